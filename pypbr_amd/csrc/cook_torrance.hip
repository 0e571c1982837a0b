// cook_torrance.hip -- launcher of the fused Cook-Torrance kernels (C ABI: pbr_cook_torrance).
//
// One launch replaces the ~130 whole-map ATen kernels of the reference's
// CookTorranceBRDF.forward (/root/reference/pypbr/models/cooktorrance.py:92-182): sRGB
// decode of albedo/specular, workflow switch (or the in-kernel metallic -> diffuse/specular
// conversion), light geometry, normal normalisation, Fresnel/GGX/Smith, compositing, clamp
// and sRGB encode, for a [B,C,H,W] planar batch and up to PBR_MAX_LIGHTS lights.
//
// Roofline: HBM-bound streaming kernel, zero reuse: 32 B read + 12 B written per pixel
// (metallic workflow, fp32).  Design for that:
//   * each lane owns 4 consecutive pixels of a row: one 16-byte load per input plane
//     (8 x global_load_dwordx4 in flight per lane, 8 KiB per wave) and one 16-byte store
//     per output plane; a wave covers 1 KiB-contiguous segments of every plane;
//   * loads/stores carry the non-temporal hint (each byte is touched once; 4K maps are
//     64 MiB per plane, nothing fits L2/MALL);
//   * view / light / intensity / grid parameters live in the kernel-argument segment and
//     are read with scalar loads: they sit in SGPRs, which IS the wave-wide broadcast on
//     CDNA (a cross-lane shuffle would cost VALU/LDS slots for something the scalar unit
//     gives for free).  Directional-light terms (L, V+L, Fresnel power) are pixel
//     independent and are folded on the host; point-light row terms are shared by the 4
//     pixels of a lane;
//   * no LDS: there is no inter-pixel reuse to stage ("LDS staging").
#include <hip/hip_runtime.h>

#include <algorithm>

#include "ct_launch.hpp"

namespace pbr {

// The process-wide values of the schedule knobs (tuning.hpp), initialised to the rules.  Indexed by PBR_TUNE_*.
static constexpr int kKnobRules[PBR_TUNE_COUNT] = {
    /* (retired: NONTEMPORAL) */ 1, /* BLOCK_LOG2 */ 0, /* F16_VEC */ 8, /* LDS_BYTES */ -1, /* BWD_VEC */ 0, /* BATCH_INNER */ -1,
    /* SCALAR_BASE */ 1, /* MAX_VEC */ 8, /* BWD_RUN */ -1, /* MSE_STREAM */ 1, /* TILE_REPEAT */ -1, /* RESIZE_UP2 */ 1};
std::atomic<int> g_knobs[PBR_TUNE_COUNT] = {
    {kKnobRules[0]}, {kKnobRules[1]}, {kKnobRules[2]}, {kKnobRules[3]}, {kKnobRules[4]}, {kKnobRules[5]}, {kKnobRules[6]}, {kKnobRules[7]},
    {kKnobRules[8]}, {kKnobRules[9]}, {kKnobRules[10]}, {kKnobRules[11]}};
thread_local const pbr_tuning *t_tuning = nullptr;

struct KernelEntry { KernelFn fn; const char *name; };

// Storage-type pairs built: (f32 -> f32), (f32 -> f16), (f16 -> f32), (f16 -> f16).
// The streaming hint is a RULE since ABI 8 (on for the 4- and 8-pixel lanes, off for the one-pixel kernels), and one light over fp32 maps
// always takes the plain-fp32 body: the instantiations that only the closed experiments' knobs reached (no hint on vector lanes; the
// packed one-light body PACK1 for the wrap-around form of tiled launches, which the repeat-inner kernels have replaced) are not built.
template <int LIGHT, int WF, typename TI, typename TO>
static KernelFn pick_variant(int vec, bool multi) {
    if constexpr (sizeof(TI) == 2) {
        if (vec == 8 && !multi) {        // pick_vec hands out 8-pixel lanes for ONE light only (several lights are VALU-bound: 4-pixel lanes);
                                         // the 8-pixel multi-light body would not fit 128 VGPRs (it spilled 220-304 bytes when it was instantiated)
            return cook_torrance_kernel<LIGHT, WF, TI, TO, 8, false, true>;
        }
    }
    if (vec == 1) return multi ? cook_torrance_kernel<LIGHT, WF, TI, TO, 1, true, false>
                               : cook_torrance_kernel<LIGHT, WF, TI, TO, 1, false, false>;
    return multi ? cook_torrance_kernel<LIGHT, WF, TI, TO, 4, true, true> : cook_torrance_kernel<LIGHT, WF, TI, TO, 4, false, true>;
}

template <int LIGHT, int WF>
static KernelFn pick_types(int in_dt, int out_dt, int vec, bool multi) {
    if (in_dt == PBR_F32) return out_dt == PBR_F32 ? pick_variant<LIGHT, WF, float, float>(vec, multi)
                                                   : pick_variant<LIGHT, WF, float, __half>(vec, multi);
    if (out_dt == PBR_F32) return pick_variant<LIGHT, WF, __half, float>(vec, multi);
    return pick_variant<LIGHT, WF, __half, __half>(vec, multi);
}

static KernelEntry pick_kernel(const pbr_render_desc *d, int vec) {
    static thread_local char name[96];
    static const char *const wf_names[3] = {"metallic", "specular", "converted"};
    const bool multi = d->n_lights > 1, point = d->light_type == PBR_LIGHT_POINT;
    const int idt = d->map_dtype, odt = d->out_dtype;
    const int nb = batch_group(d, vec);
    if (repeat_inner(d)) {
        std::snprintf(name, sizeof(name), "ctr_%s_%s_%s_%s_v4%s", point ? "point" : "directional", wf_names[d->workflow],
                      idt == PBR_F32 ? "f32" : "f16", odt == PBR_F32 ? "f32" : "f16", multi ? "_multi" : "");
        return KernelEntry{pick_repeat_kernel(d), name};
    }
    if (nb) {
        std::snprintf(name, sizeof(name), "ctb_%s_%s_%s_%s_v2_b%d", point ? "point" : "directional", wf_names[d->workflow],
                      idt == PBR_F32 ? "f32" : "f16", odt == PBR_F32 ? "f32" : "f16", nb);
        return KernelEntry{pick_batch_kernel(d, nb), name};
    }
    std::snprintf(name, sizeof(name), "ct_%s_%s_%s_%s_v%d%s", point ? "point" : "directional", wf_names[d->workflow],
                  idt == PBR_F32 ? "f32" : "f16", odt == PBR_F32 ? "f32" : "f16", vec, multi ? "_multi" : "");
    KernelFn fn = nullptr;
    switch ((point ? 3 : 0) + d->workflow) {
        case 0: fn = pick_types<PBR_LIGHT_DIRECTIONAL, PBR_WORKFLOW_METALLIC>(idt, odt, vec, multi); break;
        case 1: fn = pick_types<PBR_LIGHT_DIRECTIONAL, PBR_WORKFLOW_SPECULAR>(idt, odt, vec, multi); break;
        case 2: fn = pick_types<PBR_LIGHT_DIRECTIONAL, PBR_WORKFLOW_CONVERTED>(idt, odt, vec, multi); break;
        case 3: fn = pick_types<PBR_LIGHT_POINT, PBR_WORKFLOW_METALLIC>(idt, odt, vec, multi); break;
        case 4: fn = pick_types<PBR_LIGHT_POINT, PBR_WORKFLOW_SPECULAR>(idt, odt, vec, multi); break;
        default: fn = pick_types<PBR_LIGHT_POINT, PBR_WORKFLOW_CONVERTED>(idt, odt, vec, multi); break;
    }
    return KernelEntry{fn, name};
}

}  // namespace pbr

namespace pbr {
// pbr_prepare_device_params: one wave; lane i folds light i (fold_light: the host's own code), lane 0 the view vector and the grey flag.
// A parameter given as a NULL device pointer comes from the descriptor (host values, carried in the kernel arguments).
struct HostParams { float view[3]; float lights[PBR_MAX_LIGHTS][3]; float inten[PBR_MAX_LIGHTS][3]; };
__global__ __launch_bounds__(64) void prepare_device_params_kernel(const float *__restrict__ view, const float *__restrict__ lights,
                                                                   const float *__restrict__ inten, int intensity_rows, int n_lights, int light_type,
                                                                   const HostParams hp, DevParams *__restrict__ out) {
    const int i = threadIdx.x;
    float raw[3];
    for (int c = 0; c < 3; ++c) raw[c] = view ? view[c] : hp.view[c];
    float V[3];
    normalize_host(raw, V);
    auto intensity = [&](int l, int c) { return inten ? inten[3 * (intensity_rows == 1 ? 0 : l) + c] : hp.inten[l][c]; };
    if (i == 0) {
        int grey = 1;
        for (int l = 0; l < n_lights; ++l)
            if (intensity(l, 0) != intensity(l, 1) || intensity(l, 0) != intensity(l, 2)) grey = 0;
        for (int c = 0; c < 3; ++c) { out->V[c] = V[c]; out->raw_view[c] = raw[c]; }
        out->grey = grey;
    }
    if (i < n_lights) {
        float L[3], I[3];
        for (int c = 0; c < 3; ++c) { L[c] = lights ? lights[3 * i + c] : hp.lights[i][c]; I[c] = intensity(i, c); }
        LightU u;
        fold_light(light_type, V, L, I, u);
        out->lights[i] = u;
        for (int c = 0; c < 3; ++c) out->raw_lights[i][c] = L[c];
    }
}
}  // namespace pbr

extern "C" {

int pbr_cook_torrance(const pbr_render_desc *d, void *stream) {
    using namespace pbr;
    const TuningScope tuning(d);
    const int rc = validate(d);
    if (rc != PBR_OK) return rc;
    if (nan_light_size(d)) return fill_result_nan(d, static_cast<hipStream_t>(stream));
    const int vec = pick_vec(d);
    KArgs k;
    if (repeat_inner(d)) {           // tile(n), whole output: texels loaded and decoded once, evaluated at every repeat (ct_tiled.hip)
        fill_repeat_args(d, k);
        if (k.n_tiles < 0) return PBR_ERR_SHAPE;
        const KernelEntry e = pick_kernel(d, 4);
        hipLaunchKernelGGL(e.fn, dim3((unsigned)k.n_tiles, 1, 1), dim3(1u << k.bt_log2, 1, 1), g_lds_bytes >= 0 ? (size_t)g_lds_bytes : 0,
                           static_cast<hipStream_t>(stream), k);
        const hipError_t err = hipGetLastError();
        return err == hipSuccess ? PBR_OK : 1000 + (int)err;
    }
    const int nb = batch_group(d, vec);
    if (nb) {                        // lane_pos' material index is the group of nb consecutive materials; 2 pixels per lane
        pbr_render_desc g = *d;
        g.batch = d->batch / nb;
        fill_args(&g, 2, k);
        if (g_scalar_base != 2) k.sbase = 0;      // a group of materials is a batch (ct_launch.hpp: the rule is single materials)
    } else {
        fill_args(d, vec, k);
    }
    if (k.n_tiles < 0) return PBR_ERR_SHAPE;
    // (tiled maps in the wrap-around form -- map rows shorter than 4 texels, or the tests' PBR_TUNE_TILE_REPEAT = 0 -- are re-read from L2 /
    // Infinity Cache; the one-pixel kernels carry no streaming hint, the tests' 4-pixel form does: speed only, never values)
    const bool tiled = k.tiled != 0;
    const KernelEntry e = pick_kernel(d, vec);
    // 1-D grid, one tile per workgroup, x fastest: consecutive workgroups touch consecutive runs of every plane
    const bool fp32_one_light = d->map_dtype == PBR_F32 && d->n_lights == 1 && k.bt_log2 == 6 && !tiled;
    size_t lds = g_lds_bytes >= 0 ? (size_t)g_lds_bytes : (fp32_one_light ? kLdsFor11WavesPerCu : 0);
    if (k.xpose) lds = std::max(lds, (size_t)kXposeLdsPerWave << (k.bt_log2 - 6));     // the kernel's piece exchange needs this much
    hipLaunchKernelGGL(e.fn, dim3((unsigned)k.n_tiles, 1, 1), dim3(1u << k.bt_log2, 1, 1), lds,
                       static_cast<hipStream_t>(stream), k);
    const hipError_t err = hipGetLastError();
    return err == hipSuccess ? PBR_OK : 1000 + (int)err;
}

int pbr_cook_torrance_autotune(const pbr_render_desc *d, void *stream, int32_t *schedule) {
    using namespace pbr;
    const int rc = validate(d);
    if (rc != PBR_OK) return rc;
    if (!schedule) return PBR_ERR_NULL_MAP;
    const int32_t cands[2] = {PBR_SCHEDULE_LINEAR, PBR_SCHEDULE_XCD(6)};
    float best[2] = {3.4e38f, 3.4e38f};
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return 1000 + (int)hipGetLastError();
    pbr_render_desc t = *d;
    int err = PBR_OK;
    const int reps = 4;
    for (int round = 0; round < 3 && err == PBR_OK; ++round) {          // interleaved rounds, best of each candidate
        for (int c = 0; c < 2 && err == PBR_OK; ++c) {
            t.schedule = cands[c];
            if ((err = pbr_cook_torrance(&t, stream)) != PBR_OK) break;   // warm
            hipError_t he = hipEventRecord(e0, st);
            for (int i = 0; i < reps && err == PBR_OK; ++i) err = pbr_cook_torrance(&t, stream);
            if (he == hipSuccess) he = hipEventRecord(e1, st);
            if (he == hipSuccess) he = hipEventSynchronize(e1);
            float ms = 0.0f;
            if (he == hipSuccess) he = hipEventElapsedTime(&ms, e0, e1);
            if (he != hipSuccess) { err = 1000 + (int)he; break; }
            if (ms < best[c]) best[c] = ms;
        }
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (err != PBR_OK) return err;
    *schedule = best[1] < best[0] ? cands[1] : cands[0];
    return PBR_OK;
}

const char *pbr_kernel_name(const pbr_render_desc *d) {
    const pbr::TuningScope tuning(d);
    if (pbr::validate(d) != PBR_OK) return nullptr;
    return pbr::pick_kernel(d, pbr::pick_vec(d)).name;
}

int pbr_bytes_per_pixel(const pbr_render_desc *d) {
    if (pbr::validate(d) != PBR_OK) return 0;
    const int ein = d->map_dtype == PBR_F32 ? 4 : 2, eout = d->out_dtype == PBR_F32 ? 4 : 2;
    int ch = 3 + 1;                                        // albedo + roughness
    if (d->normal.data) ch += 3;
    ch += d->workflow == PBR_WORKFLOW_SPECULAR ? 3 : 1;    // specular | metallic
    if (pbr::is_tiled(d)) {                                // every texel is needed from HBM once, whatever the repeat count
        const int64_t reps = ((int64_t)d->height_total / d->map_height) * ((int64_t)d->width / d->map_width);
        return (int)((ch * ein + reps - 1) / reps) + 3 * eout;
    }
    return ch * ein + 3 * eout;
}

int pbr_set_tuning(int knob, int value) {
    if (knob < 0 || knob >= PBR_TUNE_COUNT) return -1;
    return pbr::g_knobs[knob].exchange(value == PBR_TUNE_UNSET ? pbr::kKnobRules[knob] : value, std::memory_order_relaxed);
}

void pbr_tuning_init(pbr_tuning *t) {
    if (!t) return;
    for (int i = 0; i < PBR_TUNE_SLOTS; ++i) t->knob[i] = PBR_TUNE_UNSET;
}

size_t pbr_device_params_bytes(void) { return sizeof(pbr::DevParams); }

int pbr_prepare_device_params(const pbr_render_desc *d, const void *view_dir, const void *lights, const void *intensities,
                              int32_t intensity_rows, void *block, void *stream) {
    using namespace pbr;
    if (!d || !block) return PBR_ERR_NULL_MAP;
    if (d->abi_version != PBR_HIP_ABI_VERSION || d->n_lights < 1 || d->n_lights > PBR_MAX_LIGHTS) return PBR_ERR_SHAPE;
    if (d->light_type != PBR_LIGHT_POINT && d->light_type != PBR_LIGHT_DIRECTIONAL) return PBR_ERR_LIGHT_TYPE;
    if (intensities && intensity_rows != 1 && intensity_rows != d->n_lights) return PBR_ERR_SHAPE;
    HostParams hp;
    for (int c = 0; c < 3; ++c) hp.view[c] = d->view_dir[c];
    for (int i = 0; i < PBR_MAX_LIGHTS; ++i)
        for (int c = 0; c < 3; ++c) { hp.lights[i][c] = d->lights[i][c]; hp.inten[i][c] = d->intensities[i][c]; }
    hipLaunchKernelGGL(prepare_device_params_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), static_cast<const float *>(view_dir),
                       static_cast<const float *>(lights), static_cast<const float *>(intensities), (int)intensity_rows, (int)d->n_lights,
                       (int)d->light_type, hp, static_cast<DevParams *>(block));
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? PBR_OK : 1000 + (int)e;
}

}  // extern "C"
