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

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/pbr_hip.h"
#include "ct_backward.hpp"
#include "ct_blend.hpp"
#include "ct_kernel.hpp"

namespace pbr {

// Measured A/B on MI355X, 4096x4096 point/metallic (tools/tune.py, DESIGN.md "Schedule experiments"):
// nt hint on: -5 % time; one-wave workgroups: -2 % vs 256 lanes (no LDS/barrier, so nothing is lost).
static int g_nontemporal = 1;
static int g_block_log2 = 6;       // workgroup size: 64 (6), 128 (7) or 256 (8) lanes
static int g_f16_vec = 8;          // pixels per lane for fp16 maps with one light: 8 (16-byte loads) or 4
// Dynamic LDS per one-wave workgroup, unused by the kernel: an occupancy governor finer than whole waves per
// SIMD (160 KiB / value = waves per CU).  amdgpu_waves_per_eu(3,3) on the kernel allows 12 waves per CU; the
// fp32 one-light kernels stream fastest with 11 in flight (in-process A/B, DESIGN.md 3.2: 115.3 vs 118.5 us on
// 4096^2, 33.3 vs 34.0 us on 2048^2, 249.8 vs 254.1 us on 8 x 2048^2 directional), the fp16 and multi-light
// kernels with no cap.  -1 = that rule; >= 0 = this many bytes for every launch (A/B runs).
static int g_lds_bytes = -1;
static int g_xcd_log2 = -1;        // >= 0 overrides the descriptor's schedule (A/B runs): tiles per XCD run = 1 << value
constexpr int kLdsFor11WavesPerCu = 14848;   // floor(163840 / 14848) = 11

static inline void normalize_host(const float v[3], float o[3]) {   // F.normalize(v, dim=0), fp32
    const float nrm = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    const float d = nrm > 1e-12f ? nrm : 1e-12f;
    o[0] = v[0] / d; o[1] = v[1] / d; o[2] = v[2] / d;
}

static int validate(const pbr_render_desc *d) {
    if (!d) return PBR_ERR_NULL_MAP;
    if (d->abi_version != PBR_HIP_ABI_VERSION) return PBR_ERR_SHAPE;
    if (d->light_type != PBR_LIGHT_DIRECTIONAL && d->light_type != PBR_LIGHT_POINT) return PBR_ERR_LIGHT_TYPE;
    if (d->workflow < 0 || d->workflow > PBR_WORKFLOW_CONVERTED) return PBR_ERR_WORKFLOW;
    if (d->workflow == PBR_WORKFLOW_SPECULAR ? !d->specular.data : !d->metallic.data) return PBR_ERR_WORKFLOW;
    if (!d->albedo.data || !d->roughness.data || !d->out) return PBR_ERR_NULL_MAP;
    if (d->batch < 1 || d->height < 1 || d->width < 1 || d->y_offset < 0 ||
        d->height_total < d->y_offset + d->height || d->n_lights < 1 || d->n_lights > PBR_MAX_LIGHTS)
        return PBR_ERR_SHAPE;
    if ((int64_t)d->batch * d->height > INT32_MAX) return PBR_ERR_SHAPE;
    if ((d->map_dtype != PBR_F32 && d->map_dtype != PBR_F16) || (d->out_dtype != PBR_F32 && d->out_dtype != PBR_F16))
        return PBR_ERR_DTYPE;
    if (d->map_dtype == PBR_F32 && d->out_dtype == PBR_F16) return PBR_ERR_DTYPE;   // not built
    if (d->schedule < PBR_SCHEDULE_AUTO || d->schedule > PBR_SCHEDULE_XCD(12)) return PBR_ERR_SHAPE;
    if (d->out_batch_stride < 0 || d->out_channel_stride < 0) return PBR_ERR_SHAPE;
    if (d->out_channel_stride && d->out_channel_stride < (int64_t)d->height * d->width) return PBR_ERR_SHAPE;
    if (d->map_height || d->map_width) {             // tiled maps: whole repeats only
        if (d->map_height < 1 || d->map_width < 1 || d->height_total % d->map_height || d->width % d->map_width)
            return PBR_ERR_SHAPE;
    }
    return PBR_OK;
}

static inline bool is_tiled(const pbr_render_desc *d) {
    return d->map_height > 0 && (d->map_height != d->height_total || d->map_width != d->width);
}

// 16-byte path: every plane start and every row start must be 16-byte (fp16: 8-byte) aligned.
static int pick_vec(const pbr_render_desc *d) {
    const int esz_in = d->map_dtype == PBR_F32 ? 4 : 2, esz_out = d->out_dtype == PBR_F32 ? 4 : 2;
    if (d->width % 4) return 1;
    const bool tiled = is_tiled(d);
    if (tiled && d->map_width % 4) return 1;          // a lane's pixels must not straddle a seam
    auto ok = [&](const pbr_map &m, int esz, bool three) {
        if (!m.data) return true;
        const uintptr_t align = esz == 4 ? 15u : 7u;
        if (reinterpret_cast<uintptr_t>(m.data) & align) return false;
        if (m.batch_stride % 4) return false;
        if (three && (m.channel_stride % 4)) return false;
        return true;
    };
    if (!ok(d->albedo, esz_in, true) || !ok(d->normal, esz_in, true) || !ok(d->roughness, esz_in, false) ||
        !ok(d->metallic, esz_in, false) || !ok(d->specular, esz_in, true))
        return 1;
    if (reinterpret_cast<uintptr_t>(d->out) & (esz_out == 4 ? 15u : 7u)) return 1;
    if (d->out_batch_stride % 8 || d->out_channel_stride % 8) return 1;      // 0 (contiguous) passes
    // fp16 maps, ONE light (HBM-bound): 8 pixels per lane keep the loads 16 bytes wide.  With several
    // lights the kernel is VALU-bound and the 4-pixel body's lower register count wins.
    if (esz_in == 2 && d->width % 8 == 0 && (!tiled || d->map_width % 8 == 0) && d->n_lights == 1 && g_f16_vec == 8) {
        auto ok16 = [&](const pbr_map &m, bool three) {
            return !m.data || ((reinterpret_cast<uintptr_t>(m.data) & 15u) == 0 && m.batch_stride % 8 == 0 &&
                               (!three || m.channel_stride % 8 == 0));
        };
        if (ok16(d->albedo, true) && ok16(d->normal, true) && ok16(d->roughness, false) && ok16(d->metallic, false) &&
            ok16(d->specular, true) && (reinterpret_cast<uintptr_t>(d->out) & 15u) == 0)
            return 8;
    }
    return 4;
}

// Workgroup -> tile order (ct_kernel.hpp: tile_of_workgroup).  Workgroups are dealt to the 8 XCDs round-robin, so
// with the linear order XCD x touches byte offsets ~ x KiB (mod 8 KiB) of every plane, all XCDs inside one narrow
// window; with runs of 64 tiles every XCD streams 64 KiB-contiguous pieces.  Measured on MI355X (tools/tune.py,
// "xcd" knob): the run order gives 6.1-6.3 TB/s whatever the shape; the linear order gives
// 6.4-6.6 TB/s when the plane streams happen to spread over the HBM channels (1024^2, 4096^2, 3072^2, ...) and
// 5.4-5.8 TB/s when they do not: rows that are not a whole number of tiles (1000^2, 3000^2: -12..14 %) and
// 8 / 16 MiB plane strides (2048^2, 4096x1024: -2..12 %).  AUTO encodes exactly that; pbr_cook_torrance_autotune
// measures instead of guessing.
static int schedule_xcd_log2(const pbr_render_desc *d, int vec) {
    if (g_xcd_log2 >= 0) return g_xcd_log2 > 12 ? 12 : g_xcd_log2;
    if (d->schedule >= PBR_SCHEDULE_LINEAR) return d->schedule - PBR_SCHEDULE_LINEAR;
    const int64_t esz = d->map_dtype == PBR_F32 ? 4 : 2;
    const int64_t row_bytes = (int64_t)d->width * esz, tile_bytes = 64 * (int64_t)vec * esz;
    const int64_t plane_bytes = d->albedo.channel_stride * esz;
    if (d->map_dtype == PBR_F16) return 6;           // fp16 maps: runs are 1.5-5 % ahead on every shape tried
    if (row_bytes % tile_bytes) return 6;
    if (plane_bytes == (8ll << 20) || plane_bytes == (16ll << 20)) return 6;
    return 0;
}

static void fill_args(const pbr_render_desc *d, int vec, KArgs &k) {
    std::memset(&k, 0, sizeof(k));
    k.albedo = d->albedo.data; k.normal = d->normal.data; k.rough = d->roughness.data;
    k.metal = d->metallic.data; k.spec = d->specular.data; k.out = d->out;
    k.a_bs = d->albedo.batch_stride; k.a_cs = d->albedo.channel_stride;
    k.n_bs = d->normal.batch_stride; k.n_cs = d->normal.channel_stride;
    k.r_bs = d->roughness.batch_stride; k.m_bs = d->metallic.batch_stride;
    k.s_bs = d->specular.batch_stride; k.s_cs = d->specular.channel_stride;
    k.o_cs = d->out_channel_stride ? d->out_channel_stride : (int64_t)d->height * d->width;
    k.o_bs = d->out_batch_stride ? d->out_batch_stride : 3 * k.o_cs;
    k.rows = d->batch * d->height; k.H = d->height; k.W = d->width;
    k.wv = d->width / vec;
    k.bt_log2 = g_block_log2 < 6 ? 6 : (g_block_log2 > 8 ? 8 : g_block_log2);
    int lg = 0;
    while ((1 << lg) < k.wv && lg < k.bt_log2) ++lg;
    k.bx_log2 = lg;
    const int bx = 1 << lg, by = (1 << k.bt_log2) >> lg;
    k.tiles_x = (k.wv + bx - 1) / bx;
    const int64_t tiles = (int64_t)k.tiles_x * ((k.rows + by - 1) / by);
    k.n_tiles = tiles > INT32_MAX ? -1 : (int32_t)tiles;      // -1: more tiles than a 1-D grid holds, rejected by the callers
    k.xcd_log2 = schedule_xcd_log2(d, vec);
    k.xcd_tiles = k.n_tiles < 0 ? 0 : (k.n_tiles >> (k.xcd_log2 + 3)) << (k.xcd_log2 + 3);
    k.div_h.init((uint32_t)d->height);
    k.div_tx.init((uint32_t)k.tiles_x);
    k.tiled = is_tiled(d);
    k.map_h = k.tiled ? d->map_height : d->height_total; k.map_w = k.tiled ? d->map_width : d->width;
    k.div_mh.init((uint32_t)k.map_h); k.div_mw.init((uint32_t)k.map_w);
    k.y_offset = d->y_offset; k.H_total = d->height_total;
    // `light_size or 1.0` (:130): 0 / NaN / negative are treated as "not given".
    const float size = (d->light_size > 0.0f) ? d->light_size : 1.0f;
    const float lo = (float)(-(double)size / 2), hi = (float)((double)size / 2);
    k.x0 = lo; k.x1 = hi; k.xstep = d->width > 1 ? (hi - lo) / (float)(d->width - 1) : 0.0f;
    k.y0 = lo; k.y1 = hi; k.ystep = d->height_total > 1 ? (hi - lo) / (float)(d->height_total - 1) : 0.0f;
    if (d->width == 1) k.x1 = k.x0;          // torch.linspace(a, b, 1) == [a]
    if (d->height_total == 1) k.y1 = k.y0;
    normalize_host(d->view_dir, k.V);
    k.n_lights = d->n_lights;
    k.albedo_srgb = d->albedo_is_srgb != 0; k.spec_srgb = d->specular_is_srgb != 0;
    k.out_srgb = d->return_srgb != 0; k.has_normal = d->normal.data != nullptr;
    for (int i = 0; i < d->n_lights; ++i) {
        LightU &u = k.lights[i];
        for (int c = 0; c < 3; ++c) u.inten[c] = d->intensities[i][c];
        if (d->light_type == PBR_LIGHT_DIRECTIONAL) {
            normalize_host(d->lights[i], u.l);                                   // :126
            float hn[3];
            for (int c = 0; c < 3; ++c) u.h[c] = k.V[c] + u.l[c];                // :155
            const float hh = u.h[0] * u.h[0] + u.h[1] * u.h[1] + u.h[2] * u.h[2];
            u.rhh = 1.0f / (hh > 1e-24f ? hh : 1e-24f);
            normalize_host(u.h, hn);
            float ct = hn[0] * k.V[0] + hn[1] * k.V[1] + hn[2] * k.V[2];          // :156-158
            ct = ct < 0.0f ? 0.0f : (ct > 1.0f ? 1.0f : ct);
            const float om = 1.0f - ct;
            u.p5 = (om * om) * (om * om) * om;                                   // :196
        } else {
            for (int c = 0; c < 3; ++c) u.l[c] = d->lights[i][c];
        }
    }
}

using KernelFn = void (*)(const KArgs);
struct KernelEntry { KernelFn fn; const char *name; };

// Storage-type pairs built: (f32 -> f32), (f16 -> f32), (f16 -> f16).
template <int LIGHT, int WF, typename TI, typename TO>
static KernelFn pick_variant(int vec, bool multi, bool nt) {
    if constexpr (sizeof(TI) == 2) {
        if (vec == 8) {
            if (multi) return nt ? cook_torrance_kernel<LIGHT, WF, TI, TO, 8, true, true>
                                 : cook_torrance_kernel<LIGHT, WF, TI, TO, 8, true, false>;
            return nt ? cook_torrance_kernel<LIGHT, WF, TI, TO, 8, false, true>
                      : cook_torrance_kernel<LIGHT, WF, TI, TO, 8, false, false>;
        }
    }
    if (vec == 1) return multi ? cook_torrance_kernel<LIGHT, WF, TI, TO, 1, true, false>
                               : cook_torrance_kernel<LIGHT, WF, TI, TO, 1, false, false>;
    if (multi) return nt ? cook_torrance_kernel<LIGHT, WF, TI, TO, 4, true, true>
                         : cook_torrance_kernel<LIGHT, WF, TI, TO, 4, true, false>;
    return nt ? cook_torrance_kernel<LIGHT, WF, TI, TO, 4, false, true>
              : cook_torrance_kernel<LIGHT, WF, TI, TO, 4, false, false>;
}

template <int LIGHT, int WF>
static KernelFn pick_types(int in_dt, int out_dt, int vec, bool multi, bool nt) {
    if (in_dt == PBR_F32) return pick_variant<LIGHT, WF, float, float>(vec, multi, nt);
    if (out_dt == PBR_F32) return pick_variant<LIGHT, WF, __half, float>(vec, multi, nt);
    return pick_variant<LIGHT, WF, __half, __half>(vec, multi, nt);
}

static KernelEntry pick_kernel(const pbr_render_desc *d, int vec, bool nt) {
    static thread_local char name[96];
    static const char *const wf_names[3] = {"metallic", "specular", "converted"};
    const bool multi = d->n_lights > 1, point = d->light_type == PBR_LIGHT_POINT;
    const int idt = d->map_dtype, odt = d->out_dtype;
    std::snprintf(name, sizeof(name), "ct_%s_%s_%s_%s_v%d%s", point ? "point" : "directional", wf_names[d->workflow],
                  idt == PBR_F32 ? "f32" : "f16", odt == PBR_F32 ? "f32" : "f16", vec, multi ? "_multi" : "");
    KernelFn fn = nullptr;
    switch ((point ? 3 : 0) + d->workflow) {
        case 0: fn = pick_types<PBR_LIGHT_DIRECTIONAL, PBR_WORKFLOW_METALLIC>(idt, odt, vec, multi, nt); break;
        case 1: fn = pick_types<PBR_LIGHT_DIRECTIONAL, PBR_WORKFLOW_SPECULAR>(idt, odt, vec, multi, nt); break;
        case 2: fn = pick_types<PBR_LIGHT_DIRECTIONAL, PBR_WORKFLOW_CONVERTED>(idt, odt, vec, multi, nt); break;
        case 3: fn = pick_types<PBR_LIGHT_POINT, PBR_WORKFLOW_METALLIC>(idt, odt, vec, multi, nt); break;
        case 4: fn = pick_types<PBR_LIGHT_POINT, PBR_WORKFLOW_SPECULAR>(idt, odt, vec, multi, nt); break;
        default: fn = pick_types<PBR_LIGHT_POINT, PBR_WORKFLOW_CONVERTED>(idt, odt, vec, multi, nt); break;
    }
    return KernelEntry{fn, name};
}

}  // namespace pbr

extern "C" {

int pbr_cook_torrance(const pbr_render_desc *d, void *stream) {
    using namespace pbr;
    const int rc = validate(d);
    if (rc != PBR_OK) return rc;
    const int vec = pick_vec(d);
    KArgs k;
    fill_args(d, vec, k);
    if (k.n_tiles < 0) return PBR_ERR_SHAPE;
    // Tiled maps are re-read from L2 / Infinity Cache, so their loads must not carry the streaming hint, and the
    // launch is then VALU-bound and wants every wave it can get (2048^2 tile(2): 81 us vs 122 us with the streaming
    // settings, 120 us for the materialised 4096^2 maps; tools/tile_probe.py).
    const bool tiled = k.tiled != 0;
    const KernelEntry e = pick_kernel(d, vec, g_nontemporal != 0 && !tiled);
    // 1-D grid, one tile per workgroup, x fastest: consecutive workgroups touch consecutive runs of every plane
    const bool fp32_one_light = d->map_dtype == PBR_F32 && d->n_lights == 1 && k.bt_log2 == 6 && !tiled;
    const size_t lds = g_lds_bytes >= 0 ? (size_t)g_lds_bytes : (fp32_one_light ? kLdsFor11WavesPerCu : 0);
    hipLaunchKernelGGL(e.fn, dim3((unsigned)k.n_tiles, 1, 1), dim3(1u << k.bt_log2, 1, 1), lds,
                       static_cast<hipStream_t>(stream), k);
    const hipError_t err = hipGetLastError();
    return err == hipSuccess ? PBR_OK : 1000 + (int)err;
}

int pbr_cook_torrance_blend(const pbr_render_desc *d, const pbr_blend_desc *bl, void *workspace, void *stream) {
    using namespace pbr;
    const int rc = validate(d);
    if (rc != PBR_OK) return rc;
    if (!bl || !workspace) return PBR_ERR_NULL_MAP;
    if (d->map_dtype != PBR_F32 || d->out_dtype != PBR_F32) return PBR_ERR_DTYPE;
    if (!d->normal.data || !bl->albedo.data || !bl->normal.data || !bl->roughness.data || !bl->mask.data)
        return PBR_ERR_NULL_MAP;
    if (d->workflow == PBR_WORKFLOW_SPECULAR ? !bl->specular.data : !bl->metallic.data) return PBR_ERR_WORKFLOW;
    int vec = pick_vec(d);
    for (const pbr_map *m : {&bl->albedo, &bl->normal, &bl->roughness, &bl->metallic, &bl->specular, &bl->mask})
        if (m->data && ((reinterpret_cast<uintptr_t>(m->data) & 15u) || m->batch_stride % 4 || m->channel_stride % 4)) vec = 1;
    KArgs k;
    fill_args(d, vec, k);
    if (k.n_tiles < 0) return PBR_ERR_SHAPE;
    KBlend b;
    std::memset(&b, 0, sizeof(b));
    b.albedo = bl->albedo.data; b.normal = bl->normal.data; b.rough = bl->roughness.data;
    b.metal = bl->metallic.data; b.spec = bl->specular.data;
    b.a_bs = bl->albedo.batch_stride; b.a_cs = bl->albedo.channel_stride;
    b.n_bs = bl->normal.batch_stride; b.n_cs = bl->normal.channel_stride;
    b.r_bs = bl->roughness.batch_stride; b.m_bs = bl->metallic.batch_stride;
    b.s_bs = bl->specular.batch_stride; b.s_cs = bl->specular.channel_stride;
    b.mask = static_cast<const float *>(bl->mask.data); b.k_bs = bl->mask.batch_stride;
    b.normal_signed = static_cast<const int *>(workspace);
    hipStream_t st = static_cast<hipStream_t>(stream);
    // pass 1: one flag per material -- does the blended normal map have a negative component?  (base.py:212)
    if (hipMemsetAsync(workspace, 0, sizeof(int) * (size_t)d->batch, st) != hipSuccess) return 1000 + (int)hipGetLastError();
    const int64_t P = (int64_t)k.map_h * k.map_w, total = P * d->batch;
    const int64_t blocks = (total + 255) / 256;
    hipLaunchKernelGGL(blend_normal_sign_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, st,
                       static_cast<const float *>(d->normal.data), static_cast<const float *>(bl->normal.data), b.mask,
                       k.n_bs, k.n_cs, b.n_bs, b.n_cs, b.k_bs, P, total, static_cast<int *>(workspace));
    // pass 2: blend + evaluate
    const bool multi = d->n_lights > 1, point = d->light_type == PBR_LIGHT_POINT;
    void (*fn)(const KArgs, const KBlend) = nullptr;
#define PBR_BLEND(L, W)                                                                                              \
    fn = vec == 4 ? (multi ? cook_torrance_blend_kernel<L, W, 4, true> : cook_torrance_blend_kernel<L, W, 4, false>) \
                  : (multi ? cook_torrance_blend_kernel<L, W, 1, true> : cook_torrance_blend_kernel<L, W, 1, false>)
    switch ((point ? 3 : 0) + d->workflow) {
        case 0: PBR_BLEND(PBR_LIGHT_DIRECTIONAL, PBR_WORKFLOW_METALLIC); break;
        case 1: PBR_BLEND(PBR_LIGHT_DIRECTIONAL, PBR_WORKFLOW_SPECULAR); break;
        case 2: PBR_BLEND(PBR_LIGHT_DIRECTIONAL, PBR_WORKFLOW_CONVERTED); break;
        case 3: PBR_BLEND(PBR_LIGHT_POINT, PBR_WORKFLOW_METALLIC); break;
        case 4: PBR_BLEND(PBR_LIGHT_POINT, PBR_WORKFLOW_SPECULAR); break;
        default: PBR_BLEND(PBR_LIGHT_POINT, PBR_WORKFLOW_CONVERTED); break;
    }
#undef PBR_BLEND
    // occupancy governor (see g_lds_bytes): the 17-stream one-light blend streams fastest with 10 waves per CU --
    // 4096^2: 255 us uncapped, 236 / 232 / 234 / 233 / 248 us at 11 / 10 / 9 / 8 / 6 (tools/blend_probe.py)
    const size_t lds = g_lds_bytes >= 0 ? (size_t)g_lds_bytes : (multi ? 0 : 16384);
    hipLaunchKernelGGL(fn, dim3((unsigned)k.n_tiles, 1, 1), dim3(1u << k.bt_log2, 1, 1), lds, st, k, b);
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

int pbr_cook_torrance_backward(const pbr_render_desc *d, const void *grad_out, void *g_albedo, void *g_normal,
                               void *g_roughness, void *g_metallic, void *g_specular, void *stream) {
    using namespace pbr;
    const int rc = validate(d);
    if (rc != PBR_OK) return rc;
    if (!grad_out) return PBR_ERR_NULL_MAP;
    if (d->out_dtype != PBR_F32) return PBR_ERR_DTYPE;        // the upstream gradient is fp32; maps (and their gradients) fp32 | fp16
    const bool half_maps = d->map_dtype == PBR_F16;
    int vec = pick_vec(d);
    for (const void *g : {grad_out, (const void *)g_albedo, (const void *)g_normal, (const void *)g_roughness,
                          (const void *)g_metallic, (const void *)g_specular})
        if (g && (reinterpret_cast<uintptr_t>(g) & 15u)) vec = 1;
    if (vec == 8) vec = 4;
    KArgs k;
    fill_args(d, vec, k);
    if (k.n_tiles < 0) return PBR_ERR_SHAPE;
    k.o_cs = (int64_t)d->height * d->width; k.o_bs = 3 * k.o_cs;     // grad_out and the g_* are contiguous, whatever `out` was
    const BArgs b = {grad_out, g_albedo, g_normal, g_roughness, g_metallic, g_specular};
    const bool multi = d->n_lights > 1, point = d->light_type == PBR_LIGHT_POINT;
    void (*fn)(const KArgs, const BArgs) = nullptr;
#define PBR_BWD_T(L, W, T)                                                                                            \
    (vec == 4 ? (multi ? cook_torrance_backward_kernel<L, W, 4, true, T> : cook_torrance_backward_kernel<L, W, 4, false, T>) \
              : (multi ? cook_torrance_backward_kernel<L, W, 1, true, T> : cook_torrance_backward_kernel<L, W, 1, false, T>))
#define PBR_BWD(L, W) fn = half_maps ? PBR_BWD_T(L, W, __half) : PBR_BWD_T(L, W, float)
    switch ((point ? 3 : 0) + d->workflow) {
        case 0: PBR_BWD(PBR_LIGHT_DIRECTIONAL, PBR_WORKFLOW_METALLIC); break;
        case 1: PBR_BWD(PBR_LIGHT_DIRECTIONAL, PBR_WORKFLOW_SPECULAR); break;
        case 2: PBR_BWD(PBR_LIGHT_DIRECTIONAL, PBR_WORKFLOW_CONVERTED); break;
        case 3: PBR_BWD(PBR_LIGHT_POINT, PBR_WORKFLOW_METALLIC); break;
        case 4: PBR_BWD(PBR_LIGHT_POINT, PBR_WORKFLOW_SPECULAR); break;
        default: PBR_BWD(PBR_LIGHT_POINT, PBR_WORKFLOW_CONVERTED); break;
    }
#undef PBR_BWD
#undef PBR_BWD_T
    hipLaunchKernelGGL(fn, dim3((unsigned)k.n_tiles, 1, 1), dim3(1u << k.bt_log2, 1, 1), 0, static_cast<hipStream_t>(stream), k, b);
    const hipError_t err = hipGetLastError();
    return err == hipSuccess ? PBR_OK : 1000 + (int)err;
}

const char *pbr_kernel_name(const pbr_render_desc *d) {
    if (pbr::validate(d) != PBR_OK) return nullptr;
    return pbr::pick_kernel(d, pbr::pick_vec(d), true).name;
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
    int *slot = nullptr;
    switch (knob) {
        case PBR_TUNE_NONTEMPORAL: slot = &pbr::g_nontemporal; break;
        case PBR_TUNE_BLOCK_LOG2: slot = &pbr::g_block_log2; break;
        case PBR_TUNE_F16_VEC: slot = &pbr::g_f16_vec; break;
        case PBR_TUNE_LDS_BYTES: slot = &pbr::g_lds_bytes; break;
        case PBR_TUNE_XCD_LOG2: slot = &pbr::g_xcd_log2; break;
        default: return -1;
    }
    const int old = *slot;
    *slot = value;
    return old;
}

}  // extern "C"
