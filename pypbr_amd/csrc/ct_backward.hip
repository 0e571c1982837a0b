// ct_backward.hip -- launchers of the backward kernels (C ABI: pbr_cook_torrance_backward, pbr_cook_torrance_backward_params);
// device code in ct_backward.hpp.
#define PBR_PARAM_GRAD_KERNELS
#include "ct_backward.hpp"
#include "ct_launch.hpp"

namespace pbr {

using BwdFn = void (*)(const KArgs, const BArgs);

// Pixels per lane are a RULE since ABI 8 (launch_backward): four for the map gradients alone -- two for fp16 maps under one light --, two with the
// light / view adjoints (PG), one for rows shorter than four pixels.  The bodies only the closed A/B knob reached (two pixels for fp32 maps
// without PG, four for fp16 maps under one light) and the one nothing reached (four pixels with PG: 256 VGPRs + 40 AGPRs) are not built.
template <int L, int W, typename T, bool PG>
static BwdFn pick_bwd_variant(int vec, bool multi) {
    if (vec == 1) return multi ? cook_torrance_backward_kernel<L, W, 1, true, T, PG> : cook_torrance_backward_kernel<L, W, 1, false, T, PG>;
    if constexpr (PG) {
        return multi ? cook_torrance_backward_kernel<L, W, 2, true, T, PG> : cook_torrance_backward_kernel<L, W, 2, false, T, PG>;
    } else if constexpr (sizeof(T) == 2) {
        if (multi) return cook_torrance_backward_kernel<L, W, 4, true, T, PG>;
        return cook_torrance_backward_kernel<L, W, 2, false, T, PG>;
    } else {
        return multi ? cook_torrance_backward_kernel<L, W, 4, true, T, PG> : cook_torrance_backward_kernel<L, W, 4, false, T, PG>;
    }
}

template <bool PG>
static BwdFn pick_bwd(const pbr_render_desc *d, int vec) {
    const bool multi = d->n_lights > 1, point = d->light_type == PBR_LIGHT_POINT, half_maps = d->map_dtype == PBR_F16;
#define PBR_BWD(L, W) return half_maps ? pick_bwd_variant<L, W, __half, PG>(vec, multi) : pick_bwd_variant<L, W, float, PG>(vec, multi)
    switch ((point ? 3 : 0) + d->workflow) {
        case 0: PBR_BWD(PBR_LIGHT_DIRECTIONAL, PBR_WORKFLOW_METALLIC);
        case 1: PBR_BWD(PBR_LIGHT_DIRECTIONAL, PBR_WORKFLOW_SPECULAR);
        case 2: PBR_BWD(PBR_LIGHT_DIRECTIONAL, PBR_WORKFLOW_CONVERTED);
        case 3: PBR_BWD(PBR_LIGHT_POINT, PBR_WORKFLOW_METALLIC);
        case 4: PBR_BWD(PBR_LIGHT_POINT, PBR_WORKFLOW_SPECULAR);
        default: PBR_BWD(PBR_LIGHT_POINT, PBR_WORKFLOW_CONVERTED);
    }
#undef PBR_BWD
}

using BwdStreamFn = void (*)(const KArgs, const BArgs, int, int);
static BwdStreamFn pick_bwd_stream(const pbr_render_desc *d, bool full) {
    const bool point = d->light_type == PBR_LIGHT_POINT;
#define PBR_BWDS(L, W) return full ? cook_torrance_backward_stream_kernel<L, W, true> : cook_torrance_backward_stream_kernel<L, W, false>
    switch ((point ? 3 : 0) + d->workflow) {
        case 0: PBR_BWDS(PBR_LIGHT_DIRECTIONAL, PBR_WORKFLOW_METALLIC);
        case 1: PBR_BWDS(PBR_LIGHT_DIRECTIONAL, PBR_WORKFLOW_SPECULAR);
        case 2: PBR_BWDS(PBR_LIGHT_DIRECTIONAL, PBR_WORKFLOW_CONVERTED);
        case 3: PBR_BWDS(PBR_LIGHT_POINT, PBR_WORKFLOW_METALLIC);
        case 4: PBR_BWDS(PBR_LIGHT_POINT, PBR_WORKFLOW_SPECULAR);
        default: PBR_BWDS(PBR_LIGHT_POINT, PBR_WORKFLOW_CONVERTED);
    }
#undef PBR_BWDS
}

// Rounds of the streamed backward kernel (ct_backward.hpp: its grid is rounds x the waves the chip holds at once), or 0 when
// the launch does not qualify: fp16 maps, one light, untiled, rows a whole number of 128-pixel tiles, 4-byte aligned planes
// with even strides (its loads and stores move two fp16 values per lane).  g_bwd_run: -1 = rule, 0 = never, N = N rounds (A/B).
int stream_run(const pbr_render_desc *d, const void *grad_out, void *const g[5]) {
    if (g_bwd_run == 0 || d->map_dtype != PBR_F16 || d->n_lights != 1 || is_tiled(d) || d->width % 128) return 0;
    if ((int64_t)d->height * d->width >= (1ll << 30) || d->batch > 65535) return 0;
    auto ok = [](const pbr_map &m) {
        return !m.data || ((reinterpret_cast<uintptr_t>(m.data) & 3u) == 0 && m.batch_stride % 2 == 0 && m.channel_stride % 2 == 0);
    };
    if (!ok(d->albedo) || !ok(d->normal) || !ok(d->roughness) || !ok(d->metallic) || !ok(d->specular)) return 0;
    if (reinterpret_cast<uintptr_t>(grad_out) & 3u) return 0;
    for (int i = 0; i < 5; ++i)
        if (reinterpret_cast<uintptr_t>(g[i]) & 3u) return 0;
    return g_bwd_run > 0 ? g_bwd_run : 4;          // 4096^2: 1 round 144-146 us, 2: 140-141, 4: 137-139, 6: 138, 8: 139-140 (tools/bwd_stream_ab.sh)
}

// Tiles of the decomposition with the smallest tiles -- one pixel per lane, 64-lane workgroups: the most any launch of
// this descriptor can have whatever the tuning knobs say (more pixels per lane or larger workgroups only merge tiles).
// Same geometry as fill_args: bx = lanes along x (a power of two covering the row, at most 64), 64 / bx rows per tile.
static int64_t max_tiles(const pbr_render_desc *d) {
    int lg = 0;
    while ((1 << lg) < d->width && lg < 6) ++lg;
    const int64_t bx = 1ll << lg, by = 64 >> lg, rows = (int64_t)d->batch * d->height;
    const int64_t tiles = ((d->width + bx - 1) / bx) * ((rows + by - 1) / by);
    return tiles > INT32_MAX ? -1 : tiles;
}

// Workspace of the light / view gradients: max_tiles rows of partial sums (fp32), then kParamStageRows rows of stage sums (fp64).
static size_t stage_offset_bytes(const pbr_render_desc *d) {
    const size_t rows = (size_t)max_tiles(d) * (size_t)(3 + 6 * d->n_lights) * sizeof(float);
    return (rows + 7) & ~(size_t)7;
}

static int launch_backward(const pbr_render_desc *d, const void *grad_out, void *g_albedo, void *g_normal, void *g_roughness,
                           void *g_metallic, void *g_specular, void *g_params, void *workspace, void *stream) {
    const int rc = validate(d);
    if (rc != PBR_OK) return rc;
    if (!grad_out) return PBR_ERR_NULL_MAP;
    if (g_params && !workspace) return PBR_ERR_NULL_MAP;
    if (d->out_dtype != PBR_F32) return PBR_ERR_DTYPE;        // the upstream gradient is fp32; maps (and their gradients) fp32 | fp16
    int vec = pick_vec(d);                                    // grad_out and the g_* only need element alignment (f32x4_e)
    if (vec == 8) vec = 4;
    // Two pixels per lane (one packed pair): (a) with the light / view adjoints (PGRAD) the four-pixel body needs 256 VGPRs +
    // 40 AGPRs = one wave per SIMD, the two-pixel body 155 = three; (b) with fp16 maps and one light the launch is VALU-bound
    // and the two-pixel body (125 VGPRs, 4 waves per SIMD) runs 155.6 us against 162.6 us on a 4096^2 material (steady state,
    // tools/bwd_ab.sh of its round); with fp32 maps the four-pixel body wins (209 vs 218 us).  A rule since ABI 8 (pick_bwd_variant).
    const bool f16_one_light = d->map_dtype == PBR_F16 && d->n_lights == 1;
    if (vec == 4 && (g_params || f16_one_light)) vec = 2;
    // The light / view adjoints are sums over pixels: the overlapping last lane of a ragged row (lane_pos: dup) would
    // count its shared pixels twice, so odd widths take the one-pixel body there.
    if (g_params && (d->width & 1)) vec = 1;
    KArgs k;
    void *const gs[5] = {g_albedo, g_normal, g_roughness, g_metallic, g_specular};
    if (const int rounds = g_params ? 0 : stream_run(d, grad_out, gs)) {
        fill_args(d, 2, k, 6);
        k.o_cs = (int64_t)d->height * d->width; k.o_bs = 3 * k.o_cs;
        const BArgs b = {grad_out, g_albedo, g_normal, g_roughness, g_metallic, g_specular, nullptr};
        const int tiles = (int)(k.o_cs / 128);
        const bool spec = d->workflow == PBR_WORKFLOW_SPECULAR;
        const int n_stores = (g_albedo ? 3 : 0) + (g_normal && d->normal.data ? 3 : 0) + (g_roughness ? 1 : 0) +
                             (spec ? (g_specular ? 3 : 0) : (g_metallic ? 1 : 0));
        // every run-time flag on and every gradient wanted: the instantiation without flag branches
        const bool full = d->albedo_is_srgb && d->return_srgb && d->normal.data && g_albedo && g_normal && g_roughness &&
                          (spec ? (g_specular && d->specular_is_srgb) : (g_metallic && (d->workflow == PBR_WORKFLOW_METALLIC || d->specular_is_srgb)));
        // one-wave workgroups, kStreamWavesPerSimd of them per SIMD: the grid covers the chip `rounds` times, split over the materials
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        const int64_t slots = (int64_t)cus * 4 * kStreamWavesPerSimd * rounds;
        int64_t per_material = (slots + d->batch - 1) / d->batch;
        if (per_material > tiles) per_material = tiles;
        if (per_material < 1) per_material = 1;
        hipLaunchKernelGGL(pick_bwd_stream(d, full), dim3((unsigned)per_material, (unsigned)d->batch, 1), dim3(64, 1, 1), 0,
                               static_cast<hipStream_t>(stream), k, b, tiles, n_stores);
        const hipError_t e = hipGetLastError();
        return e == hipSuccess ? PBR_OK : 1000 + (int)e;
    }
    // Light / view adjoints: a workgroup adds its waves' sums into LDS with atomics -- with ONE wave per workgroup the
    // order of additions is fixed and the result deterministic run to run (the rows are added in fp64 in a fixed order).
    fill_args(d, vec, k, g_params ? 6 : 0);
    if (k.n_tiles < 0) return PBR_ERR_SHAPE;
    if (g_params && k.n_tiles > max_tiles(d)) return PBR_ERR_SHAPE;   // one workspace row per workgroup: never past what the size query promised
    k.o_cs = (int64_t)d->height * d->width; k.o_bs = 3 * k.o_cs;     // grad_out and the g_* are contiguous, whatever `out` was
    const BArgs b = {grad_out, g_albedo, g_normal, g_roughness, g_metallic, g_specular, static_cast<float *>(workspace)};
    hipStream_t st = static_cast<hipStream_t>(stream);
    const BwdFn fn = g_params ? pick_bwd<true>(d, vec) : pick_bwd<false>(d, vec);
    hipLaunchKernelGGL(fn, dim3((unsigned)k.n_tiles, 1, 1), dim3(1u << k.bt_log2, 1, 1), 0, st, k, b);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return 1000 + (int)err;
    if (g_params) {
        ParamFinishArgs f;
        std::memset(&f, 0, sizeof(f));
        const int n_param = 3 + 6 * d->n_lights;
        double *stage = reinterpret_cast<double *>(static_cast<char *>(workspace) + stage_offset_bytes(d));
        hipLaunchKernelGGL(param_grad_stage_kernel, dim3(kParamStageRows), dim3(256), 0, st, static_cast<const float *>(workspace), stage,
                           (int)k.n_tiles, n_param);
        f.stage = stage; f.out = static_cast<float *>(g_params);
        f.n_rows = kParamStageRows; f.n_lights = d->n_lights; f.light_type = d->light_type;
        f.dev = k.dev;
        for (int c = 0; c < 3; ++c) f.view[c] = d->view_dir[c];
        for (int i = 0; i < d->n_lights; ++i)
            for (int c = 0; c < 3; ++c) f.lights[i][c] = d->lights[i][c];
        hipLaunchKernelGGL(param_grad_finish_kernel, dim3(1u + 2u * (unsigned)d->n_lights), dim3(256), 0, st, f);
        err = hipGetLastError();
    }
    return err == hipSuccess ? PBR_OK : 1000 + (int)err;
}

}  // namespace pbr

extern "C" {

int pbr_cook_torrance_backward(const pbr_render_desc *d, const void *grad_out, void *g_albedo, void *g_normal,
                               void *g_roughness, void *g_metallic, void *g_specular, void *stream) {
    const pbr::TuningScope tuning(d);
    return pbr::launch_backward(d, grad_out, g_albedo, g_normal, g_roughness, g_metallic, g_specular, nullptr, nullptr, stream);
}

size_t pbr_param_grad_workspace_bytes(const pbr_render_desc *d) {
    const pbr::TuningScope tuning(d);
    if (pbr::validate(d) != PBR_OK) return 0;
    const int64_t tiles = pbr::max_tiles(d);
    return tiles < 0 ? 0 : pbr::stage_offset_bytes(d) + (size_t)pbr::kParamStageRows * (size_t)(3 + 6 * d->n_lights) * sizeof(double);
}

int pbr_cook_torrance_backward_params(const pbr_render_desc *d, const void *grad_out, void *g_albedo, void *g_normal,
                                      void *g_roughness, void *g_metallic, void *g_specular, void *g_params,
                                      void *workspace, void *stream) {
    const pbr::TuningScope tuning(d);
    if (!g_params) return PBR_ERR_NULL_MAP;
    return pbr::launch_backward(d, grad_out, g_albedo, g_normal, g_roughness, g_metallic, g_specular, g_params, workspace, stream);
}

}  // extern "C"
