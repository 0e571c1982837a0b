// ct_launch.hpp -- host side shared by the launchers (cook_torrance.hip, ct_backward.hip, ct_blend.hip): descriptor
// validation, the 16-byte-path test, the workgroup-order rule and the translation of a pbr_render_desc into the
// kernel-argument block.  Tuning knobs live in cook_torrance.hip (pbr_set_tuning).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/pbr_hip.h"
#include "ct_kernel.hpp"
#include "stream_shape.hpp"
#include "tuning.hpp"

namespace pbr {

// The schedule knobs (g_block_log2, g_lds_bytes, ...) are read through tuning.hpp: per call (pbr_render_desc.tuning), else the
// process-wide test hook (pbr_set_tuning), else the rule.  What the rules are and why:
//   * nt hint on: -5 % time; one-wave workgroups: -2 % vs 256 lanes (no LDS / barrier, so nothing is lost) -- 4096x4096 point /
//     metallic, tools/tune.py, DESIGN.md "Schedule experiments";
//   * g_lds_bytes: dynamic LDS per one-wave workgroup, unused by the kernel: an occupancy governor finer than whole waves per SIMD
//     (160 KiB / value = waves per CU).  amdgpu_waves_per_eu(3,3) on the kernel allows 12 waves per CU; the fp32 one-light kernels
//     stream fastest with 11 in flight (in-process A/B, DESIGN.md 3.2: 115.3 vs 118.5 us on 4096^2, 33.3 vs 34.0 us on 2048^2,
//     249.8 vs 254.1 us on 8 x 2048^2 directional), the fp16 and multi-light kernels with no cap.  -1 = that rule; >= 0 = this
//     many bytes for every launch.
constexpr int kLdsFor11WavesPerCu = 14848;   // floor(163840 / 14848) = 11

// The folding of view / light parameters into wave-uniform values runs on the host (parameters in pbr_render_desc) or, for parameters that
// live in device memory, in prepare_device_params_kernel: ONE definition, fp contraction off, IEEE sqrt and division on both sides, so the two
// produce the same bits.
#if defined(__HIP_DEVICE_COMPILE__)         // single IEEE operations: the device build runs with -ffp-contract=fast, and HIP's __fmul_rn is a plain
__device__ __forceinline__ float pbr_unfused(float x) { asm volatile("" : "+v"(x)); return x; }   // `*` that the compiler fuses all the same -- the
#define PBR_SQRT_RN(x) __fsqrt_rn(x)                                                              // product passes through an opaque statement instead
#define PBR_DIV_RN(a, b) __fdiv_rn(a, b)
#define PBR_MUL_RN(a, b) pbr_unfused((a) * (b))
#define PBR_ADD_RN(a, b) ((a) + (b))
#else
#define PBR_SQRT_RN(x) sqrtf(x)
#define PBR_DIV_RN(a, b) ((a) / (b))
#define PBR_MUL_RN(a, b) ((a) * (b))
#define PBR_ADD_RN(a, b) ((a) + (b))
#endif
__host__ __device__ inline float dot3_rn(const float a[3], const float b[3]) {           // ((a0 b0 + a1 b1) + a2 b2), every operation rounded
    return PBR_ADD_RN(PBR_ADD_RN(PBR_MUL_RN(a[0], b[0]), PBR_MUL_RN(a[1], b[1])), PBR_MUL_RN(a[2], b[2]));
}
__host__ __device__ inline void normalize_host(const float v[3], float o[3]) {   // F.normalize(v, dim=0), fp32
    const float nrm = PBR_SQRT_RN(dot3_rn(v, v));
    const float d = nrm > 1e-12f ? nrm : 1e-12f;
    o[0] = PBR_DIV_RN(v[0], d); o[1] = PBR_DIV_RN(v[1], d); o[2] = PBR_DIV_RN(v[2], d);
}
__host__ __device__ inline void fold_light(int light_type, const float V[3], const float raw[3], const float inten[3], LightU &u) {
    for (int c = 0; c < 3; ++c) u.inten[c] = inten[c];
    u.rhh = 0.0f; u.p5 = 0.0f;
    for (int c = 0; c < 3; ++c) u.h[c] = 0.0f;
    if (light_type == PBR_LIGHT_DIRECTIONAL) {
        normalize_host(raw, u.l);                                            // :126
        float hn[3];
        for (int c = 0; c < 3; ++c) u.h[c] = PBR_ADD_RN(V[c], u.l[c]);       // :155
        const float hh = dot3_rn(u.h, u.h);
        u.rhh = PBR_DIV_RN(1.0f, (hh > 1e-24f ? hh : 1e-24f));
        normalize_host(u.h, hn);
        float ct = dot3_rn(hn, V);                                           // :156-158
        ct = ct < 0.0f ? 0.0f : (ct > 1.0f ? 1.0f : ct);
        const float om = PBR_ADD_RN(1.0f, -ct);
        u.p5 = PBR_MUL_RN(PBR_MUL_RN(PBR_MUL_RN(om, om), PBR_MUL_RN(om, om)), om);       // :196
    } else {
        for (int c = 0; c < 3; ++c) u.l[c] = raw[c];
    }
}

inline int validate(const pbr_render_desc *d) {
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
    if (d->schedule < PBR_SCHEDULE_AUTO || d->schedule > PBR_SCHEDULE_XCD(12)) return PBR_ERR_SHAPE;
    if (d->out_batch_stride < 0 || d->out_channel_stride < 0) return PBR_ERR_SHAPE;
    if (d->out_channel_stride && d->out_channel_stride < (int64_t)d->height * d->width) return PBR_ERR_SHAPE;
    if (d->map_height || d->map_width) {             // tiled maps: whole repeats only
        if (d->map_height < 1 || d->map_width < 1 || d->height_total % d->map_height || d->width % d->map_width)
            return PBR_ERR_SHAPE;
    }
    return PBR_OK;
}

inline bool is_tiled(const pbr_render_desc *d) {
    return d->map_height > 0 && (d->map_height != d->height_total || d->map_width != d->width);
}

// Pixels per lane.  4 whenever a row holds 4 pixels: the vector accesses only need element alignment (ct_kernel.hpp,
// f32x4_e), and a width that 4 does not divide is covered by overlapping the last two lanes of a row (lane_pos) --
// measured on 4090^2 fp32: 5.6 TB/s against 3.5 (3.9 with 256-lane workgroups) for the one-pixel-per-lane kernels, which
// remain for rows shorter than 4 pixels and for tiled maps whose width 4 does not divide (a lane must not straddle a
// seam).  8 (fp16 maps, one light: 16-byte loads, and the fp32 result's piece exchange) keeps its alignment demands.
inline int pick_vec(const pbr_render_desc *d) {
    if (g_max_vec == 1 || d->width < 4) return 1;
    const bool tiled = is_tiled(d);
    if (tiled && d->map_width % 4) return 1;          // a lane's pixels must not straddle a seam
    // fp16 maps, ONE light (HBM-bound): 8 pixels per lane keep the loads 16 bytes wide.  With several
    // lights the kernel is VALU-bound and the 4-pixel body's lower register count wins.
    if (d->map_dtype == PBR_F16 && d->width % 8 == 0 && (!tiled || d->map_width % 8 == 0) && d->n_lights == 1 && g_f16_vec == 8 &&
        g_max_vec >= 8) {
        auto ok16 = [&](const pbr_map &m, bool three) {
            return !m.data || ((reinterpret_cast<uintptr_t>(m.data) & 15u) == 0 && m.batch_stride % 8 == 0 &&
                               (!three || m.channel_stride % 8 == 0));
        };
        if (ok16(d->albedo, true) && ok16(d->normal, true) && ok16(d->roughness, false) && ok16(d->metallic, false) &&
            ok16(d->specular, true) && (reinterpret_cast<uintptr_t>(d->out) & 15u) == 0 && d->out_batch_stride % 8 == 0 &&
            d->out_channel_stride % 8 == 0)
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
inline int schedule_xcd_log2(const pbr_render_desc *d, int vec) {
    if (d->schedule >= PBR_SCHEDULE_LINEAR) return d->schedule - PBR_SCHEDULE_LINEAR;
    const int64_t esz = d->map_dtype == PBR_F32 ? 4 : 2;
    const int64_t row_bytes = (int64_t)d->width * esz, tile_bytes = 64 * (int64_t)vec * esz;
    const int64_t plane_bytes = d->albedo.channel_stride * esz;
    if (d->map_dtype == PBR_F16) return 6;           // fp16 maps: runs are 1.5-5 % ahead on every shape tried
    if (row_bytes % tile_bytes) return 6;            // ragged rows: a tile row does not end where a plane row ends
    // Plane strides that are a multiple of 8 MiB put the same offset of all 11 planes of a material onto the same HBM
    // channel group; under the linear order every XCD then works on that group at once.  Runs hand each XCD its own
    // 64 KiB pieces.  Measured (round 2, tools/order_sweep.sh + tune.py, runs vs linear): 8 MiB planes +0.5 %, 16 MiB
    // +8..10 %, 24 MiB -0.9 %, 32 MiB +4 %; other strides prefer the linear order: 4 MiB -3.8 %, 9 MiB -3.7 %,
    // 36 MiB -1.6 %.  From 64 MiB on (4096^2 planes) the two orders are level for one material (+-0.5 % on four boxes)
    // and the linear one is 2 % ahead for a batch (4 x 4096^2: 451.9 vs 462.1 us), so the rule stops below that.
    if (plane_bytes % (8ll << 20) == 0 && plane_bytes < (64ll << 20)) return 6;
    return 0;
}

inline void fill_args(const pbr_render_desc *d, int vec, KArgs &k, int block_log2 = 0) {
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
    k.wv = (d->width + vec - 1) / vec;             // ragged widths: the last lane of a row overlaps its neighbour (lane_pos)
    // One wave per workgroup; the one-pixel kernels cover only 256 bytes of a plane per wave and run 14 % faster in
    // four-wave workgroups (4090^2: 186 against 213 us).
    // `block_log2`: the caller's demand (the light-gradient kernels need one-wave workgroups, see ct_backward.hip).
    const int want = block_log2 ? block_log2 : g_block_log2;
    k.bt_log2 = want == 0 ? (vec == 1 ? 8 : 6) : (want < 6 ? 6 : (want > 8 ? 8 : want));
    int lg = 0;
    while ((1 << lg) < k.wv && lg < k.bt_log2) ++lg;
    k.bx_log2 = lg;
    const int bx = 1 << lg, by = (1 << k.bt_log2) >> lg;
    k.tiles_x = (k.wv + bx - 1) / bx;
    const int64_t tiles = (int64_t)k.tiles_x * ((k.rows + by - 1) / by);
    k.n_tiles = tiles > INT32_MAX ? -1 : (int32_t)tiles;      // -1: more tiles than a 1-D grid holds, rejected by the callers
    k.xcd_log2 = schedule_xcd_log2(d, vec);
    k.xcd_tiles = k.n_tiles < 0 ? 0 : (k.n_tiles >> (k.xcd_log2 + 3)) << (k.xcd_log2 + 3);
    // 8-pixel lanes with an fp32 result swap 16-byte pieces between the lanes of a row before storing (ct_kernel.hpp)
    k.xpose = vec == 8 && d->out_dtype == PBR_F32;
    // Scalar plane addresses (ct_kernel.hpp, plane_at) need every workgroup inside one material -- tile rows divide the
    // band height, or there is one material -- and 32-bit byte offsets inside a plane (< 2^30 elements).  Used for
    // single materials only: in-process A/B (tools/tune.py, knob "sb") 4096^2 fp32 115.5 vs 116.4 us, fused tile(2)
    // 80.9 vs 84.3, backward fp32 212.2 vs 217.6, fp16 -> fp16 291.8 vs 297.4; batches run 0.5-2 % SLOWER with it
    // (4 x 4096^2 fp16 334.2 vs 328.2, 16 x 2048^2 fp32 481.7 vs 474.7, 16 lights 1148 vs 1141) -- g_scalar_base = 2
    // turns it on for those too.
    const int64_t plane_px = (int64_t)d->height * d->width;
    const int64_t map_px = is_tiled(d) ? (int64_t)d->map_height * d->map_width : plane_px;
    const bool sb_allowed = (d->batch == 1 || d->height % by == 0) && plane_px < (1ll << 30) && map_px < (1ll << 30);
    k.sbase = sb_allowed && (g_scalar_base == 2 || (g_scalar_base == 1 && d->batch == 1));
    k.div_h.init((uint32_t)d->height);
    k.div_tx.init((uint32_t)k.tiles_x);
    k.tiled = is_tiled(d);
    k.map_h = k.tiled ? d->map_height : d->height_total; k.map_w = k.tiled ? d->map_width : d->width;
    k.div_mh.init((uint32_t)k.map_h); k.div_mw.init((uint32_t)k.map_w);
    k.y_offset = d->y_offset; k.H_total = d->height_total;
    k.div_reps.init(1);
    // Measured (tile_probe.py (a probe of its round, removed with its knob: git 9ce0718:tools/), 2048^2 maps, tile(2), MI355X): the fold order brings the read traffic from 2.0 x to 1.0002 x
    // the source (PMC) -- and costs fp32 maps 8-17 % of time whatever the band (78-82 us in row order, 87-98 us folded, with or
    // without the streaming hint): in row order the second visit is served by the 256 MB memory-side cache, and the launch is
    // VALU-bound either way.  fp16 maps run level (65-69 us both).  The rule therefore folds fp16 maps only.
    const bool fold_wanted = d->map_dtype == PBR_F16;
    if (k.tiled && fold_wanted && by == 1 && d->batch == 1 && d->y_offset == 0 && d->height == d->height_total &&
        d->height_total > d->map_height && k.n_tiles > 0) {
        // rows per band: the band's texels (all planes) within ~2 MiB -- 1/8 of it per XCD, beside the result streams in a 4 MiB
        // L2 -- a power of two that divides map_h, the band a whole number of XCD periods (tile_probe.py (a probe of its round, removed with its knob: git 9ce0718:tools/): "fold").
        const int64_t esz = d->map_dtype == PBR_F32 ? 4 : 2;
        const int64_t row_bytes = (int64_t)d->map_width * esz * (3 + (d->normal.data ? 3 : 0) + 1 + (d->workflow == PBR_WORKFLOW_SPECULAR ? 3 : 1));
        int fl = 0;
        while (fl < 12 && (row_bytes << (fl + 1)) <= (2ll << 20)) ++fl;
        const int64_t period = 8ll << k.xcd_log2;
        while (fl > 0 && (d->map_height % (1 << fl) || (((int64_t)k.tiles_x << fl) % period))) --fl;
        if (fl > 0) { k.fold_log2 = fl; k.fold_reps = d->height_total / d->map_height; k.div_reps.init((uint32_t)k.fold_reps); }
    }
    // `light_size or 1.0` (:130) is Python truthiness: only None / 0 / 0.0 / -0.0 mean "not given"; a NEGATIVE size is truthy
    // and mirrors the grid (linspace from +|s|/2 down to -|s|/2), NaN is truthy too (the launcher fills the result, below).
    const float size = (d->light_size != 0.0f) ? d->light_size : 1.0f;
    const float lo = (float)(-(double)size / 2), hi = (float)((double)size / 2);
    k.x0 = lo; k.x1 = hi; k.xstep = d->width > 1 ? (hi - lo) / (float)(d->width - 1) : 0.0f;
    k.y0 = lo; k.y1 = hi; k.ystep = d->height_total > 1 ? (hi - lo) / (float)(d->height_total - 1) : 0.0f;
    if (d->width == 1) k.x1 = k.x0;          // torch.linspace(a, b, 1) == [a]
    if (d->height_total == 1) k.y1 = k.y0;
    normalize_host(d->view_dir, k.V);
    k.n_lights = d->n_lights;
    k.albedo_srgb = d->albedo_is_srgb != 0; k.spec_srgb = d->specular_is_srgb != 0;
    k.out_srgb = d->return_srgb != 0; k.has_normal = d->normal.data != nullptr;
    k.grey_lights = 1;
    for (int i = 0; i < d->n_lights; ++i)
        if (d->intensities[i][0] != d->intensities[i][1] || d->intensities[i][0] != d->intensities[i][2]) k.grey_lights = 0;
    for (int i = 0; i < d->n_lights; ++i) fold_light(d->light_type, k.V, d->lights[i], d->intensities[i], k.lights[i]);
    k.dev = reinterpret_cast<uint64_t>(d->device_params);
    if (k.dev) k.grey_lights = 0;            // the intensities are not known here: the kernels ask the block's own flag
}

// A NaN light_size is truthy (`nan or 1.0` is nan): the reference's point-light grid, hence every value of its result, is NaN
// (torch.clamp and both colour transfers propagate it).  The kernels' clamps would flush it, so the launchers write that
// answer directly: the result planes are filled with quiet NaNs.  Directional lights never read light_size (:125-127).
inline bool nan_light_size(const pbr_render_desc *d) { return d->light_type == PBR_LIGHT_POINT && d->light_size != d->light_size; }
inline int fill_result_nan(const pbr_render_desc *d, hipStream_t st) {
    const int64_t plane = (int64_t)d->height * d->width;
    const int64_t cs = d->out_channel_stride ? d->out_channel_stride : plane, bs = d->out_batch_stride ? d->out_batch_stride : 3 * cs;
    for (int b = 0; b < d->batch; ++b)
        for (int c = 0; c < 3; ++c) {
            hipError_t e;
            if (d->out_dtype == PBR_F32) e = hipMemsetD32Async((hipDeviceptr_t)(static_cast<float *>(d->out) + b * bs + c * cs), 0x7fc00000, (size_t)plane, st);
            else e = hipMemsetD16Async((hipDeviceptr_t)(static_cast<uint16_t *>(d->out) + b * bs + c * cs), 0x7e00, (size_t)plane, st);
            if (e != hipSuccess) return 1000 + (int)e;
        }
    return PBR_OK;
}

using KernelFn = void (*)(const KArgs);
KernelFn pick_batch_kernel(const pbr_render_desc *d, int nb);               // ct_batch.hip
KernelFn pick_repeat_kernel(const pbr_render_desc *d);                      // ct_tiled.hip
void fill_repeat_args(const pbr_render_desc *d, KArgs &k);
// ct_repeat_backward.hip: gradients of TILED maps folded in registers (and, with `loss`, the rendering-loss step over tiled maps)
bool repeat_backward_serves(const pbr_render_desc *d);
bool repeat_loss_serves(const pbr_render_desc *d);
int64_t repeat_backward_tiles(const pbr_render_desc *d);
int launch_repeat_backward(const pbr_render_desc *d, const void *upstream, void *g_albedo, void *g_normal, void *g_roughness, void *g_metallic,
                           void *g_specular, bool loss, float scale, float *partials, hipStream_t st);
// pbr_cook_torrance_blend_backward over tiled maps through the same walk (ct_repeat_backward.hip); kblend / g1 / g2: KBlend / BArgs / BBlend by address
bool repeat_blend_backward_serves(const pbr_render_desc *d);
int launch_repeat_blend_backward(const pbr_render_desc *d, const void *kblend, const void *grad_out, const void *g1, const void *g2, hipStream_t st);

// The repeat-inner kernels (cook_torrance_repeat_kernel, cook_torrance_repeat_backward_kernel) serve every tiled launch whose map rows hold a
// 4-texel lane: the whole tiled image or a row band of it (a multi-GPU shard) of ANY height -- a band thinner than a period walks the window
// of source rows it touches -- and ragged map widths (the last lane of a row moves back and overlaps its neighbour, as everywhere).  Until
// round 6 thin bands and ragged widths took the wrap-around form of cook_torrance_kernel (a second read of every texel: 1.40 x the maps from
// HBM); it remains for map rows shorter than 4 texels and behind PBR_TUNE_TILE_REPEAT = 0 (the A/B of the tests).
inline bool repeat_inner(const pbr_render_desc *d) {
    return g_tile_repeat != 0 && is_tiled(d) && d->map_width >= 4 && g_max_vec >= 4;
}
// A band thinner than one period of the map's rows: the walk covers the cyclic window of source rows the band touches (KArgs::win_y0).
inline bool repeat_thin_band(const pbr_render_desc *d) { return d->height < d->map_height; }

// Materials per lane for this launch, or 0 for the one-material kernels.  Several lights make the launch VALU-bound, and
// the light geometry is shared by every material at a pixel position: groups of 4 (or 2) consecutive materials per lane.
inline int batch_group(const pbr_render_desc *d, int vec) {
    if (g_batch_inner == 0 || d->n_lights < 2 || vec < 4 || is_tiled(d)) return 0;
    if (g_batch_inner == 2 || g_batch_inner == 4) return d->batch % g_batch_inner == 0 ? g_batch_inner : 0;
    return d->batch % 4 == 0 ? 4 : (d->batch % 2 == 0 ? 2 : 0);
}

}  // namespace pbr
