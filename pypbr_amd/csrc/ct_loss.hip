// ct_loss.hip -- the rendering-loss step as ONE kernel (C ABI: pbr_cook_torrance_mse_step).
//
// The documented training use of the reference (docs/source/tutorials/06_advanced.rst:73-107) is
//     loss = nn.MSELoss()(brdf(predicted_material, ...), brdf(ground_truth_material, ...))
// followed by loss.backward().  The ground-truth rendering is a constant of the step (rendered once, by pbr_cook_torrance);
// what a step repeats for the PREDICTED material is: evaluate (44 B/pixel), subtract / square / mean and its backward
// (36 B/pixel of elementwise torch kernels), the backward kernel (76 B/pixel).  Here it is one pass: the predicted maps and the
// target image are read once (32 + 12 B/pixel), the colour is formed in registers, its difference to the target gives the
// pixel's share of the loss AND its upstream gradient 2 (out - target) / N, the chain rule runs back through the shading (the
// very code of cook_torrance_backward_kernel: backward_body_to with the MseLoss policy), and the four map gradients are
// written (32 B/pixel): 76 B/pixel instead of 156.
//
// The loss itself: every workgroup (one wave) leaves its sum of squared differences in a workspace row; a second small kernel
// adds the rows in fp64 in a fixed order (deterministic) and writes mean = sum / N.
#include "ct_backward.hpp"
#include "ct_launch.hpp"

namespace pbr {


// Waves per SIMD the register allocation must leave room for: the packed pair for fp16 maps fits 128 VGPRs (four waves) in the
// backward kernel, not here -- the target pixels and the squared differences come on top (56 bytes of scratch): three waves.
template <int VEC, bool MULTI, typename TM>
constexpr int mse_min_waves() { return VEC == 2 && !MULTI && sizeof(TM) == 2 ? 3 : 1; }

template <int LIGHT, int WF, int VEC, bool MULTI, typename TM>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(mse_min_waves<VEC, MULTI, TM>())))
void cook_torrance_mse_step_kernel(const KArgs a, const BArgs b, const float *__restrict__ target, float scale, float *__restrict__ partials) {
    const uint32_t tile = tile_of_workgroup(a, blockIdx.x);
    const int ty = (int)a.div_tx.div(tile);
    const LanePos p = lane_pos<VEC, true>(a, (int)tile - ty * a.tiles_x, ty);      // lanes outside the map shade a clamped position: every lane reaches the sum
    Texels<VEC> t;
    MseLoss<VEC> loss;
    loss.scale = scale;
    loss.sq = 0.0f;
    float go[3][VEC];                                                              // unused by the MseLoss policy
    const int64_t opix = p.b * a.o_bs + p.pix;
    if constexpr (sizeof(TM) == 4) {
        load_texels<WF, TM, VEC, true>(a, a.has_normal != 0, p, t);
    } else if (p.sb) {
        if (a.has_normal) load_texels_fixed<WF, TM, VEC, true, true, true>(a, p, t); else load_texels_fixed<WF, TM, VEC, true, true, false>(a, p, t);
    } else {
        if (a.has_normal) load_texels_fixed<WF, TM, VEC, true, false, true>(a, p, t); else load_texels_fixed<WF, TM, VEC, true, false, false>(a, p, t);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        Ld<float, VEC>::template load<true>(target, opix + c * a.o_cs, loss.tgt[c]);
#pragma unroll
        for (int j = 0; j < VEC; ++j) go[c][j] = 0.0f;
    }
    backward_body_to<LIGHT, WF, VEC, MULTI, TM, false>(a, b, p, t, go, nullptr, 0,
        [&](float (&ga)[3][VEC], float (&gn)[3][VEC], float (&gr)[VEC], float (&gm)[VEC], float (&gs)[3][VEC]) {
            if (p.valid) store_gradients<WF, VEC, TM>(a, b, p, ga, gn, gr, gm, gs);
        }, loss);
    const float mine = p.valid ? loss.sq : 0.0f;
    const float total = wave_sum(mine);
    if (threadIdx.x == 0) partials[blockIdx.x] = total;
}

// The streamed form (fp16 maps, one light, whole 128-pixel tiles: ct_backward.hip stream_run): cook_torrance_backward_stream_kernel's schedule --
// persistent one-wave workgroups, the next tile's texels AND target pixels prefetched global -> LDS by DMA -- with the MseLoss policy.
// The target image takes the upstream gradient's three fp32 planes in the tile buffer; a wave keeps ONE running sum over all its tiles.
// Two waves per SIMD: with the target pixels and the squared differences on top, the point-light / converted body does not fit the 168 VGPRs of
// three (8 bytes of scratch, and a compiler-inserted wait on the reload: tools/check_isa.py rejects it).
#ifndef PBR_MSE_STREAM_WAVES
#define PBR_MSE_STREAM_WAVES 2
#endif
template <int LIGHT, int WF, bool FULL>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(PBR_MSE_STREAM_WAVES)))
void cook_torrance_mse_stream_kernel(const KArgs a, const BArgs b, const int tiles_per_material, const int n_stores, float scale, float *__restrict__ partials) {
    __shared__ uint32_t buf[kStreamLdsWords];
    MseLoss<2> loss;
    loss.scale = scale;
    loss.sq = 0.0f;
    backward_stream_body<LIGHT, WF, FULL>(a, b, tiles_per_material, n_stores, buf, loss);
    const float total = wave_sum(loss.sq);
    if (threadIdx.x == 0) partials[blockIdx.y * gridDim.x + blockIdx.x] = total;
}

// partials[n] (fp32, one per workgroup of the step kernel) -> *loss = sum / count, in two stages, fp64, fixed order:
// kMseStageGroups workgroups each add a contiguous block of the partials (coalesced: lane t takes elements t, t + 256, ...) into
// stage[blockIdx]; then one workgroup adds the stage sums.  (One workgroup walking all 131 072 partials of a 4096^2 launch by
// itself took longer than the step kernel: the same lesson as the light-gradient finish kernel, ct_backward.hpp.)
constexpr int kMseStageGroups = 256;
__device__ __forceinline__ double block_sum_256(double s, double *red) {
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    return red[0];
}
__global__ __launch_bounds__(256) void mse_stage_kernel(const float *__restrict__ partials, int n, double *__restrict__ stage) {
    __shared__ double red[256];
    const int per = (n + kMseStageGroups - 1) / kMseStageGroups;
    const int i0 = blockIdx.x * per, i1 = min(n, i0 + per);
    double s = 0.0;
    for (int i = i0 + threadIdx.x; i < i1; i += 256) s += partials[i];
    const double total = block_sum_256(s, red);
    if (threadIdx.x == 0) stage[blockIdx.x] = total;
}
__global__ __launch_bounds__(256) void mse_finish_kernel(const double *__restrict__ stage, double inv_count, float *__restrict__ loss) {
    __shared__ double red[256];
    const double total = block_sum_256(threadIdx.x < kMseStageGroups ? stage[threadIdx.x] : 0.0, red);
    if (threadIdx.x == 0) *loss = (float)(total * inv_count);
}

// x *= *scalar (device scalar), in place; returns at once when the scalar is exactly 1 (the usual upstream gradient of a loss)
template <typename T>
__global__ __launch_bounds__(256) void scale_by_device_scalar_kernel(T *__restrict__ x, size_t n, const float *__restrict__ scalar) {
    const float k = *scalar;
    if (k == 1.0f) return;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) x[i] = (T)((float)x[i] * k);
}

// few partials (the streamed step leaves one per resident wave, ~3 000): ONE workgroup adds them, fp64, fixed order -- one launch instead of two
__global__ __launch_bounds__(256) void mse_reduce_small_kernel(const float *__restrict__ partials, int n, double inv_count, float *__restrict__ loss) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += partials[i];
    const double total = block_sum_256(s, red);
    if (threadIdx.x == 0) *loss = (float)(total * inv_count);
}
constexpr int kMseSmall = 16384;

// x_i *= *scalar for up to five buffers in ONE launch (the gradients of a step); returns at once when the scalar is exactly 1
struct ScaleList { void *data[5]; size_t n[5]; };
template <typename T>
__global__ __launch_bounds__(256) void scale_list_kernel(ScaleList l, const float *__restrict__ scalar) {
    const float k = *scalar;
    if (k == 1.0f) return;
    const size_t stride = (size_t)gridDim.x * blockDim.x, first = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        T *x = static_cast<T *>(l.data[j]);
        for (size_t i = first; i < l.n[j]; i += stride) x[i] = (T)((float)x[i] * k);
    }
}

using MseFn = void (*)(const KArgs, const BArgs, const float *, float, float *);

template <int L, int W>
static MseFn pick_mse(bool half_maps, int vec, bool multi) {
    if (half_maps) {
        if (vec == 2) return multi ? cook_torrance_mse_step_kernel<L, W, 2, true, __half> : cook_torrance_mse_step_kernel<L, W, 2, false, __half>;
        return multi ? cook_torrance_mse_step_kernel<L, W, 1, true, __half> : cook_torrance_mse_step_kernel<L, W, 1, false, __half>;
    }
    if (vec == 2) return multi ? cook_torrance_mse_step_kernel<L, W, 2, true, float> : cook_torrance_mse_step_kernel<L, W, 2, false, float>;
    return multi ? cook_torrance_mse_step_kernel<L, W, 1, true, float> : cook_torrance_mse_step_kernel<L, W, 1, false, float>;
}

using MseStreamFn = void (*)(const KArgs, const BArgs, int, int, float, float *);
static MseStreamFn pick_mse_stream(const pbr_render_desc *d, bool full) {
    const bool point = d->light_type == PBR_LIGHT_POINT;
#define PBR_MSES(L, W) return full ? cook_torrance_mse_stream_kernel<L, W, true> : cook_torrance_mse_stream_kernel<L, W, false>
    switch ((point ? 3 : 0) + d->workflow) {
        case 0: PBR_MSES(PBR_LIGHT_DIRECTIONAL, PBR_WORKFLOW_METALLIC);
        case 1: PBR_MSES(PBR_LIGHT_DIRECTIONAL, PBR_WORKFLOW_SPECULAR);
        case 2: PBR_MSES(PBR_LIGHT_DIRECTIONAL, PBR_WORKFLOW_CONVERTED);
        case 3: PBR_MSES(PBR_LIGHT_POINT, PBR_WORKFLOW_METALLIC);
        case 4: PBR_MSES(PBR_LIGHT_POINT, PBR_WORKFLOW_SPECULAR);
        default: PBR_MSES(PBR_LIGHT_POINT, PBR_WORKFLOW_CONVERTED);
    }
#undef PBR_MSES
}
int stream_run(const pbr_render_desc *d, const void *grad_out, void *const g[5]);      // ct_backward.hip: rounds of the streamed form, 0 = does not qualify

// Pixels per lane: the loss is a sum over pixels, so no lane may see a pixel twice (the overlapping last lane of a ragged row,
// lane_pos: dup): two when the width is even, else one.  Two also for fp32 maps with one light, where the backward kernel takes
// four: with the target pixels on top the four-pixel body needs 256 VGPRs + 21 AGPRs (one wave per SIMD; held to two waves it
// spills 72-88 bytes), and measured on a 4096^2 material it loses -- point light 295 against 228 us, directional 225 against 219
// (tools/loss_step_probe.py, round 3).
static int mse_vec(const pbr_render_desc *d) { return g_max_vec == 1 || (d->width & 1) ? 1 : 2; }

// workspace: the step kernel's partial sums (fp32, one per workgroup), then kMseStageGroups stage sums (fp64, 8-byte aligned)
static size_t mse_stage_offset(size_t tiles) { return (tiles * sizeof(float) + 7) & ~(size_t)7; }

static int64_t mse_tiles(const pbr_render_desc *d, int vec) {
    KArgs k;
    fill_args(d, vec, k, 6);
    return k.n_tiles;
}

}  // namespace pbr

extern "C" {

size_t pbr_mse_step_workspace_bytes(const pbr_render_desc *d) {
    const pbr::TuningScope tuning(d);
    if (pbr::validate(d) != PBR_OK) return 0;
    if (pbr::is_tiled(d) && !pbr::repeat_loss_serves(d)) return 0;
    // tiled maps: one partial sum per workgroup of the repeat-inner kernel; else one pixel per lane: the most workgroups any launch of this descriptor has
    const int64_t tiles = pbr::is_tiled(d) ? pbr::repeat_backward_tiles(d) : pbr::mse_tiles(d, 1);
    return tiles < 0 ? 0 : pbr::mse_stage_offset((size_t)tiles) + (size_t)pbr::kMseStageGroups * sizeof(double);
}

int pbr_cook_torrance_mse_step(const pbr_render_desc *d, const void *target, void *g_albedo, void *g_normal, void *g_roughness,
                               void *g_metallic, void *g_specular, void *loss, void *workspace, void *stream) {
    const pbr::TuningScope tuning(d);
    using namespace pbr;
    const int rc = validate(d);
    if (rc != PBR_OK) return rc;
    if (!target || !loss || !workspace) return PBR_ERR_NULL_MAP;
    if (d->out_dtype != PBR_F32) return PBR_ERR_DTYPE;        // the target image and the colour it is compared with are fp32
    if (nan_light_size(d)) return PBR_ERR_UNSUPPORTED;
    const int vec = mse_vec(d);
    KArgs k;
    const double count = 3.0 * (double)d->batch * (double)d->height * (double)d->width;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (is_tiled(d)) {
        // MaterialBase.tile fused (base.py:524-537): the repeat-inner kernel walks the maps, compares every repeat with the target and leaves
        // MAP-sized gradients (ct_repeat_backward.hpp); launches it does not serve are the caller's to split (render, loss, folded backward)
        if (!repeat_loss_serves(d)) return PBR_ERR_UNSUPPORTED;
        const int64_t tiles = repeat_backward_tiles(d);
        if (tiles < 0) return PBR_ERR_SHAPE;
        const int e = launch_repeat_backward(d, target, g_albedo, g_normal, g_roughness, g_metallic, g_specular, true, (float)(2.0 / count),
                                             static_cast<float *>(workspace), st);
        if (e != PBR_OK) return e;
        if (tiles <= kMseSmall) {
            hipLaunchKernelGGL(mse_reduce_small_kernel, dim3(1), dim3(256), 0, st, static_cast<const float *>(workspace), (int)tiles, 1.0 / count, static_cast<float *>(loss));
        } else {
            double *stage = reinterpret_cast<double *>(static_cast<char *>(workspace) + mse_stage_offset((size_t)tiles));
            hipLaunchKernelGGL(mse_stage_kernel, dim3(kMseStageGroups), dim3(256), 0, st, static_cast<const float *>(workspace), (int)tiles, stage);
            hipLaunchKernelGGL(mse_finish_kernel, dim3(1), dim3(256), 0, st, stage, 1.0 / count, static_cast<float *>(loss));
        }
        const hipError_t err = hipGetLastError();
        return err == hipSuccess ? PBR_OK : 1000 + (int)err;
    }
    void *const gs[5] = {g_albedo, g_normal, g_roughness, g_metallic, g_specular};
    if (const int rounds = g_mse_stream ? stream_run(d, target, gs) : 0) {      // fp16 maps, one light: the streamed form
        fill_args(d, 2, k, 6);
        k.o_cs = (int64_t)d->height * d->width; k.o_bs = 3 * k.o_cs;
        const BArgs b = {target, g_albedo, g_normal, g_roughness, g_metallic, g_specular, nullptr};
        const int tiles = (int)(k.o_cs / 128);
        const bool spec = d->workflow == PBR_WORKFLOW_SPECULAR;
        const int n_stores = (g_albedo ? 3 : 0) + (g_normal && d->normal.data ? 3 : 0) + (g_roughness ? 1 : 0) +
                             (spec ? (g_specular ? 3 : 0) : (g_metallic ? 1 : 0));
        // every run-time flag on and every gradient wanted: the instantiation without flag branches (as in ct_backward.hip)
        const bool full = d->albedo_is_srgb && d->return_srgb && d->normal.data && g_albedo && g_normal && g_roughness &&
                          (spec ? (g_specular && d->specular_is_srgb) : (g_metallic && (d->workflow == PBR_WORKFLOW_METALLIC || d->specular_is_srgb)));
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        // (the allocator is ALLOWED two waves per SIMD and lands on 138-167 VGPRs: three fit, and the grid is sized for three)
        int64_t per_material = ((int64_t)cus * 4 * kStreamWavesPerSimd * rounds + d->batch - 1) / d->batch;
        if (per_material > tiles) per_material = tiles;
        if (per_material < 1) per_material = 1;
        const int64_t n_partials = per_material * d->batch;             // <= tiles * batch <= the workgroups of the one-tile form
        hipLaunchKernelGGL(pick_mse_stream(d, full), dim3((unsigned)per_material, (unsigned)d->batch, 1), dim3(64, 1, 1), 0, st, k, b, tiles, n_stores,
                           (float)(2.0 / count), static_cast<float *>(workspace));
        hipError_t err = hipGetLastError();
        if (err != hipSuccess) return 1000 + (int)err;
        if (n_partials <= kMseSmall) {
            hipLaunchKernelGGL(mse_reduce_small_kernel, dim3(1), dim3(256), 0, st, static_cast<const float *>(workspace), (int)n_partials, 1.0 / count, static_cast<float *>(loss));
        } else {
            double *stage = reinterpret_cast<double *>(static_cast<char *>(workspace) + mse_stage_offset((size_t)mse_tiles(d, 1)));
            hipLaunchKernelGGL(mse_stage_kernel, dim3(kMseStageGroups), dim3(256), 0, st, static_cast<const float *>(workspace), (int)n_partials, stage);
            hipLaunchKernelGGL(mse_finish_kernel, dim3(1), dim3(256), 0, st, stage, 1.0 / count, static_cast<float *>(loss));
        }
        err = hipGetLastError();
        return err == hipSuccess ? PBR_OK : 1000 + (int)err;
    }
    fill_args(d, vec, k, 6);                                  // one-wave workgroups: one partial sum per workgroup, no LDS reduction
    if (k.n_tiles < 0) return PBR_ERR_SHAPE;
    k.o_cs = (int64_t)d->height * d->width; k.o_bs = 3 * k.o_cs;     // target and gradient planes are contiguous
    const BArgs b = {nullptr, g_albedo, g_normal, g_roughness, g_metallic, g_specular, nullptr};
    const bool multi = d->n_lights > 1, point = d->light_type == PBR_LIGHT_POINT, half_maps = d->map_dtype == PBR_F16;
    MseFn fn = nullptr;
    switch ((point ? 3 : 0) + d->workflow) {
        case 0: fn = pick_mse<PBR_LIGHT_DIRECTIONAL, PBR_WORKFLOW_METALLIC>(half_maps, vec, multi); break;
        case 1: fn = pick_mse<PBR_LIGHT_DIRECTIONAL, PBR_WORKFLOW_SPECULAR>(half_maps, vec, multi); break;
        case 2: fn = pick_mse<PBR_LIGHT_DIRECTIONAL, PBR_WORKFLOW_CONVERTED>(half_maps, vec, multi); break;
        case 3: fn = pick_mse<PBR_LIGHT_POINT, PBR_WORKFLOW_METALLIC>(half_maps, vec, multi); break;
        case 4: fn = pick_mse<PBR_LIGHT_POINT, PBR_WORKFLOW_SPECULAR>(half_maps, vec, multi); break;
        default: fn = pick_mse<PBR_LIGHT_POINT, PBR_WORKFLOW_CONVERTED>(half_maps, vec, multi); break;
    }
    hipLaunchKernelGGL(fn, dim3((unsigned)k.n_tiles, 1, 1), dim3(64, 1, 1), 0, st, k, b, static_cast<const float *>(target),
                       (float)(2.0 / count), static_cast<float *>(workspace));
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return 1000 + (int)err;
    if (k.n_tiles <= kMseSmall) {
        hipLaunchKernelGGL(mse_reduce_small_kernel, dim3(1), dim3(256), 0, st, static_cast<const float *>(workspace), (int)k.n_tiles, 1.0 / count, static_cast<float *>(loss));
    } else {
        double *stage = reinterpret_cast<double *>(static_cast<char *>(workspace) + mse_stage_offset((size_t)mse_tiles(d, 1)));
        hipLaunchKernelGGL(mse_stage_kernel, dim3(kMseStageGroups), dim3(256), 0, st, static_cast<const float *>(workspace), (int)k.n_tiles, stage);
        hipLaunchKernelGGL(mse_finish_kernel, dim3(1), dim3(256), 0, st, stage, 1.0 / count, static_cast<float *>(loss));
    }
    err = hipGetLastError();
    return err == hipSuccess ? PBR_OK : 1000 + (int)err;
}

int pbr_scale_by_device_scalar(void *data, size_t n, int dtype, const void *scalar, void *stream) {
    using namespace pbr;
    if (!data || !scalar) return PBR_ERR_NULL_MAP;
    if (dtype != PBR_F32 && dtype != PBR_F16) return PBR_ERR_DTYPE;
    if (n == 0) return PBR_OK;
    const size_t blocks = (n + 255) / 256;
    const unsigned grid = (unsigned)(blocks > 2048 ? 2048 : blocks);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == PBR_F32) hipLaunchKernelGGL((scale_by_device_scalar_kernel<float>), dim3(grid), dim3(256), 0, st, static_cast<float *>(data), n, static_cast<const float *>(scalar));
    else hipLaunchKernelGGL((scale_by_device_scalar_kernel<_Float16>), dim3(grid), dim3(256), 0, st, static_cast<_Float16 *>(data), n, static_cast<const float *>(scalar));
    const hipError_t err = hipGetLastError();
    return err == hipSuccess ? PBR_OK : 1000 + (int)err;
}

int pbr_scale_list_by_device_scalar(void *const *data, const size_t *n, int count, int dtype, const void *scalar, void *stream) {
    using namespace pbr;
    if (!data || !n || !scalar) return PBR_ERR_NULL_MAP;
    if (count < 0 || count > 5) return PBR_ERR_SHAPE;
    if (dtype != PBR_F32 && dtype != PBR_F16) return PBR_ERR_DTYPE;
    ScaleList l = {};
    size_t most = 0;
    for (int j = 0; j < count; ++j) {
        if (n[j] && !data[j]) return PBR_ERR_NULL_MAP;
        l.data[j] = data[j]; l.n[j] = n[j];
        most = n[j] > most ? n[j] : most;
    }
    if (most == 0) return PBR_OK;
    const size_t blocks = (most + 255) / 256;
    const unsigned grid = (unsigned)(blocks > 2048 ? 2048 : blocks);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == PBR_F32) hipLaunchKernelGGL((scale_list_kernel<float>), dim3(grid), dim3(256), 0, st, l, static_cast<const float *>(scalar));
    else hipLaunchKernelGGL((scale_list_kernel<_Float16>), dim3(grid), dim3(256), 0, st, l, static_cast<const float *>(scalar));
    const hipError_t err = hipGetLastError();
    return err == hipSuccess ? PBR_OK : 1000 + (int)err;
}

}  // extern "C"
