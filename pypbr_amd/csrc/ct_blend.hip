// ct_blend.hip -- launcher of the fused blend + evaluate kernels (C ABI: pbr_cook_torrance_blend); device code in ct_blend.hpp.
#include "ct_blend.hpp"
#include "ct_blend_backward.hpp"
#include "ct_launch.hpp"

namespace pbr {

// One flag per material: does the blended normal map have a negative component anywhere (base.py:212)?
// Grid-stride over the B * P source pixels; a workgroup that finds its material's flag already set skips the pixel
// block, so for real normal maps (negative components everywhere) the pass costs a launch, not a read of 7 planes.
__global__ __launch_bounds__(256) void blend_normal_sign_kernel(const float *__restrict__ n1, const float *__restrict__ n2,
                                                                const float *__restrict__ mask, int64_t n1_bs, int64_t n1_cs,
                                                                int64_t n2_bs, int64_t n2_cs, int64_t k_bs, int64_t P,
                                                                int64_t total, int *__restrict__ flag) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t bi = i / P, px = i - bi * P;
        if (__hip_atomic_load(flag + bi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) continue;   // bypasses the CU's L1
        const float wj = mask[bi * k_bs + px], iw = 1.0f - wj;
        const float *pa = n1 + bi * n1_bs + px, *pb = n2 + bi * n2_bs + px;
        const Vec3 a = {pa[0], pa[n1_cs], pa[2 * n1_cs]}, b = {pb[0], pb[n2_cs], pb[2 * n2_cs]};
        const float ra = rsq(fmaxf(dot(a, a), 1e-24f)), rb = rsq(fmaxf(dot(b, b), 1e-24f));
        const Vec3 c = {fmaf(wj, a.x * ra, iw * (b.x * rb)), fmaf(wj, a.y * ra, iw * (b.y * rb)), fmaf(wj, a.z * ra, iw * (b.z * rb))};
        const float rc = rsq(fmaxf(dot(c, c), 1e-24f));
        if (c.x * rc < 0.0f || c.y * rc < 0.0f || c.z * rc < 0.0f) flag[bi] = 1;
    }
}

static int check_blend(const pbr_render_desc *d, const pbr_blend_desc *bl, const void *workspace) {
    const int rc = validate(d);
    if (rc != PBR_OK) return rc;
    if (!bl || !workspace) return PBR_ERR_NULL_MAP;
    if (d->map_dtype != PBR_F32 || d->out_dtype != PBR_F32) return PBR_ERR_DTYPE;
    if (!d->normal.data || !bl->albedo.data || !bl->normal.data || !bl->roughness.data || !bl->mask.data)
        return PBR_ERR_NULL_MAP;
    if (d->workflow == PBR_WORKFLOW_SPECULAR ? !bl->specular.data : !bl->metallic.data) return PBR_ERR_WORKFLOW;
    if (bl->sign_mode != PBR_BLEND_SIGN_COMPUTE && bl->sign_mode != PBR_BLEND_SIGN_GIVEN) return PBR_ERR_SHAPE;
    return PBR_OK;
}

// Sets flag[b] = 1 where the blended normal of material b has a negative component, over the rows the descriptor
// holds: the whole (map_height, map_width) source maps when they are tiled, else the band's `height` rows.
static int launch_normal_sign(const pbr_render_desc *d, const pbr_blend_desc *bl, void *workspace, hipStream_t st) {
    const bool tiled = is_tiled(d);
    const int64_t P = tiled ? (int64_t)d->map_height * d->map_width : (int64_t)d->height * d->width;
    const int64_t total = P * d->batch, blocks = (total + 255) / 256;
    hipLaunchKernelGGL(blend_normal_sign_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, st,
                       static_cast<const float *>(d->normal.data), static_cast<const float *>(bl->normal.data),
                       static_cast<const float *>(bl->mask.data), d->normal.batch_stride, d->normal.channel_stride,
                       bl->normal.batch_stride, bl->normal.channel_stride, bl->mask.batch_stride, P, total,
                       static_cast<int *>(workspace));
    const hipError_t err = hipGetLastError();
    return err == hipSuccess ? PBR_OK : 1000 + (int)err;
}

static void fill_blend(const pbr_blend_desc *bl, const void *workspace, KBlend &b) {
    std::memset(&b, 0, sizeof(b));
    b.albedo = bl->albedo.data; b.normal = bl->normal.data; b.rough = bl->roughness.data;
    b.metal = bl->metallic.data; b.spec = bl->specular.data;
    b.a_bs = bl->albedo.batch_stride; b.a_cs = bl->albedo.channel_stride;
    b.n_bs = bl->normal.batch_stride; b.n_cs = bl->normal.channel_stride;
    b.r_bs = bl->roughness.batch_stride; b.m_bs = bl->metallic.batch_stride;
    b.s_bs = bl->specular.batch_stride; b.s_cs = bl->specular.channel_stride;
    b.mask = static_cast<const float *>(bl->mask.data); b.k_bs = bl->mask.batch_stride;
    b.normal_signed = static_cast<const int *>(workspace);
}

}  // namespace pbr

extern "C" {

int pbr_cook_torrance_blend_backward(const pbr_render_desc *d, const pbr_blend_desc *bl, void *workspace, const void *grad_out,
                                     const pbr_map_grads *g_material1, const pbr_map_grads *g_material2, void *g_mask, void *stream) {
    const pbr::TuningScope tuning(d);
    using namespace pbr;
    const int rc = check_blend(d, bl, workspace);
    if (rc != PBR_OK) return rc;
    if (!grad_out || !g_material1 || !g_material2) return PBR_ERR_NULL_MAP;
    if (nan_light_size(d)) return PBR_ERR_UNSUPPORTED;
    if (is_tiled(d)) {
        // A repeated texel owns the SUM over its repeats: the repeat-inner walk (ct_repeat_backward.hpp) blends once per texel, visits the
        // repeats, and runs the folded gradients through the blend's chain rule -- MAP-sized gradients (round 6).  Whole outputs and row
        // bands that hold a period of the map's rows, one light; the sign flags of tiled maps are always the whole map's.
        if (!repeat_blend_backward_serves(d)) return PBR_ERR_UNSUPPORTED;
        KBlend kb;
        fill_blend(bl, workspace, kb);
        hipStream_t tst = static_cast<hipStream_t>(stream);
        if (bl->sign_mode == PBR_BLEND_SIGN_COMPUTE) {
            if (hipMemsetAsync(workspace, 0, sizeof(int) * (size_t)d->batch, tst) != hipSuccess) return 1000 + (int)hipGetLastError();
            const int src = launch_normal_sign(d, bl, workspace, tst);
            if (src != PBR_OK) return src;
        }
        const bool tspec = d->workflow == PBR_WORKFLOW_SPECULAR;
        const BArgs t1 = {nullptr, g_material1->albedo, g_material1->normal, g_material1->roughness, tspec ? nullptr : g_material1->metallic,
                          tspec ? g_material1->specular : nullptr, nullptr};
        const BBlend t2 = {g_material2->albedo, g_material2->normal, g_material2->roughness, tspec ? nullptr : g_material2->metallic,
                           tspec ? g_material2->specular : nullptr, static_cast<float *>(g_mask)};
        return launch_repeat_blend_backward(d, &kb, grad_out, &t1, &t2, tst);
    }
    if (bl->sign_mode == PBR_BLEND_SIGN_COMPUTE && d->height != d->height_total) return PBR_ERR_UNSUPPORTED;
    const int vec = d->width >= 2 && g_max_vec >= 2 ? 2 : 1;  // two pixels per lane: both materials' raw texels stay live through the chain rule
    KArgs k;
    fill_args(d, vec, k, 6);
    if (k.n_tiles < 0) return PBR_ERR_SHAPE;
    k.o_cs = (int64_t)d->height * d->width; k.o_bs = 3 * k.o_cs;     // grad_out and every gradient plane are contiguous
    k.sbase = 0;
    KBlend b;
    fill_blend(bl, workspace, b);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (bl->sign_mode == PBR_BLEND_SIGN_COMPUTE) {            // the same map-global decision the forward launch took (base.py:212)
        if (hipMemsetAsync(workspace, 0, sizeof(int) * (size_t)d->batch, st) != hipSuccess) return 1000 + (int)hipGetLastError();
        const int src = launch_normal_sign(d, bl, workspace, st);
        if (src != PBR_OK) return src;
    }
    const bool spec = d->workflow == PBR_WORKFLOW_SPECULAR;
    const BArgs g1 = {grad_out, g_material1->albedo, g_material1->normal, g_material1->roughness, spec ? nullptr : g_material1->metallic,
                      spec ? g_material1->specular : nullptr, nullptr};
    const BBlend g2 = {g_material2->albedo, g_material2->normal, g_material2->roughness, spec ? nullptr : g_material2->metallic,
                       spec ? g_material2->specular : nullptr, static_cast<float *>(g_mask)};
    const bool multi = d->n_lights > 1, point = d->light_type == PBR_LIGHT_POINT;
    void (*fn)(const KArgs, const KBlend, const BArgs, const BBlend) = nullptr;
#define PBR_BLEND_BWD(L, W)                                                                                                              \
    fn = vec == 2 ? (multi ? cook_torrance_blend_backward_kernel<L, W, 2, true> : cook_torrance_blend_backward_kernel<L, W, 2, false>)   \
                  : (multi ? cook_torrance_blend_backward_kernel<L, W, 1, true> : cook_torrance_blend_backward_kernel<L, W, 1, false>)
    switch ((point ? 3 : 0) + d->workflow) {
        case 0: PBR_BLEND_BWD(PBR_LIGHT_DIRECTIONAL, PBR_WORKFLOW_METALLIC); break;
        case 1: PBR_BLEND_BWD(PBR_LIGHT_DIRECTIONAL, PBR_WORKFLOW_SPECULAR); break;
        case 2: PBR_BLEND_BWD(PBR_LIGHT_DIRECTIONAL, PBR_WORKFLOW_CONVERTED); break;
        case 3: PBR_BLEND_BWD(PBR_LIGHT_POINT, PBR_WORKFLOW_METALLIC); break;
        case 4: PBR_BLEND_BWD(PBR_LIGHT_POINT, PBR_WORKFLOW_SPECULAR); break;
        default: PBR_BLEND_BWD(PBR_LIGHT_POINT, PBR_WORKFLOW_CONVERTED); break;
    }
#undef PBR_BLEND_BWD
    hipLaunchKernelGGL(fn, dim3((unsigned)k.n_tiles, 1, 1), dim3(64, 1, 1), 0, st, k, b, g1, g2);
    const hipError_t err = hipGetLastError();
    return err == hipSuccess ? PBR_OK : 1000 + (int)err;
}

int pbr_blend_backward_serves(const pbr_render_desc *d) {
    const pbr::TuningScope tuning(d);
    if (pbr::validate(d) != PBR_OK || d->map_dtype != PBR_F32 || d->out_dtype != PBR_F32) return 0;
    return pbr::is_tiled(d) ? (pbr::repeat_blend_backward_serves(d) ? 1 : 0) : 1;
}

int pbr_blend_normal_sign(const pbr_render_desc *d, const pbr_blend_desc *bl, void *workspace, void *stream) {
    const pbr::TuningScope tuning(d);
    using namespace pbr;
    const int rc = check_blend(d, bl, workspace);
    if (rc != PBR_OK) return rc;
    return launch_normal_sign(d, bl, workspace, static_cast<hipStream_t>(stream));
}

int pbr_cook_torrance_blend(const pbr_render_desc *d, const pbr_blend_desc *bl, void *workspace, void *stream) {
    const pbr::TuningScope tuning(d);
    using namespace pbr;
    const int rc = check_blend(d, bl, workspace);
    if (rc != PBR_OK) return rc;
    // "Is the blended normal already signed?" is a property of the WHOLE map (base.py:212).  A row band of an untiled map
    // only holds its own rows, so its flags must come from the caller (pbr_blend_normal_sign over every band, combined).
    if (bl->sign_mode == PBR_BLEND_SIGN_COMPUTE && !is_tiled(d) && d->height != d->height_total) return PBR_ERR_UNSUPPORTED;
    if (nan_light_size(d)) return fill_result_nan(d, static_cast<hipStream_t>(stream));
    int vec = pick_vec(d);                                    // the second material and the mask only need element alignment
    if (vec == 8) vec = 4;
    const bool walk = repeat_inner(d);                        // tiled maps: blended once per texel, evaluated at every repeat (cook_torrance_repeat_blend_kernel)
    KArgs k;
    if (walk) fill_repeat_args(d, k); else fill_args(d, vec, k);
    if (k.n_tiles < 0) return PBR_ERR_SHAPE;
    KBlend b;
    fill_blend(bl, workspace, b);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (bl->sign_mode == PBR_BLEND_SIGN_COMPUTE) {
        // pass 1: one flag per material -- does the blended normal map have a negative component?  (base.py:212)
        if (hipMemsetAsync(workspace, 0, sizeof(int) * (size_t)d->batch, st) != hipSuccess) return 1000 + (int)hipGetLastError();
        const int src = launch_normal_sign(d, bl, workspace, st);
        if (src != PBR_OK) return src;
    }
    // pass 2: blend + evaluate
    const bool multi = d->n_lights > 1, point = d->light_type == PBR_LIGHT_POINT;
    if (walk) {
        void (*wfn)(const KArgs, const KBlend) = nullptr;
#define PBR_BLEND_WALK(L, W) wfn = multi ? cook_torrance_repeat_blend_kernel<L, W, true> : cook_torrance_repeat_blend_kernel<L, W, false>
        switch ((point ? 3 : 0) + d->workflow) {
            case 0: PBR_BLEND_WALK(PBR_LIGHT_DIRECTIONAL, PBR_WORKFLOW_METALLIC); break;
            case 1: PBR_BLEND_WALK(PBR_LIGHT_DIRECTIONAL, PBR_WORKFLOW_SPECULAR); break;
            case 2: PBR_BLEND_WALK(PBR_LIGHT_DIRECTIONAL, PBR_WORKFLOW_CONVERTED); break;
            case 3: PBR_BLEND_WALK(PBR_LIGHT_POINT, PBR_WORKFLOW_METALLIC); break;
            case 4: PBR_BLEND_WALK(PBR_LIGHT_POINT, PBR_WORKFLOW_SPECULAR); break;
            default: PBR_BLEND_WALK(PBR_LIGHT_POINT, PBR_WORKFLOW_CONVERTED); break;
        }
#undef PBR_BLEND_WALK
        hipLaunchKernelGGL(wfn, dim3((unsigned)k.n_tiles, 1, 1), dim3(1u << k.bt_log2, 1, 1), 0, st, k, b);
        const hipError_t werr = hipGetLastError();
        return werr == hipSuccess ? PBR_OK : 1000 + (int)werr;
    }
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

}  // extern "C"
