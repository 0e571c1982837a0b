// blend.hip -- material blending pre-stage: the per-pixel pass in
// front of the BRDF in /root/reference/examples/example_blend.py:14-16.
//
// Replaces, from /root/reference/pypbr/blending/functional.py:
//   blend_with_mask   :64-116   out = mask * map1 + (1 - mask) * map2 for every map,
//   _blend_normals    :119-145  normals: normalise both, blend, re-normalise,
//   blend_on_height / blend_on_properties :148-239  mask = sigmoid((p1 + shift - p2) / (width + 1e-6)),
//   blend_with_gradient :242-286  mask = linspace(0, 1) along x or y.
// All HBM streams: one lane per pixel walks the map's channels (planar (C,H,W) fp32).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/pbr_hip.h"
#include "brdf_math.hpp"
#include "stream_shape.hpp"

namespace pbr {

typedef float bf4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void blend_normal_one(float w, float ax, float ay, float az, float bx, float by, float bz, float &ox, float &oy, float &oz) {
    const float iw = 1.0f - w;
    const Vec3 a = {ax, ay, az}, b = {bx, by, bz};
    const float ra = rsq(fmaxf(dot(a, a), 1e-24f)), rb = rsq(fmaxf(dot(b, b), 1e-24f));
    const Vec3 c = {fmaf(w, a.x * ra, iw * (b.x * rb)), fmaf(w, a.y * ra, iw * (b.y * rb)), fmaf(w, a.z * ra, iw * (b.z * rb))};
    const float rc = rsq(fmaxf(dot(c, c), 1e-24f));
    ox = c.x * rc; oy = c.y * rc; oz = c.z * rc;
}

// map1, map2: [C][P]; mask: [P]; out: [C][P].  NORMAL: F.normalize both, blend, F.normalize.  vec_ok: P % 4 == 0 and
// 16-byte aligned planes -> 4 pixels per lane, 16-byte streaming accesses.
template <bool NORMAL>
__global__ __launch_bounds__(256) void blend_kernel(const float *__restrict__ m1, const float *__restrict__ m2,
                                                    const float *__restrict__ mask, float *__restrict__ out,
                                                    int channels, int64_t P, int vec_ok) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (vec_ok) {
        const int64_t P4 = P / 4;
        const bf4 *a4 = reinterpret_cast<const bf4 *>(m1), *b4 = reinterpret_cast<const bf4 *>(m2), *k4 = reinterpret_cast<const bf4 *>(mask);
        bf4 *o4 = reinterpret_cast<bf4 *>(out);
        for (int64_t q = tid; q < P4; q += stride) {
            const bf4 w = __builtin_nontemporal_load(k4 + q);
            if (NORMAL) {
                const bf4 ax = __builtin_nontemporal_load(a4 + q), ay = __builtin_nontemporal_load(a4 + P4 + q), az = __builtin_nontemporal_load(a4 + 2 * P4 + q);
                const bf4 bx = __builtin_nontemporal_load(b4 + q), by = __builtin_nontemporal_load(b4 + P4 + q), bz = __builtin_nontemporal_load(b4 + 2 * P4 + q);
                bf4 ox, oy, oz;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float x, y, z;
                    blend_normal_one(w[j], ax[j], ay[j], az[j], bx[j], by[j], bz[j], x, y, z);
                    ox[j] = x; oy[j] = y; oz[j] = z;
                }
                __builtin_nontemporal_store(ox, o4 + q); __builtin_nontemporal_store(oy, o4 + P4 + q); __builtin_nontemporal_store(oz, o4 + 2 * P4 + q);
            } else {
                for (int ch = 0; ch < channels; ++ch) {
                    const bf4 a = __builtin_nontemporal_load(a4 + ch * P4 + q), b = __builtin_nontemporal_load(b4 + ch * P4 + q);
                    bf4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = fmaf(w[j], a[j], (1.0f - w[j]) * b[j]);
                    __builtin_nontemporal_store(o, o4 + ch * P4 + q);
                }
            }
        }
        return;
    }
    for (int64_t p = tid; p < P; p += stride) {
        const float w = mask[p], iw = 1.0f - w;
        if (NORMAL) {
            blend_normal_one(w, m1[p], m1[P + p], m1[2 * P + p], m2[p], m2[P + p], m2[2 * P + p], out[p], out[P + p], out[2 * P + p]);
        } else {
            for (int ch = 0; ch < channels; ++ch) out[ch * P + p] = fmaf(w, m1[ch * P + p], iw * m2[ch * P + p]);
        }
    }
}

// V floats of one plane: V = 4 -> one 16-byte streaming access, V = 1 -> a dword.
template <int V> struct Run;
template <> struct Run<4> {
    static __device__ __forceinline__ void ld(const float *p, int64_t i, float v[4]) {
        const bf4 t = __builtin_nontemporal_load(reinterpret_cast<const bf4 *>(p + i));
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    static __device__ __forceinline__ void st(float *p, int64_t i, const float v[4]) {
        __builtin_nontemporal_store(bf4{v[0], v[1], v[2], v[3]}, reinterpret_cast<bf4 *>(p + i));
    }
};
template <> struct Run<1> {
    static __device__ __forceinline__ void ld(const float *p, int64_t i, float v[1]) { v[0] = p[i]; }
    static __device__ __forceinline__ void st(float *p, int64_t i, const float v[1]) { p[i] = v[0]; }
};

// Gradient of blend_kernel (what autograd derives from functional.py:103-110 / :119-145): g1, g2 [C][P] (NULL = not wanted),
// gmask [P] (NULL = not wanted; `accumulate`: added to what is there -- the mask is shared by every map of a material).
// V pixels per lane (4: P % 4 == 0 and 16-byte aligned planes; 4096^2, 3 channels: 241 -> 207 us = 5.5 TB/s).
template <bool NORMAL, int V>
__global__ __launch_bounds__(256) void blend_backward_kernel(const float *__restrict__ m1, const float *__restrict__ m2,
                                                             const float *__restrict__ mask, const float *__restrict__ gout,
                                                             float *__restrict__ g1, float *__restrict__ g2, float *__restrict__ gmask,
                                                             int channels, int64_t P, int accumulate) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * V;
    for (int64_t p = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * V; p < P; p += stride) {
        float w[V], gw[V];
        Run<V>::ld(mask, p, w);
#pragma unroll
        for (int j = 0; j < V; ++j) gw[j] = 0.0f;
        if (NORMAL) {
            float av[3][V], bv[3][V], gv[3][V], o1[3][V], o2[3][V];
#pragma unroll
            for (int c = 0; c < 3; ++c) { Run<V>::ld(m1, c * P + p, av[c]); Run<V>::ld(m2, c * P + p, bv[c]); Run<V>::ld(gout, c * P + p, gv[c]); }
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const float iw = 1.0f - w[j];
                const Vec3 a = {av[0][j], av[1][j], av[2][j]}, b = {bv[0][j], bv[1][j], bv[2][j]}, g = {gv[0][j], gv[1][j], gv[2][j]};
                const float ra = rsq(fmaxf(dot(a, a), 1e-24f)), rb = rsq(fmaxf(dot(b, b), 1e-24f));
                const Vec3 ah = {a.x * ra, a.y * ra, a.z * ra}, bh = {b.x * rb, b.y * rb, b.z * rb};
                const Vec3 c = {fmaf(w[j], ah.x, iw * bh.x), fmaf(w[j], ah.y, iw * bh.y), fmaf(w[j], ah.z, iw * bh.z)};
                const float rc = rsq(fmaxf(dot(c, c), 1e-24f));
                const Vec3 o = {c.x * rc, c.y * rc, c.z * rc};
                const float og = dot(o, g);                                  // F.normalize: (g - o (o.g)) / |c|
                const Vec3 gc = {(g.x - o.x * og) * rc, (g.y - o.y * og) * rc, (g.z - o.z * og) * rc};
                gw[j] = gc.x * (ah.x - bh.x) + gc.y * (ah.y - bh.y) + gc.z * (ah.z - bh.z);
                const Vec3 ga = {w[j] * gc.x, w[j] * gc.y, w[j] * gc.z};
                const float da = dot(ah, ga);
                o1[0][j] = (ga.x - ah.x * da) * ra; o1[1][j] = (ga.y - ah.y * da) * ra; o1[2][j] = (ga.z - ah.z * da) * ra;
                const Vec3 gb = {iw * gc.x, iw * gc.y, iw * gc.z};
                const float db = dot(bh, gb);
                o2[0][j] = (gb.x - bh.x * db) * rb; o2[1][j] = (gb.y - bh.y * db) * rb; o2[2][j] = (gb.z - bh.z * db) * rb;
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                if (g1) Run<V>::st(g1, c * P + p, o1[c]);
                if (g2) Run<V>::st(g2, c * P + p, o2[c]);
            }
        } else {
            for (int ch = 0; ch < channels; ++ch) {
                float g[V], a[V], b[V], o1[V], o2[V];
                Run<V>::ld(gout, ch * P + p, g); Run<V>::ld(m1, ch * P + p, a); Run<V>::ld(m2, ch * P + p, b);
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    o1[j] = w[j] * g[j];
                    o2[j] = (1.0f - w[j]) * g[j];
                    gw[j] = fmaf(g[j], a[j] - b[j], gw[j]);
                }
                if (g1) Run<V>::st(g1, ch * P + p, o1);
                if (g2) Run<V>::st(g2, ch * P + p, o2);
            }
        }
        if (gmask) {
            if (accumulate) {
                float old[V];
                Run<V>::ld(gmask, p, old);
#pragma unroll
                for (int j = 0; j < V; ++j) gw[j] += old[j];
            }
            Run<V>::st(gmask, p, gw);
        }
    }
}

// torch.sigmoid((p1 + shift - p2) / (width + 1e-6))
__global__ __launch_bounds__(256) void sigmoid_mask_kernel(const float *__restrict__ p1, const float *__restrict__ p2,
                                                           float *__restrict__ mask, int64_t n, float shift, float inv_width) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float x = ((p1[i] + shift) - p2[i]) * inv_width;
        mask[i] = rcp(1.0f + exp2_hw(-1.44269504f * x));           // 1 / (1 + e^-x)
    }
}

// d/d(p1, p2) of the mask above, from the mask itself: sigma' = sigma (1 - sigma); g_p1 = g sigma' / (width + 1e-6), g_p2 = -g_p1
__global__ __launch_bounds__(256) void sigmoid_mask_backward_kernel(const float *__restrict__ mask, const float *__restrict__ gout,
                                                                    float *__restrict__ g1, float *__restrict__ g2, int64_t n, float inv_width) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float sg = mask[i], g = gout[i] * (sg * (1.0f - sg)) * inv_width;
        if (g1) g1[i] = g;
        if (g2) g2[i] = -g;
    }
}

// torch.linspace(0, 1, n) along x (horizontal) or y (vertical), two-ended evaluation
__global__ __launch_bounds__(256) void gradient_mask_kernel(float *__restrict__ mask, int H, int W, int vertical) {
    const int64_t total = (int64_t)H * W, stride = (int64_t)gridDim.x * blockDim.x;
    const int n = vertical ? H : W;
    const float step = n > 1 ? 1.0f / (float)(n - 1) : 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int y = (int)(i / W), x = (int)(i - (int64_t)y * W);
        const int k = vertical ? y : x;
        mask[i] = n == 1 ? 0.0f : (k < n / 2 ? step * (float)k : 1.0f - step * (float)(n - 1 - k));
    }
}

static inline unsigned blend_grid(int64_t items) {
    const int64_t blocks = (items + 255) / 256, cap = 256 * 16;
    return (unsigned)(blocks < 1 ? 1 : (blocks > cap ? cap : blocks));
}

static inline int blend_status() {
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? PBR_OK : 1000 + (int)e;
}

}  // namespace pbr

extern "C" {

int pbr_blend_maps(const void *map1, const void *map2, const void *mask, void *out, int32_t channels, int64_t pixels,
                   int is_normal, void *stream) {
    using namespace pbr;
    if (!map1 || !map2 || !mask || !out) return PBR_ERR_NULL_MAP;
    if (channels < 1 || pixels < 1) return PBR_ERR_SHAPE;
    if (is_normal && channels != 3) return PBR_ERR_CHANNELS;
    hipStream_t s = static_cast<hipStream_t>(stream);
    auto a = static_cast<const float *>(map1), b = static_cast<const float *>(map2), m = static_cast<const float *>(mask);
    const int vec_ok = pixels % 4 == 0 && ((reinterpret_cast<uintptr_t>(map1) | reinterpret_cast<uintptr_t>(map2) | reinterpret_cast<uintptr_t>(mask) |
                                            reinterpret_cast<uintptr_t>(out)) & 15u) == 0;
    const StreamShape sh = stream_shape((size_t)(vec_ok ? pixels / 4 : pixels), is_normal ? kShapeBlendNormal : kShapeBlend);
    if (is_normal) hipLaunchKernelGGL(blend_kernel<true>, dim3(sh.grid), dim3(sh.block), sh.lds, s, a, b, m, static_cast<float *>(out), 3, pixels, vec_ok);
    else hipLaunchKernelGGL(blend_kernel<false>, dim3(sh.grid), dim3(sh.block), sh.lds, s, a, b, m, static_cast<float *>(out), (int)channels, pixels, vec_ok);
    return blend_status();
}

int pbr_blend_sigmoid_mask(const void *prop1, const void *prop2, void *mask, int64_t n, float shift, float blend_width,
                           void *stream) {
    using namespace pbr;
    if (!prop1 || !prop2 || !mask) return PBR_ERR_NULL_MAP;
    if (n < 1) return PBR_ERR_SHAPE;
    const StreamShape sh = stream_shape((size_t)n, kShapeMask);
    hipLaunchKernelGGL(sigmoid_mask_kernel, dim3(sh.grid), dim3(sh.block), sh.lds, static_cast<hipStream_t>(stream),
                       static_cast<const float *>(prop1), static_cast<const float *>(prop2), static_cast<float *>(mask), n,
                       shift, 1.0f / (blend_width + 1e-6f));
    return blend_status();
}

int pbr_blend_sigmoid_mask_backward(const void *mask, const void *grad_out, void *g_prop1, void *g_prop2, int64_t n, float blend_width,
                                    void *stream) {
    using namespace pbr;
    if (!mask || !grad_out) return PBR_ERR_NULL_MAP;
    if (n < 1) return PBR_ERR_SHAPE;
    if (!g_prop1 && !g_prop2) return PBR_OK;
    hipLaunchKernelGGL(sigmoid_mask_backward_kernel, dim3(blend_grid(n)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const float *>(mask), static_cast<const float *>(grad_out), static_cast<float *>(g_prop1),
                       static_cast<float *>(g_prop2), n, 1.0f / (blend_width + 1e-6f));
    return blend_status();
}

int pbr_blend_gradient_mask(void *mask, int32_t height, int32_t width, int vertical, void *stream) {
    using namespace pbr;
    if (!mask) return PBR_ERR_NULL_MAP;
    if (height < 1 || width < 1) return PBR_ERR_SHAPE;
    hipLaunchKernelGGL(gradient_mask_kernel, dim3(blend_grid((int64_t)height * width)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), static_cast<float *>(mask), (int)height, (int)width, vertical);
    return blend_status();
}

int pbr_blend_maps_backward(const void *map1, const void *map2, const void *mask, const void *grad_out, void *g_map1, void *g_map2,
                            void *g_mask, int32_t channels, int64_t pixels, int is_normal, int accumulate_mask, void *stream) {
    using namespace pbr;
    if (!map1 || !map2 || !mask || !grad_out) return PBR_ERR_NULL_MAP;
    if (channels < 1 || pixels < 1) return PBR_ERR_SHAPE;
    if (is_normal && channels != 3) return PBR_ERR_CHANNELS;
    const int64_t blocks = (pixels + 255) / 256;
    const dim3 grid((unsigned)(blocks > 256 * 16 ? 256 * 16 : blocks));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const float *a = static_cast<const float *>(map1), *b = static_cast<const float *>(map2), *k = static_cast<const float *>(mask);
    const float *g = static_cast<const float *>(grad_out);
    float *ga = static_cast<float *>(g_map1), *gb = static_cast<float *>(g_map2), *gk = static_cast<float *>(g_mask);
    bool vec = pixels % 4 == 0;
    for (const void *p : {map1, map2, mask, grad_out, (const void *)g_map1, (const void *)g_map2, (const void *)g_mask})
        if (p && (reinterpret_cast<uintptr_t>(p) & 15u)) vec = false;
    if (vec) {
        const StreamShape sh = stream_shape((size_t)(pixels / 4), kShapeBlendBwd);
        if (is_normal) hipLaunchKernelGGL((blend_backward_kernel<true, 4>), dim3(sh.grid), dim3(sh.block), sh.lds, s, a, b, k, g, ga, gb, gk, (int)channels, pixels, accumulate_mask);
        else hipLaunchKernelGGL((blend_backward_kernel<false, 4>), dim3(sh.grid), dim3(sh.block), sh.lds, s, a, b, k, g, ga, gb, gk, (int)channels, pixels, accumulate_mask);
    } else if (is_normal) {
        hipLaunchKernelGGL((blend_backward_kernel<true, 1>), grid, dim3(256), 0, s, a, b, k, g, ga, gb, gk, (int)channels, pixels, accumulate_mask);
    } else {
        hipLaunchKernelGGL((blend_backward_kernel<false, 1>), grid, dim3(256), 0, s, a, b, k, g, ga, gb, gk, (int)channels, pixels, accumulate_mask);
    }
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? PBR_OK : 1000 + (int)e;
}

}  // extern "C"
