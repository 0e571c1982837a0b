// map_ops.hip -- stand-alone map conversions (same arithmetic as the fused kernel) and the
// small introspection entry points of the C ABI.  All kernels are HBM-bound streams:
// 16-byte accesses per lane, grid-stride over the element count, non-temporal hints.
//
// Reference functions replaced (paths under /root/reference/pypbr/):
//   utils/functions.py:31-47   srgb_to_linear
//   utils/functions.py:50-66   linear_to_srgb
//   materials/metallic.py:98-108   to_diffuse_specular_material (arithmetic part)
//   materials/diffuse.py:128-147   to_basecolor_metallic_material (arithmetic part)
//   materials/base.py:191-242      _process_normal_map / _compute_normal_map_z_component
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <cstdint>

#include "../../include/pbr_hip.h"
#include "brdf_math.hpp"
#include "stream_shape.hpp"

namespace pbr {

template <typename T> struct Elem;
template <> struct Elem<float> {
    static __device__ __forceinline__ float ld(const void *p, size_t i) { return static_cast<const float *>(p)[i]; }
    static __device__ __forceinline__ void st(void *p, size_t i, float v) { static_cast<float *>(p)[i] = v; }
};
template <> struct Elem<__half> {
    static __device__ __forceinline__ float ld(const void *p, size_t i) { return (float)static_cast<const _Float16 *>(p)[i]; }
    static __device__ __forceinline__ void st(void *p, size_t i, float v) { static_cast<_Float16 *>(p)[i] = (_Float16)v; }
};

// 4 elements per lane when everything is 16-byte (fp32) / 8-byte (fp16) aligned.
template <typename T> struct Quad;
template <> struct Quad<float> {
    typedef float v4 __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ void ld(const void *p, size_t q, float v[4]) {
        v4 t = __builtin_nontemporal_load(reinterpret_cast<const v4 *>(p) + q);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    static __device__ __forceinline__ void st(void *p, size_t q, const float v[4]) {
        v4 t = {v[0], v[1], v[2], v[3]};
        __builtin_nontemporal_store(t, reinterpret_cast<v4 *>(p) + q);
    }
};
template <> struct Quad<__half> {
    typedef _Float16 v4 __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ void ld(const void *p, size_t q, float v[4]) {
        v4 t = __builtin_nontemporal_load(reinterpret_cast<const v4 *>(p) + q);
        v[0] = (float)t.x; v[1] = (float)t.y; v[2] = (float)t.z; v[3] = (float)t.w;
    }
    static __device__ __forceinline__ void st(void *p, size_t q, const float v[4]) {
        v4 t = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
        __builtin_nontemporal_store(t, reinterpret_cast<v4 *>(p) + q);
    }
};

// ---- colour transfer functions ---------------------------------------------------
template <typename T, bool TO_LINEAR>
__global__ __launch_bounds__(256) void colour_kernel(const void *src, void *dst, size_t n, int vec_ok) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nq = vec_ok ? n / 4 : 0;
    for (size_t q = tid; q < nq; q += stride) {
        float v[4];
        Quad<T>::ld(src, q, v);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = TO_LINEAR ? srgb_to_linear(v[j]) : linear_to_srgb(v[j]);
        Quad<T>::st(dst, q, v);
    }
    for (size_t i = nq * 4 + tid; i < n; i += stride) {
        const float x = Elem<T>::ld(src, i);
        Elem<T>::st(dst, i, TO_LINEAR ? srgb_to_linear(x) : linear_to_srgb(x));
    }
}

// ---- metallic -> diffuse/specular (metallic.py:98-108) ---------------------------
template <typename T>
__global__ __launch_bounds__(256) void metallic_to_specular_kernel(const void *__restrict__ albedo, const void *__restrict__ metallic,
                                                                   void *__restrict__ diffuse, void *__restrict__ specular, int batch,
                                                                   int64_t P, int albedo_srgb, int vec_ok) {
    const size_t stride = (size_t)gridDim.x * blockDim.x, tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (vec_ok) {                       // P % 4 == 0, 16-byte accesses: 4 pixels of one material per lane
        const size_t P4 = (size_t)P / 4, total4 = (size_t)batch * P4;
        for (size_t q = tid; q < total4; q += stride) {
            const size_t b = q / P4, pq = q - b * P4;
            // all four loads of the lane first (outputs never alias inputs: the results are new maps), then the six stores: with
            // the loads of a channel behind the previous channel's stores a lane had one load in flight at a time
            float m[4], a[3][4];
            Quad<T>::ld(metallic, q, m);
#pragma unroll
            for (int c = 0; c < 3; ++c) Quad<T>::ld(albedo, (b * 3 + c) * P4 + pq, a[c]);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const size_t o = (b * 3 + c) * P4 + pq;
                float d[4], sp[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float lin = albedo_srgb ? srgb_to_linear(a[c][j]) : a[c][j], om = 1.0f - m[j];
                    d[j] = lin * om;
                    sp[j] = fmaf(lin, m[j], kDielectricF0 * om);
                }
                Quad<T>::st(diffuse, o, d);
                Quad<T>::st(specular, o, sp);
            }
        }
        return;
    }
    const size_t total = (size_t)batch * (size_t)P;
    for (size_t i = tid; i < total; i += stride) {
        const size_t b = i / (size_t)P, p = i - b * (size_t)P;
        const float m = Elem<T>::ld(metallic, i), om = 1.0f - m;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const size_t o = (b * 3 + c) * (size_t)P + p;
            float a = Elem<T>::ld(albedo, o);
            if (albedo_srgb) a = srgb_to_linear(a);
            Elem<T>::st(diffuse, o, a * om);
            Elem<T>::st(specular, o, fmaf(a, m, kDielectricF0 * om));
        }
    }
}

// ---- diffuse/specular -> basecolor/metallic (diffuse.py:128-147) -----------------
// Thresholded selects (den < eps, metallic >= 0.95) make this one discontinuous, so the
// divisions are done with a Newton-refined reciprocal: the quotient then rounds like the
// reference's IEEE division except in rare half-ulp ties.
__device__ __forceinline__ float div_refined(float a, float b) {
    float r = rcp(b);
    r = fmaf(fmaf(-b, r, 1.0f), r, r);
    const float q = a * r;
    return fmaf(fmaf(-b, q, a), r, q);
}

__device__ __forceinline__ void specular_to_metallic_one(float d, float s, int albedo_srgb, float &bc, float &m) {
    const float eps = 1e-6f;
    if (albedo_srgb) d = srgb_to_linear(d);
    const float num = s - kDielectricF0;                           // RAW specular (diffuse.py:120-124)
    const float den = d - kDielectricF0 + eps;
    m = clamp01(div_refined(num, den + eps));
    if (den < eps) m = 0.0f;
    bc = div_refined(d, 1.0f - m + eps);
    if (m >= 0.95f) bc = s;
    bc = clamp01(bc);
}

template <typename T>
__global__ __launch_bounds__(256) void specular_to_metallic_kernel(const void *diffuse, const void *specular,
                                                                   void *basecolor, void *metallic, size_t n,
                                                                   int albedo_srgb, int vec_ok) {
    const size_t stride = (size_t)gridDim.x * blockDim.x, tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nq = vec_ok ? n / 4 : 0;
    for (size_t q = tid; q < nq; q += stride) {
        float d[4], sp[4], bc[4], m[4];
        Quad<T>::ld(diffuse, q, d);
        Quad<T>::ld(specular, q, sp);
#pragma unroll
        for (int j = 0; j < 4; ++j) specular_to_metallic_one(d[j], sp[j], albedo_srgb, bc[j], m[j]);
        Quad<T>::st(basecolor, q, bc);
        Quad<T>::st(metallic, q, m);
    }
    for (size_t i = nq * 4 + tid; i < n; i += stride) {
        float bc, m;
        specular_to_metallic_one(Elem<T>::ld(diffuse, i), Elem<T>::ld(specular, i), albedo_srgb, bc, m);
        Elem<T>::st(basecolor, i, bc);
        Elem<T>::st(metallic, i, m);
    }
}

// ---- normal decode (base.py:191-242) ---------------------------------------------
// Read-only pass (in-place calls only; see the one-pass path below): 16-byte loads, four in flight per lane (a lone dword
// load per iteration left it at 3.7 TB/s); every wave leaves as soon as the flag is set.
template <typename T>
__global__ __launch_bounds__(256) void any_negative_kernel(const void *src, size_t n, int *flag, int vec_ok) {
    const size_t stride = (size_t)gridDim.x * blockDim.x, tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool neg = false;
    const size_t nq = vec_ok ? n / 4 : 0;
    size_t q = tid;
    for (; q + 3 * stride < nq; q += 4 * stride) {
        if (__atomic_load_n(flag, __ATOMIC_RELAXED) != 0) return;   // settled by the probe or by another wave
        float v[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) Quad<T>::ld(src, q + u * stride, v[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) neg |= (v[u][0] < 0.0f) | (v[u][1] < 0.0f) | (v[u][2] < 0.0f) | (v[u][3] < 0.0f);
    }
    for (; q < nq; q += stride) {
        float v[4];
        Quad<T>::ld(src, q, v);
        neg |= (v[0] < 0.0f) | (v[1] < 0.0f) | (v[2] < 0.0f) | (v[3] < 0.0f);
    }
    for (size_t i = nq * 4 + tid; i < n; i += stride) neg |= Elem<T>::ld(src, i) < 0.0f;
    if (__ballot(neg) != 0ull && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

template <typename T, int CH>
__global__ __launch_bounds__(256) void decode_normal_kernel(const void *src, void *dst, int64_t P, const int *flag) {
    const bool keep = CH == 3 && *flag != 0;                        // base.py:212-213
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < (size_t)P; p += stride) {
        float x = Elem<T>::ld(src, p), y = Elem<T>::ld(src, (size_t)P + p), z;
        if (CH == 3) {
            z = Elem<T>::ld(src, 2 * (size_t)P + p);
            if (!keep) { x = fmaf(x, 2.0f, -1.0f); y = fmaf(y, 2.0f, -1.0f); z = fmaf(z, 2.0f, -1.0f); }
        } else {
            x = fmaf(x, 2.0f, -1.0f); y = fmaf(y, 2.0f, -1.0f);     // base.py:235
            z = sqrt_hw(fmaxf(1.0f - (x * x + y * y), 1e-6f));      // base.py:238-240
        }
        if (!keep) {
            const float r = rsq(fmaxf(fmaf(z, z, fmaf(y, y, x * x)), 1e-24f));   // F.normalize
            x *= r; y *= r; z *= r;
        }
        Elem<T>::st(dst, p, x); Elem<T>::st(dst, (size_t)P + p, y); Elem<T>::st(dst, 2 * (size_t)P + p, z);
    }
}

// 3-channel maps in ONE pass over the data when source and destination do not overlap.  Three launches, no host decision:
//  1. normal_probe_kernel looks at 4096 values (64 runs of 64) spread over the map and WRITES the flag (a signed map -- a predicted or a
//     blended normal -- has a negative value among them practically always; an encoded PNG map never);
//  2. decode_normal_speculative_kernel returns at once when the flag is already set; otherwise it decodes as if no value
//     were negative and records exactly whether one was;
//  3. keep_normal_kernel returns at once unless the flag is set, in which case the result is the map as it is
//     (base.py:212-213) -- also after a speculative decode that met a negative value late (rare, costs the second pass).
// The flag is exact in every case: the probe only ever sets it on a negative value it has seen.
template <typename T>
__global__ __launch_bounds__(256) void normal_probe_kernel(const void *src, size_t n, int *flag) {
    // 64 runs of 64 consecutive values, evenly spread over the map: every wave-level load is one 256-byte request
    // (4096 single values 48 KB apart took 9 us, each its own DRAM page)
    const size_t step = n / 64;
    bool neg = false;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const size_t i = ((size_t)k * 4 + (threadIdx.x >> 6)) * step + (threadIdx.x & 63);
        if (i < n) neg |= Elem<T>::ld(src, i) < 0.0f;
    }
    const int any = __syncthreads_or(neg);
    if (threadIdx.x == 0) *flag = any ? 1 : 0;
}

template <typename T>
__global__ __launch_bounds__(256) void decode_normal_speculative_kernel(const void *src, void *dst, int64_t P, int *flag) {
    if (*flag != 0) return;                                        // the probe met a negative value: kept as it is
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    bool neg = false;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < (size_t)P; p += stride) {
        float x = Elem<T>::ld(src, p), y = Elem<T>::ld(src, (size_t)P + p), z = Elem<T>::ld(src, 2 * (size_t)P + p);
        neg |= (x < 0.0f) | (y < 0.0f) | (z < 0.0f);
        x = fmaf(x, 2.0f, -1.0f); y = fmaf(y, 2.0f, -1.0f); z = fmaf(z, 2.0f, -1.0f);
        const float r = rsq(fmaxf(fmaf(z, z, fmaf(y, y, x * x)), 1e-24f));       // F.normalize
        Elem<T>::st(dst, p, x * r); Elem<T>::st(dst, (size_t)P + p, y * r); Elem<T>::st(dst, 2 * (size_t)P + p, z * r);
    }
    if (__ballot(neg) != 0ull && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

template <typename T>
__global__ __launch_bounds__(256) void keep_normal_kernel(const void *src, void *dst, size_t n, const int *flag, int vec_ok) {
    if (*flag == 0) return;
    const size_t stride = (size_t)gridDim.x * blockDim.x, tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nq = vec_ok ? n / 4 : 0;
    for (size_t q = tid; q < nq; q += stride) {
        float v[4];
        Quad<T>::ld(src, q, v);
        Quad<T>::st(dst, q, v);
    }
    for (size_t i = nq * 4 + tid; i < n; i += stride) Elem<T>::st(dst, i, Elem<T>::ld(src, i));
}

// Gradient folding (autograd plumbing of the backward kernel): the backward kernel writes one gradient per OUTPUT
// pixel and material; a map that is shared by the whole batch, or repeated ny x nx times by a fused tile(), owns the
// sum of those.  dst[bo][c][y][x] = sum_{b in group} sum_{ty,tx} src[b][c][ty*h + y][tx*w + x], fixed order.
// Four consecutive columns per lane (w % 4 == 0, 16-byte aligned planes): the terms of a sum are independent 16-byte
// loads, four of them in flight at a time; added in the same fixed order as the one-column form below.
template <typename T>
__global__ __launch_bounds__(256) void fold_gradient_quad_kernel(const void *__restrict__ src, void *__restrict__ dst,
                                                                 int batch, int channels, int h, int w, int ny, int nx, int fold_batch) {
    const int wq = w >> 2;
    const int64_t plane_q = (int64_t)h * wq, total = (int64_t)(fold_batch ? 1 : batch) * channels * plane_q;
    const int64_t W = (int64_t)nx * w, src_plane = (int64_t)ny * h * W;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int reps = ny * nx, terms = (fold_batch ? batch : 1) * reps;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t pq = i % plane_q, t = i / plane_q;
        const int c = (int)(t % channels), bo = (int)(t / channels);
        const int y = (int)(pq / wq), x = (int)(pq - (int64_t)y * wq) * 4;
        const int64_t p0 = ((int64_t)(fold_batch ? 0 : bo) * channels + c) * src_plane + (int64_t)y * W + x;
        auto term = [&](int k) {                         // k = (b, ty, tx) in the order of the sum; in units of 4 elements
            const int b = k / reps, r = k - b * reps, ty = r / nx, tx = r - ty * nx;
            return (size_t)((p0 + (int64_t)b * channels * src_plane + (int64_t)ty * h * W + (int64_t)tx * w) >> 2);
        };
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        int k = 0;
        for (; k + 4 <= terms; k += 4) {
            float v[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) Quad<T>::ld(src, term(k + u), v[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] += v[u][j];
        }
        for (; k < terms; ++k) {
            float v[4];
            Quad<T>::ld(src, term(k), v);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] += v[j];
        }
        Quad<T>::st(dst, (size_t)i, acc);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void fold_gradient_kernel(const void *__restrict__ src, void *__restrict__ dst,
                                                            int batch, int channels, int h, int w, int ny, int nx, int fold_batch) {
    const int64_t plane = (int64_t)h * w, total = (int64_t)(fold_batch ? 1 : batch) * channels * plane;
    const int64_t W = (int64_t)nx * w, src_plane = (int64_t)ny * h * W;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t px = i % plane, t = i / plane;
        const int c = (int)(t % channels), bo = (int)(t / channels);
        const int y = (int)(px / w), x = (int)(px - (int64_t)y * w);
        float acc = 0.0f;
        const int b0 = fold_batch ? 0 : bo, b1 = fold_batch ? batch : bo + 1;
        for (int b = b0; b < b1; ++b) {
            const int64_t p = ((int64_t)b * channels + c) * src_plane;
            for (int ty = 0; ty < ny; ++ty)
                for (int tx = 0; tx < nx; ++tx) acc += Elem<T>::ld(src, (size_t)(p + ((int64_t)ty * h + y) * W + (int64_t)tx * w + x));
        }
        Elem<T>::st(dst, (size_t)i, acc);
    }
}

// Gradient of decode_normal_kernel (MaterialBase._process_normal_map, base.py:191-242) w.r.t. the stored map, with
// torch's conventions: a map that was kept as is passes the gradient through; x*2-1 contributes a factor 2;
// F.normalize projects out the radial component; clamp(1 - x^2 - y^2, min=1e-6) passes where it did not clamp.
template <int CH, int V>
__global__ __launch_bounds__(256) void decode_normal_backward_kernel(const float *__restrict__ src, const float *__restrict__ gout,
                                                                     float *__restrict__ gin, int64_t P, const int *flag) {
    const bool keep = CH == 3 && *flag != 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x * V, Ps = (size_t)P;
    auto ld = [](const float *p, size_t i, float v[V]) {          // V = 4: one 16-byte streaming load (P % 4 == 0, aligned planes)
        if constexpr (V == 4) Quad<float>::ld(p, i / 4, v); else v[0] = p[i];
    };
    auto st = [](float *p, size_t i, const float v[V]) {
        if constexpr (V == 4) Quad<float>::st(p, i / 4, v); else p[i] = v[0];
    };
    for (size_t p = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * V; p < Ps; p += stride) {
        float g[3][V], s[3][V], o[3][V];
#pragma unroll
        for (int c = 0; c < 3; ++c) ld(gout, c * Ps + p, g[c]);
        if (keep) {
#pragma unroll
            for (int c = 0; c < 3; ++c) st(gin, c * Ps + p, g[c]);
            continue;
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) ld(src, c * Ps + p, s[c]);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float gx = g[0][k], gy = g[1][k], gz = g[2][k];
            const float x = fmaf(s[0][k], 2.0f, -1.0f), y = fmaf(s[1][k], 2.0f, -1.0f);
            float z, q = 0.0f;
            if (CH == 3) z = fmaf(s[2][k], 2.0f, -1.0f);
            else { q = 1.0f - (x * x + y * y); z = sqrt_hw(fmaxf(q, 1e-6f)); }
            const float r = rsq(fmaxf(fmaf(z, z, fmaf(y, y, x * x)), 1e-24f));
            const float nx = x * r, ny = y * r, nz = z * r;
            const float radial = fmaf(nz, gz, fmaf(ny, gy, nx * gx));
            const float px = (gx - nx * radial) * r, py = (gy - ny * radial) * r, pz = (gz - nz * radial) * r;
            if (CH == 3) {
                o[0][k] = 2.0f * px; o[1][k] = 2.0f * py; o[2][k] = 2.0f * pz;
            } else {                                                     // z = sqrt(clamp(q)), dz/dx = -x / z where q >= 1e-6
                const float dz = q >= 1e-6f ? -pz * rcp(z) : 0.0f;
                o[0][k] = 2.0f * fmaf(dz, x, px); o[1][k] = 2.0f * fmaf(dz, y, py);
            }
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) st(gin, c * Ps + p, o[c]);
    }
}


// ---- gradients of the colour transfers and of the workflow conversions ------------------------------------------------
// What torch.autograd derives from the reference's plain torch ops (functions.py:31-66, metallic.py:98-108, diffuse.py:128-147),
// so that a rendering loss differentiates through material.to_linear() / to_srgb() / to_diffuse_specular_material() /
// to_basecolor_metallic_material() as it does upstream.  torch's sub-gradient conventions: clamp passes on the closed
// interval, masked assignment / torch.where route the gradient to the selected branch only, pow differentiates as
// e x^(e-1).  Gradients travel in the maps' storage type; arithmetic is fp32.  Same streaming form as the forward kernels.

// d/dx srgb_to_linear(x) = [0 <= x <= 1] * (x <= 0.04045 ? 1/12.92 : 2.4/1.055 ((x + 0.055)/1.055)^1.4); the final clamp
// always passes (its argument lies in [0,1]).
__device__ __forceinline__ float srgb_to_linear_slope(float x) {
    const float t = clamp01(x);
    const float hi = exp2_hw(fmaf(1.4f, log2_hw(t + 0.055f), -0.10814020f /* 1.4 log2 1.055 */)) * 2.2748815f /* 2.4 / 1.055 */;
    const float d = t <= 0.04045f ? 1.0f / 12.92f : hi;
    return (x >= 0.0f && x <= 1.0f) ? d : 0.0f;
}
// d/dx linear_to_srgb(x) = [0 <= x <= 1] * (x <= 0.0031308 ? 12.92 : 1.055/2.4 x^(1/2.4 - 1))
__device__ __forceinline__ float linear_to_srgb_slope(float x) {
    const float c = clamp01(x);
    const float hi = exp2_hw(log2_hw(c) * (1.0f / 2.4f - 1.0f)) * 0.43958333f /* 1.055 / 2.4 */;
    const float d = c <= 0.0031308f ? 12.92f : hi;
    return (x >= 0.0f && x <= 1.0f) ? d : 0.0f;
}

template <typename T, bool TO_LINEAR>
__global__ __launch_bounds__(256) void colour_backward_kernel(const void *src, const void *gout, void *gin, size_t n, int vec_ok) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nq = vec_ok ? n / 4 : 0;
    for (size_t q = tid; q < nq; q += stride) {
        float x[4], g[4];
        Quad<T>::ld(src, q, x);
        Quad<T>::ld(gout, q, g);
#pragma unroll
        for (int j = 0; j < 4; ++j) g[j] *= TO_LINEAR ? srgb_to_linear_slope(x[j]) : linear_to_srgb_slope(x[j]);
        Quad<T>::st(gin, q, g);
    }
    for (size_t i = nq * 4 + tid; i < n; i += stride) {
        const float x = Elem<T>::ld(src, i);
        Elem<T>::st(gin, i, Elem<T>::ld(gout, i) * (TO_LINEAR ? srgb_to_linear_slope(x) : linear_to_srgb_slope(x)));
    }
}

// metallic.py:98-108 backward: diffuse = lin (1 - m), specular = 0.04 (1 - m) + lin m, lin = srgb_to_linear(albedo) | albedo.
//   g_lin = g_d (1 - m) + g_s m;  g_albedo = g_lin * slope;  g_m = sum_c (g_s (lin - 0.04) - g_d lin)
// Either upstream gradient may be absent (NULL: that output was not used), either result may be unwanted (NULL).
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void metallic_to_specular_backward_kernel(const void *albedo, const void *metallic, const void *g_diffuse,
                                                                            const void *g_specular, void *g_albedo, void *g_metallic,
                                                                            int batch, int64_t P, int albedo_srgb) {
    constexpr int V = VEC ? 4 : 1;
    const size_t stride = (size_t)gridDim.x * blockDim.x, tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t PV = (size_t)P / V, total = (size_t)batch * PV;
    auto ld = [](const void *p, size_t i, float v[V]) { if constexpr (VEC) Quad<T>::ld(p, i, v); else v[0] = Elem<T>::ld(p, i); };
    auto st = [](void *p, size_t i, const float v[V]) { if constexpr (VEC) Quad<T>::st(p, i, v); else Elem<T>::st(p, i, v[0]); };
    for (size_t q = tid; q < total; q += stride) {
        const size_t b = q / PV, pq = q - b * PV;
        float m[V], gm[V];
        ld(metallic, q, m);
#pragma unroll
        for (int j = 0; j < V; ++j) gm[j] = 0.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const size_t o = (b * 3 + c) * PV + pq;
            float a[V], gd[V], gs[V], ga[V];
            ld(albedo, o, a);
#pragma unroll
            for (int j = 0; j < V; ++j) { gd[j] = 0.0f; gs[j] = 0.0f; }
            if (g_diffuse) ld(g_diffuse, o, gd);
            if (g_specular) ld(g_specular, o, gs);
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const float lin = albedo_srgb ? srgb_to_linear(a[j]) : a[j];
                const float slope = albedo_srgb ? srgb_to_linear_slope(a[j]) : 1.0f;
                ga[j] = fmaf(gs[j] - gd[j], m[j], gd[j]) * slope;
                gm[j] = fmaf(gs[j], lin - kDielectricF0, fmaf(-gd[j], lin, gm[j]));
            }
            if (g_albedo) st(g_albedo, o, ga);
        }
        if (g_metallic) st(g_metallic, q, gm);
    }
}

// diffuse.py:128-147 backward (elementwise over all three channels; the metallic map it yields has three):
//   num = s - 0.04, den = d - 0.04 + eps, q = num / (den + eps), m0 = clamp(q, 0, 1), m = den < eps ? 0 : m0,
//   bc0 = d / (1 - m + eps), bc1 = m >= 0.95 ? s : bc0, bc = clamp(bc1, 0, 1);  d = srgb_to_linear(diffuse) | diffuse.
// The selects are re-taken with the forward's own arithmetic (div_refined), so forward and backward agree on every branch.
__device__ __forceinline__ void specular_to_metallic_backward_one(float draw, float s, float g_bc, float g_m, int albedo_srgb,
                                                                   float &g_d, float &g_s) {
    const float eps = 1e-6f;
    const float d = albedo_srgb ? srgb_to_linear(draw) : draw;
    const float num = s - kDielectricF0, den = d - kDielectricF0 + eps;
    const float q = div_refined(num, den + eps);
    float m = clamp01(q);
    const bool dead = den < eps;
    if (dead) m = 0.0f;
    const float w = div_refined(1.0f, 1.0f - m + eps);
    const bool metal = m >= 0.95f;
    const float bc1 = metal ? s : d * w;
    const float g_bc1 = (bc1 >= 0.0f && bc1 <= 1.0f) ? g_bc : 0.0f;
    const float g_bc0 = metal ? 0.0f : g_bc1;
    g_s = metal ? g_bc1 : 0.0f;
    float gd = g_bc0 * w;
    const float g_mt = fmaf(g_bc0 * d, w * w, g_m);
    const float g_q = (!dead && q >= 0.0f && q <= 1.0f) ? g_mt : 0.0f;
    const float r = div_refined(1.0f, den + eps);
    g_s = fmaf(g_q, r, g_s);
    gd = fmaf(-g_q * q, r, gd);
    g_d = albedo_srgb ? gd * srgb_to_linear_slope(draw) : gd;
}

template <typename T>
__global__ __launch_bounds__(256) void specular_to_metallic_backward_kernel(const void *diffuse, const void *specular, const void *g_basecolor,
                                                                            const void *g_metallic, void *g_diffuse, void *g_specular,
                                                                            size_t n, int albedo_srgb, int vec_ok) {
    const size_t stride = (size_t)gridDim.x * blockDim.x, tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nq = vec_ok ? n / 4 : 0;
    for (size_t q = tid; q < nq; q += stride) {
        float d[4], sp[4], gb[4] = {0.0f, 0.0f, 0.0f, 0.0f}, gm[4] = {0.0f, 0.0f, 0.0f, 0.0f}, gd[4], gs[4];
        Quad<T>::ld(diffuse, q, d);
        Quad<T>::ld(specular, q, sp);
        if (g_basecolor) Quad<T>::ld(g_basecolor, q, gb);
        if (g_metallic) Quad<T>::ld(g_metallic, q, gm);
#pragma unroll
        for (int j = 0; j < 4; ++j) specular_to_metallic_backward_one(d[j], sp[j], gb[j], gm[j], albedo_srgb, gd[j], gs[j]);
        if (g_diffuse) Quad<T>::st(g_diffuse, q, gd);
        if (g_specular) Quad<T>::st(g_specular, q, gs);
    }
    for (size_t i = nq * 4 + tid; i < n; i += stride) {
        float gd, gs;
        specular_to_metallic_backward_one(Elem<T>::ld(diffuse, i), Elem<T>::ld(specular, i), g_basecolor ? Elem<T>::ld(g_basecolor, i) : 0.0f,
                                          g_metallic ? Elem<T>::ld(g_metallic, i) : 0.0f, albedo_srgb, gd, gs);
        if (g_diffuse) Elem<T>::st(g_diffuse, i, gd);
        if (g_specular) Elem<T>::st(g_specular, i, gs);
    }
}


static inline unsigned stream_grid(size_t work_items) {
    const size_t blocks = (work_items + 255) / 256;
    const size_t cap = 256 * 8;                                     // 256 CUs x 8 blocks, grid-stride beyond
    return (unsigned)(blocks < 1 ? 1 : (blocks > cap ? cap : blocks));
}

static inline int hip_status() {
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? PBR_OK : 1000 + (int)e;
}

static inline bool is_aligned(const void *p, size_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

template <typename T>
static int fold_launch(const void *src, void *dst, int32_t batch, int32_t channels, int32_t h, int32_t w, int32_t ny, int32_t nx,
                       int fold_batch, void *stream) {
    const size_t items = (size_t)(fold_batch ? 1 : batch) * channels * h * w, al = 4 * sizeof(T);
    if (w % 4 == 0 && is_aligned(src, al) && is_aligned(dst, al)) {
        const StreamShape sh = stream_shape(items / 4, kShapeFold);
        hipLaunchKernelGGL((fold_gradient_quad_kernel<T>), dim3(sh.grid), dim3(sh.block), sh.lds, static_cast<hipStream_t>(stream),
                           src, dst, (int)batch, (int)channels, (int)h, (int)w, (int)ny, (int)nx, fold_batch);
        return hip_status();
    }
    hipLaunchKernelGGL((fold_gradient_kernel<T>), dim3(stream_grid(items)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       src, dst, (int)batch, (int)channels, (int)h, (int)w, (int)ny, (int)nx, fold_batch);
    return hip_status();
}

}  // namespace pbr

extern "C" {

static int colour_launch(const void *src, void *dst, size_t n, int dtype, void *stream, bool to_linear) {
    using namespace pbr;
    if (!src || !dst) return PBR_ERR_NULL_MAP;
    if (dtype != PBR_F32 && dtype != PBR_F16) return PBR_ERR_DTYPE;
    if (n == 0) return PBR_OK;
    const size_t al = dtype == PBR_F32 ? 16 : 8;
    const int vec_ok = is_aligned(src, al) && is_aligned(dst, al);
    const StreamShape sh = stream_shape(vec_ok ? (n + 3) / 4 : n, kShapeColour);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == PBR_F32) {
        if (to_linear) hipLaunchKernelGGL((colour_kernel<float, true>), dim3(sh.grid), dim3(sh.block), sh.lds, s, src, dst, n, vec_ok);
        else hipLaunchKernelGGL((colour_kernel<float, false>), dim3(sh.grid), dim3(sh.block), sh.lds, s, src, dst, n, vec_ok);
    } else {
        if (to_linear) hipLaunchKernelGGL((colour_kernel<__half, true>), dim3(sh.grid), dim3(sh.block), sh.lds, s, src, dst, n, vec_ok);
        else hipLaunchKernelGGL((colour_kernel<__half, false>), dim3(sh.grid), dim3(sh.block), sh.lds, s, src, dst, n, vec_ok);
    }
    return hip_status();
}

int pbr_srgb_to_linear(const void *src, void *dst, size_t n, int dtype, void *stream) {
    return colour_launch(src, dst, n, dtype, stream, true);
}

int pbr_linear_to_srgb(const void *src, void *dst, size_t n, int dtype, void *stream) {
    return colour_launch(src, dst, n, dtype, stream, false);
}

int pbr_metallic_to_specular(const void *albedo, const void *metallic, void *diffuse, void *specular,
                             int32_t batch, int64_t pixels, int albedo_is_srgb, int dtype, void *stream) {
    using namespace pbr;
    if (!albedo || !metallic || !diffuse || !specular) return PBR_ERR_NULL_MAP;
    if (batch < 1 || pixels < 1) return PBR_ERR_SHAPE;
    if (dtype != PBR_F32 && dtype != PBR_F16) return PBR_ERR_DTYPE;
    const size_t al = dtype == PBR_F32 ? 16 : 8;
    const int vec_ok = pixels % 4 == 0 && is_aligned(albedo, al) && is_aligned(metallic, al) && is_aligned(diffuse, al) && is_aligned(specular, al);
    const size_t items = (size_t)batch * (size_t)pixels;
    const StreamShape sh = stream_shape(vec_ok ? items / 4 : items, kShapeM2S);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == PBR_F32)
        hipLaunchKernelGGL((metallic_to_specular_kernel<float>), dim3(sh.grid), dim3(sh.block), sh.lds, s, albedo, metallic, diffuse, specular, (int)batch, pixels, albedo_is_srgb, vec_ok);
    else
        hipLaunchKernelGGL((metallic_to_specular_kernel<__half>), dim3(sh.grid), dim3(sh.block), sh.lds, s, albedo, metallic, diffuse, specular, (int)batch, pixels, albedo_is_srgb, vec_ok);
    return hip_status();
}

int pbr_specular_to_metallic(const void *diffuse, const void *specular, void *basecolor, void *metallic,
                             size_t n, int albedo_is_srgb, int dtype, void *stream) {
    using namespace pbr;
    if (!diffuse || !specular || !basecolor || !metallic) return PBR_ERR_NULL_MAP;
    if (dtype != PBR_F32 && dtype != PBR_F16) return PBR_ERR_DTYPE;
    if (n == 0) return PBR_OK;
    const size_t al = dtype == PBR_F32 ? 16 : 8;
    const int vec_ok = is_aligned(diffuse, al) && is_aligned(specular, al) && is_aligned(basecolor, al) && is_aligned(metallic, al);
    const StreamShape sh = stream_shape(vec_ok ? (n + 3) / 4 : n, kShapeS2M);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == PBR_F32)
        hipLaunchKernelGGL((specular_to_metallic_kernel<float>), dim3(sh.grid), dim3(sh.block), sh.lds, s, diffuse, specular, basecolor, metallic, n, albedo_is_srgb, vec_ok);
    else
        hipLaunchKernelGGL((specular_to_metallic_kernel<__half>), dim3(sh.grid), dim3(sh.block), sh.lds, s, diffuse, specular, basecolor, metallic, n, albedo_is_srgb, vec_ok);
    return hip_status();
}


static int colour_backward_launch(const void *src, const void *gout, void *gin, size_t n, int dtype, void *stream, bool to_linear) {
    using namespace pbr;
    if (!src || !gout || !gin) return PBR_ERR_NULL_MAP;
    if (dtype != PBR_F32 && dtype != PBR_F16) return PBR_ERR_DTYPE;
    if (n == 0) return PBR_OK;
    const size_t al = dtype == PBR_F32 ? 16 : 8;
    const int vec_ok = is_aligned(src, al) && is_aligned(gout, al) && is_aligned(gin, al);
    const StreamShape sh = stream_shape(vec_ok ? (n + 3) / 4 : n, kShapeColourBwd);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == PBR_F32) {
        if (to_linear) hipLaunchKernelGGL((colour_backward_kernel<float, true>), dim3(sh.grid), dim3(sh.block), sh.lds, s, src, gout, gin, n, vec_ok);
        else hipLaunchKernelGGL((colour_backward_kernel<float, false>), dim3(sh.grid), dim3(sh.block), sh.lds, s, src, gout, gin, n, vec_ok);
    } else {
        if (to_linear) hipLaunchKernelGGL((colour_backward_kernel<__half, true>), dim3(sh.grid), dim3(sh.block), sh.lds, s, src, gout, gin, n, vec_ok);
        else hipLaunchKernelGGL((colour_backward_kernel<__half, false>), dim3(sh.grid), dim3(sh.block), sh.lds, s, src, gout, gin, n, vec_ok);
    }
    return hip_status();
}

int pbr_srgb_to_linear_backward(const void *src, const void *grad_out, void *grad_in, size_t n, int dtype, void *stream) {
    return colour_backward_launch(src, grad_out, grad_in, n, dtype, stream, true);
}

int pbr_linear_to_srgb_backward(const void *src, const void *grad_out, void *grad_in, size_t n, int dtype, void *stream) {
    return colour_backward_launch(src, grad_out, grad_in, n, dtype, stream, false);
}

int pbr_metallic_to_specular_backward(const void *albedo, const void *metallic, const void *g_diffuse, const void *g_specular,
                                      void *g_albedo, void *g_metallic, int32_t batch, int64_t pixels, int albedo_is_srgb, int dtype,
                                      void *stream) {
    using namespace pbr;
    if (!albedo || !metallic) return PBR_ERR_NULL_MAP;
    if (batch < 1 || pixels < 1) return PBR_ERR_SHAPE;
    if (dtype != PBR_F32 && dtype != PBR_F16) return PBR_ERR_DTYPE;
    if (!g_albedo && !g_metallic) return PBR_OK;
    const size_t al = dtype == PBR_F32 ? 16 : 8;
    auto ok = [&](const void *p) { return !p || is_aligned(p, al); };
    const bool vec = pixels % 4 == 0 && ok(albedo) && ok(metallic) && ok(g_diffuse) && ok(g_specular) && ok(g_albedo) && ok(g_metallic);
    const size_t items = (size_t)batch * (size_t)pixels;
    const StreamShape sh = stream_shape(vec ? items / 4 : items, kShapeM2SBwd);
    hipStream_t s = static_cast<hipStream_t>(stream);
#define PBR_M2S_BWD(T, V) hipLaunchKernelGGL((metallic_to_specular_backward_kernel<T, V>), dim3(sh.grid), dim3(sh.block), sh.lds, s, albedo, metallic, \
                                             g_diffuse, g_specular, g_albedo, g_metallic, (int)batch, pixels, albedo_is_srgb)
    if (dtype == PBR_F32) { if (vec) PBR_M2S_BWD(float, true); else PBR_M2S_BWD(float, false); }
    else { if (vec) PBR_M2S_BWD(__half, true); else PBR_M2S_BWD(__half, false); }
#undef PBR_M2S_BWD
    return hip_status();
}

int pbr_specular_to_metallic_backward(const void *diffuse, const void *specular, const void *g_basecolor, const void *g_metallic,
                                      void *g_diffuse, void *g_specular, size_t n, int albedo_is_srgb, int dtype, void *stream) {
    using namespace pbr;
    if (!diffuse || !specular) return PBR_ERR_NULL_MAP;
    if (dtype != PBR_F32 && dtype != PBR_F16) return PBR_ERR_DTYPE;
    if (n == 0 || (!g_diffuse && !g_specular)) return PBR_OK;
    const size_t al = dtype == PBR_F32 ? 16 : 8;
    auto ok = [&](const void *p) { return !p || is_aligned(p, al); };
    const int vec_ok = ok(diffuse) && ok(specular) && ok(g_basecolor) && ok(g_metallic) && ok(g_diffuse) && ok(g_specular);
    const StreamShape sh = stream_shape(vec_ok ? (n + 3) / 4 : n, kShapeS2MBwd);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == PBR_F32)
        hipLaunchKernelGGL((specular_to_metallic_backward_kernel<float>), dim3(sh.grid), dim3(sh.block), sh.lds, s, diffuse, specular, g_basecolor, g_metallic,
                           g_diffuse, g_specular, n, albedo_is_srgb, vec_ok);
    else
        hipLaunchKernelGGL((specular_to_metallic_backward_kernel<__half>), dim3(sh.grid), dim3(sh.block), sh.lds, s, diffuse, specular, g_basecolor, g_metallic,
                           g_diffuse, g_specular, n, albedo_is_srgb, vec_ok);
    return hip_status();
}

int pbr_fold_gradient_typed(const void *src, void *dst, int32_t batch, int32_t channels, int32_t h, int32_t w, int32_t ny,
                            int32_t nx, int fold_batch, int dtype, void *stream) {
    if (!src || !dst) return PBR_ERR_NULL_MAP;
    if (batch < 1 || channels < 1 || h < 1 || w < 1 || ny < 1 || nx < 1) return PBR_ERR_SHAPE;
    if (dtype == PBR_F32) return pbr::fold_launch<float>(src, dst, batch, channels, h, w, ny, nx, fold_batch, stream);
    if (dtype == PBR_F16) return pbr::fold_launch<__half>(src, dst, batch, channels, h, w, ny, nx, fold_batch, stream);
    return PBR_ERR_DTYPE;
}

int pbr_fold_gradient(const void *src, void *dst, int32_t batch, int32_t channels, int32_t h, int32_t w, int32_t ny,
                      int32_t nx, int fold_batch, void *stream) {
    return pbr_fold_gradient_typed(src, dst, batch, channels, h, w, ny, nx, fold_batch, PBR_F32, stream);
}

int pbr_decode_normal(const void *src, void *dst, int32_t channels, int64_t pixels, int dtype,
                      void *workspace, void *stream) {
    using namespace pbr;
    if (channels != 2 && channels != 3) return PBR_ERR_CHANNELS;
    if (!src || !dst || !workspace) return PBR_ERR_NULL_MAP;
    if (pixels < 1) return PBR_ERR_SHAPE;
    if (dtype != PBR_F32 && dtype != PBR_F16) return PBR_ERR_DTYPE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    int *flag = static_cast<int *>(workspace);
    const unsigned grid = stream_grid((size_t)pixels);
    const size_t esz = dtype == PBR_F32 ? 4 : 2, n = (size_t)pixels * 3;
    if (channels == 2) {
        if (hipMemsetAsync(flag, 0, sizeof(int), s) != hipSuccess) return hip_status();
    } else if (dtype == PBR_F32) {                                          // the probe writes the flag, 0 or 1
        hipLaunchKernelGGL((normal_probe_kernel<float>), dim3(1), dim3(256), 0, s, src, n, flag);
    } else {
        hipLaunchKernelGGL((normal_probe_kernel<__half>), dim3(1), dim3(256), 0, s, src, n, flag);
    }
    const char *s0 = static_cast<const char *>(src), *d0 = static_cast<const char *>(dst);
    if (channels == 3 && (d0 + n * esz <= s0 || s0 + n * esz <= d0)) {      // disjoint: one pass (+ a conditional fix-up)
        const int vec_ok = is_aligned(src, 4 * esz) && is_aligned(dst, 4 * esz);
        const unsigned fix_grid = stream_grid(n / 16 + 1);
        if (dtype == PBR_F32) {
            hipLaunchKernelGGL((decode_normal_speculative_kernel<float>), dim3(grid), dim3(256), 0, s, src, dst, pixels, flag);
            hipLaunchKernelGGL((keep_normal_kernel<float>), dim3(fix_grid), dim3(256), 0, s, src, dst, n, flag, vec_ok);
        } else {
            hipLaunchKernelGGL((decode_normal_speculative_kernel<__half>), dim3(grid), dim3(256), 0, s, src, dst, pixels, flag);
            hipLaunchKernelGGL((keep_normal_kernel<__half>), dim3(fix_grid), dim3(256), 0, s, src, dst, n, flag, vec_ok);
        }
        return hip_status();
    }
    if (dtype == PBR_F32) {
        if (channels == 3) {
            hipLaunchKernelGGL((any_negative_kernel<float>), dim3(stream_grid((size_t)pixels * 3 / 16)), dim3(256), 0, s, src, (size_t)pixels * 3, flag, (int)is_aligned(src, 16));
            hipLaunchKernelGGL((decode_normal_kernel<float, 3>), dim3(grid), dim3(256), 0, s, src, dst, pixels, flag);
        } else {
            hipLaunchKernelGGL((decode_normal_kernel<float, 2>), dim3(grid), dim3(256), 0, s, src, dst, pixels, flag);
        }
    } else {
        if (channels == 3) {
            hipLaunchKernelGGL((any_negative_kernel<__half>), dim3(stream_grid((size_t)pixels * 3 / 16)), dim3(256), 0, s, src, (size_t)pixels * 3, flag, (int)is_aligned(src, 8));
            hipLaunchKernelGGL((decode_normal_kernel<__half, 3>), dim3(grid), dim3(256), 0, s, src, dst, pixels, flag);
        } else {
            hipLaunchKernelGGL((decode_normal_kernel<__half, 2>), dim3(grid), dim3(256), 0, s, src, dst, pixels, flag);
        }
    }
    return hip_status();
}

int pbr_decode_normal_backward(const void *src, const void *grad_out, void *grad_in, int32_t channels, int64_t pixels,
                               const void *workspace, void *stream) {
    using namespace pbr;
    if (channels != 2 && channels != 3) return PBR_ERR_CHANNELS;
    if (!src || !grad_out || !grad_in || !workspace) return PBR_ERR_NULL_MAP;
    if (pixels < 1) return PBR_ERR_SHAPE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const unsigned grid = stream_grid((size_t)pixels);
    auto a = static_cast<const float *>(src), g = static_cast<const float *>(grad_out);
    float *gi = static_cast<float *>(grad_in);
    const int *flag = static_cast<const int *>(workspace);
    if (pixels % 4 == 0 && is_aligned(src, 16) && is_aligned(grad_out, 16) && is_aligned(grad_in, 16)) {
        const unsigned vgrid = stream_grid((size_t)pixels / 4);
        if (channels == 3) hipLaunchKernelGGL((decode_normal_backward_kernel<3, 4>), dim3(vgrid), dim3(256), 0, s, a, g, gi, pixels, flag);
        else hipLaunchKernelGGL((decode_normal_backward_kernel<2, 4>), dim3(vgrid), dim3(256), 0, s, a, g, gi, pixels, flag);
    } else if (channels == 3) {
        hipLaunchKernelGGL((decode_normal_backward_kernel<3, 1>), dim3(grid), dim3(256), 0, s, a, g, gi, pixels, flag);
    } else {
        hipLaunchKernelGGL((decode_normal_backward_kernel<2, 1>), dim3(grid), dim3(256), 0, s, a, g, gi, pixels, flag);
    }
    return hip_status();
}

int pbr_abi_version(void) { return PBR_HIP_ABI_VERSION; }

size_t pbr_render_desc_size(void) { return sizeof(pbr_render_desc); }

const char *pbr_error_string(int code) {
    switch (code) {
        case PBR_OK: return "ok";
        case PBR_ERR_NULL_MAP: return "a required map pointer is NULL";
        case PBR_ERR_WORKFLOW: return "Material must have either 'metallic' or 'specular' property.";
        case PBR_ERR_LIGHT_TYPE: return "Unsupported light_type. Must be 'directional' or 'point'.";
        case PBR_ERR_SHAPE: return "bad extents, band, light count or ABI version";
        case PBR_ERR_DTYPE: return "unsupported map dtype";
        case PBR_ERR_CHANNELS: return "Normal map must have 2 or 3 channels.";
        case PBR_ERR_NO_DEVICE: return "no HIP device";
        case PBR_ERR_UNSUPPORTED: return "not implemented for this configuration";
        default: break;
    }
    if (code >= 1000) return hipGetErrorString(static_cast<hipError_t>(code - 1000));
    return "unknown error";
}

}  // extern "C"
