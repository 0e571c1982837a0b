// ct_kernel.hpp -- the fused Cook-Torrance kernels (device side) for gfx950.
//
// Work decomposition.  A *tile* is what one workgroup (T = 64..256 lanes, no LDS, no barrier)
// evaluates at a time: bx lanes along x (each lane VEC = 4 consecutive pixels -> 16-byte
// accesses) by T/bx rows, with bx = min(T, next_pow2(W/4)).  For 4K maps every wave touches one
// 1 KiB-contiguous run of each of the 8 input and 3 output planes.
//
// Schedule: one tile per workgroup, 1-D grid in row-major tile order.  A persistent grid-stride
// variant with register double buffering (next tile's loads issued before the current tile's
// arithmetic) was built and measured 10-20 % SLOWER (135 VGPRs -> 3 waves/SIMD; DESIGN.md,
// "Schedule experiments"): with 5 waves/SIMD of independent one-shot waves the memory pipe is
// already saturated and the chip-wide dispatcher is the cheaper software pipeline.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <cstdint>

#include "../../include/pbr_hip.h"
#include "brdf_math.hpp"

namespace pbr {

// Per-light, wave-uniform block (host-folded, all fp32).
struct LightU {
    float l[3];       // directional: normalised L (:126); point: position (:129)
    float h[3];       // directional: V + L
    float hh;         // directional: |V+L|^2
    float p5;         // directional: (1 - clamp(Hv.V))^5
    float inten[3];   // :96
};

// Multiply-shift division of n < 2^31 by a fixed d (Granlund-Montgomery round-up form).
struct FastDiv {
    uint32_t mul; int32_t sh1, sh2;
    __host__ void init(uint32_t d) {
        int l = 0;
        while ((1ull << l) < d) ++l;
        mul = (uint32_t)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
        sh1 = l < 1 ? l : 1;
        sh2 = l < 1 ? 0 : l - 1;
    }
    __device__ __forceinline__ uint32_t div(uint32_t n) const {
        const uint32_t t = __umulhi(mul, n);
        return (t + ((n - t) >> sh1)) >> sh2;
    }
};

struct KArgs {
    const void *albedo, *normal, *rough, *metal, *spec;
    void *out;
    int64_t a_bs, a_cs, n_bs, n_cs, r_bs, m_bs, s_bs, s_cs;   // element strides (batch, channel)
    int64_t o_bs, o_cs;
    int32_t rows;            // B * H
    int32_t H, W;            // band rows, width (pixels)
    int32_t wv;              // lanes per row: W / VEC
    int32_t bx_log2;         // tile = (1<<bx_log2) lanes along x by (block>>bx_log2) rows
    int32_t bt_log2;         // log2 of the workgroup size (64..256 lanes)
    int32_t tiles_x;         // tiles per row
    int32_t n_tiles;         // tiles_x * ceil(rows / tile rows)
    FastDiv div_h;           // row / H
    FastDiv div_tx;          // tile / tiles_x
    int32_t y_offset, H_total;
    float x0, x1, xstep;     // torch.linspace(-s/2, s/2, W)   :132
    float y0, y1, ystep;     // torch.linspace(-s/2, s/2, H_total)   :133
    float V[3];              // F.normalize(view_dir)   :95
    int32_t n_lights;
    int32_t albedo_srgb, spec_srgb, out_srgb, has_normal;
    LightU lights[PBR_MAX_LIGHTS];
};

// ------------------------------------------------------------------ typed vector I/O
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

template <typename T, int VEC> struct Ld;
template <> struct Ld<float, 4> {
    template <bool NT> static __device__ __forceinline__ void load(const void *p, int64_t i, float v[4]) {
        const f32x4 *q = reinterpret_cast<const f32x4 *>(static_cast<const float *>(p) + i);
        const f32x4 t = NT ? __builtin_nontemporal_load(q) : *q;
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    template <bool NT> static __device__ __forceinline__ void store(void *p, int64_t i, const float v[4]) {
        const f32x4 t = {v[0], v[1], v[2], v[3]};
        f32x4 *q = reinterpret_cast<f32x4 *>(static_cast<float *>(p) + i);
        if (NT) __builtin_nontemporal_store(t, q); else *q = t;
    }
};
template <> struct Ld<float, 1> {
    template <bool NT> static __device__ __forceinline__ void load(const void *p, int64_t i, float v[1]) {
        v[0] = static_cast<const float *>(p)[i];
    }
    template <bool NT> static __device__ __forceinline__ void store(void *p, int64_t i, const float v[1]) {
        static_cast<float *>(p)[i] = v[0];
    }
};
template <> struct Ld<__half, 4> {
    template <bool NT> static __device__ __forceinline__ void load(const void *p, int64_t i, float v[4]) {
        const f16x4 *q = reinterpret_cast<const f16x4 *>(static_cast<const _Float16 *>(p) + i);
        const f16x4 t = NT ? __builtin_nontemporal_load(q) : *q;
        v[0] = (float)t.x; v[1] = (float)t.y; v[2] = (float)t.z; v[3] = (float)t.w;
    }
    template <bool NT> static __device__ __forceinline__ void store(void *p, int64_t i, const float v[4]) {
        const f16x4 t = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
        f16x4 *q = reinterpret_cast<f16x4 *>(static_cast<_Float16 *>(p) + i);
        if (NT) __builtin_nontemporal_store(t, q); else *q = t;
    }
};
template <> struct Ld<__half, 1> {
    template <bool NT> static __device__ __forceinline__ void load(const void *p, int64_t i, float v[1]) {
        v[0] = (float)static_cast<const _Float16 *>(p)[i];
    }
    template <bool NT> static __device__ __forceinline__ void store(void *p, int64_t i, const float v[1]) {
        static_cast<_Float16 *>(p)[i] = (_Float16)v[0];
    }
};

// 8 pixels per lane: fp16 maps read as one 16-byte load per plane; fp32 results leave as two 16-byte stores.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <> struct Ld<__half, 8> {
    template <bool NT> static __device__ __forceinline__ void load(const void *p, int64_t i, float v[8]) {
        const f16x8 *q = reinterpret_cast<const f16x8 *>(static_cast<const _Float16 *>(p) + i);
        const f16x8 t = NT ? __builtin_nontemporal_load(q) : *q;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (float)t[j];
    }
    template <bool NT> static __device__ __forceinline__ void store(void *p, int64_t i, const float v[8]) {
        f16x8 t;
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = (_Float16)v[j];
        f16x8 *q = reinterpret_cast<f16x8 *>(static_cast<_Float16 *>(p) + i);
        if (NT) __builtin_nontemporal_store(t, q); else *q = t;
    }
};
template <> struct Ld<float, 8> {
    template <bool NT> static __device__ __forceinline__ void load(const void *p, int64_t i, float v[8]) {
        Ld<float, 4>::load<NT>(p, i, v); Ld<float, 4>::load<NT>(p, i + 4, v + 4);
    }
    template <bool NT> static __device__ __forceinline__ void store(void *p, int64_t i, const float v[8]) {
        Ld<float, 4>::store<NT>(p, i, v); Ld<float, 4>::store<NT>(p, i + 4, v + 4);
    }
};

// torch.linspace two-ended evaluation (what ATen's device kernel computes), branch-free:
// i < n/2 ? a + step*i : b - step*(n-1-i).
__device__ __forceinline__ float linspace_at(float a, float b, float step, int n, int i) {
    const bool lo = i < (n >> 1);
    return fmaf(lo ? step : -step, (float)(lo ? i : n - 1 - i), lo ? a : b);
}

// ------------------------------------------------------------------ one lane's share of a tile
struct LanePos {
    int b, y, x;          // material, row inside the band, first pixel column
    int64_t pix;          // y * W + x
    bool valid;
};

template <int VEC>
__device__ __forceinline__ LanePos lane_pos(const KArgs &a, int tile_x, int tile_y) {
    const int tid = threadIdx.x;
    const int xv = (tile_x << a.bx_log2) + (tid & ((1 << a.bx_log2) - 1));
    const int row = (tile_y << (a.bt_log2 - a.bx_log2)) + (tid >> a.bx_log2);  // b * H + y
    LanePos p;
    p.valid = xv < a.wv && row < a.rows;
    p.b = (int)a.div_h.div((uint32_t)row);
    p.y = row - p.b * a.H;
    p.x = xv * VEC;
    p.pix = (int64_t)p.y * a.W + p.x;
    return p;
}

template <int VEC> struct Texels { float al[3][VEC], nm[3][VEC], ro[VEC], me[VEC], sp[3][VEC]; };

// Issues every load of the lane's texels; nothing here waits on memory.
template <int WF, typename TI, int VEC, bool NT>
__device__ __forceinline__ void load_texels(const KArgs &a, const LanePos &p, Texels<VEC> &t) {
#pragma unroll
    for (int c = 0; c < 3; ++c) Ld<TI, VEC>::template load<NT>(a.albedo, p.b * a.a_bs + c * a.a_cs + p.pix, t.al[c]);
    if (a.has_normal) {
#pragma unroll
        for (int c = 0; c < 3; ++c) Ld<TI, VEC>::template load<NT>(a.normal, p.b * a.n_bs + c * a.n_cs + p.pix, t.nm[c]);
    }
    Ld<TI, VEC>::template load<NT>(a.rough, p.b * a.r_bs + p.pix, t.ro);
    if (WF != PBR_WORKFLOW_SPECULAR) Ld<TI, VEC>::template load<NT>(a.metal, p.b * a.m_bs + p.pix, t.me);
    if (WF == PBR_WORKFLOW_SPECULAR) {
#pragma unroll
        for (int c = 0; c < 3; ++c) Ld<TI, VEC>::template load<NT>(a.spec, p.b * a.s_bs + c * a.s_cs + p.pix, t.sp[c]);
    }
}

// Material terms of pixel j of the lane (:99-118): base colour, F0, kD scale, normal, roughness.
template <int WF, int VEC>
__device__ __forceinline__ void material_terms(const Texels<VEC> &t, int j, const Vec3 &V, PixelTerms &pt) {
    float base[3], f0[3], kd_scale = 1.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) base[c] = t.al[c][j];
    if (WF == PBR_WORKFLOW_METALLIC) {
        const float m = t.me[j];
#pragma unroll
        for (int c = 0; c < 3; ++c) f0[c] = fmaf(m, base[c] - kDielectricF0, kDielectricF0);   // lerp :107
        kd_scale = 1.0f - m;                                                            // :170
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) f0[c] = t.sp[c][j];                                 // :112-113
    }
    pixel_terms(Vec3{t.nm[0][j], t.nm[1][j], t.nm[2][j]}, V, t.ro[j], base, f0, kd_scale, pt);
}

// Everything between the loads and the stores (cooktorrance.py:99-180 and the conversions it calls).
// Run-time flags (sRGB decode/encode, normal present) are wave-uniform and each guards ONE hoisted
// block over all VEC pixels, so the shading code stays one basic block and the scheduler can
// interleave the pixels' transcendental latencies.
template <int LIGHT, int WF, typename TO, int VEC, bool MULTI, bool NT>
__device__ __forceinline__ void shade_and_store(const KArgs &a, const LanePos &p, Texels<VEC> &t) {
    if (!a.has_normal) {                                                    // +Z, :147-152
#pragma unroll
        for (int j = 0; j < VEC; ++j) { t.nm[0][j] = 0.0f; t.nm[1][j] = 0.0f; t.nm[2][j] = 1.0f; }
    }
    // ---- colour decode (base.py:262-277; diffuse.py:76-91; metallic.py:98-108)
    if (a.albedo_srgb) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int j = 0; j < VEC; ++j) t.al[c][j] = srgb_to_linear(t.al[c][j]);
    }
    if (WF == PBR_WORKFLOW_CONVERTED) {   // to_diffuse_specular_material, then the specular workflow
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const float m = t.me[j], om = 1.0f - m;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                t.sp[c][j] = fmaf(t.al[c][j], m, kDielectricF0 * om);       // metallic.py:108
                t.al[c][j] = t.al[c][j] * om;                               // metallic.py:105
            }
        }
    }
    if (WF != PBR_WORKFLOW_METALLIC && a.spec_srgb) {   // CONVERTED: upstream default decodes again (F6)
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int j = 0; j < VEC; ++j) t.sp[c][j] = srgb_to_linear(t.sp[c][j]);
    }

    const Vec3 V = {a.V[0], a.V[1], a.V[2]};
    float ys = 0.0f;
    if (LIGHT == PBR_LIGHT_POINT) ys = linspace_at(a.y0, a.y1, a.ystep, a.H_total, p.y + a.y_offset);

    float res[3][VEC];
    if constexpr (!MULTI) {
        // ---- one light: pixel by pixel, terms and shading back to back (shortest live ranges: 87 VGPRs)
        const LightU &lu = a.lights[0];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            PixelTerms pt;
            material_terms<WF, VEC>(t, j, V, pt);
            LightGeom g;
            if (LIGHT == PBR_LIGHT_POINT) {
                g = point_light_geom(V, Vec3{lu.l[0], lu.l[1], lu.l[2]}, linspace_at(a.x0, a.x1, a.xstep, a.W, p.x + j), ys);
            } else {
                g.L = {lu.l[0], lu.l[1], lu.l[2]};
                g.h = {lu.h[0], lu.h[1], lu.h[2]};
                g.hh = lu.hh; g.p5 = lu.p5; g.att = 1.0f;
            }
            float col[3];
            shade_light(pt, g, lu.inten, col);
#pragma unroll
            for (int c = 0; c < 3; ++c) res[c][j] = col[c];
        }
    } else {
        // ---- several lights (H12): light-independent terms of every pixel first, then lights in the
        // OUTER (uniform) loop and the lane's pixels inside it, so each light's scalar loads and loop
        // overhead are paid once per VEC pixels and the pixels' transcendental latencies interleave
        PixelTerms pt[VEC];
        float xs[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            material_terms<WF, VEC>(t, j, V, pt[j]);
            xs[j] = LIGHT == PBR_LIGHT_POINT ? linspace_at(a.x0, a.x1, a.xstep, a.W, p.x + j) : 0.0f;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int j = 0; j < VEC; ++j) res[c][j] = 0.0f;
        for (int l = 0; l < a.n_lights; ++l) {
            const LightU &lu = a.lights[l];
            LightGeom g;
            if (LIGHT == PBR_LIGHT_DIRECTIONAL) {
                g.L = {lu.l[0], lu.l[1], lu.l[2]};
                g.h = {lu.h[0], lu.h[1], lu.h[2]};
                g.hh = lu.hh; g.p5 = lu.p5; g.att = 1.0f;
            }
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                if (LIGHT == PBR_LIGHT_POINT) g = point_light_geom(V, Vec3{lu.l[0], lu.l[1], lu.l[2]}, xs[j], ys);
                float col[3];
                shade_light(pt[j], g, lu.inten, col);
#pragma unroll
                for (int c = 0; c < 3; ++c) res[c][j] += col[c];
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c)                                         // sum of per-light clamped terms, clamped
#pragma unroll
            for (int j = 0; j < VEC; ++j) res[c][j] = clamp01(res[c][j]);
    }
    if (a.out_srgb) {                                                       // :179-180
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int j = 0; j < VEC; ++j) res[c][j] = linear_to_srgb_unit(res[c][j]);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) Ld<TO, VEC>::template store<NT>(a.out, p.b * a.o_bs + c * a.o_cs + p.pix, res[c]);
}

// ------------------------------------------------------------------ kernel
//   LIGHT: PBR_LIGHT_*     WF: PBR_WORKFLOW_*     TI/TO: map / output storage types
//   VEC: pixels per lane (4 = 16-byte fp32 accesses, 8 = 16-byte fp16 accesses, 1 = ragged widths /
//        unaligned views)
//   MULTI: more than one light (uniform loop) -- the single-light body is straight-line
// 1-D grid, one tile per workgroup, tiles ordered x fastest.
template <int LIGHT, int WF, typename TI, typename TO, int VEC, bool MULTI, bool NT>
__global__ __launch_bounds__(256, MULTI ? 4 : 1) void cook_torrance_kernel(const KArgs a) {
    const int ty = (int)a.div_tx.div(blockIdx.x);
    const LanePos p = lane_pos<VEC>(a, (int)blockIdx.x - ty * a.tiles_x, ty);
    if (!p.valid) return;
    Texels<VEC> t;
    load_texels<WF, TI, VEC, NT>(a, p, t);
    shade_and_store<LIGHT, WF, TO, VEC, MULTI, NT>(a, p, t);
}

}  // namespace pbr
