// ct_blend.hpp -- material blending fused in front of the Cook-Torrance evaluation.
//
// /root/reference/examples/example_blend.py:14-32 blends two materials (pypbr/blending/functional.py:64-145) and
// hands the result straight to the BRDF.  Unfused that is, per pixel, 68 B read + 32 B written by the blend, a
// second pass over the blended normal map (MaterialBase._process_normal_map runs again when the normal is assigned,
// base.py:191-242), then 32 B read + 12 B written by the render.  Fused: both materials and the mask are read once
// (68 B), the blended texels never leave registers, 12 B are written.
//
// Arithmetic = the stand-alone kernels' (blend.hip: blend_kernel; map_ops.hip: decode_normal_kernel), in the same
// order, so the fused result equals blend -> assign -> render:
//   every map      mask * map1 + (1 - mask) * map2                                      functional.py:103-110
//   normals        normalise both, blend, normalise                                     functional.py:119-145
//   re-assignment  the blended normal is re-read as [0,1]-encoded (x*2-1, normalise) unless some component of it,
//                  anywhere in the map, is negative (base.py:212-213).  That is a property of the whole map, so a
//                  reduction kernel sets one flag per material first (blend_normal_sign_kernel; it stops at the
//                  first negative value it sees, i.e. immediately for real normal maps) and this kernel reads it.
#pragma once
#include "ct_kernel.hpp"

namespace pbr {

// The second material of a fused blend + the weights of the first.  Member names match KArgs (load_texels).
struct KBlend {
    const void *albedo, *normal, *rough, *metal, *spec;
    int64_t a_bs, a_cs, n_bs, n_cs, r_bs, m_bs, s_bs, s_cs;
    const float *mask;        // [B|1][map_h][map_w] fp32
    int64_t k_bs;             // mask batch stride (0: one mask for the whole batch)
    const int *normal_signed; // [B]: 1 = the blended normal map of material b has a negative component
};

__device__ __forceinline__ float lerp_mask(float w, float iw, float a, float b) { return fmaf(w, a, iw * b); }

// Blends u into t (t = mask * t + (1 - mask) * u) for the lane's VEC pixels.
template <int WF, int VEC>
__device__ __forceinline__ void blend_texels(Texels<VEC> &t, const Texels<VEC> &u, const float w[VEC], bool keep_signed) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const float wj = w[j], iw = 1.0f - wj;
#pragma unroll
        for (int c = 0; c < 3; ++c) t.al[c][j] = lerp_mask(wj, iw, t.al[c][j], u.al[c][j]);
        t.ro[j] = lerp_mask(wj, iw, t.ro[j], u.ro[j]);
        if (WF != PBR_WORKFLOW_SPECULAR) t.me[j] = lerp_mask(wj, iw, t.me[j], u.me[j]);
        if (WF == PBR_WORKFLOW_SPECULAR) {
#pragma unroll
            for (int c = 0; c < 3; ++c) t.sp[c][j] = lerp_mask(wj, iw, t.sp[c][j], u.sp[c][j]);
        }
        // _blend_normals (functional.py:119-145)
        const Vec3 a = {t.nm[0][j], t.nm[1][j], t.nm[2][j]}, b = {u.nm[0][j], u.nm[1][j], u.nm[2][j]};
        const float ra = rsq(fmaxf(dot(a, a), 1e-24f)), rb = rsq(fmaxf(dot(b, b), 1e-24f));
        Vec3 c = {fmaf(wj, a.x * ra, iw * (b.x * rb)), fmaf(wj, a.y * ra, iw * (b.y * rb)), fmaf(wj, a.z * ra, iw * (b.z * rb))};
        const float rc = rsq(fmaxf(dot(c, c), 1e-24f));
        c = {c.x * rc, c.y * rc, c.z * rc};
        if (!keep_signed) {   // base.py:214-216 on re-assignment: read as [0,1]-encoded, then F.normalize
            c = {fmaf(c.x, 2.0f, -1.0f), fmaf(c.y, 2.0f, -1.0f), fmaf(c.z, 2.0f, -1.0f)};
            const float r = rsq(fmaxf(fmaf(c.z, c.z, fmaf(c.y, c.y, c.x * c.x)), 1e-24f));
            c = {c.x * r, c.y * r, c.z * r};
        }
        t.nm[0][j] = c.x; t.nm[1][j] = c.y; t.nm[2][j] = c.z;
    }
}

//   fp32 maps and output; both materials carry all four maps.  Same grid / tile order as cook_torrance_kernel.
template <int LIGHT, int WF, int VEC, bool MULTI>
__global__ __launch_bounds__(256)
#ifndef PBR_BLEND_WAVES
#define PBR_BLEND_WAVES 2, 3
#endif
__attribute__((amdgpu_waves_per_eu(PBR_BLEND_WAVES)))
void cook_torrance_blend_kernel(const KArgs a, const KBlend b) {
    const uint32_t tile = tile_of_workgroup(a, blockIdx.x);
    const int ty = (int)a.div_tx.div(tile);
    const LanePos p = lane_pos<VEC>(a, (int)tile - ty * a.tiles_x, ty);
    if (!p.valid) return;
    Texels<VEC> t, u;
    float w[VEC];
    load_texels<WF, float, VEC, true>(a, true, p, t);
    load_texels<WF, float, VEC, true>(b, true, p, u);
    if (p.sb) Ld<float, VEC>::template load<true>(plane_at<float>(b.mask, p.b0 * b.k_bs, (uint32_t)p.src), 0, w);
    else Ld<float, VEC>::template load<true>(b.mask, p.b * b.k_bs + p.src, w);
    const bool keep_signed = b.normal_signed[p.sb ? p.b0 : p.b] != 0;
    blend_texels<WF, VEC>(t, u, w, keep_signed);
    shade_and_store<LIGHT, WF, float, VEC, MULTI, true, MULTI>(a, p, t);
}

// The fused blend over TILED maps (round 6): the repeat-inner walk of cook_torrance_repeat_kernel with the blend in front of the decode -- both
// materials' texels and the mask loaded once, blended and re-decoded ONCE per texel, evaluated at every repeat (the wrap-around form above
// blends at every OUTPUT pixel and re-reads both materials through the caches: 2 x 2048^2 under tile(2): 154 us at 0.40 of HBM).
template <int LIGHT, int WF, bool MULTI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 8)))
void cook_torrance_repeat_blend_kernel(const KArgs a, const KBlend b) {
    repeat_forward_body<LIGHT, WF, float, float, false, true, MULTI>(a, [&](const LanePos &p, Texels<4> &t) {
        Texels<4> u;
        float w[4];
        load_texels<WF, float, 4, false>(b, true, p, u);
        const int mat = p.sb ? p.b0 : p.b;
        Ld<float, 4>::template load<false>(b.mask, mat * b.k_bs + p.src, w);
        blend_texels<WF, 4>(t, u, w, b.normal_signed[mat] != 0);
    });
}

}  // namespace pbr
