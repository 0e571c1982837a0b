// ct_blend_backward.hpp -- gradient of the fused blend + evaluate kernel (ct_blend.hpp) w.r.t. BOTH materials and the mask.
//
// The documented training use of the reference is a rendering loss (docs/source/tutorials/06_advanced.rst:73-107); with a
// blended material (examples/example_blend.py:14-32) its autograd runs back through CookTorranceBRDF.forward, the
// re-assignment of the blended normal (base.py:191-242) and blend_with_mask (blending/functional.py:64-145).  Unfused, that
// is three backward passes over materialised maps (render 76 B/pixel, decode 36 B, blend ~250 B); here it is ONE pass:
// both materials, the mask and the upstream gradient are read once (80 B/pixel), the forward terms are re-evaluated in
// registers, the chain rule runs through the shading (backward_body_to, the very code of cook_torrance_backward_kernel),
// the re-decode and the blend, and the gradients of both materials and of the mask are written (68 B/pixel).
//
// Sub-gradient conventions are torch's (ct_backward.hpp); the blend's arithmetic is blend_texels' (ct_blend.hpp), so forward
// and backward agree on every value.  fp32 maps; both materials carry all four maps.
#pragma once
#include "ct_backward.hpp"
#include "ct_blend.hpp"

namespace pbr {

struct BBlend {
    void *g_albedo, *g_normal, *g_rough, *g_metal, *g_spec;      // material 2, contiguous [B][C][H][W] fp32, NULL = not wanted
    float *g_mask;                                               // [B][1][H][W] (one value per output pixel), NULL = not wanted
};

// The blend's own chain rule: the gradients w.r.t. the BLENDED texels (what the shading read) -> both materials' maps and the mask, stored.
// `t` / `u`: the two materials' raw texels, `w` the mask, `at`-style indexing through a.o_cs (elements per gradient plane) and p.pix
// (the lane's first pixel inside a plane): one gradient value per pixel the LANE OWNS -- output pixels for cook_torrance_blend_backward_kernel,
// source texels for the repeat-inner walk over tiled maps (ct_repeat_backward.hpp, where ga ... gs are already the sums over the repeats).
template <int WF, int VEC>
__device__ __forceinline__ void blend_backward_sink(const KArgs &a, const LanePos &p, int mat, const Texels<VEC> &t, const Texels<VEC> &u, const float (&w)[VEC],
                                                    bool keep_signed, const BArgs &g1, const BBlend &g2, float (&ga)[3][VEC], float (&gn)[3][VEC],
                                                    float (&gr)[VEC], float (&gm)[VEC], float (&gs)[3][VEC]) {
    float gw[VEC], o1[VEC], o2[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) gw[j] = 0.0f;
    // every plain map: x = w a + (1 - w) b  ->  g_a = w g, g_b = (1 - w) g, g_w += g (a - b)      (functional.py:103-110)
    auto lerp_back = [&](const float (&g)[VEC], const float (&av)[VEC], const float (&bv)[VEC], void *p1, void *p2, int channels, int c) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            o1[j] = w[j] * g[j];
            o2[j] = (1.0f - w[j]) * g[j];
            gw[j] = fmaf(g[j], av[j] - bv[j], gw[j]);
        }
        const int64_t at = ((int64_t)mat * channels + c) * a.o_cs + p.pix;
        if (p1) Ld<float, VEC>::template store<true>(p1, at, o1);
        if (p2) Ld<float, VEC>::template store<true>(p2, at, o2);
    };
#pragma unroll
    for (int c = 0; c < 3; ++c) lerp_back(ga[c], t.al[c], u.al[c], g1.g_albedo, g2.g_albedo, 3, c);
    lerp_back(gr, t.ro, u.ro, g1.g_rough, g2.g_rough, 1, 0);
    if (WF != PBR_WORKFLOW_SPECULAR) {
        lerp_back(gm, t.me, u.me, g1.g_metal, g2.g_metal, 1, 0);
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) lerp_back(gs[c], t.sp[c], u.sp[c], g1.g_spec, g2.g_spec, 3, c);
    }
    // the normal: o = normalize(w a^ + (1 - w) b^) (functional.py:119-145), then on re-assignment -- unless the blended map
    // counts as signed -- normalize(2 o - 1) (base.py:214-216).  gn is the adjoint of what the shading read.
    float n1[3][VEC], n2[3][VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const float wj = w[j], iw = 1.0f - wj;
        const Vec3 av = {t.nm[0][j], t.nm[1][j], t.nm[2][j]}, bv = {u.nm[0][j], u.nm[1][j], u.nm[2][j]};
        const float ra = rsq(fmaxf(dot(av, av), 1e-24f)), rb = rsq(fmaxf(dot(bv, bv), 1e-24f));
        const Vec3 ah = {av.x * ra, av.y * ra, av.z * ra}, bh = {bv.x * rb, bv.y * rb, bv.z * rb};
        const Vec3 c = {fmaf(wj, ah.x, iw * bh.x), fmaf(wj, ah.y, iw * bh.y), fmaf(wj, ah.z, iw * bh.z)};
        const float rc = rsq(fmaxf(dot(c, c), 1e-24f));
        const Vec3 o = {c.x * rc, c.y * rc, c.z * rc};
        Vec3 g = {gn[0][j], gn[1][j], gn[2][j]};
        if (!keep_signed) {
            const Vec3 d = {fmaf(o.x, 2.0f, -1.0f), fmaf(o.y, 2.0f, -1.0f), fmaf(o.z, 2.0f, -1.0f)};
            const float rd = rsq(fmaxf(dot(d, d), 1e-24f));
            const Vec3 e = {d.x * rd, d.y * rd, d.z * rd};
            const float eg = dot(e, g);
            g = {2.0f * (g.x - e.x * eg) * rd, 2.0f * (g.y - e.y * eg) * rd, 2.0f * (g.z - e.z * eg) * rd};
        }
        const float og = dot(o, g);                                  // F.normalize: (g - o (o.g)) / |c|
        const Vec3 gc = {(g.x - o.x * og) * rc, (g.y - o.y * og) * rc, (g.z - o.z * og) * rc};
        gw[j] += gc.x * (ah.x - bh.x) + gc.y * (ah.y - bh.y) + gc.z * (ah.z - bh.z);
        const Vec3 gA = {wj * gc.x, wj * gc.y, wj * gc.z};
        const float da = dot(ah, gA);
        n1[0][j] = (gA.x - ah.x * da) * ra; n1[1][j] = (gA.y - ah.y * da) * ra; n1[2][j] = (gA.z - ah.z * da) * ra;
        const Vec3 gB = {iw * gc.x, iw * gc.y, iw * gc.z};
        const float db = dot(bh, gB);
        n2[0][j] = (gB.x - bh.x * db) * rb; n2[1][j] = (gB.y - bh.y * db) * rb; n2[2][j] = (gB.z - bh.z * db) * rb;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int64_t at = ((int64_t)mat * 3 + c) * a.o_cs + p.pix;
        if (g1.g_normal) Ld<float, VEC>::template store<true>(g1.g_normal, at, n1[c]);
        if (g2.g_normal) Ld<float, VEC>::template store<true>(g2.g_normal, at, n2[c]);
    }
    if (g2.g_mask) Ld<float, VEC>::template store<true>(g2.g_mask, (int64_t)mat * a.o_cs + p.pix, gw);
}

template <int LIGHT, int WF, int VEC, bool MULTI>
__global__ __launch_bounds__(64) void cook_torrance_blend_backward_kernel(const KArgs a, const KBlend b, const BArgs g1, const BBlend g2) {
    const uint32_t tile = tile_of_workgroup(a, blockIdx.x);
    const int ty = (int)a.div_tx.div(tile);
    const LanePos p = lane_pos<VEC>(a, (int)tile - ty * a.tiles_x, ty);
    if (!p.valid) return;
    Texels<VEC> t, u;
    float w[VEC], go[3][VEC];
    load_texels<WF, float, VEC, true>(a, true, p, t);
    load_texels<WF, float, VEC, true>(b, true, p, u);
    const int mat = p.sb ? p.b0 : p.b;
    Ld<float, VEC>::template load<true>(b.mask, mat * b.k_bs + p.src, w);
#pragma unroll
    for (int c = 0; c < 3; ++c) Ld<float, VEC>::template load<true>(g1.gout, mat * a.o_bs + c * a.o_cs + p.pix, go[c]);
    const bool keep_signed = b.normal_signed[mat] != 0;
    Texels<VEC> x = t;                                           // the blended texels: what the shading reads
    blend_texels<WF, VEC>(x, u, w, keep_signed);
    backward_body_to<LIGHT, WF, VEC, MULTI, float, false>(a, g1, p, x, go, nullptr, 0,
        [&](float (&ga)[3][VEC], float (&gn)[3][VEC], float (&gr)[VEC], float (&gm)[VEC], float (&gs)[3][VEC]) {
            blend_backward_sink<WF, VEC>(a, p, mat, t, u, w, keep_signed, g1, g2, ga, gn, gr, gm, gs);
        });
}


}  // namespace pbr
