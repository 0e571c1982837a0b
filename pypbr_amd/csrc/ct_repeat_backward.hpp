// ct_repeat_backward.hpp -- gradient of the fused evaluation w.r.t. TILED maps, folded in registers (round 5).
//
// material.tile(n) (/root/reference/pypbr/materials/base.py:524-537) is `map.repeat(1, n, n)`: autograd gives a texel the SUM of
// the gradients of its n x n repeats.  The reference's example material is `resize(512).tile(2)` (examples/example_brdf.py:11) and
// its documented ML use a rendering loss over such a material (docs/source/tutorials/06_advanced.rst:73-107).  Until round 4 the
// backward kernel wrote one gradient per OUTPUT pixel (32 B x n^2 per texel) and pbr_fold_gradient read them back and summed.
// Here the grid walks the SOURCE maps, like cook_torrance_repeat_kernel in the forward direction: a lane loads its texels once,
// re-evaluates the light-independent forward terms once (colour decode and its slope, F0, normal, the PixelTerms), then visits the
// rep_y x rep_x output positions of its texels -- upstream gradient in (12 B per output pixel), light geometry, the chain rule of
// ct_backward.hpp (eval_light / backprop_light), the light-independent tail -- and adds each position's gradient to register
// accumulators in the order pbr_fold_gradient adds them (repeat rows outer, repeat columns inner).  Map-sized gradients are
// written once: 12 B per output pixel + 64 B per texel instead of 76 + 32 (fold reads) + 32/n^2 per output pixel.
//
// Same functions per position as backward_body_to.  What differs from pbr_cook_torrance_backward + pbr_fold_gradient is the ORDER OF
// SUMMATION, never a formula (round 6): the chain rule splits into a per-position part (light geometry, eval_light, backprop_light:
// adjoints of kb / f0 / a2 / k / N.V / the unit normal) and a light-independent tail (bwd_tail) that is LINEAR in those adjoints with
// coefficients that depend on the texel only.  The positions' adjoints are accumulated (backprop_light's fused multiply-adds, straight
// into one PixelAdjoint) and the tail runs once per texel -- in round 5 the tail ran per position and the per-position gradients were
// added (bit-identical to the two-kernel form, 45 packed instructions per position more; measured: DESIGN.md 3.2).  Equal to the
// two-kernel form in real arithmetic and to fp32 rounding in practice (tests/test_gpu_round5.py: <= 4e-6 of the largest gradient;
// both forms against float64 autograd through repeat()).  fp16 maps: the sum is rounded once.  Under a DIRECTIONAL light
// every repeat evaluates to the same colour and the chain rule is linear in the upstream gradient: the repeats' upstream values are
// summed first and the texel is differentiated once.
//
// With the MseLoss policy the upstream gradient is formed in the kernel from the target image (the rendering-loss step for tiled
// maps: pbr_cook_torrance_mse_step lifts its "untiled" restriction through this kernel).
//
// One packed pair (2 texels) per lane: the texel state that must stay live across the repeat loop (PixelTerms, colour slopes,
// 8-10 accumulators) plus one position's working set is ~150 VGPRs per pair; two pairs would leave one wave per SIMD.
#pragma once
#include <type_traits>
#include "ct_backward.hpp"
#include "ct_blend_backward.hpp"

namespace pbr {

// Forward decode of one pixel group, as the chain rule needs it again (cooktorrance.py:99-118 and the conversions it calls):
// the statements of backward_body_to's "forward: decoded colours and their derivatives" block.
template <class R> struct BwdTexelT {
    R base[3], dbase[3], f0[3], df0[3], alin[3];
    R m, om, kd_scale, rough;
    Vec3T<R> nraw;
    PixelTermsT<R> pt;
};

template <int WF, int VEC, class R>
__device__ __forceinline__ void bwd_decode(const KArgs &a, const Texels<VEC> &t, int g, const Vec3 &V, BwdTexelT<R> &x) {
    x.kd_scale = splat<R>(1.0f);
    x.m = WF != PBR_WORKFLOW_SPECULAR ? gather<R>(t.me, g) : splat<R>(0.0f);
    x.om = splat<R>(1.0f) - x.m;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const R al = gather<R>(t.al[c], g);
        x.alin[c] = x.base[c] = al; x.dbase[c] = splat<R>(1.0f);
        if (a.albedo_srgb) srgb_to_linear_and_grad(al, x.base[c], x.dbase[c]);
        x.alin[c] = x.base[c];
        if (WF == PBR_WORKFLOW_METALLIC) {
            x.f0[c] = fma_(x.m, x.base[c], x.om * kDielectricF0);                  // lerp(0.04, base, m) :107
            x.df0[c] = splat<R>(0.0f);
        } else if (WF == PBR_WORKFLOW_SPECULAR) {
            const R sp = gather<R>(t.sp[c], g);
            x.f0[c] = sp; x.df0[c] = splat<R>(1.0f);
            if (a.spec_srgb) srgb_to_linear_and_grad(sp, x.f0[c], x.df0[c]);
        } else {   // CONVERTED: to_diffuse_specular_material (metallic.py:98-108), then the specular workflow
            const R sp = fma_(x.alin[c], x.m, x.om * kDielectricF0);
            x.base[c] = x.alin[c] * x.om;
            x.f0[c] = sp; x.df0[c] = splat<R>(1.0f);
            if (a.spec_srgb) srgb_to_linear_and_grad(sp, x.f0[c], x.df0[c]);
        }
    }
    if (WF == PBR_WORKFLOW_METALLIC) x.kd_scale = x.om;
    x.nraw = {gather<R>(t.nm[0], g), gather<R>(t.nm[1], g), gather<R>(t.nm[2], g)};
    x.rough = gather<R>(t.ro, g);
    pixel_terms(x.nraw, V, x.rough, x.base, x.f0, x.kd_scale, x.pt);
}

// The light-independent tail of the chain rule (backward_body_to's last block): adjoints of kb / f0 / a2 / k / N.V / the unit
// normal -> gradients w.r.t. the stored texels of the group.
template <int WF, class R>
__device__ __forceinline__ void bwd_tail(const BwdTexelT<R> &x, const PixelAdjointT<R> &adj, const Vec3 &V, R (&ga)[3], R (&gn)[3], R &gr, R &gm,
                                         R (&gs)[3]) {
    // kb = kd_scale * base / pi ; kd_scale = 1 - m  (:169-174)
    R g_m = splat<R>(0.0f);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        R g_base = adj.g_kb[c] * (x.kd_scale * kInvPi);
        gs[c] = splat<R>(0.0f);
        if (WF == PBR_WORKFLOW_METALLIC) {                               // kd_scale = 1 - m; lerp(0.04, base, m)  (:107)
            g_m = fma_(adj.g_kb[c], x.base[c] * (-kInvPi), g_m);
            g_base = fma_(adj.g_f0[c], x.m, g_base);
            g_m = fma_(adj.g_f0[c], x.base[c] - kDielectricF0, g_m);
        } else if (WF == PBR_WORKFLOW_SPECULAR) {
            gs[c] = adj.g_f0[c] * x.df0[c];
        } else {   // diffuse = a (1-m) ; specular = 0.04 (1-m) + a m
            const R g_sp = adj.g_f0[c] * x.df0[c], g_diff = adj.g_kb[c] * kInvPi;
            g_base = fma_(g_diff, x.om, g_sp * x.m);
            g_m = fma_(g_sp, x.alin[c] - kDielectricF0, fma_(-g_diff, x.alin[c], g_m));
        }
        ga[c] = g_base * x.dbase[c];
    }
    gm = g_m;
    gr = fma_(adj.g_k, (x.rough + 1.0f) * 0.25f, adj.g_a2 * (x.rough * 2.0f));   // k = (r+1)^2/8, a2 = r^2
    // N.V clamp, then F.normalize: g_n = (g - n (n.g)) / |n|
    const R gv = masked(in_unit(x.pt.ndv_raw, x.pt.ndv), adj.g_ndv);
    const Vec3T<R> gnh = {fma_(gv, splat<R>(V.x), adj.g_n.x), fma_(gv, splat<R>(V.y), adj.g_n.y), fma_(gv, splat<R>(V.z), adj.g_n.z)};
    const R rn = rsq(dot_plus(x.nraw, x.nraw, 1e-24f));
    const R radial = dot(x.pt.n, gnh);
    gn[0] = (gnh.x - x.pt.n.x * radial) * rn;
    gn[1] = (gnh.y - x.pt.n.y * radial) * rn;
    gn[2] = (gnh.z - x.pt.n.z * radial) * rn;
}

// The texel state only the TAIL reads (decoded colours and their slopes, metallic, roughness, the stored normal) is parked in LDS across the
// position loop -- written once, read once (round 6).  With the tail behind the loop the gradient kernels of ONE light then need 164
// registers instead of 196: THREE waves per SIMD without a spill (2048^2 tile(2) fp32 point: 140 -> 128 us, fp16 maps 136 -> 120).  The
// loss policy and the several-lights form stay at two waves: they need 184 / 180 with the state parked, and the spills that a forced
// third wave costs them (28-52 bytes of scratch per lane) cost more than it brings (fp32 loss step 135 -> 152 us; fp16 134 -> 131).
// Round 5 had parked the tail's inputs with the tail INSIDE the loop (read back at every position): +12...21 %, not adopted then.
template <bool LOSS, bool MULTI> struct RepeatBwdShape { static constexpr bool park = true; static constexpr int waves = 3; };
template <int WF>
__device__ __forceinline__ void park_texel(float2 *slot, BwdTexelT<f32x2> &x, bool restore) {
    int i = 0;
    auto io = [&](f32x2 &v) {
        float2 *q = slot + 64 * i++;
        if (restore) { const float2 t = *q; v = f32x2{t.x, t.y}; } else { *q = make_float2(v.x, v.y); }
    };
#pragma unroll
    for (int c = 0; c < 3; ++c) { io(x.base[c]); io(x.dbase[c]); }
    io(x.m); io(x.rough); io(x.nraw.x); io(x.nraw.y); io(x.nraw.z);
    if (WF != PBR_WORKFLOW_METALLIC) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { io(x.df0[c]); io(x.alin[c]); }
        io(x.om);
    }
}
constexpr int kParkSlots = 18;

// Extra kernel arguments: where the upstream gradient (or the target image) of the OUTPUT lives.
struct RBArgs {
    const float *gout;          // upstream gradient, or with the loss policy the target image: [B][3][band rows][out_W] fp32 contiguous
    int64_t gout_cs;            // elements between its channel planes (band rows * out_W); materials are 3 of them apart
    float scale;                // loss: 2 / N
    float *partials;            // loss: one sum of squared differences per workgroup
};

//   LIGHT / WF: as everywhere.  TM: storage type of the maps and of their gradients.  LOSS: the rendering-loss step.
// KArgs as fill_repeat_args leaves them (the grid of an untiled launch over the source maps; rep_y / rep_x / out_W / out_Ht /
// y_offset / H_total describe the output) with o_cs = the MAP's plane (the gradient planes are dense [B][C][map_h * map_w]).
//   MULTI: several lights (H12: per-light clamp, sum, clamp, encode) -- per position the two passes over the lights of backward_body_to (the
//   summed colour decides the outer clamp and the encode's slope; then every light's chain rule into one adjoint); no loss policy.
//   BLEND (round 6): the texels are the blend of TWO materials under a mask (ct_blend.hpp: blend_texels, the blended normal re-decoded), formed
//   once per texel in front of the walk; behind it the folded gradients run on through the blend's own chain rule (blend_backward_sink) into
//   map-sized gradients of both materials and of the mask -- pbr_cook_torrance_blend_backward over tiled maps.  fp32 maps, no loss policy.
template <int LIGHT, int WF, typename TM, bool LOSS, bool MULTI, bool BLEND>
__device__ __forceinline__ void repeat_backward_body(const KArgs &a, const BArgs &b, const RBArgs &rb, const KBlend *kb, const BBlend *g2) {
    static_assert(!(MULTI && LOSS), "the loss step over tiled maps is built for one light");
    static_assert(!BLEND || (sizeof(TM) == 4 && !LOSS), "the fused blend's tiled backward: fp32 maps, gradients only");
    constexpr int VEC = 2;
    using R = f32x2;
    const uint32_t tile = tile_of_workgroup(a, blockIdx.x);
    const int ty = (int)a.div_tx.div(tile);
    LanePos p = lane_pos<VEC, LOSS>(a, (int)tile - ty * a.tiles_x, ty);            // over the SOURCE maps; LOSS: every lane reaches the wave sum
    if (!LOSS && !p.valid) return;
    const int PH = a.map_h;                                                        // the period of the map's rows
    if (!LOSS && a.H != PH) repeat_window(a, p);                                   // a thin band: the walk's rows are a cyclic window of the map's (ct_kernel.hpp)
    Texels<VEC> t;
    if constexpr (sizeof(TM) == 4) {
        load_texels<WF, TM, VEC, true>(a, a.has_normal != 0, p, t);
    } else if (p.sb) {
        if (a.has_normal) load_texels_fixed<WF, TM, VEC, true, true, true>(a, p, t); else load_texels_fixed<WF, TM, VEC, true, true, false>(a, p, t);
    } else {
        if (a.has_normal) load_texels_fixed<WF, TM, VEC, true, false, true>(a, p, t); else load_texels_fixed<WF, TM, VEC, true, false, false>(a, p, t);
    }
    Texels<VEC> t1, u;                        // BLEND: the two materials' own texels (the sink needs them again), `t` becomes the blended texel
    float w[VEC] = {};
    bool keep_signed = false;
    const int mat = p.sb ? p.b0 : p.b;
    if constexpr (BLEND) {
        t1 = t;
        load_texels<WF, float, VEC, true>(*kb, true, p, u);
        Ld<float, VEC>::template load<true>(kb->mask, mat * kb->k_bs + p.src, w);
        keep_signed = kb->normal_signed[mat] != 0;
        blend_texels<WF, VEC>(t, u, w, keep_signed);
    }
    // Positions k = ry * rep_x + rx in pbr_fold_gradient's order; (ry, rx) are wave-uniform (scalar plane addresses need that), whether
    // a repeat's row lies inside the band `gout` holds -- rows [y_offset, y_offset + H_total) of the tiled image -- is the lane's own test.
    const int n_pos = a.rep_y * a.rep_x;
    const uint32_t lane_out = (uint32_t)(p.y * a.out_W + p.x);                  // inside the first repeat; < 2^30 when p.sb (launch_repeat_backward)
    auto in_band = [&](int ry) { const int yy = p.y + ry * PH - a.y_offset; return yy >= 0 && yy < a.H_total; };
    auto load_upstream = [&](int k, float (&go)[3][VEC]) {
        const int ry = k / a.rep_x, rx = k - ry * a.rep_x;
        if (!in_band(ry)) return;
        const int64_t rep = ((int64_t)ry * PH - a.y_offset) * a.out_W + (int64_t)rx * a.W;      // may be negative; rep + the lane's part never is
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (p.sb) Ld<float, VEC>::template load<true>(plane_at<float>(rb.gout, (p.b0 * 3 + c) * rb.gout_cs + rep, lane_out), 0, go[c]);
            else Ld<float, VEC>::template load<true>(rb.gout, ((int64_t)p.b * 3 + c) * rb.gout_cs + rep + (int64_t)p.y * a.out_W + p.x, go[c]);
        }
    };
    float go[3][VEC] = {}, go_next[3][VEC] = {};
    // (a point light / several lights: the position loop below keeps its own running addresses; one directional light sums its positions in chunks)

    if (!a.has_normal) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) { t.nm[0][j] = 0.0f; t.nm[1][j] = 0.0f; t.nm[2][j] = 1.0f; }
    }
    const Vec3 V = view_of(a);
    const LightU lu = light_of(a, 0);
    BwdTexelT<R> x;
    bwd_decode<WF, VEC, R>(a, t, 0, V, x);

    R acc_a[3], acc_n[3], acc_s[3], acc_r = splat<R>(0.0f), acc_m = splat<R>(0.0f);
#pragma unroll
    for (int c = 0; c < 3; ++c) { acc_a[c] = splat<R>(0.0f); acc_n[c] = splat<R>(0.0f); acc_s[c] = splat<R>(0.0f); }
    float sq = 0.0f;

    if constexpr (LIGHT == PBR_LIGHT_DIRECTIONAL && !MULTI) {
        // A directional light does not know where the pixel is (:125-127): every repeat of a texel evaluates to the SAME colour, and the
        // chain rule is linear in the upstream gradient -- so the repeats' upstream values are summed first (three adds per position) and
        // the texel is differentiated ONCE: n^2 fewer evaluations, and the kernel is bound by its bytes (12 B per output pixel) instead of
        // by instruction issue.  Exact in real arithmetic; in fp32 the sum of the per-position gradients (the two-kernel form) and the
        // gradient of the summed upstream differ by rounding (~1e-7 relative), so THIS branch is not bit-identical to backward + fold.
        const LightGeomT<R> lg = light_geom<LIGHT, R>(lu, V, splat<R>(0.0f), 0.0f);
        LightEvalT<R> e;
        eval_light(x.pt, lg, lu.inten, e);
        R out[3], gsum[3] = {splat<R>(0.0f), splat<R>(0.0f), splat<R>(0.0f)};
        R enc_slope[3] = {splat<R>(1.0f), splat<R>(1.0f), splat<R>(1.0f)};
#pragma unroll
        for (int c = 0; c < 3; ++c) {                                                                           // :179-180 (the loss compares this) and its slope
            out[c] = e.uc[c];
            if (a.out_srgb) { if (LOSS) linear_to_srgb_unit_and_grad(e.uc[c], out[c], enc_slope[c]); else enc_slope[c] = linear_to_srgb_grad_unit(e.uc[c]); }
        }
        // the positions' loads do not depend on one another: four positions' worth (12 loads) in flight before the first add
        constexpr int kChunk = 4;
        for (int k0 = 0; k0 < n_pos; k0 += kChunk) {
            float gq[kChunk][3][VEC] = {};
#pragma unroll
            for (int q = 0; q < kChunk; ++q)
                if (k0 + q < n_pos) load_upstream(k0 + q, gq[q]);
#pragma unroll
            for (int q = 0; q < kChunk; ++q) {
                if (k0 + q < n_pos && in_band((k0 + q) / a.rep_x)) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        if constexpr (LOSS) {
                            const R d = out[c] - gather<R>(gq[q][c], 0);
                            if (p.valid) sq += hsum(d * d);
                            gsum[c] = gsum[c] + d;
                        } else {
                            gsum[c] = gsum[c] + gather<R>(gq[q][c], 0);
                        }
                    }
                }
            }
        }
        R g_col[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const R gc = LOSS ? gsum[c] * rb.scale : gsum[c];
            g_col[c] = a.out_srgb ? gc * enc_slope[c] : gc;
        }
        PixelAdjointT<R> adj;
#pragma unroll
        for (int c = 0; c < 3; ++c) { adj.g_kb[c] = splat<R>(0.0f); adj.g_f0[c] = splat<R>(0.0f); }
        adj.g_a2 = adj.g_k = adj.g_ndv = splat<R>(0.0f);
        adj.g_n = {splat<R>(0.0f), splat<R>(0.0f), splat<R>(0.0f)};
        LightParamAdjT<R> pa;
        backprop_light<LIGHT, false>(x.pt, lg, lu.inten, e, g_col, adj, V, pa);
        bwd_tail<WF, R>(x, adj, V, acc_a, acc_n, acc_r, acc_m, acc_s);
    } else {
        // The position loop, written for the SCALAR unit as much as for the vector one (round 6).  With two waves per SIMD a wave's own
        // scalar instructions are time its vector pipe idles unless the other wave happens to issue: the loop used to spend ~130 scalar
        // instructions and ~15 branches per position on the upstream addresses (a division k / rep_x for this position and the next, a
        // 64-bit product per channel, the scalar-base / 64-bit-lane choice tested per load) beside ~300 vector ones.  Now: ONE per-lane
        // 64-bit address for the lane's place in the first repeat (channel 0), the channel planes and the repeat as a running SCALAR
        // offset that is advanced by additions; (ry, rx) of this position and of the next as running counters; the point light's row
        // coordinate once per repeat ROW.  (Measured: 151 -> 148 us alone; with the branches below and the tail hoisted 151 -> 138.)
        const char *const lane_base = reinterpret_cast<const char *>(rb.gout) +
            4 * (((int64_t)p.b * 3) * rb.gout_cs + ((int64_t)p.y - a.y_offset) * a.out_W + p.x);     // (b, channel 0, repeat (0,0)); below the band's first row for y < y_offset: never dereferenced there
        const int64_t plane_bytes = 4 * rb.gout_cs;
        const int64_t col_step = 4 * (int64_t)a.W, row_step = 4 * ((int64_t)PH * a.out_W - (int64_t)(a.rep_x - 1) * a.W);
        auto row_in_band = [&](int yrow) { const int yy = yrow - a.y_offset; return yy >= 0 && yy < a.H_total; };
        auto fetch = [&](int64_t rep_bytes, float (&dst)[3][VEC]) {
#pragma unroll
            for (int c = 0; c < 3; ++c) Ld<float, VEC>::template load<true>(lane_base + (rep_bytes + c * plane_bytes), 0, dst[c]);
        };
        // the position in flight: loaded one iteration ahead
        int n_rx = 0, n_yrow = p.y;
        int64_t n_rep = 0;
        // The upstream values (the loss step: the target's) of a position: the plain gradient kernels keep the NEXT position's in flight under this
        // one's arithmetic (a second set of 6 registers: they fit three waves per SIMD with it); the loss step and the several-lights form request
        // a position's values at its own start -- their registers peak in backprop_light, where a set in flight was what kept them at two waves
        // (and several lights first read the values behind a whole pass over the lights).
        constexpr bool kAhead = !LOSS && !MULTI;
        if (kAhead && row_in_band(n_yrow)) fetch(n_rep, go);
        float ys = 0.0f;
        PixelAdjointT<R> adj;                   // summed over the positions; the tail is applied once, behind the loop
        auto clear_adjoint = [&]() {
#pragma unroll
            for (int c = 0; c < 3; ++c) { adj.g_kb[c] = splat<R>(0.0f); adj.g_f0[c] = splat<R>(0.0f); }
            adj.g_a2 = adj.g_k = adj.g_ndv = splat<R>(0.0f);
            adj.g_n = {splat<R>(0.0f), splat<R>(0.0f), splat<R>(0.0f)};
        };
        clear_adjoint();
        constexpr bool kPark = RepeatBwdShape<LOSS, MULTI>::park && !BLEND;       // (the blend form carries two materials' texels anyway: two waves)
        __shared__ float2 s_park[kPark ? kParkSlots * 64 : 1];
        if constexpr (kPark) park_texel<WF>(s_park + threadIdx.x, x, false);
        // The output encode is a launch-wide flag: tested once, outside the loop (two copies of the loop) -- inside it was three scalar
        // branches per position, and a branch costs a wave of a two-wave SIMD ~12 cycles (tools/valu_occupancy.hip).
        auto positions = [&](auto srgb_tag) {
        constexpr bool SRGB = decltype(srgb_tag)::value;
        for (int k = 0; k < n_pos; ++k) {
            const int rx = n_rx, yrow = n_yrow;
            const int64_t rep = n_rep;
            // advance to the next position and start its loads: they travel under this position's arithmetic
            ++n_rx; n_rep += col_step;
            if (n_rx == a.rep_x) { n_rx = 0; n_yrow += PH; n_rep += row_step - col_step; }
            if constexpr (kAhead) {
                if (k + 1 < n_pos && row_in_band(n_yrow)) fetch(n_rep, go_next);
            }
            if (row_in_band(yrow)) {                                                  // (a repeat outside this rank's band: nothing to add)
                if constexpr (!kAhead) fetch(rep, go);
                R xs[1] = {splat<R>(0.0f)};
                if (LIGHT == PBR_LIGHT_POINT) {
                    if (rx == 0) ys = linspace_at(a.y0, a.y1, a.ystep, a.out_Ht, yrow);
                    // x_grid_w's one-sided form without its wave vote: the output's width is a multiple of 4 (repeat_inner: map_w % 4 == 0),
                    // so its midpoint is even and a lane's pair of columns never straddles it -- the same statements, no branch
                    const int x0 = p.x + rx * a.W;
                    if (a.out_W & 3) {                    // ragged map widths (round 6): a pair may straddle the midpoint -- the general form
                        x_grid_w<R, 1, VEC>(a, a.out_W, x0, xs);
                    } else {
                        const bool lo = x0 < (a.out_W >> 1);
                        const float f0 = (float)(lo ? x0 : a.out_W - 1 - x0);
                        xs[0] = fma_(splat<R>(lo ? a.xstep : -a.xstep), fma_(splat<R>(lo ? 1.0f : -1.0f), lane_offsets<R>(0), splat<R>(f0)), splat<R>(lo ? a.x0 : a.x1));
                    }
                }
                if constexpr (MULTI) {
                    R g_col[3], sum[3] = {splat<R>(0.0f), splat<R>(0.0f), splat<R>(0.0f)};
                    for (int l = 0; l < a.n_lights; ++l) {          // pass 1: the summed colour decides the outer clamp / the encode's slope
                        const LightU ll = light_of(a, l);
                        LightEvalT<R> e;
                        eval_light(x.pt, light_geom<LIGHT, R>(ll, V, xs[0], ys), ll.inten, e);
    #pragma unroll
                        for (int c = 0; c < 3; ++c) sum[c] = sum[c] + e.uc[c];
                    }
    #pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const R slope = SRGB ? linear_to_srgb_grad_unit(clamp01(sum[c])) : splat<R>(1.0f);
                        g_col[c] = masked(in_unit(sum[c]), gather<R>(go[c], 0) * slope);
                    }
                    for (int l = 0; l < a.n_lights; ++l) {          // pass 2: every light's chain rule into the one adjoint
                        const LightU ll = light_of(a, l);
                        const LightGeomT<R> lg = light_geom<LIGHT, R>(ll, V, xs[0], ys);
                        LightEvalT<R> e;
                        eval_light(x.pt, lg, ll.inten, e);
                        LightParamAdjT<R> pa;
                        backprop_light<LIGHT, false>(x.pt, lg, ll.inten, e, g_col, adj, V, pa);
                    }
                } else {
                    const LightGeomT<R> lg = light_geom<LIGHT, R>(lu, V, xs[0], ys);
                    LightEvalT<R> e;
                    eval_light(x.pt, lg, lu.inten, e);
                    R gout_c[3], g_col[3];
                    if constexpr (LOSS) {
            #pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            R out = e.uc[c], slope = splat<R>(1.0f);
                            if (SRGB) linear_to_srgb_unit_and_grad(e.uc[c], out, slope);               // :179-180 and its slope, one log2
                            const R d = out - gather<R>(go[c], 0);
                            if (p.valid) sq += hsum(d * d);
                            gout_c[c] = d * rb.scale;
                            g_col[c] = SRGB ? gout_c[c] * slope : gout_c[c];
                        }
                    } else {
            #pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            gout_c[c] = gather<R>(go[c], 0);
                            g_col[c] = SRGB ? gout_c[c] * linear_to_srgb_grad_unit(e.uc[c]) : gout_c[c];
                        }
                    }
                    LightParamAdjT<R> pa;
                    backprop_light<LIGHT, false>(x.pt, lg, lu.inten, e, g_col, adj, V, pa);
                }
            }
            if constexpr (kAhead) {
    #pragma unroll
                for (int c = 0; c < 3; ++c)
    #pragma unroll
                    for (int j = 0; j < VEC; ++j) go[c][j] = go_next[c][j];
            }
        }
        };
        if (a.out_srgb) positions(std::true_type{}); else positions(std::false_type{});
        // The light-independent tail of the chain rule is LINEAR in the adjoints it receives, with coefficients that do not depend on the
        // position: the positions' adjoints were summed above (fused multiply-adds straight into `adj`), the tail runs ONCE per texel.
        if constexpr (kPark) {
            park_texel<WF>(s_park + threadIdx.x, x, true);
            if (WF == PBR_WORKFLOW_METALLIC) { x.om = splat<R>(1.0f) - x.m; x.kd_scale = x.om; }      // (1 - m: recomputed, the same subtraction)
        }
        bwd_tail<WF, R>(x, adj, V, acc_a, acc_n, acc_r, acc_m, acc_s);
    }
    if constexpr (LOSS) {
        const float total = wave_sum(sq);
        if (threadIdx.x == 0) rb.partials[blockIdx.x] = total;
        if (!p.valid) return;
    }
    float oa[3][VEC], on[3][VEC], os[3][VEC], orr[VEC], om[VEC];
#pragma unroll
    for (int c = 0; c < 3; ++c) { scatter(oa[c], 0, acc_a[c]); scatter(on[c], 0, acc_n[c]); scatter(os[c], 0, acc_s[c]); }
    scatter(orr, 0, acc_r); scatter(om, 0, acc_m);
    if constexpr (BLEND) {
        blend_backward_sink<WF, VEC>(a, p, mat, t1, u, w, keep_signed, b, *g2, oa, on, orr, om, os);
    } else {
        // the lane's place in the gradient planes is formed AGAIN here (a few scalar and vector instructions) instead of being carried across
        // the position loop: its 64-bit offsets are registers the loop does not have to hold (the loss form: the last one short of three waves)
        LanePos q = lane_pos<VEC, LOSS>(a, (int)tile - ty * a.tiles_x, ty);
        if (!LOSS && a.H != PH) repeat_window(a, q);
        asm volatile("" : "+v"(q.x), "+v"(q.y));                   // (kept opaque: otherwise the compiler merges q with p and carries p)
        q.pix = q.src = (int64_t)q.y * a.W + q.x;
        store_gradients<WF, VEC, TM>(a, b, q, oa, on, orr, om, os);
    }
}

template <int LIGHT, int WF, typename TM, bool LOSS, bool MULTI = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(RepeatBwdShape<LOSS, MULTI>::waves, 4)))
void cook_torrance_repeat_backward_kernel(const KArgs a, const BArgs b, const RBArgs rb) {
    repeat_backward_body<LIGHT, WF, TM, LOSS, MULTI, false>(a, b, rb, nullptr, nullptr);
}

template <int LIGHT, int WF>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 4)))
void cook_torrance_repeat_blend_backward_kernel(const KArgs a, const KBlend kb, const BArgs b, const BBlend g2, const RBArgs rb) {
    repeat_backward_body<LIGHT, WF, float, false, false, true>(a, b, rb, &kb, &g2);
}

}  // namespace pbr
