// ct_backward.hpp -- gradient of the fused Cook-Torrance evaluation w.r.t. the material maps
// (the documented ML use of the reference is a rendering loss,
// /root/reference/docs/source/tutorials/06_advanced.rst:73-107, where autograd differentiates
// pypbr/models/cooktorrance.py:92-182 back to albedo / normal / roughness / metallic|specular).
//
// One streaming kernel: per pixel it re-evaluates the forward terms from the maps (cheaper than
// storing ~20 intermediates per pixel: 32 B of maps + 12 B of upstream gradient in, 32 B of
// gradients out) and applies the chain rule by hand.  Sub-gradient conventions are torch's, because
// that is what the reference's autograd graph uses: clamp passes the gradient on the closed interval
// [min, max]; the piecewise sRGB functions differentiate the branch the value takes; lerp(0.04, base,
// m) gives m to base and sum_c(base_c - 0.04) to the single-channel metallic map; F.normalize
// projects out the radial component.  Light geometry does not depend on the maps.
//
// Like brdf_math.hpp the chain rule is written over the real type R: float, or f32x2 = the lane's pixels two at a
// time, so that its adds / multiplies / fmas are packed instructions; branches on per-pixel conditions are selects.
// With fp16 maps the kernel moves half the bytes and takes the same time, i.e. it is bound by instruction issue --
// which is what the packed form halves.
#pragma once
#include "ct_kernel.hpp"

namespace pbr {

struct BArgs {
    const void *gout;                      // upstream gradient, [B][3][H][W] contiguous fp32
    void *g_albedo, *g_normal, *g_rough, *g_metal, *g_spec;   // contiguous, NULL = not wanted
    float *g_param_partials;               // PGRAD kernels: [n_tiles][3 + 6 L] per-workgroup sums of the adjoints of
                                           // V (3), the lights' L | position (L x 3) and their intensities (L x 3)
};

// Adjoints of one light's parameters, for the lane's pixel group (the reference's autograd reaches view_dir,
// light_dir_or_position and light_intensity: cooktorrance.py:95-96, :126-140 are plain torch ops on them).
template <class R> struct LightParamAdjT {
    Vec3T<R> g_L;     // directional: adjoint of the normalised L; point: of the light position
    Vec3T<R> g_V;     // this light's share of the adjoint of the normalised view vector (through h = V + L and Hv.V)
    R g_I[3];
};

__device__ __forceinline__ float hsum(float v) { return v; }
__device__ __forceinline__ float hsum(f32x2 v) { return v.x + v.y; }
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// Sums EIGHT values over the wave with 10 exchanges instead of 48: three halving steps (lanes whose bit 5 / 4 / 3 is set keep
// the upper half of the values and send the lower one, the others the opposite), then a 3-step butterfly on the one value
// left.  Lane l ends up with the wave total of value number (l >> 3) & 7, the same in all 8 lanes that share those bits.
template <int C, int M>
__device__ __forceinline__ void wave_halve(float (&v)[8], bool up) {     // C values -> C / 2: exchange across lane bit M
#pragma unroll
    for (int j = 0; j < C / 2; ++j) {
        const float keep = up ? v[j + C / 2] : v[j], send = up ? v[j] : v[j + C / 2];
        v[j] = keep + __shfl_xor(send, M, 64);
    }
}
__device__ __forceinline__ float wave_sum8(float (&v)[8]) {
    const int lane = threadIdx.x & 63;
    wave_halve<8, 32>(v, (lane & 32) != 0);
    wave_halve<4, 16>(v, (lane & 16) != 0);
    wave_halve<2, 8>(v, (lane & 8) != 0);
    float t = v[0];
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    return t;
}

template <class R> using MaskT = typename MaskOf<R>::type;
template <class R> __device__ __forceinline__ MaskT<R> in_unit(R x) {
    return and_(ge_(x, splat<R>(0.0f)), le_(x, splat<R>(1.0f)));
}
template <class R> __device__ __forceinline__ R masked(MaskT<R> m, R x) { return select_(m, x, splat<R>(0.0f)); }
// 0 <= x <= 1 given clamp01(x), which the forward terms already hold: one compare per value instead of two.
template <class R> __device__ __forceinline__ MaskT<R> in_unit(R x, R clamped) { return eq_(clamped, x); }

// d/dx of utils.srgb_to_linear (functions.py:31-47): clamp (closed interval), branch, clamp.
template <class R> __device__ __forceinline__ R srgb_to_linear_grad(R x) {
    const R hi = exp2_hw(fma_(splat<R>(1.4f), log2_hw(x + 0.055f), splat<R>(-0.10814020f) /* 1.4*log2(1.055) */)) * 2.2748815f /* 2.4/1.055 */;
    const R d = select_(le_(x, splat<R>(0.04045f)), splat<R>(1.0f / 12.92f), hi);
    return masked(in_unit(x), d);
}

// utils.srgb_to_linear and its derivative from ONE log / exp pair: with u = clamp(x) + 0.055 and
// e = (u / 1.055)^1.4 = exp2(1.4 log2 u - 1.4 log2 1.055), the value is e u / 1.055 and the slope 2.4 e / 1.055.
template <class R> __device__ __forceinline__ void srgb_to_linear_and_grad(R x, R &value, R &slope) {
    const R t = clamp01(x), u = t + 0.055f;
    const R e = exp2_hw(fma_(splat<R>(1.4f), log2_hw(u), splat<R>(-0.10814020f) /* 1.4*log2(1.055) */));
    const MaskT<R> low = le_(t, splat<R>(0.04045f));
    value = select_(low, t * (1.0f / 12.92f), (e * u) * (1.0f / 1.055f));
    slope = masked(in_unit(x, t), select_(low, splat<R>(1.0f / 12.92f), e * 2.2748815f /* 2.4/1.055 */));
}

// d/dc of utils.linear_to_srgb (functions.py:50-66) for c already in [0,1].
template <class R> __device__ __forceinline__ R linear_to_srgb_grad_unit(R c) {
    const R hi = exp2_hw(log2_hw(c) * (1.0f / 2.4f - 1.0f)) * 0.43958333f /* 1.055/2.4 */;
    return select_(le_(c, splat<R>(0.0031308f)), splat<R>(12.92f), hi);
}

// utils.linear_to_srgb (functions.py:50-66) and its derivative for c already in [0,1], from ONE log2 (the loss steps need both of the
// same value): the two functions above / in brdf_math.hpp evaluate log2_hw(c) each -- the very same instruction on the very same operand,
// which the compiler does not merge across their inline assembly.  Same statements otherwise: bit-identical to calling them separately.
template <class R> __device__ __forceinline__ void linear_to_srgb_unit_and_grad(R c, R &value, R &slope) {
    const R l = log2_hw(c);
    const MaskT<R> low = le_(c, splat<R>(0.0031308f));
    const R hi = fma_sat_after_trans(splat<R>(1.055f), exp2_hw(l * (1.0f / 2.4f)), splat<R>(-0.055f));
    value = select_(low, c * 12.92f, hi);
    slope = select_(low, splat<R>(12.92f), exp2_hw(l * (1.0f / 2.4f - 1.0f)) * 0.43958333f /* 1.055/2.4 */);
}

// Forward terms of one (pixel, light) pair that the chain rule needs again.
template <class R> struct LightEvalT {
    R ndl_raw, ndl, c, s2, den, dl, dD, ds, q, dg, rad;
    MaskT<R> nh_pos;
    R F[3], u[3], uc[3];      // uc = clamp01(u): the light's clamped contribution, also the test of the clamp's sub-gradient
};

template <class R>
__device__ __forceinline__ void eval_light(const PixelTermsT<R> &t, const LightGeomT<R> &g, const float inten[3], LightEvalT<R> &e) {
    e.ndl_raw = dot(t.n, g.d) * g.rinv;                             // N.L = (N.d) rinv: the forward kernels' form (shade_light), L never formed
    e.ndl = clamp01(e.ndl_raw);
    const R nh = e.ndl_raw + t.ndv_raw;                              // N.h = N.L + N.V
    e.den = ggx_den(t, g, nh, e.s2, e.nh_pos);
    e.c = masked(e.nh_pos, nh * g.rh);                               // clamp(N.H), :215
    e.dl = fma_(e.ndl, t.omk, t.kk);
    e.dD = fma_(e.den * e.den, splat<R>(kPi), splat<R>(1e-7f));
    e.ds = fma_(t.ndv * 4.0f, e.ndl, splat<R>(1e-7f));
    e.q = rcp((e.dD * t.dv) * (e.dl * e.ds));
    e.dg = t.a2ndv * e.ndl * e.q;
    e.rad = e.ndl * g.att;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        e.F[ch] = fma_(t.f0[ch], g.om5, g.p5);
        e.u[ch] = fma_(e.F[ch], e.dg - t.kb[ch], t.kb[ch]) * (e.rad * inten[ch]);
        e.uc[ch] = clamp01(e.u[ch]);
    }
}

// Accumulators that do not depend on the light.
template <class R> struct PixelAdjointT {
    R g_kb[3];            // adjoint of kb = kd_scale * base / pi
    R g_f0[3];
    R g_a2, g_k, g_ndv;
    Vec3T<R> g_n;         // adjoint of the unit normal
};

// Chain rule through one light's contribution, given the adjoint of its clamped colour.
template <int LIGHT, bool PG, class R>
__device__ __forceinline__ void backprop_light(const PixelTermsT<R> &t, const LightGeomT<R> &g, const float inten[3],
                                               const LightEvalT<R> &e, const R g_col[3], PixelAdjointT<R> &adj,
                                               const Vec3 &V, LightParamAdjT<R> &pa) {
    R g_dg = splat<R>(0.0f), g_rad = splat<R>(0.0f), g_p5 = splat<R>(0.0f);
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        const R gu = masked(in_unit(e.u[ch], e.uc[ch]), g_col[ch]);          // clamp :177
        const R S = fma_(e.F[ch], e.dg - t.kb[ch], t.kb[ch]);                // F dg + (1 - F) kb
        const R gS = gu * (e.rad * inten[ch]);
        g_rad = fma_(gu * S, splat<R>(inten[ch]), g_rad);
        adj.g_kb[ch] = fma_(gS, splat<R>(1.0f) - e.F[ch], adj.g_kb[ch]);
        g_dg = fma_(gS, e.F[ch], g_dg);
        const R gF = gS * (e.dg - t.kb[ch]);
        adj.g_f0[ch] = fma_(gF, g.om5, adj.g_f0[ch]);                        // F = f0 + (1-f0) p5, :196
        if constexpr (PG) {
            g_p5 = fma_(gF, splat<R>(1.0f) - t.f0[ch], g_p5);
            pa.g_I[ch] = (gu * S) * e.rad;                                   // u = S rad I   (:175-176)
        }
    }
    R g_ndl = g_rad * g.att;                                                 // :175
    // dg = (a2 ndv ndl) q,  q = 1 / (dD dv dl ds)
    const R g_num = g_dg * e.q;
    const R g_Q = -(g_dg * e.dg) * e.q;                                      // adjoint of the product dD dv dl ds
    adj.g_a2 = fma_(g_num, t.ndv * e.ndl, adj.g_a2);
    R g_ndv = g_num * t.a2 * e.ndl;
    g_ndl = fma_(g_num, t.a2ndv, g_ndl);
    const R dvdl = t.dv * e.dl, dDds = e.dD * e.ds;
    const R g_dD = g_Q * (dvdl * e.ds);
    const R g_dv = g_Q * (dDds * e.dl);
    const R g_dl = g_Q * (dDds * t.dv);
    const R g_ds = g_Q * (dvdl * e.dD);
    // dD = pi den^2 + 1e-7 ; den = c^2 (a2 - 1) + 1  (:216-217)
    const R g_den = masked(e.nh_pos, g_dD * (2.0f * kPi) * e.den);
    adj.g_a2 = fma_(g_den, splat<R>(1.0f) - e.s2, adj.g_a2);                 // d den / d a2 = c^2
    const R g_c = masked(in_unit(e.c), g_den * 2.0f * e.c * (t.a2 - 1.0f));
    // dv = ndv (1-k) + k + 1e-7 ; dl likewise ; ds = 4 ndv ndl + 1e-7
    g_ndv = fma_(g_dv, t.omk, g_ndv);
    adj.g_k = fma_(g_dv, splat<R>(1.0f) - t.ndv, adj.g_k);
    g_ndl = fma_(g_dl, t.omk, g_ndl);
    adj.g_k = fma_(g_dl, splat<R>(1.0f) - e.ndl, adj.g_k);
    g_ndv = fma_(g_ds, e.ndl * 4.0f, g_ndv);
    g_ndl = fma_(g_ds, t.ndv * 4.0f, g_ndl);
    adj.g_ndv = adj.g_ndv + g_ndv;                                           // N.V does not depend on the light
    // dots -> unit normal (clamps pass on the closed interval); c = N . h / |h|
    const R gl = masked(in_unit(e.ndl_raw, e.ndl), g_ndl);
    const R gch = g_c * g.rh, glr = gl * g.rinv;                             // d N.L / d n = L = d rinv
    adj.g_n.x = fma_(glr, g.d.x, fma_(gch, g.h.x, adj.g_n.x));
    adj.g_n.y = fma_(glr, g.d.y, fma_(gch, g.h.y, adj.g_n.y));
    adj.g_n.z = fma_(glr, g.d.z, fma_(gch, g.h.z, adj.g_n.z));
    if constexpr (PG) {
        // ---- light / view parameters.  h = V + L (un-normalised); c = (n.h)/|h| (:215), cos = clamp((h.V)/|h|) (:156-158),
        // p5 = (1 - cos)^5 (:196).  d(x.h / |h|)/dh = (x - (x.h / |h|^2) h) / |h|.
        const R rh = g.rh;
        const R hv = dotu(g.h, V);
        const R cos_raw = hv * rh;
        const R om = splat<R>(1.0f) - clamp01(cos_raw), om2 = om * om;
        const R gcs = masked(in_unit(cos_raw), g_p5 * (om2 * om2) * -5.0f) * rh;
        const R nhr = (e.ndl_raw + t.ndv_raw) * g.rhh, hvr = hv * g.rhh;
        const R wh = fma_(gch, nhr, gcs * hvr);                              // g_h = gch n + gcs V - wh h
        const Vec3T<R> g_h = {fma_(gch, t.n.x, fma_(gcs, splat<R>(V.x), -wh * g.h.x)),
                              fma_(gch, t.n.y, fma_(gcs, splat<R>(V.y), -wh * g.h.y)),
                              fma_(gch, t.n.z, fma_(gcs, splat<R>(V.z), -wh * g.h.z))};
        pa.g_V = {fma_(gcs, g.h.x, g_h.x), fma_(gcs, g.h.y, g_h.y), fma_(gcs, g.h.z, g_h.z)};   // + the direct V of Hv.V
        const Vec3T<R> g_Ld = {fma_(gl, t.n.x, g_h.x), fma_(gl, t.n.y, g_h.y), fma_(gl, t.n.z, g_h.z)};   // N.L and h
        if (LIGHT == PBR_LIGHT_POINT) {
            // L = d rinv, rinv = 1/(|d| + 1e-7) (:139); att = 1/(|d|^2 + 1e-7) (:140); rad = ndl att (:175)
            const R g_rinv = dot(g_Ld, g.d);
            const R g_att = g_rad * e.ndl;
            const R coef = fma_(g_rinv * g.rdist, -(g.rinv * g.rinv), (g_att * -2.0f) * (g.att * g.att));
            pa.g_L = {fma_(g_Ld.x, g.rinv, coef * g.d.x), fma_(g_Ld.y, g.rinv, coef * g.d.y), fma_(g_Ld.z, g.rinv, coef * g.d.z)};
        } else {
            pa.g_L = g_Ld;
        }
    }
}

// Everything after the loads: forward re-evaluation, chain rule, stores.  `t` holds the lane's texels (as floats), `go` the
// upstream gradient.  s_param / n_param: the PGRAD instantiations' LDS accumulators.
//   `sink(ga, gn, gr, gm, gs)`: what happens to the lane's gradients w.r.t. the texels the shading read -- by default they are
//   stored (backward_body below); the fused blend's backward (ct_blend_backward.hpp) carries them on through the blend.
//   `loss`: NoLoss -- the upstream gradient is `go`, as loaded; MseLoss<VEC> -- the fused rendering-loss step (ct_loss.hip): the
//   upstream gradient of a pixel is formed HERE from the colour the forward re-evaluation just produced, 2 (out - target) * scale,
//   and the squared differences are summed into loss.sq.
struct NoLoss { static constexpr bool on = false; };
template <int VEC> struct MseLoss {
    static constexpr bool on = true;
    float tgt[3][VEC];      // the lane's pixels of the target image
    float scale;            // 2 / N  (d mean((out - target)^2) / d out = scale * (out - target))
    float sq;               // sum of squared differences over the lane's pixels
};

#ifndef PBR_MSE_PACKED
#define PBR_MSE_PACKED 1      // the loss step with fp32 maps in packed pairs too (A/B: build with -DPBR_MSE_PACKED=0)
#endif
template <int LIGHT, int WF, int VEC, bool MULTI, typename TM, bool PGRAD, class Sink, class Loss>
__device__ __forceinline__ void backward_body_to(const KArgs &a, const BArgs &b, const LanePos &p, Texels<VEC> &t, float (&go)[3][VEC],
                                                 float *s_param, int n_param, Sink &&sink, Loss &loss) {
    constexpr bool kPacked = sizeof(TM) == 2 || MULTI || (Loss::on && PBR_MSE_PACKED);
    using R = typename RealOf<VEC, kPacked>::type;
    constexpr int NG = RealOf<VEC, kPacked>::N;
    float accV[3] = {0.0f, 0.0f, 0.0f}, accL[3] = {0.0f, 0.0f, 0.0f}, accI[3] = {0.0f, 0.0f, 0.0f};   // one-light PGRAD
    if (!a.has_normal) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) { t.nm[0][j] = 0.0f; t.nm[1][j] = 0.0f; t.nm[2][j] = 1.0f; }
    }
    const Vec3 V = view_of(a);
    float ys = 0.0f;
    if (LIGHT == PBR_LIGHT_POINT) ys = linspace_at(a.y0, a.y1, a.ystep, a.H_total, p.y + a.y_offset);

    float ga[3][VEC], gn[3][VEC], gr[VEC], gm[VEC], gs[3][VEC];
    R xgrid[NG];
    if (LIGHT == PBR_LIGHT_POINT) x_grid<R, NG, VEC>(a, p.x, xgrid);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        // ---- forward: decoded colours and their derivatives
        R base[3], dbase[3], f0[3], df0[3], alin[3], kd_scale = splat<R>(1.0f);
        const R m = WF != PBR_WORKFLOW_SPECULAR ? gather<R>(t.me, g) : splat<R>(0.0f);
        const R om = splat<R>(1.0f) - m;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const R al = gather<R>(t.al[c], g);
            alin[c] = base[c] = al; dbase[c] = splat<R>(1.0f);
            if (a.albedo_srgb) srgb_to_linear_and_grad(al, base[c], dbase[c]);
            alin[c] = base[c];
            if (WF == PBR_WORKFLOW_METALLIC) {
                f0[c] = fma_(m, base[c], om * kDielectricF0);                  // lerp(0.04, base, m) :107
                df0[c] = splat<R>(0.0f);
            } else if (WF == PBR_WORKFLOW_SPECULAR) {
                const R sp = gather<R>(t.sp[c], g);
                f0[c] = sp; df0[c] = splat<R>(1.0f);
                if (a.spec_srgb) srgb_to_linear_and_grad(sp, f0[c], df0[c]);
            } else {   // CONVERTED: to_diffuse_specular_material (metallic.py:98-108), then the specular workflow
                const R sp = fma_(alin[c], m, om * kDielectricF0);
                base[c] = alin[c] * om;
                f0[c] = sp; df0[c] = splat<R>(1.0f);
                if (a.spec_srgb) srgb_to_linear_and_grad(sp, f0[c], df0[c]);
            }
        }
        if (WF == PBR_WORKFLOW_METALLIC) kd_scale = om;
        const Vec3T<R> nraw = {gather<R>(t.nm[0], g), gather<R>(t.nm[1], g), gather<R>(t.nm[2], g)};
        const R rough = gather<R>(t.ro, g);
        PixelTermsT<R> pt;
        pixel_terms(nraw, V, rough, base, f0, kd_scale, pt);
        const R xs = LIGHT == PBR_LIGHT_POINT ? xgrid[g] : splat<R>(0.0f);
        R gout_c[3];
        if constexpr (!Loss::on) {
#pragma unroll
            for (int c = 0; c < 3; ++c) gout_c[c] = gather<R>(go[c], g);
        }
        R enc_slope[3] = {splat<R>(1.0f), splat<R>(1.0f), splat<R>(1.0f)};       // the loss step: the encode's slope at `lin`, from the encode's own log2
        auto loss_gradient = [&](const R (&lin)[3]) {           // lin: the clamped linear colour of the pixel (:177), as the forward kernel holds it
            if constexpr (Loss::on) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    R out = lin[c];
                    if (a.out_srgb) linear_to_srgb_unit_and_grad(lin[c], out, enc_slope[c]);        // :179-180
                    const R d = out - gather<R>(loss.tgt[c], g);
                    loss.sq += hsum(d * d);
                    gout_c[c] = d * loss.scale;
                }
            }
        };

        // ---- adjoint of the linear colour before per-light clamps
        R g_col[3];
        const int nl = MULTI ? a.n_lights : 1;
        if (MULTI) {                                 // pass 1: the summed colour decides the outer clamp / encode slope
            R sum[3] = {splat<R>(0.0f), splat<R>(0.0f), splat<R>(0.0f)};
            for (int l = 0; l < nl; ++l) {
                const LightU lu = light_of(a, l);
                LightEvalT<R> e;
                eval_light(pt, light_geom<LIGHT, R>(lu, V, xs, ys), lu.inten, e);
#pragma unroll
                for (int c = 0; c < 3; ++c) sum[c] = sum[c] + e.uc[c];
            }
            if constexpr (Loss::on) {
                const R lin[3] = {clamp01(sum[0]), clamp01(sum[1]), clamp01(sum[2])};
                loss_gradient(lin);
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const R slope = Loss::on ? enc_slope[c] : (a.out_srgb ? linear_to_srgb_grad_unit(clamp01(sum[c])) : splat<R>(1.0f));
                g_col[c] = masked(in_unit(sum[c]), gout_c[c] * slope);
            }
        }
        PixelAdjointT<R> adj;
#pragma unroll
        for (int c = 0; c < 3; ++c) { adj.g_kb[c] = splat<R>(0.0f); adj.g_f0[c] = splat<R>(0.0f); }
        adj.g_a2 = adj.g_k = adj.g_ndv = splat<R>(0.0f);
        adj.g_n = {splat<R>(0.0f), splat<R>(0.0f), splat<R>(0.0f)};
        for (int l = 0; l < nl; ++l) {
            const LightU lu = light_of(a, l);
            const LightGeomT<R> lg = light_geom<LIGHT, R>(lu, V, xs, ys);
            LightEvalT<R> e;
            eval_light(pt, lg, lu.inten, e);
            if (!MULTI) {
                if constexpr (Loss::on) loss_gradient(e.uc);
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    g_col[c] = a.out_srgb ? gout_c[c] * (Loss::on ? enc_slope[c] : linear_to_srgb_grad_unit(e.uc[c])) : gout_c[c];
            }
            LightParamAdjT<R> pa;
            backprop_light<LIGHT, PGRAD>(pt, lg, lu.inten, e, g_col, adj, V, pa);
            if constexpr (PGRAD) {
                accV[0] += hsum(pa.g_V.x); accV[1] += hsum(pa.g_V.y); accV[2] += hsum(pa.g_V.z);
                if constexpr (MULTI) {        // per-light sums leave the lane here: the light loop is a run-time loop
                    float v8[8] = {hsum(pa.g_L.x), hsum(pa.g_L.y), hsum(pa.g_L.z), hsum(pa.g_I[0]), hsum(pa.g_I[1]), hsum(pa.g_I[2]), 0.0f, 0.0f};
                    const float w = wave_sum8(v8);
                    const int j = (threadIdx.x >> 3) & 7;                    // the value this lane holds the wave total of
                    if ((threadIdx.x & 7) == 0 && j < 6)
                        atomicAdd(&s_param[j < 3 ? 3 + 3 * l + j : 3 + 3 * a.n_lights + 3 * l + (j - 3)], w);
                } else {
                    accL[0] += hsum(pa.g_L.x); accL[1] += hsum(pa.g_L.y); accL[2] += hsum(pa.g_L.z);
                    accI[0] += hsum(pa.g_I[0]); accI[1] += hsum(pa.g_I[1]); accI[2] += hsum(pa.g_I[2]);
                }
            }
        }
        // ---- light-independent tail
        // kb = kd_scale * base / pi ; kd_scale = 1 - m  (:169-174)
        R g_m = splat<R>(0.0f);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            R g_base = adj.g_kb[c] * (kd_scale * kInvPi);
            if (WF == PBR_WORKFLOW_METALLIC) {                               // kd_scale = 1 - m; lerp(0.04, base, m)  (:107)
                g_m = fma_(adj.g_kb[c], base[c] * (-kInvPi), g_m);
                g_base = fma_(adj.g_f0[c], m, g_base);
                g_m = fma_(adj.g_f0[c], base[c] - kDielectricF0, g_m);
            } else if (WF == PBR_WORKFLOW_SPECULAR) {
                scatter(gs[c], g, adj.g_f0[c] * df0[c]);
            } else {   // diffuse = a (1-m) ; specular = 0.04 (1-m) + a m
                const R g_sp = adj.g_f0[c] * df0[c], g_diff = adj.g_kb[c] * kInvPi;
                g_base = fma_(g_diff, om, g_sp * m);
                g_m = fma_(g_sp, alin[c] - kDielectricF0, fma_(-g_diff, alin[c], g_m));
            }
            scatter(ga[c], g, g_base * dbase[c]);
        }
        scatter(gm, g, g_m);
        scatter(gr, g, fma_(adj.g_k, (rough + 1.0f) * 0.25f, adj.g_a2 * (rough * 2.0f)));   // k = (r+1)^2/8, a2 = r^2
        // N.V clamp, then F.normalize: g_n = (g - n (n.g)) / |n|
        const R gv = masked(in_unit(pt.ndv_raw, pt.ndv), adj.g_ndv);
        if constexpr (PGRAD) {                                                // N.V (:163): the direct share of V
            accV[0] += hsum(gv * pt.n.x); accV[1] += hsum(gv * pt.n.y); accV[2] += hsum(gv * pt.n.z);
        }
        const Vec3T<R> gnh = {fma_(gv, splat<R>(V.x), adj.g_n.x), fma_(gv, splat<R>(V.y), adj.g_n.y), fma_(gv, splat<R>(V.z), adj.g_n.z)};
        const R rn = rsq(dot_plus(nraw, nraw, 1e-24f));
        const R radial = dot(pt.n, gnh);
        scatter(gn[0], g, (gnh.x - pt.n.x * radial) * rn);
        scatter(gn[1], g, (gnh.y - pt.n.y * radial) * rn);
        scatter(gn[2], g, (gnh.z - pt.n.z * radial) * rn);
    }
    if constexpr (PGRAD) {
        {   // slots 0..2 = V; one light: 3..5 = L, 6..8 = I (n_param = 9), so values 0..7 of the first batch map to slots 0..7
            float v8[8] = {accV[0], accV[1], accV[2], MULTI ? 0.0f : accL[0], MULTI ? 0.0f : accL[1], MULTI ? 0.0f : accL[2],
                           MULTI ? 0.0f : accI[0], MULTI ? 0.0f : accI[1]};
            const float w = wave_sum8(v8);
            const int j = (threadIdx.x >> 3) & 7;
            if ((threadIdx.x & 7) == 0 && j < (MULTI ? 3 : 8)) atomicAdd(&s_param[j], w);
            if constexpr (!MULTI) {
                const float wi = wave_sum(accI[2]);
                if ((threadIdx.x & 63) == 0) atomicAdd(&s_param[8], wi);
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < n_param; i += blockDim.x) b.g_param_partials[(int64_t)blockIdx.x * n_param + i] = s_param[i];
        if (!p.valid) return;
    }
    sink(ga, gn, gr, gm, gs);
}

template <int LIGHT, int WF, int VEC, bool MULTI, typename TM, bool PGRAD, class Sink>
__device__ __forceinline__ void backward_body_to(const KArgs &a, const BArgs &b, const LanePos &p, Texels<VEC> &t, float (&go)[3][VEC],
                                                 float *s_param, int n_param, Sink &&sink) {
    NoLoss none;
    backward_body_to<LIGHT, WF, VEC, MULTI, TM, PGRAD>(a, b, p, t, go, s_param, n_param, sink, none);
}

// The default sink: the lane's gradients go to the dense gradient planes [B][C][H*W] (the result's channel stride).
template <int WF, int VEC, typename TM>
__device__ __forceinline__ void store_gradients(const KArgs &a, const BArgs &b, const LanePos &p, float (&ga)[3][VEC], float (&gn)[3][VEC],
                                                float (&gr)[VEC], float (&gm)[VEC], float (&gs)[3][VEC]) {
    auto put = [&](void *plane, int channels, int c, const float *v) {
        if (p.sb) Ld<TM, VEC>::template store<true>(plane_at<TM>(plane, ((int64_t)p.b0 * channels + c) * a.o_cs, (uint32_t)p.pix), 0, v);
        else Ld<TM, VEC>::template store<true>(plane, ((int64_t)p.b * channels + c) * a.o_cs + p.pix, v);
    };
    if (b.g_albedo) {
#pragma unroll
        for (int c = 0; c < 3; ++c) put(b.g_albedo, 3, c, ga[c]);
    }
    if (b.g_normal && a.has_normal) {
#pragma unroll
        for (int c = 0; c < 3; ++c) put(b.g_normal, 3, c, gn[c]);
    }
    if (b.g_rough) put(b.g_rough, 1, 0, gr);
    if (WF != PBR_WORKFLOW_SPECULAR) {
        if (b.g_metal) put(b.g_metal, 1, 0, gm);
    } else if (b.g_spec) {
#pragma unroll
        for (int c = 0; c < 3; ++c) put(b.g_spec, 3, c, gs[c]);
    }
}

template <int LIGHT, int WF, int VEC, bool MULTI, typename TM, bool PGRAD>
__device__ __forceinline__ void backward_body(const KArgs &a, const BArgs &b, const LanePos &p, Texels<VEC> &t, float (&go)[3][VEC],
                                              float *s_param, int n_param) {
    backward_body_to<LIGHT, WF, VEC, MULTI, TM, PGRAD>(a, b, p, t, go, s_param, n_param,
        [&](float (&ga)[3][VEC], float (&gn)[3][VEC], float (&gr)[VEC], float (&gm)[VEC], float (&gs)[3][VEC]) {
            store_gradients<WF, VEC, TM>(a, b, p, ga, gn, gr, gm, gs);
        });
}

//   LIGHT: PBR_LIGHT_*    WF: PBR_WORKFLOW_*    VEC: 4 | 2 | 1    TM: storage type of the maps AND of their gradients
//   (float | __half; arithmetic and the upstream gradient are fp32)
//   PGRAD: also the adjoints of view / light / intensity, summed over the workgroup's pixels into b.g_param_partials
//   (one row per workgroup; param_grad_finish_kernel adds the rows up).  In these instantiations no lane leaves early:
//   lanes outside the map shade a clamped (valid) position with a zero upstream gradient, so that every lane takes
//   part in the wave reductions.
#ifndef PBR_BWD_F16_WAVES
#define PBR_BWD_F16_WAVES 4
#endif
// Two-pixel lanes, one light, fp16 maps: the allocator lands on 129 VGPRs = 3 waves per SIMD, one register past 4 waves.
// (The point-light / converted-workflow body needs four registers more: held to 128 it spilled 20 bytes per lane -- found by the resource scan
// of tools/check_isa.py in round 6, present since round 4 -- so that one instantiation runs three waves per SIMD, without scratch.)
template <int LIGHT, int WF, int VEC, bool MULTI, typename TM, bool PGRAD>
constexpr int bwd_min_waves() {
    if (!(VEC == 2 && !MULTI && !PGRAD && sizeof(TM) == 2)) return 1;
    return LIGHT == PBR_LIGHT_POINT && WF == PBR_WORKFLOW_CONVERTED ? 3 : PBR_BWD_F16_WAVES;
}

template <int LIGHT, int WF, int VEC, bool MULTI, typename TM = float, bool PGRAD = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(bwd_min_waves<LIGHT, WF, VEC, MULTI, TM, PGRAD>())))
void cook_torrance_backward_kernel(const KArgs a, const BArgs b) {
    // Packed two-pixel arithmetic for fp16 maps and for several lights, as in the forward kernels: A/B on 4096^2 maps --
    // fp16 maps 182 us packed vs 194 us scalar; 4 lights fp32 423 us vs 491 us; one light fp32 221 us packed vs 206 us
    // scalar (that launch is bound by the memory system, and there the shorter arithmetic phases only make its 19
    // streams burstier).
    const uint32_t tile = tile_of_workgroup(a, blockIdx.x);
    const int ty = (int)a.div_tx.div(tile);
    const LanePos p = lane_pos<VEC, PGRAD>(a, (int)tile - ty * a.tiles_x, ty);
    if (!PGRAD && !p.valid) return;
    constexpr int kParamSlots = 3 + 6 * PBR_MAX_LIGHTS;
    __shared__ float s_param[PGRAD ? kParamSlots : 1];
    const int n_param = 3 + 6 * a.n_lights;
    if constexpr (PGRAD) {
        for (int i = threadIdx.x; i < n_param; i += blockDim.x) s_param[i] = 0.0f;
        __syncthreads();
    }
    // Texels and upstream gradient in ONE branch-free block per (scalar addresses, normal map) combination: with the flags
    // tested between the loads, the fp16 instantiations close every block with the conversions of its values, i.e. with a wait,
    // and a wave pays three memory round trips (albedo | normal | roughness, metallic | upstream gradient) instead of one.
    Texels<VEC> t;
    float go[3][VEC];
    const int64_t opix = p.b * a.o_bs + p.pix;
    auto load_all = [&](auto sb, auto hn) {
        load_texels_fixed<WF, TM, VEC, true, decltype(sb)::value, decltype(hn)::value>(a, p, t);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if constexpr (decltype(sb)::value) Ld<float, VEC>::template load<true>(plane_at<float>(b.gout, p.b0 * a.o_bs + c * a.o_cs, (uint32_t)p.pix), 0, go[c]);
            else Ld<float, VEC>::template load<true>(b.gout, opix + c * a.o_cs, go[c]);
        }
    };
    if constexpr (sizeof(TM) == 4) {            // fp32 maps: no conversions, no waits -- and the paced form measures 1 % faster (ct_kernel.hpp)
        load_texels<WF, TM, VEC, true>(a, a.has_normal != 0, p, t);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (p.sb) Ld<float, VEC>::template load<true>(plane_at<float>(b.gout, p.b0 * a.o_bs + c * a.o_cs, (uint32_t)p.pix), 0, go[c]);
            else Ld<float, VEC>::template load<true>(b.gout, opix + c * a.o_cs, go[c]);
        }
    } else if (p.sb) {
        if (a.has_normal) load_all(std::true_type{}, std::true_type{}); else load_all(std::true_type{}, std::false_type{});
    } else {
        if (a.has_normal) load_all(std::false_type{}, std::true_type{}); else load_all(std::false_type{}, std::false_type{});
    }
    if (PGRAD && !p.valid) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int j = 0; j < VEC; ++j) go[c][j] = 0.0f;
    }
    backward_body<LIGHT, WF, VEC, MULTI, TM, PGRAD>(a, b, p, t, go, s_param, n_param);
}

// ------------------------------------------------------------------ streamed form (fp16 maps, one light)
// With fp16 maps the backward pass moves 44 B per pixel and is bound by instruction issue, not by HBM: a one-tile wave
// spends about as long being dispatched, forming its 19 plane addresses and waiting for its first loads as it spends
// computing, and its 4 waves per SIMD (125 VGPRs) cannot cover that: the VALUs idle a third of the time.  Here the grid is as many
// waves as the chip holds at once (or a small multiple), and wave w of a material walks its 128-pixel tiles w, w + G,
// w + 2 G ... (G = waves per material): at any moment the chip works on one compact window of every plane, as with
// one-tile waves, every wave does the same amount of work (no tail round), and the texels and the upstream gradient of the NEXT tile travel global ->
// LDS (global_load_lds_dword: no VGPR destination, so the prefetch costs no registers) while the current tile is
// differentiated from registers; 3 waves per SIMD (the loop needs 124-150 VGPRs; a fourth wave measured level).  Per tile: 8 (10) map planes x 256 B + 3 gradient
// planes x 512 B = 3.5 (4) KiB of LDS per wave, ONE buffer -- a tile's values are copied to registers (11 ds_reads)
// before the next tile's loads are issued into the same buffer.  Plane addresses, the material index and the light block
// are formed once per wave; a tile costs a scalar row / column split and the lane-constant offset.  Same backward_body as
// cook_torrance_backward_kernel<.., 2, false, __half, false>: bit-identical gradients.
//   Requires (checked by the launcher): one light, untiled maps, W % 128 == 0, planes below 2^30 pixels, every plane
//   4-byte aligned with even strides, no light / view adjoints.
#ifndef PBR_BWD_STREAM_WAVES
#define PBR_BWD_STREAM_WAVES 3
#endif
constexpr int kStreamWavesPerSimd = PBR_BWD_STREAM_WAVES;
constexpr int kStreamMapPlanes = 10;                                 // albedo 3, normal 3, roughness, metallic | specular 3
constexpr int kStreamLdsWords = kStreamMapPlanes * 64 + 3 * 128;
typedef __attribute__((address_space(3))) void *lds_ptr;

// One LDS-DMA load: lane l's dword goes to LDS byte OFF + 4 l of the tile buffer.  M0 (the LDS base of the instruction) is
// the buffer for EVERY load of the kernel -- the plane's place inside it travels in the instruction's offset field, which
// the hardware adds to the global address too, so the plane's (scalar) base is biased by -OFF: no s_mov m0 + wait state
// per load (14 loads per tile).
template <typename T, int OFF>
__device__ __forceinline__ void dma_dword(const void *plane, int64_t uniform_elems, uint32_t lane_elems, uint32_t *buf) {
    const uint64_t base = reinterpret_cast<uint64_t>(plane) + (uint64_t)uniform_elems * sizeof(T) - (uint64_t)OFF;
    const __attribute__((address_space(1))) void *g = (const __attribute__((address_space(1))) void *)(reinterpret_cast<global_ptr>(base) + lane_elems * (uint32_t)sizeof(T));
    __builtin_amdgcn_global_load_lds(g, (lds_ptr)buf, 4, OFF, 2);
}

// The body, shared with the streamed rendering-loss step (ct_loss.hip): with a Loss policy `b.gout` is the TARGET image -- same three fp32
// planes, same DMA slots -- and the upstream gradient is formed from it inside backward_body_to.
template <int LIGHT, int WF, bool FULL, class Loss>
__device__ __forceinline__ void backward_stream_body(const KArgs &a, const BArgs &b, const int tiles_per_material, const int n_stores, uint32_t *buf, Loss &loss) {
    const int lane = threadIdx.x, mat = blockIdx.y;
    const int t0 = blockIdx.x, t1 = tiles_per_material, step = gridDim.x;      // tiles t0, t0 + step, ... of material `mat`
    if (FULL) {      // the usual rendering-loss launch: sRGB in and out, a normal map, every gradient wanted -- no flag branches
        // exactly the facts the launcher has checked for this workflow (a false assumption would be undefined behaviour): the
        // specular decode flag only where the workflow reads it, the metallic | specular gradient only where it is written
        __builtin_assume(a.albedo_srgb != 0); __builtin_assume(a.out_srgb != 0); __builtin_assume(a.has_normal != 0);
        __builtin_assume(b.g_albedo != nullptr); __builtin_assume(b.g_normal != nullptr); __builtin_assume(b.g_rough != nullptr);
        if (WF != PBR_WORKFLOW_METALLIC) __builtin_assume(a.spec_srgb != 0);
        if (WF == PBR_WORKFLOW_SPECULAR) __builtin_assume(b.g_spec != nullptr); else __builtin_assume(b.g_metal != nullptr);
    }
    const bool has_normal = FULL || a.has_normal != 0;
    auto issue = [&](int t) {
        const uint32_t l2 = (uint32_t)t * 128u + 2u * lane, l1 = (uint32_t)t * 128u + lane;   // lane offsets in pixels: pairs | singles
        dma_dword<__half, 0 * 256>(a.albedo, mat * a.a_bs, l2, buf);
        dma_dword<__half, 1 * 256>(a.albedo, mat * a.a_bs + a.a_cs, l2, buf);
        dma_dword<__half, 2 * 256>(a.albedo, mat * a.a_bs + 2 * a.a_cs, l2, buf);
        if (has_normal) {
            dma_dword<__half, 3 * 256>(a.normal, mat * a.n_bs, l2, buf);
            dma_dword<__half, 4 * 256>(a.normal, mat * a.n_bs + a.n_cs, l2, buf);
            dma_dword<__half, 5 * 256>(a.normal, mat * a.n_bs + 2 * a.n_cs, l2, buf);
        }
        dma_dword<__half, 6 * 256>(a.rough, mat * a.r_bs, l2, buf);
        if (WF != PBR_WORKFLOW_SPECULAR) {
            dma_dword<__half, 7 * 256>(a.metal, mat * a.m_bs, l2, buf);
        } else {
            dma_dword<__half, 7 * 256>(a.spec, mat * a.s_bs, l2, buf);
            dma_dword<__half, 8 * 256>(a.spec, mat * a.s_bs + a.s_cs, l2, buf);
            dma_dword<__half, 9 * 256>(a.spec, mat * a.s_bs + 2 * a.s_cs, l2, buf);
        }
        constexpr int G = kStreamMapPlanes * 256;                // the upstream gradient: 128 floats per plane, two loads each
        dma_dword<float, G + 0>(b.gout, mat * a.o_bs, l1, buf);
        dma_dword<float, G + 256>(b.gout, mat * a.o_bs + 64, l1, buf);
        dma_dword<float, G + 512>(b.gout, mat * a.o_bs + a.o_cs, l1, buf);
        dma_dword<float, G + 768>(b.gout, mat * a.o_bs + a.o_cs + 64, l1, buf);
        dma_dword<float, G + 1024>(b.gout, mat * a.o_bs + 2 * a.o_cs, l1, buf);
        dma_dword<float, G + 1280>(b.gout, mat * a.o_bs + 2 * a.o_cs + 64, l1, buf);
    };
    // The tile's values leave LDS through hand-written ds_reads: for an LDS read the compiler can see it waits until EVERY
    // vector-memory operation in flight has finished (it knows the buffer is the target of LDS-DMA loads, not which of
    // them) -- that includes the previous tile's stores, a full write round trip per tile.
    const uint32_t lds0 = (uint32_t)(size_t)(lds_ptr)buf;
    const uint32_t at4 = lds0 + 4u * lane, at8 = lds0 + kStreamMapPlanes * 256u + 8u * lane;
    issue(t0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int t = t0; t < t1; t += step) {
        // the tile's loads were issued BEFORE the previous tile's stores (vmcnt counts both, in order): when at most the
        // stores are still in flight, the loads have landed
        if (FULL && WF != PBR_WORKFLOW_SPECULAR) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (FULL) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else if (n_stores == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (n_stores == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        uint32_t w[kStreamMapPlanes];
        float2 g2[3];
#pragma unroll
        for (int q = 0; q < kStreamMapPlanes; ++q) {
            w[q] = 0u;
            const bool used = q < 3 || (q < 6 && has_normal) || q == 6 || q == 7 || (q > 7 && WF == PBR_WORKFLOW_SPECULAR);
            if (used) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(w[q]) : "v"(at4), "i"(q * 256) : "memory");
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(g2[c]) : "v"(at8), "i"(c * 512) : "memory");
        // the values are in registers: the buffer may be overwritten (every later use of them depends on this wait)
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]), "+v"(w[8]), "+v"(w[9]),
                       "+v"(g2[0]), "+v"(g2[1]), "+v"(g2[2]) :: "memory");
        Texels<2> tx;
        float go[3][2];
        auto halves = [&](int plane, float v[2]) {
            const f16x2 h = __builtin_bit_cast(f16x2, w[plane]);
            v[0] = (float)h.x; v[1] = (float)h.y;
        };
#pragma unroll
        for (int c = 0; c < 3; ++c) halves(c, tx.al[c]);
        if (has_normal) {
#pragma unroll
            for (int c = 0; c < 3; ++c) halves(3 + c, tx.nm[c]);
        }
        halves(6, tx.ro);
        if (WF != PBR_WORKFLOW_SPECULAR) {
            halves(7, tx.me);
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) halves(7 + c, tx.sp[c]);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) { go[c][0] = g2[c].x; go[c][1] = g2[c].y; }
        if (t + step < t1) issue(t + step);
        const int ty = (int)a.div_tx.div((uint32_t)t);          // row of the material; tiles_x = W / 128 tiles per row
        LanePos p;
        p.b = p.b0 = mat; p.y = ty; p.x = (t - ty * a.tiles_x) * 128 + 2 * lane;
        p.pix = p.src = (int64_t)t * 128 + 2 * lane;
        p.valid = true; p.sb = true; p.dup = 0;
        if constexpr (Loss::on) {
#pragma unroll
            for (int c = 0; c < 3; ++c) { loss.tgt[c][0] = go[c][0]; loss.tgt[c][1] = go[c][1]; }
            backward_body_to<LIGHT, WF, 2, false, __half, false>(a, b, p, tx, go, nullptr, 0,
                [&](float (&ga)[3][2], float (&gn)[3][2], float (&gr)[2], float (&gm)[2], float (&gs)[3][2]) {
                    store_gradients<WF, 2, __half>(a, b, p, ga, gn, gr, gm, gs);
                }, loss);
        } else {
            backward_body<LIGHT, WF, 2, false, __half, false>(a, b, p, tx, go, nullptr, 0);
        }
    }
}

template <int LIGHT, int WF, bool FULL>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(PBR_BWD_STREAM_WAVES)))
void cook_torrance_backward_stream_kernel(const KArgs a, const BArgs b, const int tiles_per_material, const int n_stores) {
    __shared__ uint32_t buf[kStreamLdsWords];
    NoLoss none;
    backward_stream_body<LIGHT, WF, FULL>(a, b, tiles_per_material, n_stores, buf, none);
}

// (A form of this kernel with 16-byte memory instructions -- 4 + 2 vector-memory instructions per tile instead of 14 + 8 -- was built in round 3
// and measured level, 143.5 against 142.9 us: the kernel is not limited by how its requests are shaped.  Removed in round 5; profiles/EXPERIMENTS.md.)

#ifdef PBR_PARAM_GRAD_KERNELS      // the two reduction kernels are not templates: defined in ct_backward.hip only (the header is shared)
// Adds up the per-workgroup rows of the PGRAD kernels (fp64 sums, fixed order: deterministic) and applies the part of
// the chain rule that sits in front of the kernel: view_dir and a directional light enter through F.normalize
// (cooktorrance.py:95, :126), whose Jacobian is (I - v v^T) / max(|x|, 1e-12).  One workgroup per 3-vector:
// vector 0 = view, 1..L = lights, L+1..2L = intensities.  out: [3 + 6 L] floats in that order.
// Stage 1 of the row sum: kParamStageRows workgroups, each adds a contiguous block of rows.  The rows are read as one flat
// array, lane t taking elements t, t + T, t + 2 T ... with T the largest multiple of n_param <= 256, so a lane always
// meets the same parameter and the loads are coalesced; fp64, fixed order.  (One workgroup per 3-vector walking all
// 131 072 rows of a 4096^2 launch by itself took 155 us -- as long as the kernel that produced them.)
constexpr int kParamStageRows = 256;
__global__ __launch_bounds__(256) void param_grad_stage_kernel(const float *__restrict__ partials, double *__restrict__ stage,
                                                               int n_rows, int n_param) {
    const int T = (256 / n_param) * n_param, per = (n_rows + kParamStageRows - 1) / kParamStageRows;
    const int r0 = blockIdx.x * per, r1 = min(n_rows, r0 + per);
    __shared__ double part[256];
    double s = 0.0;
    if ((int)threadIdx.x < T && r0 < r1) {
        const float *p = partials + (int64_t)r0 * n_param;
        const int64_t n = (int64_t)(r1 - r0) * n_param;
        for (int64_t e = threadIdx.x; e < n; e += T) s += p[e];
    }
    part[threadIdx.x] = s;
    __syncthreads();
    if ((int)threadIdx.x < n_param) {
        double t = 0.0;
        for (int k = threadIdx.x; k < T; k += n_param) t += part[k];
        stage[(int64_t)blockIdx.x * n_param + threadIdx.x] = t;
    }
}

struct ParamFinishArgs {
    const double *stage; float *out; int32_t n_rows, n_lights, light_type;      // n_rows = kParamStageRows rows of stage sums
    float view[3]; float lights[PBR_MAX_LIGHTS][3];
    uint64_t dev;                     // parameters in device memory (address of the DevParams block): the raw vectors come from it
};
__global__ __launch_bounds__(256) void param_grad_finish_kernel(const ParamFinishArgs a) {
    const int vec = blockIdx.x, n_param = 3 + 6 * a.n_lights;
    double s[3] = {0.0, 0.0, 0.0};
    for (int r = threadIdx.x; r < a.n_rows; r += 256) {
        const double *row = a.stage + (int64_t)r * n_param + 3 * vec;
        s[0] += row[0]; s[1] += row[1]; s[2] += row[2];
    }
    __shared__ double red[3][256];
#pragma unroll
    for (int j = 0; j < 3; ++j) red[j][threadIdx.x] = s[j];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
#pragma unroll
            for (int j = 0; j < 3; ++j) red[j][threadIdx.x] += red[j][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    double g[3] = {red[0][0], red[1][0], red[2][0]};
    const bool normalised = vec == 0 || (vec <= a.n_lights && a.light_type == PBR_LIGHT_DIRECTIONAL);
    if (normalised) {
        float x[3];
        if (a.dev) {
            const DevParams *d = reinterpret_cast<const DevParams *>(a.dev);
            for (int j = 0; j < 3; ++j) x[j] = vec == 0 ? d->raw_view[j] : d->raw_lights[vec - 1][j];
        } else {
            for (int j = 0; j < 3; ++j) x[j] = vec == 0 ? a.view[j] : a.lights[vec - 1][j];
        }
        const double nrm = sqrt((double)x[0] * x[0] + (double)x[1] * x[1] + (double)x[2] * x[2]);
        if (nrm > 1e-12) {
            const double u[3] = {x[0] / nrm, x[1] / nrm, x[2] / nrm};
            const double ug = u[0] * g[0] + u[1] * g[1] + u[2] * g[2];
            for (int j = 0; j < 3; ++j) g[j] = (g[j] - u[j] * ug) / nrm;
        } else {                     // F.normalize divides by the clamp there: d(x / 1e-12)/dx
            for (int j = 0; j < 3; ++j) g[j] *= 1e12;
        }
    }
    for (int j = 0; j < 3; ++j) a.out[3 * vec + j] = (float)g[j];
}

#endif  // PBR_PARAM_GRAD_KERNELS

}  // namespace pbr
