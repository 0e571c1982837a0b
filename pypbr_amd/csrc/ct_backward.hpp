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
};

template <class R> using MaskT = typename MaskOf<R>::type;
template <class R> __device__ __forceinline__ MaskT<R> in_unit(R x) {
    return and_(ge_(x, splat<R>(0.0f)), le_(x, splat<R>(1.0f)));
}
template <class R> __device__ __forceinline__ R masked(MaskT<R> m, R x) { return select_(m, x, splat<R>(0.0f)); }

// d/dx of utils.srgb_to_linear (functions.py:31-47): clamp (closed interval), branch, clamp.
template <class R> __device__ __forceinline__ R srgb_to_linear_grad(R x) {
    const R hi = exp2_hw(fma_(splat<R>(1.4f), log2_hw(x + 0.055f), splat<R>(-0.10814020f) /* 1.4*log2(1.055) */)) * 2.2748815f /* 2.4/1.055 */;
    const R d = select_(le_(x, splat<R>(0.04045f)), splat<R>(1.0f / 12.92f), hi);
    return masked(in_unit(x), d);
}

// d/dc of utils.linear_to_srgb (functions.py:50-66) for c already in [0,1].
template <class R> __device__ __forceinline__ R linear_to_srgb_grad_unit(R c) {
    const R hi = exp2_hw(log2_hw(c) * (1.0f / 2.4f - 1.0f)) * 0.43958333f /* 1.055/2.4 */;
    return select_(le_(c, splat<R>(0.0031308f)), splat<R>(12.92f), hi);
}

// Forward terms of one (pixel, light) pair that the chain rule needs again.
template <class R> struct LightEvalT {
    R ndl_raw, ndl, c, s2, den, dl, dD, ds, q, dg, rad;
    MaskT<R> nh_pos;
    R F[3], u[3];
};

template <class R>
__device__ __forceinline__ void eval_light(const PixelTermsT<R> &t, const LightGeomT<R> &g, const float inten[3], LightEvalT<R> &e) {
    e.ndl_raw = dot(t.n, g.L);
    e.ndl = clamp01(e.ndl_raw);
    const R nh = e.ndl_raw + t.ndv_raw;                              // N.h = N.L + N.V
    e.den = ggx_den(t, g, nh, e.s2, e.nh_pos);
    e.c = masked(e.nh_pos, nh * sqrt_hw(g.rhh));                     // clamp(N.H), :215
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
template <class R>
__device__ __forceinline__ void backprop_light(const PixelTermsT<R> &t, const LightGeomT<R> &g, const float inten[3],
                                               const LightEvalT<R> &e, const R g_col[3], PixelAdjointT<R> &adj) {
    R g_dg = splat<R>(0.0f), g_rad = splat<R>(0.0f);
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        const R gu = masked(in_unit(e.u[ch]), g_col[ch]);                    // clamp :177
        const R S = fma_(e.F[ch], e.dg - t.kb[ch], t.kb[ch]);                // F dg + (1 - F) kb
        const R gS = gu * (e.rad * inten[ch]);
        g_rad = fma_(gu * S, splat<R>(inten[ch]), g_rad);
        adj.g_kb[ch] = fma_(gS, splat<R>(1.0f) - e.F[ch], adj.g_kb[ch]);
        g_dg = fma_(gS, e.F[ch], g_dg);
        const R gF = gS * (e.dg - t.kb[ch]);
        adj.g_f0[ch] = fma_(gF, g.om5, adj.g_f0[ch]);                        // F = f0 + (1-f0) p5, :196
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
    const R gl = masked(in_unit(e.ndl_raw), g_ndl);
    const R gch = g_c * sqrt_hw(g.rhh);
    adj.g_n.x = fma_(gl, g.L.x, fma_(gch, g.h.x, adj.g_n.x));
    adj.g_n.y = fma_(gl, g.L.y, fma_(gch, g.h.y, adj.g_n.y));
    adj.g_n.z = fma_(gl, g.L.z, fma_(gch, g.h.z, adj.g_n.z));
}

//   LIGHT: PBR_LIGHT_*    WF: PBR_WORKFLOW_*    VEC: 4 | 1    TM: storage type of the maps AND of their gradients
//   (float | __half; arithmetic and the upstream gradient are fp32)
template <int LIGHT, int WF, int VEC, bool MULTI, typename TM = float>
__global__ __launch_bounds__(256) void cook_torrance_backward_kernel(const KArgs a, const BArgs b) {
    // Packed two-pixel arithmetic for fp16 maps and for several lights, as in the forward kernels: A/B on 4096^2 maps --
    // fp16 maps 182 us packed vs 194 us scalar; 4 lights fp32 423 us vs 491 us; one light fp32 221 us packed vs 206 us
    // scalar (that launch is bound by the memory system, and there the shorter arithmetic phases only make its 19
    // streams burstier).
    constexpr bool kPacked = sizeof(TM) == 2 || MULTI;
    using R = typename RealOf<VEC, kPacked>::type;
    constexpr int NG = RealOf<VEC, kPacked>::N;
    const uint32_t tile = tile_of_workgroup(a, blockIdx.x);
    const int ty = (int)a.div_tx.div(tile);
    const LanePos p = lane_pos<VEC>(a, (int)tile - ty * a.tiles_x, ty);
    if (!p.valid) return;
    Texels<VEC> t;
    load_texels<WF, TM, VEC, true>(a, a.has_normal != 0, p, t);
    float go[3][VEC];
    const int64_t opix = p.b * a.o_bs + p.pix;
#pragma unroll
    for (int c = 0; c < 3; ++c) Ld<float, VEC>::template load<true>(b.gout, opix + c * a.o_cs, go[c]);
    if (!a.has_normal) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) { t.nm[0][j] = 0.0f; t.nm[1][j] = 0.0f; t.nm[2][j] = 1.0f; }
    }
    const Vec3 V = {a.V[0], a.V[1], a.V[2]};
    float ys = 0.0f;
    if (LIGHT == PBR_LIGHT_POINT) ys = linspace_at(a.y0, a.y1, a.ystep, a.H_total, p.y + a.y_offset);

    float ga[3][VEC], gn[3][VEC], gr[VEC], gm[VEC], gs[3][VEC];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        // ---- forward: decoded colours and their derivatives
        R base[3], dbase[3], f0[3], df0[3], alin[3], kd_scale = splat<R>(1.0f);
        const R m = WF != PBR_WORKFLOW_SPECULAR ? gather<R>(t.me, g) : splat<R>(0.0f);
        const R om = splat<R>(1.0f) - m;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const R al = gather<R>(t.al[c], g);
            alin[c] = base[c] = a.albedo_srgb ? srgb_to_linear(al) : al;
            dbase[c] = a.albedo_srgb ? srgb_to_linear_grad(al) : splat<R>(1.0f);
            if (WF == PBR_WORKFLOW_METALLIC) {
                f0[c] = fma_(m, base[c], om * kDielectricF0);                  // lerp(0.04, base, m) :107
                df0[c] = splat<R>(0.0f);
            } else if (WF == PBR_WORKFLOW_SPECULAR) {
                const R sp = gather<R>(t.sp[c], g);
                f0[c] = a.spec_srgb ? srgb_to_linear(sp) : sp;
                df0[c] = a.spec_srgb ? srgb_to_linear_grad(sp) : splat<R>(1.0f);
            } else {   // CONVERTED: to_diffuse_specular_material (metallic.py:98-108), then the specular workflow
                const R sp = fma_(alin[c], m, om * kDielectricF0);
                base[c] = alin[c] * om;
                f0[c] = a.spec_srgb ? srgb_to_linear(sp) : sp;
                df0[c] = a.spec_srgb ? srgb_to_linear_grad(sp) : splat<R>(1.0f);
            }
        }
        if (WF == PBR_WORKFLOW_METALLIC) kd_scale = om;
        const Vec3T<R> nraw = {gather<R>(t.nm[0], g), gather<R>(t.nm[1], g), gather<R>(t.nm[2], g)};
        const R rough = gather<R>(t.ro, g);
        PixelTermsT<R> pt;
        pixel_terms(nraw, V, rough, base, f0, kd_scale, pt);
        const R xs = LIGHT == PBR_LIGHT_POINT ? xs_of<R>(a, p.x, g) : splat<R>(0.0f);
        R gout_c[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) gout_c[c] = gather<R>(go[c], g);

        // ---- adjoint of the linear colour before per-light clamps
        R g_col[3];
        const int nl = MULTI ? a.n_lights : 1;
        if (MULTI) {                                 // pass 1: the summed colour decides the outer clamp / encode slope
            R sum[3] = {splat<R>(0.0f), splat<R>(0.0f), splat<R>(0.0f)};
            for (int l = 0; l < nl; ++l) {
                const LightU &lu = a.lights[l];
                LightEvalT<R> e;
                eval_light(pt, light_geom<LIGHT, R>(lu, V, xs, ys), lu.inten, e);
#pragma unroll
                for (int c = 0; c < 3; ++c) sum[c] = sum[c] + clamp01(e.u[c]);
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const R slope = a.out_srgb ? linear_to_srgb_grad_unit(clamp01(sum[c])) : splat<R>(1.0f);
                g_col[c] = masked(in_unit(sum[c]), gout_c[c] * slope);
            }
        }
        PixelAdjointT<R> adj;
#pragma unroll
        for (int c = 0; c < 3; ++c) { adj.g_kb[c] = splat<R>(0.0f); adj.g_f0[c] = splat<R>(0.0f); }
        adj.g_a2 = adj.g_k = adj.g_ndv = splat<R>(0.0f);
        adj.g_n = {splat<R>(0.0f), splat<R>(0.0f), splat<R>(0.0f)};
        for (int l = 0; l < nl; ++l) {
            const LightU &lu = a.lights[l];
            const LightGeomT<R> lg = light_geom<LIGHT, R>(lu, V, xs, ys);
            LightEvalT<R> e;
            eval_light(pt, lg, lu.inten, e);
            if (!MULTI) {
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    g_col[c] = a.out_srgb ? gout_c[c] * linear_to_srgb_grad_unit(clamp01(e.u[c])) : gout_c[c];
            }
            backprop_light(pt, lg, lu.inten, e, g_col, adj);
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
        const R gv = masked(in_unit(pt.ndv_raw), adj.g_ndv);
        const Vec3T<R> gnh = {fma_(gv, splat<R>(V.x), adj.g_n.x), fma_(gv, splat<R>(V.y), adj.g_n.y), fma_(gv, splat<R>(V.z), adj.g_n.z)};
        const R rn = rsq(dot_plus(nraw, nraw, 1e-24f));
        const R radial = dot(pt.n, gnh);
        scatter(gn[0], g, (gnh.x - pt.n.x * radial) * rn);
        scatter(gn[1], g, (gnh.y - pt.n.y * radial) * rn);
        scatter(gn[2], g, (gnh.z - pt.n.z * radial) * rn);
    }
    const int64_t gp3 = (int64_t)p.b * 3 * a.o_cs + p.pix, gp1 = (int64_t)p.b * a.o_cs + p.pix;
    if (b.g_albedo) {
#pragma unroll
        for (int c = 0; c < 3; ++c) Ld<TM, VEC>::template store<true>(b.g_albedo, gp3 + c * a.o_cs, ga[c]);
    }
    if (b.g_normal && a.has_normal) {
#pragma unroll
        for (int c = 0; c < 3; ++c) Ld<TM, VEC>::template store<true>(b.g_normal, gp3 + c * a.o_cs, gn[c]);
    }
    if (b.g_rough) Ld<TM, VEC>::template store<true>(b.g_rough, gp1, gr);
    if (WF != PBR_WORKFLOW_SPECULAR) {
        if (b.g_metal) Ld<TM, VEC>::template store<true>(b.g_metal, gp1, gm);
    } else if (b.g_spec) {
#pragma unroll
        for (int c = 0; c < 3; ++c) Ld<TM, VEC>::template store<true>(b.g_spec, gp3 + c * a.o_cs, gs[c]);
    }
}

}  // namespace pbr
