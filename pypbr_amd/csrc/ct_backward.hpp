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
#pragma once
#include "ct_kernel.hpp"

namespace pbr {

struct BArgs {
    const void *gout;                      // upstream gradient, [B][3][H][W] contiguous fp32
    void *g_albedo, *g_normal, *g_rough, *g_metal, *g_spec;   // contiguous, NULL = not wanted
};

// d/dx of utils.srgb_to_linear (functions.py:31-47): clamp (closed interval), branch, clamp.
__device__ __forceinline__ float srgb_to_linear_grad(float x) {
    const float lo = 1.0f / 12.92f;
    const float hi = 2.2748815f /* 2.4/1.055 */ * exp2_hw(fmaf(1.4f, log2_hw(x + 0.055f), -0.10814020f /* 1.4*log2(1.055) */));
    const float d = x <= 0.04045f ? lo : hi;
    return (x >= 0.0f && x <= 1.0f) ? d : 0.0f;
}

// d/dc of utils.linear_to_srgb (functions.py:50-66) for c already in [0,1].
__device__ __forceinline__ float linear_to_srgb_grad_unit(float c) {
    const float hi = 0.43958333f /* 1.055/2.4 */ * exp2_hw(log2_hw(c) * (1.0f / 2.4f - 1.0f));
    return c <= 0.0031308f ? 12.92f : hi;
}

__device__ __forceinline__ bool in_unit(float x) { return x >= 0.0f && x <= 1.0f; }

// Forward terms of one (pixel, light) pair that the chain rule needs again.
struct LightEval {
    float ndl_raw, ndl, c, s2, den, dl, dD, ds, q, dg, rad;
    bool nh_pos;
    float F[3], u[3];
};

__device__ __forceinline__ void eval_light(const PixelTerms &t, const LightGeom &g, const float inten[3], LightEval &e) {
    e.ndl_raw = dot(t.n, g.L);
    e.ndl = clamp01(e.ndl_raw);
    const float nh = e.ndl_raw + t.ndv_raw;                          // N.h = N.L + N.V
    e.den = ggx_den(t, g, nh, e.s2, e.nh_pos);
    e.c = e.nh_pos ? nh * sqrt_hw(g.rhh) : 0.0f;                    // clamp(N.H), :215
    e.dl = fmaf(e.ndl, 1.0f - t.k, t.k) + 1e-7f;
    e.dD = fmaf(kPi, e.den * e.den, 1e-7f);
    e.ds = fmaf(4.0f * t.ndv, e.ndl, 1e-7f);
    e.q = rcp((e.dD * t.dv) * (e.dl * e.ds));
    e.dg = t.a2ndv * e.ndl * e.q;
    e.rad = e.ndl * g.att;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        e.F[ch] = fmaf(1.0f - t.f0[ch], g.p5, t.f0[ch]);
        e.u[ch] = fmaf(e.F[ch], e.dg - t.kb[ch], t.kb[ch]) * (inten[ch] * e.rad);
    }
}

// Accumulators that do not depend on the light.
struct PixelAdjoint {
    float g_kb[3];        // adjoint of kb = kd_scale * base / pi
    float g_f0[3];
    float g_a2, g_k, g_ndv;
    Vec3 g_n;             // adjoint of the unit normal
};

// Chain rule through one light's contribution, given the adjoint of its clamped colour.
__device__ __forceinline__ void backprop_light(const PixelTerms &t, const LightGeom &g, const float inten[3],
                                               const LightEval &e, const float g_col[3], PixelAdjoint &adj) {
    float g_dg = 0.0f, g_rad = 0.0f;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        const float gu = in_unit(e.u[ch]) ? g_col[ch] : 0.0f;               // clamp :177
        const float S = fmaf(e.F[ch], e.dg - t.kb[ch], t.kb[ch]);           // F dg + (1 - F) kb
        const float gS = gu * (inten[ch] * e.rad);
        g_rad = fmaf(gu * S, inten[ch], g_rad);
        adj.g_kb[ch] = fmaf(gS, 1.0f - e.F[ch], adj.g_kb[ch]);
        g_dg = fmaf(gS, e.F[ch], g_dg);
        const float gF = gS * (e.dg - t.kb[ch]);
        adj.g_f0[ch] = fmaf(gF, 1.0f - g.p5, adj.g_f0[ch]);                  // F = f0 + (1-f0) p5, :196
    }
    float g_ndl = g_rad * g.att;                                             // :175
    // dg = (a2 ndv ndl) q,  q = 1 / (dD dv dl ds)
    const float g_num = g_dg * e.q;
    const float g_Q = -g_dg * e.dg * e.q;                                    // adjoint of the product dD dv dl ds
    adj.g_a2 = fmaf(g_num, t.ndv * e.ndl, adj.g_a2);
    float g_ndv = g_num * t.a2 * e.ndl;
    g_ndl = fmaf(g_num, t.a2ndv, g_ndl);
    const float g_dD = g_Q * (t.dv * e.dl * e.ds);
    const float g_dv = g_Q * (e.dD * e.dl * e.ds);
    const float g_dl = g_Q * (e.dD * t.dv * e.ds);
    const float g_ds = g_Q * (e.dD * t.dv * e.dl);
    // dD = pi den^2 + 1e-7 ; den = c^2 (a2 - 1) + 1  (:216-217)
    const float g_den = g_dD * (2.0f * kPi) * e.den;
    float g_c = 0.0f;
    if (e.nh_pos) {
        adj.g_a2 = fmaf(g_den, 1.0f - e.s2, adj.g_a2);                       // d den / d a2 = c^2
        g_c = in_unit(e.c) ? g_den * 2.0f * e.c * (t.a2 - 1.0f) : 0.0f;
    }
    // dv = ndv (1-k) + k + 1e-7 ; dl likewise ; ds = 4 ndv ndl + 1e-7
    const float omk = 1.0f - t.k;
    g_ndv = fmaf(g_dv, omk, g_ndv);
    adj.g_k = fmaf(g_dv, 1.0f - t.ndv, adj.g_k);
    g_ndl = fmaf(g_dl, omk, g_ndl);
    adj.g_k = fmaf(g_dl, 1.0f - e.ndl, adj.g_k);
    g_ndv = fmaf(g_ds, 4.0f * e.ndl, g_ndv);
    g_ndl = fmaf(g_ds, 4.0f * t.ndv, g_ndl);
    adj.g_ndv += g_ndv;                                                      // N.V does not depend on the light
    // dots -> unit normal (clamps pass on the closed interval); c = N . h / |h|
    const float gl = in_unit(e.ndl_raw) ? g_ndl : 0.0f;
    const float gch = g_c * sqrt_hw(g.rhh);
    adj.g_n.x = fmaf(gl, g.L.x, fmaf(gch, g.h.x, adj.g_n.x));
    adj.g_n.y = fmaf(gl, g.L.y, fmaf(gch, g.h.y, adj.g_n.y));
    adj.g_n.z = fmaf(gl, g.L.z, fmaf(gch, g.h.z, adj.g_n.z));
}

//   LIGHT: PBR_LIGHT_*    WF: PBR_WORKFLOW_*    VEC: 4 | 1    TM: storage type of the maps AND of their gradients
//   (float | __half; arithmetic and the upstream gradient are fp32)
template <int LIGHT, int WF, int VEC, bool MULTI, typename TM = float>
__global__ __launch_bounds__(256) void cook_torrance_backward_kernel(const KArgs a, const BArgs b) {
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
    for (int j = 0; j < VEC; ++j) {
        // ---- forward: decoded colours and their derivatives
        float base[3], dbase[3], f0[3], df0[3], alin[3], kd_scale = 1.0f;
        const float m = WF != PBR_WORKFLOW_SPECULAR ? t.me[j] : 0.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            alin[c] = base[c] = a.albedo_srgb ? srgb_to_linear(t.al[c][j]) : t.al[c][j];
            dbase[c] = a.albedo_srgb ? srgb_to_linear_grad(t.al[c][j]) : 1.0f;
            if (WF == PBR_WORKFLOW_METALLIC) {
                f0[c] = fmaf(m, base[c] - kDielectricF0, kDielectricF0);
                df0[c] = 0.0f;
            } else if (WF == PBR_WORKFLOW_SPECULAR) {
                f0[c] = a.spec_srgb ? srgb_to_linear(t.sp[c][j]) : t.sp[c][j];
                df0[c] = a.spec_srgb ? srgb_to_linear_grad(t.sp[c][j]) : 1.0f;
            } else {   // CONVERTED: to_diffuse_specular_material (metallic.py:98-108), then the specular workflow
                const float sp = fmaf(alin[c], m, kDielectricF0 * (1.0f - m));
                base[c] = alin[c] * (1.0f - m);
                f0[c] = a.spec_srgb ? srgb_to_linear(sp) : sp;
                df0[c] = a.spec_srgb ? srgb_to_linear_grad(sp) : 1.0f;
            }
        }
        if (WF == PBR_WORKFLOW_METALLIC) kd_scale = 1.0f - m;
        const Vec3 nraw = {t.nm[0][j], t.nm[1][j], t.nm[2][j]};
        PixelTerms pt;
        pixel_terms(nraw, V, t.ro[j], base, f0, kd_scale, pt);
        const float ndv_raw = dotu(pt.n, V);
        const float xs = LIGHT == PBR_LIGHT_POINT ? linspace_at(a.x0, a.x1, a.xstep, a.W, p.x + j) : 0.0f;

        // ---- adjoint of the linear colour before per-light clamps
        float g_col[3];
        const int nl = MULTI ? a.n_lights : 1;
        if (MULTI) {                                 // pass 1: the summed colour decides the outer clamp / encode slope
            float sum[3] = {0.0f, 0.0f, 0.0f};
            for (int l = 0; l < nl; ++l) {
                const LightU &lu = a.lights[l];
                LightEval e;
                eval_light(pt, light_geom<LIGHT, float>(lu, V, xs, ys), lu.inten, e);
#pragma unroll
                for (int c = 0; c < 3; ++c) sum[c] += clamp01(e.u[c]);
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float tot = clamp01(sum[c]);
                const float slope = a.out_srgb ? linear_to_srgb_grad_unit(tot) : 1.0f;
                g_col[c] = in_unit(sum[c]) ? go[c][j] * slope : 0.0f;
            }
        }
        PixelAdjoint adj = {};
        for (int l = 0; l < nl; ++l) {
            const LightU &lu = a.lights[l];
            const LightGeom g = light_geom<LIGHT, float>(lu, V, xs, ys);
            LightEval e;
            eval_light(pt, g, lu.inten, e);
            if (!MULTI) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float col = clamp01(e.u[c]);
                    g_col[c] = go[c][j] * (a.out_srgb ? linear_to_srgb_grad_unit(col) : 1.0f);
                }
            }
            backprop_light(pt, g, lu.inten, e, g_col, adj);
        }
        // ---- light-independent tail
        // kb = kd_scale * base / pi ; kd_scale = 1 - m  (:169-174)
        float g_m = 0.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float g_base = adj.g_kb[c] * (kd_scale * kInvPi);
            if (WF == PBR_WORKFLOW_METALLIC) g_m = fmaf(adj.g_kb[c], -base[c] * kInvPi, g_m);
            if (WF == PBR_WORKFLOW_METALLIC) {                               // lerp(0.04, base, m)  (:107)
                g_base = fmaf(adj.g_f0[c], m, g_base);
                g_m = fmaf(adj.g_f0[c], base[c] - kDielectricF0, g_m);
            } else if (WF == PBR_WORKFLOW_SPECULAR) {
                gs[c][j] = adj.g_f0[c] * df0[c];
            } else {   // diffuse = a (1-m) ; specular = 0.04 (1-m) + a m  (kd_scale == 1 here, so g_m above is 0-weighted)
                const float g_sp = adj.g_f0[c] * df0[c], g_diff = adj.g_kb[c] * kInvPi;
                g_base = fmaf(g_diff, 1.0f - m, g_sp * m);
                g_m = fmaf(g_sp, alin[c] - kDielectricF0, fmaf(-g_diff, alin[c], g_m));
            }
            ga[c][j] = g_base * dbase[c];
        }
        gm[j] = g_m;
        const float r = t.ro[j];
        gr[j] = fmaf(adj.g_k, (r + 1.0f) * 0.25f, adj.g_a2 * (2.0f * r));    // k = (r+1)^2/8, a2 = r^2
        // N.V clamp, then F.normalize: g_n = (g - n (n.g)) / |n|
        Vec3 gnh = adj.g_n;
        if (in_unit(ndv_raw)) { gnh.x = fmaf(adj.g_ndv, V.x, gnh.x); gnh.y = fmaf(adj.g_ndv, V.y, gnh.y); gnh.z = fmaf(adj.g_ndv, V.z, gnh.z); }
        const float nn = dot(nraw, nraw);
        const float rn = rsq(fmaxf(nn, 1e-24f));
        const float radial = dot(pt.n, gnh);
        gn[0][j] = (gnh.x - pt.n.x * radial) * rn;
        gn[1][j] = (gnh.y - pt.n.y * radial) * rn;
        gn[2][j] = (gnh.z - pt.n.z * radial) * rn;
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
