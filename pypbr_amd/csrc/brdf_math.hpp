// brdf_math.hpp -- per-pixel Cook-Torrance arithmetic for gfx950 (CDNA4), fp32.
//
// What is computed follows the reference line by line (citations are to
// /root/reference/pypbr/models/cooktorrance.py unless another file is named).  HOW it
// is computed is chosen for the MI355X VALU budget (about 430
// lane-instructions per pixel at the 44 B/pixel HBM roofline):
//
//  * every division / sqrt / pow goes to the quarter-rate transcendental unit
//    (v_rcp_f32, v_rsq_f32, v_sqrt_f32, v_log_f32, v_exp_f32: 1 ulp each) -- no IEEE
//    division sequences, no ocml powf;
//  * the four denominators of D, G(V), G(L) and the specular term share ONE v_rcp_f32;
//  * x^5 is three multiplies; x^2.4 and x^(1/2.4) are exp2(c*log2(x)) on the restricted
//    domains the sRGB transfer functions reach;
//  * the GGX denominator NdotH^2 (a^2-1) + 1 is evaluated as a^2 + (1-a^2) sin^2(N,H) with
//    sin^2 = |n x h|^2 / (|n|^2 |h|^2).  The reference's form cancels catastrophically
//    for small roughness near the highlight: its own fp32 result is only ~2e-5..5e-5 from
//    the same code run in fp64 there (measured again in DESIGN.md).  The
//    cross-product form has no cancellation, so this kernel tracks the fp64 evaluation
//    of the reference to ~1e-6 and its distance to the reference's fp32 output is the
//    reference's own rounding error, not the sum of two.
#pragma once
#include <hip/hip_runtime.h>

namespace pbr {

constexpr float kPi = 3.14159265358979323846f;      // torch.pi / math.pi rounded to fp32
constexpr float kInvPi = 0.31830988618379067154f;
constexpr float kDielectricF0 = 0.04f;              // :107

__device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float sqrt_hw(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float log2_hw(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float exp2_hw(float x) { return __builtin_amdgcn_exp2f(x); }
// med3(x,0,1): one VALU op, folded into the producer's clamp modifier where possible.
__device__ __forceinline__ float clamp01(float x) { return __builtin_amdgcn_fmed3f(x, 0.0f, 1.0f); }

// utils/functions.py:31-47.  ((t+0.055)/1.055)^2.4 = exp2(2.4 log2(t+0.055) - 2.4 log2(1.055)).
__device__ __forceinline__ float srgb_to_linear(float x) {
    const float t = clamp01(x);
    const float lo = t * (1.0f / 12.92f);
    const float hi = exp2_hw(fmaf(2.4f, log2_hw(t + 0.055f), -0.18538320f /* 2.4*log2(1.055) = 0.185383197... */));
    return fminf(t <= 0.04045f ? lo : hi, 1.0f);
}

// utils/functions.py:50-66; `c` must already be in [0,1] (callers clamp).
__device__ __forceinline__ float linear_to_srgb_unit(float c) {
    const float lo = c * 12.92f;
    const float hi = fmaf(1.055f, exp2_hw(log2_hw(c) * (1.0f / 2.4f)), -0.055f);
    return clamp01(c <= 0.0031308f ? lo : hi);
}
__device__ __forceinline__ float linear_to_srgb(float x) { return linear_to_srgb_unit(clamp01(x)); }

struct Vec3 { float x, y, z; };
__device__ __forceinline__ float dot(const Vec3 &a, const Vec3 &b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }

// Light-dependent, material-independent terms of one pixel (:122-140, :155-159).
struct LightGeom {
    Vec3 L;        // light direction as the reference uses it (point: d/(dist+1e-7))
    Vec3 h;        // V + L, un-normalised
    float hh;      // |h|^2
    float att;     // 1/(dist^2+1e-7), 1 for directional
    float p5;      // (1 - clamp(Hv.V))^5
};

__device__ __forceinline__ float pow5(float x) { const float x2 = x * x; return x2 * x2 * x; }

// Point light (:129-140): surface point (xs, -ys, 0).
__device__ __forceinline__ LightGeom point_light_geom(const Vec3 &V, const Vec3 &Lpos, float xs, float ys) {
    LightGeom g;
    const Vec3 d = {Lpos.x - xs, Lpos.y + ys, Lpos.z};
    const float dist = sqrt_hw(dot(d, d));                         // torch.norm :138
    const float rinv = rcp(dist + 1e-7f);                          // :139
    g.L = {d.x * rinv, d.y * rinv, d.z * rinv};
    g.att = rcp(fmaf(dist, dist, 1e-7f));                          // :140 (distances**2, re-squared)
    g.h = {V.x + g.L.x, V.y + g.L.y, V.z + g.L.z};                 // :155
    g.hh = dot(g.h, g.h);
    const float rh = rsq(fmaxf(g.hh, 1e-24f));                     // F.normalize eps 1e-12 on the norm
    g.p5 = pow5(1.0f - clamp01(dot(g.h, V) * rh));                 // :156-158, :196
    return g;
}

// Light-independent terms of one pixel, computed once and reused by every light.
struct PixelTerms {
    Vec3 n;            // unit normal: stored normal * 1/max(|n|, 1e-12)   (F.normalize :154)
    float ndv;         // clamp(N.V)                  (:163)
    float a2;          // roughness^2                 (alpha = roughness, :213-214)
    float k;           // (r+1)^2/8                   (:232-233)
    float dv;          // NdotV (1-k) + k + 1e-7      (:234)
    float a2ndv;       // a2 * NdotV
    float base[3];     // linear albedo / diffuse colour, pre-multiplied by 1/pi (:174)
    float f0[3];       // reflectance at normal incidence
    float kd_scale;    // (1 - metallic) or 1          (:169-172)
};

__device__ __forceinline__ void pixel_terms(const Vec3 &n, const Vec3 &V, float rough, const float base[3],
                                            const float f0[3], float kd_scale, PixelTerms &t) {
    const float rn = rsq(fmaxf(dot(n, n), 1e-24f));
    t.n = {n.x * rn, n.y * rn, n.z * rn};
    t.ndv = clamp01(dot(t.n, V));
    t.a2 = rough * rough;
    const float r1 = rough + 1.0f;
    t.k = r1 * r1 * 0.125f;
    t.dv = fmaf(t.ndv, 1.0f - t.k, t.k) + 1e-7f;
    t.a2ndv = t.a2 * t.ndv;
#pragma unroll
    for (int c = 0; c < 3; ++c) { t.base[c] = base[c] * kInvPi; t.f0[c] = f0[c]; }
    t.kd_scale = kd_scale;
}

// One light's linear RGB contribution, clamped to [0,1] (:160-177).
__device__ __forceinline__ void shade_light(const PixelTerms &t, const LightGeom &g, const float inten[3], float out[3]) {
    const float ndl = clamp01(dot(t.n, g.L));                      // :164
    // GGX (:213-217), cancellation-free: den = a2 + (1-a2) sin^2(N,H) when N.H > 0, else 1;
    // sin^2 = |N x h|^2 / |h|^2 with N unit.
    const float nh = dot(t.n, g.h);
    const Vec3 c = {fmaf(t.n.y, g.h.z, -(t.n.z * g.h.y)), fmaf(t.n.z, g.h.x, -(t.n.x * g.h.z)),
                    fmaf(t.n.x, g.h.y, -(t.n.y * g.h.x))};
    const float s2 = fminf(dot(c, c) * rcp(fmaxf(g.hh, 1e-36f)), 1.0f);
    const float den = nh > 0.0f ? fmaf(s2, 1.0f - t.a2, t.a2) : 1.0f;
    // D * G / (4 NdotV NdotL + 1e-7) with one reciprocal (:217, :232-235, :165-166).
    const float dl = fmaf(ndl, 1.0f - t.k, t.k) + 1e-7f;
    const float dD = fmaf(kPi, den * den, 1e-7f);
    const float ds = fmaf(4.0f * t.ndv, ndl, 1e-7f);
    const float dg = t.a2ndv * ndl * rcp((dD * t.dv) * (dl * ds));
    const float rad = ndl * g.att;                                 // :175
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        const float F = fmaf(1.0f - t.f0[ch], g.p5, t.f0[ch]);     // :196
        const float kd = (1.0f - F) * t.kd_scale;                  // :169-172
        out[ch] = clamp01(fmaf(F, dg, kd * t.base[ch]) * (inten[ch] * rad));   // :174-177
    }
}

}  // namespace pbr
