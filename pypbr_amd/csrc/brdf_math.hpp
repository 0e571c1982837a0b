// brdf_math.hpp -- per-pixel Cook-Torrance arithmetic for gfx950 (CDNA4), fp32.
//
// What is computed follows the reference line by line (citations are to
// /root/reference/pypbr/models/cooktorrance.py unless another file is named).  HOW it
// is computed is chosen for the MI355X VALU: rocprofv3 SQ counters on this kernel show a plain
// fp32 VALU wave-instruction holding its SIMD for ~4.25 cycles (SQ_ACTIVE_INST_VALU x 4 /
// SQ_INSTS_VALU), i.e. 16 lanes per clock; only the PACKED forms (v_pk_fma_f32, v_pk_mul_f32,
// v_pk_add_f32: two fp32 values per lane per instruction, same 4 cycles) reach the chip's fp32 peak.
// At ~160 instructions per pixel the one-light kernel has its VALUs 62 % busy next to a
// 77 %-busy HBM, and the 16-light configuration is VALU-bound outright.  Hence:
//
//  * every function here is a template over the real type R: float (one pixel) or f32x2 (TWO pixels
//    of the lane at once).  The multi-light and fp16 kernels use f32x2, so that nearly all adds /
//    multiplies / fmas are packed instructions; transcendental and min/max/select instructions have
//    no packed form and are issued per component;
//  * every division / sqrt / pow goes to the transcendental unit (v_rcp_f32, v_rsq_f32,
//    v_log_f32, v_exp_f32: 1 ulp each) -- no IEEE division sequences, no ocml powf; one light
//    evaluation costs 4 of them (point light) or 1 (directional);
//  * the four denominators of D, G(V), G(L) and the specular term share ONE v_rcp_f32;
//  * x^5 is three multiplies; x^2.4 and x^(1/2.4) are exp2(c*log2(x)) on the restricted
//    domains the sRGB transfer functions reach;
//  * F dg + (1-F) kd base/pi is evaluated as kb + F (dg - kb) with kb = kd_scale base/pi hoisted
//    out of the light loop;
//  * N.h is N.L + N.V (h = L + V) and N.L is (N.d) rinv: the normalised light vector is never formed;
//  * the GGX denominator NdotH^2 (a^2-1) + 1 is evaluated as a^2 + (1-a^2) sin^2(N,H) with
//    sin^2 = |h - (N.h) N|^2 / |h|^2, N the unit normal.  The reference's form cancels
//    catastrophically for small roughness near the highlight: its own fp32 result is only
//    ~2e-5..5e-5 from the same code run in fp64 there.  The
//    projection form has no cancellation, so this kernel tracks the fp64 evaluation of
//    the reference to ~1e-6 and its distance to the reference's fp32 output is the
//    reference's own rounding error, not the sum of two.
#pragma once
#include <hip/hip_runtime.h>

namespace pbr {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

constexpr float kPi = 3.14159265358979323846f;      // torch.pi / math.pi rounded to fp32
constexpr float kInvPi = 0.31830988618379067154f;
constexpr float kDielectricF0 = 0.04f;              // :107

// ---- scalar primitives
__device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float sqrt_hw(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float log2_hw(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float exp2_hw(float x) { return __builtin_amdgcn_exp2f(x); }
// med3(x,0,1): one VALU op, folded into the producer's clamp modifier where possible.
__device__ __forceinline__ float clamp01(float x) { return __builtin_amdgcn_fmed3f(x, 0.0f, 1.0f); }
__device__ __forceinline__ float fma_(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ float max_(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ float min_(float a, float b) { return fminf(a, b); }
__device__ __forceinline__ float select_(bool m, float a, float b) { return m ? a : b; }
__device__ __forceinline__ bool gt_(float a, float b) { return a > b; }
__device__ __forceinline__ bool le_(float a, float b) { return a <= b; }
__device__ __forceinline__ bool ge_(float a, float b) { return a >= b; }
__device__ __forceinline__ bool eq_(float a, float b) { return a == b; }
__device__ __forceinline__ bool and_(bool a, bool b) { return a && b; }

// ---- two pixels per lane: +, -, * on f32x2 compile to v_pk_add/mul_f32, fma to v_pk_fma_f32
__device__ __forceinline__ f32x2 rcp(f32x2 x) { return f32x2{rcp(x.x), rcp(x.y)}; }
__device__ __forceinline__ f32x2 rsq(f32x2 x) { return f32x2{rsq(x.x), rsq(x.y)}; }
__device__ __forceinline__ f32x2 sqrt_hw(f32x2 x) { return f32x2{sqrt_hw(x.x), sqrt_hw(x.y)}; }
__device__ __forceinline__ f32x2 log2_hw(f32x2 x) { return f32x2{log2_hw(x.x), log2_hw(x.y)}; }
__device__ __forceinline__ f32x2 exp2_hw(f32x2 x) { return f32x2{exp2_hw(x.x), exp2_hw(x.y)}; }
// Packed fp32 has no min/max/med3, but VOP3P carries the clamp output modifier ([0,1] saturation): one packed
// instruction clamps two pixels, and where the value to clamp is a product or an fma the clamp rides on it for free.
// The compiler does not form these (it sees two scalar med3), hence the inline assembly (verified on MI355X against
// fminf(fmaxf(x,0),1) over sign, range and both halves).
__device__ __forceinline__ f32x2 clamp01(f32x2 x) {
    f32x2 r;
    asm("v_pk_mul_f32 %0, %1, 1.0 op_sel_hi:[1,0] clamp" : "=v"(r) : "v"(x));
    return r;
}
__device__ __forceinline__ f32x2 mul_sat(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x2 fma_sat(f32x2 a, f32x2 b, f32x2 c) {
    f32x2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// gfx940+ "trans forwarding" hazard: a non-transcendental VALU instruction that reads a VGPR written by the transcendental instruction
// IMMEDIATELY before it needs one wait state.  The compiler inserts it for its own instructions, NOT in front of inline assembly -- the
// packed v_pk_fma_f32 right behind a v_exp_f32 then reads the OLD register in lanes 0-3 of every 8 (found in round 3: a loss kernel whose
// schedule put them back to back; tools/check_isa.py now scans every kernel of the build for the pattern).  Where an operand IS the direct result
// of v_exp / v_log / v_rcp / v_rsq, use these forms: the wait state travels inside the assembly.
__device__ __forceinline__ f32x2 mul_sat_after_trans(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("s_nop 0\n\tv_pk_mul_f32 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x2 fma_sat_after_trans(f32x2 a, f32x2 b, f32x2 c) {
    f32x2 r;
    asm("s_nop 0\n\tv_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float mul_sat(float a, float b) { return clamp01(a * b); }          // v_mul_f32 ... clamp
__device__ __forceinline__ float mul_sat_after_trans(float a, float b) { return clamp01(a * b); }
__device__ __forceinline__ float fma_sat_after_trans(float a, float b, float c) { return clamp01(fmaf(a, b, c)); }
__device__ __forceinline__ float fma_sat(float a, float b, float c) { return clamp01(fmaf(a, b, c)); }
__device__ __forceinline__ f32x2 fma_(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 max_(f32x2 a, f32x2 b) { return f32x2{fmaxf(a.x, b.x), fmaxf(a.y, b.y)}; }
__device__ __forceinline__ f32x2 min_(f32x2 a, f32x2 b) { return f32x2{fminf(a.x, b.x), fminf(a.y, b.y)}; }
__device__ __forceinline__ f32x2 select_(i32x2 m, f32x2 a, f32x2 b) { return m ? a : b; }
__device__ __forceinline__ i32x2 gt_(f32x2 a, f32x2 b) { return a > b; }
__device__ __forceinline__ i32x2 le_(f32x2 a, f32x2 b) { return a <= b; }
__device__ __forceinline__ i32x2 ge_(f32x2 a, f32x2 b) { return a >= b; }
__device__ __forceinline__ i32x2 eq_(f32x2 a, f32x2 b) { return a == b; }
__device__ __forceinline__ i32x2 and_(i32x2 a, i32x2 b) { return a & b; }

template <class R> __device__ __forceinline__ R splat(float v);
template <> __device__ __forceinline__ float splat<float>(float v) { return v; }
template <> __device__ __forceinline__ f32x2 splat<f32x2>(float v) { return f32x2{v, v}; }

// utils/functions.py:31-47.  ((t+0.055)/1.055)^2.4 = exp2(2.4 log2(t+0.055) - 2.4 log2(1.055)).
template <class R> __device__ __forceinline__ R srgb_to_linear(R x) {
    const R t = clamp01(x);
    const R lo = t * (1.0f / 12.92f);
    const R hi = exp2_hw(fma_(splat<R>(2.4f), log2_hw(t + 0.055f), splat<R>(-0.18538320f) /* 2.4*log2(1.055) */));
    return select_(le_(t, splat<R>(0.04045f)), lo, hi);      // hi(1) = 1 +- 1 ulp, like the reference's pow
}

// utils/functions.py:50-66; `c` must already be in [0,1] (callers clamp).
template <class R> __device__ __forceinline__ R linear_to_srgb_unit(R c) {
    const R lo = c * 12.92f;
    const R hi = fma_sat_after_trans(splat<R>(1.055f), exp2_hw(log2_hw(c) * (1.0f / 2.4f)), splat<R>(-0.055f));   // clamp = output modifier
    return select_(le_(c, splat<R>(0.0031308f)), lo, hi);                            // lo <= 0.0405 needs none
}
template <class R> __device__ __forceinline__ R linear_to_srgb(R x) { return linear_to_srgb_unit(clamp01(x)); }

template <class R> struct Vec3T { R x, y, z; };
using Vec3 = Vec3T<float>;
template <class R> __device__ __forceinline__ R dot(const Vec3T<R> &a, const Vec3T<R> &b) {
    return fma_(a.z, b.z, fma_(a.y, b.y, a.x * b.x));
}
// a.b + c with c seeding the fma chain: the "+ tiny" that keeps rsq finite on a zero vector costs nothing
template <class R> __device__ __forceinline__ R dot_plus(const Vec3T<R> &a, const Vec3T<R> &b, float c) {
    return fma_(a.z, b.z, fma_(a.y, b.y, fma_(a.x, b.x, splat<R>(c))));
}
template <class R> __device__ __forceinline__ R dotu(const Vec3T<R> &a, const Vec3 &b) {   // b wave-uniform
    return fma_(a.z, splat<R>(b.z), fma_(a.y, splat<R>(b.y), a.x * b.x));
}

// Light-dependent, material-independent terms of one pixel (:122-140, :155-159).
template <class R> struct LightGeomT {
    Vec3T<R> d;    // point: light position - surface point, un-normalised; directional: L
    R rinv;        // point: 1/(dist + 1e-7); directional: 1.   N.L = (N.d) rinv costs 4 ops, never forming L
    R rdist;       // point: 1/dist (the v_rsq itself); only the light-parameter gradient reads it
    Vec3T<R> h;    // V + L, un-normalised (L = d * rinv, the light direction as the reference uses it, is never kept: N.L = (N.d) rinv)
    R rhh;         // 1/|h|^2  (|h|^2 clamped at 1e-24: F.normalize's 1e-12 on the norm)
    R rh;          // 1/|h|: the v_rsq that rhh is the square of; only the backward kernels read it (round 6: they took sqrt(rhh) twice per light)
    R att;         // 1/(dist^2+1e-7), 1 for directional
    R p5;          // (1 - clamp(Hv.V))^5
    R om5;         // 1 - p5:  F = f0 + (1 - f0) p5 = f0 om5 + p5, one fma per channel
};
using LightGeom = LightGeomT<float>;

template <class R> __device__ __forceinline__ R pow5(R x) { const R x2 = x * x; return x2 * x2 * x; }

// Point light (:129-140): surface point (xs, -ys, 0); ys is the lane's row, shared by its pixels.
//   dist = sqrt(dd) and 1/(dist + 1e-7) come from ONE v_rsq: r = rsq(dd), dist = dd r,
//   1/(dist + 1e-7) = r / (1 + 1e-7 r) = r (1 - 1e-7 r) + O((1e-7 r)^2) -- exact to 1e-8 relative for any
//   light further than 1e-3 from the surface point.  dist^2 carries + 1e-12 and |h|^2 + 1e-24, seeded into their
//   fma chains (below one ulp for any distance above 4e-3 / any |h| above 4e-9): r stays <= 1e6, so a light sitting
//   exactly on a pixel, or L = -V, give L = 0 / a zero half vector like the reference's F.normalize instead of
//   NaN.  (They replace max(., tiny): packed fp32 has no max, and the seed is free.)
template <class R>
__device__ __forceinline__ LightGeomT<R> point_light_geom(const Vec3 &V, const Vec3 &Lpos, R xs, float ys) {
    LightGeomT<R> g;
    const Vec3T<R> d = {splat<R>(Lpos.x) - xs, splat<R>(Lpos.y + ys), splat<R>(Lpos.z)};
    const R dd = dot_plus(d, d, 1e-12f);                            // dist^2, torch.norm :138
    const R r = rsq(dd);
    const R rinv = r * fma_(splat<R>(-1e-7f), r, splat<R>(1.0f));   // 1/(dist + 1e-7)  :139
    g.d = d; g.rinv = rinv; g.rdist = r;
    g.att = rcp(dd + 1e-7f);                                        // :140
    g.h = {fma_(d.x, rinv, splat<R>(V.x)), fma_(d.y, rinv, splat<R>(V.y)), fma_(d.z, rinv, splat<R>(V.z))};   // :155
    const R rh = rsq(dot_plus(g.h, g.h, 1e-24f));
    g.rhh = rh * rh; g.rh = rh;
    g.p5 = pow5(splat<R>(1.0f) - mul_sat_after_trans(dotu(g.h, V), rh));         // :156-158, :196  (rh: the v_rsq's own result)
    g.om5 = splat<R>(1.0f) - g.p5;
    return g;
}

// Light-independent terms of one pixel, computed once and reused by every light.
template <class R> struct PixelTermsT {
    Vec3T<R> n;        // unit normal: stored normal * 1/max(|n|, 1e-12)   (F.normalize :154)
    R ndv_raw;         // N.V before the clamp: N.h = N.L + N.V, one add per light
    R ndv;             // clamp(N.V)                  (:163)
    R ndv4;            // 4 clamp(N.V): the specular denominator 4 NdotV NdotL + 1e-7 is one fma per light
    R a2;              // roughness^2                 (alpha = roughness, :213-214)
    R oma2;            // 1 - a2: the GGX denominator a2 + (1 - a2) sin^2 is one fma per light
    R k;               // (r+1)^2/8                   (:232-233)
    R omk, kk;         // 1 - k, k + 1e-7: both G denominators are one fma, ndx omk + kk
    R dv;              // NdotV (1-k) + k + 1e-7      (:234)
    R a2ndv;           // a2 * NdotV
    R a2ndv_pi;        // a2 * NdotV / pi: the pi of D's denominator, paid per pixel instead of per light
    R kb[3];           // kd_scale * base / pi: the diffuse term is (1 - F) kb   (:169-174)
    R f0[3];           // reflectance at normal incidence
};
using PixelTerms = PixelTermsT<float>;

// base: linear albedo / diffuse colour; kd_scale: (1 - metallic) or 1.
template <class R>
__device__ __forceinline__ void pixel_terms(const Vec3T<R> &n, const Vec3 &V, R rough, const R base[3], const R f0[3],
                                            R kd_scale, PixelTermsT<R> &t) {
    const R rn = rsq(dot_plus(n, n, 1e-24f));
    t.n = {n.x * rn, n.y * rn, n.z * rn};
    t.ndv_raw = dotu(t.n, V);
    t.ndv = clamp01(t.ndv_raw);
    t.ndv4 = t.ndv * 4.0f;
    t.a2 = rough * rough;
    t.oma2 = splat<R>(1.0f) - t.a2;
    const R r1 = rough + 1.0f;
    t.k = r1 * r1 * 0.125f;
    t.omk = splat<R>(1.0f) - t.k;
    t.kk = t.k + 1e-7f;
    t.dv = fma_(t.ndv, t.omk, t.kk);
    t.a2ndv = t.a2 * t.ndv;
    t.a2ndv_pi = (t.a2 * kInvPi) * t.ndv;
    const R s = kd_scale * kInvPi;
#pragma unroll
    for (int c = 0; c < 3; ++c) { t.kb[c] = base[c] * s; t.f0[c] = f0[c]; }
}

// GGX denominator (:213-217), cancellation-free: a2 + (1-a2) sin^2(N,H) when N.H > 0, else 1, with
// sin^2 = |h - (N.h) N|^2 / |h|^2 (N unit).  Rounding errors of N.h and of |N| move the projection ALONG N, i.e.
// orthogonally to it, so they enter sin^2 only to second order; 9 ops against 12 for the cross product.
// `nh` = N.h, which the caller has as N.L + N.V.
template <class R, class M>
__device__ __forceinline__ R ggx_den(const PixelTermsT<R> &t, const LightGeomT<R> &g, R nh, R &s2, M &nh_pos) {
    const Vec3T<R> p = {fma_(-nh, t.n.x, g.h.x), fma_(-nh, t.n.y, g.h.y), fma_(-nh, t.n.z, g.h.z)};
    s2 = min_(dot(p, p) * g.rhh, splat<R>(1.0f));
    nh_pos = gt_(nh, splat<R>(0.0f));
    return select_(nh_pos, fma_(s2, splat<R>(1.0f) - t.a2, t.a2), splat<R>(1.0f));
}

// The forward kernels' form: no N.H > 0 select and no clamp of sin^2 at 1.  N.h = N.L + N.V <= 0 means that N.L or
// N.V is <= 0, so its clamp is 0 and the specular term a2 NdotV NdotL / (...) is exactly 0 whatever D is; and
// sin^2 <= 1 + 1e-7 keeps den finite and positive.  Two compare/select pairs and a min per light evaluation less.
template <class R>
__device__ __forceinline__ R ggx_den_fwd(const PixelTermsT<R> &t, const LightGeomT<R> &g, R nh) {
    const Vec3T<R> p = {fma_(-nh, t.n.x, g.h.x), fma_(-nh, t.n.y, g.h.y), fma_(-nh, t.n.z, g.h.z)};
    return fma_(dot(p, p) * g.rhh, t.oma2, t.a2);
}

template <class R> struct MaskOf { using type = bool; };
template <> struct MaskOf<f32x2> { using type = i32x2; };

// One light's linear RGB contribution, clamped to [0,1] (:160-177).  `inten` is wave-uniform.  GREY: the light's three
// intensities are equal (decided on the host for the whole launch), so radiance * intensity is computed once.
template <bool GREY = false, class R>
__device__ __forceinline__ void shade_light(const PixelTermsT<R> &t, const LightGeomT<R> &g, const float inten[3], R out[3]) {
    const R nd = dot(t.n, g.d);
    const R ndl = mul_sat(nd, g.rinv);                             // clamp(N.L) :164
    const R den = ggx_den_fwd(t, g, fma_(nd, g.rinv, t.ndv_raw));  // N.h = N.L + N.V
    // D * G / (4 NdotV NdotL + 1e-7) with one reciprocal (:217, :232-235, :165-166); D's pi sits in a2ndv_pi.
    const R dl = fma_(ndl, t.omk, t.kk);
    const R dD = fma_(den, den, splat<R>(1e-7f * kInvPi));
    const R ds = fma_(t.ndv4, ndl, splat<R>(1e-7f));
    const R dg = t.a2ndv_pi * ndl * rcp((dD * t.dv) * (dl * ds));
    const R rad = ndl * g.att;                                     // :175
    const R w0 = rad * inten[0];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        const R F = fma_(t.f0[ch], g.om5, g.p5);                   // :196
        // F dg + (1 - F) kb  ==  kb + F (dg - kb)                    (:166, :169-174)
        out[ch] = mul_sat(fma_(F, dg - t.kb[ch], t.kb[ch]), (GREY || ch == 0) ? w0 : rad * inten[ch]);      // :175-177
    }
}

}  // namespace pbr
