// ct_kernel.hpp -- the fused Cook-Torrance kernels (device side) for gfx950.
//
// Work decomposition.  A *tile* is what one workgroup (T = 64..256 lanes, no LDS, no barrier)
// evaluates at a time: bx lanes along x (each lane VEC = 4 consecutive pixels -> 16-byte
// accesses) by T/bx rows, with bx = min(T, next_pow2(W/4)).  For 4K maps every wave touches one
// 1 KiB-contiguous run of each of the 8 input and 3 output planes.
//
// Schedule: one tile per one-wave workgroup, 1-D grid; the workgroup -> tile map is either the identity or
// XCD-aware runs (tile_of_workgroup below).  A persistent grid-stride variant with register double buffering
// (next tile's loads issued before the current tile's arithmetic) was built and measured 10-20 % SLOWER
// ("Schedule experiments"): ~11 independent one-shot waves per CU already saturate the memory
// pipe, and the chip-wide dispatcher is the cheaper software pipeline.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <cstdint>
#include <type_traits>

#include "../../include/pbr_hip.h"
#include "brdf_math.hpp"

namespace pbr {

// Per-light, wave-uniform block (host-folded, all fp32).
struct LightU {
    float l[3];       // directional: normalised L (:126); point: position (:129)
    float h[3];       // directional: V + L
    float rhh;        // directional: 1/|V+L|^2
    float p5;         // directional: (1 - clamp(Hv.V))^5
    float inten[3];   // :96
};

// pbr_render_desc.device_params: the same wave-uniform values, prepared on the device (cook_torrance.hip: prepare_device_params_kernel)
// from parameter tensors that live there.  `raw_*`: the un-normalised inputs, for the chain rule through F.normalize (param_grad_finish_kernel).
struct DevParams {
    float V[3];
    int32_t grey;                      // every light's three intensities are equal
    LightU lights[PBR_MAX_LIGHTS];
    float raw_view[3];
    float raw_lights[PBR_MAX_LIGHTS][3];
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
    int32_t xcd_log2;        // workgroup -> tile remap: each XCD takes runs of (1 << xcd_log2) consecutive tiles (0 = identity)
    int32_t xcd_tiles;       // tiles covered by the remap: n_tiles rounded down to a multiple of 8 << xcd_log2
    int32_t fold_log2, fold_reps;   // tiled maps: rows visited in bands of (1 << fold_log2) source rows, all fold_reps vertical repeats of a band back to back (0 = off)
    FastDiv div_reps;        // band visit / fold_reps
    int32_t xpose;           // 8-pixel lanes, fp32 result: exchange the lanes' 16-byte pieces through LDS before storing
    int32_t sbase;           // every tile lies inside one material and every plane is < 4 GiB: scalar plane addresses (plane_at)
    FastDiv div_h;           // row / H
    FastDiv div_tx;          // tile / tiles_x
    int32_t tiled;           // maps are (map_h, map_w) and repeat over the (H_total, W) output (MaterialBase.tile)
    int32_t map_h, map_w;
    FastDiv div_mh, div_mw;  // (y_offset + y) / map_h, x / map_w
    int32_t y_offset, H_total;
    float x0, x1, xstep;     // torch.linspace(-s/2, s/2, W)   :132
    float y0, y1, ystep;     // torch.linspace(-s/2, s/2, H_total)   :133
    float V[3];              // F.normalize(view_dir)   :95
    int32_t n_lights;
    int32_t albedo_srgb, spec_srgb, out_srgb, has_normal;
    int32_t grey_lights;     // every light's three intensities are equal: radiance * intensity once per light, not per channel
    LightU lights[PBR_MAX_LIGHTS];
    int32_t rep_y, rep_x;    // cook_torrance_repeat_kernel: the grid walks the SOURCE maps (H x W texels), every lane evaluates its texels at all rep_y * rep_x positions of the output
    int32_t win_y0;          // repeat kernels, a row band THINNER than a period of the map's rows (round 6): the walk covers only the H source rows the band
                             // touches, row i of it is source row (win_y0 + i) mod map_h; map_h is the period there (H == map_h: the whole map, win_y0 = 0)
    int32_t out_W, out_Ht;   // ... whose rows are out_W pixels wide and whose point-light grid spans out_Ht x out_W; `out` holds the rows
                             // [y_offset, y_offset + H_total) of it (the repeat kernel's H_total: the band's rows; it has no other use there)
    uint64_t dev;            // pbr_render_desc.device_params (address of a DevParams block, 0 = none): when set, V and the light blocks are read from it (view_of / light_of)
};

// View vector and light block of a launch: kernel arguments, or the device block when the parameters live in device memory.
// Wave-uniform either way (scalar loads); by value, so that neither the kernel-argument struct nor the block has its address taken.
// The block is read through the CONSTANT address space: written by an earlier kernel of the stream, never by this one, so its loads are scalar
// (s_load) like the kernel arguments' -- through a plain pointer the compiler issues per-lane vector loads (the ISA assertions of the streamed
// kernels caught exactly that).
typedef const __attribute__((address_space(4))) DevParams *DevParamsPtr;
__device__ __forceinline__ DevParamsPtr dev_params(uint64_t address) { return (DevParamsPtr)address; }
__device__ __forceinline__ Vec3 view_of(const KArgs &a) {
    if (a.dev) { const DevParamsPtr d = dev_params(a.dev); return Vec3{d->V[0], d->V[1], d->V[2]}; }
    return Vec3{a.V[0], a.V[1], a.V[2]};
}
__device__ __forceinline__ LightU light_of(const KArgs &a, int l) {
    if (a.dev) {
        const DevParamsPtr d = dev_params(a.dev);
        LightU u;
#pragma unroll
        for (int c = 0; c < 3; ++c) { u.l[c] = d->lights[l].l[c]; u.h[c] = d->lights[l].h[c]; u.inten[c] = d->lights[l].inten[c]; }
        u.rhh = d->lights[l].rhh; u.p5 = d->lights[l].p5;
        return u;
    }
    return a.lights[l];
}
__device__ __forceinline__ bool grey_lights_of(const KArgs &a) { return a.dev ? dev_params(a.dev)->grey != 0 : a.grey_lights != 0; }

// ------------------------------------------------------------------ typed vector I/O
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
// The same vectors with ELEMENT alignment: what a lane's 4 (2) pixels have when the row pitch or a view's origin is not
// a multiple of 16 (8) bytes.  The instruction is the same global_load_dwordx4 / dwordx2 (gfx950 runs in unaligned
// access mode); a misaligned wave touches one more 128-byte line per plane.  Aligned data loses nothing.
typedef f32x4 f32x4_e __attribute__((aligned(4)));
typedef f16x4 f16x4_e __attribute__((aligned(2)));

template <typename T, int VEC> struct Ld;
template <> struct Ld<float, 4> {
    template <bool NT> static __device__ __forceinline__ void load(const void *p, int64_t i, float v[4]) {
        const f32x4_e *q = reinterpret_cast<const f32x4_e *>(static_cast<const float *>(p) + i);
        const f32x4 t = NT ? __builtin_nontemporal_load(q) : *q;
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    template <bool NT> static __device__ __forceinline__ void store(void *p, int64_t i, const float v[4]) {
        const f32x4 t = {v[0], v[1], v[2], v[3]};
        f32x4_e *q = reinterpret_cast<f32x4_e *>(static_cast<float *>(p) + i);
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
        const f16x4_e *q = reinterpret_cast<const f16x4_e *>(static_cast<const _Float16 *>(p) + i);
        const f16x4 t = NT ? __builtin_nontemporal_load(q) : *q;
        v[0] = (float)t.x; v[1] = (float)t.y; v[2] = (float)t.z; v[3] = (float)t.w;
    }
    template <bool NT> static __device__ __forceinline__ void store(void *p, int64_t i, const float v[4]) {
        const f16x4 t = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
        f16x4_e *q = reinterpret_cast<f16x4_e *>(static_cast<_Float16 *>(p) + i);
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

// 2 pixels per lane (one packed pair): the register-light form of the backward kernel for fp16 maps
typedef float f32x2v __attribute__((ext_vector_type(2), aligned(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2), aligned(2)));
template <> struct Ld<float, 2> {
    template <bool NT> static __device__ __forceinline__ void load(const void *p, int64_t i, float v[2]) {
        const f32x2v *q = reinterpret_cast<const f32x2v *>(static_cast<const float *>(p) + i);
        const f32x2v t = NT ? __builtin_nontemporal_load(q) : *q;
        v[0] = t.x; v[1] = t.y;
    }
    template <bool NT> static __device__ __forceinline__ void store(void *p, int64_t i, const float v[2]) {
        const f32x2v t = {v[0], v[1]};
        f32x2v *q = reinterpret_cast<f32x2v *>(static_cast<float *>(p) + i);
        if (NT) __builtin_nontemporal_store(t, q); else *q = t;
    }
};
template <> struct Ld<__half, 2> {
    template <bool NT> static __device__ __forceinline__ void load(const void *p, int64_t i, float v[2]) {
        const f16x2 *q = reinterpret_cast<const f16x2 *>(static_cast<const _Float16 *>(p) + i);
        const f16x2 t = NT ? __builtin_nontemporal_load(q) : *q;
        v[0] = (float)t.x; v[1] = (float)t.y;
    }
    template <bool NT> static __device__ __forceinline__ void store(void *p, int64_t i, const float v[2]) {
        const f16x2 t = {(_Float16)v[0], (_Float16)v[1]};
        f16x2 *q = reinterpret_cast<f16x2 *>(static_cast<_Float16 *>(p) + i);
        if (NT) __builtin_nontemporal_store(t, q); else *q = t;
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

// Scalar plane addressing (KArgs::sbase).  The general form of an access is map + b * batch_stride + c * channel_stride +
// lane offset with a per-lane material index b: a 64-bit multiply-add and a 64-bit add per plane in vector registers
// (19 planes in the backward kernel: 9 % of its VALU instructions and two address registers per plane).  When the
// whole workgroup works on one material, everything but the lane offset is uniform: the plane's address is formed
// with scalar instructions and the access becomes `global_load ... v_offset, s[base:base+1]` -- no vector address
// arithmetic at all, and ONE 32-bit offset register shared by all planes of the same element size.  The empty asm keeps
// the compiler from folding the lane offset back into a 64-bit vector address.
typedef __attribute__((address_space(1))) char *global_ptr;
template <typename T>
__device__ __forceinline__ void *plane_at(const void *plane, int64_t uniform_elems, uint32_t lane_elems) {
    uint64_t base = reinterpret_cast<uint64_t>(plane) + (uint64_t)uniform_elems * sizeof(T);
    asm("" : "+s"(base));
    return (void *)(reinterpret_cast<global_ptr>(base) + lane_elems * (uint32_t)sizeof(T));
}

// torch.linspace two-ended evaluation (what ATen's device kernel computes), branch-free:
// i < n/2 ? a + step*i : b - step*(n-1-i).
__device__ __forceinline__ float linspace_at(float a, float b, float step, int n, int i) {
    const bool lo = i < (n >> 1);
    return fmaf(lo ? step : -step, (float)(lo ? i : n - 1 - i), lo ? a : b);
}

// Workgroups are dealt to the 8 XCDs round-robin (workgroup i runs on XCD i % 8).  With the identity order every
// XCD therefore walks each plane in 1 KiB steps 8 KiB apart; the remap hands XCD x the tiles
// [x << c, (x + 1) << c) of every block of 8 << c tiles, i.e. (1 KiB << c)-contiguous runs per XCD, while the chip-wide
// front stays one compact window.  Scalar arithmetic only.
__device__ __forceinline__ uint32_t tile_of_workgroup(const KArgs &a, uint32_t wg) {
    uint32_t s = wg;
    if (a.xcd_log2 != 0 && wg < (uint32_t)a.xcd_tiles) {
        const uint32_t c = (uint32_t)a.xcd_log2, xcd = wg & 7u, slot = wg >> 3;
        s = ((slot >> c) << (c + 3)) + (xcd << c) + (slot & ((1u << c) - 1u));
    }
    if (a.fold_log2 > 0) {
        // tile(n) (MaterialBase.tile, base.py:524-537): output rows y and y + map_h read the same texels.  In row order the
        // second visit comes map_h rows -- tens of MB of traffic -- later and misses every L2; here the launch walks bands of
        // 1 << fold_log2 source rows and visits all vertical repeats of a band back to back.  A band is a whole number of
        // 8-XCD periods of tiles (fill_args), so the XCD that fetched a texel is the one that reads it again.
        const uint32_t f = (uint32_t)a.fold_log2, row = a.div_tx.div(s), tx = s - row * (uint32_t)a.tiles_x;
        const uint32_t visit = row >> f, band = a.div_reps.div(visit), rep = visit - band * (uint32_t)a.fold_reps;
        s = (rep * (uint32_t)a.map_h + (band << f) + (row & ((1u << f) - 1u))) * (uint32_t)a.tiles_x + tx;
    }
    return s;
}

constexpr int kXposeLdsPerWave = 3 * 144 * 16;       // shade_and_store's piece exchange (8-pixel lanes, fp32 result)

// ------------------------------------------------------------------ one lane's share of a tile
struct LanePos {
    int b, y, x;          // material, row inside the band, first pixel column
    int64_t pix;          // y * W + x: where the result goes
    int64_t src;          // where the texels come from: pix, or the wrapped position inside the (map_h, map_w) maps
    bool valid;
    bool sb;              // KArgs::sbase: b == b0 for every lane of the workgroup, pix and src fit 30 bits
    int b0;               // material of the tile's first row (scalar)
    int dup;              // ragged rows: this many of the lane's first pixels are also the previous lane's last ones
};

// CLAMP: lanes outside the map get the nearest position inside it (valid = false): they may load, must not store.
template <int VEC, bool CLAMP = false>
__device__ __forceinline__ LanePos lane_pos(const KArgs &a, int tile_x, int tile_y) {
    const int tid = threadIdx.x;
    int xv = (tile_x << a.bx_log2) + (tid & ((1 << a.bx_log2) - 1));
    int row = (tile_y << (a.bt_log2 - a.bx_log2)) + (tid >> a.bx_log2);  // b * H + y
    LanePos p;
    p.valid = xv < a.wv && row < a.rows;
    if (CLAMP) { xv = xv < a.wv ? xv : a.wv - 1; row = row < a.rows ? row : a.rows - 1; }
    p.sb = a.sbase != 0;
    p.b0 = (int)a.div_h.div((uint32_t)(tile_y << (a.bt_log2 - a.bx_log2)));
    p.b = p.sb ? p.b0 : (int)a.div_h.div((uint32_t)row);
    p.y = row - p.b * a.H;
    // Widths that VEC does not divide: the last lane of a row moves back so that it ends with the row, overlapping its
    // neighbour by `dup` pixels.  Both lanes compute the same values for those pixels and store them twice -- every
    // access stays a full vector, and there is no tail code.  (Reductions over pixels must skip the duplicates.)
    p.x = xv * VEC;
    p.dup = 0;
    if (VEC > 1 && p.x > a.W - VEC) { p.dup = p.x - (a.W - VEC); p.x = a.W - VEC; }
    p.pix = (int64_t)p.y * a.W + p.x;
    p.src = p.pix;
    if (a.tiled) {        // MaterialBase.tile (base.py:524-537) as wrap-around addressing: texel (y mod h, x mod w)
        const uint32_t yg = (uint32_t)(p.y + a.y_offset), xg = (uint32_t)p.x;
        const uint32_t ys = yg - a.div_mh.div(yg) * (uint32_t)a.map_h;
        const uint32_t xs = xg - a.div_mw.div(xg) * (uint32_t)a.map_w;     // VEC divides map_w: a lane never straddles a seam
        p.src = (int64_t)ys * a.map_w + xs;
    }
    return p;
}

template <int VEC> struct Texels { float al[3][VEC], nm[3][VEC], ro[VEC], me[VEC], sp[3][VEC]; };

// Issues every load of the lane's texels; nothing here waits on memory.
// `Src` is KArgs, or KBlend (the second material of a fused blend): same member names.
// SB / HN: scalar plane addresses / a normal map, as compile-time facts -- no branch between the loads, so that a caller that
// loads several materials back to back (the batch-inner kernel) keeps ALL their loads in one basic block: with the flags
// tested per material the compiler closes every material's block with the conversions of its values, i.e. with a wait for
// its loads, and the materials' memory latencies add up.
template <int WF, typename TI, int VEC, bool NT, bool SB, bool HN, class Src>
__device__ __forceinline__ void load_texels_fixed(const Src &a, const LanePos &p, Texels<VEC> &t, int material = -1) {
    if constexpr (SB) {                                 // scalar plane addresses, one lane offset for all planes
        const int b = material < 0 ? p.b0 : material;
        const uint32_t src = (uint32_t)p.src;
#pragma unroll
        for (int c = 0; c < 3; ++c) Ld<TI, VEC>::template load<NT>(plane_at<TI>(a.albedo, b * a.a_bs + c * a.a_cs, src), 0, t.al[c]);
        if constexpr (HN) {
#pragma unroll
            for (int c = 0; c < 3; ++c) Ld<TI, VEC>::template load<NT>(plane_at<TI>(a.normal, b * a.n_bs + c * a.n_cs, src), 0, t.nm[c]);
        }
        Ld<TI, VEC>::template load<NT>(plane_at<TI>(a.rough, b * a.r_bs, src), 0, t.ro);
        if (WF != PBR_WORKFLOW_SPECULAR) Ld<TI, VEC>::template load<NT>(plane_at<TI>(a.metal, b * a.m_bs, src), 0, t.me);
        if (WF == PBR_WORKFLOW_SPECULAR) {
#pragma unroll
            for (int c = 0; c < 3; ++c) Ld<TI, VEC>::template load<NT>(plane_at<TI>(a.spec, b * a.s_bs + c * a.s_cs, src), 0, t.sp[c]);
        }
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) Ld<TI, VEC>::template load<NT>(a.albedo, p.b * a.a_bs + c * a.a_cs + p.src, t.al[c]);
        if constexpr (HN) {
#pragma unroll
            for (int c = 0; c < 3; ++c) Ld<TI, VEC>::template load<NT>(a.normal, p.b * a.n_bs + c * a.n_cs + p.src, t.nm[c]);
        }
        Ld<TI, VEC>::template load<NT>(a.rough, p.b * a.r_bs + p.src, t.ro);
        if (WF != PBR_WORKFLOW_SPECULAR) Ld<TI, VEC>::template load<NT>(a.metal, p.b * a.m_bs + p.src, t.me);
        if (WF == PBR_WORKFLOW_SPECULAR) {
#pragma unroll
            for (int c = 0; c < 3; ++c) Ld<TI, VEC>::template load<NT>(a.spec, p.b * a.s_bs + c * a.s_cs + p.src, t.sp[c]);
        }
    }
}

template <int WF, typename TI, int VEC, bool NT, class Src>
__device__ __forceinline__ void load_texels_paced(const Src &a, bool has_normal, const LanePos &p, Texels<VEC> &t, int material = -1) {
    if (p.sb) {                                         // scalar plane addresses, one lane offset for all planes
        const int b = material < 0 ? p.b0 : material;
        const uint32_t src = (uint32_t)p.src;
#pragma unroll
        for (int c = 0; c < 3; ++c) Ld<TI, VEC>::template load<NT>(plane_at<TI>(a.albedo, b * a.a_bs + c * a.a_cs, src), 0, t.al[c]);
        if (has_normal) {
#pragma unroll
            for (int c = 0; c < 3; ++c) Ld<TI, VEC>::template load<NT>(plane_at<TI>(a.normal, b * a.n_bs + c * a.n_cs, src), 0, t.nm[c]);
        }
        Ld<TI, VEC>::template load<NT>(plane_at<TI>(a.rough, b * a.r_bs, src), 0, t.ro);
        if (WF != PBR_WORKFLOW_SPECULAR) Ld<TI, VEC>::template load<NT>(plane_at<TI>(a.metal, b * a.m_bs, src), 0, t.me);
        if (WF == PBR_WORKFLOW_SPECULAR) {
#pragma unroll
            for (int c = 0; c < 3; ++c) Ld<TI, VEC>::template load<NT>(plane_at<TI>(a.spec, b * a.s_bs + c * a.s_cs, src), 0, t.sp[c]);
        }
        return;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) Ld<TI, VEC>::template load<NT>(a.albedo, p.b * a.a_bs + c * a.a_cs + p.src, t.al[c]);
    if (has_normal) {
#pragma unroll
        for (int c = 0; c < 3; ++c) Ld<TI, VEC>::template load<NT>(a.normal, p.b * a.n_bs + c * a.n_cs + p.src, t.nm[c]);
    }
    Ld<TI, VEC>::template load<NT>(a.rough, p.b * a.r_bs + p.src, t.ro);
    if (WF != PBR_WORKFLOW_SPECULAR) Ld<TI, VEC>::template load<NT>(a.metal, p.b * a.m_bs + p.src, t.me);
    if (WF == PBR_WORKFLOW_SPECULAR) {
#pragma unroll
        for (int c = 0; c < 3; ++c) Ld<TI, VEC>::template load<NT>(a.spec, p.b * a.s_bs + c * a.s_cs + p.src, t.sp[c]);
    }
}

// fp32 maps keep the flags between the loads (load_texels_paced): their blocks end without a wait (no conversion), and the
// one-light fp32 kernels measure 1-2 % FASTER with the normal-map test sitting between the albedo loads and the rest
// (4096^2: 114.2 against 116.4 us; backward 213 / 215) -- the memory system likes the loads of a wave slightly paced.
template <int WF, typename TI, int VEC, bool NT, class Src>
__device__ __forceinline__ void load_texels(const Src &a, bool has_normal, const LanePos &p, Texels<VEC> &t, int material = -1) {
    if constexpr (sizeof(TI) == 4) {
        load_texels_paced<WF, TI, VEC, NT>(a, has_normal, p, t, material);
    } else if (p.sb) {
        if (has_normal) load_texels_fixed<WF, TI, VEC, NT, true, true>(a, p, t, material);
        else load_texels_fixed<WF, TI, VEC, NT, true, false>(a, p, t, material);
    } else {
        if (has_normal) load_texels_fixed<WF, TI, VEC, NT, false, true>(a, p, t, material);
        else load_texels_fixed<WF, TI, VEC, NT, false, false>(a, p, t, material);
    }
}

// The real type of the shading code.  PACKED: two pixels per instruction (v_pk_* fp32).  Measured
// in-process A/B on MI355X (tools/tune.py --altlib): packed math makes the VALU-bound 16-light
// kernel 22 % faster (940 vs 1199 us on 2 x 4096^2 fp16 maps) but the HBM-bound one-light fp32
// kernel 15 % SLOWER (137 vs 119 us) although it halves its VALU-busy time; with fp16 maps and
// one light (neither unit saturated) it is 5 % faster -- so it is used for several lights and
// for fp16 maps, not for the fp32 one-light kernels.
template <int VEC, bool PACKED> struct RealOf { using type = float; static constexpr int N = VEC; };
template <int VEC> struct RealOf<VEC, true> { using type = f32x2; static constexpr int N = VEC / 2; };
template <> struct RealOf<1, true> { using type = float; static constexpr int N = 1; };

template <class R> __device__ __forceinline__ R gather(const float *v, int g);
template <> __device__ __forceinline__ float gather<float>(const float *v, int g) { return v[g]; }
template <> __device__ __forceinline__ f32x2 gather<f32x2>(const float *v, int g) { return f32x2{v[2 * g], v[2 * g + 1]}; }
__device__ __forceinline__ void scatter(float *v, int g, float r) { v[g] = r; }
__device__ __forceinline__ void scatter(float *v, int g, f32x2 r) { v[2 * g] = r.x; v[2 * g + 1] = r.y; }

// Material terms of pixel group g of the lane (:99-118): base colour, F0, kD scale, normal, roughness.
template <int WF, int VEC, class R>
__device__ __forceinline__ void material_terms(const Texels<VEC> &t, int g, const Vec3 &V, PixelTermsT<R> &pt) {
    R base[3], f0[3], kd_scale = splat<R>(1.0f);
#pragma unroll
    for (int c = 0; c < 3; ++c) base[c] = gather<R>(t.al[c], g);
    if (WF == PBR_WORKFLOW_METALLIC) {
        const R m = gather<R>(t.me, g);
        kd_scale = splat<R>(1.0f) - m;                                                            // :170
        const R d0 = kd_scale * kDielectricF0;
#pragma unroll
        for (int c = 0; c < 3; ++c) f0[c] = fma_(m, base[c], d0);                                   // lerp(0.04, base, m) :107
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) f0[c] = gather<R>(t.sp[c], g);                                // :112-113
    }
    const Vec3T<R> n = {gather<R>(t.nm[0], g), gather<R>(t.nm[1], g), gather<R>(t.nm[2], g)};
    pixel_terms(n, V, gather<R>(t.ro, g), base, f0, kd_scale, pt);
}

// x grid values of the lane's pixels (torch.linspace over W, :132), per pixel group.  The two-ended formula is
// x_i = a + step i below the midpoint, b - step (n-1-i) above it.  A lane's VEC pixels are consecutive, so unless the
// lane straddles the midpoint they share the side, and float(i0 + k) = float(i0) + k exactly (integers below 2^24):
// one conversion and three selects per LANE, then one add and one fma per pixel (packed: per pair) -- bit-identical
// to evaluating the formula per pixel, which costs a compare, three selects, a conversion, a subtraction and the fma
// per PIXEL (8 % of the fp16 kernel's VALU work).  Straddling lanes (only when VEC does not divide n/2) take the
// per-pixel form; the choice is made per wave.
template <class R> __device__ __forceinline__ R xs_slow(const KArgs &a, int x0, int g);
template <> __device__ __forceinline__ float xs_slow<float>(const KArgs &a, int x0, int g) {
    return linspace_at(a.x0, a.x1, a.xstep, a.W, x0 + g);
}
template <> __device__ __forceinline__ f32x2 xs_slow<f32x2>(const KArgs &a, int x0, int g) {
    return f32x2{linspace_at(a.x0, a.x1, a.xstep, a.W, x0 + 2 * g), linspace_at(a.x0, a.x1, a.xstep, a.W, x0 + 2 * g + 1)};
}
template <class R> __device__ __forceinline__ R lane_offsets(int g);           // {first pixel of group g, ...} as floats
template <> __device__ __forceinline__ float lane_offsets<float>(int g) { return (float)g; }
template <> __device__ __forceinline__ f32x2 lane_offsets<f32x2>(int g) { return f32x2{(float)(2 * g), (float)(2 * g + 1)}; }

template <class R> __device__ __forceinline__ R xs_slow_w(const KArgs &a, int W, int x0, int g);
template <> __device__ __forceinline__ float xs_slow_w<float>(const KArgs &a, int W, int x0, int g) {
    return linspace_at(a.x0, a.x1, a.xstep, W, x0 + g);
}
template <> __device__ __forceinline__ f32x2 xs_slow_w<f32x2>(const KArgs &a, int W, int x0, int g) {
    return f32x2{linspace_at(a.x0, a.x1, a.xstep, W, x0 + 2 * g), linspace_at(a.x0, a.x1, a.xstep, W, x0 + 2 * g + 1)};
}

// `W`: the width the grid spans (a.W, or the output's width when the lanes walk source texels: cook_torrance_repeat_kernel)
template <class R, int NG, int VEC>
__device__ __forceinline__ void x_grid_w(const KArgs &a, int W, int x0, R xs[NG]) {
    const int half = W >> 1;
    const bool lo = x0 < half;
    const bool one_side = lo == (x0 + VEC - 1 < half);
    if (__all(one_side)) {
        const float sgn = lo ? 1.0f : -1.0f;
        const float f0 = (float)(lo ? x0 : W - 1 - x0);
        const float st = lo ? a.xstep : -a.xstep, base = lo ? a.x0 : a.x1;
#pragma unroll
        for (int g = 0; g < NG; ++g)
            xs[g] = fma_(splat<R>(st), fma_(splat<R>(sgn), lane_offsets<R>(g), splat<R>(f0)), splat<R>(base));
    } else {
#pragma unroll
        for (int g = 0; g < NG; ++g) xs[g] = xs_slow_w<R>(a, W, x0, g);
    }
}
template <class R, int NG, int VEC>
__device__ __forceinline__ void x_grid(const KArgs &a, int x0, R xs[NG]) { x_grid_w<R, NG, VEC>(a, a.W, x0, xs); }

template <int LIGHT, class R>
__device__ __forceinline__ LightGeomT<R> light_geom(const LightU &lu, const Vec3 &V, R xs, float ys) {
    if (LIGHT == PBR_LIGHT_POINT) return point_light_geom<R>(V, Vec3{lu.l[0], lu.l[1], lu.l[2]}, xs, ys);
    LightGeomT<R> g;                       // directional: everything folded on the host, wave-uniform
    g.d = {splat<R>(lu.l[0]), splat<R>(lu.l[1]), splat<R>(lu.l[2])};
    g.rinv = splat<R>(1.0f); g.rdist = splat<R>(1.0f);
    g.h = {splat<R>(lu.h[0]), splat<R>(lu.h[1]), splat<R>(lu.h[2])};
    g.rhh = splat<R>(lu.rhh); g.rh = sqrt_hw(g.rhh); g.p5 = splat<R>(lu.p5); g.om5 = splat<R>(1.0f - lu.p5); g.att = splat<R>(1.0f);
    return g;
}

// Decodes the lane's texels in place (cooktorrance.py:99-118 and the conversions it calls): +Z for a missing normal map,
// sRGB -> linear albedo, the in-kernel metallic -> diffuse/specular conversion, sRGB -> linear specular.  Run-time flags
// are wave-uniform and each guards ONE hoisted block over all VEC pixels.
template <int WF, int VEC, bool PACKED>
__device__ __forceinline__ void decode_texels(const KArgs &a, Texels<VEC> &t) {
    using R = typename RealOf<VEC, PACKED>::type;
    constexpr int NG = RealOf<VEC, PACKED>::N;
    if (!a.has_normal) {                                                    // +Z, :147-152
#pragma unroll
        for (int j = 0; j < VEC; ++j) { t.nm[0][j] = 0.0f; t.nm[1][j] = 0.0f; t.nm[2][j] = 1.0f; }
    }
    // ---- colour decode (base.py:262-277; diffuse.py:76-91; metallic.py:98-108)
    if (a.albedo_srgb) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int g = 0; g < NG; ++g) scatter(t.al[c], g, srgb_to_linear(gather<R>(t.al[c], g)));
    }
    if (WF == PBR_WORKFLOW_CONVERTED) {   // to_diffuse_specular_material, then the specular workflow
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const R m = gather<R>(t.me, g), om = splat<R>(1.0f) - m;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const R al = gather<R>(t.al[c], g);
                scatter(t.sp[c], g, fma_(al, m, om * kDielectricF0));       // metallic.py:108
                scatter(t.al[c], g, al * om);                               // metallic.py:105
            }
        }
    }
    if (WF != PBR_WORKFLOW_METALLIC && a.spec_srgb) {   // CONVERTED: upstream default decodes again (F6)
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int g = 0; g < NG; ++g) scatter(t.sp[c], g, srgb_to_linear(gather<R>(t.sp[c], g)));
    }

}

// Everything between the loads and the stores (cooktorrance.py:99-180 and the conversions it calls).
// Run-time flags (sRGB decode/encode, normal present) are wave-uniform and each guards ONE hoisted
// block over all VEC pixels, so the shading code stays one basic block and the scheduler can
// interleave the pixel groups' transcendental latencies.
template <int LIGHT, int WF, typename TO, int VEC, bool MULTI, bool NT, bool PACKED>
__device__ __forceinline__ void shade_and_store(const KArgs &a, const LanePos &p, Texels<VEC> &t) {
    using R = typename RealOf<VEC, PACKED>::type;
    constexpr int NG = RealOf<VEC, PACKED>::N;                             // pixel groups per lane
    decode_texels<WF, VEC, PACKED>(a, t);

    const Vec3 V = view_of(a);
    float ys = 0.0f;
    if (LIGHT == PBR_LIGHT_POINT) ys = linspace_at(a.y0, a.y1, a.ystep, a.H_total, p.y + a.y_offset);

    R res[3][NG];
    if constexpr (!MULTI) {
        // ---- one light: group by group, terms and shading back to back (shortest live ranges)
        const LightU lu = light_of(a, 0);
        R xs[NG];
        if (LIGHT == PBR_LIGHT_POINT) x_grid<R, NG, VEC>(a, p.x, xs);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            PixelTermsT<R> pt;
            material_terms<WF, VEC, R>(t, g, V, pt);
            const LightGeomT<R> lg = light_geom<LIGHT, R>(lu, V, LIGHT == PBR_LIGHT_POINT ? xs[g] : splat<R>(0.0f), ys);
            R col[3];
            shade_light(pt, lg, lu.inten, col);
#pragma unroll
            for (int c = 0; c < 3; ++c) res[c][g] = col[c];
        }
    } else {
        // ---- several lights (H12): light-independent terms of every pixel first, then lights in the
        // OUTER (uniform) loop and the lane's pixel groups inside it, so each light's scalar loads and
        // loop overhead are paid once per VEC pixels and the groups' transcendental latencies interleave
        PixelTermsT<R> pt[NG];
        R xs[NG];
        if (LIGHT == PBR_LIGHT_POINT) {
            x_grid<R, NG, VEC>(a, p.x, xs);
        } else {
#pragma unroll
            for (int g = 0; g < NG; ++g) xs[g] = splat<R>(0.0f);
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) material_terms<WF, VEC, R>(t, g, V, pt[g]);
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int g = 0; g < NG; ++g) res[c][g] = splat<R>(0.0f);
        for (int l = 0; l < a.n_lights; ++l) {
            const LightU lu = light_of(a, l);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const LightGeomT<R> lg = light_geom<LIGHT, R>(lu, V, xs[g], ys);
                R col[3];
                shade_light(pt[g], lg, lu.inten, col);
#pragma unroll
                for (int c = 0; c < 3; ++c) res[c][g] += col[c];
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c)                                         // sum of per-light clamped terms, clamped
#pragma unroll
            for (int g = 0; g < NG; ++g) res[c][g] = clamp01(res[c][g]);
    }
    if (a.out_srgb) {                                                       // :179-180
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int g = 0; g < NG; ++g) res[c][g] = linear_to_srgb_unit(res[c][g]);
    }
    if constexpr (VEC == 8 && sizeof(TO) == 4) {
        // A lane of the 8-pixel kernels holds 32 contiguous bytes of every result plane.  Stored as they stand, each
        // global_store_dwordx4 covers 16 of every 32 bytes of a 2 KiB span -- half-filled write requests, measured
        // 5 % slower (4 x 4096^2 fp16 maps: 329 -> 313 us) than stores that each cover a contiguous 1 KiB.  So the lanes
        // of a row swap their 16-byte pieces first: piece q of the row's 2 bx pieces is produced by lane q / 2 and
        // stored by lane q mod bx, through 2.25 KiB of LDS per wave and plane (first halves in one 1 KiB block, second
        // halves 128 bytes -- 32 banks -- further on, so both the b128 writes and the interleaved reads are
        // conflict-free).  One wave only ever talks to itself here: a wave barrier orders the LDS traffic.
        if (a.xpose) {
            extern __shared__ f32x4 xp_all[];                                 // kXposeLdsPerWave bytes per wave, sized by the launcher
            f32x4 (*xp)[3][144] = reinterpret_cast<f32x4 (*)[3][144]>(xp_all);   // [wave of the workgroup][plane][64 + 8 pad + 64 + 8]
            const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
            const int bx = 1 << a.bx_log2, col = lane & (bx - 1), row0 = lane - col;
            // lanes of this row that exist (the last tile of a row may be partial): its 2 nvt pieces are dealt to them
            const int nvt = min(bx, a.wv - (p.x / VEC - col));
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float o[VEC];
#pragma unroll
                for (int g = 0; g < NG; ++g) scatter(o, g, res[c][g]);
                xp[wave][c][lane] = f32x4{o[0], o[1], o[2], o[3]};
                xp[wave][c][72 + lane] = f32x4{o[4], o[5], o[6], o[7]};
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int64_t base = p.b * a.o_bs + p.pix - 4 * col;              // where piece `col` of the row goes
#pragma unroll
            for (int c = 0; c < 3; ++c) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const int q = s2 * nvt + col;                             // piece index inside the row
                    const f32x4 v = xp[wave][c][(q & 1) * 72 + row0 + (q >> 1)];
                    f32x4 *dst = p.sb ? static_cast<f32x4 *>(plane_at<float>(a.out, p.b0 * a.o_bs + c * a.o_cs, (uint32_t)p.pix - 4 * col + 4 * s2 * nvt))
                                      : reinterpret_cast<f32x4 *>(static_cast<float *>(a.out) + base + c * a.o_cs + 4 * s2 * nvt);
                    if (NT) __builtin_nontemporal_store(v, dst); else *dst = v;
                }
            }
            return;
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float o[VEC];
#pragma unroll
        for (int g = 0; g < NG; ++g) scatter(o, g, res[c][g]);
        if (p.sb) Ld<TO, VEC>::template store<NT>(plane_at<TO>(a.out, p.b0 * a.o_bs + c * a.o_cs, (uint32_t)p.pix), 0, o);
        else Ld<TO, VEC>::template store<NT>(a.out, p.b * a.o_bs + c * a.o_cs + p.pix, o);
    }
}

// ------------------------------------------------------------------ kernel
//   LIGHT: PBR_LIGHT_*     WF: PBR_WORKFLOW_*     TI/TO: map / output storage types
//   VEC: pixels per lane (4 = 16-byte fp32 accesses, 8 = 16-byte fp16 accesses, 1 = ragged widths /
//        unaligned views)
//   MULTI: more than one light (uniform loop) -- the single-light body is straight-line
//   PACK1: the one-light fp32 body with packed arithmetic too.  Streaming launches lose with it (see RealOf above), but a
//        launch over TILED maps (MaterialBase.tile fused as wrap-around addressing) re-reads its texels from L2 / the
//        memory-side cache and is VALU-bound on the scalar body (valu_busy 0.885, profiles/r03_kernels.json): there the
//        packed body is the faster one (cook_torrance.hip: pick_kernel).
// 1-D grid, one tile per workgroup, tiles ordered x fastest.
template <int LIGHT, int WF, typename TI, typename TO, int VEC, bool MULTI, bool NT, bool PACK1 = false>
// Occupancy: the one-light kernels are HBM-bound and want exactly 3 waves per SIMD (12 per CU): fewer cannot
// cover the latency, more only add concurrent plane streams that fight for DRAM pages (measured, DESIGN.md 3.2:
// 2 -> 126 us, 3 -> 114 us, 4 -> 123 us, uncapped (7) -> 129 us on a 4096^2 map).  amdgpu_waves_per_eu(3,3)
// makes the register allocation enforce it.  The multi-light body is VALU-bound and wants >= 4.
#ifndef PBR_WAVES_PER_EU
#define PBR_WAVES_PER_EU 3
#endif
__global__ __launch_bounds__(256)
__attribute__((amdgpu_waves_per_eu((MULTI || PACK1) ? 4 : PBR_WAVES_PER_EU, (MULTI || PACK1) ? 8 : PBR_WAVES_PER_EU)))
void cook_torrance_kernel(const KArgs a) {
    const uint32_t tile = tile_of_workgroup(a, blockIdx.x);
    const int ty = (int)a.div_tx.div(tile);
    const LanePos p = lane_pos<VEC>(a, (int)tile - ty * a.tiles_x, ty);
    if (!p.valid) return;
    Texels<VEC> t;
    load_texels<WF, TI, VEC, NT>(a, a.has_normal != 0, p, t);
    shade_and_store<LIGHT, WF, TO, VEC, MULTI, NT, (MULTI || sizeof(TI) == 2 || PACK1)>(a, p, t);
}

// ------------------------------------------------------------------ batch-inner kernel (several lights)
// With several lights the launch is VALU-bound (config 5: 16 lights, VALUs 94 % busy, HBM at 15 %), and ~45 % of a
// light evaluation -- the light geometry: d, 1/dist, attenuation, half vector, 1/|h|^2, the Fresnel power -- depends on
// the pixel POSITION only, not on the material ("batch reuse of light geometry").  Here a lane owns
// VEC pixel positions and NB materials of the batch at those positions: lights in the outer loop, the geometry of a
// light once per position, then the NB materials' shading (NB independent dependency chains per light: the ILP a
// VALU-bound body wants).  Same functions, same operation order per pixel as cook_torrance_kernel<.., MULTI = true>.
//   rows = (B / NB) * H: lane_pos' "material" index is the GROUP of NB consecutive materials.
template <int LIGHT, int WF, typename TI, typename TO, int VEC, int NB, bool NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 4)))
void cook_torrance_batch_kernel(const KArgs a) {
    using R = typename RealOf<VEC, true>::type;
    constexpr int NG = RealOf<VEC, true>::N;
    const uint32_t tile = tile_of_workgroup(a, blockIdx.x);
    const int ty = (int)a.div_tx.div(tile);
    const LanePos p = lane_pos<VEC>(a, (int)tile - ty * a.tiles_x, ty);
    if (!p.valid) return;
    Texels<VEC> t[NB];
    LanePos pj[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) { pj[j] = p; pj[j].b = p.b * NB + j; }
    // the flags tested ONCE around the loads of all NB materials (load_texels_fixed): 8 NB loads in flight, one wait
    auto load_all = [&](auto sb, auto hn) {
#pragma unroll
        for (int j = 0; j < NB; ++j)
            load_texels_fixed<WF, TI, VEC, NT, decltype(sb)::value, decltype(hn)::value>(a, pj[j], t[j], p.b0 * NB + j);
    };
    if (p.sb) {
        if (a.has_normal) load_all(std::true_type{}, std::true_type{}); else load_all(std::true_type{}, std::false_type{});
    } else {
        if (a.has_normal) load_all(std::false_type{}, std::true_type{}); else load_all(std::false_type{}, std::false_type{});
    }
    const Vec3 V = view_of(a);
    PixelTermsT<R> pt[NB][NG];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        decode_texels<WF, VEC, true>(a, t[j]);
#pragma unroll
        for (int g = 0; g < NG; ++g) material_terms<WF, VEC, R>(t[j], g, V, pt[j][g]);
    }
    float ys = 0.0f;
    R xs[NG];
    if (LIGHT == PBR_LIGHT_POINT) {
        ys = linspace_at(a.y0, a.y1, a.ystep, a.H_total, p.y + a.y_offset);
        x_grid<R, NG, VEC>(a, p.x, xs);
    } else {
#pragma unroll
        for (int g = 0; g < NG; ++g) xs[g] = splat<R>(0.0f);
    }
    R res[NB][3][NG];
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int g = 0; g < NG; ++g) res[j][c][g] = splat<R>(0.0f);
    auto light_loop = [&](auto grey) {
        for (int l = 0; l < a.n_lights; ++l) {
            const LightU lu = light_of(a, l);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const LightGeomT<R> lg = light_geom<LIGHT, R>(lu, V, xs[g], ys);
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    R col[3];
                    shade_light<decltype(grey)::value>(pt[j][g], lg, lu.inten, col);
#pragma unroll
                    for (int c = 0; c < 3; ++c) res[j][c][g] += col[c];
                }
            }
        }
    };
    if (grey_lights_of(a)) light_loop(std::true_type{}); else light_loop(std::false_type{});
#pragma unroll
    for (int j = 0; j < NB; ++j) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float o[VEC];
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                R v = clamp01(res[j][c][g]);                                  // sum of per-light clamped terms, clamped
                if (a.out_srgb) v = linear_to_srgb_unit(v);                     // :179-180
                scatter(o, g, v);
            }
            if (p.sb) Ld<TO, VEC>::template store<NT>(plane_at<TO>(a.out, (p.b0 * NB + j) * a.o_bs + c * a.o_cs, (uint32_t)p.pix), 0, o);
            else Ld<TO, VEC>::template store<NT>(a.out, pj[j].b * a.o_bs + c * a.o_cs + p.pix, o);
        }
    }
}

// ------------------------------------------------------------------ repeat-inner kernel (MaterialBase.tile fused, whole output)
// material.tile(n) (base.py:524-537) repeats every map n x n times; the evaluation of the repeated maps differs between the
// repeats only in the point light's geometry.  cook_torrance_kernel handles it as wrap-around addressing: every texel is
// loaded, decoded (the sRGB transfer: 6 transcendentals) and turned into its light-independent terms once per REPEAT, and
// whether the second read comes from a cache is left to the memory system (measured: 1.40 x the maps from HBM in row order).
// Here the grid walks the SOURCE maps instead: a lane loads its texels once (streaming loads: nothing is read twice any
// more), decodes them once, forms PixelTerms once, and then evaluates light geometry + shading + encode at each of the
// rep_y x rep_x output positions and stores there.  HBM reads are 1.0 x the maps by construction; the decode and the pixel
// terms cost 1 / (rep_y rep_x) per output pixel; a directional light (position-independent) is evaluated ONCE and stored
// rep_y x rep_x times.  Same functions in the same order per pixel as cook_torrance_kernel: bit-identical to evaluating the
// materialised repeat.  One light, 4 texels per lane (fp16 maps: 8-byte loads -- loads are the minor stream here), packed
// arithmetic; row bands of the tiled output (multi-GPU shards) of any height: a band thinner than a period walks the window of source rows
// it touches (repeat_window); ragged map widths keep their 4-texel lanes (the last lane of a row moves back: lane_pos).
// NTL / NTS: the streaming hint on the loads / on the stores.
// MULTI (round 5): several lights -- the light loop sits INSIDE the position loop (a position's lights are summed, clamped and encoded as
// cook_torrance_kernel<.., MULTI = true> does: per-light clamp, sum, clamp, encode), so texels are still loaded, decoded and turned into
// pixel terms once for all repeats and all lights.  The launch is VALU-bound; what the walk saves is the second read of every texel
// (the wrap-around form: 1.40 x the maps from HBM) and 1 - 1/(rep_y rep_x) of the decode.
// A row band thinner than one period of the map's rows (a multi-GPU shard of ONE tiled material): the band's rows touch a cyclic
// window of the source rows -- row i of the walk is source row (win_y0 + i) mod map_h -- and each of them exactly once vertically (the band
// test of the position loop picks that repeat), while the horizontal repeats still share the texel: every texel the band needs is loaded
// and decoded once, none that it does not need is touched (until round 6 such bands took the wrap-around form: 1.40 x the maps from HBM).
__device__ __forceinline__ void repeat_window(const KArgs &a, LanePos &p) {
    int sy = a.win_y0 + p.y;
    sy -= sy >= a.map_h ? a.map_h : 0;
    p.y = sy;
    p.src = p.pix = (int64_t)sy * a.W + p.x;
}

//   `prepare(p, t)`: what happens to the lane's texels between the loads and the decode -- nothing, or (ct_blend.hpp, round 6) the blend with a second
//   material under a mask: blended ONCE per texel, evaluated at every repeat.
template <int LIGHT, int WF, typename TI, typename TO, bool NTL, bool NTS, bool MULTI, class Prepare>
__device__ __forceinline__ void repeat_forward_body(const KArgs &a, Prepare &&prepare) {
    constexpr int VEC = 4, NG = 2;
    using R = f32x2;
    const uint32_t tile = tile_of_workgroup(a, blockIdx.x);
    const int ty = (int)a.div_tx.div(tile);
    LanePos p = lane_pos<VEC>(a, (int)tile - ty * a.tiles_x, ty);            // over the SOURCE maps: a.H x a.W texels per material
    if (!p.valid) return;
    const int PH = a.map_h;                                                  // the period of the map's rows
    if (a.H != PH) repeat_window(a, p);                                      // a thin band: the walk's rows are a cyclic window of the map's
    Texels<VEC> t;
    load_texels<WF, TI, VEC, NTL>(a, a.has_normal != 0, p, t);
    prepare(p, t);
    decode_texels<WF, VEC, true>(a, t);
    const Vec3 V = view_of(a);
    const LightU lu = light_of(a, 0);
    PixelTermsT<R> pt[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) material_terms<WF, VEC, R>(t, g, V, pt[g]);

    const uint32_t lane_out = (uint32_t)(p.y * a.out_W + p.x);              // inside the first repeat; < 2^30 when p.sb (fill_args)
    // `out` holds the rows [y_offset, y_offset + H_total) of the tiled image (all of it, or a multi-GPU shard's band): a repeat whose
    // row falls outside is skipped; the others land y_offset rows higher.  (rep may be negative; rep + the lane's part never is.)
    auto in_band = [&](int ry) { const int yy = p.y + ry * PH - a.y_offset; return yy >= 0 && yy < a.H_total; };
    auto store_at = [&](int ry, int rx, const R (&res)[3][NG]) {
        const int64_t rep = ((int64_t)ry * PH - a.y_offset) * a.out_W + (int64_t)rx * a.W;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float o[VEC];
#pragma unroll
            for (int g = 0; g < NG; ++g) scatter(o, g, res[c][g]);
            if (p.sb) Ld<TO, VEC>::template store<NTS>(plane_at<TO>(a.out, p.b0 * a.o_bs + c * a.o_cs + rep, lane_out), 0, o);
            else Ld<TO, VEC>::template store<NTS>(a.out, p.b * a.o_bs + c * a.o_cs + rep + (int64_t)p.y * a.out_W + p.x, o);
        }
    };
    auto shade = [&](const R (&xs)[NG], float ys, R (&res)[3][NG]) {
        if constexpr (MULTI) {            // lights outer (uniform), the lane's pixel groups inside: shade_and_store's order, its sums
            R sum[3][NG];
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int g = 0; g < NG; ++g) sum[c][g] = splat<R>(0.0f);
            for (int l = 0; l < a.n_lights; ++l) {
                const LightU ll = light_of(a, l);
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const LightGeomT<R> lg = light_geom<LIGHT, R>(ll, V, xs[g], ys);
                    R col[3];
                    shade_light(pt[g], lg, ll.inten, col);
#pragma unroll
                    for (int c = 0; c < 3; ++c) sum[c][g] += col[c];
                }
            }
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const R lin = clamp01(sum[c][g]);                                                     // sum of per-light clamped terms, clamped
                    res[c][g] = a.out_srgb ? linear_to_srgb_unit(lin) : lin;                              // :179-180
                }
        } else {
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const LightGeomT<R> lg = light_geom<LIGHT, R>(lu, V, xs[g], ys);
                R col[3];
                shade_light(pt[g], lg, lu.inten, col);
#pragma unroll
                for (int c = 0; c < 3; ++c) res[c][g] = a.out_srgb ? linear_to_srgb_unit(col[c]) : col[c];      // :179-180
            }
        }
    };
    if (LIGHT == PBR_LIGHT_DIRECTIONAL) {                                   // the light does not know where the pixel is (:125-127)
        R res[3][NG];
        const R xs[NG] = {splat<R>(0.0f), splat<R>(0.0f)};
        shade(xs, 0.0f, res);
        for (int ry = 0; ry < a.rep_y; ++ry) {
            if (!in_band(ry)) continue;
            for (int rx = 0; rx < a.rep_x; ++rx) store_at(ry, rx, res);
        }
        return;
    }
    for (int ry = 0; ry < a.rep_y; ++ry) {
        if (!in_band(ry)) continue;
        const float ys = linspace_at(a.y0, a.y1, a.ystep, a.out_Ht, p.y + ry * PH);
        for (int rx = 0; rx < a.rep_x; ++rx) {
            R xs[NG], res[3][NG];
            x_grid_w<R, NG, VEC>(a, a.out_W, p.x + rx * a.W, xs);
            shade(xs, ys, res);
            store_at(ry, rx, res);
        }
    }
}

template <int LIGHT, int WF, typename TI, typename TO, bool NTL, bool NTS, bool MULTI = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8)))
void cook_torrance_repeat_kernel(const KArgs a) {
    repeat_forward_body<LIGHT, WF, TI, TO, NTL, NTS, MULTI>(a, [](const LanePos &, Texels<4> &) {});
}

}  // namespace pbr
