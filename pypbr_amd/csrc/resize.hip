// resize.hip -- antialiased bilinear resize of planar maps.
//
// Replaces MaterialBase.resize (/root/reference/pypbr/materials/base.py:490-504), which calls
// torchvision.transforms.functional.resize on every (C,H,W) float map; for float tensors that is
// torch.nn.functional.interpolate(mode="bilinear", align_corners=False, antialias=...).
// ATen's antialiased kernel is separable (width pass, then height pass, fp32 intermediate); per
// output index i along an axis of input size n_in and output size n_out:
//     scale   = n_in / n_out                 support = antialias && scale >= 1 ? scale : 1
//     center  = scale * (i + 0.5)            invscale = antialias && scale >= 1 ? 1/scale : 1
//     xmin    = max(0, (int)(center - support + 0.5))
//     xsize   = min(n_in, (int)(center + support + 0.5)) - xmin
//     w_j     = max(0, 1 - |(j + xmin - center + 0.5) * invscale|),  normalised to sum 1
// With antialias off (or when up-scaling) this reduces to the ordinary 2-tap bilinear rule with
// edge clamping, so one kernel covers both settings.
//
// (The kernel families and which shapes each takes: pbr_resize_form in include/pbr_hip.h; the register-only ones live in resize_down.hpp and below, the row walk in
// resize_stream.hpp.)  Schedule of the general one: ONE kernel (resize_strip_kernel below), HEIGHT pass first.  The tap pattern down the rows is the same for
// every column, so the height pass needs no exchange between lanes: it runs on registers straight from global memory,
// and only the height-reduced strip of a tile goes through LDS for the width pass.  Tap weights are normalised once per
// tile (as ATen does) instead of per output.  4096^2 -> 2048^2, 3 planes, antialiased: 450 MB of HBM traffic for two
// passes through a workspace -> 252 MB.
// History of the schedule (resize_sweep.py (a probe of its round, removed with its knob: git 9ce0718:tools/), 3 x 4096^2 -> 2048^2 | 1024^2 | 6144^2 up-scale, us): two kernels through the
// workspace 89 | 94 | 412; the whole raw window of a tile in LDS, width pass LDS -> LDS, height pass LDS -> output
// 81.5 | 95.8 | 232 -- counters: LDS pipe 65-80 % busy (13.5 cycles per LDS instruction, ~28 bytes per clock: the width pass
// reads dwords `scale` floats apart, a bank conflict for even scales), VALUs 40 %, global loads fully hidden; this form
// with one piece per lane and step 77 | 64 | 181 (LDS pipe 24 % busy, but 83 % of the wave cycles waiting on memory), with
// 2-4 pieces = 8-16 loads in flight per lane 55 | 44 | 164 = 4.6 | 4.8 | 4.0 TB/s.  Tried and measured level or worse:
// persistent workgroups with the next window prefetched through registers, unmasked tap loops for interior tiles, four
// output columns per lane with 16-byte stores (more conflicts: 86 | 72 | 195).
#include <hip/hip_runtime.h>

#include <climits>
#include <cstdint>

#include "../../include/pbr_hip.h"
#include "tuning.hpp"

namespace pbr {


struct AxisFilter {
    float scale, support, invscale;
    int n_in;
};

__host__ __device__ __forceinline__ void tap_window(const AxisFilter &f, int i, int &xmin, int &xsize, float &center) {
    center = f.scale * ((float)i + 0.5f);
    const int lo = (int)(center - f.support + 0.5f), hi = (int)(center + f.support + 0.5f);
    xmin = lo > 0 ? lo : 0;
    xsize = (hi < f.n_in ? hi : f.n_in) - xmin;
}
// first tap of output i (the window's start): what the gradient kernels bracket their contributors with; host and device agree bit for bit
__host__ __device__ __forceinline__ int first_tap(const AxisFilter &f, int i) {
    int xmin, n; float c;
    tap_window(f, i, xmin, n, c);
    return xmin;
}

__host__ __device__ __forceinline__ float tap_weight(const AxisFilter &f, int j, int xmin, float center) {
    const float x = ((float)(j + xmin) - center + 0.5f) * f.invscale;
    return fmaxf(0.0f, 1.0f - fabsf(x));
}

// rows x n_in -> rows x n_out along the contiguous axis
__global__ __launch_bounds__(256) void resize_width_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                                           int64_t rows, int n_out, AxisFilter f) {
    const int64_t total = rows * n_out;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        const int64_t row = idx / n_out;
        const int i = (int)(idx - row * n_out);
        int xmin, xsize; float center;
        tap_window(f, i, xmin, xsize, center);
        const float *p = src + row * f.n_in + xmin;
        float acc = 0.0f, wsum = 0.0f;
        for (int j = 0; j < xsize; ++j) {
            const float w = tap_weight(f, j, xmin, center);
            acc = fmaf(w, p[j], acc);
            wsum += w;
        }
        dst[idx] = wsum != 0.0f ? acc / wsum : 0.0f;
    }
}

// planes x n_in x width -> planes x n_out x width down the rows
__global__ __launch_bounds__(256) void resize_height_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                                            int64_t planes, int n_out, int width, AxisFilter f) {
    const int64_t total = planes * n_out * width;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        const int x = (int)(idx % width);
        const int64_t t = idx / width;
        const int i = (int)(t % n_out);
        const int64_t plane = t / n_out;
        int ymin, ysize; float center;
        tap_window(f, i, ymin, ysize, center);
        const float *p = src + (plane * f.n_in + ymin) * width + x;
        float acc = 0.0f, wsum = 0.0f;
        for (int j = 0; j < ysize; ++j) {
            const float w = tap_weight(f, j, ymin, center);
            acc = fmaf(w, p[(int64_t)j * width], acc);
            wsum += w;
        }
        dst[idx] = wsum != 0.0f ? acc / wsum : 0.0f;
    }
}


// ---- gradient of the resize (what autograd derives from F.interpolate(mode="bilinear", antialias=...)) -------------------------
// The forward is out = Wy in Wx^T with the banded tap matrices above; the gradient is g_in = Wy^T g_out Wx.  Two passes through a
// workspace, each a GATHER by input index (no atomics, fixed summation order): tap windows are monotone in the output index, so the
// outputs whose window holds input k are a contiguous range; it is bracketed from the window geometry and every candidate's exact
// window is re-derived with the forward's own arithmetic (tap_window / tap_weight), the per-output normalisation 1 / sum_j w_j
// coming from a small table computed first.
__global__ __launch_bounds__(256) void resize_norm_kernel(float *__restrict__ inv, int n_out, AxisFilter f) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_out) return;
    int xmin, n; float center, wsum = 0.0f;
    tap_window(f, i, xmin, n, center);
    for (int j = 0; j < n; ++j) wsum += tap_weight(f, j, xmin, center);
    inv[i] = wsum != 0.0f ? 1.0f / wsum : 0.0f;
}

// first / last output index whose window can hold input k: |k + 0.5 - scale (i + 0.5)| <= support + 1, one more on each side
__device__ __forceinline__ void candidates(const AxisFilter &f, int k, int n_out, int &lo, int &hi) {
    const float inv = 1.0f / f.scale;
    lo = max(0, (int)floorf(((float)k - f.support - 1.0f) * inv - 0.5f) - 1);
    hi = min(n_out - 1, (int)ceilf(((float)k + f.support + 2.0f) * inv - 0.5f) + 1);
}

// pass A: tmp[plane][k][x] = sum_i Wy[i][k] g_out[plane][i][x]      (k over input rows, x over OUTPUT columns)
__global__ __launch_bounds__(256) void resize_backward_rows_kernel(const float *__restrict__ gout, float *__restrict__ tmp, const float *__restrict__ inv,
                                                                   int64_t planes, int n_out, int width, AxisFilter f) {
    const int64_t total = planes * f.n_in * width, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        const int x = (int)(idx % width);
        const int64_t t = idx / width;
        const int k = (int)(t % f.n_in);
        const int64_t plane = t / f.n_in;
        int lo, hi;
        candidates(f, k, n_out, lo, hi);
        const float *g = gout + plane * n_out * width + x;
        float acc = 0.0f;
        for (int i = lo; i <= hi; ++i) {
            int ymin, n; float center;
            tap_window(f, i, ymin, n, center);
            if (k >= ymin && k < ymin + n) acc = fmaf(tap_weight(f, k - ymin, ymin, center) * inv[i], g[(int64_t)i * width], acc);
        }
        tmp[idx] = acc;
    }
}

// pass B: g_in[row][k] = sum_i Wx[i][k] tmp[row][i]                 (rows = planes * h_in, k over input columns)
__global__ __launch_bounds__(256) void resize_backward_cols_kernel(const float *__restrict__ tmp, float *__restrict__ gin, const float *__restrict__ inv,
                                                                   int64_t rows, int n_out, AxisFilter f) {
    const int64_t total = rows * f.n_in, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        const int64_t row = idx / f.n_in;
        const int k = (int)(idx - row * f.n_in);
        int lo, hi;
        candidates(f, k, n_out, lo, hi);
        const float *g = tmp + row * n_out;
        float acc = 0.0f;
        for (int i = lo; i <= hi; ++i) {
            int xmin, n; float center;
            tap_window(f, i, xmin, n, center);
            if (k >= xmin && k < xmin + n) acc = fmaf(tap_weight(f, k - xmin, xmin, center) * inv[i], g[i], acc);
        }
        gin[idx] = acc;
    }
}

constexpr int kTileW = 64;

// A workgroup owns a toh x 64 tile of the output:
//   0. tap tables in LDS: per output column / row of the tile its first tap and its taps' weights, NORMALISED once (ATen
//      normalises the weights, then accumulates sum w_j x_j: _compute_indices_weights_aa) -- the inner loops are pure fma
//      streams, no weight arithmetic, no division;
//   1. height pass: a lane owns a 16-byte piece (4 columns) of one output row, loads that piece of each of the row's K
//      input rows straight from global memory (coalesced along x; an input row serves ~K / scale output rows and is re-read
//      from L2, not from HBM) and accumulates in registers -> strip mid[toh][in_cols] in LDS, one ds_write_b128 per piece;
//   2. width pass mid -> output: a lane keeps its column and its K weights, reads its taps from LDS, stores coalesced.
// Only taps inside a window are ever used (0 x inf must not become NaN).  The sums are formed height-first, ATen's
// width-first: the same products added in another order, a few ulp apart (tests: <= 2e-6 from ATen).
// LDS: wx[K][64] wy[K][toh] | xo[64] xn[64] yo[toh] yn[toh] | mid[toh][pitch] + 16 floats of slack.
struct StripGeom { int toh, tiles_x, tiles_y, kx, ky, pitch, vec_ok, xcd_chunk, xcd_tiles, quads; };     // XCD-contiguous order: tiles per chunk (0 = identity), tiles covered by whole blocks of 8 chunks

template <int K, bool VEC>
__device__ __forceinline__ void height_from_global(const float *__restrict__ sp, float *mid, const float *wy, const int *yo, const int *yn,
                                                   int toh, int oh, int in_cols, int pitch, int w_in, int cols_left, int tid) {
    const int cn = VEC ? (in_cols + 3) >> 2 : in_cols, total = oh * cn;
    const float inv = 1.0f / (float)cn;
    if (VEC) {
        // U pieces per lane and step, their U x K loads all in flight before the first fma: the kernel waits on memory
        // (counters: 83 % of the wave cycles), and every load in flight shortens the phase.
        constexpr int U = K <= 4 ? 4 : (K <= 8 ? 2 : 1);
        for (int e0 = tid; e0 < total; e0 += U * 256) {
            float4 v[U][K];
            int o[U], n[U], at[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e = e0 + u * 256;
                const bool live = e < total;
                o[u] = live ? (int)(((float)e + 0.5f) * inv) : 0;          // (e + 0.5) / cn is never within rounding of an integer
                const int c = e - o[u] * cn;
                n[u] = live && 4 * c < cols_left ? yn[o[u]] : 0;              // the window's last piece may start past the row's end
                at[u] = live ? o[u] * pitch + 4 * c : -1;
                const float *q = sp + (int64_t)yo[o[u]] * w_in + 4 * c;
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    v[u][j] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    if (j < n[u]) v[u][j] = *reinterpret_cast<const float4 *>(q + (int64_t)j * w_in);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    const float w = j < n[u] ? wy[j * toh + o[u]] : 0.0f;
                    acc.x = fmaf(w, v[u][j].x, acc.x); acc.y = fmaf(w, v[u][j].y, acc.y);
                    acc.z = fmaf(w, v[u][j].z, acc.z); acc.w = fmaf(w, v[u][j].w, acc.w);
                }
                if (at[u] >= 0) *reinterpret_cast<float4 *>(mid + at[u]) = acc;
            }
        }
        return;
    }
    for (int e = tid; e < total; e += 256) {
        const int o = (int)(((float)e + 0.5f) * inv), c = e - o * cn;
        const int n = c < cols_left ? yn[o] : 0;
        const float *q = sp + (int64_t)yo[o] * w_in + c;
        float acc = 0.0f;
#pragma unroll
        for (int j = 0; j < K; ++j)
            if (j < n) acc = fmaf(wy[j * toh + o], q[(int64_t)j * w_in], acc);
        mid[o * pitch + c] = acc;
    }
}

template <int K>
__device__ __forceinline__ void width_to_global(const float *mid, float *dp, const float *wx, const int *xo, const int *xn, int pitch,
                                                int oh, int ow, int oy0, int ox0, int w_out, int tid) {
    const int i = tid & (kTileW - 1), off = xo[i], n = xn[i];
    if (i >= ow) return;
    float w[K];
#pragma unroll
    for (int j = 0; j < K; ++j) w[j] = j < n ? wx[j * kTileW + i] : 0.0f;
    for (int r = tid / kTileW; r < oh; r += 2 * (256 / kTileW)) {            // a lane keeps its column; two rows per step for ILP
        const int r2 = r + 256 / kTileW;
        const bool second = r2 < oh;
        const float *q0 = mid + r * pitch + off, *q1 = mid + (second ? r2 : r) * pitch + off;
        float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
        for (int j = 0; j < K; ++j) { a0 = fmaf(w[j], j < n ? q0[j] : 0.0f, a0); a1 = fmaf(w[j], j < n ? q1[j] : 0.0f, a1); }
        dp[(int64_t)(oy0 + r) * w_out + ox0 + i] = a0;
        if (second) dp[(int64_t)(oy0 + r2) * w_out + ox0 + i] = a1;
    }
}

// The same pass with FOUR consecutive columns per lane and 16-byte stores (rows of the result 16-byte aligned, whole quads): the
// launches that write more than they read -- the gradient of a down-scale, 4 output bytes per input byte at 2x -- are bound by
// their stores, and a wave's 4-byte stores fill a 256-byte piece of a row where its 16-byte stores fill four rows of the tile.
// Same taps, same order per column: bit-identical to the one-column form.
template <int K>
__device__ __forceinline__ void width_to_global_quads(const float *mid, float *dp, const float *wx, const int *xo, const int *xn, int pitch,
                                                      int oh, int ow, int oy0, int ox0, int w_out, int tid) {
    constexpr int kLanes = kTileW / 4;                                       // lanes per tile row
    const int i4 = (tid & (kLanes - 1)) * 4;
    if (i4 >= ow) return;
    int off[4], n[4];
    float w[4][K];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        off[c] = xo[i4 + c]; n[c] = xn[i4 + c];
#pragma unroll
        for (int j = 0; j < K; ++j) w[c][j] = j < n[c] ? wx[j * kTileW + i4 + c] : 0.0f;
    }
    for (int r = tid / kLanes; r < oh; r += 256 / kLanes) {
        const float *q = mid + r * pitch;
        float a[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int j = 0; j < K; ++j) a[c] = fmaf(w[c][j], j < n[c] ? q[off[c] + j] : 0.0f, a[c]);
        *reinterpret_cast<float4 *>(dp + (int64_t)(oy0 + r) * w_out + ox0 + i4) = make_float4(a[0], a[1], a[2], a[3]);
    }
}

// TABLES (the gradient, pbr_resize_bilinear_backward): the same two phases with the tap tables TRANSPOSED -- per gradient-input
// index k the first upstream index that reads it, their number and the normalised weights, as resize_backward_tables_kernel
// wrote them to global memory -- instead of derived from the filter: "dst" is the gradient of the resize's input, "src" the
// upstream gradient.  Phase 0 copies the tile's slices of the tables into the same LDS arrays; phases 1 and 2 do not change.
struct StripTables { const int *lo_x, *cnt_x, *lo_y, *cnt_y; const float *w_x, *w_y; int nx, ny, h_src; const float *band; const int *col_base; const float *col_w; };

// WIDE: the instantiation for 17 ... 36 taps per axis (down-scales of 7x ... 17x: (int)(2 s) + 3 taps; round 5) -- its own kernel, so that its registers (36 pieces of a column in
// flight: 190 VGPRs) are not the occupancy of the common one (89).
template <bool TABLES, bool QUADS, bool WIDE = false>
__global__ __launch_bounds__(256) void resize_strip_kernel(const float *__restrict__ src, float *__restrict__ dst, int h_out,
                                                           int w_out, int w_in, StripGeom tg, AxisFilter fw, AxisFilter fh, StripTables tb) {
    extern __shared__ float lds[];
    float *wx = lds, *wy = wx + tg.kx * kTileW;
    int *xo = reinterpret_cast<int *>(wy + tg.ky * tg.toh), *xn = xo + kTileW, *yo = xn + kTileW, *yn = yo + tg.toh;
    float *mid = reinterpret_cast<float *>(yn + tg.toh);            // [toh][pitch] + 16 floats of slack (taps past a window are loaded, never used)
    __shared__ int tap_max[2];
    // Workgroups are dealt to the 8 XCDs round-robin; each XCD has its own L2.  With the identity order the left / right / upper /
    // lower neighbours of a tile -- which share its halo rows and the 128-byte lines its window starts and ends in -- run on
    // OTHER XCDs, and every shared line leaves HBM once per XCD that touches it (PMC: 1.3-1.4 x the algorithmic bytes, 2 x the input
    // when up-scaling).  Here XCD x takes the x-th CHUNK of consecutive tiles out of every block of 8 chunks, so the left / right
    // neighbours (and, with chunks of two tile rows, half of the upper / lower ones) meet in one L2.
    int tile = blockIdx.x;
    if (tg.xcd_chunk > 0 && tile < tg.xcd_tiles) {          // blocks of 8 chunks: XCD x takes chunk x of every block
        const int span = 8 * tg.xcd_chunk, blk = tile / span, r = tile - blk * span;
        tile = blk * span + (r & 7) * tg.xcd_chunk + (r >> 3);
    }
    const int per_plane = tg.tiles_x * tg.tiles_y;
    const int plane = tile / per_plane, t2 = tile - plane * per_plane;
    const int ty = t2 / tg.tiles_x, tx = t2 - ty * tg.tiles_x;
    const int ox0 = tx * kTileW, oy0 = ty * tg.toh;
    const int ow = min(kTileW, w_out - ox0), oh = min(tg.toh, h_out - oy0);
    const int tid = threadIdx.x;
    int xlo = 0, n0, xl = 0, nl = 0; float c0;                       // tap windows are monotone in the output index
    __shared__ int win[2];
    if (TABLES) {
        // the upstream columns the tile reads: first and one-past-last over its columns WITH contributors (without antialiasing a
        // down-scale leaves columns that no output reads: cnt = 0)
        if (tid < kTileW) {
            int lo = INT32_MAX, hi = 0;
            if (tid < ow) {
                const int n = tb.cnt_x[ox0 + tid];
                if (n > 0) { lo = tb.lo_x[ox0 + tid]; hi = lo + n; }
            }
            for (int o = 32; o > 0; o >>= 1) { lo = min(lo, __shfl_xor(lo, o, 64)); hi = max(hi, __shfl_xor(hi, o, 64)); }
            if (tid == 0) { win[0] = lo == INT32_MAX ? 0 : lo; win[1] = lo == INT32_MAX ? 0 : hi; }
        }
    } else {
        tap_window(fw, ox0, xlo, n0, c0);
        tap_window(fw, ox0 + ow - 1, xl, nl, c0);
    }
    // ---- phase 0: tap tables (wave 0: columns; waves 1-3: rows, with their ABSOLUTE first input row) and the largest tap counts
    if (tid < 2) tap_max[tid] = 0;
    __syncthreads();
    if (TABLES) { xlo = win[0]; xl = win[1]; }
    const int xbase = tg.vec_ok ? (xlo & ~3) : xlo;
    const int in_cols = xl + nl - xbase;
    if (tid < kTileW) {
        int xmin = 0, n = 0; float center = 0.0f, wsum = 0.0f;
        if (TABLES) {
            if (tid < ow) { xmin = tb.lo_x[ox0 + tid]; n = min(tb.cnt_x[ox0 + tid], tg.kx); }
            for (int j = 0; j < tg.kx; ++j) wx[j * kTileW + tid] = j < n ? tb.w_x[(size_t)j * tb.nx + ox0 + tid] : 0.0f;
        } else {
            if (tid < ow) {
                tap_window(fw, ox0 + tid, xmin, n, center);
                for (int j = 0; j < n; ++j) wsum += tap_weight(fw, j, xmin, center);
            }
            const float inv = wsum != 0.0f ? 1.0f / wsum : 0.0f;
            for (int j = 0; j < tg.kx; ++j) wx[j * kTileW + tid] = j < n ? tap_weight(fw, j, xmin, center) * inv : 0.0f;
        }
        xo[tid] = tid < ow && n > 0 ? xmin - xbase : 0;
        xn[tid] = n;
        int m = n;
        for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o, 64));
        if (tid == 0) atomicMax(&tap_max[0], m);
    } else {
        int m = 0;
        for (int o = tid - kTileW; o < tg.toh; o += 256 - kTileW) {
            int ymin = 0, n = 0; float center = 0.0f, wsum = 0.0f;
            if (TABLES) {
                if (o < oh) { ymin = tb.lo_y[oy0 + o]; n = min(tb.cnt_y[oy0 + o], tg.ky); }
                for (int j = 0; j < tg.ky; ++j) wy[j * tg.toh + o] = j < n ? tb.w_y[(size_t)j * tb.ny + oy0 + o] : 0.0f;
            } else {
                if (o < oh) {
                    tap_window(fh, oy0 + o, ymin, n, center);
                    for (int j = 0; j < n; ++j) wsum += tap_weight(fh, j, ymin, center);
                }
                const float inv = wsum != 0.0f ? 1.0f / wsum : 0.0f;
                for (int j = 0; j < tg.ky; ++j) wy[j * tg.toh + o] = j < n ? tap_weight(fh, j, ymin, center) * inv : 0.0f;
            }
            yo[o] = ymin;
            yn[o] = n;
            m = max(m, n);
        }
        for (int k = 32; k > 0; k >>= 1) m = max(m, __shfl_xor(m, k, 64));
        if ((tid & 63) == 0) atomicMax(&tap_max[1], m);
    }
    __syncthreads();
    const int kx = tap_max[0], ky = tap_max[1];
    // ---- phase 1: height pass, global -> mid
    const float *sp = src + (int64_t)plane * (TABLES ? tb.h_src : fh.n_in) * w_in + xbase;
    const int cols_left = w_in - xbase;
#define PBR_HEIGHT(KK) (tg.vec_ok ? height_from_global<KK, true>(sp, mid, wy, yo, yn, tg.toh, oh, in_cols, tg.pitch, w_in, cols_left, tid) \
                                  : height_from_global<KK, false>(sp, mid, wy, yo, yn, tg.toh, oh, in_cols, tg.pitch, w_in, cols_left, tid))
    if (WIDE) {
        if (ky <= 24) PBR_HEIGHT(24);
        else PBR_HEIGHT(36);
    } else if (ky <= 4) PBR_HEIGHT(4);
    else if (ky <= 6) PBR_HEIGHT(6);
    else if (ky <= 8) PBR_HEIGHT(8);
    else if (ky <= 12) PBR_HEIGHT(12);
    else PBR_HEIGHT(16);
#undef PBR_HEIGHT
    __syncthreads();
    // ---- phase 2: width pass, mid -> output
    float *dp = dst + (int64_t)plane * h_out * w_out;
    const bool quads = QUADS && (ow & 3) == 0;
#define PBR_WIDTH(KK) (quads ? width_to_global_quads<KK>(mid, dp, wx, xo, xn, tg.pitch, oh, ow, oy0, ox0, w_out, tid) \
                             : width_to_global<KK>(mid, dp, wx, xo, xn, tg.pitch, oh, ow, oy0, ox0, w_out, tid))
    if (WIDE) {
        if (kx <= 24) width_to_global<24>(mid, dp, wx, xo, xn, tg.pitch, oh, ow, oy0, ox0, w_out, tid);
        else width_to_global<36>(mid, dp, wx, xo, xn, tg.pitch, oh, ow, oy0, ox0, w_out, tid);
    } else if (kx <= 4) PBR_WIDTH(4);
    else if (kx <= 6) PBR_WIDTH(6);
    else if (kx <= 8) PBR_WIDTH(8);
    else if (kx <= 12) PBR_WIDTH(12);
    else width_to_global<16>(mid, dp, wx, xo, xn, tg.pitch, oh, ow, oy0, ox0, w_out, tid);
#undef PBR_WIDTH
}


// ---- table-driven form of the two passes (round 3, after the counters: the generic kernels above re-derive every candidate's
// window per element and are VALU-bound -- 132 + 221 us for a 2048^2 -> 4096^2 gradient, VALUs saturated, 0.15 of HBM).
// A small kernel writes, per INPUT index k of an axis, the first contributing output `lo[k]`, their number `cnt[k]` and the
// normalised weights w[j][k] (j-major: lanes that walk k read them coalesced); the passes are then pure fma streams.
// More than kBwdMaxTaps contributors per input (up-scales from ~3x on) keep the generic kernels: the launcher decides from the
// scale (contributors <= (2 support + 2) / scale + 2).
constexpr int kBwdMaxTaps = 12;
__device__ __forceinline__ void backward_table_entry(int *__restrict__ lo_out, int *__restrict__ cnt_out, float *__restrict__ w, int n_out,
                                                     const AxisFilter &f, int k) {
    if (k >= f.n_in) return;
    int lo, hi, first = -1, n = 0;
    candidates(f, k, n_out, lo, hi);
    for (int i = lo; i <= hi; ++i) {
        int xmin, sz; float center;
        tap_window(f, i, xmin, sz, center);
        if (k < xmin || k >= xmin + sz) continue;
        if (first < 0) first = i;
        const int j = i - first;                            // contributors are contiguous (windows are monotone in i)
        float wsum = 0.0f;                                  // the output's normalisation, as resize_norm_kernel forms it
        for (int q = 0; q < sz; ++q) wsum += tap_weight(f, q, xmin, center);
        if (j < kBwdMaxTaps) w[(size_t)j * f.n_in + k] = tap_weight(f, k - xmin, xmin, center) * (wsum != 0.0f ? 1.0f / wsum : 0.0f);
        n = j + 1;
    }
    n = min(n, kBwdMaxTaps);                                // (the launcher only comes here when the bound on n fits)
    for (int j = n; j < kBwdMaxTaps; ++j) w[(size_t)j * f.n_in + k] = 0.0f;
    lo_out[k] = first < 0 ? 0 : first;
    cnt_out[k] = n;
}

// both axes in one launch: workgroups [0, groups_y) write the row tables, the others the column tables
// `band` != nullptr: besides, per BAND of kBandRows consecutive gradient rows (what one wave of resize_backward_gather_kernel owns), the
// rows' weights as a dense matrix over the band's union of upstream rows -- record of kBandWords words: [0] first upstream row,
// [1] number of upstream rows (<= kBandMaxRows), [8 + 8 j + r] weight of upstream row first + j in gradient row r -- so that the
// gather kernel reads eight wave-uniform weights with one scalar load instead of looking each up through lo / cnt (the look-ups made
// it scalar-bound: 953 scalar against 752 vector instructions per wave).  A lane reads back only the entries of its own row k; the
// band's first / last upstream row come from its eight lanes by shuffles.
constexpr int kBandRows = 8, kBandMaxRows = 16, kBandWords = 8 + kBandRows * kBandMaxRows;
__global__ __launch_bounds__(256) void resize_backward_tables_kernel(int *__restrict__ lo_y, int *__restrict__ cnt_y, float *__restrict__ wy, int h_out,
                                                                     AxisFilter fh, int *__restrict__ lo_x, int *__restrict__ cnt_x,
                                                                     float *__restrict__ wx, int w_out, AxisFilter fw, int groups_y,
                                                                     float *__restrict__ band, int *__restrict__ col_base, float *__restrict__ col_w,
                                                                     int col_window) {
    if ((int)blockIdx.x >= groups_y) {
        // ... and per GROUP of four consecutive gradient columns (what one lane of the gather kernel owns): the first upstream column of
        // the group's window, col_base[group], and the 4 x col_window matrix of column weights over that window, col_w[(c W + j) groups +
        // group] -- group-minor, so that the gather kernel's lanes read each entry coalesced, with no look-up through lo / cnt in between.
        const int k = (blockIdx.x - groups_y) * 256 + threadIdx.x;
        backward_table_entry(lo_x, cnt_x, wx, w_out, fw, k);
        if (col_w == nullptr) return;
        const bool live = k < fw.n_in;
        const int first = live ? lo_x[k] : 0, n = live ? cnt_x[k] : 0;
        int lo = n > 0 ? first : INT32_MAX;
        lo = min(lo, __shfl_xor(lo, 1, 64)); lo = min(lo, __shfl_xor(lo, 2, 64));
        if (lo == INT32_MAX) lo = 0;
        const int base = lo < w_out - col_window ? lo : w_out - col_window, group = k >> 2, c = k & 3, groups = (fw.n_in + 3) >> 2;
        if (group >= groups) return;
        if (c == 0) col_base[group] = base;
        for (int j = 0; j < col_window; ++j) {
            const int d = base + j - first;
            col_w[(size_t)(c * col_window + j) * groups + group] = d >= 0 && d < n ? wx[(size_t)d * fw.n_in + k] : 0.0f;
        }
        return;
    }
    const int k = blockIdx.x * 256 + threadIdx.x;
    backward_table_entry(lo_y, cnt_y, wy, h_out, fh, k);
    if (band == nullptr) return;
    const bool live = k < fh.n_in;
    const int first = live ? lo_y[k] : 0, n = live ? cnt_y[k] : 0;
    int lo = n > 0 ? first : INT32_MAX, hi = n > 0 ? first + n : 0;
    for (int o = 1; o < kBandRows; o <<= 1) { lo = min(lo, __shfl_xor(lo, o, 64)); hi = max(hi, __shfl_xor(hi, o, 64)); }
    if (lo == INT32_MAX) lo = 0;
    const int rows = min(max(hi - lo, 0), kBandMaxRows);
    if (k - (int)(threadIdx.x & (kBandRows - 1)) >= fh.n_in) return;        // a band past the last row
    float *rec = band + (size_t)(k / kBandRows) * kBandWords;
    const int r = threadIdx.x & (kBandRows - 1);
    if (r == 0) { reinterpret_cast<int *>(rec)[0] = lo; reinterpret_cast<int *>(rec)[1] = rows; }
    for (int j = 0; j < kBandMaxRows; ++j) {
        const int d = lo + j - first;
        rec[8 + kBandRows * j + r] = d >= 0 && d < n ? wy[(size_t)d * fh.n_in + k] : 0.0f;
    }
}

// rows pass: tmp[plane][k][x] = sum_j wy[j][k] g[plane][lo[k] + j][x].  One input row k per workgroup row: lo / cnt / weights are
// uniform (scalar loads), a lane owns four consecutive columns (16-byte accesses).
__global__ __launch_bounds__(256) void resize_backward_rows_table_kernel(const float *__restrict__ gout, float *__restrict__ tmp, const int *__restrict__ lo,
                                                                         const int *__restrict__ cnt, const float *__restrict__ w, int h_in, int n_out,
                                                                         int width) {
    typedef float v4 __attribute__((ext_vector_type(4), aligned(4)));
    const int chunks = (width + 1023) / 1024;               // 1-D grid (grid.y stops at 65 535): workgroup -> (row, chunk of 1024 columns)
    const int row = blockIdx.x / chunks, plane = row / h_in, k = row - plane * h_in;
    const int x = ((blockIdx.x - row * chunks) * 256 + threadIdx.x) * 4;
    if (x >= width) return;
    const int first = lo[k], n = cnt[k];
    const float *g = gout + ((int64_t)plane * n_out + first) * width + x;
    float *t = tmp + (int64_t)row * width + x;
    if (x + 4 <= width) {
        v4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int j = 0; j < n; ++j) acc += w[(size_t)j * h_in + k] * *reinterpret_cast<const v4 *>(g + (int64_t)j * width);
        *reinterpret_cast<v4 *>(t) = acc;
    } else {
        for (int c = 0; c < width - x; ++c) {
            float acc = 0.0f;
            for (int j = 0; j < n; ++j) acc = fmaf(w[(size_t)j * h_in + k], g[(int64_t)j * width + c], acc);
            t[c] = acc;
        }
    }
}

// columns pass: g_in[row][k] = sum_j wx[j][k] tmp[row][lo[k] + j].  A lane owns input column k for kBwdRows consecutive rows: its
// offsets and weights are read once (coalesced over k) and reused for every row.
constexpr int kBwdRows = 8;
__global__ __launch_bounds__(256) void resize_backward_cols_table_kernel(const float *__restrict__ tmp, float *__restrict__ gin, const int *__restrict__ lo,
                                                                         const int *__restrict__ cnt, const float *__restrict__ w, int64_t rows, int w_in,
                                                                         int n_out) {
    const int chunks = (w_in + 255) / 256;                  // 1-D grid: workgroup -> (group of kBwdRows rows, chunk of 256 columns)
    const int64_t rgroup = blockIdx.x / chunks;
    const int k = (int)(blockIdx.x - rgroup * chunks) * 256 + threadIdx.x;
    if (k >= w_in) return;
    const int first = lo[k], n = cnt[k];
    float wk[kBwdMaxTaps];
#pragma unroll
    for (int j = 0; j < kBwdMaxTaps; ++j) wk[j] = j < n ? w[(size_t)j * w_in + k] : 0.0f;
    const int64_t r0 = rgroup * kBwdRows;
#pragma unroll 2
    for (int r = 0; r < kBwdRows; ++r) {
        const int64_t row = r0 + r;
        if (row >= rows) break;
        const float *t = tmp + row * n_out + first;
        float acc = 0.0f;
#pragma unroll
        for (int j = 0; j < kBwdMaxTaps; ++j)
            if (j < n) acc = fmaf(wk[j], t[j], acc);
        gin[row * w_in + k] = acc;
    }
}

// ---- up-scaling on both axes: two taps per axis, registers only (round 3) --------------------------------------------------
// With scale <= 1 on both axes an output pixel has at most two taps per axis (support = 1), so neither the tap tables nor the
// LDS strip of resize_strip_kernel are needed: a lane owns FOUR consecutive output pixels of a row.  Their taps lie in at most five
// consecutive input columns; the lane loads six from each of the row's two input rows (one 16-byte and one 8-byte load per row,
// element-aligned), blends the rows first (the same order as the strip kernel: height, then width), and picks each output's two
// columns out of the six with selects.  Five vector-memory instructions per four output pixels instead of tables + LDS + barriers;
// the launch is bound by its writes (2.25 output pixels per input pixel at 1.5x).  One-wave workgroups = 256 output pixels of a
// row; workgroups are dealt to the XCDs in runs of kUpRun (every XCD keeps whole bands of output rows, so the input rows two
// output rows share meet in one L2).
constexpr int kUpRunLog2 = 9;
typedef float rf4 __attribute__((ext_vector_type(4), aligned(4)));
typedef float rf2 __attribute__((ext_vector_type(2), aligned(4)));

// kUpRows (template parameter ROWS): output rows per lane -- the column taps are formed once for all of them

__device__ __forceinline__ void two_taps(const AxisFilter &f, int i, int &first, float &w0, float &w1) {
    int n; float center;
    tap_window(f, i, first, n, center);
    const float a = tap_weight(f, 0, first, center), b = n > 1 ? tap_weight(f, 1, first, center) : 0.0f;
    const float inv = __builtin_amdgcn_rcpf(a + b);         // a + b > 0: the window always holds the tap nearest the centre (1 ulp; the
    w0 = a * inv; w1 = b * inv;                             // strip kernel divides -- results agree to ~1e-7, both <= 2e-6 from ATen)
}

template <int kUpRows>
__global__ __launch_bounds__(64) void resize_up2_kernel(const float *__restrict__ src, float *__restrict__ dst, int h_in, int w_in, int h_out,
                                                        int w_out, int groups_x, int groups_y, uint32_t xcd_groups, AxisFilter fw, AxisFilter fh) {
    uint32_t wg = blockIdx.x;
    if (wg < xcd_groups) {                                  // XCD x takes runs of 1 << kUpRunLog2 consecutive workgroups (tile_of_workgroup's map)
        const uint32_t c = kUpRunLog2, xcd = wg & 7u, slot = wg >> 3;
        wg = ((slot >> c) << (c + 3)) + (xcd << c) + (slot & ((1u << c) - 1u));
    }
    const uint32_t band = wg / (uint32_t)groups_x, gx = wg - band * (uint32_t)groups_x;    // band = plane * groups_y + (y / kUpRows)
    const int plane = (int)(band / (uint32_t)groups_y), yb = (int)(band - (uint32_t)plane * (uint32_t)groups_y) * kUpRows;
    // Lanes past the row's end stay in the wave (they work on column 0 and store nothing): the rows' taps below are read ACROSS lanes,
    // and a lane that has left has no defined values (the compiler is free to form them after the exit).
    const int x_raw = ((int)gx * 64 + (int)threadIdx.x) * 4;
    const bool live = x_raw < w_out;
    const int x0 = live ? x_raw : 0;
    // The rows' taps are wave-uniform (every lane of the workgroup works on rows yb .. yb + kUpRows - 1) but floating-point: the scalar unit
    // cannot form them, and formed per lane they were ~30 vector instructions per row and lane (VALU busy 0.69 in a launch that should wait
    // for its stores).  Lane r forms row r's taps ONCE; the others read them through v_readlane (same arithmetic, same bits).
    int ty0; float tw0, tw1;
    two_taps(fh, min(yb + (int)(threadIdx.x & (kUpRows - 1)), h_out - 1), ty0, tw0, tw1);
    int first[4]; float wa[4], wb[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) two_taps(fw, min(x0 + k, w_out - 1), first[k], wa[k], wb[k]);
    const int xb = min(first[0], w_in - 6);                 // six columns from xb on, inside the row
    int o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = first[k] - xb;       // 0 .. 5 (5 only at the right edge, where the second tap's weight is 0)
    const float *sp = src + (int64_t)plane * h_in * w_in + xb;
    float *dp = dst + ((int64_t)plane * h_out) * w_out + x0;
    const bool whole = x0 + 4 <= w_out;
#pragma unroll
    for (int r = 0; r < kUpRows; ++r) {
        const int y = yb + r;
        if (y >= h_out) break;
        const int y0 = __builtin_amdgcn_readlane(ty0, r);
        const float wy0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tw0), r));
        const float wy1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tw1), r));
        const int y1 = min(y0 + 1, h_in - 1);               // a one-tap window at the last row: weight 0 on a valid row
        const float *r0 = sp + (int64_t)y0 * w_in, *r1 = sp + (int64_t)y1 * w_in;
        const rf4 a4 = *reinterpret_cast<const rf4 *>(r0), b4 = *reinterpret_cast<const rf4 *>(r1);
        const rf2 a2 = *reinterpret_cast<const rf2 *>(r0 + 4), b2 = *reinterpret_cast<const rf2 *>(r1 + 4);
        float mid[6];                                       // height pass first, as the strip kernel: acc = w0 v0, then fma(w1, v1, acc)
        mid[0] = fmaf(wy1, b4.x, wy0 * a4.x); mid[1] = fmaf(wy1, b4.y, wy0 * a4.y); mid[2] = fmaf(wy1, b4.z, wy0 * a4.z);
        mid[3] = fmaf(wy1, b4.w, wy0 * a4.w); mid[4] = fmaf(wy1, b2.x, wy0 * a2.x); mid[5] = fmaf(wy1, b2.y, wy0 * a2.y);
        float out[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ok = o[k];
            const float va = ok == 0 ? mid[0] : (ok == 1 ? mid[1] : (ok == 2 ? mid[2] : (ok == 3 ? mid[3] : (ok == 4 ? mid[4] : mid[5]))));
            const float vb = ok == 0 ? mid[1] : (ok == 1 ? mid[2] : (ok == 2 ? mid[3] : (ok == 3 ? mid[4] : mid[5])));
            out[k] = fmaf(wb[k], vb, wa[k] * va);
        }
        float *q = dp + (int64_t)y * w_out;
        if (!live) continue;
        if (whole) {
            typedef float sf4 __attribute__((ext_vector_type(4), aligned(4)));
            const sf4 v = {out[0], out[1], out[2], out[3]};
            __builtin_nontemporal_store(v, reinterpret_cast<sf4 *>(q));
        } else {
            for (int k = 0; k < w_out - x0; ++k) q[k] = out[k];
        }
    }
}


// ---- gradient of an up-scale: the transpose of resize_up2_kernel, registers only (round 4) ---------------------------------------
// out[y][x] = sum over two rows and two columns of wy wx in[..]; the gradient g_in[ky][kx] gathers g_out over the outputs whose
// two-tap windows hold (ky, kx).  Windows start at first_tap(i), which is monotone in i, so the outputs that touch gradient columns
// k0 .. k0 + 3 are the contiguous range first_tap(i) in [k0 - 1, k0 + 3]: at most 5 / scale + 1 of them.  A lane owns FOUR consecutive
// gradient columns of R rows: it finds the start of its range once (a short search around the closed-form estimate, with the forward's
// own arithmetic), builds the 4 x W matrix of column weights in registers (W = 8 | 12 | 16 upstream columns; zero where an output
// does not touch a column), then walks the upstream rows that touch its R gradient rows: W / 4 16-byte loads, 4 W fma for the width
// sum, 4 R fma into the accumulators with the row's (wave-uniform) weights.  No tables, no LDS, no barriers -- the strip kernel with
// transposed tables (resize_strip_kernel<true>) spends most of its time in per-tile set-up and between its barriers on these shapes
// (0.52 of HBM).  A gather by gradient element with a fixed summation order: deterministic, no atomics.  The launcher checks on the
// host (the same float arithmetic) that W and the row bound hold for every lane; other shapes keep the table-driven passes.
template <int W, int R>
__global__ __launch_bounds__(64) void resize_up2_backward_kernel(const float *__restrict__ gout, float *__restrict__ gin, int h_in, int w_in, int h_out,
                                                                 int w_out, int groups_x, int groups_y, uint32_t xcd_groups, AxisFilter fw, AxisFilter fh) {
    uint32_t wg = blockIdx.x;
    if (wg < xcd_groups) {                                  // XCD x takes runs of 1 << kUpRunLog2 consecutive workgroups, as the forward
        const uint32_t c = kUpRunLog2, xcd = wg & 7u, slot = wg >> 3;
        wg = ((slot >> c) << (c + 3)) + (xcd << c) + (slot & ((1u << c) - 1u));
    }
    const uint32_t band = wg / (uint32_t)groups_x, gx = wg - band * (uint32_t)groups_x;
    const int plane = (int)(band / (uint32_t)groups_y), r0 = (int)(band - (uint32_t)plane * (uint32_t)groups_y) * R;
    const int k0 = ((int)gx * 64 + (int)threadIdx.x) * 4;
    if (k0 >= w_in) return;
    // ---- columns: the first output whose window reaches column k0 - 1 or beyond
    int i_lo = 0;
    if (k0 > 1) {
        i_lo = (int)(((float)k0 - 0.5f) / fw.scale - 0.5f) - 1;
        i_lo = i_lo < 0 ? 0 : (i_lo > w_out - 1 ? w_out - 1 : i_lo);
        while (i_lo > 0 && first_tap(fw, i_lo - 1) >= k0 - 1) --i_lo;
        while (i_lo < w_out - 1 && first_tap(fw, i_lo) < k0 - 1) ++i_lo;
    }
    const int i_base = i_lo < w_out - W ? i_lo : w_out - W;            // W upstream columns from here, inside the row (w_out >= W: the launcher)
    float wx[4][W];
#pragma unroll
    for (int j = 0; j < W; ++j) {
        int first; float wa, wb;
        two_taps(fw, i_base + j, first, wa, wb);
#pragma unroll
        for (int c = 0; c < 4; ++c) wx[c][j] = (first == k0 + c ? wa : 0.0f) + (first + 1 == k0 + c ? wb : 0.0f);
    }
    // ---- rows: the upstream rows whose windows reach gradient rows r0 .. r0 + R - 1 (wave-uniform)
    int y = 0;
    if (r0 > 1) {
        y = (int)(((float)r0 - 0.5f) / fh.scale - 0.5f) - 1;
        y = y < 0 ? 0 : (y > h_out - 1 ? h_out - 1 : y);
        while (y > 0 && first_tap(fh, y - 1) >= r0 - 1) --y;
        while (y < h_out - 1 && first_tap(fh, y) < r0 - 1) ++y;
    }
    float acc[R][4];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[r][c] = 0.0f;
    const float *gp = gout + (int64_t)plane * h_out * w_out + i_base;
    for (; y < h_out; ++y) {
        int yf; float wy0, wy1;
        two_taps(fh, y, yf, wy0, wy1);
        if (yf > r0 + R - 1) break;
        const int y1 = min(yf + 1, h_in - 1);               // the forward's second row (weight 0 when the window holds one tap)
        const float *row = gp + (int64_t)y * w_out;
        float g[W];
#pragma unroll
        for (int q = 0; q < W / 4; ++q) {
            const rf4 v = *reinterpret_cast<const rf4 *>(row + 4 * q);       // cached: neighbouring lanes' and rows' windows overlap
            g[4 * q] = v.x; g[4 * q + 1] = v.y; g[4 * q + 2] = v.z; g[4 * q + 3] = v.w;
        }
        float t[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float a = wx[c][0] * g[0];
#pragma unroll
            for (int j = 1; j < W; ++j) a = fmaf(wx[c][j], g[j], a);
            t[c] = a;
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float cy = (yf == r0 + r ? wy0 : 0.0f) + (y1 == r0 + r ? wy1 : 0.0f);
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[r][c] = fmaf(cy, t[c], acc[r][c]);
        }
    }
    float *dp = gin + (int64_t)plane * h_in * w_in + k0;
    const bool whole = k0 + 4 <= w_in;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (r0 + r >= h_in) break;
        float *q = dp + (int64_t)(r0 + r) * w_in;
        if (whole) {
            typedef float sf4 __attribute__((ext_vector_type(4), aligned(4)));
            const sf4 v = {acc[r][0], acc[r][1], acc[r][2], acc[r][3]};
            __builtin_nontemporal_store(v, reinterpret_cast<sf4 *>(q));
        } else {
            for (int c = 0; c < w_in - k0; ++c) q[c] = acc[r][c];
        }
    }
}

// Host side of the kernel above: the largest number of outputs whose windows start in [k0 - 1, k0 + 3] over all lanes' k0 (multiples
// of 4), computed with the kernel's own first_tap -- the kernel's W must cover it.
static int up2_backward_window(const AxisFilter &f, int n_out) {
    int worst = 0, lo = 0, hi = 0;                       // [lo, hi): outputs with first_tap in [k0 - 1, k0 + 3], both ends monotone in k0
    for (int k0 = 0; k0 < f.n_in; k0 += 4) {
        while (lo < n_out && first_tap(f, lo) < k0 - 1) ++lo;
        if (hi < lo) hi = lo;
        while (hi < n_out && first_tap(f, hi) <= k0 + 3) ++hi;
        worst = hi - lo > worst ? hi - lo : worst;
    }
    return worst;
}


// ---- gradient of a down-scale, registers only (round 4): the two table-driven passes in one kernel without the LDS strip --------
// With the transposed tap tables in global memory (resize_backward_tables_kernel: per gradient index k the first upstream index
// lo[k] that read it, their number cnt[k] and the normalised weights w[j][k]) the gradient is a gather with short, contiguous ranges
// on both axes.  A lane owns FOUR consecutive gradient columns of R rows.  Its columns' upstream ranges overlap and are monotone, so
// their union is W <= 16 consecutive upstream columns: the lane builds the 4 x W matrix of column weights once (4 W table reads,
// coalesced over the lanes), then walks the union of its rows' upstream rows: W / 4 16-byte loads, 4 W fma for the width sums, and per
// gradient row one wave-uniform weight (scalar loads) times the four sums.  resize_strip_kernel<true> does the same work through a
// tile of LDS with three barrier-separated phases and reaches 0.52 of HBM on 4096^2 <- 2048^2; this form has no set-up to amortise.
// Gather by gradient element, fixed order: deterministic.  The launcher checks W on the host (same float arithmetic).
template <int W, int R, bool BAND>
__global__ __launch_bounds__(64) void resize_backward_gather_kernel(const float *__restrict__ gout, float *__restrict__ gin, int h_in, int w_in, int h_out,
                                                                    int w_out, int groups_x, int groups_y, uint32_t xcd_groups, StripTables tb) {
    uint32_t wg = blockIdx.x;
    if (wg < xcd_groups) {
        const uint32_t c = kUpRunLog2, xcd = wg & 7u, slot = wg >> 3;
        wg = ((slot >> c) << (c + 3)) + (xcd << c) + (slot & ((1u << c) - 1u));
    }
    const uint32_t band = wg / (uint32_t)groups_x, gx = wg - band * (uint32_t)groups_x;
    const int plane = (int)(band / (uint32_t)groups_y), r0 = (int)(band - (uint32_t)plane * (uint32_t)groups_y) * R;
    const int k0 = ((int)gx * 64 + (int)threadIdx.x) * 4;
    if (k0 >= w_in) return;
    // ---- columns
    float wx[4][W];
    int i_base;
    if (BAND) {                                  // the group's record (resize_backward_tables_kernel): 1 + 4 W coalesced loads, none waits for another
        const int group = k0 >> 2, groups = (w_in + 3) >> 2;
        i_base = tb.col_base[group];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int j = 0; j < W; ++j) wx[c][j] = tb.col_w[(size_t)(c * W + j) * groups + group];
    } else {
        int lo[4], n[4], i_lo = INT32_MAX;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int k = min(k0 + c, w_in - 1);
            lo[c] = tb.lo_x[k]; n[c] = k0 + c < w_in ? min(tb.cnt_x[k], kBwdMaxTaps) : 0;
            if (n[c] > 0) i_lo = min(i_lo, lo[c]);
        }
        if (i_lo == INT32_MAX) i_lo = 0;
        i_base = i_lo < w_out - W ? i_lo : w_out - W;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int k = min(k0 + c, w_in - 1);
#pragma unroll
            for (int j = 0; j < W; ++j) {
                const int d = i_base + j - lo[c];
                wx[c][j] = d >= 0 && d < n[c] ? tb.w_x[(size_t)d * tb.nx + k] : 0.0f;
            }
        }
    }
    // ---- rows (wave-uniform: scalar loads)
    float acc[R][4];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[r][c] = 0.0f;
    const float *gp = gout + (int64_t)plane * h_out * w_out + i_base;
    auto load_row = [&](int y, float g[W]) {
        const float *row = gp + (int64_t)y * w_out;
#pragma unroll
        for (int q = 0; q < W / 4; ++q) {
            const rf4 v = *reinterpret_cast<const rf4 *>(row + 4 * q);       // cached: neighbouring lanes' and rows' windows overlap
            g[4 * q] = v.x; g[4 * q + 1] = v.y; g[4 * q + 2] = v.z; g[4 * q + 3] = v.w;
        }
    };
    auto width_sums = [&](const float g[W], float t[4]) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float a = wx[c][0] * g[0];
#pragma unroll
            for (int j = 1; j < W; ++j) a = fmaf(wx[c][j], g[j], a);
            t[c] = a;
        }
    };
    if (BAND) {
        // the band's record (resize_backward_tables_kernel): first upstream row, their number, eight weights per upstream row
        static_assert(!BAND || R == kBandRows, "a band is what one wave owns");
        const float *rec = tb.band + (size_t)(r0 / R) * kBandWords;
        const int y_lo = reinterpret_cast<const int *>(rec)[0], y_n = reinterpret_cast<const int *>(rec)[1];
        for (int j = 0; j < y_n; ++j) {
            float g[W], t[4];
            load_row(y_lo + j, g);
            width_sums(g, t);
            const float *cw = rec + 8 + kBandRows * j;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float cy = cw[r];
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[r][c] = fmaf(cy, t[c], acc[r][c]);
            }
        }
    } else {
        int ylo[R], yn[R], y_lo = INT32_MAX, y_hi = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int k = min(r0 + r, h_in - 1);
            ylo[r] = tb.lo_y[k]; yn[r] = r0 + r < h_in ? min(tb.cnt_y[k], kBwdMaxTaps) : 0;
            if (yn[r] > 0) { y_lo = min(y_lo, ylo[r]); y_hi = max(y_hi, ylo[r] + yn[r]); }
        }
        for (int y = y_lo; y < y_hi; ++y) {
            float g[W], t[4];
            load_row(y, g);
            width_sums(g, t);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int d = y - ylo[r];
                const float cy = d >= 0 && d < yn[r] ? tb.w_y[(size_t)d * tb.ny + min(r0 + r, h_in - 1)] : 0.0f;
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[r][c] = fmaf(cy, t[c], acc[r][c]);
            }
        }
    }
    float *dp = gin + (int64_t)plane * h_in * w_in + k0;
    const bool whole = k0 + 4 <= w_in;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (r0 + r >= h_in) break;
        float *q = dp + (int64_t)(r0 + r) * w_in;
        if (whole) {
            typedef float sf4 __attribute__((ext_vector_type(4), aligned(4)));
            const sf4 v = {acc[r][0], acc[r][1], acc[r][2], acc[r][3]};
            __builtin_nontemporal_store(v, reinterpret_cast<sf4 *>(q));
        } else {
            for (int c = 0; c < w_in - k0; ++c) q[c] = acc[r][c];
        }
    }
}

// Host side: the widest union of upstream columns over all lanes' four gradient columns, from the forward's own windows (an output's
// window [xmin, xmin + size) is monotone in the output index at both ends): the outputs whose window meets [k0, k0 + 3].
static int gather_window(const AxisFilter &f, int n_out) {
    int worst = 0, lo = 0, hi = 0;
    for (int k0 = 0; k0 < f.n_in; k0 += 4) {
        int xmin, n; float c;
        while (lo < n_out) { tap_window(f, lo, xmin, n, c); if (xmin + n > k0) break; ++lo; }
        if (hi < lo) hi = lo;
        while (hi < n_out) { tap_window(f, hi, xmin, n, c); if (xmin > k0 + 3) break; ++hi; }
        worst = hi - lo > worst ? hi - lo : worst;
    }
    return worst;
}

// The same for the rows of a band: the most upstream rows any kBandRows consecutive gradient rows (starting at a multiple) gather from.
static int band_window(const AxisFilter &f, int n_out, int rows) {
    int worst = 0, lo = 0, hi = 0;
    for (int k0 = 0; k0 < f.n_in; k0 += rows) {
        int xmin, n; float c;
        while (lo < n_out) { tap_window(f, lo, xmin, n, c); if (xmin + n > k0) break; ++lo; }
        if (hi < lo) hi = lo;
        while (hi < n_out) { tap_window(f, hi, xmin, n, c); if (xmin > k0 + rows - 1) break; ++hi; }
        worst = hi - lo > worst ? hi - lo : worst;
    }
    return worst;
}

}  // namespace pbr
#include "resize_down.hpp"
#include "resize_stream.hpp"
namespace pbr {

static AxisFilter make_filter(int n_in, int n_out, bool antialias) {
    AxisFilter f;
    f.scale = (float)n_in / (float)n_out;            // area_pixel_compute_scale<float>, align_corners = False
    const bool aa = antialias && f.scale >= 1.0f;
    f.support = aa ? f.scale : 1.0f;                 // interp_size / 2 * scale, interp_size = 2
    f.invscale = aa ? 1.0f / f.scale : 1.0f;
    f.n_in = n_in;
    return f;
}

// The three weight vectors of resize_down_kernel for the whole factor S (antialiased): an axis of 16 outputs has them all -- output 0 (window clipped
// on the left), output 5 (whole window) and output 15 (clipped on the right) -- and they do not depend on the axis' length: with n_in = S n_out the
// tap positions relative to the window are small whole and half numbers, exact in float, whatever the output's index.  Formed with the strip kernel's statements (phase 0 of resize_strip_kernel).
static DownTaps down_taps(int S) {
    DownTaps t;
    const AxisFilter f = make_filter(16 * S, 16, true);
    const int which[3] = {5, 0, 15};
    float *const into[3] = {t.wi, t.wl, t.wr};
    for (int s = 0; s < 3; ++s) {
        const int i = which[s];
        int xmin, n; float center, wsum = 0.0f;
        tap_window(f, i, xmin, n, center);
        for (int j = 0; j < n; ++j) wsum += tap_weight(f, j, xmin, center);
        const float inv = wsum != 0.0f ? 1.0f / wsum : 0.0f;
        for (int j = 0; j < 32; ++j) into[s][j] = 0.0f;
        const int shift = xmin - (S * i - S / 2);            // the window's first tap among the K = 2 S of an unclipped one
        for (int j = 0; j < n; ++j)
            if (shift + j >= 0 && shift + j < 32) into[s][shift + j] = tap_weight(f, j, xmin, center) * inv;
    }
    return t;
}

// The TRANSPOSE of an up-scale by the whole factor S has the same shape: gradient element k gathers the 2 S upstream elements S k - S/2 ... S k + 3 S/2 - 1
// (the outputs whose two-tap windows hold input k), with one weight vector for every interior k and clipped ones for the first and the last --
// resize_down_kernel with other numbers in its three vectors.  Weights from the forward's own two-tap rule (resize_up2_kernel: two_taps).
static DownTaps up_transpose_taps(int S) {
    DownTaps t;
    const AxisFilter f = make_filter(16, 16 * S, false);     // 16 gradient elements, 16 S upstream; up-scales: antialiasing changes nothing
    const int which[3] = {5, 0, 15};
    float *const into[3] = {t.wi, t.wl, t.wr};
    for (int s = 0; s < 3; ++s) {
        const int k = which[s];
        for (int j = 0; j < 32; ++j) into[s][j] = 0.0f;
        for (int j = 0; j < 2 * S; ++j) {
            const int i = S * k - S / 2 + j;
            if (i < 0 || i >= 16 * S) continue;
            int first, n; float center;
            tap_window(f, i, first, n, center);
            const float a = tap_weight(f, 0, first, center), b = n > 1 ? tap_weight(f, 1, first, center) : 0.0f, inv = 1.0f / (a + b);
            into[s][j] = (first == k ? a * inv : 0.0f) + (first + 1 == k ? b * inv : 0.0f);
        }
    }
    return t;
}

// Launch of resize_down_kernel: `small` = the side with 1 / S^2 of the elements (the down-scale's result, the up-scale's gradient).  False when the
// shape is not the kernel's (the caller goes on to its other forms).
static bool launch_down(const float *large, float *small, int64_t planes, int h_small, int w_small, int S, const DownTaps &taps, hipStream_t s, bool dry = false) {
    // A lane owns 32 bytes of every row of the large side (8 / S columns of the small one) and walks down a band of rows, R rows of the small side per turn
    // of its loop, the next large row in flight.  Bands are cut so that the launch has ~1 536 waves -- six per CU, which then run side by side from
    // the first to the last row: 3 x 4096^2 -> 2048^2 | 1024^2 | 512^2 (us) with 1 280 / 1 536 / 1 792 / 2 048 / 2 560 / 3 072 / 4 096 / 6 144 waves:
    // 37.6 / 37.2 / 39.3 / 39.4 / 42.1 / 40.0 / 40.1 / 42.7 | 34.2 / 32.4 / 32.3 / 32.3 / 34.2 / 34.4 / 34.3 / 39.2 | 35.1 / 33.8 / 33.9 / 33.9 / 35.6 / 37.8 / 37.8 / 46.1
    // (tools/resize_down_probe.py; the strip kernel: 45.6 | 36.4, and 185 for the two passes the 19 taps of an 8 x down-scale fell to).  64 bytes per
    // lane (16 lines per load instruction instead of 8), 3 or 7 rows in flight, 4 or 8 rows per turn: level or 2-8 % slower; non-temporal loads:
    // 1.4-1.8 x slower (a lane's two 16-byte loads of a row are two instructions on the same lines).
    // These shapes -- one 3-plane 4096^2 map, 201 MB -- sit in the 256 MB memory-side cache between launches, and only cached (plain) loads find them there.
    // A large side that cannot (EIGHT planes, 537 MB: nothing survives a launch) reads 122.4 / 117.6 us at 1 536 / 2 048 waves for 2x (strip kernel 128), 123.0 / 112.3
    // for 4x (118), 99 / 96 for 8x (two passes: 516) -- 0.64-0.71 of HBM, what cached loads stream at on this part (tools/membench.hip: read-only plain 5.6 TB/s,
    // non-temporal 6.2).  There the lanes own 16 bytes of a row instead (4 / S columns: every line is touched by ONE instruction), loaded non-temporally,
    // three rows in flight: 111.8 (0.75) | 108.4 (0.66); on the 3-plane shapes that form costs 36.8 -> 50 | 32.0 -> 46 (it streams past the memory-side cache).
    if (S < 2 || (S > 8 && S != 16) || w_small % 4 != 0 || w_small < 8 || h_small < 2) return false;
    if (((reinterpret_cast<uintptr_t>(large) | reinterpret_cast<uintptr_t>(small)) & 15u) != 0) return false;
    const bool streams = (int64_t)planes * h_small * w_small * S * S * 4 > (256ll << 20);      // the large side does not fit the memory-side cache
    const bool narrow = streams && (S == 2 || S == 4);           // 16 bytes of a row per lane, non-temporal loads
    // columns of the small side per lane: 32 bytes of the large side's row where S divides 8, else the fewest whose S-fold is a whole number of 16-byte pieces
    const int cols = narrow ? 4 / S : (S == 16 ? 1 : (8 % S == 0 ? 8 / S : (S == 6 ? 2 : 4))), R = S == 2 ? 4 : 2;      // ... and its rows per turn of the kernel's loop (16 x: 64 bytes per lane)
    const int64_t groups_x = (w_small + 64 * cols - 1) / (64 * cols);
    int64_t bands = (streams ? 2048 : 1536) / (planes * groups_x);
    bands = bands < 1 ? 1 : bands;
    int64_t band_rows = (h_small + bands - 1) / bands;
    band_rows = (band_rows + R - 1) / R * R;
    bands = (h_small + band_rows - 1) / band_rows;               // every band holds at least one row
    const int64_t pairs = planes * bands, n_groups = pairs * groups_x;
    if (n_groups > INT32_MAX) return false;
    const uint32_t mapped = (uint32_t)((pairs / 8) * 8 * groups_x);      // the (plane, band) pairs dealt to the XCDs by eights
    void (*fn)(const float *, float *, int, int, int, int, int, uint32_t, const DownTaps) = nullptr;
    switch (S) {
        case 2: fn = resize_down_kernel<2, 4, 4, 1>; break;
        case 3: fn = resize_down_kernel<3, 2, 4, 1>; break;
        case 4: fn = resize_down_kernel<4, 2, 2, 1>; break;
        case 5: fn = resize_down_kernel<5, 2, 4, 1>; break;
        case 6: fn = resize_down_kernel<6, 2, 2, 1>; break;
        case 7: fn = resize_down_kernel<7, 2, 4, 1>; break;
        case 8: fn = resize_down_kernel<8, 2, 1, 1>; break;
        default: fn = resize_down_kernel<16, 2, 1, 1>; break;
    }
    if (narrow) fn = S == 2 ? resize_down_kernel<2, 4, 2, 3, true> : resize_down_kernel<4, 2, 1, 3, true>;
    if (dry) return true;                                   // (pbr_resize_form: the shape is this kernel's)
    hipLaunchKernelGGL(fn, dim3((unsigned)n_groups), dim3(64), 0, s, large, small, h_small, w_small, (int)groups_x, (int)bands, (int)band_rows, mapped, taps);
    return true;
}

// Launch of resize_stream_kernel (resize_stream.hpp): antialiased down-scales by any factor 1.01 <= s < 17 on both axes.  False when the shape is
// not the kernel's (the caller goes on to the strip form).  `workspace` holds the tables: pbr_resize_workspace_bytes is planes x h_in x w_out floats,
// the tables need 5 h_in + 2 h_out + (kt + 2) w_out.
static bool launch_stream(const float *src, float *dst, int64_t planes, int h_in, int w_in, int h_out, int w_out, const AxisFilter &fw, const AxisFilter &fh,
                          float *workspace, hipStream_t s, bool dry = false) {
    if (fw.scale < 1.01f || fh.scale < 1.01f || fw.support != fw.scale || fh.support != fh.scale) return false;      // (support = scale: antialiased)
    if ((int)(2.0f * fw.support) + 3 > 36 || (int)(2.0f * fh.support) + 3 > 36 || w_in % 4 != 0 || w_out < 16 || h_out < 4) return false;      // (the strip form's range of taps)
    if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(workspace)) & 15u) != 0) return false;
    const int kt = (int)(2.0f * fw.support) + 2, kw = (kt + 3) & ~3;    // rows of the column table (a window holds at most ceil(2 s) taps: hi - lo < 2 s + 1); taps the width pass walks: whole groups of four
    const size_t words = 4 * (size_t)h_in + (size_t)h_in + 2 * (size_t)h_out + (size_t)(kt + 2) * (size_t)w_out;
    if (words > (size_t)planes * (size_t)h_in * (size_t)w_out) return false;
    // One 16-byte piece of a row per lane (a strip spans 256 input columns), four rows in flight, ~2 048 workgroups, whatever the factor and the number of
    // planes (tools/resize_stream_probe.py on 3 | 8 x 4096^2 -> 3000^2 ... 300^2, every repetition on freshly allocated buffers, with boost clocks AND after 150 ms
    // of launches: two pieces per lane level to 20 % slower; 1 024 | 1 536 | 3 072 | 4 096 workgroups 20 | 3 | 2-20 | 5-20 % slower; eight rows in flight 3-6 % slower;
    // 6 144 ... 24 576 short bands dispatched in memory order: 5-25 % slower from 2 x up).
    constexpr int P = 1, D = 4;
    const bool nt = (int64_t)planes * h_in * w_in * 4 > (256ll << 20);      // the input does not fit the memory-side cache: it streams (8 x 4096^2: 110 -> 101 us at 10 x)
    // output columns per strip: as many as keep every strip's window (its start aligned down to 16 bytes) within 256 P input columns
    auto window_fits = [&](int oc) {
        for (int xb = 0; xb < w_out; xb += oc) {
            const int last = (xb + oc < w_out ? xb + oc : w_out) - 1;
            int lo, n, lo2, n2; float c;
            tap_window(fw, xb, lo, n, c);
            tap_window(fw, last, lo2, n2, c);
            if (lo2 + n2 - (lo & ~3) > 256 * P) return false;
        }
        return true;
    };
    int oc = (int)(((float)(256 * P - 5) - 2.0f * fw.support) / fw.scale);
    if (oc > w_out) oc = w_out;
    while (oc >= 8 && !window_fits(oc)) --oc;
    if (oc < 8) return false;
    const int64_t strips = (w_out + oc - 1) / oc;
    int64_t bands = 2048 / (planes * strips);
    bands = bands < 1 ? 1 : bands;
    int64_t band_rows = (h_out + bands - 1) / bands;
    band_rows = band_rows < 4 ? 4 : band_rows;
    bands = (h_out + band_rows - 1) / band_rows;
    const int64_t pairs = planes * bands, n_groups = pairs * strips;
    if (n_groups > INT32_MAX) return false;
    const size_t lds = sizeof(float) * ((size_t)2 * (4 / P) * (256 * P + 40) + (size_t)(kw + 2) * oc + 8);      // two buffers of finished rows | the strip's column table | two notes
    if (lds > 32 * 1024) return false;
    if (dry) return true;
    float4 *rec = reinterpret_cast<float4 *>(workspace);
    int *orow = reinterpret_cast<int *>(rec + h_in), *ylo = orow + h_in, *yhi = ylo + h_out, *xlo = yhi + h_out, *xn = xlo + w_out;
    float *wx = reinterpret_cast<float *>(xn + w_out);
    const int groups_y = (h_in + 255) / 256, groups_x = (w_out + 255) / 256;
    hipLaunchKernelGGL(resize_stream_tables_kernel, dim3(groups_y + groups_x), dim3(256), 0, s, rec, orow, ylo, yhi, xlo, xn, wx, kt, h_out, w_out, fh, fw, groups_y);
    const StreamGeom g = {h_out, w_out, h_in, w_in, kw, kt, oc, (int)strips, (int)bands, (int)band_rows, (uint32_t)((pairs / 8) * 8 * strips)};
    auto fn = nt ? resize_stream_kernel<P, D, true> : resize_stream_kernel<P, D, false>;
    hipLaunchKernelGGL(fn, dim3((unsigned)n_groups), dim3(128), lds, s, src, dst, g, rec, orow, ylo, yhi, xlo, xn, wx);
    return true;
}

static inline unsigned stream_grid(int64_t items) {
    const int64_t blocks = (items + 255) / 256, cap = 256 * 16;
    return (unsigned)(blocks < 1 ? 1 : (blocks > cap ? cap : blocks));
}

}  // namespace pbr

extern "C" {

size_t pbr_resize_workspace_bytes(int64_t planes, int32_t h_in, int32_t w_out) {
    return planes < 1 || h_in < 1 || w_out < 1 ? 0 : (size_t)planes * (size_t)h_in * (size_t)w_out * sizeof(float);
}

// pbr_resize_bilinear and pbr_resize_form: `form` receives the kernel family; `dry` = decide, launch nothing
static int resize_forward(const void *src, void *dst, int64_t planes, int32_t h_in, int32_t w_in, int32_t h_out,
                          int32_t w_out, int antialias, void *workspace, void *stream, bool dry, int *form) {
    using namespace pbr;
    if (!src || !dst || !workspace) return PBR_ERR_NULL_MAP;
    if (planes < 1 || h_in < 1 || w_in < 1 || h_out < 1 || w_out < 1) return PBR_ERR_SHAPE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    float *tmp = static_cast<float *>(workspace);
    const AxisFilter fw = make_filter(w_in, w_out, antialias != 0), fh = make_filter(h_in, h_out, antialias != 0);
    if (g_resize_up2 && fw.scale <= 1.0f && fh.scale <= 1.0f && w_in >= 6) {
        // up-scaling (or 1:1) on both axes: the two-tap register form.  3 x 4096^2 -> 6144^2 (resize_sweep.py (a probe of its round, removed with its knob: git 9ce0718:tools/), round 3).
        // Output rows per lane.  Measured (resize_up_ab.py (a probe of its round, removed with its knob: git 9ce0718:tools/), 3 planes, us, rows 2 / 4 / 8; strip kernel for scale):
        //   4096^2 -> 6144^2  131.4 / 121.6 / 110.8 (157)    -> 8192^2  215.3 / 195.5 / 184.7 (270)    2048^2 -> 4096^2  59.0 / 53.6 / 53.0 (64)
        //   4096^2 -> 4608^2   79.1 /  70.0 /  73.1 (101)    -> 4096^2   63.7 /  61.2 /  66.0 (95)
        // 8 from 1.25x up (the column taps' share shrinks as the rows grow), 4 below.
        const int rows = fh.scale <= 0.8f ? 8 : 4;
        const int64_t groups_x = (w_out + 255) / 256, groups_y = (h_out + rows - 1) / rows, n_groups = groups_x * groups_y * planes;
        if (n_groups <= INT32_MAX) {
            const uint32_t span = 8u << kUpRunLog2;
            const uint32_t xcd_groups = (uint32_t)(n_groups / span) * span;
            auto fn = rows == 8 ? resize_up2_kernel<8> : resize_up2_kernel<4>;
            *form = PBR_RESIZE_TWO_TAP;
            if (dry) return PBR_OK;
            hipLaunchKernelGGL(fn, dim3((unsigned)n_groups), dim3(64), 0, s, static_cast<const float *>(src), static_cast<float *>(dst),
                               (int)h_in, (int)w_in, (int)h_out, (int)w_out, (int)groups_x, (int)groups_y, xcd_groups, fw, fh);
            const hipError_t e = hipGetLastError();
            return e == hipSuccess ? PBR_OK : 1000 + (int)e;
        }
    }
    if (g_resize_up2 && antialias && h_in % h_out == 0 && w_in % w_out == 0 && h_in / h_out == w_in / w_out && h_in / h_out >= 2 && h_in / h_out <= 16 &&
        launch_down(static_cast<const float *>(src), static_cast<float *>(dst), planes, h_out, w_out, h_in / h_out, down_taps(h_in / h_out), s, dry)) {
        // a whole factor 2 ... 8 | 16 on both axes: the register form (resize_down.hpp)
        *form = PBR_RESIZE_BAND_WALK;
        if (dry) return PBR_OK;
        const hipError_t e = hipGetLastError();
        return e == hipSuccess ? PBR_OK : 1000 + (int)e;
    }
    // Antialiased down-scales from 7 x up that are not a whole factor (17 ... 36 taps per axis, (int)(2 s) + 3: the strip form's WIDE instantiation): every input row once
    // (resize_stream.hpp).  tools/resize_stream_probe.py, us, walk | strip, after 150 ms of launches (settled clocks), every repetition on freshly allocated buffers:
    //   8 x 4096^2 -> 400^2  100 | 126     -> 300^2  97 | 128     3 x 4096^2 -> 400^2  35-40 | 44     -> 300^2  42 | 60
    // The walk is built and bit-identical for every factor from 1.01 x (knob value 2 takes it wherever the shape allows) but NOT the rule below 7 x, where
    // its STORES decide: the walking wave alone streams 8 x 4096^2 at its read-only pattern (87-95 us) at every factor and clock, the width pass's arithmetic is free from
    // 2 x up, and the result's stores cost 0.45-0.9 us per MB where the strip form pays 0.26 (not understood: DESIGN.md section 9).  At boost clocks (the first ~20 launches
    // after an idle moment) 8 x 4096^2 -> 2000^2 | 1365^2 | 1000^2 read 120 | 107 | 102 against the strip form's 140 | 119 | 108; after 150 ms of launches 139 | 125-132 | 107
    // against 140 | 117-122 | 108, and below 2 x 187-243 against 154-174 -- on three boxes of four: on the fourth the stores drain fast enough for the walk to win from 1.5 x up
    // (106 against 117 at 1365^2, equally settled); the strip form's figures do not move.  A cache-resident input (3 planes) between 2.2 x and 7 x is 2-6 % faster
    // through the strip form.  Every step: profiles/EXPERIMENTS.md.
    const bool walk = (int)(2.0f * fw.support) + 3 > 16 || (int)(2.0f * fh.support) + 3 > 16;
    if (g_resize_up2 && antialias && (walk || g_resize_up2 == 2) &&
        launch_stream(static_cast<const float *>(src), static_cast<float *>(dst), planes, h_in, w_in, h_out, w_out, fw, fh, tmp, s, dry)) {
        *form = PBR_RESIZE_ROW_WALK;
        if (dry) return PBR_OK;
        const hipError_t e = hipGetLastError();
        return e == hipSuccess ? PBR_OK : 1000 + (int)e;
    }
    {   // strip form: tap tables + the height-reduced strip [toh][pitch] of a toh x 64 output tile in LDS, up to 36 taps per axis
        const int kx = (int)(2.0f * fw.support) + 3, ky = (int)(2.0f * fh.support) + 3;      // taps per output: xsize <= 2 support + 2
        const bool vec_ok = w_in % 4 == 0 && (reinterpret_cast<uintptr_t>(src) & 15u) == 0;
        if (kx <= 36 && ky <= 36) {                                                           // (round 5: 16 -> 36 taps, i.e. down-scales below 17x keep the one-kernel form)
            const int cols_max = (int)(kTileW * fw.scale + 2.0f * fw.support) + 4 + 3;       // + 3: window start aligned down to 16 bytes
            const int pitch = ((cols_max + 3) & ~3) + 4;                                      // + 4 floats: rows land on different banks
            auto lds_for = [&](int rows) {
                return sizeof(float) * ((size_t)kx * kTileW + (size_t)ky * rows + 2 * (kTileW + rows) + (size_t)rows * pitch + 40);      // slack: a window's taps are read up to the template's K
            };
            // Output rows per workgroup: as many as keep the workgroup's LDS within 24 KiB (6 workgroups per CU).  More rows
            // amortise the tables and re-read fewer input rows; more resident workgroups overlap the phases
            // (resize_sweep.py (a probe of its round, removed with its knob: git 9ce0718:tools/): 2x down 32-64 rows, 3-4x down 16, up-scales 64-128).
            int toh = 8;
            for (int rows : {128, 64, 32, 16})
                if (lds_for(rows) <= 24 * 1024) { toh = rows; break; }
            const size_t lds = lds_for(toh);
            const int64_t tx = (w_out + kTileW - 1) / kTileW, tyy = (h_out + toh - 1) / toh;
            if (lds <= 64 * 1024 && planes * tx * tyy <= INT32_MAX) {
                const int64_t n_tiles = planes * tx * tyy;
                // XCD-contiguous order (resize_strip_kernel) in chunks of 64 tiles.  Identity order -> chunks of 64, 3 x 4096^2 (us):
                // -> 2048^2 55.1 -> 48.0, -> 1024^2 44.4 -> 36.8, -> 1365^2 46.3 -> 41.0, -> 3000^2 74.7 -> 72.7, -> 5000^2 127.6 -> 114.3,
                // -> 6144^2 163.5 -> 158.1, -> 8192^2 269.1 -> 271.5; HBM reads 301.5 -> 201.7 MB for the 2x down-scale (the input is
                // 201.3 MB), 399.7 -> 201.7 MB for the 1.5x up-scale.  One chunk per XCD (an eighth of all tiles each) is as good for
                // down-scales but 4 % slower for large up-scales (eight write fronts far apart); 32 ... 1024 tiles are within 2 %.
                int64_t chunk = 64;
                if (chunk > n_tiles / 8) chunk = n_tiles / 8;
                const int quads = 0;      // 16-byte stores in the forward width pass: 62.7 against 54.0 us with them (4096^2 -> 2048^2): never
                const StripGeom tg = {toh, (int)tx, (int)tyy, kx, ky, pitch, vec_ok ? 1 : 0, (int)chunk, (int)(chunk ? (n_tiles / (8 * chunk)) * 8 * chunk : 0), quads};
                auto strip = kx <= 16 && ky <= 16 ? resize_strip_kernel<false, false> : resize_strip_kernel<false, false, true>;
                *form = PBR_RESIZE_STRIP;
                if (dry) return PBR_OK;
                hipLaunchKernelGGL(strip, dim3((unsigned)(planes * tx * tyy)), dim3(256), lds, s,
                                   static_cast<const float *>(src), static_cast<float *>(dst), (int)h_out, (int)w_out, (int)w_in, tg, fw, fh, StripTables{});
                const hipError_t e = hipGetLastError();
                return e == hipSuccess ? PBR_OK : 1000 + (int)e;
            }
        }
    }
    // more than 36 taps per axis (down-scales from 17x): two passes through `workspace`
    *form = PBR_RESIZE_TWO_PASS;
    if (dry) return PBR_OK;
    hipLaunchKernelGGL(resize_width_kernel, dim3(stream_grid(planes * h_in * w_out)), dim3(256), 0, s,
                       static_cast<const float *>(src), tmp, planes * h_in, (int)w_out, fw);
    hipLaunchKernelGGL(resize_height_kernel, dim3(stream_grid(planes * h_out * w_out)), dim3(256), 0, s,
                       tmp, static_cast<float *>(dst), planes, (int)h_out, (int)w_out, fh);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? PBR_OK : 1000 + (int)e;
}

int pbr_resize_bilinear(const void *src, void *dst, int64_t planes, int32_t h_in, int32_t w_in, int32_t h_out,
                        int32_t w_out, int antialias, void *workspace, void *stream) {
    int form = 0;
    return resize_forward(src, dst, planes, h_in, w_in, h_out, w_out, antialias, workspace, stream, false, &form);
}

int pbr_resize_form(const void *src, const void *dst, int64_t planes, int32_t h_in, int32_t w_in, int32_t h_out,
                    int32_t w_out, int antialias, const void *workspace) {
    int form = -1;
    const int rc = resize_forward(src, const_cast<void *>(dst), planes, h_in, w_in, h_out, w_out, antialias, const_cast<void *>(workspace), nullptr, true, &form);
    return rc == PBR_OK ? form : -1;
}

// workspace layout (floats): tmp [planes][h_in][w_out] | inv_y [h_out] | inv_x [w_out] | wy [kBwdMaxTaps][h_in] | wx [kBwdMaxTaps][w_in] |
// then ints: lo_y, cnt_y [h_in] | lo_x, cnt_x [w_in]
size_t pbr_resize_backward_workspace_bytes(int64_t planes, int32_t h_in, int32_t w_in, int32_t h_out, int32_t w_out) {
    if (planes < 1 || h_in < 1 || w_in < 1 || h_out < 1 || w_out < 1) return 0;
    const size_t words = (size_t)planes * (size_t)h_in * (size_t)w_out + (size_t)h_out + (size_t)w_out +
                         (size_t)pbr::kBwdMaxTaps * ((size_t)h_in + (size_t)w_in) + 2 * ((size_t)h_in + (size_t)w_in) + 4;
    return words * sizeof(float);
}

int pbr_resize_bilinear_backward(const void *grad_out, void *grad_in, int64_t planes, int32_t h_in, int32_t w_in, int32_t h_out,
                                 int32_t w_out, int antialias, void *workspace, void *stream) {
    using namespace pbr;
    if (!grad_out || !grad_in || !workspace) return PBR_ERR_NULL_MAP;
    if (planes < 1 || h_in < 1 || w_in < 1 || h_out < 1 || w_out < 1) return PBR_ERR_SHAPE;
    if (planes * h_in > INT32_MAX) return PBR_ERR_SHAPE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const AxisFilter fw = make_filter(w_in, w_out, antialias != 0), fh = make_filter(h_in, h_out, antialias != 0);
    float *tmp = static_cast<float *>(workspace);                    // [planes][h_in][w_out]
    float *inv_y = tmp + (size_t)planes * h_in * w_out, *inv_x = inv_y + h_out;
    float *wy = inv_x + w_out, *wx = wy + (size_t)kBwdMaxTaps * h_in;
    int *lo_y = reinterpret_cast<int *>(wx + (size_t)kBwdMaxTaps * w_in), *cnt_y = lo_y + h_in, *lo_x = cnt_y + h_in, *cnt_x = lo_x + w_in;
    const auto g = static_cast<const float *>(grad_out);
    float *gi = static_cast<float *>(grad_in);
    const int up = h_out % h_in == 0 && w_out % w_in == 0 && h_out / h_in == w_out / w_in ? h_out / h_in : 0;
    if (g_resize_up2 && (up == 2 || up == 4 || up == 8 || up == 16) && launch_down(g, gi, planes, h_in, w_in, up, up_transpose_taps(up), s)) {
        // gradient of an up-scale by 2 | 4 | 8: the band walk of resize_down.hpp over the upstream gradient, with the transposed two-tap weights.  (Powers of two only:
        // the forward's scale 1 / S is then exact and its two weights are the same for every S-th output; with 1/3, 1/5 ... the forward's fp32 tap positions drift
        // by ~6e-8 of the index, and the exact transpose of THAT is what the two-tap transpose below forms.)
        // 3 x 4096^2 upstream -> 2048^2: see DESIGN.md section 3 (the two-tap transpose below: 44.4 us, 1.17 x the bytes -- its lanes' windows overlap past L2)
        const hipError_t e = hipGetLastError();
        return e == hipSuccess ? PBR_OK : 1000 + (int)e;
    }
    if (g_resize_up2 && fw.scale <= 1.0f && fh.scale <= 1.0f && fw.scale >= 0.34f && fh.scale >= 0.25f && w_out >= 16) {
        // gradient of an up-scale (up to 3x across, 4x down the rows): the register-only transpose of the two-tap forward (round 4;
        // resize_bwd_probe.py (a probe of its round, removed with its knob: git 9ce0718:tools/)).  W from the exact window count of THIS shape; rows per lane 4.
        const int need = up2_backward_window(fw, w_out);
        constexpr int R = 4;
        const int64_t groups_x = (w_in + 255) / 256, groups_y = (h_in + R - 1) / R, n_groups = groups_x * groups_y * planes;
        if (need <= 16 && n_groups <= INT32_MAX) {
            const uint32_t span = 8u << kUpRunLog2;
            const uint32_t xcd_groups = (uint32_t)(n_groups / span) * span;
            auto fn = need <= 8 ? resize_up2_backward_kernel<8, R> : (need <= 12 ? resize_up2_backward_kernel<12, R> : resize_up2_backward_kernel<16, R>);
            hipLaunchKernelGGL(fn, dim3((unsigned)n_groups), dim3(64), 0, s, static_cast<const float *>(grad_out), static_cast<float *>(grad_in),
                               (int)h_in, (int)w_in, (int)h_out, (int)w_out, (int)groups_x, (int)groups_y, xcd_groups, fw, fh);
            const hipError_t e = hipGetLastError();
            return e == hipSuccess ? PBR_OK : 1000 + (int)e;
        }
    }
    auto fits = [](const AxisFilter &f) { return (int)((2.0f * f.support + 2.0f) / f.scale) + 2 <= kBwdMaxTaps; };
    const int64_t grid_rows = (int64_t)((w_out + 1023) / 1024) * planes * h_in;
    const int64_t grid_cols = (int64_t)((w_in + 255) / 256) * ((planes * h_in + kBwdRows - 1) / kBwdRows);
    if (fits(fw) && fits(fh) && grid_rows <= INT32_MAX && grid_cols <= INT32_MAX) {          // table-driven
        const int groups_y = (h_in + 255) / 256, groups_x = (w_in + 255) / 256;
        // Register-only gather over the tables (round 4, resize_backward_gather_kernel): 4 gradient columns x 8 rows per lane; the rows'
        // weights from the per-band matrices the tables kernel leaves in the (otherwise unused) pass-to-pass area of the workspace.
        const bool gather = w_out >= 16;
        const int need = gather ? gather_window(fw, w_out) : 0;
        constexpr int R = 8;                                          // gradient rows per lane
        const int64_t ggx = (w_in + 255) / 256, ggy = (h_in + R - 1) / R, n_groups = ggx * ggy * planes;
        const bool gather_ok = gather && need <= 16 && n_groups <= INT32_MAX;
        const int window = need <= 8 ? 8 : (need <= 12 ? 12 : 16);    // the gather kernel's W
        const size_t band_words = (size_t)((h_in + kBandRows - 1) / kBandRows) * kBandWords, col_groups = (size_t)(w_in + 3) / 4;
        const bool banded = gather_ok && R == kBandRows && band_window(fh, h_out, kBandRows) <= kBandMaxRows &&
                            band_words + col_groups * (1 + 4 * (size_t)window) <= (size_t)planes * h_in * w_out;
        float *band = banded ? tmp : nullptr, *col_w = banded ? tmp + band_words + col_groups : nullptr;
        int *col_base = banded ? reinterpret_cast<int *>(tmp + band_words) : nullptr;
        hipLaunchKernelGGL(resize_backward_tables_kernel, dim3(groups_y + groups_x), dim3(256), 0, s, lo_y, cnt_y, wy, (int)h_out, fh, lo_x, cnt_x, wx,
                           (int)w_out, fw, groups_y, band, col_base, col_w, window);
        if (gather_ok) {
            const uint32_t span = 8u << kUpRunLog2;
            const uint32_t xcd_groups = (uint32_t)(n_groups / span) * span;
            const StripTables tb = {lo_x, cnt_x, lo_y, cnt_y, wx, wy, (int)w_in, (int)h_in, (int)h_out, band, col_base, col_w};
            auto fn = banded ? (need <= 8 ? resize_backward_gather_kernel<8, 8, true> : (need <= 12 ? resize_backward_gather_kernel<12, 8, true> : resize_backward_gather_kernel<16, 8, true>))
                             : (need <= 8 ? resize_backward_gather_kernel<8, 8, false> : (need <= 12 ? resize_backward_gather_kernel<12, 8, false> : resize_backward_gather_kernel<16, 8, false>));
            hipLaunchKernelGGL(fn, dim3((unsigned)n_groups), dim3(64), 0, s, g, gi, (int)h_in, (int)w_in, (int)h_out, (int)w_out, (int)ggx, (int)ggy, xcd_groups, tb);
            const hipError_t e = hipGetLastError();
            return e == hipSuccess ? PBR_OK : 1000 + (int)e;
        }
        // One pass: the strip kernel with the transposed tables (resize_strip_kernel<true>): a toh x 64 tile of the gradient, the
        // rows pass from global memory into the LDS strip, the columns pass out of it.  3 x 2048^2 gradient -> 4096^2: see DESIGN.md 3.8.
        const int kx = (int)((2.0f * fw.support + 2.0f) / fw.scale) + 2, ky = (int)((2.0f * fh.support + 2.0f) / fh.scale) + 2;     // <= kBwdMaxTaps
        const bool vec_ok = w_out % 4 == 0 && (reinterpret_cast<uintptr_t>(grad_out) & 15u) == 0;
        const int cols_max = (int)((float)(kTileW - 1 + 2.0f * fw.support) / fw.scale) + 8;     // upstream columns a tile of 64 reads, + alignment
        const int pitch = ((cols_max + 3) & ~3) + 4;
        auto lds_for = [&](int rows) {
            return sizeof(float) * ((size_t)kx * kTileW + (size_t)ky * rows + 2 * (kTileW + rows) + (size_t)rows * pitch + 16);
        };
        // Rows per tile: here more rows win up to ~48 KiB of LDS (resize_bwd_probe.py (a probe of its round, removed with its knob: git 9ce0718:tools/), us at 32 / 64 / 128 rows: 2048^2 -> 4096^2
        // 123 / 79 / 67, 3000^2 -> 4096^2 138 / 98 / 84, 6144^2 -> 4096^2 184 / 150 / -, 4096^2 -> 2048^2 59 / 60 / 115): the strip's
        // halo rows are re-read per tile, and a gradient tile reads few bytes for what it writes.
        int toh = 8;
        for (int rows : {128, 64, 32, 16})
            if (lds_for(rows) <= 48 * 1024) { toh = rows; break; }
        const size_t lds = lds_for(toh);
        const int64_t tx = (w_in + kTileW - 1) / kTileW, tyy = (h_in + toh - 1) / toh, n_tiles = planes * tx * tyy;
        if (lds <= 64 * 1024 && n_tiles <= INT32_MAX) {
            int64_t chunk = 64;
            if (chunk > n_tiles / 8) chunk = n_tiles / 8;
            // 16-byte stores where the gradient is at least twice its upstream (2048^2 -> 4096^2: 66.7 against 69.3 us; the other way,
            // 4096^2 -> 2048^2, 70.6 against 59.4: a quarter of the lanes then walk the LDS strip)
            const int quads = (int64_t)h_in * w_in >= 2 * (int64_t)h_out * w_out && w_in % 4 == 0 &&
                              (reinterpret_cast<uintptr_t>(grad_in) & 15u) == 0;
            const StripGeom tg = {toh, (int)tx, (int)tyy, kx, ky, pitch, vec_ok ? 1 : 0, (int)chunk, (int)(chunk ? (n_tiles / (8 * chunk)) * 8 * chunk : 0), quads};
            const StripTables tb = {lo_x, cnt_x, lo_y, cnt_y, wx, wy, (int)w_in, (int)h_in, (int)h_out, nullptr, nullptr, nullptr};
            auto strip = quads ? resize_strip_kernel<true, true> : resize_strip_kernel<true, false>;
            hipLaunchKernelGGL(strip, dim3((unsigned)n_tiles), dim3(256), lds, s, g, gi, (int)h_in, (int)w_in, (int)w_out, tg, fw, fh, tb);
        } else {                                                                              // two passes through the workspace
            hipLaunchKernelGGL(resize_backward_rows_table_kernel, dim3((unsigned)grid_rows), dim3(256), 0, s, g, tmp, lo_y, cnt_y, wy, (int)h_in, (int)h_out, (int)w_out);
            hipLaunchKernelGGL(resize_backward_cols_table_kernel, dim3((unsigned)grid_cols), dim3(256), 0, s, tmp, gi, lo_x, cnt_x, wx, planes * h_in, (int)w_in, (int)w_out);
        }
    } else {                                                                                  // many contributors per input: the generic passes
        hipLaunchKernelGGL(resize_norm_kernel, dim3((h_out + 255) / 256), dim3(256), 0, s, inv_y, (int)h_out, fh);
        hipLaunchKernelGGL(resize_norm_kernel, dim3((w_out + 255) / 256), dim3(256), 0, s, inv_x, (int)w_out, fw);
        hipLaunchKernelGGL(resize_backward_rows_kernel, dim3(stream_grid(planes * h_in * w_out)), dim3(256), 0, s, g, tmp, inv_y, planes, (int)h_out, (int)w_out, fh);
        hipLaunchKernelGGL(resize_backward_cols_kernel, dim3(stream_grid(planes * h_in * w_in)), dim3(256), 0, s, tmp, gi, inv_x, planes * h_in, (int)w_out, fw);
    }
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? PBR_OK : 1000 + (int)e;
}

}  // extern "C"
