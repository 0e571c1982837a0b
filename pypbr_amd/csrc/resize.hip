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
// Schedule: ONE kernel (resize_tile_kernel below), the separable passes fused through LDS: a workgroup stages the raw
// input window of its output tile with 16-byte loads, runs the width pass LDS -> LDS and the height pass LDS -> output,
// with the tap weights normalised once per tile (as ATen does) instead of per output.  Against the two-pass form (kept
// for extreme down-scales whose windows do not fit) this saves the write and the re-read of the width-pass result:
// 4096^2 -> 2048^2, 3 planes, antialiased: 450 MB of HBM traffic -> 252 MB.  Measured (profiles/, rocprofv3 SQ / LDS
// counters): the kernel is bound by the LDS pipe (65-80 % busy: SQ_LDS_IDX_ACTIVE 145 k of 170-218 k cycles per CU, 13.5 cycles
// per LDS instruction, ~28 bytes per clock: dword reads two floats apart are 2-way bank conflicts), VALUs 40 %, not by HBM
// (3.1 TB/s of algorithmic bytes).  Tried on top and measured level or worse, hence not here: persistent workgroups with the
// next tile's window prefetched through registers (86 us against 81.5), unmasked tap loops for interior tiles (81.0).
// Phase elimination (4096^2 -> 2048^2, 81.5 us): without global loads 80.1, without the width pass 59.5, without the height
// pass 60.5, without both 46.2, tables + LDS writes only 25.2.
#include <hip/hip_runtime.h>

#include <climits>
#include <cstdint>

#include "../../include/pbr_hip.h"

namespace pbr {

struct AxisFilter {
    float scale, support, invscale;
    int n_in;
};

__device__ __forceinline__ void tap_window(const AxisFilter &f, int i, int &xmin, int &xsize, float &center) {
    center = f.scale * ((float)i + 0.5f);
    xmin = max(0, (int)(center - f.support + 0.5f));
    xsize = min(f.n_in, (int)(center + f.support + 0.5f)) - xmin;
}

__device__ __forceinline__ float tap_weight(const AxisFilter &f, int j, int xmin, float center) {
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

constexpr int kTileW = 64;

// Tile form (the default).  A workgroup owns a toh x 64 tile of the output and works in three phases through LDS:
//   0. tap tables: per output column / row of the tile its first tap, and its taps' weights NORMALISED once (ATen
//      normalises the weights, then accumulates sum w_j x_j: _compute_indices_weights_aa) -- the inner loops below are
//      pure fma streams, no weight arithmetic, no division;
//   1. the raw input window of the tile, [in_rows][in_cols], fetched with 16-byte loads, all of a lane's loads in
//      flight together (the previous form walked taps with dword loads, two rows in flight per wave: 2.8 TB/s and
//      VALU-bound on address arithmetic);
//   2. width pass raw -> mid[in_rows][64];   3. height pass mid -> output, coalesced stores.
// Both passes read LDS with lanes consecutive in x and touch exactly the taps ATen touches (a tap outside a window is never
// read: 0 x inf would be NaN).  LDS: wx[K][64] wy[K][toh] | xo[64] xn[64] yo[toh] yn[toh] | mid[rows][64] raw[rows][pitch].
struct TileGeom { int toh, tiles_x, tiles_y, kx, ky, rows_max, pitch, vec_ok, vec_out; };

// The two tap loops, unrolled to a compile-time bound K >= the largest tap count in the tile (taps past a lane's own count
// are loaded -- from LDS the kernel owns -- but replaced by 0 before use).
template <int K>
__device__ __forceinline__ void width_pass(const float *raw, float *mid, const float *wx, const int *xo, const int *xn, int pitch,
                                           int in_rows, int tid) {
    const int i = tid & (kTileW - 1), off = xo[i], n = xn[i];
    float w[K];
#pragma unroll
    for (int j = 0; j < K; ++j) w[j] = j < n ? wx[j * kTileW + i] : 0.0f;
    for (int r = tid / kTileW; r < in_rows; r += 2 * (256 / kTileW)) {       // a lane keeps its column; two rows per step for ILP
        const int r2 = r + 256 / kTileW;
        const bool second = r2 < in_rows;
        const float *q0 = raw + r * pitch + off, *q1 = raw + (second ? r2 : r) * pitch + off;
        float v0[K], v1[K];
#pragma unroll
        for (int j = 0; j < K; ++j) { v0[j] = j < n ? q0[j] : 0.0f; v1[j] = j < n ? q1[j] : 0.0f; }
        float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
        for (int j = 0; j < K; ++j) { a0 = fmaf(w[j], v0[j], a0); a1 = fmaf(w[j], v1[j], a1); }
        mid[r * kTileW + i] = a0;
        if (second) mid[r2 * kTileW + i] = a1;
    }
}

// Four consecutive output columns per lane: one ds_read_b128 per tap, one 16-byte store per output row (when the output
// row pitch and the tile allow; else column by column).
template <int K>
__device__ __forceinline__ void height_pass(const float *mid, float *dp, const float *wy, const int *yo, const int *yn, int toh, int oh,
                                            int ow, int oy0, int ox0, int w_out, bool vec_out, int tid) {
    if (vec_out) {
        for (int e = tid; e < oh * (kTileW / 4); e += 256) {
            const int o = e / (kTileW / 4), i = (e - o * (kTileW / 4)) * 4;
            if (i >= ow) continue;
            const float *q = mid + yo[o] * kTileW + i;
            const int n = yn[o];
            float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const float4 v = *reinterpret_cast<const float4 *>(q + j * kTileW);
                const float w = j < n ? wy[j * toh + o] : 0.0f;
                acc.x = fmaf(w, j < n ? v.x : 0.0f, acc.x); acc.y = fmaf(w, j < n ? v.y : 0.0f, acc.y);
                acc.z = fmaf(w, j < n ? v.z : 0.0f, acc.z); acc.w = fmaf(w, j < n ? v.w : 0.0f, acc.w);
            }
            *reinterpret_cast<float4 *>(dp + (int64_t)(oy0 + o) * w_out + ox0 + i) = acc;     // ow % 4 == 0 here
        }
        return;
    }
    for (int e = tid; e < oh * kTileW; e += 256) {
        const int o = e / kTileW, i = e - o * kTileW;
        if (i >= ow) continue;
        const float *q = mid + yo[o] * kTileW + i;
        const int n = yn[o];
        float acc = 0.0f;
#pragma unroll
        for (int j = 0; j < K; ++j) acc = fmaf(j < n ? wy[j * toh + o] : 0.0f, j < n ? q[j * kTileW] : 0.0f, acc);
        dp[(int64_t)(oy0 + o) * w_out + ox0 + i] = acc;
    }
}

__global__ __launch_bounds__(256) void resize_tile_kernel(const float *__restrict__ src, float *__restrict__ dst, int h_out,
                                                          int w_out, int w_in, TileGeom tg, AxisFilter fw, AxisFilter fh) {
    extern __shared__ float lds[];
    float *wx = lds, *wy = wx + tg.kx * kTileW;
    int *xo = reinterpret_cast<int *>(wy + tg.ky * tg.toh), *xn = xo + kTileW, *yo = xn + kTileW, *yn = yo + tg.toh;
    // mid before raw, 16 floats of slack behind raw: the unrolled tap loops may LOAD (never use) up to 16 taps past a window
    float *mid = reinterpret_cast<float *>(yn + tg.toh), *raw = mid + tg.rows_max * kTileW;
    __shared__ int tap_max[2];
    const int tile = blockIdx.x, per_plane = tg.tiles_x * tg.tiles_y;
    const int plane = tile / per_plane, t2 = tile - plane * per_plane;
    const int ty = t2 / tg.tiles_x, tx = t2 - ty * tg.tiles_x;
    const int ox0 = tx * kTileW, oy0 = ty * tg.toh;
    const int ow = min(kTileW, w_out - ox0), oh = min(tg.toh, h_out - oy0);
    const int tid = threadIdx.x;
    // window of the tile (tap windows are monotone in the output index)
    int xlo, ylo, n0, xl, nl, yl, nyl; float c0;
    tap_window(fw, ox0, xlo, n0, c0);
    tap_window(fw, ox0 + ow - 1, xl, nl, c0);
    tap_window(fh, oy0, ylo, n0, c0);
    tap_window(fh, oy0 + oh - 1, yl, nyl, c0);
    const int xbase = tg.vec_ok ? (xlo & ~3) : xlo;
    const int in_cols = xl + nl - xbase, in_rows = yl + nyl - ylo;
    // ---- phase 0: tap tables (wave 0: columns, wave 1: rows) and the largest tap count per axis
    if (tid < kTileW) {
        int xmin = 0, n = 0; float center = 0.0f, wsum = 0.0f;
        if (tid < ow) {
            tap_window(fw, ox0 + tid, xmin, n, center);
            for (int j = 0; j < n; ++j) wsum += tap_weight(fw, j, xmin, center);
        }
        const float inv = wsum != 0.0f ? 1.0f / wsum : 0.0f;
        for (int j = 0; j < tg.kx; ++j) wx[j * kTileW + tid] = j < n ? tap_weight(fw, j, xmin, center) * inv : 0.0f;
        xo[tid] = tid < ow ? xmin - xbase : 0;
        xn[tid] = n;
        int m = n;
        for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o, 64));
        if (tid == 0) tap_max[0] = m;
    } else if (tid < 2 * kTileW) {
        const int o = tid - kTileW;
        int ymin = 0, n = 0; float center = 0.0f, wsum = 0.0f;
        if (o < oh) {
            tap_window(fh, oy0 + o, ymin, n, center);
            for (int j = 0; j < n; ++j) wsum += tap_weight(fh, j, ymin, center);
        }
        const float inv = wsum != 0.0f ? 1.0f / wsum : 0.0f;
        if (o < tg.toh) {
            for (int j = 0; j < tg.ky; ++j) wy[j * tg.toh + o] = j < n ? tap_weight(fh, j, ymin, center) * inv : 0.0f;
            yo[o] = o < oh ? ymin - ylo : 0;
            yn[o] = n;
        }
        int m = n;
        for (int k = 32; k > 0; k >>= 1) m = max(m, __shfl_xor(m, k, 64));
        if (o == 0) tap_max[1] = m;
    }
    // ---- phase 1: raw window -> LDS
    const float *sp = src + ((int64_t)plane * fh.n_in + ylo) * w_in + xbase;
    if (tg.vec_ok) {
        // The window's 16-byte pieces dealt to the lanes in linear order (piece e = row e / c4n, column e % c4n: every lane
        // busy whatever the row length), ALL of a lane's loads -- up to 8 -- in flight before the first LDS store: one
        // memory round trip per workgroup instead of one per four rows.  The division is a float multiply: (e + 0.5) / c4n is
        // at least 0.5 / c4n away from an integer, far more than the rounding error for e < 2^16, c4n <= 256.
        const int c4n = (in_cols + 3) >> 2, total = in_rows * c4n;
        const float inv = 1.0f / (float)c4n;
        for (int e0 = tid; e0 < total; e0 += 8 * 256) {
            float4 v[8];
            int at[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = e0 + u * 256;
                const int r = (int)(((float)e + 0.5f) * inv), c = e - r * c4n;
                v[u] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                at[u] = e < total ? r * tg.pitch + 4 * c : -1;
                if (e < total && xbase + 4 * c < w_in) v[u] = *reinterpret_cast<const float4 *>(sp + (int64_t)r * w_in + 4 * c);   // w_in % 4 == 0
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (at[u] >= 0) *reinterpret_cast<float4 *>(raw + at[u]) = v[u];
        }
    } else {
        for (int c = tid & 63; c < in_cols; c += 64)
            for (int r = tid >> 6; r < in_rows; r += 4) raw[r * tg.pitch + c] = sp[(int64_t)r * w_in + c];
    }
    __syncthreads();
    const int kx = tap_max[0], ky = tap_max[1];
    // ---- phase 2: width pass raw -> mid
    if (kx <= 4) width_pass<4>(raw, mid, wx, xo, xn, tg.pitch, in_rows, tid);
    else if (kx <= 6) width_pass<6>(raw, mid, wx, xo, xn, tg.pitch, in_rows, tid);
    else if (kx <= 8) width_pass<8>(raw, mid, wx, xo, xn, tg.pitch, in_rows, tid);
    else if (kx <= 12) width_pass<12>(raw, mid, wx, xo, xn, tg.pitch, in_rows, tid);
    else width_pass<16>(raw, mid, wx, xo, xn, tg.pitch, in_rows, tid);
    __syncthreads();
    // ---- phase 3: height pass mid -> output
    float *dp = dst + (int64_t)plane * h_out * w_out;
    const bool vec_out = tg.vec_out && (ow & 3) == 0;
    if (ky <= 4) height_pass<4>(mid, dp, wy, yo, yn, tg.toh, oh, ow, oy0, ox0, w_out, vec_out, tid);
    else if (ky <= 6) height_pass<6>(mid, dp, wy, yo, yn, tg.toh, oh, ow, oy0, ox0, w_out, vec_out, tid);
    else if (ky <= 8) height_pass<8>(mid, dp, wy, yo, yn, tg.toh, oh, ow, oy0, ox0, w_out, vec_out, tid);
    else if (ky <= 12) height_pass<12>(mid, dp, wy, yo, yn, tg.toh, oh, ow, oy0, ox0, w_out, vec_out, tid);
    else height_pass<16>(mid, dp, wy, yo, yn, tg.toh, oh, ow, oy0, ox0, w_out, vec_out, tid);
}

static AxisFilter make_filter(int n_in, int n_out, bool antialias) {
    AxisFilter f;
    f.scale = (float)n_in / (float)n_out;            // area_pixel_compute_scale<float>, align_corners = False
    const bool aa = antialias && f.scale >= 1.0f;
    f.support = aa ? f.scale : 1.0f;                 // interp_size / 2 * scale, interp_size = 2
    f.invscale = aa ? 1.0f / f.scale : 1.0f;
    f.n_in = n_in;
    return f;
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

int pbr_resize_bilinear(const void *src, void *dst, int64_t planes, int32_t h_in, int32_t w_in, int32_t h_out,
                        int32_t w_out, int antialias, void *workspace, void *stream) {
    using namespace pbr;
    if (!src || !dst || !workspace) return PBR_ERR_NULL_MAP;
    if (planes < 1 || h_in < 1 || w_in < 1 || h_out < 1 || w_out < 1) return PBR_ERR_SHAPE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    float *tmp = static_cast<float *>(workspace);
    const AxisFilter fw = make_filter(w_in, w_out, antialias != 0), fh = make_filter(h_in, h_out, antialias != 0);
    // tile form: the raw window of a toh x 64 output tile, its width pass and the tap tables in at most 40 KiB of LDS (4
    // workgroups per CU; 64 KiB if no tile height fits that), for up to 16 taps per axis (scale <= 6.5)
    {
        const int kx = (int)(2.0f * fw.support) + 3, ky = (int)(2.0f * fh.support) + 3;      // taps per output: xsize <= 2 support + 2
        const bool vec_ok = w_in % 4 == 0 && (reinterpret_cast<uintptr_t>(src) & 15u) == 0;
        const bool vec_out = w_out % 4 == 0 && (reinterpret_cast<uintptr_t>(dst) & 15u) == 0;
        const bool fits = kx <= 16 && ky <= 16;
        for (size_t budget : {(size_t)40 * 1024, (size_t)64 * 1024}) {
            for (int toh : {32, 16, 8, 4}) {
                if (!fits) break;
                const int rows_max = (int)(toh * fh.scale + 2.0f * fh.support) + 4;
                const int cols_max = (int)(kTileW * fw.scale + 2.0f * fw.support) + 4 + 3;   // + 3: window start aligned down to 16 bytes
                const int pitch = ((cols_max + 3) & ~3) + 4;                                  // + 4 floats: rows land on different banks
                const size_t lds = sizeof(float) * ((size_t)kx * kTileW + (size_t)ky * toh + 2 * (kTileW + toh) + (size_t)rows_max * pitch +
                                                    (size_t)rows_max * kTileW + 16);
                const int64_t tx = (w_out + kTileW - 1) / kTileW, tyy = (h_out + toh - 1) / toh;
                if (lds > budget || planes * tx * tyy > INT32_MAX) continue;
                const TileGeom tg = {toh, (int)tx, (int)tyy, kx, ky, rows_max, pitch, vec_ok ? 1 : 0, vec_out ? 1 : 0};
                hipLaunchKernelGGL(resize_tile_kernel, dim3((unsigned)(planes * tx * tyy)), dim3(256), lds, s,
                                   static_cast<const float *>(src), static_cast<float *>(dst), (int)h_out, (int)w_out, (int)w_in, tg, fw, fh);
                const hipError_t e = hipGetLastError();
                return e == hipSuccess ? PBR_OK : 1000 + (int)e;
            }
        }
    }
    // windows too large for the tile form (more than 16 taps per axis, i.e. down-scales beyond ~6.5x): two passes through `workspace`
    hipLaunchKernelGGL(resize_width_kernel, dim3(stream_grid(planes * h_in * w_out)), dim3(256), 0, s,
                       static_cast<const float *>(src), tmp, planes * h_in, (int)w_out, fw);
    hipLaunchKernelGGL(resize_height_kernel, dim3(stream_grid(planes * h_out * w_out)), dim3(256), 0, s,
                       tmp, static_cast<float *>(dst), planes, (int)h_out, (int)w_out, fh);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? PBR_OK : 1000 + (int)e;
}

}  // extern "C"
