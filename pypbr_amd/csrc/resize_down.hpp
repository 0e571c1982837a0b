// resize_down.hpp -- antialiased down-scale by a whole factor S = 2 ... 8 | 16 on both axes, registers only (round 5).
//
// What MaterialBase.resize (/root/reference/pypbr/materials/base.py:490-504) does to a 1024^2 / 2048^2 / 4096^2 texture on its way to
// 512^2: torchvision's resize = F.interpolate(mode="bilinear", antialias=True), i.e. ATen's separable triangle filter of support S
// (resize.hip's header has the rule).  With n_in = S n_out the tap pattern is the same for every output index: output i reads the
// K = 2 S inputs S i - floor(S/2) ... S i - floor(S/2) + 2 S - 1 with the triangle's weights (an odd S: the last one is 0); only the FIRST and the LAST
// output of an axis have a clipped window (the taps that fall outside are dropped; the rest are normalised by their own sum).  So there is nothing to
// look up: the three weight vectors (interior, first, last -- the same for both axes) are formed on the host with resize.hip's own
// tap_window / tap_weight arithmetic and travel as kernel arguments, i.e. in scalar registers.
//
// A lane owns C consecutive output columns of a BAND of output rows and walks down the band.  It streams the input rows that feed
// them: C S / 4 16-byte loads per row (its own C S input columns, contiguous), the height pass as fma into the accumulators of the one or two output rows a input row
// feeds (taps in ascending order, as resize_strip_kernel forms them: bit-identical results).  When an output row's height pass is
// complete, the floor(S/2) | ceil(S/2) height-reduced columns left | right of the lane's own come from the neighbouring lanes (one cross-lane move each;
// the wave's first and last lane load the neighbouring 16-byte piece themselves and run the height pass on it too), and the width pass
// is C K fma out of registers.  No tables, no LDS strip, no barriers: the strip kernel's three barrier-separated phases per tile kept
// it at 0.70-0.74 of HBM on these shapes with every byte read once.
#pragma once
#include <type_traits>

namespace pbr {

template <int I, int END, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < END) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, END>(f);
    }
}

struct DownTaps { float wi[32], wl[32], wr[32]; };       // K = 2 S normalised weights of an interior output, of output 0, of the last output; zero where the window is clipped

template <int S, int R, int C, int D, bool NT = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 4))) void resize_down_kernel(const float *__restrict__ src, float *__restrict__ dst, int h_out, int w_out,
                                                                                             int groups_x, int bands, int band_rows, uint32_t mapped, const DownTaps t) {
    constexpr int K = 2 * S, HL = S / 2, HR = S - HL, N = C * S;      // taps per axis, halo columns left / right, input columns a lane owns (C output columns)
    constexpr int E = (HR + 3) / 4, X = 4 * E;           // 16-byte pieces (columns) the wave's first / last lane loads beside its own: the halo no neighbour lane holds
    constexpr int CH = S * R;                            // input rows per turn of the loop: R output rows
    static_assert(N % 4 == 0 && HR <= N && K <= 32, "whole 16-byte pieces; the halo comes out of the neighbouring lane's own columns");
    static_assert(R % 2 == 0 && CH % (D + 1) == 0, "the accumulator pair and the ring of rows come round with every turn of the loop");
    typedef float lf4 __attribute__((ext_vector_type(4)));
    // (plane, band) pairs are dealt to the XCDs round-robin, ALL column strips of a pair to the same XCD: neighbouring strips share the 128-byte
    // lines their pieces' ends lie in, and walk down their rows side by side -- the second one finds the line in its XCD's L2.
    const uint32_t wg = blockIdx.x;
    uint32_t pair, gx;
    if (wg < mapped) { const uint32_t k = wg >> 3; pair = (k / (uint32_t)groups_x) * 8u + (wg & 7u); gx = k % (uint32_t)groups_x; }
    else { pair = wg / (uint32_t)groups_x; gx = wg - pair * (uint32_t)groups_x; }
    const int plane = (int)(pair / (uint32_t)bands), yb = (int)(pair - (uint32_t)plane * (uint32_t)bands) * band_rows;
    const int rows_here = min(band_rows, h_out - yb);    // >= 1: bands = ceil(h_out / band_rows)
    const int lane = (int)threadIdx.x;
    // lanes past the row's end stay in the wave (they work on column 0 and store nothing): values are read ACROSS lanes below
    const int x_raw = ((int)gx * 64 + lane) * C;
    const bool live = x_raw < w_out;
    const int x0 = live ? x_raw : 0;
    const int w_in = S * w_out, h_in = S * h_out;
    const bool at_left = x0 == 0, at_right = x0 + C >= w_out;
    const float *sp = src + (int64_t)plane * h_in * w_in + S * x0;
    // the wave's first / last lane: the E 16-byte pieces left / right of its own columns (inside the row)
    const bool extra = (lane == 0 && !at_left) || (lane == 63 && !at_right);
    const float *ep = sp + (lane == 0 ? -X : N);
    // column weights: the inner columns of a lane are interior outputs always; column 0 is output 0 in the row's first lane, column C - 1 the last output in its last
    float wc0[K], wcl[K];
#pragma unroll
    for (int j = 0; j < K; ++j) { wc0[j] = at_left ? t.wl[j] : (C == 1 && at_right ? t.wr[j] : t.wi[j]); wcl[j] = at_right ? t.wr[j] : t.wi[j]; }     // (C = 1: the lane's one column is both)

    float acc[2][N + X];                                 // the two output rows in flight: own columns, then the first / last lane's extra pieces
#pragma unroll
    for (int k = 0; k < N + X; ++k) acc[0][k] = acc[1][k] = 0.0f;
    float *dp = dst + (int64_t)plane * h_out * w_out + x0;
    // The band's input rows S yb - HL ... S (yb + rows_here) + HR - 1, `rel` counted from the first.  Row rel feeds output row rel / S (its taps
    // 0 .. S - 1) and the one before (taps S .. 2 S - 1), which is complete with the last of them.  Rows are loaded D rows ahead of their use into a
    // ring of D + 1 rows; every array index is a compile-time constant (static_for over one turn of the loop), so the arrays are registers.
    const int row0 = S * yb - HL, n_rows = S * rows_here + S;
    float ring[D + 1][N + X];
    auto load_row = [&](auto ic, int base) {             // relative row base + i into slot i mod (D + 1)
        constexpr int i = decltype(ic)::value;
        const int rel = base + i, yi = row0 + rel;
        float *v = ring[i % (D + 1)];
#pragma unroll
        for (int k = 0; k < N + X; ++k) v[k] = 0.0f;
        if (rel < n_rows && yi >= 0 && yi < h_in) {      // wave-uniform; rows outside the image are taps outside a clipped window (weight 0, value 0)
            const float *row = sp + (int64_t)yi * w_in;
#pragma unroll
            for (int q = 0; q < N / 4; ++q) {
                const lf4 a = NT ? __builtin_nontemporal_load(reinterpret_cast<const lf4 *>(row + 4 * q)) : *reinterpret_cast<const lf4 *>(row + 4 * q);      // cached: the two loads of a 32-byte run of a lane are two instructions
                v[4 * q] = a.x; v[4 * q + 1] = a.y; v[4 * q + 2] = a.z; v[4 * q + 3] = a.w;
            }
            if (extra) {
#pragma unroll
                for (int q = 0; q < E; ++q) {
                    const lf4 a = *reinterpret_cast<const lf4 *>(ep + (int64_t)yi * w_in + 4 * q);
                    v[N + 4 * q] = a.x; v[N + 4 * q + 1] = a.y; v[N + 4 * q + 2] = a.z; v[N + 4 * q + 3] = a.w;
                }
            }
        }
    };
    static_for<0, D>([&](auto ic) { load_row(ic, 0); });
    for (int base = 0; base < n_rows; base += CH) {
        static_for<0, CH>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            load_row(std::integral_constant<int, (i + D) % CH>{}, i + D < CH ? base : base + CH);      // slot (i + D) mod (D + 1), as CH is a multiple of D + 1
            __builtin_amdgcn_sched_barrier(0);
            const float *v = ring[i % (D + 1)];
            constexpr int r = i / S, j = i - S * r;      // the output row this input row opens or continues (tap j); the one before gets tap j + S
            const int y = yb + base / S + r;
            float *a_new = acc[r & 1], *a_old = acc[(r + 1) & 1];
            const float w_new = y == 0 ? t.wl[j] : (y == h_out - 1 ? t.wr[j] : t.wi[j]);
            const float w_old = y == 1 ? t.wl[j + S] : (y == h_out ? t.wr[j + S] : t.wi[j + S]);
#pragma unroll
            for (int k = 0; k < N + X; ++k) {
                a_new[k] = fmaf(w_new, v[k], j == 0 ? 0.0f : a_new[k]);
                a_old[k] = fmaf(w_old, v[k], a_old[k]);
                asm volatile("" : "+v"(a_new[k]), "+v"(a_old[k]));      // formed HERE: otherwise the chains sink to the row's end and all K input rows stay live
            }
            if constexpr (j == S - 1) {
                // ---- output row y - 1 is reduced down the rows: halo columns, width pass, store
                const float *a = a_old;
                float e[N + HL + HR];                    // height-reduced columns S x0 - HL ... S x0 + N + HR - 1
#pragma unroll
                for (int k = 0; k < N; ++k) e[HL + k] = a[k];
#pragma unroll
                for (int k = 0; k < HL; ++k) {
                    const float from_left = __shfl_up(a[N - HL + k], 1, 64);
                    e[k] = lane == 0 ? a[N + X - HL + k] : from_left;                         // the extra pieces are zero where the row starts
                }
#pragma unroll
                for (int k = 0; k < HR; ++k) {
                    const float from_right = __shfl_down(a[k], 1, 64);
                    e[HL + N + k] = lane == 63 ? a[N + k] : (at_right ? 0.0f : from_right);
                }
                float o[C];
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    float sum = 0.0f;
#pragma unroll
                    for (int jj = 0; jj < K; ++jj) sum = fmaf(c == 0 ? wc0[jj] : (c == C - 1 ? wcl[jj] : t.wi[jj]), e[S * c + jj], sum);
                    o[c] = sum;
                }
                if (live && y - 1 >= yb && y - 1 < yb + rows_here) {
                    float *q = dp + (int64_t)(y - 1) * w_out;
                    if constexpr (C == 1) {
                        __builtin_nontemporal_store(o[0], q);
                    } else {
                        typedef float lfc __attribute__((ext_vector_type(C)));
                        lfc out;
#pragma unroll
                        for (int c = 0; c < C; ++c) out[c] = o[c];
                        __builtin_nontemporal_store(out, reinterpret_cast<lfc *>(q));
                    }
                }
            }
        });
    }
}

}  // namespace pbr
