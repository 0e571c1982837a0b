// resize_stream.hpp -- antialiased down-scale by ANY factor 1.01 <= s < 17 on both axes: every input row read ONCE (round 6; the rule from 7 x up: resize.hip).
//
// What MaterialBase.resize (/root/reference/pypbr/materials/base.py:490-504) does when the target is not a whole fraction of the
// texture (4096^2 -> 1365^2, -> 400^2, 2048^2 -> 1000^2 ...): torchvision's resize = F.interpolate(mode="bilinear", antialias=True), ATen's
// separable triangle filter (resize.hip's header has the rule).  resize_strip_kernel serves these shapes tile by tile: per OUTPUT row
// of a tile it loads the row's K input rows, so every input row crosses L2 -> CU about twice, a tile's halo rows (102 read for the 82
// owned at 10.24 x) leave HBM a second time (PMC: 1.15 x), and three barrier-separated phases share a workgroup's time.
//
// Here a workgroup owns a STRIP of output columns (the 256 P input columns their windows span, a 16-byte piece per lane and P) and a BAND
// of output rows.  Its first wave walks down the band's INPUT rows: row r is loaded once and added to the (at most three) output rows
// whose windows hold it -- ceil(2 + 1/s) = 3 windows overlap for a triangle of support s whose centres are s apart -- each output row's
// taps in ascending order, exactly the chain resize_strip_kernel forms (bit-identical results).  The three accumulators are a shift
// register: slot 0 is the oldest output row still open; when its last tap is in, the row goes to LDS and the slots move up.  The
// workgroup's second wave runs the width pass over the finished rows (per-lane columns, weights from an LDS copy of the strip's slice
// of the column table) and stores (why two waves: at the kernel).
// What a row contributes to which slot is WAVE-UNIFORM and the same for every strip and plane: resize_stream_tables_kernel writes one
// 16-byte record per input row (three normalised weights, the slots that START with this row, the number of output rows that END
// with it) into the workspace, and the walk reads it with one scalar load per row: the weights are scalar operands of the fma.
// The tables are formed with the strip kernel's own statements (tap_window / tap_weight, weights normalised by their sum).
#pragma once
#include <type_traits>

namespace pbr {

struct StreamGeom { int h_out, w_out, h_in, w_in, kx, kt, oc, strips, bands, band_rows; uint32_t mapped; };

// one launch: blocks [0, groups_y) the row tables (256 input rows each), the others the column tables
__global__ __launch_bounds__(256) void resize_stream_tables_kernel(float4 *__restrict__ rec, int *__restrict__ orow, int *__restrict__ ylo, int *__restrict__ yhi,
                                                                   int *__restrict__ xlo, int *__restrict__ xn, float *__restrict__ wx, int kx, int h_out, int w_out,
                                                                   AxisFilter fh, AxisFilter fw, int groups_y) {
    if ((int)blockIdx.x >= groups_y) {
        const int c = ((int)blockIdx.x - groups_y) * 256 + (int)threadIdx.x;
        if (c >= w_out) return;
        int xmin, n; float center, wsum = 0.0f;
        tap_window(fw, c, xmin, n, center);
        for (int j = 0; j < n; ++j) wsum += tap_weight(fw, j, xmin, center);
        const float inv = wsum != 0.0f ? 1.0f / wsum : 0.0f;
        for (int j = 0; j < kx; ++j) wx[(size_t)j * w_out + c] = j < n ? tap_weight(fw, j, xmin, center) * inv : 0.0f;
        xlo[c] = xmin; xn[c] = n < kx ? n : kx;
        return;
    }
    // The output rows whose windows meet this block's 256 input rows: their windows and 1 / (sum of their taps) once, in LDS (at most 256 / s + 4 of them);
    // an input row then looks its three slots up instead of summing three windows itself.
    constexpr int kOut = 264;
    __shared__ int lo_s[kOut], hi_s[kOut];
    __shared__ float cen_s[kOut], inv_s[kOut];
    auto end_of = [&](int i) { int ymin, n; float c; tap_window(fh, i, ymin, n, c); return ymin + n; };
    // the first output row whose window ends after r (ends are monotone in the output index): a closed-form estimate, then the exact rule
    auto first_open = [&](int r) {
        int o = (int)(((float)r - fh.support - 0.5f) * fh.invscale - 0.5f) - 2;
        o = o < 0 ? 0 : (o > h_out - 1 ? h_out - 1 : o);
        while (o > 0 && end_of(o - 1) > r) --o;
        while (o < h_out - 1 && end_of(o) <= r) ++o;
        return o;
    };
    const int rb = (int)blockIdx.x * 256, tid = (int)threadIdx.x;
    const int i0 = first_open(rb);                       // (uniform)
    for (int k = tid; k < kOut; k += 256) {
        const int i = i0 + k;
        int ymin = INT32_MAX, n = 0; float center = 0.0f, wsum = 0.0f;
        if (i < h_out) {
            tap_window(fh, i, ymin, n, center);
            if (ymin < rb + 256) {
                for (int j = 0; j < n; ++j) wsum += tap_weight(fh, j, ymin, center);
                if (i >= 0 && ymin + n > 0) { ylo[i] = ymin; yhi[i] = ymin + n; }      // (every output row lies in some block's range; neighbours write the same values)
            }
        }
        lo_s[k] = ymin; hi_s[k] = i < h_out ? ymin + n : INT32_MAX; cen_s[k] = center;
        inv_s[k] = wsum != 0.0f ? 1.0f / wsum : 0.0f;
    }
    __syncthreads();
    const int r = rb + tid;
    if (r >= fh.n_in) return;
    // the first output row whose window ends after r, counted from i0: the estimate again, then the exact rule on the LDS copy
    int k0 = (int)(((float)r - fh.support - 0.5f) * fh.invscale - 0.5f) - 2 - i0;
    k0 = k0 < 0 ? 0 : (k0 > kOut - 4 ? kOut - 4 : k0);
    while (k0 > 0 && hi_s[k0 - 1] > r) --k0;
    while (k0 < kOut - 4 && hi_s[k0] <= r) ++k0;
    float w[3] = {0.0f, 0.0f, 0.0f};
    int bits = 0, ends = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int ymin = lo_s[k0 + k], yend = hi_s[k0 + k];
        if (ymin <= r && r < yend && yend != INT32_MAX) {
            w[k] = tap_weight(fh, r - ymin, ymin, cen_s[k0 + k]) * inv_s[k0 + k];
            if (r == ymin) bits |= 1 << k;
        }
        if (ends == k && yend == r + 1) ++ends;
    }
    rec[r] = make_float4(w[0], w[1], w[2], __int_as_float(bits | (ends << 3)));
    orow[r] = i0 + k0;
}

// Why three slots are enough, and why the launcher need not look: with centres c_i = s (i + 1/2) and windows [(int)(c_i - s + 1/2), (int)(c_i + s + 1/2)),
// a fourth window open at a row needs c_o + 2 s - 1/2 < c_o + s + 1/2, i.e. s < 1; consecutive windows leave no gap for s >= 1; and only the LAST
// window's end is clipped to the axis for s > 1 (ends are distinct otherwise).  The launcher takes s >= 1.01 (the centres are floats: ~5e-4 at 4096).
typedef float stream_f4 __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ void stream_load16(stream_f4 &dst, const float *q) {      // issued, NOT waited for: the caller counts (resize_stream_kernel)
    if (NT) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(dst) : "v"(q) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(q) : "memory");
}
__device__ __forceinline__ void stream_pin(stream_f4 &x) { asm volatile("" : "+v"(x)); }      // uses of x stay behind the wait in front of this

// The walk.  P: 16-byte pieces of a row per lane (a strip spans 256 P input columns); D: rows in flight; NT: non-temporal loads (an input
// that streams past the memory-side cache: every 128-byte line is touched by one instruction here).
//
// A workgroup is TWO waves.  Wave 0 walks (loads, height pass, finished rows -> LDS); wave 1 runs the width pass and stores.  Apart they are because
// of the counter of outstanding memory operations (vmcnt): loads and stores retire through it in issue order, so a wave that stores between its loads
// either waits for its whole ring of rows after every width pass or has to know at each row how many stores are still in front of the row it wants
// (both measured: 28 of 126 us on 8 x 4096^2 -> 1365^2) -- and the width pass's LDS round trips would sit in the walk's way besides.  The walk's counter
// now holds its D rows of loads and nothing else: s_waitcnt vmcnt((D - 1) P) before every row, exact.  The waves meet at ONE s_barrier per B finished
// rows (two buffers: wave 1 works on one while wave 0 fills the other), written as bare instructions: __syncthreads() carries a fence that waits for
// the ring as well.
//
// The tables (resize_stream_tables_kernel wrote them; __restrict__ kernel arguments, so that the wave-uniform reads are SCALAR loads):
//   rec  [h_in]   {w slot 0, w slot 1, w slot 2, bits: slots that start with this row (0-2) | output rows that end with it << 3}
//   orow [h_in]   the output row in slot 0 at this input row (the first whose window ends after it)
//   ylo, yhi [h_out]  first input row of an output row's window, one past its last
//   xlo, xn  [w_out]  first input column of an output column's window, its taps;   wx [kt][w_out] normalised column weights (0 past a window)
template <int P, int D, bool NT>
__global__ __launch_bounds__(128) void resize_stream_kernel(const float *__restrict__ src, float *__restrict__ dst, const StreamGeom g,
                                                            const float4 *__restrict__ t_rec, const int *__restrict__ t_orow, const int *__restrict__ t_ylo,
                                                            const int *__restrict__ t_yhi, const int *__restrict__ t_xlo, const int *__restrict__ t_xn,
                                                            const float *__restrict__ t_wx) {
    typedef stream_f4 lf4;
    extern __shared__ float lds[];
    constexpr int B = 4 / P;                             // finished rows per turn of the width pass
    constexpr int kMid = 256 * P + 40;                   // a height-reduced row of the strip + slack (taps past a window are read, never used)
    float *wxs = lds + 2 * B * kMid;                     // lds: mid[2][B][kMid] | wxs[kx / 4][oc][4] | xo[oc] | xc[oc] | note[2][4]
    int *xo = reinterpret_cast<int *>(wxs + g.kx * g.oc), *xc = xo + g.oc, *note = xc + g.oc;
    // (plane, band) pairs are dealt to the XCDs round-robin, ALL strips of a pair to the same XCD (resize_down.hpp): neighbouring strips share the
    // input columns their edge windows overlap in, and walk down their rows side by side -- the second one finds them in its XCD's L2.
    const uint32_t wg = blockIdx.x;
    uint32_t pair, gx;
    if (wg < g.mapped) { const uint32_t k = wg >> 3; pair = (k / (uint32_t)g.strips) * 8u + (wg & 7u); gx = k % (uint32_t)g.strips; }
    else { pair = wg / (uint32_t)g.strips; gx = wg - pair * (uint32_t)g.strips; }
    const int plane = (int)(pair / (uint32_t)g.bands), y0 = (int)(pair - (uint32_t)plane * (uint32_t)g.bands) * g.band_rows;
    const int y1 = min(y0 + g.band_rows, g.h_out);
    const int xb = (int)gx * g.oc, oc = min(g.oc, g.w_out - xb);
    const int lane = (int)threadIdx.x & 63;
    const int xbase = t_xlo[xb] & ~3;                    // uniform: scalar loads
    // LDS instructions of one wave execute in order; the other wave sees them behind "all of mine are done" + the barrier
    auto meet = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : : : "memory"); };

    if (__builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6) != 0) {
        // ================= wave 1: the width pass.  A lane keeps an output column; its weights come four taps to a read and serve all B rows.
        int nmin = INT32_MAX;                            // the fewest taps of a column of this strip: the tap groups below it need no "inside the window?" test
        for (int c = lane; c < oc; c += 64) { const int n = t_xn[xb + c]; xo[c] = t_xlo[xb + c] - xbase; xc[c] = n; nmin = min(nmin, n); }
        for (int k = 32; k > 0; k >>= 1) nmin = min(nmin, __shfl_xor(nmin, k, 64));
        for (int j = 0; j < g.kx; ++j)
            for (int c = lane; c < oc; c += 64) wxs[(((j >> 2) * oc + c) << 2) + (j & 3)] = j < g.kt ? t_wx[(size_t)j * g.w_out + xb + c] : 0.0f;
        float *dp = dst + (int64_t)plane * g.h_out * g.w_out + xb;
        for (int buf = 0;; buf ^= 1) {
            meet();
            const int o_first = __builtin_amdgcn_readfirstlane(note[4 * buf]), rows = __builtin_amdgcn_readfirstlane(note[4 * buf + 1]);
            const int last = __builtin_amdgcn_readfirstlane(note[4 * buf + 2]);
            const float *mid = lds + buf * B * kMid;
            for (int c = lane; c < (rows ? oc : 0); c += 64) {
                const int off = xo[c], n = xc[c];
                float a[B];
#pragma unroll
                for (int b = 0; b < B; ++b) a[b] = 0.0f;
                const float *q = mid + off;
                const lf4 *wq = reinterpret_cast<const lf4 *>(wxs) + c;
                int j = 0;
                for (; j + 4 <= nmin; j += 4) {          // every column of the strip has these taps
                    const lf4 w = wq[(j >> 2) * oc];
                    float v[B][4];
#pragma unroll
                    for (int b = 0; b < B; ++b)
#pragma unroll
                        for (int u = 0; u < 4; ++u) v[b][u] = q[b * kMid + j + u];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int b = 0; b < B; ++b) a[b] = fmaf(w[u], v[b][u], a[b]);
                }
                for (; j < g.kx; j += 4) {               // kx is a multiple of 4 (weights 0 past the table's rows; the strip's slack past a window: read, never used)
                    const lf4 w = wq[(j >> 2) * oc];
                    float v[B][4];
#pragma unroll
                    for (int b = 0; b < B; ++b)
#pragma unroll
                        for (int u = 0; u < 4; ++u) v[b][u] = q[b * kMid + j + u];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int b = 0; b < B; ++b) a[b] = fmaf(w[u], j + u < n ? v[b][u] : 0.0f, a[b]);
                }
#pragma unroll
                for (int b = 0; b < B; ++b)
                    if (b < rows) __builtin_nontemporal_store(a[b], dp + (int64_t)(o_first + b) * g.w_out + c);
            }
            if (last) break;
        }
        return;
    }

    // ================= wave 0: the walk
    const int r0 = t_ylo[y0], r1 = t_yhi[y1 - 1];
    int o = t_orow[r0];
    const int x_end = min(g.w_in, t_xlo[xb + oc - 1] + t_xn[xb + oc - 1]);      // one past the strip's last input column
    // Every load is UNCONDITIONAL (the counter of outstanding loads then tells the wave which row has arrived): a lane past the strip's window
    // reads the window's last piece again, a row past the band's end the band's last row -- values no tap uses.
    const float *sp[P];
#pragma unroll
    for (int p = 0; p < P; ++p) sp[p] = src + (int64_t)plane * g.h_in * g.w_in + min(xbase + 4 * lane + 256 * p, (x_end - 1) & ~3);

    lf4 ring[D][P], acc[3][P];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int p = 0; p < P; ++p) acc[k][p] = lf4{0.0f, 0.0f, 0.0f, 0.0f};
    // The loads are written out (inline assembly) and waited for BY COUNT: the compiler's own bookkeeping gives up at the joins of this loop and drains
    // the ring with every turn (s_waitcnt vmcnt(7, 6, ... 0)).  D rows of P loads are outstanding whenever a row is consumed, the wanted one the oldest.
    auto load_row = [&](auto ic, int r) {
        constexpr int i = decltype(ic)::value;
        const int64_t at = (int64_t)min(r, r1 - 1) * g.w_in;
#pragma unroll
        for (int p = 0; p < P; ++p) stream_load16<NT>(ring[i][p], sp[p] + at);
    };
    auto arrived = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
        asm volatile("s_waitcnt vmcnt(%0)" : : "n"((D - 1) * P) : "memory");
#pragma unroll
        for (int p = 0; p < P; ++p) stream_pin(ring[i][p]);
    };
    static_for<0, D>([&](auto ic) { load_row(ic, r0 + decltype(ic)::value); });
    int pend = 0, o_first = 0, buf = 0;                  // finished rows in the buffer being filled: output rows o_first ... o_first + pend - 1
    auto hand_over = [&](int last) {
        if (lane == 0) { note[4 * buf] = o_first; note[4 * buf + 1] = pend; note[4 * buf + 2] = last; }
        meet();
        buf ^= 1;
        pend = 0;
    };

    for (int base = r0; base < r1; base += D) {
        float4 rec[D];
#pragma unroll
        for (int i = 0; i < D; ++i) rec[i] = t_rec[min(base + i, r1 - 1)];      // uniform: D scalar loads, one wait
        static_for<0, D>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const int r = base + i;
            if (r < r1) {
                const int bits = __float_as_int(rec[i].w);
                const float w0 = rec[i].x, w1 = rec[i].y, w2 = rec[i].z;
                if (bits & 7) {                          // a row's first tap starts from zero: what the slot held (the row that left it, rows at weight 0) does not matter
#pragma unroll
                    for (int p = 0; p < P; ++p) {
                        if (bits & 1) acc[0][p] = lf4{0.0f, 0.0f, 0.0f, 0.0f};
                        if (bits & 2) acc[1][p] = lf4{0.0f, 0.0f, 0.0f, 0.0f};
                        if (bits & 4) acc[2][p] = lf4{0.0f, 0.0f, 0.0f, 0.0f};
                    }
                }
                arrived(ic);
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    const lf4 v = ring[i][p];
                    acc[0][p] = lf4{fmaf(w0, v.x, acc[0][p].x), fmaf(w0, v.y, acc[0][p].y), fmaf(w0, v.z, acc[0][p].z), fmaf(w0, v.w, acc[0][p].w)};
                    acc[1][p] = lf4{fmaf(w1, v.x, acc[1][p].x), fmaf(w1, v.y, acc[1][p].y), fmaf(w1, v.z, acc[1][p].z), fmaf(w1, v.w, acc[1][p].w)};
                    acc[2][p] = lf4{fmaf(w2, v.x, acc[2][p].x), fmaf(w2, v.y, acc[2][p].y), fmaf(w2, v.z, acc[2][p].z), fmaf(w2, v.w, acc[2][p].w)};
                }
                load_row(ic, r + D);
                for (int e = bits >> 3; e > 0; --e) {
                    // ---- output row o is reduced down the rows: it waits in LDS for the width pass
                    if (o >= y0 && o < y1) {
                        if (pend == 0) o_first = o;
                        float *mid = lds + (buf * B + pend) * kMid;
#pragma unroll
                        for (int p = 0; p < P; ++p) *reinterpret_cast<lf4 *>(mid + 4 * lane + 256 * p) = acc[0][p];
                        if (++pend == B) hand_over(0);
                    }
#pragma unroll
                    for (int p = 0; p < P; ++p) { acc[0][p] = acc[1][p]; acc[1][p] = acc[2][p]; }
                    ++o;
                }
            }
        });
    }
    asm volatile("s_waitcnt vmcnt(0)" : : : "memory");      // (the ring's last loads: rows past the band)
    hand_over(1);
}

}  // namespace pbr
