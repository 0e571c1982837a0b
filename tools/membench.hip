// membench.hip -- stand-alone ceiling probe for the Cook-Torrance access pattern on MI355X:
// 8 planar fp32 input planes -> 3 output planes, 16 B per lane per plane, optional synthetic
// VALU work per pixel.  Tells how much of the fused kernel's time is the memory pattern
// itself (no arithmetic) and how much VALU work that pattern hides.
//   hipcc -O3 --offload-arch=gfx950 tools/membench.hip -o tools/bin/membench && tools/bin/membench [S] [resize: the ceilings and the resize family's patterns only]
#include <hip/hip_runtime.h>
#include <cstring>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <bool NT> __device__ __forceinline__ f4 ld(const f4 *p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ void st(f4 *p, f4 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// FMAS: synthetic dependent-chain-free VALU work per pixel (transcendental every 10th op)
template <int FMAS> __device__ __forceinline__ f4 work(f4 a, f4 b, f4 c) {
    f4 r = a;
#pragma unroll
    for (int i = 0; i < FMAS; ++i) {
        if (i % 10 == 9) { r.x = __builtin_amdgcn_rcpf(r.x + 2.0f); r.y = __builtin_amdgcn_rcpf(r.y + 2.0f); r.z = __builtin_amdgcn_rcpf(r.z + 2.0f); r.w = __builtin_amdgcn_rcpf(r.w + 2.0f); }
        else r = r * b + c;
    }
    return r;
}

template <bool NT, int FMAS, int BLOCK, bool NTS = NT>
__global__ __launch_bounds__(BLOCK) void oneshot(const f4 *__restrict__ in, f4 *__restrict__ out, size_t nv, size_t plane, size_t oplane) {
    const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= nv) return;
    f4 v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = ld<NT>(in + c * plane + i);
    f4 r0 = work<FMAS>(v[0] + v[3] + v[6], v[6], v[7]);
    f4 r1 = work<FMAS>(v[1] + v[4] + v[7], v[6], v[7]);
    f4 r2 = work<FMAS>(v[2] + v[5], v[6], v[7]);
    st<NTS>(out + i, r0); st<NTS>(out + oplane + i, r1); st<NTS>(out + 2 * oplane + i, r2);
}

template <bool NT, int FMAS, int BLOCK>
__global__ __launch_bounds__(BLOCK) void persistent(const f4 *__restrict__ in, f4 *__restrict__ out, size_t nv, size_t plane) {
    const size_t stride = (size_t)gridDim.x * BLOCK;
    size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= nv) return;
    f4 v[8], w[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = ld<NT>(in + c * plane + i);
    while (true) {
        const size_t n = i + stride;
        if (n < nv) {
#pragma unroll
            for (int c = 0; c < 8; ++c) w[c] = ld<NT>(in + c * plane + n);
        }
        f4 r0 = work<FMAS>(v[0] + v[3] + v[6], v[6], v[7]);
        f4 r1 = work<FMAS>(v[1] + v[4] + v[7], v[6], v[7]);
        f4 r2 = work<FMAS>(v[2] + v[5], v[6], v[7]);
        st<NT>(out + i, r0); st<NT>(out + plane + i, r1); st<NT>(out + 2 * plane + i, r2);
        if (n >= nv) break;
        i = n;
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = w[c];
    }
}

// LDS-DMA variant: the 8 plane loads go global -> LDS (global_load_lds_dwordx4, no VGPR destination), the lane then
// reads its 16 bytes per plane back with ds_read_b128.  One-wave workgroups, 8 KiB of LDS each.
template <bool NT, int FMAS>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3)))
void oneshot_ldsdma(const f4 *__restrict__ in, f4 *__restrict__ out, size_t nv, size_t plane) {
    __shared__ f4 buf[8][64];
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    const size_t ic = i < nv ? i : nv - 1;
#pragma unroll
    for (int c = 0; c < 8; ++c)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(in + c * plane + ic),
                                         (__attribute__((address_space(3))) void *)&buf[c][0], 16, 0, NT ? 2 : 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    f4 v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = buf[c][threadIdx.x];
    if (i >= nv) return;
    f4 r0 = work<FMAS>(v[0] + v[3] + v[6], v[6], v[7]);
    f4 r1 = work<FMAS>(v[1] + v[4] + v[7], v[6], v[7]);
    f4 r2 = work<FMAS>(v[2] + v[5], v[6], v[7]);
    st<NT>(out + i, r0); st<NT>(out + plane + i, r1); st<NT>(out + 2 * plane + i, r2);
}

// the register path at the same geometry: one-wave workgroups, 3 waves per SIMD
template <bool NT, int FMAS>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3)))
void oneshot_w3(const f4 *__restrict__ in, f4 *__restrict__ out, size_t nv, size_t plane) {
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= nv) return;
    f4 v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = ld<NT>(in + c * plane + i);
    f4 r0 = work<FMAS>(v[0] + v[3] + v[6], v[6], v[7]);
    f4 r1 = work<FMAS>(v[1] + v[4] + v[7], v[6], v[7]);
    f4 r2 = work<FMAS>(v[2] + v[5], v[6], v[7]);
    st<NT>(out + i, r0); st<NT>(out + plane + i, r1); st<NT>(out + 2 * plane + i, r2);
}

// the BACKWARD kernel's pattern: 8 map planes + 3 upstream-gradient planes in, 8 gradient planes out (19 streams, 76 B per
// pixel), one-wave groups at WPE waves per SIMD, optional arithmetic
template <bool NT, int FMAS, int WPE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
void oneshot_bwd(const f4 *__restrict__ in, const f4 *__restrict__ gout, f4 *__restrict__ out, size_t nv, size_t plane) {
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= nv) return;
    f4 v[11];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = ld<NT>(in + c * plane + i);
#pragma unroll
    for (int c = 0; c < 3; ++c) v[8 + c] = ld<NT>(gout + c * plane + i);
    const f4 r = work<FMAS>(v[0] + v[3] + v[6] + v[8], v[9], v[10]);
#pragma unroll
    for (int c = 0; c < 8; ++c) st<NT>(out + c * plane + i, r + v[c]);
}

// the backward kernel's pattern with fp16 maps (44 B per pixel: 8 fp16 planes + 3 fp32 gradient planes in, 8 fp16 planes out).
// PX = pixels per lane: 2 = what the kernels do (4-byte map accesses, 8-byte gradient accesses), 8 = the same bytes through
// 16-byte accesses.  One-wave groups, WPE waves per SIMD.
template <int PX, int WPE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
void oneshot_bwd16(const unsigned *__restrict__ in, const float *__restrict__ gout, unsigned *__restrict__ out, size_t npx, size_t plane_px) {
    const size_t i = ((size_t)blockIdx.x * 64 + threadIdx.x) * PX;            // first pixel of the lane
    if (i >= npx) return;
    constexpr int W = PX / 2, G = PX;                                           // dwords per lane and plane: maps | gradient
    unsigned v[8][W]; float g[3][G];
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int k = 0; k < W; ++k) v[c][k] = __builtin_nontemporal_load(in + (c * plane_px + i) / 2 + k);
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int k = 0; k < G; ++k) g[c][k] = __builtin_nontemporal_load(gout + c * plane_px + i + k);
    unsigned mix = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int k = 0; k < G; ++k) mix ^= __float_as_uint(g[c][k]);
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int k = 0; k < W; ++k) __builtin_nontemporal_store(v[c][k] ^ mix, out + (c * plane_px + i) / 2 + k);
}

// the forward kernel's pattern with fp16 maps and an fp32 result (28 B per pixel), 8 pixels per lane: 16-byte loads, two
// 16-byte stores per result plane (each covering 16 of every 32 bytes -- the kernel swaps pieces through LDS to fill them)
template <int WPE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
void oneshot_fwd16(const f4 *__restrict__ in, f4 *__restrict__ out, size_t npx, size_t plane_px) {
    const size_t i = ((size_t)blockIdx.x * 64 + threadIdx.x) * 8;
    if (i >= npx) return;
    f4 v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = __builtin_nontemporal_load(in + (c * plane_px + i) / 8);
    const f4 r = v[0] + v[1] + v[2] + v[3] + v[4] + v[5] + v[6] + v[7];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        __builtin_nontemporal_store(r + v[c], out + (c * plane_px + i) / 4);
        __builtin_nontemporal_store(r - v[c], out + (c * plane_px + i) / 4 + 1);
    }
}

// the repeat-inner kernel's pattern (MaterialBase.tile(2) fused, ct_tiled.hip): 8 source planes of (S/2)^2 read once -- 16 bytes per
// lane and plane for fp32 maps, 8 for fp16 (HALF) --, 3 result planes of S^2 written: every lane's four pixels go to the four repeats
// (two across, two down).  One-wave workgroups, source order, stores non-temporal (the kernel's rule).  268 / 335 MB per launch at
// S = 4096, three quarters / three fifths of them writes.
template <bool HALF, int WPE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, 8)))
void oneshot_repeat(const float *__restrict__ in, f4 *__restrict__ out, int src, size_t src_plane) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    const size_t q = (size_t)blockIdx.x * 64 + threadIdx.x;                 // quad of the source
    const int qx = src / 4, y = (int)(q / qx), x = (int)(q - (size_t)y * qx) * 4;
    if (y >= src) return;
    f4 r = {0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        if (HALF) { const f2 v = *reinterpret_cast<const f2 *>(in + (c * src_plane + (size_t)y * src + x) / 2); r += f4{v.x, v.y, v.x, v.y}; }
        else r += *reinterpret_cast<const f4 *>(in + c * src_plane + (size_t)y * src + x);
    }
    const size_t W = 2 * (size_t)src, plane = W * W;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int ry = 0; ry < 2; ++ry)
#pragma unroll
            for (int rx = 0; rx < 2; ++rx)
                __builtin_nontemporal_store(r + (float)c, out + (c * plane + ((size_t)ry * src + y) * W + (size_t)rx * src + x) / 4);
}

// BASELINE config 3's layout: B materials as five [B][C][S^2] tensors (albedo 3, normal 3, roughness 1, metallic 1 -> result 3),
// 11 streams whose plane strides are all multiples of 16 MiB at S = 2048.  RUNS: XCD x takes runs of 64 consecutive tiles
// (tile_of_workgroup's remap); else the identity order.  One-wave groups, 3 waves per SIMD, as the fused kernel.
template <bool RUNS, int FMAS>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3)))
void batch_w3(const f4 *__restrict__ al, const f4 *__restrict__ nm, const f4 *__restrict__ ro, const f4 *__restrict__ me, f4 *__restrict__ out,
              unsigned tiles_per_mat, unsigned n_tiles, size_t plane) {
    unsigned tile = blockIdx.x;
    if (RUNS && tile < (n_tiles >> 9 << 9)) { const unsigned xcd = tile & 7u, slot = tile >> 3; tile = ((slot >> 6) << 9) + (xcd << 6) + (slot & 63u); }
    const unsigned b = tile / tiles_per_mat;
    const size_t i = (size_t)(tile - b * tiles_per_mat) * 64 + threadIdx.x;
    f4 v[8];
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] = ld<true>(al + ((size_t)b * 3 + c) * plane + i);
#pragma unroll
    for (int c = 0; c < 3; ++c) v[3 + c] = ld<true>(nm + ((size_t)b * 3 + c) * plane + i);
    v[6] = ld<true>(ro + (size_t)b * plane + i);
    v[7] = ld<true>(me + (size_t)b * plane + i);
    f4 r0 = work<FMAS>(v[0] + v[3] + v[6], v[6], v[7]);
    f4 r1 = work<FMAS>(v[1] + v[4] + v[7], v[6], v[7]);
    f4 r2 = work<FMAS>(v[2] + v[5], v[6], v[7]);
    st<true>(out + ((size_t)b * 3) * plane + i, r0); st<true>(out + ((size_t)b * 3 + 1) * plane + i, r1); st<true>(out + ((size_t)b * 3 + 2) * plane + i, r2);
}

// the fused blend + render pattern: two materials (8 planes each) and a mask in, 3 planes out (80 B per pixel, 20 streams)
template <int WPE, int FMAS>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
void oneshot_blend(const f4 *__restrict__ a, const f4 *__restrict__ b, const f4 *__restrict__ mask, f4 *__restrict__ out, size_t nv, size_t plane) {
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= nv) return;
    f4 v[17];
#pragma unroll
    for (int c = 0; c < 8; ++c) { v[c] = ld<true>(a + c * plane + i); v[8 + c] = ld<true>(b + c * plane + i); }
    v[16] = ld<true>(mask + i);
    f4 m0 = v[16], s0 = v[0] * m0 + v[8], s1 = v[1] * m0 + v[9], s2 = v[2] * m0 + v[10];
#pragma unroll
    for (int c = 3; c < 8; ++c) { s0 += v[c] * m0; s1 += v[8 + c]; s2 += v[c] - v[8 + c]; }
    st<true>(out + i, work<FMAS>(s0, s1, s2)); st<true>(out + plane + i, work<FMAS>(s1, s2, s0)); st<true>(out + 2 * plane + i, work<FMAS>(s2, s0, s1));
}

// the fused blend + render BACKWARD pattern: both materials (8 planes each), the mask and 3 upstream-gradient planes in, the gradients of both
// materials and of the mask out (20 in / 17 out, 148 B per pixel, 37 streams), PX pixels per lane (the kernel: 2 -> 8-byte accesses).
template <int WPE, int PX>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
void oneshot_blend_bwd(const float *__restrict__ a, const float *__restrict__ b, const float *__restrict__ mask, const float *__restrict__ gout,
                       float *__restrict__ ga, float *__restrict__ gb, float *__restrict__ gm, size_t npx) {
    typedef float vf __attribute__((ext_vector_type(PX)));
    const size_t i = ((size_t)blockIdx.x * 64 + threadIdx.x) * PX;
    if (i >= npx) return;
    auto L = [&](const float *p, size_t c) { return __builtin_nontemporal_load(reinterpret_cast<const vf *>(p + c * npx + i)); };
    vf s = L(mask, 0);
#pragma unroll
    for (int c = 0; c < 8; ++c) s += L(a, c) - L(b, c);
#pragma unroll
    for (int c = 0; c < 3; ++c) s += L(gout, c);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        __builtin_nontemporal_store(s + (float)c, reinterpret_cast<vf *>(ga + c * npx + i));
        __builtin_nontemporal_store(s - (float)c, reinterpret_cast<vf *>(gb + c * npx + i));
    }
    __builtin_nontemporal_store(s, reinterpret_cast<vf *>(gm + i));
}

// two 1 KiB pieces per plane and wave (lane l: pieces l and l + 64 of a 2 KiB run): 16 loads in flight per lane, every
// instruction still covers a contiguous 1 KiB.  WPE waves per SIMD.
template <bool NT, int FMAS, int WPE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
void oneshot_x2(const f4 *__restrict__ in, f4 *__restrict__ out, size_t nv, size_t plane) {
    const size_t i = (size_t)blockIdx.x * 128 + threadIdx.x;
    if (i + 64 >= nv) return;
    f4 v[8], w[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) { v[c] = ld<NT>(in + c * plane + i); w[c] = ld<NT>(in + c * plane + i + 64); }
    f4 r0 = work<FMAS>(v[0] + v[3] + v[6], v[6], v[7]), r1 = work<FMAS>(v[1] + v[4] + v[7], v[6], v[7]), r2 = work<FMAS>(v[2] + v[5], v[6], v[7]);
    f4 q0 = work<FMAS>(w[0] + w[3] + w[6], w[6], w[7]), q1 = work<FMAS>(w[1] + w[4] + w[7], w[6], w[7]), q2 = work<FMAS>(w[2] + w[5], w[6], w[7]);
    st<NT>(out + i, r0); st<NT>(out + plane + i, r1); st<NT>(out + 2 * plane + i, r2);
    st<NT>(out + i + 64, q0); st<NT>(out + plane + i + 64, q1); st<NT>(out + 2 * plane + i + 64, q2);
}


// the resize family's patterns (csrc/resize.hip, resize_down.hpp): RI 16-byte streams in, WO out per lane, nothing else -- a 2x down-scale or the
// gradient of a 2x up-scale is 4 : 1, a 4x down-scale 16 : 1, a 1.5x up-scale 4 : 9, the gradient of a 2x down-scale 1 : 4.  NT: the loads'
// hint (the stores are non-temporal always, as the kernels').  TWO: the lane's loads are pairs of neighbouring 16-byte pieces (a lane of
// resize_down_kernel owns 32 contiguous bytes of a row: two instructions on the same lines) instead of RI streams a plane apart.
template <int RI, int WO, bool NT, bool TWO>
__global__ __launch_bounds__(64) void ratio_pattern(const f4 *__restrict__ in, f4 *__restrict__ out, size_t units) {
    const size_t u = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (u >= units) return;
    f4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int r = 0; r < RI; ++r) {
        const f4 *p = TWO ? in + (size_t)(r / 2) * 2 * units + 2 * u + (r & 1) : in + (size_t)r * units + u;
        acc += NT ? __builtin_nontemporal_load(p) : *p;
    }
#pragma unroll
    for (int w = 0; w < WO; ++w) __builtin_nontemporal_store(acc, out + (size_t)w * units + u);
}

// read-only and write-only ceilings
template <bool NT> __global__ __launch_bounds__(256) void readonly(const f4 *__restrict__ in, f4 *__restrict__ out, size_t nv, size_t plane) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nv) return;
    f4 s = {0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < 8; ++c) s += ld<NT>(in + c * plane + i);
    if (s.x == 12345.678f) out[i] = s;
}
template <bool NT> __global__ __launch_bounds__(256) void writeonly(f4 *__restrict__ out, size_t nv, size_t plane) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nv) return;
    f4 s = {1, 2, 3, 4};
    st<NT>(out + i, s); st<NT>(out + plane + i, s); st<NT>(out + 2 * plane + i, s);
}

struct Result { const char *name; double us; double gbs; };

template <typename L> static double time_us(L launch, int iters) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch(i);
    std::vector<double> t;
    for (int r = 0; r < 5; ++r) {
        CHECK(hipEventRecord(e0, 0));
        for (int i = 0; i < iters; ++i) launch(i);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        t.push_back(ms * 1e3 / iters);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main(int argc, char **argv) {
    const size_t S = argc > 1 ? atoi(argv[1]) : 4096;
    const size_t px = S * S, nv = px / 4, plane = nv;
    const int NSETS = 3;
    f4 *in[NSETS], *out[NSETS];
    for (int s = 0; s < NSETS; ++s) {
        CHECK(hipMalloc(&in[s], 8 * px * 4)); CHECK(hipMalloc(&out[s], 3 * px * 4));
        CHECK(hipMemset(in[s], 0x3c, 8 * px * 4));
    }
    const double bytes_rw = 44.0 * px, bytes_r = 32.0 * px, bytes_w = 12.0 * px;
    const int iters = 20;
    const bool only_resize = argc > 2 && !strcmp(argv[2], "resize");
    auto report = [&](const char *name, double us, double bytes) { printf("%-44s %8.2f us  %7.1f GB/s\n", name, us, bytes / us / 1e3); fflush(stdout); };

    report("read-only 8 planes nt", time_us([&](int i) { hipLaunchKernelGGL(readonly<true>, dim3((nv + 255) / 256), dim3(256), 0, 0, in[i % NSETS], out[i % NSETS], nv, plane); }, iters), bytes_r);
    report("read-only 8 planes", time_us([&](int i) { hipLaunchKernelGGL(readonly<false>, dim3((nv + 255) / 256), dim3(256), 0, 0, in[i % NSETS], out[i % NSETS], nv, plane); }, iters), bytes_r);
    report("write-only 3 planes nt", time_us([&](int i) { hipLaunchKernelGGL(writeonly<true>, dim3((nv + 255) / 256), dim3(256), 0, 0, out[i % NSETS], nv, plane); }, iters), bytes_w);
    report("write-only 3 planes", time_us([&](int i) { hipLaunchKernelGGL(writeonly<false>, dim3((nv + 255) / 256), dim3(256), 0, 0, out[i % NSETS], nv, plane); }, iters), bytes_w);

    {   // the resize family: 3 planes of S^2 on the large side
        const size_t big = 3 * px / 4;                     // f4 units
#define RATIO(RI, WO, NT, TWO, UNITS, WHAT) report(WHAT, time_us([&](int i) { hipLaunchKernelGGL((ratio_pattern<RI, WO, NT, TWO>), dim3((unsigned)(((UNITS) + 63) / 64)), dim3(64), 0, 0, in[i % NSETS], out[i % NSETS], (size_t)(UNITS)); }, iters), 16.0 * (UNITS) * (RI + WO))
        RATIO(4, 1, false, false, big / 4, "resize pattern 4:1 (2x down, grad of 2x up), plain loads");
        RATIO(4, 1, true, false, big / 4, "resize pattern 4:1, nt loads");
        RATIO(4, 1, false, true, big / 4, "resize pattern 4:1, plain loads in 32-byte pairs");
        RATIO(4, 1, true, true, big / 4, "resize pattern 4:1, nt loads in 32-byte pairs");
        RATIO(16, 1, false, false, big / 16, "resize pattern 16:1 (4x down), plain loads");
        RATIO(16, 1, true, false, big / 16, "resize pattern 16:1, nt loads");
        RATIO(16, 1, false, true, big / 16, "resize pattern 16:1, plain loads in 32-byte pairs");
        {   // 2.25 x the large side out: its own buffer
            f4 *up = nullptr; CHECK(hipMalloc(&up, 9 * (big / 4) * 16));
#define RATIO_UP(NT, WHAT) report(WHAT, time_us([&](int i) { hipLaunchKernelGGL((ratio_pattern<4, 9, NT, false>), dim3((unsigned)((big / 4 + 63) / 64)), dim3(64), 0, 0, in[i % NSETS], up, big / 4); }, iters), 16.0 * (big / 4) * 13)
            RATIO_UP(false, "resize pattern 4:9 (1.5x up), plain loads");
            RATIO_UP(true, "resize pattern 4:9, nt loads");
            CHECK(hipFree(up));
        }
        // EIGHT planes on the large side (537 MB: nothing survives a launch): what the row walk (resize_stream.hpp) is measured against
        RATIO(9, 1, true, false, 2 * px / 9, "resize pattern 9:1 over 8 planes (3x down), nt loads");
        RATIO(9, 1, false, false, 2 * px / 9, "resize pattern 9:1 over 8 planes, plain loads");
        RATIO(105, 1, true, false, 2 * px / 105, "resize pattern 105:1 over 8 planes (10.24x down), nt loads");
        RATIO(1, 4, false, false, big / 4, "resize pattern 1:4 (grad of 2x down), plain loads");
        RATIO(1, 4, true, false, big / 4, "resize pattern 1:4, nt loads");
        RATIO(4, 1, false, false, big / 4, "resize pattern 4:1 (2x down, grad of 2x up), plain loads");
    }
    if (only_resize) return 0;

#define ONESHOT(NT, F, B) report("oneshot nt=" #NT " valu/px=" #F " block=" #B, time_us([&](int i) { hipLaunchKernelGGL((oneshot<NT, F, B>), dim3((nv + B - 1) / B), dim3(B), 0, 0, in[i % NSETS], out[i % NSETS], nv, plane, plane); }, iters), bytes_rw)
    ONESHOT(true, 0, 256); ONESHOT(false, 0, 256); ONESHOT(true, 0, 128); ONESHOT(true, 0, 512); ONESHOT(true, 0, 1024);
    ONESHOT(true, 40, 256); ONESHOT(true, 80, 256);
    report("oneshot loads nt, stores plain", time_us([&](int i) { hipLaunchKernelGGL((oneshot<true, 0, 256, false>), dim3((nv + 255) / 256), dim3(256), 0, 0, in[i % NSETS], out[i % NSETS], nv, plane, plane); }, iters), bytes_rw);
    report("oneshot loads plain, stores nt", time_us([&](int i) { hipLaunchKernelGGL((oneshot<false, 0, 256, true>), dim3((nv + 255) / 256), dim3(256), 0, 0, in[i % NSETS], out[i % NSETS], nv, plane, plane); }, iters), bytes_rw);
    // padded plane strides: do the 64 MiB-aligned planes camp on the same HBM channels/banks?
    for (size_t pad : {(size_t)0, (size_t)16, (size_t)64, (size_t)272, (size_t)1040, (size_t)4112, (size_t)65552}) {   // in f4 units (16 B)
        char nm[96]; snprintf(nm, sizeof(nm), "oneshot nt plane stride +%zu B", pad * 16);
        if (8 * (plane + pad) * 16 > 8 * px * 4 + 0) { /* need bigger buffers: allocate once */ }
        static f4 *bin = nullptr, *bout = nullptr;
        if (!bin) { CHECK(hipMalloc(&bin, 8 * (px * 4 + 2 * 1048576))); CHECK(hipMalloc(&bout, 3 * (px * 4 + 2 * 1048576))); CHECK(hipMemset(bin, 0x3c, 8 * (px * 4 + 2 * 1048576))); }
        report(nm, time_us([&](int i) { hipLaunchKernelGGL((oneshot<true, 0, 256>), dim3((nv + 255) / 256), dim3(256), 0, 0, bin, bout, nv, plane + pad, plane + pad); }, iters), bytes_rw);
    }
#define W3(K, NT, F) report(#K " nt=" #NT " valu/px=" #F " (1-wave groups, 3 waves/SIMD)", time_us([&](int i) { hipLaunchKernelGGL((K<NT, F>), dim3((nv + 63) / 64), dim3(64), 0, 0, in[i % NSETS], out[i % NSETS], nv, plane); }, iters), bytes_rw)
    W3(oneshot_w3, true, 0); W3(oneshot_ldsdma, true, 0); W3(oneshot_ldsdma, false, 0);
    W3(oneshot_w3, true, 60); W3(oneshot_ldsdma, true, 60);
    W3(oneshot_w3, true, 0); W3(oneshot_ldsdma, true, 0);
#define X2(WPE, F, LDS) report("oneshot_x2 valu/px=" #F " waves/SIMD=" #WPE " lds=" #LDS, time_us([&](int i) { hipLaunchKernelGGL((oneshot_x2<true, F, WPE>), dim3((nv + 127) / 128), dim3(64), LDS, 0, in[i % NSETS], out[i % NSETS], nv, plane); }, iters), bytes_rw)
#define W3L(F, LDS) report("oneshot_w3 valu/px=" #F " lds=" #LDS, time_us([&](int i) { hipLaunchKernelGGL((oneshot_w3<true, F>), dim3((nv + 63) / 64), dim3(64), LDS, 0, in[i % NSETS], out[i % NSETS], nv, plane); }, iters), bytes_rw)
    W3L(0, 14848); W3L(60, 14848); W3L(160, 14848);
    X2(1, 0, 0); X2(2, 0, 0); X2(2, 0, 24576); X2(2, 0, 32768); X2(3, 0, 0);
    X2(1, 60, 0); X2(2, 60, 0); X2(2, 60, 24576); X2(2, 60, 32768); X2(2, 160, 0); X2(2, 160, 24576);
    W3L(0, 14848); X2(2, 0, 24576);
    {   // backward pattern: 19 streams
        f4 *bo = nullptr; CHECK(hipMalloc(&bo, 8 * px * 4));
        const double bytes_b = 76.0 * px;
#define BWD(WPE, F) report("backward pattern 11 in / 8 out, valu/px=" #F " waves/SIMD=" #WPE, time_us([&](int i) { hipLaunchKernelGGL((oneshot_bwd<true, F, WPE>), dim3((nv + 63) / 64), dim3(64), 0, 0, in[i % NSETS], out[i % NSETS], bo, nv, plane); }, iters), bytes_b)
        BWD(2, 0); BWD(3, 0); BWD(4, 0); BWD(2, 60); BWD(2, 160); BWD(3, 60); BWD(2, 0);
        CHECK(hipFree(bo));
    }
    {   // backward pattern with fp16 maps: 4-byte accesses (what the kernels do) against 16-byte accesses on the same bytes
        unsigned *hi = nullptr, *ho = nullptr; float *hg = nullptr;
        CHECK(hipMalloc(&hi, 8 * px * 2)); CHECK(hipMalloc(&ho, 8 * px * 2)); CHECK(hipMalloc(&hg, 3 * px * 4));
        CHECK(hipMemset(hi, 0x3c, 8 * px * 2)); CHECK(hipMemset(hg, 0x3c, 3 * px * 4));
        const double bytes_h = 44.0 * px;
#define BWD16(PX, WPE) report("fp16 backward pattern, " #PX " px per lane, waves/SIMD=" #WPE, time_us([&](int i) { hipLaunchKernelGGL((oneshot_bwd16<PX, WPE>), dim3((unsigned)((px / PX + 63) / 64)), dim3(64), 0, 0, hi, hg, ho, px, px); }, iters), bytes_h)
        BWD16(2, 4); BWD16(2, 3); BWD16(2, 8); BWD16(4, 4); BWD16(8, 3); BWD16(8, 2); BWD16(2, 4); BWD16(8, 3);
        CHECK(hipFree(hi)); CHECK(hipFree(ho)); CHECK(hipFree(hg));
    }
    {   // forward pattern with fp16 maps -> fp32 result, four materials (the config 5' probe shape)
        const size_t p4 = 4 * px;
        f4 *fi = nullptr, *fo = nullptr;
        CHECK(hipMalloc(&fi, 8 * p4 * 2)); CHECK(hipMalloc(&fo, 3 * p4 * 4)); CHECK(hipMemset(fi, 0x3c, 8 * p4 * 2));
#define FWD16(WPE) report("fp16 forward pattern 4 x S^2, 8 px per lane, waves/SIMD=" #WPE, time_us([&](int i) { hipLaunchKernelGGL((oneshot_fwd16<WPE>), dim3((unsigned)((p4 / 8 + 63) / 64)), dim3(64), 0, 0, fi, fo, p4, p4); }, iters), 28.0 * p4)
        FWD16(2); FWD16(3); FWD16(4); FWD16(3);
        CHECK(hipFree(fi)); CHECK(hipFree(fo));
    }
    {   // repeat-inner pattern: (S/2)^2 source, tile(2) -> S^2
        const int src = (int)(S / 2);
        const size_t sp = (size_t)src * src;
#define REPEAT(HALF, WPE) report(HALF ? "repeat pattern, fp16 source, waves/SIMD>=" #WPE : "repeat pattern, fp32 source, waves/SIMD>=" #WPE, time_us([&](int i) { \
            hipLaunchKernelGGL((oneshot_repeat<HALF, WPE>), dim3((unsigned)(sp / 4 / 64)), dim3(64), 0, 0, (const float *)in[i % NSETS], out[i % NSETS], src, sp); }, iters), \
            (HALF ? 16.0 : 32.0) * sp + 12.0 * px)
        REPEAT(false, 4); REPEAT(true, 4); REPEAT(false, 2); REPEAT(true, 2); REPEAT(false, 8); REPEAT(true, 8); REPEAT(false, 4); REPEAT(true, 4);
    }
    {   // fused blend + render pattern: 17 planes in, 3 out
        f4 *mk = nullptr; CHECK(hipMalloc(&mk, px * 4)); CHECK(hipMemset(mk, 0x3c, px * 4));
#define BLEND(WPE, F) report("blend+render pattern 17 in / 3 out, valu/px=" #F " waves/SIMD=" #WPE, time_us([&](int i) { hipLaunchKernelGGL((oneshot_blend<WPE, F>), dim3((nv + 63) / 64), dim3(64), 0, 0, in[0], in[1], mk, out[i % NSETS], nv, plane); }, iters), 80.0 * px)
        BLEND(2, 0); BLEND(3, 0); BLEND(2, 60); BLEND(3, 60); BLEND(2, 0);
        CHECK(hipFree(mk));
    }
    {   // fused blend + render backward pattern: 20 planes in, 17 out
        float *mk, *go, *ga, *gb, *gm;
        CHECK(hipMalloc(&mk, px * 4)); CHECK(hipMalloc(&go, 3 * px * 4)); CHECK(hipMalloc(&ga, 8 * px * 4)); CHECK(hipMalloc(&gb, 8 * px * 4)); CHECK(hipMalloc(&gm, px * 4));
        CHECK(hipMemset(mk, 0x3c, px * 4)); CHECK(hipMemset(go, 0x3c, 3 * px * 4));
#define BLENDBWD(WPE, PX) report("blend+render backward pattern 20 in / 17 out, " #PX " px per lane, waves/SIMD=" #WPE, time_us([&](int i) { \
            hipLaunchKernelGGL((oneshot_blend_bwd<WPE, PX>), dim3((unsigned)((px / PX + 63) / 64)), dim3(64), 0, 0, (const float *)in[0], (const float *)in[1], mk, go, ga, gb, gm, px); }, 10), 148.0 * px)
        BLENDBWD(2, 2); BLENDBWD(3, 2); BLENDBWD(4, 2); BLENDBWD(2, 4); BLENDBWD(3, 4); BLENDBWD(2, 2);
        CHECK(hipFree(mk)); CHECK(hipFree(go)); CHECK(hipFree(ga)); CHECK(hipFree(gb)); CHECK(hipFree(gm));
    }
    for (int cfg = 0; cfg < 2; ++cfg) {   // config 3 (64 x 2048^2) and config 4's share (64 x 1024^2): the batch layout's own ceiling, both orders
        const size_t s = cfg == 0 ? 2048 : 1024, B = 64, pp = s * s / 4;          // plane in f4 units
        f4 *al, *nm, *ro, *me, *ou;
        CHECK(hipMalloc(&al, B * 3 * pp * 16)); CHECK(hipMalloc(&nm, B * 3 * pp * 16)); CHECK(hipMalloc(&ro, B * pp * 16)); CHECK(hipMalloc(&me, B * pp * 16));
        CHECK(hipMalloc(&ou, B * 3 * pp * 16));
        CHECK(hipMemset(al, 0x3c, B * 3 * pp * 16)); CHECK(hipMemset(nm, 0x3c, B * 3 * pp * 16)); CHECK(hipMemset(ro, 0x3c, B * pp * 16)); CHECK(hipMemset(me, 0x3c, B * pp * 16));
        const unsigned tpm = (unsigned)(pp / 64), nt = (unsigned)(B * tpm);
        const double bytes_b = 44.0 * B * s * s;
        char nm2[96];
#define BATCH(RUNS, F) snprintf(nm2, sizeof(nm2), "batch 64 x %zu^2, order=%s, valu/px=%d", s, RUNS ? "runs of 64" : "identity", F); \
        report(nm2, time_us([&](int i) { hipLaunchKernelGGL((batch_w3<RUNS, F>), dim3(nt), dim3(64), 0, 0, al, nm, ro, me, ou, tpm, nt, pp); }, 10), bytes_b)
        BATCH(false, 0); BATCH(true, 0); BATCH(false, 60); BATCH(true, 60); BATCH(false, 0); BATCH(true, 0);
        if (cfg == 0) {      // the same batch with every plane stride padded off the 16 MiB grid (one allocation per tensor, planes pp + pad apart)
            for (size_t pad : {(size_t)272, (size_t)4112, (size_t)65552}) {          // f4 units: 4 352 B, 64.25 KiB, 1 MiB + 256 B
                f4 *al2, *nm2b, *ro2, *me2, *ou2;
                const size_t ps = pp + pad;
                CHECK(hipMalloc(&al2, B * 3 * ps * 16)); CHECK(hipMalloc(&nm2b, B * 3 * ps * 16)); CHECK(hipMalloc(&ro2, B * ps * 16)); CHECK(hipMalloc(&me2, B * ps * 16));
                CHECK(hipMalloc(&ou2, B * 3 * ps * 16));
                CHECK(hipMemset(al2, 0x3c, B * 3 * ps * 16)); CHECK(hipMemset(nm2b, 0x3c, B * 3 * ps * 16)); CHECK(hipMemset(ro2, 0x3c, B * ps * 16)); CHECK(hipMemset(me2, 0x3c, B * ps * 16));
                snprintf(nm2, sizeof(nm2), "batch 64 x 2048^2, plane stride +%zu B, identity", pad * 16);
                report(nm2, time_us([&](int i) { hipLaunchKernelGGL((batch_w3<false, 0>), dim3(nt), dim3(64), 0, 0, al2, nm2b, ro2, me2, ou2, tpm, nt, ps); }, 10), bytes_b);
                snprintf(nm2, sizeof(nm2), "batch 64 x 2048^2, plane stride +%zu B, runs of 64", pad * 16);
                report(nm2, time_us([&](int i) { hipLaunchKernelGGL((batch_w3<true, 0>), dim3(nt), dim3(64), 0, 0, al2, nm2b, ro2, me2, ou2, tpm, nt, ps); }, 10), bytes_b);
                CHECK(hipFree(al2)); CHECK(hipFree(nm2b)); CHECK(hipFree(ro2)); CHECK(hipFree(me2)); CHECK(hipFree(ou2));
            }
        }
        CHECK(hipFree(al)); CHECK(hipFree(nm)); CHECK(hipFree(ro)); CHECK(hipFree(me)); CHECK(hipFree(ou));
    }
#define PERSIST(NT, F, B, G) report("persistent nt=" #NT " valu/px=" #F " block=" #B " grid=" #G, time_us([&](int i) { hipLaunchKernelGGL((persistent<NT, F, B>), dim3(G), dim3(B), 0, 0, in[i % NSETS], out[i % NSETS], nv, plane); }, iters), bytes_rw)
    PERSIST(true, 0, 256, 1024); PERSIST(true, 0, 256, 2048);
    PERSIST(true, 60, 256, 2048);
    return 0;
}
