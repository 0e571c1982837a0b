// Issue rates of the integer instructions the kernels' address arithmetic uses, against v_fma_f32 (4 cycles per wave
// instruction).  hipcc -O3 --offload-arch=gfx950 tools/valu_rate.hip -o tools/bin/valu_rate && tools/bin/valu_rate
// One wave per SIMD (grid = 256 CUs x 4), CHAINS independent dependency chains per lane, N instructions per chain.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int CHAINS = 8, ITERS = 4096;

template <int OP>
__global__ __launch_bounds__(64) void rate_kernel(uint32_t *out, uint32_t seed, uint32_t mul) {
    uint32_t v[CHAINS];
    uint64_t w[CHAINS];
    float f[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) { v[c] = seed + threadIdx.x * 7 + c; w[c] = v[c]; f[c] = (float)v[c]; }
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) {
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[c]) : "v"(1.0001f));
            if (OP == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v[c]) : "v"(mul));
            if (OP == 2) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(v[c]) : "v"(mul));
            if (OP == 3) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[c]) : "v"(v[c]), "v"(mul) : "vcc");
            if (OP == 4) asm volatile("v_lshl_add_u64 %0, %0, 2, %1" : "+v"(w[c]) : "v"(w[(c + 1) % CHAINS]));
            if (OP == 5) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(v[c]) : "v"(mul));
            if (OP == 6) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[c]) : "v"(mul));
            if (OP == 7) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(w[c]) : "v"(w[(c + 1) % CHAINS]));
            if (OP == 8) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(v[c]) : "v"(mul));
            if (OP == 9) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[c]) : "v"(mul));
            if (OP == 10) asm volatile("v_exp_f32 %0, %0" : "+v"(f[c]));
            if (OP == 11) asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(f[c]));
        }
    }
    uint32_t acc = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc += v[c] + (uint32_t)w[c] + (uint32_t)f[c];
    if (acc == 0x12345678u) out[0] = acc;
}

static double g_base_ms = 0;
template <int OP> static void run(const char *name, uint32_t *out) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(rate_kernel<OP>, dim3(1024), dim3(64), 0, 0, out, 3u, 77u);
    (void)hipEventRecord(e0, 0);
    for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(rate_kernel<OP>, dim3(1024), dim3(64), 0, 0, out, 3u, 77u);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= 20;
    if (OP == 0) g_base_ms = ms;
    printf("%-16s %8.3f ms per launch  = %.2f x v_fma_f32  (~%.1f cycles per wave instruction)\n", name, ms, ms / g_base_ms, 4.0 * ms / g_base_ms);
}

int main() {
    uint32_t *out; (void)hipMalloc(&out, 64);
    for (int w = 0; w < 200; ++w) hipLaunchKernelGGL(rate_kernel<0>, dim3(1024), dim3(64), 0, 0, out, 3u, 77u);   // settle the clocks
    run<0>("v_fma_f32", out);
    run<1>("v_mul_lo_u32", out);
    run<2>("v_mul_u32_u24", out);
    run<3>("v_mad_u64_u32", out);
    run<4>("v_lshl_add_u64", out);
    run<5>("v_mul_hi_u32", out);
    run<6>("v_add_u32", out);
    run<7>("v_pk_fma_f32", out);
    run<8>("v_mad_u32_u24", out);
    run<9>("v_cndmask_b32", out);
    run<10>("v_exp_f32", out);
    run<11>("v_cvt_f32_f16", out);
    (void)hipFree(out);
    return 0;
}
