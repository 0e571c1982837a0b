// How many cycles a SIMD of gfx950 spends per wave64 vector instruction as a function of the waves it holds: v_fma_f32, v_pk_fma_f32,
// v_exp_f32, a v_cmp + v_cndmask pair, and a mix.  One question: is packed fp32 arithmetic double rate, and does a second / third / fourth
// wave on a SIMD raise the issue rate of plain fp32 instructions?  (round 6; decides what the VALU-bound kernels should look like)
//   hipcc -O3 --offload-arch=gfx950 tools/valu_occupancy.hip -o tools/bin/valu_occupancy && tools/bin/valu_occupancy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int CHAINS = 8, ITERS = 2048;

template <int OP>
__global__ __launch_bounds__(64) void rate_kernel(uint32_t *out, float seed, uint32_t useed) {
    float f[CHAINS];
    double w[CHAINS];
    uint32_t sc[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) { f[c] = seed + threadIdx.x * 1e-3f + c; w[c] = (double)f[c]; sc[c] = useed + c; }
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) {
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[c]) : "v"(1.0001f));
            if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(w[c]) : "v"(w[(c + 1) % CHAINS]));
            if (OP == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(f[c]));
            if (OP == 3) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(f[c]) : "v"(seed) : "vcc");
            if (OP == 4) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(w[c]) : "v"(w[(c + 1) % CHAINS]));
            if (OP == 5) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[c]) : "v"(1.0001f));
            if (OP == 6) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(w[c]) : "v"(w[(c + 1) % CHAINS]));
            if (OP == 7) asm volatile("v_fma_f32 %0, %0, %2, %0\n\ts_add_u32 %1, %1, 3" : "+v"(f[c]), "+s"(sc[c]) : "v"(1.0001f) : "scc");
            if (OP == 8) asm volatile("v_fma_f32 %0, %0, %2, %0\n\ts_mul_i32 %1, %1, 3\n\ts_add_u32 %1, %1, 5" : "+v"(f[c]), "+s"(sc[c]) : "v"(1.0001f) : "scc");
            if (OP == 9) asm volatile("v_fma_f32 %0, %0, %1, %0\n\ts_nop 0" : "+v"(f[c]) : "v"(1.0001f));
            if (OP == 10) asm volatile("v_pk_fma_f32 %0, %0, %2, %0\n\ts_add_u32 %1, %1, 3" : "+v"(w[c]), "+s"(sc[c]) : "v"(w[(c + 1) % CHAINS]) : "scc");
            if (OP == 11) asm volatile("v_fma_f32 %0, %0, %2, %0\n\ts_cmp_eq_u32 %1, 77\n\ts_cbranch_scc1 1f\n\ts_nop 0\n1:" : "+v"(f[c]), "+s"(sc[c]) : "v"(1.0001f) : "scc");
        }
    }
    float acc = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc += f[c] + (float)w[c] + (float)(sc[c] & 1);
    if (acc == 1234.5f) out[0] = 1;
}

template <int OP> static void run(const char *name, uint32_t *out, int per_instr_ops) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    printf("%-26s", name);
    for (int waves : {1, 2, 3, 4, 6, 8}) {
        const int grid = 1024 * waves;
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(rate_kernel<OP>, dim3(grid), dim3(64), 0, 0, out, 3.0f, 5u);
        (void)hipEventRecord(e0, 0);
        for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(rate_kernel<OP>, dim3(grid), dim3(64), 0, 0, out, 3.0f, 5u);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        ms /= 10;
        // SIMD cycles per wave instruction at an assumed 2.4 GHz: time * clock / (instructions issued on one SIMD)
        const double per = ms * 1e-3 * 2.4e9 / ((double)waves * ITERS * CHAINS * per_instr_ops);
        printf("  %dw %.2f", waves, per);
    }
    printf("   (SIMD cycles per wave instruction at 2.4 GHz, by waves per SIMD)\n");
}

int main() {
    uint32_t *out; (void)hipMalloc(&out, 64);
    for (int w = 0; w < 300; ++w) hipLaunchKernelGGL(rate_kernel<0>, dim3(4096), dim3(64), 0, 0, out, 3.0f, 5u);   // settle the clocks
    run<0>("v_fma_f32", out, 1);
    run<5>("v_mul_f32", out, 1);
    run<1>("v_pk_fma_f32", out, 1);
    run<4>("v_pk_mul_f32", out, 1);
    run<6>("v_pk_add_f32", out, 1);
    run<2>("v_exp_f32", out, 1);
    run<3>("v_cmp + v_cndmask (per instr)", out, 2);
    run<7>("v_fma + s_add  (per PAIR)", out, 1);
    run<8>("v_fma + 2 SALU (per TRIPLE)", out, 1);
    run<9>("v_fma + s_nop 0 (per PAIR)", out, 1);
    run<10>("v_pk_fma + s_add (per PAIR)", out, 1);
    run<11>("v_fma + s_cmp + branch not taken + s_nop (per GROUP)", out, 1);
    return 0;
}
