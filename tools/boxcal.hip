// boxcal.hip -- the bench line's box calibration (VERDICT r4, next #1a): the BARE access pattern of the headline kernel -- 8 fp32
// planes in, 3 planes out, one 16-byte streaming access per lane and plane, one-wave workgroups at the same occupancy
// (amdgpu_waves_per_eu(3,3) + 14 848 B of unused dynamic LDS = 11 waves per CU) -- with no arithmetic worth the name.  bench.py runs it
// on the very buffers of the timed launches right after the steady region and puts its time beside the kernel's
// (`roofline.box_pattern_us`, `roofline.kernel_over_box_pattern`): the fused kernel has sat within +-3 % of this pattern on every box
// (profiles/r04_membench.txt), so a slow line with ratio ~1 is a slow BOX, and a ratio that grows is a slow BINARY.
// Bench-side helper: built by __graft_entry__.build() into tools/libboxcal.so, NOT part of libpbr_hip.so (the product has no use for it).
//   hipcc -O3 --offload-arch=gfx950 -shared -fPIC tools/boxcal.hip -o tools/libboxcal.so
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f4 __attribute__((ext_vector_type(4)));

struct BoxPlanes { const f4 *in[8]; f4 *out[3]; };

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3)))
void boxcal_forward_pattern_kernel(const BoxPlanes p, const uint32_t n_vec) {
    const uint32_t i = blockIdx.x * 64u + threadIdx.x;
    if (i >= n_vec) return;
    f4 v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = __builtin_nontemporal_load(p.in[c] + i);
    const f4 r0 = v[0] + v[3] + v[6], r1 = v[1] + v[4] + v[7], r2 = v[2] + v[5];
    __builtin_nontemporal_store(r0, p.out[0] + i);
    __builtin_nontemporal_store(r1, p.out[1] + i);
    __builtin_nontemporal_store(r2, p.out[2] + i);
}

extern "C" {

// in[8] / out[3]: device pointers to planes of `pixels` floats each (16-byte aligned, pixels % 4 == 0); enqueues ONE launch on `stream`.
// Returns 0, -1 on bad arguments, or 1000 + the HIP error.
int boxcal_forward_pattern(const void *const *in, void *const *out, uint64_t pixels, void *stream) {
    if (!in || !out || pixels == 0 || pixels % 4 || pixels / 4 > 0xffffffffull) return -1;
    BoxPlanes p;
    for (int c = 0; c < 8; ++c) {
        if (!in[c] || (reinterpret_cast<uintptr_t>(in[c]) & 15u)) return -1;
        p.in[c] = static_cast<const f4 *>(in[c]);
    }
    for (int c = 0; c < 3; ++c) {
        if (!out[c] || (reinterpret_cast<uintptr_t>(out[c]) & 15u)) return -1;
        p.out[c] = static_cast<f4 *>(out[c]);
    }
    const uint32_t n_vec = (uint32_t)(pixels / 4);
    hipLaunchKernelGGL(boxcal_forward_pattern_kernel, dim3((n_vec + 63) / 64), dim3(64), 14848, static_cast<hipStream_t>(stream), p, n_vec);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : 1000 + (int)e;
}

int boxcal_version(void) { return 1; }

}  // extern "C"
