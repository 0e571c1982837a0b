// unpack.hip -- decoded image samples (uint8 / uint16, as PIL hands them out) -> planar float32 maps, on the device.
//
// Reference function replaced: /root/reference/pypbr/materials/base.py:122-168 `MaterialBase._to_tensor` for PIL images
// (16-bit modes: samples / 65535.0; every other mode through torchvision's to_tensor: uint8 (H,W,C) -> float32 (C,H,W) / 255),
// and -- for a normal map -- base.py:191-242 `_process_normal_map` behind it in the same pass: an image's samples are never
// negative, so the "already signed?" test of :212 is decided before the data is looked at and the map is always decoded.
//
// Why on the device: the loader's float conversion was the second largest item of examples/example_brdf.py after the PNG inflate
// itself, and float maps are four times the bytes of their samples on the host-to-device copy.  The samples travel as they are;
// this kernel is one HBM-bound pass: 4 consecutive pixels per lane, their samples in whole dwords, one 16-byte store per plane.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/pbr_hip.h"
#include "brdf_math.hpp"

namespace pbr {

// a / b correctly rounded (IEEE): torch's CPU `div` is a true division, and 1/255 is not a float
__device__ __forceinline__ float sample_to_unit(unsigned v, float divisor) { return (float)v / divisor; }

// base.py:191-242 on one pixel whose components are in [0, 1]: the arithmetic of map_ops.hip's decode_normal_kernel, operation for operation
template <int C> __device__ __forceinline__ void decode_unsigned_normal(float &x, float &y, float &z) {
    x = fmaf(x, 2.0f, -1.0f); y = fmaf(y, 2.0f, -1.0f);
    if (C == 3) z = fmaf(z, 2.0f, -1.0f);
    else z = sqrt_hw(fmaxf(1.0f - (x * x + y * y), 1e-6f));                       // base.py:235-240
    const float r = rsq(fmaxf(fmaf(z, z, fmaf(y, y, x * x)), 1e-24f));              // F.normalize
    x *= r; y *= r; z *= r;
}

// Dense (H,W,C) samples, W % 4 == 0, dword-aligned rows: lane = 4 consecutive pixels = 4*C samples = C (uint8) or 2*C (uint16) dwords.
template <typename U, int C, bool NORMAL>
__global__ __launch_bounds__(256) void unpack_dense_kernel(const uint32_t *__restrict__ src, float *__restrict__ dst, int64_t quads, int64_t plane,
                                                           float divisor) {
    constexpr int WORDS = C * (int)sizeof(U);
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= quads) return;
    uint32_t w[WORDS];
#pragma unroll
    for (int k = 0; k < WORDS; ++k) w[k] = src[q * WORDS + k];
    auto sample = [&](int k) -> unsigned {                 // k-th sample of the lane's 4*C
        return sizeof(U) == 1 ? (w[k >> 2] >> (8 * (k & 3))) & 0xffu : (w[k >> 1] >> (16 * (k & 1))) & 0xffffu;
    };
    constexpr int CO = NORMAL ? 3 : C;
    float v[CO][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float s[C];
#pragma unroll
        for (int c = 0; c < C; ++c) s[c] = sample_to_unit(sample(j * C + c), divisor);
        if (NORMAL) {
            float x = s[0], y = s[1], z = C == 3 ? s[C - 1] : 0.0f;
            decode_unsigned_normal<C>(x, y, z);
            v[0][j] = x; v[1][j] = y; v[CO - 1][j] = z;
        } else {
#pragma unroll
            for (int c = 0; c < C; ++c) v[c][j] = s[c];
        }
    }
#pragma unroll
    for (int c = 0; c < CO; ++c)
        *reinterpret_cast<float4 *>(dst + (int64_t)c * plane + 4 * q) = make_float4(v[c][0], v[c][1], v[c][2], v[c][3]);
}

// Any strides (element units), any extents: one pixel per lane.
template <typename U, bool NORMAL>
__global__ __launch_bounds__(256) void unpack_strided_kernel(const U *__restrict__ src, float *__restrict__ dst, int channels, int height, int width,
                                                             int64_t sc, int64_t sh, int64_t sw, float divisor) {
    const int64_t plane = (int64_t)height * width, p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= plane) return;
    const int64_t y = p / width, x = p - y * width, base = y * sh + x * sw;
    if (NORMAL) {
        float a = sample_to_unit(src[base], divisor), b = sample_to_unit(src[base + sc], divisor), c = 0.0f;
        if (channels == 3) { c = sample_to_unit(src[base + 2 * sc], divisor); decode_unsigned_normal<3>(a, b, c); }
        else decode_unsigned_normal<2>(a, b, c);
        dst[p] = a; dst[plane + p] = b; dst[2 * plane + p] = c;
    } else {
        for (int c = 0; c < channels; ++c) dst[(int64_t)c * plane + p] = sample_to_unit(src[base + c * sc], divisor);
    }
}

template <typename U, int C, bool NORMAL>
static void launch_dense(const void *src, float *dst, int64_t plane, float divisor, hipStream_t s) {
    const int64_t quads = plane / 4;
    hipLaunchKernelGGL((unpack_dense_kernel<U, C, NORMAL>), dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, s,
                       static_cast<const uint32_t *>(src), dst, quads, plane, divisor);
}

template <typename U>
static int unpack(const void *src, int channels, int height, int width, int64_t sc, int64_t sh, int64_t sw, float *dst, int normal, float divisor,
                  hipStream_t s) {
    const int64_t plane = (int64_t)height * width;
    const bool dense = sc == 1 && sw == channels && sh == (int64_t)width * channels && width % 4 == 0 &&
                       (reinterpret_cast<uintptr_t>(src) & 3) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0;
    if (dense && normal && channels == 3) launch_dense<U, 3, true>(src, dst, plane, divisor, s);
    else if (dense && normal && channels == 2) launch_dense<U, 2, true>(src, dst, plane, divisor, s);
    else if (dense && !normal && channels == 1) launch_dense<U, 1, false>(src, dst, plane, divisor, s);
    else if (dense && !normal && channels == 3) launch_dense<U, 3, false>(src, dst, plane, divisor, s);
    else {
        const dim3 grid((unsigned)((plane + 255) / 256));
        if (normal) hipLaunchKernelGGL((unpack_strided_kernel<U, true>), grid, dim3(256), 0, s, static_cast<const U *>(src), dst, channels, height, width, sc, sh, sw, divisor);
        else hipLaunchKernelGGL((unpack_strided_kernel<U, false>), grid, dim3(256), 0, s, static_cast<const U *>(src), dst, channels, height, width, sc, sh, sw, divisor);
    }
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? PBR_OK : 1000 + (int)e;
}

}  // namespace pbr

extern "C" int pbr_unpack_image(const void *src, int32_t bits, int32_t channels, int32_t height, int32_t width, int64_t stride_c,
                                int64_t stride_h, int64_t stride_w, float *dst, int32_t decode_normal, void *stream) {
    if (!src || !dst) return PBR_ERR_NULL_MAP;
    if (bits != 8 && bits != 16) return PBR_ERR_DTYPE;
    if (height < 1 || width < 1 || (int64_t)height * width > ((int64_t)1 << 40)) return PBR_ERR_SHAPE;
    if (decode_normal ? (channels != 2 && channels != 3) : (channels < 1 || channels > 4)) return PBR_ERR_CHANNELS;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (bits == 8) return pbr::unpack<uint8_t>(src, channels, height, width, stride_c, stride_h, stride_w, dst, decode_normal, 255.0f, s);
    return pbr::unpack<uint16_t>(src, channels, height, width, stride_c, stride_h, stride_w, dst, decode_normal, 65535.0f, s);
}
