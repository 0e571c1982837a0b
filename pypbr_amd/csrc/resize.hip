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
// Two HBM-streaming passes; each lane produces one output element and walks its taps (<= 2*scale+2)
// along the contiguous axis (width pass) or down a column (height pass: lanes of a wave are
// consecutive in x, so every tap row is a coalesced run).
#include <hip/hip_runtime.h>

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
    hipLaunchKernelGGL(resize_width_kernel, dim3(stream_grid(planes * h_in * w_out)), dim3(256), 0, s,
                       static_cast<const float *>(src), tmp, planes * h_in, (int)w_out, fw);
    hipLaunchKernelGGL(resize_height_kernel, dim3(stream_grid(planes * h_out * w_out)), dim3(256), 0, s,
                       tmp, static_cast<float *>(dst), planes, (int)h_out, (int)w_out, fh);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? PBR_OK : 1000 + (int)e;
}

}  // extern "C"
