// stream_shape.hpp -- launch shape of the streaming map kernels (map_ops.hip, blend.hip).
#pragma once
#include <cstddef>

#include "tuning.hpp"

namespace pbr {

// Launch shape of the streaming map kernels (map_ops.hip, blend.hip): all of them are grid-stride loops over blockDim-agnostic
// indices, so the shape is the launcher's choice.  shape 0 = 2048 workgroups of 256 lanes walking the data, 1 = one item per lane in
// 256-lane workgroups, 2 = one item per lane in one-wave workgroups; lds = unused dynamic LDS per workgroup, which caps the resident
// waves the way the render kernel's occupancy governor does.  Each launcher names its rule (measured below; the override knobs of
// ABI 6 went with their experiment: profiles/EXPERIMENTS.md).
struct StreamRule { int shape, lds; };
struct StreamShape { unsigned grid, block; size_t lds; };
inline StreamShape stream_shape(size_t work_items, StreamRule rule) {
    const int shape = rule.shape, lds = rule.lds;
    const unsigned block = shape == 2 ? 64u : 256u;
    size_t blocks = (work_items + block - 1) / block;
    if (shape == 0 && blocks > 256 * 8) blocks = 256 * 8;
    if (blocks > 0x7fffffffu) blocks = 0x7fffffffu;
    return {(unsigned)(blocks < 1 ? 1 : blocks), block, (size_t)(lds > 0 ? lds : 0)};
}
// Rules, measured on 4096^2 fp32 maps (stream_shape_probe.py (a probe of its round, removed with its knob: git 9ce0718:tools/), two boxes; us, shape 0 -> the rule): one item per lane beats the
// walking workgroups by 4-9 % on every kernel but the one-pixel-per-lane mask kernel (30.9 -> 32.8); the kernels with ten planes and
// little arithmetic per byte gain another few per cent from one-wave workgroups held to two waves per SIMD (20 KiB of LDS each).
//   metallic_to_specular 109.9 / 125.8 -> 101.6 / 105.1     specular_to_metallic 43.6 -> 41.7     colour 65.7 / 68.4 -> 59.8 / 60.8
//   colour backward 100.3 -> 91.2     metallic_to_specular backward 157.7 -> 152.3     specular_to_metallic backward 66.2 -> 63.1
//   blend 3 channels 110.7 -> 104.2     blend normals 109.6 -> 98.8     blend backward 190.5 -> 182.0
constexpr StreamRule kShapeM2S = {2, 20480}, kShapeS2M = {1, 0}, kShapeColour = {2, 0}, kShapeColourBwd = {2, 0}, kShapeM2SBwd = {1, 0},
                     kShapeS2MBwd = {1, 0}, kShapeBlend = {1, 0}, kShapeBlendNormal = {2, 20480}, kShapeBlendBwd = {1, 0}, kShapeMask = {0, 0}, kShapeFold = {1, 0};      // fold: tile(2) of 3 x 2048^2 44.0 -> 40.8 us, a map shared by 8 materials 81.3 -> 77.9, nine repeats level (fold_probe.py (a probe of its round, removed with its knob: git 9ce0718:tools/))

}  // namespace pbr
