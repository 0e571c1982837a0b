// ct_repeat_backward.hip -- instantiations and launch of the repeat-inner backward kernel (ct_repeat_backward.hpp), and the C ABI
// entry that hands out FOLDED gradients of tiled maps: pbr_cook_torrance_backward_folded.  The rendering-loss step over tiled maps
// (pbr_cook_torrance_mse_step, ct_loss.hip) launches the same kernel with the loss policy through launch_repeat_backward.
#include "ct_repeat_backward.hpp"
#include "ct_launch.hpp"

namespace pbr {

using RepBwdFn = void (*)(const KArgs, const BArgs, const RBArgs);

template <int L, int W>
static RepBwdFn repeat_bwd_types(bool half_maps, bool loss, bool multi) {
    if (multi) return half_maps ? cook_torrance_repeat_backward_kernel<L, W, __half, false, true> : cook_torrance_repeat_backward_kernel<L, W, float, false, true>;
    if (half_maps) return loss ? cook_torrance_repeat_backward_kernel<L, W, __half, true> : cook_torrance_repeat_backward_kernel<L, W, __half, false>;
    return loss ? cook_torrance_repeat_backward_kernel<L, W, float, true> : cook_torrance_repeat_backward_kernel<L, W, float, false>;
}

static RepBwdFn pick_repeat_bwd(const pbr_render_desc *d, bool loss) {
    const bool point = d->light_type == PBR_LIGHT_POINT, half_maps = d->map_dtype == PBR_F16, multi = d->n_lights > 1;
    switch ((point ? 3 : 0) + d->workflow) {
        case 0: return repeat_bwd_types<PBR_LIGHT_DIRECTIONAL, PBR_WORKFLOW_METALLIC>(half_maps, loss, multi);
        case 1: return repeat_bwd_types<PBR_LIGHT_DIRECTIONAL, PBR_WORKFLOW_SPECULAR>(half_maps, loss, multi);
        case 2: return repeat_bwd_types<PBR_LIGHT_DIRECTIONAL, PBR_WORKFLOW_CONVERTED>(half_maps, loss, multi);
        case 3: return repeat_bwd_types<PBR_LIGHT_POINT, PBR_WORKFLOW_METALLIC>(half_maps, loss, multi);
        case 4: return repeat_bwd_types<PBR_LIGHT_POINT, PBR_WORKFLOW_SPECULAR>(half_maps, loss, multi);
        default: return repeat_bwd_types<PBR_LIGHT_POINT, PBR_WORKFLOW_CONVERTED>(half_maps, loss, multi);
    }
}

// The launches the repeat-inner backward serves: what the forward's repeat-inner kernel serves (map rows a whole number of
// 4-texel lanes, an output band that holds a full period of the map's rows) with planes small enough for its 32-bit lane offsets
// when it addresses them through a scalar base.  fp32 result / upstream gradient only (as every backward entry).
bool repeat_backward_serves(const pbr_render_desc *d) {
    return repeat_inner(d) && d->out_dtype == PBR_F32 && g_max_vec >= 2;
}
// ... and the rendering-loss step through it: one light (the two passes over several lights are built for the gradient only)
//     (its squared-difference sum counts every pixel once: no overlapping last lanes, i.e. map rows of whole pairs; bands of at least a period)
bool repeat_loss_serves(const pbr_render_desc *d) {
    return repeat_backward_serves(d) && d->n_lights == 1 && d->map_width % 4 == 0 && !repeat_thin_band(d);
}

// Workgroups of the launch for this descriptor (one partial sum each under the loss policy), -1 when it does not fit a 1-D grid.
int64_t repeat_backward_tiles(const pbr_render_desc *d) {
    KArgs k;
    pbr_render_desc g = *d;
    g.height = g.height_total = d->map_height; g.width = d->map_width; g.map_height = g.map_width = 0; g.y_offset = 0;
    fill_args(&g, 2, k, 6);
    return k.n_tiles;
}

// KArgs of the walk over the SOURCE maps for this descriptor (the output described through rep / out_* / y_offset / H_total).
static int fill_repeat_backward(const pbr_render_desc *d, KArgs &k) {
    KArgs full;
    fill_args(d, 4, full);                               // the output's point-light grid, view, light, flags
    pbr_render_desc g = *d;
    const bool thin = repeat_thin_band(d);               // a band thinner than a period: the cyclic window of source rows it touches (KArgs::win_y0)
    g.height = g.height_total = thin ? d->height : d->map_height; g.width = d->map_width; g.map_height = g.map_width = 0; g.y_offset = 0;
    g.out_batch_stride = g.out_channel_stride = 0;
    fill_args(&g, 2, k, 6);                              // one-wave workgroups over the source maps, two texels per lane
    if (k.n_tiles < 0) return PBR_ERR_SHAPE;
    // ONE directional light: the kernel sums the repeats' upstream values and differentiates once -- it is bound by its 8-byte streams (3 upstream
    // planes x n^2 positions, 8 map planes, 8 gradient planes per wave), and those stream best when every XCD walks long runs of tiles: 2048^2
    // tile(2) fp32, runs of 64 (the forward's rule) / 512 / 1024 / 4096 tiles: 86.4 / 79.7 / 76.3 / 76.1 us (tools/repeat_bwd_probe.py --schedule).
    // The point-light form is VALU-bound and does not care (144-147 us whatever the order).
    if (d->schedule == PBR_SCHEDULE_AUTO && d->light_type == PBR_LIGHT_DIRECTIONAL && d->n_lights == 1) {
        k.xcd_log2 = 10;
        k.xcd_tiles = (k.n_tiles >> (k.xcd_log2 + 3)) << (k.xcd_log2 + 3);
    }
    k.x0 = full.x0; k.x1 = full.x1; k.xstep = full.xstep;
    k.y0 = full.y0; k.y1 = full.y1; k.ystep = full.ystep;
    k.rep_y = d->height_total / d->map_height; k.rep_x = d->width / d->map_width;
    k.out_W = d->width; k.out_Ht = d->height_total;
    k.y_offset = d->y_offset; k.H_total = d->height;     // `upstream` holds the rows [y_offset, y_offset + height) of the tiled image
    k.map_h = d->map_height; k.win_y0 = thin ? d->y_offset % d->map_height : 0;
    k.o_cs = (int64_t)d->map_height * d->map_width; k.o_bs = 3 * k.o_cs;      // the gradient planes: dense, map-sized
    if ((int64_t)d->height * d->width >= (1ll << 30)) k.sbase = 0;           // the lane's offset inside the output's first repeat must fit 32 bits of bytes
    return PBR_OK;
}

// `upstream`: the gradient w.r.t. the output, or (loss != nullptr) the target image; [B][3][d->height][d->width] fp32 contiguous.
// g_*: MAP-sized ([B][C][map_height][map_width], dense), in the maps' storage type.
int launch_repeat_backward(const pbr_render_desc *d, const void *upstream, void *g_albedo, void *g_normal, void *g_roughness, void *g_metallic,
                           void *g_specular, bool loss, float scale, float *partials, hipStream_t st) {
    KArgs k;
    const int rc = fill_repeat_backward(d, k);
    if (rc != PBR_OK) return rc;
    const BArgs b = {nullptr, g_albedo, g_normal, g_roughness, g_metallic, g_specular, nullptr};
    const RBArgs rb = {static_cast<const float *>(upstream), (int64_t)d->height * d->width, scale, partials};
    hipLaunchKernelGGL(pick_repeat_bwd(d, loss), dim3((unsigned)k.n_tiles, 1, 1), dim3(64, 1, 1), 0, st, k, b, rb);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? PBR_OK : 1000 + (int)e;
}

// pbr_cook_torrance_blend_backward over TILED maps (round 6): what the one-kernel folded backward serves, with one light, fp32 maps.
bool repeat_blend_backward_serves(const pbr_render_desc *d) {
    return repeat_backward_serves(d) && d->n_lights == 1 && d->map_dtype == PBR_F32 && d->normal.data != nullptr && !repeat_thin_band(d);
}
// kblend: ct_blend.hpp's KBlend (the second material, the mask, the normal-sign flags); g1: BArgs with material 1's MAP-sized gradient planes
// (gout unused); g2: BBlend with material 2's and the mask's.  `grad_out`: [B][3][d->height][d->width] fp32 contiguous.
int launch_repeat_blend_backward(const pbr_render_desc *d, const void *kblend, const void *grad_out, const void *g1, const void *g2, hipStream_t st) {
    KArgs k;
    const int rc = fill_repeat_backward(d, k);
    if (rc != PBR_OK) return rc;
    k.sbase = 0;                                          // (the second material and the mask are addressed per lane, as in the untiled blend backward)
    const RBArgs rb = {static_cast<const float *>(grad_out), (int64_t)d->height * d->width, 0.0f, nullptr};
    void (*fn)(const KArgs, const KBlend, const BArgs, const BBlend, const RBArgs) = nullptr;
    const bool point = d->light_type == PBR_LIGHT_POINT;
    switch ((point ? 3 : 0) + d->workflow) {
        case 0: fn = cook_torrance_repeat_blend_backward_kernel<PBR_LIGHT_DIRECTIONAL, PBR_WORKFLOW_METALLIC>; break;
        case 1: fn = cook_torrance_repeat_blend_backward_kernel<PBR_LIGHT_DIRECTIONAL, PBR_WORKFLOW_SPECULAR>; break;
        case 2: fn = cook_torrance_repeat_blend_backward_kernel<PBR_LIGHT_DIRECTIONAL, PBR_WORKFLOW_CONVERTED>; break;
        case 3: fn = cook_torrance_repeat_blend_backward_kernel<PBR_LIGHT_POINT, PBR_WORKFLOW_METALLIC>; break;
        case 4: fn = cook_torrance_repeat_blend_backward_kernel<PBR_LIGHT_POINT, PBR_WORKFLOW_SPECULAR>; break;
        default: fn = cook_torrance_repeat_blend_backward_kernel<PBR_LIGHT_POINT, PBR_WORKFLOW_CONVERTED>; break;
    }
    hipLaunchKernelGGL(fn, dim3((unsigned)k.n_tiles, 1, 1), dim3(64, 1, 1), 0, st, k, *static_cast<const KBlend *>(kblend), *static_cast<const BArgs *>(g1),
                       *static_cast<const BBlend *>(g2), rb);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? PBR_OK : 1000 + (int)e;
}

static size_t folded_fallback_bytes(const pbr_render_desc *d) {
    // output-sized gradients of every map the descriptor holds, in the maps' storage type (the two-kernel form's intermediate)
    const size_t esz = d->map_dtype == PBR_F32 ? 4 : 2;
    const size_t planes = 3 + (d->normal.data ? 3 : 0) + 1 + (d->workflow == PBR_WORKFLOW_SPECULAR ? 3 : 1);
    return planes * (size_t)d->batch * (size_t)d->height * (size_t)d->width * esz;
}

}  // namespace pbr

extern "C" {

size_t pbr_backward_folded_workspace_bytes(const pbr_render_desc *d) {
    const pbr::TuningScope tuning(d);
    if (pbr::validate(d) != PBR_OK || !pbr::is_tiled(d)) return 0;
    return pbr::repeat_backward_serves(d) ? 0 : pbr::folded_fallback_bytes(d);
}

int pbr_cook_torrance_backward_folded(const pbr_render_desc *d, const void *grad_out, void *g_albedo, void *g_normal, void *g_roughness,
                                      void *g_metallic, void *g_specular, void *workspace, void *stream) {
    const pbr::TuningScope tuning(d);
    using namespace pbr;
    const int rc = validate(d);
    if (rc != PBR_OK) return rc;
    if (!grad_out) return PBR_ERR_NULL_MAP;
    if (d->out_dtype != PBR_F32) return PBR_ERR_DTYPE;
    if (!is_tiled(d))                                         // nothing to fold: the gradients are map-sized as they come
        return pbr_cook_torrance_backward(d, grad_out, g_albedo, g_normal, g_roughness, g_metallic, g_specular, stream);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (repeat_backward_serves(d)) {
        if (repeat_thin_band(d)) {
            // the band touches only a window of the map's rows: the texels outside it have no repeat inside the band -- their partial sum is 0
            const size_t esz = d->map_dtype == PBR_F32 ? 4 : 2, plane = (size_t)d->batch * d->map_height * d->map_width * esz;
            void *const bufs[5] = {g_albedo, d->normal.data ? g_normal : nullptr, g_roughness, d->workflow == PBR_WORKFLOW_SPECULAR ? nullptr : g_metallic,
                                   d->workflow == PBR_WORKFLOW_SPECULAR ? g_specular : nullptr};
            const int chans[5] = {3, 3, 1, 1, 3};
            for (int i = 0; i < 5; ++i)
                if (bufs[i] && hipMemsetAsync(bufs[i], 0, chans[i] * plane, st) != hipSuccess) return 1000 + (int)hipGetLastError();
        }
        return launch_repeat_backward(d, grad_out, g_albedo, g_normal, g_roughness, g_metallic, g_specular, false, 0.0f, nullptr, st);
    }
    // the two-kernel form (map rows shorter than 4 texels; PBR_TUNE_TILE_REPEAT = 0): per-output-pixel gradients into the workspace, then the
    // sums (whole outputs only: a band's repeats are not whole)
    if (!workspace) return PBR_ERR_NULL_MAP;
    if (d->height != d->height_total) return PBR_ERR_UNSUPPORTED;
    const size_t esz = d->map_dtype == PBR_F32 ? 4 : 2;
    const size_t plane = (size_t)d->batch * d->height * d->width * esz;
    char *w = static_cast<char *>(workspace);
    void *const wanted[5] = {g_albedo, d->normal.data ? g_normal : nullptr, g_roughness,
                             d->workflow == PBR_WORKFLOW_SPECULAR ? nullptr : g_metallic, d->workflow == PBR_WORKFLOW_SPECULAR ? g_specular : nullptr};
    const int channels[5] = {3, 3, 1, 1, 3};
    void *big[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    for (int i = 0; i < 5; ++i)
        if (wanted[i]) { big[i] = w; w += (size_t)channels[i] * plane; }
    int e = pbr_cook_torrance_backward(d, grad_out, big[0], big[1], big[2], big[3], big[4], stream);
    for (int i = 0; i < 5 && e == PBR_OK; ++i)
        if (wanted[i])
            e = pbr_fold_gradient_typed(big[i], wanted[i], d->batch, channels[i], d->map_height, d->map_width, d->height / d->map_height,
                                        d->width / d->map_width, 0, d->map_dtype, stream);
    return e;
}

}  // extern "C"
