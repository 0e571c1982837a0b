// ct_tiled.hip -- instantiations, selection and launch set-up of the repeat-inner kernel (ct_kernel.hpp:
// cook_torrance_repeat_kernel): MaterialBase.tile (/root/reference/pypbr/materials/base.py:524-537) fused into the
// evaluation with every texel loaded and decoded ONCE and evaluated at all its positions of the output.
#include "ct_launch.hpp"

namespace pbr {

// The streaming hint sits on the STORES only -- a rule since ABI 8 (tile_pack_probe.py of round 4, git 9ce0718:tools/, 2048^2 tile(2) -> 4096^2: fp32 maps 53.3 us
// against 59.6 with the hint on loads and stores, 65.9 with none; fp16 maps 52.3 / 53.6-56.1 / 51.8): the stores are the major stream (every byte
// written once, never read here); a lane's texel loads are few and short, and without the hint they stay clear of the write stream's path.
// The three other hint combinations were instantiations that only the closed experiment's knob reached: not built any more.
template <int LIGHT, int WF, typename TI, typename TO>
static KernelFn repeat_hints(bool multi) {
    return multi ? cook_torrance_repeat_kernel<LIGHT, WF, TI, TO, false, true, true> : cook_torrance_repeat_kernel<LIGHT, WF, TI, TO, false, true>;
}

template <int LIGHT, int WF>
static KernelFn repeat_types(int in_dt, int out_dt, bool multi) {
    if (in_dt == PBR_F32) return out_dt == PBR_F32 ? repeat_hints<LIGHT, WF, float, float>(multi) : repeat_hints<LIGHT, WF, float, __half>(multi);
    return out_dt == PBR_F32 ? repeat_hints<LIGHT, WF, __half, float>(multi) : repeat_hints<LIGHT, WF, __half, __half>(multi);
}

KernelFn pick_repeat_kernel(const pbr_render_desc *d) {
    const bool point = d->light_type == PBR_LIGHT_POINT;
    const int idt = d->map_dtype, odt = d->out_dtype;
    const bool multi = d->n_lights > 1;
    switch ((point ? 3 : 0) + d->workflow) {
        case 0: return repeat_types<PBR_LIGHT_DIRECTIONAL, PBR_WORKFLOW_METALLIC>(idt, odt, multi);
        case 1: return repeat_types<PBR_LIGHT_DIRECTIONAL, PBR_WORKFLOW_SPECULAR>(idt, odt, multi);
        case 2: return repeat_types<PBR_LIGHT_DIRECTIONAL, PBR_WORKFLOW_CONVERTED>(idt, odt, multi);
        case 3: return repeat_types<PBR_LIGHT_POINT, PBR_WORKFLOW_METALLIC>(idt, odt, multi);
        case 4: return repeat_types<PBR_LIGHT_POINT, PBR_WORKFLOW_SPECULAR>(idt, odt, multi);
        default: return repeat_types<PBR_LIGHT_POINT, PBR_WORKFLOW_CONVERTED>(idt, odt, multi);
    }
}

// The kernel's argument block: the grid of an UNTILED launch over the source maps (rows = B * map_height, W = map_width), the
// result's strides and the point-light grid of the full output.
void fill_repeat_args(const pbr_render_desc *d, KArgs &k) {
    KArgs full;
    fill_args(d, 4, full);                               // the output's point-light grid (x0 .. ystep), view, light, flags
    pbr_render_desc g = *d;
    const bool thin = repeat_thin_band(d);               // the walk covers the source rows the band touches: all of them, or a cyclic window
    g.height = g.height_total = thin ? d->height : d->map_height;
    g.width = d->map_width;
    g.map_height = g.map_width = 0;
    g.y_offset = 0;
    const int64_t plane = (int64_t)d->height * d->width;
    g.out_channel_stride = d->out_channel_stride ? d->out_channel_stride : plane;
    g.out_batch_stride = d->out_batch_stride ? d->out_batch_stride : 3 * g.out_channel_stride;
    fill_args(&g, 4, k);
    k.x0 = full.x0; k.x1 = full.x1; k.xstep = full.xstep;
    k.y0 = full.y0; k.y1 = full.y1; k.ystep = full.ystep;
    k.rep_y = d->height_total / d->map_height; k.rep_x = d->width / d->map_width;
    k.out_W = d->width; k.out_Ht = d->height_total;
    k.y_offset = d->y_offset; k.H_total = d->height;     // the rows [y_offset, y_offset + H_total) of the tiled image are what `out` holds (KArgs: out_Ht)
    k.map_h = d->map_height; k.win_y0 = thin ? d->y_offset % d->map_height : 0;
    if (plane >= (1ll << 30)) k.sbase = 0;               // the lane's offset inside the result's first repeat must fit 32 bits of bytes
}

}  // namespace pbr
