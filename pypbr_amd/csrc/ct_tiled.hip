// ct_tiled.hip -- instantiations, selection and launch set-up of the repeat-inner kernel (ct_kernel.hpp:
// cook_torrance_repeat_kernel): MaterialBase.tile (/root/reference/pypbr/materials/base.py:524-537) fused into the
// evaluation with every texel loaded and decoded ONCE and evaluated at all its positions of the output.
#include "ct_launch.hpp"

namespace pbr {

template <int LIGHT, int WF>
static KernelFn repeat_types(int in_dt, int out_dt, bool nt) {
    if (in_dt == PBR_F32) {
        if (out_dt == PBR_F32) return nt ? cook_torrance_repeat_kernel<LIGHT, WF, float, float, true> : cook_torrance_repeat_kernel<LIGHT, WF, float, float, false>;
        return nt ? cook_torrance_repeat_kernel<LIGHT, WF, float, __half, true> : cook_torrance_repeat_kernel<LIGHT, WF, float, __half, false>;
    }
    if (out_dt == PBR_F32) return nt ? cook_torrance_repeat_kernel<LIGHT, WF, __half, float, true> : cook_torrance_repeat_kernel<LIGHT, WF, __half, float, false>;
    return nt ? cook_torrance_repeat_kernel<LIGHT, WF, __half, __half, true> : cook_torrance_repeat_kernel<LIGHT, WF, __half, __half, false>;
}

KernelFn pick_repeat_kernel(const pbr_render_desc *d, bool nt) {
    const bool point = d->light_type == PBR_LIGHT_POINT;
    const int idt = d->map_dtype, odt = d->out_dtype;
    switch ((point ? 3 : 0) + d->workflow) {
        case 0: return repeat_types<PBR_LIGHT_DIRECTIONAL, PBR_WORKFLOW_METALLIC>(idt, odt, nt);
        case 1: return repeat_types<PBR_LIGHT_DIRECTIONAL, PBR_WORKFLOW_SPECULAR>(idt, odt, nt);
        case 2: return repeat_types<PBR_LIGHT_DIRECTIONAL, PBR_WORKFLOW_CONVERTED>(idt, odt, nt);
        case 3: return repeat_types<PBR_LIGHT_POINT, PBR_WORKFLOW_METALLIC>(idt, odt, nt);
        case 4: return repeat_types<PBR_LIGHT_POINT, PBR_WORKFLOW_SPECULAR>(idt, odt, nt);
        default: return repeat_types<PBR_LIGHT_POINT, PBR_WORKFLOW_CONVERTED>(idt, odt, nt);
    }
}

// The kernel's argument block: the grid of an UNTILED launch over the source maps (rows = B * map_height, W = map_width), the
// result's strides and the point-light grid of the full output.
void fill_repeat_args(const pbr_render_desc *d, KArgs &k) {
    KArgs full;
    fill_args(d, 4, full);                               // the output's point-light grid (x0 .. ystep), view, light, flags
    pbr_render_desc g = *d;
    g.height = g.height_total = d->map_height;
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
    if (plane >= (1ll << 30)) k.sbase = 0;               // the lane's offset inside the result's first repeat must fit 32 bits of bytes
}

}  // namespace pbr
