// ct_batch.hip -- instantiations and selection of the batch-inner kernels (ct_kernel.hpp: cook_torrance_batch_kernel):
// several lights, NB materials per lane, light geometry computed once per pixel position.
#include "ct_launch.hpp"

namespace pbr {

template <int LIGHT, int WF, typename TI, typename TO>
static KernelFn batch_variant(int nb, bool nt) {
    if (nb == 4) return nt ? cook_torrance_batch_kernel<LIGHT, WF, TI, TO, 2, 4, true> : cook_torrance_batch_kernel<LIGHT, WF, TI, TO, 2, 4, false>;
    return nt ? cook_torrance_batch_kernel<LIGHT, WF, TI, TO, 2, 2, true> : cook_torrance_batch_kernel<LIGHT, WF, TI, TO, 2, 2, false>;
}

template <int LIGHT, int WF>
static KernelFn batch_types(int in_dt, int out_dt, int nb, bool nt) {
    if (in_dt == PBR_F32) return out_dt == PBR_F32 ? batch_variant<LIGHT, WF, float, float>(nb, nt) : batch_variant<LIGHT, WF, float, __half>(nb, nt);
    return out_dt == PBR_F32 ? batch_variant<LIGHT, WF, __half, float>(nb, nt) : batch_variant<LIGHT, WF, __half, __half>(nb, nt);
}

KernelFn pick_batch_kernel(const pbr_render_desc *d, int nb, bool nt) {
    const bool point = d->light_type == PBR_LIGHT_POINT;
    const int idt = d->map_dtype, odt = d->out_dtype;
    switch ((point ? 3 : 0) + d->workflow) {
        case 0: return batch_types<PBR_LIGHT_DIRECTIONAL, PBR_WORKFLOW_METALLIC>(idt, odt, nb, nt);
        case 1: return batch_types<PBR_LIGHT_DIRECTIONAL, PBR_WORKFLOW_SPECULAR>(idt, odt, nb, nt);
        case 2: return batch_types<PBR_LIGHT_DIRECTIONAL, PBR_WORKFLOW_CONVERTED>(idt, odt, nb, nt);
        case 3: return batch_types<PBR_LIGHT_POINT, PBR_WORKFLOW_METALLIC>(idt, odt, nb, nt);
        case 4: return batch_types<PBR_LIGHT_POINT, PBR_WORKFLOW_SPECULAR>(idt, odt, nb, nt);
        default: return batch_types<PBR_LIGHT_POINT, PBR_WORKFLOW_CONVERTED>(idt, odt, nb, nt);
    }
}

}  // namespace pbr
