// ct_batch.hip -- instantiations and selection of the batch-inner kernels (ct_kernel.hpp: cook_torrance_batch_kernel):
// several lights, NB materials per lane, light geometry computed once per pixel position.
#include "ct_launch.hpp"

namespace pbr {

template <int LIGHT, int WF, typename TI, typename TO>
static KernelFn batch_variant(int nb) {        // (the streaming hint: a rule since ABI 8 -- these launches are never tiled)
    return nb == 4 ? cook_torrance_batch_kernel<LIGHT, WF, TI, TO, 2, 4, true> : cook_torrance_batch_kernel<LIGHT, WF, TI, TO, 2, 2, true>;
}

template <int LIGHT, int WF>
static KernelFn batch_types(int in_dt, int out_dt, int nb) {
    if (in_dt == PBR_F32) return out_dt == PBR_F32 ? batch_variant<LIGHT, WF, float, float>(nb) : batch_variant<LIGHT, WF, float, __half>(nb);
    return out_dt == PBR_F32 ? batch_variant<LIGHT, WF, __half, float>(nb) : batch_variant<LIGHT, WF, __half, __half>(nb);
}

KernelFn pick_batch_kernel(const pbr_render_desc *d, int nb) {
    const bool point = d->light_type == PBR_LIGHT_POINT;
    const int idt = d->map_dtype, odt = d->out_dtype;
    switch ((point ? 3 : 0) + d->workflow) {
        case 0: return batch_types<PBR_LIGHT_DIRECTIONAL, PBR_WORKFLOW_METALLIC>(idt, odt, nb);
        case 1: return batch_types<PBR_LIGHT_DIRECTIONAL, PBR_WORKFLOW_SPECULAR>(idt, odt, nb);
        case 2: return batch_types<PBR_LIGHT_DIRECTIONAL, PBR_WORKFLOW_CONVERTED>(idt, odt, nb);
        case 3: return batch_types<PBR_LIGHT_POINT, PBR_WORKFLOW_METALLIC>(idt, odt, nb);
        case 4: return batch_types<PBR_LIGHT_POINT, PBR_WORKFLOW_SPECULAR>(idt, odt, nb);
        default: return batch_types<PBR_LIGHT_POINT, PBR_WORKFLOW_CONVERTED>(idt, odt, nb);
    }
}

}  // namespace pbr
