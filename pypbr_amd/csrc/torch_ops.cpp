// torch_ops.cpp -- the PyTorch-ROCm face of libpbr_hip.so: TORCH_LIBRARY(pbr_hip, ...) operators over the C ABI
// (include/pbr_hip.h), as SURVEY.md 8b sketches ("pbr_hip::cook_torrance(Tensor albedo[B,3,H,W], ...)").
//
// Host-only C++: no device code lives here.  Every operator validates its tensors, fills the C-ABI descriptor and
// calls the same extern "C" entry point the ctypes binding calls, on torch's CURRENT HIP stream of the maps' device;
// outputs are allocated through ATen (torch is plumbing: device memory, streams, the dispatcher).  Registered for
// the CUDA dispatch key (= HIP on a ROCm build); fake (meta) kernels and the autograd formula are registered from
// Python (pypbr_amd/torch_ops.py: torch.library.register_fake / register_autograd), so torch.library.opcheck,
// torch.compile and FakeTensor tracing see complete operators.
//
// Reference behaviour replaced: pypbr/models/cooktorrance.py:92-182 (forward, and what autograd derives from it),
// pypbr/utils/functions.py:31-66, pypbr/materials/metallic.py:98-108, pypbr/materials/diffuse.py:128-147.
#include <ATen/ATen.h>
#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include <cstring>
#include <tuple>
#include <vector>

#include "../../include/pbr_hip.h"

namespace {

using at::Tensor;
using OptTensor = std::optional<Tensor>;

void check_status(int rc, const char *what) {
    if (rc == PBR_OK) return;
    const char *msg = pbr_error_string(rc);
    // same mapping as pypbr_amd._native.check: caller errors -> ValueError / TypeError / NotImplementedError
    if (rc == PBR_ERR_DTYPE) TORCH_CHECK_TYPE(false, what, ": ", msg);
    if (rc == PBR_ERR_UNSUPPORTED) TORCH_CHECK_NOT_IMPLEMENTED(false, what, ": ", msg);
    if (rc < 0) TORCH_CHECK_VALUE(false, what, ": ", msg);
    TORCH_CHECK(false, what, ": HIP error ", rc - 1000, " (", msg, ")");
}

int dtype_code(const Tensor &t, const char *name) {
    if (t.scalar_type() == at::kFloat) return PBR_F32;
    if (t.scalar_type() == at::kHalf) return PBR_F16;
    TORCH_CHECK_TYPE(false, name, " must be float32 or float16, got ", t.scalar_type());
}

// [B,C,H,W] on a HIP device, rows contiguous (planes / materials may be strided: the ABI takes element strides)
Tensor as_map(const Tensor &t, int64_t channels, const char *name) {
    TORCH_CHECK_VALUE(t.dim() == 4 && t.size(1) == channels, name, " must be [B,", channels, ",H,W], got ", t.sizes());
    TORCH_CHECK(t.is_cuda(), name, " must live on a ROCm device; pbr_hip has no CPU path");
    if (t.stride(3) != 1 || t.stride(2) != t.size(3)) return t.contiguous();
    return t;
}

pbr_map map_of(const OptTensor &t) {
    pbr_map m = {nullptr, 0, 0};
    if (t.has_value() && t->defined()) {
        m.data = t->data_ptr();
        m.batch_stride = t->size(0) > 1 ? t->stride(0) : 0;
        m.channel_stride = t->stride(1);
    }
    return m;
}

std::vector<float> host_floats(const Tensor &t, int64_t cols, const char *name) {
    TORCH_CHECK_VALUE(t.numel() % cols == 0 && t.numel() > 0, name, " must hold a multiple of ", cols, " values, got ", t.sizes());
    const Tensor h = t.detach().to(at::kCPU, at::kFloat).contiguous();      // device tensors: one small D2H copy
    return std::vector<float>(h.data_ptr<float>(), h.data_ptr<float>() + h.numel());
}

struct Prepared {
    pbr_render_desc d;
    Tensor albedo, roughness;
    OptTensor normal, metallic, specular;
    int64_t B, H, W;           // extent of the OUTPUT band
    // parameters that live on the device (ABI 5): read there, never copied to the host
    OptTensor dev_view, dev_lights, dev_intensities;
    Tensor param_block;
};

// view / light / intensity tensors on the maps' device stay there: pbr_prepare_device_params folds them into a block the kernels read
void use_device_parameters(Prepared &p, void *stream) {
    if (!p.dev_view.has_value() && !p.dev_lights.has_value() && !p.dev_intensities.has_value()) return;
    p.param_block = at::empty({(int64_t)((pbr_device_params_bytes() + 3) / 4)}, p.albedo.options().dtype(at::kFloat));
    auto ptr = [](const OptTensor &t) -> const void * { return t.has_value() ? t->data_ptr() : nullptr; };
    const int32_t rows = p.dev_intensities.has_value() ? (int32_t)(p.dev_intensities->numel() / 3) : 1;
    check_status(pbr_prepare_device_params(&p.d, ptr(p.dev_view), ptr(p.dev_lights), ptr(p.dev_intensities), rows, p.param_block.data_ptr(), stream),
                 "pbr_hip::prepare_device_params");
    p.d.device_params = p.param_block.data_ptr();
}

// Shared by forward and backward: validates the maps and fills everything of the descriptor except `out`.
Prepared prepare(const Tensor &albedo, const OptTensor &normal, const Tensor &roughness, const OptTensor &metallic,
                 const OptTensor &specular, const Tensor &view_dir, const Tensor &lights, const Tensor &intensities,
                 double light_size, int64_t light_type, bool albedo_is_srgb, bool specular_is_srgb, bool convert,
                 bool return_srgb, int64_t y_offset, int64_t height_total, int64_t tile_y, int64_t tile_x, int64_t rows) {
    Prepared p;
    std::memset(&p.d, 0, sizeof(p.d));
    p.albedo = as_map(albedo, 3, "albedo");
    p.roughness = as_map(roughness, 1, "roughness");
    if (normal.has_value() && normal->defined()) p.normal = as_map(*normal, 3, "normal");
    if (metallic.has_value() && metallic->defined()) p.metallic = as_map(*metallic, 1, "metallic");
    if (specular.has_value() && specular->defined()) p.specular = as_map(*specular, 3, "specular");
    TORCH_CHECK_VALUE(p.metallic.has_value() || p.specular.has_value(),
                      "Material must have either 'metallic' or 'specular' property.");          // cooktorrance.py:115-118
    TORCH_CHECK_VALUE(!(convert && !p.metallic.has_value()), "convert_to_diffuse_specular needs a metallic map");
    const int64_t B = p.albedo.size(0), H = p.albedo.size(2), W = p.albedo.size(3);
    auto same = [&](const OptTensor &t, const char *name) {
        if (!t.has_value()) return;
        TORCH_CHECK_VALUE(t->size(2) == H && t->size(3) == W && (t->size(0) == B || t->size(0) == 1), name, " ", t->sizes(),
                          " does not match albedo ", p.albedo.sizes());
        TORCH_CHECK_TYPE(t->scalar_type() == p.albedo.scalar_type() && t->device() == p.albedo.device(),
                         "all maps must share dtype and device (", name, ")");
    };
    same(p.normal, "normal"); same(OptTensor(p.roughness), "roughness"); same(p.metallic, "metallic"); same(p.specular, "specular");
    TORCH_CHECK_VALUE(light_type == PBR_LIGHT_DIRECTIONAL || light_type == PBR_LIGHT_POINT,
                      "Unsupported light_type: ", light_type, ". Must be 'directional' (0) or 'point' (1).");   // :62-65

    pbr_render_desc &d = p.d;
    d.abi_version = PBR_HIP_ABI_VERSION;
    TORCH_CHECK_VALUE(tile_y >= 1 && tile_x >= 1, "tile counts must be >= 1");
    if (tile_y == 1 && tile_x == 1) {
        TORCH_CHECK_VALUE(rows == 0, "`rows` selects a band of a tiled map; without tiling pass the band's own maps");
        d.batch = (int32_t)B; d.height = (int32_t)H; d.width = (int32_t)W;
        d.height_total = (int32_t)(height_total > 0 ? height_total : H);
    } else {       // MaterialBase.tile fused as wrap-around addressing (base.py:524-537)
        TORCH_CHECK_VALUE(height_total == 0 || height_total == tile_y * H, "with tiling the full map has ", tile_y * H, " rows");
        d.batch = (int32_t)B; d.height = (int32_t)(rows > 0 ? rows : tile_y * H - y_offset); d.width = (int32_t)(tile_x * W);
        d.height_total = (int32_t)(tile_y * H); d.map_height = (int32_t)H; d.map_width = (int32_t)W;
        TORCH_CHECK_VALUE(d.height >= 1 && y_offset + d.height <= d.height_total, "band outside the tiled map");
    }
    d.y_offset = (int32_t)y_offset;
    d.map_dtype = dtype_code(p.albedo, "maps");
    d.workflow = p.metallic.has_value() ? (convert ? PBR_WORKFLOW_CONVERTED : PBR_WORKFLOW_METALLIC) : PBR_WORKFLOW_SPECULAR;
    d.light_type = (int32_t)light_type;
    d.albedo_is_srgb = albedo_is_srgb; d.specular_is_srgb = specular_is_srgb; d.return_srgb = return_srgb;
    d.albedo = map_of(p.albedo); d.normal = map_of(p.normal); d.roughness = map_of(p.roughness);
    d.metallic = map_of(p.metallic);
    d.specular = d.workflow == PBR_WORKFLOW_SPECULAR ? map_of(p.specular) : pbr_map{nullptr, 0, 0};
    auto on_device = [&](const Tensor &t, int64_t cols, const char *name) {
        if (!t.is_cuda()) return false;
        TORCH_CHECK_VALUE(t.numel() % cols == 0 && t.numel() > 0, name, " must hold a multiple of ", cols, " values, got ", t.sizes());
        return true;
    };
    // a parameter on ANOTHER GPU than the maps is brought over, as the reference's `.to(device)` does (cooktorrance.py:95-96) and as the
    // ctypes plan path does (functional._device_parameter_tensors): the same call must not depend on which binding serves it
    auto here = [&](const Tensor &t) { return t.detach().to(p.albedo.device(), at::kFloat).contiguous(); };
    std::vector<float> v(3, 0.0f), l, it;
    if (on_device(view_dir, 3, "view_dir")) p.dev_view = here(view_dir); else v = host_floats(view_dir, 3, "view_dir");
    TORCH_CHECK_VALUE(view_dir.numel() == 3, "view_dir must have 3 components");
    if (on_device(lights, 3, "lights")) { p.dev_lights = here(lights); l.assign((size_t)lights.numel(), 0.0f); }
    else l = host_floats(lights, 3, "lights");
    if (on_device(intensities, 3, "intensities")) { p.dev_intensities = here(intensities); it.assign((size_t)intensities.numel(), 0.0f); }
    else it = host_floats(intensities, 3, "intensities");
    const size_t L = l.size() / 3;
    TORCH_CHECK_VALUE(L >= 1 && L <= PBR_MAX_LIGHTS, "between 1 and ", PBR_MAX_LIGHTS, " lights are supported, got ", L);
    if (it.size() == 3 && L > 1) { it.resize(3 * L); for (size_t i = 1; i < L; ++i) for (int c = 0; c < 3; ++c) it[3 * i + c] = it[c]; }
    TORCH_CHECK_VALUE(it.size() == 3 * L, "lights [", L, ",3] and intensities [", it.size() / 3, ",3] disagree");
    d.n_lights = (int32_t)L;
    for (int c = 0; c < 3; ++c) d.view_dir[c] = v[c];
    for (size_t i = 0; i < L; ++i)
        for (int c = 0; c < 3; ++c) { d.lights[i][c] = l[3 * i + c]; d.intensities[i][c] = it[3 * i + c]; }
    d.light_size = (float)light_size;                                       // 0 = falsy -> 1.0 inside; negative / NaN pass through (cooktorrance.py:130)
    d.schedule = PBR_SCHEDULE_AUTO;
    p.B = B; p.H = d.height; p.W = d.width;
    return p;
}

void *current_stream(const Tensor &t) { return c10::hip::getCurrentHIPStream(t.device().index()).stream(); }

// ---------------------------------------------------------------------------------------------- operators
Tensor cook_torrance(const Tensor &albedo, const OptTensor &normal, const Tensor &roughness, const OptTensor &metallic,
                     const OptTensor &specular, const Tensor &view_dir, const Tensor &lights, const Tensor &intensities,
                     double light_size, int64_t light_type, bool albedo_is_srgb, bool specular_is_srgb, bool convert,
                     bool return_srgb, int64_t y_offset, int64_t height_total, int64_t tile_y, int64_t tile_x, int64_t rows,
                     bool half_result) {
    Prepared p = prepare(albedo, normal, roughness, metallic, specular, view_dir, lights, intensities, light_size, light_type,
                         albedo_is_srgb, specular_is_srgb, convert, return_srgb, y_offset, height_total, tile_y, tile_x, rows);
    const c10::DeviceGuard guard(p.albedo.device());
    Tensor out = at::empty({p.B, 3, p.H, p.W}, p.albedo.options().dtype(half_result ? at::kHalf : at::kFloat));
    p.d.out = out.data_ptr();
    p.d.out_dtype = half_result ? PBR_F16 : PBR_F32;
    use_device_parameters(p, current_stream(out));
    check_status(pbr_cook_torrance(&p.d, current_stream(out)), "pbr_hip::cook_torrance");
    return out;
}

// Gradients w.r.t. the maps (in the maps' storage type) and, when `want_params`, [3 + 6 L] floats: d/d view_dir | d/d lights |
// d/d intensities.  Unwanted gradients come back as empty (0-element) tensors.  Untiled maps: OUTPUT-sized (the sum over a batch that
// shares a map is pbr_hip::fold_gradient).  TILED maps without light / view gradients and without batch-shared maps: MAP-sized, every
// texel's sum over its repeats (pbr_cook_torrance_backward_folded: one kernel that walks the maps, or backward + fold through a workspace);
// with light / view gradients or shared maps OUTPUT-sized as before (the autograd formula folds: pypbr_amd/torch_ops.py).
std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor, Tensor> cook_torrance_backward(
    const Tensor &grad_out, const Tensor &albedo, const OptTensor &normal, const Tensor &roughness, const OptTensor &metallic,
    const OptTensor &specular, const Tensor &view_dir, const Tensor &lights, const Tensor &intensities, double light_size,
    int64_t light_type, bool albedo_is_srgb, bool specular_is_srgb, bool convert, bool return_srgb, int64_t y_offset,
    int64_t height_total, int64_t tile_y, int64_t tile_x, int64_t rows, bool want_albedo, bool want_normal, bool want_roughness,
    bool want_metallic, bool want_specular, bool want_params) {
    Prepared p = prepare(albedo, normal, roughness, metallic, specular, view_dir, lights, intensities, light_size, light_type,
                         albedo_is_srgb, specular_is_srgb, convert, return_srgb, y_offset, height_total, tile_y, tile_x, rows);
    const c10::DeviceGuard guard(p.albedo.device());
    TORCH_CHECK_VALUE(grad_out.numel() == p.B * 3 * p.H * p.W, "grad_out ", grad_out.sizes(), " does not match the result [", p.B,
                      ",3,", p.H, ",", p.W, "]");
    const Tensor g = grad_out.to(p.albedo.device(), at::kFloat).contiguous();
    p.d.out = const_cast<void *>(g.data_ptr());      // ignored by the backward entry points; must be non-NULL to validate
    p.d.out_dtype = PBR_F32;
    const auto opts = p.albedo.options();
    const bool tiled = tile_y != 1 || tile_x != 1;
    auto lone = [&](const OptTensor &t) { return t.has_value() && t->size(0) == 1 && p.B > 1; };
    const bool shared = lone(OptTensor(p.albedo)) || lone(p.normal) || lone(OptTensor(p.roughness)) || lone(p.metallic) || lone(p.specular);
    const bool folded = tiled && !want_params && !shared;
    const int64_t gh = folded ? p.albedo.size(2) : p.H, gw = folded ? p.albedo.size(3) : p.W;
    auto buf = [&](bool want, int64_t c) { return want ? at::empty({p.B, c, gh, gw}, opts) : at::empty({0}, opts); };
    Tensor ga = buf(want_albedo, 3), gn = buf(want_normal && p.normal.has_value(), 3), gr = buf(want_roughness, 1);
    Tensor gm = buf(want_metallic && p.d.workflow != PBR_WORKFLOW_SPECULAR, 1);
    Tensor gs = buf(want_specular && p.d.workflow == PBR_WORKFLOW_SPECULAR, 3);
    auto ptr = [](const Tensor &t) -> void * { return t.numel() ? t.data_ptr() : nullptr; };
    Tensor gp = at::empty({0}, opts.dtype(at::kFloat));
    void *stream = current_stream(g);
    use_device_parameters(p, stream);
    if (want_params) {
        gp = at::empty({3 + 6 * (int64_t)p.d.n_lights}, opts.dtype(at::kFloat));
        Tensor ws = at::empty({(int64_t)(pbr_param_grad_workspace_bytes(&p.d) / 4 + 1)}, opts.dtype(at::kFloat));
        check_status(pbr_cook_torrance_backward_params(&p.d, g.data_ptr(), ptr(ga), ptr(gn), ptr(gr), ptr(gm), ptr(gs), gp.data_ptr(),
                                                       ws.data_ptr(), stream), "pbr_hip::cook_torrance_backward");
    } else if (folded) {
        const size_t ws_bytes = pbr_backward_folded_workspace_bytes(&p.d);
        Tensor ws = at::empty({(int64_t)ws_bytes}, opts.dtype(at::kByte));
        check_status(pbr_cook_torrance_backward_folded(&p.d, g.data_ptr(), ptr(ga), ptr(gn), ptr(gr), ptr(gm), ptr(gs),
                                                       ws_bytes ? ws.data_ptr() : nullptr, stream), "pbr_hip::cook_torrance_backward (folded)");
    } else {
        check_status(pbr_cook_torrance_backward(&p.d, g.data_ptr(), ptr(ga), ptr(gn), ptr(gr), ptr(gm), ptr(gs), stream),
                     "pbr_hip::cook_torrance_backward");
    }
    return {ga, gn, gr, gm, gs, gp};
}

// src [B,C,ny*h,nx*w] fp32 | fp16 -> [fold_batch ? 1 : B, C, h, w] of the same type: the sums autograd performs for a repeat() / a broadcast
// (formed in fp32, rounded once for fp16 gradients).
Tensor fold_gradient(const Tensor &src, int64_t h, int64_t w, bool fold_batch) {
    TORCH_CHECK_VALUE(src.dim() == 4 && src.is_cuda(), "fold_gradient needs a [B,C,H,W] device tensor");
    TORCH_CHECK_VALUE(h >= 1 && w >= 1 && src.size(2) % h == 0 && src.size(3) % w == 0, "whole repeats only");
    const Tensor s = src.contiguous();
    const c10::DeviceGuard guard(s.device());
    Tensor dst = at::empty({fold_batch ? 1 : s.size(0), s.size(1), h, w}, s.options());
    check_status(pbr_fold_gradient_typed(s.data_ptr(), dst.data_ptr(), (int32_t)s.size(0), (int32_t)s.size(1), (int32_t)h, (int32_t)w,
                                         (int32_t)(s.size(2) / h), (int32_t)(s.size(3) / w), fold_batch ? 1 : 0, dtype_code(s, "src"),
                                         current_stream(s)), "pbr_hip::fold_gradient");
    return dst;
}

Tensor colour(const Tensor &x, bool to_linear) {
    TORCH_CHECK(x.is_cuda(), "pbr_hip colour transfer needs a tensor on a ROCm device; there is no CPU path");
    const Tensor t = x.contiguous();
    const int dt = dtype_code(t, "texture");
    const c10::DeviceGuard guard(t.device());
    Tensor out = at::empty_like(t);
    const int rc = to_linear ? pbr_srgb_to_linear(t.data_ptr(), out.data_ptr(), (size_t)t.numel(), dt, current_stream(t))
                             : pbr_linear_to_srgb(t.data_ptr(), out.data_ptr(), (size_t)t.numel(), dt, current_stream(t));
    check_status(rc, to_linear ? "pbr_hip::srgb_to_linear" : "pbr_hip::linear_to_srgb");
    return out;
}
Tensor srgb_to_linear(const Tensor &x) { return colour(x, true); }
Tensor linear_to_srgb(const Tensor &x) { return colour(x, false); }

std::tuple<Tensor, Tensor> metallic_to_diffuse_specular(const Tensor &albedo, const Tensor &metallic, bool albedo_is_srgb) {
    TORCH_CHECK(albedo.is_cuda() && metallic.is_cuda(), "pbr_hip::metallic_to_diffuse_specular needs tensors on a ROCm device");
    TORCH_CHECK_VALUE(albedo.dim() >= 3 && metallic.dim() == albedo.dim() && albedo.size(-3) == 3 && metallic.size(-3) == 1 &&
                      albedo.size(-1) == metallic.size(-1) && albedo.size(-2) == metallic.size(-2) &&
                      albedo.numel() == 3 * metallic.numel(), "albedo [..,3,H,W] / metallic [..,1,H,W] expected");
    const Tensor a = albedo.contiguous(), m = metallic.to(albedo.scalar_type()).contiguous();
    const c10::DeviceGuard guard(a.device());
    Tensor diffuse = at::empty_like(a), spec = at::empty_like(a);
    const int64_t P = a.size(-1) * a.size(-2);
    check_status(pbr_metallic_to_specular(a.data_ptr(), m.data_ptr(), diffuse.data_ptr(), spec.data_ptr(), (int32_t)(a.numel() / (3 * P)), P,
                                          albedo_is_srgb, dtype_code(a, "albedo"), current_stream(a)), "pbr_hip::metallic_to_diffuse_specular");
    return {diffuse, spec};
}

std::tuple<Tensor, Tensor> diffuse_specular_to_basecolor_metallic(const Tensor &diffuse, const Tensor &specular, bool albedo_is_srgb) {
    TORCH_CHECK(diffuse.is_cuda() && specular.is_cuda(), "pbr_hip::diffuse_specular_to_basecolor_metallic needs tensors on a ROCm device");
    TORCH_CHECK_VALUE(diffuse.sizes() == specular.sizes(), "diffuse and specular must have the same shape");
    const Tensor d = diffuse.contiguous(), s = specular.to(diffuse.scalar_type()).contiguous();
    const c10::DeviceGuard guard(d.device());
    Tensor base = at::empty_like(d), met = at::empty_like(d);
    check_status(pbr_specular_to_metallic(d.data_ptr(), s.data_ptr(), base.data_ptr(), met.data_ptr(), (size_t)d.numel(), albedo_is_srgb,
                                          dtype_code(d, "diffuse"), current_stream(d)), "pbr_hip::diffuse_specular_to_basecolor_metallic");
    return {base, met};
}

// ---- gradients of the map ops (pbr_*_backward): registered as operators of their own so that the autograd formulas
// (pypbr_amd/torch_ops.py) stay traceable.  Gradients travel in the maps' storage type.
Tensor colour_backward(const Tensor &x, const Tensor &grad_out, bool to_linear) {
    TORCH_CHECK(x.is_cuda() && grad_out.is_cuda(), "pbr_hip colour transfer gradients need tensors on a ROCm device");
    TORCH_CHECK_VALUE(x.sizes() == grad_out.sizes(), "texture and grad_out must have the same shape");
    const Tensor t = x.contiguous(), g = grad_out.to(x.scalar_type()).contiguous();
    const int dt = dtype_code(t, "texture");
    const c10::DeviceGuard guard(t.device());
    Tensor gin = at::empty_like(t);
    const int rc = to_linear ? pbr_srgb_to_linear_backward(t.data_ptr(), g.data_ptr(), gin.data_ptr(), (size_t)t.numel(), dt, current_stream(t))
                             : pbr_linear_to_srgb_backward(t.data_ptr(), g.data_ptr(), gin.data_ptr(), (size_t)t.numel(), dt, current_stream(t));
    check_status(rc, "pbr_hip::colour_backward");
    return gin;
}

std::tuple<Tensor, Tensor> metallic_to_diffuse_specular_backward(const Tensor &albedo, const Tensor &metallic, const OptTensor &g_diffuse,
                                                                 const OptTensor &g_specular, bool albedo_is_srgb) {
    const Tensor a = albedo.contiguous(), m = metallic.to(albedo.scalar_type()).contiguous();
    const c10::DeviceGuard guard(a.device());
    auto opt = [&](const OptTensor &t) { return t.has_value() && t->defined() ? t->to(a.scalar_type()).contiguous() : Tensor(); };
    const Tensor gd = opt(g_diffuse), gs = opt(g_specular);
    Tensor ga = at::empty_like(a), gm = at::empty_like(m);
    const int64_t P = a.size(-1) * a.size(-2);
    check_status(pbr_metallic_to_specular_backward(a.data_ptr(), m.data_ptr(), gd.defined() ? gd.data_ptr() : nullptr,
                                                   gs.defined() ? gs.data_ptr() : nullptr, ga.data_ptr(), gm.data_ptr(),
                                                   (int32_t)(a.numel() / (3 * P)), P, albedo_is_srgb, dtype_code(a, "albedo"), current_stream(a)),
                 "pbr_hip::metallic_to_diffuse_specular_backward");
    return {ga, gm};
}

std::tuple<Tensor, Tensor> diffuse_specular_to_basecolor_metallic_backward(const Tensor &diffuse, const Tensor &specular, const OptTensor &g_basecolor,
                                                                           const OptTensor &g_metallic, bool albedo_is_srgb) {
    const Tensor d = diffuse.contiguous(), s = specular.to(diffuse.scalar_type()).contiguous();
    const c10::DeviceGuard guard(d.device());
    auto opt = [&](const OptTensor &t) { return t.has_value() && t->defined() ? t->to(d.scalar_type()).contiguous() : Tensor(); };
    const Tensor gb = opt(g_basecolor), gm = opt(g_metallic);
    Tensor gd = at::empty_like(d), gs = at::empty_like(s);
    check_status(pbr_specular_to_metallic_backward(d.data_ptr(), s.data_ptr(), gb.defined() ? gb.data_ptr() : nullptr,
                                                   gm.defined() ? gm.data_ptr() : nullptr, gd.data_ptr(), gs.data_ptr(), (size_t)d.numel(),
                                                   albedo_is_srgb, dtype_code(d, "diffuse"), current_stream(d)),
                 "pbr_hip::diffuse_specular_to_basecolor_metallic_backward");
    return {gd, gs};
}

// MaterialBase.resize for one map (base.py:490-504): [..., H, W] fp32 -> [..., h_out, w_out]
Tensor resize(const Tensor &texture, int64_t h_out, int64_t w_out, bool antialias) {
    TORCH_CHECK(texture.is_cuda(), "pbr_hip::resize needs a tensor on a ROCm device; there is no CPU path");
    TORCH_CHECK_TYPE(texture.scalar_type() == at::kFloat, "pbr_hip::resize supports float32 maps, got ", texture.scalar_type());
    TORCH_CHECK_VALUE(texture.dim() >= 2 && h_out >= 1 && w_out >= 1, "resize needs [..., H, W] and a positive size");
    const Tensor t = texture.contiguous();
    const c10::DeviceGuard guard(t.device());
    const int64_t h = t.size(-2), w = t.size(-1), planes = t.numel() / (h * w);
    std::vector<int64_t> shape(t.sizes().begin(), t.sizes().end());
    shape[shape.size() - 2] = h_out; shape[shape.size() - 1] = w_out;
    Tensor out = at::empty(shape, t.options());
    Tensor ws = at::empty({(int64_t)(pbr_resize_workspace_bytes(planes, (int32_t)h, (int32_t)w_out) / 4 + 1)}, t.options());
    check_status(pbr_resize_bilinear(t.data_ptr(), out.data_ptr(), planes, (int32_t)h, (int32_t)w, (int32_t)h_out, (int32_t)w_out, antialias,
                                     ws.data_ptr(), current_stream(t)), "pbr_hip::resize");
    return out;
}

Tensor resize_backward(const Tensor &grad_out, int64_t h_in, int64_t w_in, bool antialias) {
    TORCH_CHECK(grad_out.is_cuda(), "pbr_hip::resize_backward needs a tensor on a ROCm device");
    const Tensor g = grad_out.to(at::kFloat).contiguous();
    const c10::DeviceGuard guard(g.device());
    const int64_t ho = g.size(-2), wo = g.size(-1), planes = g.numel() / (ho * wo);
    std::vector<int64_t> shape(g.sizes().begin(), g.sizes().end());
    shape[shape.size() - 2] = h_in; shape[shape.size() - 1] = w_in;
    Tensor gin = at::empty(shape, g.options());
    Tensor ws = at::empty({(int64_t)(pbr_resize_backward_workspace_bytes(planes, (int32_t)h_in, (int32_t)w_in, (int32_t)ho, (int32_t)wo) / 4 + 1)}, g.options());
    check_status(pbr_resize_bilinear_backward(g.data_ptr(), gin.data_ptr(), planes, (int32_t)h_in, (int32_t)w_in, (int32_t)ho, (int32_t)wo, antialias,
                                              ws.data_ptr(), current_stream(g)), "pbr_hip::resize_backward");
    return gin;
}

}  // namespace

TORCH_LIBRARY(pbr_hip, m) {
    m.def("cook_torrance(Tensor albedo, Tensor? normal, Tensor roughness, Tensor? metallic, Tensor? specular, Tensor view_dir, "
          "Tensor lights, Tensor intensities, float light_size, int light_type, bool albedo_is_srgb, bool specular_is_srgb, "
          "bool convert_to_diffuse_specular, bool return_srgb, int y_offset=0, int height_total=0, int tile_y=1, int tile_x=1, "
          "int rows=0, bool half_result=False) -> Tensor");
    m.def("cook_torrance_backward(Tensor grad_out, Tensor albedo, Tensor? normal, Tensor roughness, Tensor? metallic, Tensor? specular, "
          "Tensor view_dir, Tensor lights, Tensor intensities, float light_size, int light_type, bool albedo_is_srgb, "
          "bool specular_is_srgb, bool convert_to_diffuse_specular, bool return_srgb, int y_offset, int height_total, int tile_y, "
          "int tile_x, int rows, bool want_albedo, bool want_normal, bool want_roughness, bool want_metallic, bool want_specular, "
          "bool want_params) -> (Tensor, Tensor, Tensor, Tensor, Tensor, Tensor)");
    m.def("fold_gradient(Tensor src, int h, int w, bool fold_batch) -> Tensor");
    m.def("srgb_to_linear(Tensor texture) -> Tensor");
    m.def("linear_to_srgb(Tensor texture) -> Tensor");
    m.def("metallic_to_diffuse_specular(Tensor albedo, Tensor metallic, bool albedo_is_srgb) -> (Tensor, Tensor)");
    m.def("diffuse_specular_to_basecolor_metallic(Tensor diffuse, Tensor specular, bool albedo_is_srgb) -> (Tensor, Tensor)");
    m.def("colour_backward(Tensor texture, Tensor grad_out, bool to_linear) -> Tensor");
    m.def("metallic_to_diffuse_specular_backward(Tensor albedo, Tensor metallic, Tensor? g_diffuse, Tensor? g_specular, bool albedo_is_srgb) "
          "-> (Tensor, Tensor)");
    m.def("diffuse_specular_to_basecolor_metallic_backward(Tensor diffuse, Tensor specular, Tensor? g_basecolor, Tensor? g_metallic, "
          "bool albedo_is_srgb) -> (Tensor, Tensor)");
    m.def("resize(Tensor texture, int h_out, int w_out, bool antialias) -> Tensor");
    m.def("resize_backward(Tensor grad_out, int h_in, int w_in, bool antialias) -> Tensor");
}

TORCH_LIBRARY_IMPL(pbr_hip, CUDA, m) {       // the CUDA dispatch key is the HIP device on a ROCm build of torch
    m.impl("cook_torrance", &cook_torrance);
    m.impl("cook_torrance_backward", &cook_torrance_backward);
    m.impl("fold_gradient", &fold_gradient);
    m.impl("srgb_to_linear", &srgb_to_linear);
    m.impl("linear_to_srgb", &linear_to_srgb);
    m.impl("metallic_to_diffuse_specular", &metallic_to_diffuse_specular);
    m.impl("diffuse_specular_to_basecolor_metallic", &diffuse_specular_to_basecolor_metallic);
    m.impl("colour_backward", &colour_backward);
    m.impl("metallic_to_diffuse_specular_backward", &metallic_to_diffuse_specular_backward);
    m.impl("diffuse_specular_to_basecolor_metallic_backward", &diffuse_specular_to_basecolor_metallic_backward);
    m.impl("resize", &resize);
    m.impl("resize_backward", &resize_backward);
}
