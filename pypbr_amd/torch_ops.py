"""`torch.ops.pbr_hip.*`: the C ABI of libpbr_hip.so registered as PyTorch operators (north_star's
"surfaced to Python via a PyTorch-ROCm C++/HIP extension").

The operators themselves are C++ (pypbr_amd/csrc/torch_ops.cpp -> pypbr_amd/_pbr_torch_ops.so, built by
`__graft_entry__.build()` / `make -C pypbr_amd/csrc torch_ops`): TORCH_LIBRARY(pbr_hip) definitions with HIP
("CUDA" dispatch key) kernels that fill the C-ABI descriptor and call pbr_cook_torrance & co. on torch's current
stream.  This module loads that library and completes the operators from Python:

  * fake (meta) kernels -- shapes / dtypes without running anything, for FakeTensor tracing and torch.compile;
  * the autograd formula of `pbr_hip::cook_torrance`: one call of `pbr_hip::cook_torrance_backward` (the backward
    kernel: gradients of the maps and of view / lights / intensities, what the reference's autograd derives from
    cooktorrance.py:92-182), then the folds autograd would do for a tile repeat or a batch-shared map.

`pypbr_amd.functional.cook_torrance` goes through `torch.ops.pbr_hip.cook_torrance` when this extension is present and
falls back to the ctypes binding of the same C ABI when it is not (`available()`); both end in the same kernels.
"""
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_pbr_torch_ops.so")
_loaded = None


def available() -> bool:
    """Loads pypbr_amd/_pbr_torch_ops.so on first use; False when it has not been built (PBR_NO_TORCH_OPS=1 forces that) or
    does not load against this torch (another torch build, another C++ ABI): one warning, then every call takes the ctypes
    binding of the same C ABI -- not an exception on the first call and a silent fallback afterwards."""
    global _loaded
    if _loaded is None:
        _loaded = False
        if os.environ.get("PBR_NO_TORCH_OPS") != "1" and os.path.exists(LIB_PATH):
            from . import _native
            _native.lib()                       # libpbr_hip.so first: the extension links against it (a missing one raises)
            try:
                torch.ops.load_library(LIB_PATH)
                _register()
                _loaded = True
            except (OSError, RuntimeError) as e:
                import warnings
                warnings.warn("pypbr_amd: %s does not load against torch %s (%s); using the ctypes binding of libpbr_hip.so. "
                              "Rebuild with `make -C pypbr_amd/csrc torch_ops`." % (LIB_PATH, torch.__version__, str(e).splitlines()[0]),
                              RuntimeWarning, stacklevel=2)
    return _loaded


def _out_extent(albedo, tile_y, tile_x, y_offset, rows):
    B, _, H, W = albedo.shape
    if tile_y == 1 and tile_x == 1:
        return B, H, W
    return B, (rows if rows > 0 else tile_y * H - y_offset), tile_x * W


def _register():
    lib = torch.library

    @lib.register_fake("pbr_hip::cook_torrance")
    def _(albedo, normal, roughness, metallic, specular, view_dir, lights, intensities, light_size, light_type, albedo_is_srgb,
          specular_is_srgb, convert_to_diffuse_specular, return_srgb, y_offset=0, height_total=0, tile_y=1, tile_x=1, rows=0,
          half_result=False):
        B, H, W = _out_extent(albedo, tile_y, tile_x, y_offset, rows)
        return albedo.new_empty((B, 3, H, W), dtype=torch.float16 if half_result else torch.float32)

    @lib.register_fake("pbr_hip::cook_torrance_backward")
    def _(grad_out, albedo, normal, roughness, metallic, specular, view_dir, lights, intensities, light_size, light_type,
          albedo_is_srgb, specular_is_srgb, convert_to_diffuse_specular, return_srgb, y_offset, height_total, tile_y, tile_x, rows,
          want_albedo, want_normal, want_roughness, want_metallic, want_specular, want_params):
        B, H, W = _out_extent(albedo, tile_y, tile_x, y_offset, rows)
        maps = [t for t in (albedo, normal, roughness, metallic, specular) if t is not None]
        shared = B > 1 and any(t.shape[0] == 1 for t in maps)
        if (tile_y, tile_x) != (1, 1) and not want_params and not shared:      # tiled maps: MAP-sized, folded over the repeats (torch_ops.cpp)
            H, W = albedo.shape[-2:]

        def buf(want, c):
            return albedo.new_empty((B, c, H, W)) if want else albedo.new_empty((0,))
        n_lights = lights.numel() // 3
        return (buf(want_albedo, 3), buf(want_normal and normal is not None, 3), buf(want_roughness, 1),
                buf(want_metallic and metallic is not None, 1), buf(want_specular and metallic is None, 3),
                albedo.new_empty((3 + 6 * n_lights,) if want_params else (0,), dtype=torch.float32))

    @lib.register_fake("pbr_hip::fold_gradient")
    def _(src, h, w, fold_batch):
        return src.new_empty((1 if fold_batch else src.shape[0], src.shape[1], h, w))

    for name in ("srgb_to_linear", "linear_to_srgb"):
        lib.register_fake("pbr_hip::" + name)(lambda texture: torch.empty_like(texture))
    lib.register_fake("pbr_hip::metallic_to_diffuse_specular")(
        lambda albedo, metallic, albedo_is_srgb: (torch.empty_like(albedo), torch.empty_like(albedo)))
    lib.register_fake("pbr_hip::diffuse_specular_to_basecolor_metallic")(
        lambda diffuse, specular, albedo_is_srgb: (torch.empty_like(diffuse), torch.empty_like(diffuse)))

    lib.register_fake("pbr_hip::colour_backward")(lambda texture, grad_out, to_linear: torch.empty_like(texture))
    lib.register_fake("pbr_hip::metallic_to_diffuse_specular_backward")(
        lambda albedo, metallic, g_diffuse, g_specular, albedo_is_srgb: (torch.empty_like(albedo), torch.empty_like(metallic)))
    lib.register_fake("pbr_hip::diffuse_specular_to_basecolor_metallic_backward")(
        lambda diffuse, specular, g_basecolor, g_metallic, albedo_is_srgb: (torch.empty_like(diffuse), torch.empty_like(specular)))
    lib.register_fake("pbr_hip::resize")(lambda texture, h_out, w_out, antialias: texture.new_empty(tuple(texture.shape[:-2]) + (h_out, w_out)))
    lib.register_fake("pbr_hip::resize_backward")(
        lambda grad_out, h_in, w_in, antialias: grad_out.new_empty(tuple(grad_out.shape[:-2]) + (h_in, w_in), dtype=torch.float32))

    # autograd formulas of the map ops: one backward operator each (what the reference's autograd derives from its plain torch
    # ops: functions.py:31-66, metallic.py:98-108, diffuse.py:128-147, base.py:490-504)
    def colour_autograd(name, to_linear):
        def setup(ctx, inputs, output):
            ctx.save_for_backward(inputs[0])

        def backward(ctx, grad_out):
            (x,) = ctx.saved_tensors
            return torch.ops.pbr_hip.colour_backward(x, grad_out, to_linear)
        lib.register_autograd("pbr_hip::" + name, backward, setup_context=setup)
    colour_autograd("srgb_to_linear", True)
    colour_autograd("linear_to_srgb", False)

    def pair_autograd(name):
        def setup(ctx, inputs, output):
            ctx.save_for_backward(inputs[0], inputs[1])
            ctx.flag = inputs[2]

        def backward(ctx, g0, g1):
            x, y = ctx.saved_tensors
            gx, gy = getattr(torch.ops.pbr_hip, name + "_backward")(x, y, g0, g1, ctx.flag)
            return (gx if ctx.needs_input_grad[0] else None, gy.to(y.dtype) if ctx.needs_input_grad[1] else None, None)
        lib.register_autograd("pbr_hip::" + name, backward, setup_context=setup)
    pair_autograd("metallic_to_diffuse_specular")
    pair_autograd("diffuse_specular_to_basecolor_metallic")

    def resize_setup(ctx, inputs, output):
        ctx.geom = (inputs[0].shape[-2], inputs[0].shape[-1], inputs[3])

    def resize_backward(ctx, grad_out):
        h, w, antialias = ctx.geom
        return torch.ops.pbr_hip.resize_backward(grad_out, h, w, antialias), None, None, None
    lib.register_autograd("pbr_hip::resize", resize_backward, setup_context=resize_setup)

    names = ("albedo", "normal", "roughness", "metallic", "specular", "view_dir", "lights", "intensities")

    def setup_context(ctx, inputs, output):
        tensors, rest = inputs[:8], tuple(inputs[8:])
        assert len(rest) == 12, "pbr_hip::cook_torrance takes 8 tensors and 12 scalars"
        ctx.rest = rest
        ctx.present = [t is not None for t in tensors]
        ctx.save_for_backward(*[t for t in tensors if t is not None])
        if output.dtype != torch.float32:
            ctx.half = True

    def backward(ctx, grad_out):
        if getattr(ctx, "half", False):
            raise NotImplementedError("gradients need a float32 result (fp16 maps are fine: their gradients come back in fp16)")
        saved = list(ctx.saved_tensors)
        tensors = [saved.pop(0) if p else None for p in ctx.present]
        albedo, normal, roughness, metallic, specular, view_dir, lights, intensities = tensors
        (light_size, light_type, a_srgb, s_srgb, convert, r_srgb, y_offset, height_total, tile_y, tile_x, rows, _half) = ctx.rest
        need = ctx.needs_input_grad
        want_params = bool(need[5] or need[6] or need[7])
        ga, gn, gr, gm, gs, gp = torch.ops.pbr_hip.cook_torrance_backward(
            grad_out, albedo, normal, roughness, metallic, specular, view_dir, lights, intensities, light_size, light_type, a_srgb,
            s_srgb, convert, r_srgb, y_offset, height_total, tile_y, tile_x, rows, bool(need[0]), bool(need[1] and normal is not None),
            bool(need[2]), bool(need[3] and metallic is not None), bool(need[4] and specular is not None and metallic is None),
            want_params)
        B, _, H, W = albedo.shape
        tiled = (tile_y, tile_x) != (1, 1)
        grads = []
        for g, t, needed in zip((ga, gn, gr, gm, gs), tensors[:5], need[:5]):
            if not needed or t is None or g.numel() == 0:
                grads.append(None)
                continue
            shared = B > 1 and t.shape[0] == 1
            if tiled and tuple(g.shape[-2:]) == (H, W) and not shared:         # already the map's: the operator folded over the repeats
                grads.append(g)
                continue
            if tiled and grad_out.shape[-2] != tile_y * H:
                raise NotImplementedError("gradients of a tiled evaluation with light / view gradients or batch-shared maps need the whole output, not a row band")
            if tiled or shared:        # a map repeated by the fused tile(), or shared by the batch, owns the SUM over its uses
                g = torch.ops.pbr_hip.fold_gradient(g, H, W, shared)   # (fp16 gradients: summed in fp32, rounded once)
            grads.append(g)
        L = lights.numel() // 3
        pg = [None, None, None]
        if want_params:
            def like(g, t, lights_many):
                if g.dim() == 2 and t.numel() == 3 and lights_many:      # one intensity given for several lights: the sum
                    g = g.sum(dim=0)
                return g.reshape(t.shape).to(device=t.device, dtype=t.dtype)
            if need[5]:
                pg[0] = like(gp[0:3], view_dir, False)
            if need[6]:
                pg[1] = like(gp[3:3 + 3 * L].reshape(L, 3), lights, False)
            if need[7]:
                pg[2] = like(gp[3 + 3 * L:].reshape(L, 3), intensities, L > 1)
        return (*grads, *pg, *([None] * 12))

    lib.register_autograd("pbr_hip::cook_torrance", backward, setup_context=setup_context)
