"""Runs the snippets of README.md "Use" on a GPU box: python tools/readme_check.py (from the repository root)."""
import sys, torch
sys.path.insert(0, ".")
from pypbr_amd.io import load_material_from_folder
from pypbr_amd.models import CookTorranceBRDF
import pypbr_amd.blending as B
from pypbr_amd import functional as F
view_dir, light_position, light_intensity, light_size = torch.tensor([0.0, 0.0, 1.0]), torch.tensor([0.1, 0.1, 1.0]), torch.tensor([1.0, 1.0, 1.0]), 1.0
material = load_material_from_folder("tests/golden/tiles", preferred_workflow="metallic").resize((512, 512)).to("cuda")
brdf = CookTorranceBRDF(light_type="point")
color = brdf(material, view_dir, light_position, light_intensity, light_size)
print(color.shape, color.device, float(color.mean()))
material_a = load_material_from_folder("tests/golden/tiles", preferred_workflow="metallic").resize((256, 256)).to("cuda")
material_b = material_a.clone(); material_b.albedo = material_b.albedo.flip(-1)
material_a.tile(2, lazy=True)
print(brdf(material_a, view_dir, light_position, light_intensity, light_size).shape)
material_a.materialize_tile()
with B.lazy_blending():
    blended, mask = B.HeightBlend(blend_width=0.1, shift=-0.5)(material_a, material_b.tile(2))
print(blended.__dict__.get("_lazy_blend") is not None, brdf(blended, view_dir, light_position, light_intensity, light_size).shape)
a, n, r, m = (material._maps[k] for k in ("albedo", "normal", "roughness", "metallic"))
plan = F.plan_cook_torrance(a, n, r, m, view_dir=[0, 0, 1], light=[[0.1, 0.1, 1.0], [0.3, 0.2, 0.8]], light_intensity=[[0.5] * 3] * 2,
                            light_type="point", light_size=1.0, autotune=True)
print(plan.launch().shape, plan.desc.schedule)
a2 = a.clone().requires_grad_()
light = torch.tensor([0.1, 0.1, 1.0], device="cuda", requires_grad=True)
F.cook_torrance(a2, n, r, m, view_dir=[0, 0, 1], light=light, light_intensity=[1, 1, 1]).mean().backward()
print(a2.grad.abs().sum().item() > 0, light.grad, [t.shape for t in F.pack_maps(a, n, r, m)])
out = torch.ops.pbr_hip.cook_torrance(a[None], n[None], r[None], m[None], None, torch.tensor([0.0, 0.0, 1.0]), torch.tensor([[0.1, 0.1, 1.0]]),
                                      torch.ones(1, 3), 1.0, 1, True, True, False, True)
print(out.shape, torch.equal(out[0], color))
from pypbr_amd.losses import RenderingLoss
pred = material.clone()
pred.albedo = pred._maps["albedo"].clone().requires_grad_()
loss = RenderingLoss(light_type="point")(pred, material_a)
loss.backward()
print(type(loss.grad_fn).__name__, loss.item(), pred._maps["albedo"].grad is not None)
a3 = a.clone().requires_grad_()
loss = F.rendering_loss_mse(a3, n, r, m, target=color, view_dir=[0, 0, 1], light=[0.2, 0.1, 1.0], light_intensity=[1, 1, 1])
loss.backward()
print(loss.item(), a3.grad.abs().sum().item() > 0)
