"""`RenderingLoss`: the rendering loss of the reference's tutorial (docs/source/tutorials/06_advanced.rst:73-107) as a module.

Upstream the class lives in the tutorial, not in the package:

    rendered_pred = self.brdf(predicted_material, self.view_dir, self.light_dir, self.light_intensity)
    rendered_gt = self.brdf(ground_truth_material, self.view_dir, self.light_dir, self.light_intensity)
    loss = nn.MSELoss()(rendered_pred, rendered_gt)

Same constructor, same forward arguments, same value and gradients here.  The ground-truth rendering is one launch of the fused
evaluation; it runs without a graph when nothing it depends on requires grad (the usual case: the predicted material's evaluation,
the MSE and their backward are then ONE kernel, functional.rendering_loss_mse -> pbr_cook_torrance_mse_step, whenever the predicted
maps are plain tensors on a ROCm device).  When `view_dir` / `light_dir` / `light_intensity` -- shared by both renderings upstream --
or a ground-truth map require grad, the ground truth is rendered differentiably, as upstream renders it: a light being fitted receives
the gradient of BOTH branches.
"""
import torch
import torch.nn as nn

from . import functional as F_
from .models import CookTorranceBRDF


class RenderingLoss(nn.Module):
    def __init__(self, light_type='point', view_dir=torch.tensor([0.0, 0.0, 1.0]), light_dir=torch.tensor([0.1, 0.1, 1.0]),
                 light_intensity=torch.tensor([1.0, 1.0, 1.0]), light_size=None):
        super().__init__()
        self.brdf = CookTorranceBRDF(light_type=light_type)
        self.view_dir = view_dir
        self.light_dir = light_dir
        self.light_intensity = light_intensity
        self.light_size = light_size

    @staticmethod
    def _maps(material):
        store = material.__dict__.get("_store", {})
        return [store.get(k) for k in ("albedo", "normal", "roughness", "metallic", "specular")]

    def forward(self, predicted_material, ground_truth_material):
        """`ground_truth_material`: a material, or the reference rendering itself ((3,H,W) / (B,3,H,W) tensor)."""
        if isinstance(ground_truth_material, torch.Tensor):
            rendered_gt = ground_truth_material
        else:
            # 06_advanced.rst:101-102 renders both materials under autograd.  Without a graph only when nothing could receive a gradient
            # through this branch: no ground-truth map and none of the shared light / view tensors requires grad.
            gt_maps = list(ground_truth_material.__dict__.get("_store", {}).values())
            pending = ground_truth_material.__dict__.get("_lazy_blend")
            if pending is not None:
                gt_maps += list(pending[0].values()) + [pending[1]]
            differentiable = torch.is_grad_enabled() and any(
                isinstance(t, torch.Tensor) and t.requires_grad for t in gt_maps + [self.view_dir, self.light_dir, self.light_intensity])
            with torch.enable_grad() if differentiable else torch.no_grad():
                rendered_gt = self.brdf(ground_truth_material, self.view_dir, self.light_dir, self.light_intensity, self.light_size)
        a, n, r, m, s = self._maps(predicted_material)
        fusable = (predicted_material.__dict__.get("_lazy_blend") is None
                   and torch.device(predicted_material.device).type == "cuda" and not predicted_material._has_pending()
                   and a is not None and r is not None and "normal" in predicted_material.__dict__.get("_store", {})
                   and (m is not None or s is not None) and all(t is None or t.is_cuda for t in (a, n, r, m, s))
                   and self.brdf.override_device is None)
        if not fusable:
            rendered_pred = self.brdf(predicted_material, self.view_dir, self.light_dir, self.light_intensity, self.light_size)
            return nn.MSELoss()(rendered_pred, rendered_gt.to(rendered_pred.device))
        if m is not None:
            s = None
        return F_.rendering_loss_mse(a, n, r, m, s, target=rendered_gt.to(a.device), view_dir=self.view_dir, light=self.light_dir,
                                     light_intensity=self.light_intensity, light_type=self.brdf.light_type, light_size=self.light_size,
                                     tile=predicted_material.lazy_tile,      # a recorded tile(n): evaluated and differentiated at every repeat
                                     albedo_is_srgb=bool(predicted_material.albedo_is_srgb),
                                     specular_is_srgb=bool(getattr(predicted_material, "specular_is_srgb", True)))
