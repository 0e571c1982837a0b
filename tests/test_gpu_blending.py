"""GPU parity of the blending pre-stage against pypbr.blending run by the real
reference on 96x96 crops of its two PNG materials (tests/golden/blend.npz), through the class API
the reference's examples/example_blend.py uses."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _materials(z, device):
    from pypbr_amd.materials import BasecolorMetallicMaterial
    mats = []
    for i in (1, 2):
        m = BasecolorMetallicMaterial(device=torch.device(device))
        for key in z:
            if key.startswith(f"in_m{i}_"):
                m._maps[key[len(f"in_m{i}_"):]] = torch.from_numpy(z[key]).to(device)    # maps exactly as the reference loaded them
        mats.append(m)
    return mats


@pytest.mark.parametrize("device", ["cuda", "cpu"])
def test_blends_match_reference(device, golden):
    import pypbr_amd.blending as B
    z = golden("blend")
    m1, m2 = _materials(z, device)
    blends = {"height": B.HeightBlend(blend_width=0.1, shift=-0.5), "mask": B.MaskBlend(torch.from_numpy(z["in_mask"]).to(device)),
              "prop": B.PropertyBlend(property_name="roughness", blend_width=0.1),
              "gradh": B.GradientBlend("horizontal"), "gradv": B.BlendFactory.get_blend_method("gradient", direction="vertical")}
    for name, blender in blends.items():
        out, mask = blender(m1, m2)
        assert type(out) is type(m1) and mask.shape == (1, 96, 96)
        assert np.abs(mask.cpu().numpy() - z[f"out_{name}_mask"]).max() <= 2e-6, name
        expected = sorted(k[len(f"out_{name}_"):] for k in z if k.startswith(f"out_{name}_") and not k.endswith("_mask"))
        assert sorted(out._maps) == expected
        for k, v in out._maps.items():
            assert v.device.type == device
            assert np.abs(v.cpu().numpy() - z[f"out_{name}_{k}"]).max() <= 3e-6, (name, k)
    assert out.albedo_is_srgb == m1.albedo_is_srgb


def test_blend_then_render_pipeline(golden):
    """example_blend.py's sequence on the crops: HeightBlend -> resize -> tile -> point-light render, against
    the same sequence evaluated by the oracle on the reference's blended maps."""
    import torch_oracle as O
    import pypbr_amd.blending as B
    from pypbr_amd.models import CookTorranceBRDF
    z = golden("blend")
    m1, m2 = _materials(z, "cuda")
    material, mask = B.HeightBlend(blend_width=0.1, shift=-0.5)(m1, m2)
    material.tile(2)
    out = CookTorranceBRDF("point")(material, torch.tensor([0.0, 0.0, 1.0]), torch.tensor([0.1, 0.1, 1.0]),
                                    torch.tensor([1.0, 1.0, 1.0]), 1.0)
    ref_maps = {k: torch.from_numpy(z[f"out_height_{k}"]).repeat(1, 2, 2) for k in ("albedo", "normal", "roughness", "metallic")}
    ref = O.cook_torrance(ref_maps["albedo"], ref_maps["normal"], ref_maps["roughness"], ref_maps["metallic"], None,
                          view=torch.tensor([0.0, 0.0, 1.0]), light=torch.tensor([0.1, 0.1, 1.0]),
                          intensity=torch.tensor([1.0, 1.0, 1.0]), light_type="point", light_size=1.0)
    assert (out.cpu() - ref).abs().max().item() <= 1e-5


def test_blend_errors():
    import pypbr_amd.blending as B
    from pypbr_amd.materials import BasecolorMetallicMaterial
    a = BasecolorMetallicMaterial(albedo=torch.rand(3, 8, 8), roughness=torch.rand(1, 8, 8))
    with pytest.raises(ValueError, match="height maps"):
        B.blend_on_height(a, a)
    with pytest.raises(ValueError, match="'metallic' maps"):
        B.blend_on_properties(a, a)
    with pytest.raises(ValueError, match="Mask must have shape"):
        B.blend_with_mask(a, a, torch.rand(2, 8, 8))
    with pytest.raises(ValueError, match="Direction must be"):
        B.blend_with_gradient(a, a, "diagonal")
    with pytest.raises(ValueError, match="Unknown blending method"):
        B.blend_materials(a, a, method="magic")
    with pytest.raises(ValueError, match="Mask must be provided"):
        B.blend_materials(a, a, method="mask")
