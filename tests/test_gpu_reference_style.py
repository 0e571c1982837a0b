"""The checks the reference's own test-suite makes on this path (tests/test_models.py, test_blending.py, test_materials.py,
test_utils.py of giuvecchio/PyPBR), written against the `pypbr` alias of this package and run the way the reference's
tests run: CPU-resident random materials, default arguments.  They are property checks (shapes, value ranges, round
trips, averages) -- the numeric parity with the reference is pinned elsewhere (golden vectors)."""
import os
import warnings

import pytest
import torch

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture
def pypbr_alias():
    from pypbr_amd import compat
    compat.install()
    yield
    compat.uninstall()


def _random_maps(h, w, seed):
    g = torch.Generator().manual_seed(seed)
    return dict(albedo=torch.rand(3, h, w, generator=g), normal=torch.rand(3, h, w, generator=g) * 2 - 1,
                roughness=torch.rand(1, h, w, generator=g)), g


def test_directional_light_on_a_metallic_material(pypbr_alias):
    from pypbr.materials import BasecolorMetallicMaterial
    from pypbr.models import CookTorranceBRDF
    maps, g = _random_maps(64, 64, 1)
    material = BasecolorMetallicMaterial(metallic=torch.rand(1, 64, 64, generator=g), **maps)
    color = CookTorranceBRDF(light_type="directional")(material, torch.tensor([0.0, 0.0, 1.0]), torch.tensor([0.0, 0.0, 1.0]),
                                                        torch.tensor([1.0, 1.0, 1.0]))
    assert color.shape == (3, 64, 64) and color.device.type == "cpu"
    assert bool((color >= 0).all()) and bool((color <= 1).all())


def test_point_light_on_a_specular_material(pypbr_alias):
    from pypbr.materials import DiffuseSpecularMaterial
    from pypbr.models import CookTorranceBRDF
    maps, g = _random_maps(64, 64, 2)
    material = DiffuseSpecularMaterial(specular=torch.rand(3, 64, 64, generator=g), **maps)
    color = CookTorranceBRDF(light_type="point")(material, torch.tensor([0.0, 0.0, 1.0]), torch.tensor([0.0, 10.0, 10.0]),
                                                  torch.tensor([1.0, 1.0, 1.0]), light_size=5.0)
    assert color.shape == (3, 64, 64)
    assert bool((color >= 0).all()) and bool((color <= 1).all())
    with pytest.raises(ValueError, match="Unsupported light_type"):
        CookTorranceBRDF(light_type="spot")


def test_half_and_half_blend_of_two_real_materials(pypbr_alias):
    from pypbr.blending.functional import blend_with_mask
    from pypbr.io import load_material_from_folder
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mat1 = load_material_from_folder(os.path.join(GOLDEN, "rocks"))
        mat2 = load_material_from_folder(os.path.join(GOLDEN, "tiles"))
    h, w = mat1.size
    blended, mask = blend_with_mask(mat1, mat2, torch.full((1, h, w), 0.5))
    expected = (mat1.albedo.mean().item() + mat2.albedo.mean().item()) / 2
    assert abs(blended.albedo.mean().item() - expected) < 0.1 and mask.shape == (1, h, w)
    assert type(blended) is type(mat1)


def test_srgb_round_trip(pypbr_alias):
    from pypbr.utils import linear_to_srgb, srgb_to_linear
    texture = torch.linspace(0, 1, steps=100).view(1, 10, 10).cuda()
    assert torch.allclose(texture, linear_to_srgb(srgb_to_linear(texture)), atol=1e-4)


def test_clone_is_deep_and_to_moves_every_map(pypbr_alias):
    from pypbr.materials import BasecolorMetallicMaterial
    maps, g = _random_maps(16, 16, 3)
    material = BasecolorMetallicMaterial(metallic=torch.rand(1, 16, 16, generator=g), **maps)
    clone = material.clone()
    for key in material._maps:
        assert material._maps[key] is not clone._maps[key] and torch.equal(material._maps[key], clone._maps[key])
    clone.to(torch.device("cuda"))
    assert all(v.device.type == "cuda" for v in clone._maps.values() if v is not None)
    clone.to(torch.device("cpu"))
    assert all(v.device == torch.device("cpu") for v in clone._maps.values() if v is not None)
    assert all(torch.equal(material._maps[k], clone._maps[k]) for k in material._maps)
