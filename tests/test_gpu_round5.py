"""Round 5 on the GPU box: ADVICE r4 fixes on the device side; the repeat-inner backward / loss step; the repeat-inner walk with
several lights (added as they are built)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ARGS = (torch.tensor([0.0, 0.0, 1.0]), torch.tensor([0.1, 0.1, 1.0]), torch.tensor([1.0, 1.0, 1.0]), 1.0)


def _maps(H, W, seed=0, device="cuda", dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    n = torch.cat([(torch.rand(2, H, W, generator=g) - 0.5), torch.ones(1, H, W)], 0)
    n = n / n.norm(dim=0, keepdim=True)
    maps = dict(albedo=torch.rand(3, H, W, generator=g), normal=n, roughness=torch.rand(1, H, W, generator=g) * 0.8 + 0.2,
                metallic=torch.rand(1, H, W, generator=g))
    return {k: v.to(device=device, dtype=dtype) for k, v in maps.items()}


def test_material_moves_between_gpus_by_device_copy():
    """to('cuda:1') of a material on cuda:0 must not take the host staging memcpy."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two ROCm devices")
    from pypbr_amd.materials import BasecolorMetallicMaterial
    from pypbr_amd.models import CookTorranceBRDF
    m = BasecolorMetallicMaterial(**_maps(40, 64, seed=4, device="cuda:0"))
    want = CookTorranceBRDF("point")(m, *ARGS).cpu()
    m.to("cuda:1")
    assert all(t.device == torch.device("cuda", 1) for t in m._raw.values())
    assert torch.equal(CookTorranceBRDF("point")(m, *ARGS).cpu(), want)


def test_map_assigned_after_tile_on_a_device_material_keeps_its_size():
    """ADVICE r4 (medium), device side: load -> to(cuda) -> tile(2) records the repeat; a map assigned afterwards is not repeated."""
    from pypbr_amd.materials import BasecolorMetallicMaterial
    from pypbr_amd.models import CookTorranceBRDF
    host = _maps(16, 24, seed=2, device="cpu")
    m = BasecolorMetallicMaterial(**host)
    m.resize((16, 24))                                 # first whole-material operation: the maps now wait on the device
    m.tile(2)
    assert m.lazy_tile == (2, 2)
    r = torch.full((1, 32, 48), 0.5)
    m.roughness = r
    assert m.lazy_tile == (1, 1) and m.size == (32, 48)
    ref = BasecolorMetallicMaterial(albedo=host["albedo"].repeat(1, 2, 2), normal=None, roughness=r, metallic=host["metallic"].repeat(1, 2, 2))
    ref._maps["normal"] = host["normal"].repeat(1, 2, 2)
    brdf = CookTorranceBRDF("point")
    got, want = brdf(m, *ARGS), brdf(ref, *ARGS)
    assert got.shape == (3, 32, 48) and torch.allclose(got, want, atol=2e-6)
