"""Pins oracle/blend_oracle.py (the ATen-level restatement of pypbr.blending) against tests/golden/blend.npz, which the
REAL reference produced: masks, every blended map of every blend kind, and the renders of the blended materials
(blend -> re-decoded normal -> CookTorranceBRDF).  Runs on the CPU; bit for bit wherever the reference's op order is
reproduced exactly."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import blend_oracle as BO  # noqa: E402
import torch_oracle as O  # noqa: E402

KINDS = ("height", "mask", "prop", "gradh", "gradv")
LIGHTS = {"pt1": ("point", [0.1, 0.1, 1.0], 1.0), "dir": ("directional", [0.3, -0.2, 1.0], None)}


def _materials(z):
    return [{k[len(f"in_m{i}_"):]: torch.from_numpy(z[k]) for k in z if k.startswith(f"in_m{i}_")} for i in (1, 2)]


def _mask(kind, m1, m2, z):
    if kind == "height":
        return BO.sigmoid_mask(m1["height"], m2["height"], 0.1, -0.5)
    if kind == "mask":
        return torch.from_numpy(z["in_mask"]).unsqueeze(0)
    if kind == "prop":
        return BO.sigmoid_mask(m1["roughness"], m2["roughness"], 0.1)
    return BO.gradient_mask(96, 96, "horizontal" if kind == "gradh" else "vertical")


@pytest.mark.parametrize("kind", KINDS)
def test_blend_oracle_reproduces_the_reference(kind, golden):
    z = golden("blend")
    m1, m2 = _materials(z)
    mask = _mask(kind, m1, m2, z)
    assert mask.shape == (1, 96, 96) and np.array_equal(mask.numpy(), z[f"out_{kind}_mask"])
    blended = BO.blend_materials(m1, m2, mask)
    expected = sorted(k[len(f"out_{kind}_"):] for k in z if k.startswith(f"out_{kind}_") and not k.endswith("_mask"))
    assert sorted(blended) == expected
    for name, t in blended.items():
        assert np.array_equal(t.numpy(), z[f"out_{kind}_{name}"]), (kind, name, float(np.abs(t.numpy() - z[f"out_{kind}_{name}"]).max()))
    for lk, (ltype, lvec, lsize) in LIGHTS.items():
        out = O.cook_torrance(blended["albedo"], blended["normal"], blended["roughness"], blended["metallic"], None,
                              view=torch.tensor([0.0, 0.0, 1.0]), light=torch.tensor(lvec), intensity=torch.tensor([1.0, 1.0, 1.0]),
                              light_type=ltype, light_size=lsize)
        assert np.array_equal(out.numpy(), z[f"render_{kind}_{lk}"]), (kind, lk, float(np.abs(out.numpy() - z[f"render_{kind}_{lk}"]).max()))
