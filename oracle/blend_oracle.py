"""oracle/blend_oracle.py -- TEST INFRASTRUCTURE, not product code.

ATen-level restatement of the reference's material blending, the pre-stage of the BRDF in
/root/reference/examples/example_blend.py:14-16.  Same ATen ops in the same order as
/root/reference/pypbr/blending/functional.py, so that it can be pinned bit for bit against tests/golden/blend.npz
(generated from the real reference by oracle/gen_golden.py) on the CPU:

  blend_maps        functional.py:103-110  mask * map1 + (1 - mask) * map2
  blend_normals     functional.py:119-145  F.normalize both, blend, F.normalize
  sigmoid_mask      functional.py:181-190 / :226-233   sigmoid((p1 + shift - p2) / (blend_width + 1e-6))
  gradient_mask     functional.py:262-280  linspace(0, 1) along x or y
  blend_materials   functional.py:64-116   every map; the blended normal passes through
                    MaterialBase._process_normal_map again when it is assigned (base.py:191-242)

Only tests/ may import this module (like everything under oracle/).
"""
from typing import Dict, Optional

import torch
import torch.nn.functional as F

import torch_oracle


def blend_maps(map1: torch.Tensor, map2: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    return mask * map1 + (1 - mask) * map2


def blend_normals(normal1: torch.Tensor, normal2: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    n1, n2 = F.normalize(normal1, dim=0), F.normalize(normal2, dim=0)
    return F.normalize(mask * n1 + (1 - mask) * n2, dim=0)


def sigmoid_mask(prop1: torch.Tensor, prop2: torch.Tensor, blend_width: float, shift: float = 0.0) -> torch.Tensor:
    # blend_on_height adds the shift first; blend_on_properties has none (x + 0.0 == x, so one form serves both)
    return torch.sigmoid(((prop1 + shift) - prop2) / (blend_width + 1e-6))


def gradient_mask(height: int, width: int, direction: str) -> torch.Tensor:
    if direction == "horizontal":
        return torch.linspace(0, 1, steps=width).unsqueeze(0).unsqueeze(0).expand(1, height, width)
    return torch.linspace(0, 1, steps=height).unsqueeze(1).unsqueeze(0).expand(1, height, width)


def blend_materials(maps1: Dict[str, Optional[torch.Tensor]], maps2: Dict[str, Optional[torch.Tensor]],
                    mask: torch.Tensor) -> Dict[str, Optional[torch.Tensor]]:
    """blend_with_mask on name -> tensor dicts; `mask` [1,H,W]."""
    out = {}
    for name in list(maps1.keys()) + [k for k in maps2.keys() if k not in maps1]:
        m1, m2 = maps1.get(name), maps2.get(name)
        if m1 is None or m2 is None:
            blended = m2 if m1 is None else m1
        elif name == "normal":
            blended = blend_normals(m1, m2, mask)
        else:
            blended = blend_maps(m1, m2, mask)
        if name == "normal" and blended is not None:
            blended = torch_oracle.decode_normal(blended)        # setattr(material, "normal", ...) decodes again
        out[name] = blended
    return out
