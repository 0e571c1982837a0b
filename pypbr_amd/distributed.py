"""Multi-GPU sharding of Cook-Torrance evaluation: one process per GPU (torch.distributed,
backend "nccl" = RCCL over xGMI on ROCm; "gloo" for the CPU tests).

Every pixel of every material is independent, so the data path has NO
collective: rank r evaluates its slice and the results stay sharded.  The only
communication is one small broadcast of the light/view parameter block from the rank
that owns it -- latency-bound (~100 floats), done once per parameter change.

Partitioning:
  * B >= world: contiguous batch slices, sizes differ by at most one material;
  * B <  world (e.g. one 4K material on 8 GPUs): split the rows of each material into
    bands; a band is evaluated with (y_offset, height_total) so the point-light grid is
    that of the full map.
The reference has no distributed code at all; this module is a build extension.
"""
from typing import Dict, NamedTuple, Optional, Tuple

import torch

# packed parameter block: [n_lights, light_size, view(3), lights(L,3), intensities(L,3)], padded to MAX_LIGHTS
_MAX_LIGHTS = 16
_BLOCK_FLOATS = 2 + 3 + _MAX_LIGHTS * 6


class Shard(NamedTuple):
    batch_start: int     # first material
    batch_stop: int      # one past the last material
    row_start: int       # first row (0 unless B < world)
    row_stop: int        # one past the last row


def partition(batch: int, height: int, world: int, rank: int) -> Shard:
    """This rank's share of a [batch, C, height, W] evaluation.  Shards are disjoint and
    cover everything; a rank may get an empty shard only when batch*height < world."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("bad world/rank %d/%d" % (rank, world))
    if batch >= world:
        base, extra = divmod(batch, world)
        start = rank * base + min(rank, extra)
        return Shard(start, start + base + (1 if rank < extra else 0), 0, height)
    # fewer materials than ranks: ranks_per_material ranks share one material's rows
    per_mat = world // batch
    spare = world - per_mat * batch            # the first `spare` materials get one more rank
    b, r0 = 0, 0
    for b in range(batch):
        k = per_mat + (1 if b < spare else 0)
        if rank < r0 + k:
            idx = rank - r0
            base, extra = divmod(height, k)
            y0 = idx * base + min(idx, extra)
            return Shard(b, b + 1, y0, y0 + base + (1 if idx < extra else 0))
        r0 += k
    raise AssertionError("unreachable")


def pack_light_block(params: Dict) -> torch.Tensor:
    lights = torch.as_tensor(params["light"], dtype=torch.float32).reshape(-1, 3)
    inten = torch.as_tensor(params["light_intensity"], dtype=torch.float32).reshape(-1, 3)
    if inten.shape[0] == 1 and lights.shape[0] > 1:
        inten = inten.expand(lights.shape[0], 3)
    L = lights.shape[0]
    if not 1 <= L <= _MAX_LIGHTS or inten.shape[0] != L:
        raise ValueError("1..%d lights with matching intensities expected" % _MAX_LIGHTS)
    blk = torch.zeros(_BLOCK_FLOATS, dtype=torch.float32)
    blk[0] = float(L)
    blk[1] = float(params.get("light_size") or 0.0)
    blk[2:5] = torch.as_tensor(params["view_dir"], dtype=torch.float32).reshape(3)
    blk[5:5 + 3 * L] = lights.reshape(-1)
    blk[5 + 3 * _MAX_LIGHTS:5 + 3 * _MAX_LIGHTS + 3 * L] = inten.reshape(-1)
    return blk


def unpack_light_block(blk: torch.Tensor) -> Dict:
    blk = blk.detach().to("cpu", torch.float32)
    L = int(blk[0].item())
    size = float(blk[1].item())
    o = 5 + 3 * _MAX_LIGHTS
    return {"view_dir": blk[2:5].tolist(), "light": blk[5:5 + 3 * L].reshape(L, 3).tolist(),
            "light_intensity": blk[o:o + 3 * L].reshape(L, 3).tolist(), "light_size": size if size != 0 else None}     # `light_size or 1.0`: only 0 is falsy; negative / NaN sizes travel as they are


def broadcast_light_block(params: Optional[Dict], device: torch.device, src: int = 0, group=None) -> Dict:
    """Rank `src` passes its parameters, the others pass None; everyone returns the same
    dict.  One broadcast of a 404-byte block (RCCL over xGMI on GPUs)."""
    import torch.distributed as dist
    device = _collective_device(device, group)
    if dist.get_rank(group) == src:
        if params is None:
            raise ValueError("the source rank must provide the parameters")
        blk = pack_light_block(params).to(device)
    else:
        blk = torch.empty(_BLOCK_FLOATS, dtype=torch.float32, device=device)
    dist.broadcast(blk, src=src, group=group)
    return unpack_light_block(blk)


def _collective_device(device, group=None) -> torch.device:
    """Where the small collectives' buffers live: on the GPU for "nccl" (RCCL over xGMI), on the host for "gloo"
    (CPU tests; several ranks sharing one GPU on a single-GPU box)."""
    import torch.distributed as dist
    return torch.device(device) if dist.get_backend(group) == "nccl" else torch.device("cpu")


def shard_maps(maps: Dict[str, Optional[torch.Tensor]], shard: Shard) -> Dict[str, Optional[torch.Tensor]]:
    """Zero-copy views of [B,C,H,W] maps for one shard (row bands stay strided views:
    the C ABI takes batch/channel strides)."""
    out = {}
    for name, t in maps.items():
        out[name] = None if t is None else t[shard.batch_start:shard.batch_stop, :, shard.row_start:shard.row_stop, :]
    return out


def cook_torrance_sharded(maps: Dict[str, Optional[torch.Tensor]], params: Optional[Dict], *, light_type: str,
                          src: int = 0, group=None, render=None, owned: Optional[Shard] = None,
                          global_shape: Optional[Tuple[int, int]] = None, plan: bool = False, **flags):
    """Evaluates this rank's shard of `maps` ([B,C,H,W] tensors: albedo, normal, roughness,
    metallic | specular -- every rank holds, or can index, the full batch) with the
    parameters broadcast from `src`.  Returns (shard, output or None for an empty shard).
    `owned=partition(...)` with `global_shape=(B, H)`: `maps` (and a blend's second material / mask) hold exactly THIS
    rank's shard -- the materials [batch_start, batch_stop) and rows [row_start, row_stop) of a [B,C,H,W] job that no rank
    holds as a whole (every rank generates or loads its own slice: BASELINE config 4, 512 materials over 8 GPUs); nothing
    is cut.  `plan=True` returns (shard, RenderPlan or None) instead of launching: the caller launches it as often as it
    likes (bench.py), parameters and blend flags already exchanged.
    `tile=` and `blend=` (see functional.plan_cook_torrance) shard too: with a fused tile the ranks split the rows of
    the tiled OUTPUT over whole source maps; a blend's second material and mask are sliced like the first.  A fused
    blend over ROW BANDS has the path's one real exchange step: whether the blended normal map counts as already
    signed is a property of the whole map (base.py:212), so every rank works the flag out for its rows
    (pbr_blend_normal_sign) and the ranks that share a material combine them with one all-reduce (MAX) of B ints.
    `render` defaults to the HIP path (tests inject a recorder on CPU)."""
    import torch.distributed as dist
    from .functional import tile_counts
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    albedo = maps["albedo"]
    ny, nx = tile_counts(flags.pop("tile", 1))
    blend = flags.pop("blend", None)
    tiled = (ny, nx) != (1, 1)
    if owned is not None:
        if global_shape is None:
            raise ValueError("owned= needs global_shape=(B, H): the extent of the whole job")
        B, H = int(global_shape[0]), int(global_shape[1])
        shard = Shard(*owned)
        if shard != partition(B, H * ny, world, rank):
            raise ValueError("owned=%s is not partition(%d, %d, %d, %d)" % (tuple(shard), B, H * ny, world, rank))
        want_rows = H if tiled else shard.row_stop - shard.row_start
        if albedo.shape[0] != shard.batch_stop - shard.batch_start or albedo.shape[-2] != want_rows:
            raise ValueError("maps %s do not hold the shard %s" % (tuple(albedo.shape), tuple(shard)))
    else:
        B, _, H, _ = albedo.shape
        shard = partition(B, H * ny, world, rank)          # rows of the OUTPUT: with a fused tile() that is ny * H
    p = broadcast_light_block(params, device=albedo.device, src=src, group=group)
    empty = shard.batch_stop <= shard.batch_start or shard.row_stop <= shard.row_start
    # untiled row bands of a fused blend: the one exchange step (every rank takes part, also one with an empty shard)
    exchange = blend is not None and not tiled and B < world and render is None
    if empty and not exchange:
        return shard, None
    if owned is not None:
        cut = lambda t: t
    elif tiled:    # the kernel wraps its texel addresses: every rank keeps whole source maps and evaluates a band of the output
        cut = lambda t: None if t is None else t[shard.batch_start:shard.batch_stop]
    else:
        cut = lambda t: None if t is None else t[shard.batch_start:shard.batch_stop, :, shard.row_start:shard.row_stop, :]
    if tiled:
        flags.update(tile=(ny, nx), rows=shard.row_stop - shard.row_start)
        total = None
    else:
        total = H
    nb = shard.batch_stop - shard.batch_start
    if blend is not None:   # second material and mask are sharded exactly like the first ([B|1,C,H,W] tensors)
        def cut2(t):
            if t is None:
                return None
            t = t if t.dim() == 4 else t.unsqueeze(0)
            if owned is not None:
                return t
            return cut(t if t.shape[0] > 1 else t.expand(B, *t.shape[1:]))
        flags["blend"] = tuple(cut2(t) for t in blend)
    kw = dict(view_dir=p["view_dir"], light=p["light"], light_intensity=p["light_intensity"], light_type=light_type,
              light_size=p["light_size"], y_offset=shard.row_start, height_total=total, **flags)
    m = {name: cut(t) for name, t in maps.items()}
    if render is not None:
        return shard, render(m["albedo"], m.get("normal"), m["roughness"], m.get("metallic"), m.get("specular"), **kw)
    from .functional import cook_torrance, plan_cook_torrance
    if not exchange:
        if plan:
            return shard, plan_cook_torrance(m["albedo"], m.get("normal"), m["roughness"], m.get("metallic"), m.get("specular"), **kw)
        return shard, cook_torrance(m["albedo"], m.get("normal"), m["roughness"], m.get("metallic"), m.get("specular"), **kw)
    signed = torch.zeros(B, dtype=torch.int32, device=albedo.device)       # one flag per material of the FULL batch
    rp = None
    if not empty:
        with torch.no_grad():
            rp = plan_cook_torrance(*[None if t is None else t.detach() for t in (m["albedo"], m.get("normal"), m["roughness"], m.get("metallic"), m.get("specular"))],
                                    blend_flags=signed[shard.batch_start:shard.batch_stop],
                                    **dict(kw, blend=tuple(None if t is None else t.detach() for t in kw["blend"])))
        signed[shard.batch_start:shard.batch_stop] = rp.blend_normal_sign()
    where = _collective_device(albedo.device, group)
    combined = signed.to(where)
    dist.all_reduce(combined, op=dist.ReduceOp.MAX, group=group)
    if empty:
        return shard, None
    mine = combined[shard.batch_start:shard.batch_stop].to(albedo.device)
    wants_grad = torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad
                                                 for t in list(m.values()) + list(kw["blend"]) + [kw["view_dir"], kw["light"], kw["light_intensity"]])
    if wants_grad and not plan:
        # with the whole map's flags in hand the band is an ordinary differentiable evaluation: the fused blend's own backward
        # kernel (functional._FusedBlendFn) gives the gradients of both materials' maps and of the mask for this rank's rows
        return shard, cook_torrance(m["albedo"], m.get("normal"), m["roughness"], m.get("metallic"), m.get("specular"), blend_flags=mine, **kw)
    rp.use_blend_flags(mine)
    if plan:
        return shard, rp
    with torch.cuda.device(rp.device):
        return shard, rp.launch()
