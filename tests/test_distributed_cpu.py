"""Multi-process path on CPU: world_size 2 and 3 over gloo (the GPU box runs the same code over
RCCL).  Checks the partition (disjoint, covering), the light-block broadcast, and that the
shards every rank evaluates -- here with the ATen oracle injected as the renderer, as the
checker of the plumbing -- reassemble into the unsharded result."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pypbr_amd.distributed import (Shard, broadcast_light_block, cook_torrance_sharded, pack_light_block, partition,
                                   unpack_light_block)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("batch,height,world", [(1, 4096, 8), (512, 1024, 8), (32, 4096, 8), (3, 10, 8), (5, 7, 2),
                                                (8, 16, 8), (1, 5, 8), (2, 3, 7)])
def test_partition_is_disjoint_and_covering(batch, height, world):
    owner = {}
    for r in range(world):
        s = partition(batch, height, world, r)
        assert 0 <= s.batch_start <= s.batch_stop <= batch and 0 <= s.row_start <= s.row_stop <= height
        for b in range(s.batch_start, s.batch_stop):
            for y in range(s.row_start, s.row_stop):
                assert (b, y) not in owner, "overlap"
                owner[(b, y)] = r
    assert len(owner) == batch * height
    sizes = [sum(1 for v in owner.values() if v == r) for r in range(world)]
    if batch >= world:
        assert max(sizes) - min(sizes) <= height                 # at most one material apart
    assert partition(1, 4096, 8, 3) == Shard(0, 1, 1536, 2048)  # B < G -> row bands
    assert partition(512, 1024, 8, 7) == Shard(448, 512, 0, 1024)
    with pytest.raises(ValueError):
        partition(4, 4, 2, 2)


def test_light_block_roundtrip():
    p = {"view_dir": [0.1, -0.2, 1.0], "light": [[0.1, 0.1, 1.0], [1.0, 0.5, 2.0]], "light_intensity": [[1, 1, 1], [0.5, 0.25, 0.125]],
         "light_size": 2.5}
    q = unpack_light_block(pack_light_block(p))
    assert q["light_size"] == 2.5 and len(q["light"]) == 2
    assert torch.allclose(torch.tensor(q["light"]), torch.tensor(p["light"]))
    assert torch.allclose(torch.tensor(q["light_intensity"]), torch.tensor(p["light_intensity"], dtype=torch.float32))
    assert torch.allclose(torch.tensor(q["view_dir"]), torch.tensor(p["view_dir"]))
    assert unpack_light_block(pack_light_block({**p, "light_size": None}))["light_size"] is None
    # `light_size or 1.0` (cooktorrance.py:130): only 0 is falsy -- a negative size (mirrored grid) and NaN travel as they are
    assert unpack_light_block(pack_light_block({**p, "light_size": -2.5}))["light_size"] == -2.5
    assert unpack_light_block(pack_light_block({**p, "light_size": 0.0}))["light_size"] is None
    nan = unpack_light_block(pack_light_block({**p, "light_size": float("nan")}))["light_size"]
    assert nan is not None and nan != nan
    with pytest.raises(ValueError):
        pack_light_block({**p, "light": [[0, 0, 1]] * 17, "light_intensity": [[1, 1, 1]] * 17})


def _worker(rank, world, port, batch, tmpdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
        import torch_oracle as O
        torch.set_num_threads(1)
        g = torch.Generator().manual_seed(77)                      # every rank builds the same full batch
        H, W = 12, 16
        maps = {"albedo": torch.rand(batch, 3, H, W, generator=g), "normal": torch.rand(batch, 3, H, W, generator=g) * 2 - 1,
                "roughness": torch.rand(batch, 1, H, W, generator=g) * 0.8 + 0.2, "metallic": torch.rand(batch, 1, H, W, generator=g)}
        params = {"view_dir": [0.0, 0.1, 1.0], "light": [[0.2, 0.1, 0.9]], "light_intensity": [[1.0, 0.9, 0.8]], "light_size": 1.5}
        # only rank 0 knows the parameters; the others learn them from the broadcast
        got = broadcast_light_block(params if rank == 0 else None, device=torch.device("cpu"), src=0)
        assert abs(got["light"][0][2] - 0.9) < 1e-6 and got["light_size"] == 1.5

        calls = []

        def render(albedo, normal, roughness, metallic, specular, **kw):
            calls.append(kw)
            return O.cook_torrance_batched(albedo, normal, roughness, metallic, specular, view=torch.tensor(kw["view_dir"]),
                                           light=torch.tensor(kw["light"][0]), intensity=torch.tensor(kw["light_intensity"][0]),
                                           light_type=kw["light_type"], light_size=kw["light_size"],
                                           y_offset=kw["y_offset"], H_total=kw["height_total"])

        shard, out = cook_torrance_sharded(maps, params if rank == 0 else None, light_type="point", render=render)
        if out is not None:
            assert calls[0]["height_total"] == H and calls[0]["y_offset"] == shard.row_start
            assert out.shape == (shard.batch_stop - shard.batch_start, 3, shard.row_stop - shard.row_start, W)
        # a rank that OWNS only its shard (bench.py --config 4: 512 materials nobody holds as a whole) gets the same call
        mine = partition(batch, H, world, rank)
        local = {k: v[mine.batch_start:mine.batch_stop, :, mine.row_start:mine.row_stop].clone() for k, v in maps.items()}
        calls.clear()
        oshard, oout = cook_torrance_sharded(local, params if rank == 0 else None, light_type="point", render=render, owned=mine,
                                             global_shape=(batch, H))
        assert oshard == shard and (oout is None) == (out is None)
        if out is not None:
            assert torch.equal(oout, out) and calls[0]["y_offset"] == shard.row_start and calls[0]["height_total"] == H
        with pytest.raises(ValueError):
            cook_torrance_sharded(local, params if rank == 0 else None, light_type="point", render=render, owned=mine)
        with pytest.raises(ValueError):
            cook_torrance_sharded(local, params if rank == 0 else None, light_type="point", render=render,
                                  owned=partition(batch, H, world, (rank + 1) % world), global_shape=(batch, H))
        # fused tile(2) + fused blend shard too: bands of the tiled OUTPUT over whole source maps; material 2 and the
        # mask cut like material 1 (a recorder stands in for the kernel: this checks the plumbing only)
        seen_kw = []

        def recorder(albedo, normal, roughness, metallic, specular, **kw):
            seen_kw.append((albedo.shape, kw))
            return torch.zeros(albedo.shape[0], 3, kw.get("rows") or albedo.shape[-2], albedo.shape[-1] * (kw.get("tile") or (1, 1))[1])
        mask = torch.rand(1, H, W, generator=g)
        second = (maps["albedo"].flip(0), maps["normal"], maps["roughness"], maps["metallic"], None, mask)
        tshard, tout = cook_torrance_sharded(maps, params if rank == 0 else None, light_type="point", render=recorder, tile=2, blend=second)
        if tout is not None:
            shape, kw = seen_kw[0]
            nb = tshard.batch_stop - tshard.batch_start
            assert shape == (nb, 3, H, W) and kw["tile"] == (2, 2) and kw["rows"] == tshard.row_stop - tshard.row_start
            assert kw["y_offset"] == tshard.row_start and kw["height_total"] is None and 0 <= tshard.row_start < tshard.row_stop <= 2 * H
            assert [None if t is None else tuple(t.shape) for t in kw["blend"]] == [(nb, 3, H, W), (nb, 3, H, W), (nb, 1, H, W), (nb, 1, H, W), None, (nb, 1, H, W)]
            assert torch.equal(kw["blend"][0], maps["albedo"].flip(0)[tshard.batch_start:tshard.batch_stop])
        bshard, _ = cook_torrance_sharded(maps, params if rank == 0 else None, light_type="point", render=recorder, blend=second)
        if len(seen_kw) > (1 if tout is not None else 0):
            shape, kw = seen_kw[-1]
            assert kw["blend"][5].shape == (bshard.batch_stop - bshard.batch_start, 1, bshard.row_stop - bshard.row_start, W) == shape[:1] + (1,) + shape[2:]
        if batch < world:
            # a fused blend over row bands with inputs that require grad must not come back without a grad_fn.  Since
            # round 3 the band goes through the differentiable call (tests/test_gpu_distributed.py checks the gradients); on this
            # CPU-only box both forms reach the kernel library's device check -- after the flags' exchange was prepared
            leaf = {k: (v.clone().requires_grad_(True) if k == "albedo" else v) for k, v in maps.items()}
            with pytest.raises(RuntimeError, match="ROCm device|no CPU"):
                cook_torrance_sharded(leaf, params if rank == 0 else None, light_type="point", blend=second)
        torch.save({"shard": tuple(shard), "out": out, "tshard": tuple(tshard)}, os.path.join(tmpdir, f"rank{rank}.pt"))
        dist.barrier()
        if rank == 0:
            full = O.cook_torrance_batched(maps["albedo"], maps["normal"], maps["roughness"], maps["metallic"], None,
                                           view=torch.tensor(params["view_dir"]), light=torch.tensor(params["light"][0]),
                                           intensity=torch.tensor(params["light_intensity"][0]), light_type="point", light_size=1.5)
            seen = torch.zeros(batch, H, dtype=torch.bool)
            for r in range(world):
                rec = torch.load(os.path.join(tmpdir, f"rank{r}.pt"))
                b0, b1, y0, y1 = rec["shard"]
                if rec["out"] is None:
                    continue
                assert not seen[b0:b1, y0:y1].any()
                seen[b0:b1, y0:y1] = True
                # ATen rounds by position inside a SIMD chunk, so bands agree to the ulp, not always bit for bit
                assert (rec["out"] - full[b0:b1, :, y0:y1]).abs().max().item() <= 1e-5
            assert bool(seen.all())
            tiled_rows = torch.zeros(batch, 2 * H, dtype=torch.bool)            # the tiled output is covered exactly once too
            for r in range(world):
                b0, b1, y0, y1 = torch.load(os.path.join(tmpdir, f"rank{r}.pt"))["tshard"]
                assert not tiled_rows[b0:b1, y0:y1].any()
                tiled_rows[b0:b1, y0:y1] = True
            assert bool(tiled_rows.all())
            if batch < world:
                # row bands of ONE tiled material: tile(2) over `world` ranks cuts bands of 2 H / world rows -- from world 3 on THINNER than a period
                # of the map's rows (H), the case the repeat-inner walk serves through its window of source rows since round 6 (VERDICT r5 next #5;
                # values: tests/test_gpu_round6.py::test_thin_bands_... and tests/test_gpu_distributed.py)
                heights = [rec[3] - rec[2] for rec in (torch.load(os.path.join(tmpdir, f"rank{r}.pt"))["tshard"] for r in range(world)) if rec[3] > rec[2]]
                assert heights and (world < 3 or batch > 1 or min(heights) < H), heights
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,batch", [(2, 5), (2, 1), (3, 2), (3, 1)])
def test_sharded_evaluation_over_gloo(world, batch, tmp_path):
    mp.spawn(_worker, args=(world, _free_port(), batch, str(tmp_path)), nprocs=world, join=True)
