"""The HIP path under pypbr_amd.distributed.cook_torrance_sharded with world > 1, one process per rank
(SURVEY.md 8e): every rank evaluates its shard with the fused kernel and checks it BIT-EQUAL against the rows of
its own unsharded launch.  Backend: "nccl" (= RCCL over xGMI) with one GPU per rank when the box has at least two
GPUs; on a single-GPU box the two ranks share cuda:0 and the 404-byte light block / the blend flags travel over
"gloo" (RCCL refuses two ranks on one device) -- the data path, which has no collective, is the same."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu

WORLD = 2


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _maps(B, H, W, seed, dtype=torch.float32, specular=False):
    g = torch.Generator().manual_seed(seed)
    a = torch.rand(B, 3, H, W, generator=g)
    n = torch.cat([(torch.rand(B, 2, H, W, generator=g) - 0.5), torch.ones(B, 1, H, W)], 1)
    n = n / n.norm(dim=1, keepdim=True)
    r = torch.rand(B, 1, H, W, generator=g) * 0.9 + 0.1
    second = torch.rand(B, 3 if specular else 1, H, W, generator=g)
    out = {"albedo": a, "normal": n, "roughness": r, ("specular" if specular else "metallic"): second}
    return {k: v.to(dtype) for k, v in out.items()}


def _rank(rank, world, port, backend, results):
    import math
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    dev = torch.device("cuda", rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pypbr_amd import functional as F
        from pypbr_amd.distributed import cook_torrance_sharded
        ring = [[math.cos(2 * math.pi * i / 16), math.sin(2 * math.pi * i / 16), 1.0] for i in range(16)]
        cases = [
            # (name, maps, params (rank 0 only knows them), light_type, flags)
            ("row bands of one material (config 2 on 2 GPUs)", _maps(1, 96, 128, 1),
             dict(view_dir=[0.0, 0.1, 1.0], light=[[0.1, 0.1, 1.0]], light_intensity=[[1.0, 0.9, 0.8]], light_size=1.0), "point", {}),
            ("batch slices (config 4 scaled down)", _maps(5, 64, 64, 2),
             dict(view_dir=[0.0, 0.0, 1.0], light=[[0.1, 0.1, 1.0]], light_intensity=[[1.0, 1.0, 1.0]], light_size=1.0), "point", {}),
            ("16 lights, fp16 maps, batch slices (config 5 scaled down)", _maps(4, 32, 64, 3, torch.float16),
             dict(view_dir=[0.0, 0.0, 1.0], light=ring, light_intensity=[[1.0 / 16] * 3] * 16, light_size=1.0), "point", {}),
            ("16 lights, fp16 maps, row bands", _maps(1, 50, 64, 4, torch.float16),
             dict(view_dir=[0.0, 0.0, 1.0], light=ring, light_intensity=[[1.0 / 16] * 3] * 16, light_size=1.0), "point", {}),
            ("converted + directional (config 3 scaled down), odd split", _maps(3, 40, 72, 5),
             dict(view_dir=[0.0, 0.0, 1.0], light=[[0.3, -0.2, 1.0]], light_intensity=[[1.0, 1.0, 1.0]], light_size=None), "directional",
             dict(convert_to_diffuse_specular=True)),
            ("fused tile(2), row bands of the tiled output", _maps(1, 24, 32, 6, specular=True),
             dict(view_dir=[0.0, 0.0, 1.0], light=[[0.2, 0.1, 0.9]], light_intensity=[[1.0, 1.0, 1.0]], light_size=2.0), "point", dict(tile=2)),
        ]
        log = []
        for name, host, params, ltype, flags in cases:
            maps = {k: v.to(dev) for k, v in host.items()}
            shard, out = cook_torrance_sharded(maps, params if rank == 0 else None, light_type=ltype, **flags)
            full = F.cook_torrance(maps["albedo"], maps["normal"], maps["roughness"], maps.get("metallic"), maps.get("specular"),
                                   view_dir=params["view_dir"], light=params["light"], light_intensity=params["light_intensity"],
                                   light_type=ltype, light_size=params["light_size"], **flags)
            assert out is not None, (name, shard)
            want = full[shard.batch_start:shard.batch_stop, :, shard.row_start:shard.row_stop]
            assert out.shape == want.shape and torch.equal(out, want), (name, rank, shard, float((out - want).abs().max()))
            log.append((name, tuple(shard)))
        # fused blend over ROW BANDS: the blended normal's "already signed?" flag is a whole-map property, so the ranks
        # exchange their bands' flags (the path's one collective).  Material 2 / mask chosen so that the ONLY negative
        # component of the blended normal sits in the LAST rows: rank 0's band alone would decode it as [0,1]-encoded.
        H, W = 64, 64
        one = {"albedo": torch.rand(1, 3, H, W), "roughness": torch.rand(1, 1, H, W) * 0.8 + 0.2, "metallic": torch.rand(1, 1, H, W)}
        nrm = torch.zeros(1, 3, H, W); nrm[:, 0] = 0.3; nrm[:, 1] = 0.2; nrm[:, 2] = 0.9
        nrm2 = nrm.clone(); nrm2[:, 0, H - 1, :] = -0.6
        for flip in (False, True):                      # all-positive blended normal, then one negative row at the bottom
            g = torch.Generator().manual_seed(9)
            m1 = {k: v.to(dev) for k, v in {**one, "normal": nrm / nrm.norm(dim=1, keepdim=True)}.items()}
            n2 = (nrm2 if flip else nrm); n2 = (n2 / n2.norm(dim=1, keepdim=True)).to(dev)
            second = (torch.rand(1, 3, H, W, generator=g).to(dev), n2, (torch.rand(1, 1, H, W, generator=g) * 0.8 + 0.2).to(dev),
                      torch.rand(1, 1, H, W, generator=g).to(dev), None, torch.rand(1, 1, H, W, generator=g).to(dev))
            params = dict(view_dir=[0.0, 0.0, 1.0], light=[[0.1, 0.1, 1.0]], light_intensity=[[1.0, 1.0, 1.0]], light_size=1.0)
            shard, out = cook_torrance_sharded(m1, params if rank == 0 else None, light_type="point", blend=second)
            full = F.cook_torrance(m1["albedo"], m1["normal"], m1["roughness"], m1["metallic"], view_dir=params["view_dir"], light=params["light"],
                                   light_intensity=params["light_intensity"], light_type="point", light_size=1.0, blend=second)
            want = full[:, :, shard.row_start:shard.row_stop]
            assert torch.equal(out, want), ("blend row bands", flip, rank, float((out - want).abs().max()))
            log.append(("fused blend, row bands, negative row at the bottom: %s" % flip, tuple(shard)))
            # with inputs that require grad the band keeps its graph -- the fused blend's own backward
            # kernel, the whole map's flags exchanged -- and its gradients are the rows of the unsharded gradients
            leaf1 = {k: v.clone().requires_grad_(True) for k, v in m1.items()}
            leaf2 = tuple(None if t is None else t.clone().requires_grad_(True) for t in second)
            shard_g, out_g = cook_torrance_sharded(leaf1, params if rank == 0 else None, light_type="point", blend=leaf2)
            assert out_g.requires_grad and torch.equal(out_g.detach(), want)
            wt = torch.rand(1, 3, H, W, generator=torch.Generator().manual_seed(10)).to(dev)
            (out_g * wt[:, :, shard.row_start:shard.row_stop]).sum().backward()
            full1 = {k: v.clone().requires_grad_(True) for k, v in m1.items()}
            full2 = tuple(None if t is None else t.clone().requires_grad_(True) for t in second)
            full_g = F.cook_torrance(full1["albedo"], full1["normal"], full1["roughness"], full1["metallic"], view_dir=params["view_dir"],
                                     light=params["light"], light_intensity=params["light_intensity"], light_type="point", light_size=1.0, blend=full2)
            (full_g * wt).sum().backward()
            rows = slice(shard.row_start, shard.row_stop)
            outside = torch.ones(H, dtype=torch.bool); outside[rows] = False
            for k in leaf1:
                assert torch.equal(leaf1[k].grad[:, :, rows], full1[k].grad[:, :, rows]), ("sharded blend gradient", k, rank)
                assert float(leaf1[k].grad[:, :, outside].abs().sum()) == 0.0            # rows this rank does not own: no contribution
            for a, b in zip(leaf2, full2):
                if a is not None:
                    assert torch.equal(a.grad[:, :, rows], b.grad[:, :, rows])
            log.append(("fused blend, row bands, gradients: %s" % flip, tuple(shard)))
            # and a band WITHOUT the exchanged flags is refused, not silently decoded from its own rows
            with pytest.raises(NotImplementedError):
                F.cook_torrance(*[t[:, :, :32] for t in (m1["albedo"], m1["normal"], m1["roughness"], m1["metallic"])], view_dir=params["view_dir"],
                                light=params["light"], light_intensity=params["light_intensity"], light_type="point", light_size=1.0,
                                y_offset=0, height_total=H, blend=tuple(None if t is None else t[:, :, :32] for t in second))
        torch.cuda.synchronize()
        results.put((rank, backend, log))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_sharded_evaluation_hip_path_world_2():
    import torch.multiprocessing as mp
    backend = "nccl" if torch.cuda.device_count() >= WORLD else "gloo"
    ctx = mp.get_context("forkserver")
    results = ctx.SimpleQueue()
    mp.start_processes(_rank, args=(WORLD, _free_port(), backend, results), nprocs=WORLD, join=True, start_method="forkserver")
    got = sorted(results.get() for _ in range(WORLD))
    assert [g[0] for g in got] == list(range(WORLD))
    shards = {name: [dict(g[2])[name] for g in got] for name, _ in got[0][2]}
    print(f"\n[sharded, backend {backend}, {torch.cuda.device_count()} GPU(s)]")
    for name, ss in shards.items():
        print("   ", name, ss)
        assert ss[0] != ss[1]                                            # the two ranks really evaluated different shards
