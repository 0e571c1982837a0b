#!/usr/bin/env python3
"""A whole training step -- rendering loss through autograd (the one-kernel step), then an SGD update of the four maps -- captured into a
HIP graph (torch.cuda.graph) and replayed, against the same step run eagerly: small maps are launch-bound, and the library's calls only
enqueue (no host synchronisation, no allocation outside torch's allocator), so the capture is legal.  python tools/graph_step_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_material  # noqa: E402
from pypbr_amd import functional as F  # noqa: E402

dev = torch.device("cuda", 0)
KW = dict(view_dir=[0, 0, 1], light=[0.1, 0.1, 1.0], light_intensity=[1, 1, 1], light_type="point", light_size=1.0)


def make(S, dtype=torch.float32):
    target = F.cook_torrance(*synth_material(S, dev, 4), **KW).float()
    leaves = [t.clone().to(dtype).requires_grad_(True) for t in synth_material(S, dev, 3)]
    return leaves, target


def train_step(leaves, target, lr=0.05):
    for t in leaves:
        t.grad = None
    loss = F.rendering_loss_mse(*leaves, target=target, **KW)
    loss.backward()
    with torch.no_grad():
        for t in leaves:
            t.add_(t.grad, alpha=-lr)
    return loss


def capture(leaves, target):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            train_step(leaves, target)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    for t in leaves:
        t.grad = None
    with torch.cuda.graph(graph):
        loss = train_step(leaves, target)
    return graph, loss


def fit_light_step(maps, target, light, lr=0.5):
    """Fit a point light's position to a target image: evaluation, MSE, backward to the light (light-gradient kernels), SGD update."""
    light.grad = None
    loss = torch.nn.functional.mse_loss(F.cook_torrance(*maps, view_dir=[0, 0, 1], light=light, light_intensity=[1, 1, 1], light_type="point", light_size=1.0), target)
    loss.backward()
    with torch.no_grad():
        light.add_(light.grad, alpha=-lr)
    return loss


def light_fitting(S):
    maps = synth_material(S, dev, 3)
    target = F.cook_torrance(*maps, view_dir=[0, 0, 1], light=[0.25, -0.15, 0.9], light_intensity=[1, 1, 1], light_type="point", light_size=1.0)
    out = {}
    for where in ("cpu", "cuda"):
        light = torch.tensor([-0.2, 0.2, 1.2], device=where, requires_grad=True)
        for _ in range(10):
            fit_light_step(maps, target, light)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            fit_light_step(maps, target, light)
        torch.cuda.synchronize()
        out[where] = (time.perf_counter() - t0) / 100 * 1e6
    light = torch.tensor([-0.2, 0.2, 1.2], device="cuda", requires_grad=True)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fit_light_step(maps, target, light)
    torch.cuda.current_stream().wait_stream(side)
    light.grad = None
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        fit_light_step(maps, target, light)
    for _ in range(10):
        graph.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        graph.replay()
    torch.cuda.synchronize()
    out["graph"] = (time.perf_counter() - t0) / 100 * 1e6
    print(f"{S}^2 light-fitting step (evaluate, MSE, backward to the light, SGD): light on the host {out['cpu']:7.1f} us (one blocking read-back per step), "
          f"light on the device {out['cuda']:7.1f} us, captured {out['graph']:7.1f} us", flush=True)


if __name__ == "__main__":
    for S in (256, 1024, 2048):
        light_fitting(S)
    for S in (256, 512, 1024, 2048):
        leaves, target = make(S)
        for _ in range(20):
            train_step(leaves, target)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            train_step(leaves, target)
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / 200 * 1e6
        graph, loss = capture(leaves, target)
        for _ in range(20):
            graph.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            graph.replay()
        torch.cuda.synchronize()
        replay = (time.perf_counter() - t0) / 200 * 1e6
        print(f"{S}^2 fp32 training step (loss + gradients + SGD update): eager {eager:7.1f} us, HIP graph replay {replay:7.1f} us ({eager / replay:.1f}x), loss {float(loss):.6f}", flush=True)
