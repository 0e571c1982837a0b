#!/usr/bin/env python3
"""Host <-> device copy rates for a 4096^2 material (512 MiB of maps in, 192 MiB of result out): pageable vs pinned."""
import time

import torch

dev = torch.device("cuda", 0)
n_in, n_out = 8 * 4096 * 4096, 3 * 4096 * 4096
for pinned in (False, True):
    src = torch.rand(n_in).pin_memory() if pinned else torch.rand(n_in)
    dst_host = torch.empty(n_out).pin_memory() if pinned else torch.empty(n_out)
    d_in, d_out = torch.empty(n_in, device=dev), torch.rand(n_out, device=dev)
    for _ in range(2):
        d_in.copy_(src, non_blocking=pinned); dst_host.copy_(d_out, non_blocking=pinned); torch.cuda.synchronize()
    t0 = time.perf_counter(); d_in.copy_(src, non_blocking=pinned); torch.cuda.synchronize(); t1 = time.perf_counter()
    dst_host.copy_(d_out, non_blocking=pinned); torch.cuda.synchronize(); t2 = time.perf_counter()
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    t3 = time.perf_counter()
    with torch.cuda.stream(s1):
        d_in.copy_(src, non_blocking=True)
    with torch.cuda.stream(s2):
        dst_host.copy_(d_out, non_blocking=True)
    torch.cuda.synchronize(); t4 = time.perf_counter()
    print(f"{'pinned  ' if pinned else 'pageable'}: H2D 512 MiB {1e3 * (t1 - t0):6.1f} ms ({n_in * 4 / (t1 - t0) / 1e9:5.1f} GB/s)  "
          f"D2H 192 MiB {1e3 * (t2 - t1):6.1f} ms ({n_out * 4 / (t2 - t1) / 1e9:5.1f} GB/s)  both at once {1e3 * (t4 - t3):6.1f} ms")
