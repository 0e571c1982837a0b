import math, sys, torch
sys.path.insert(0, '.')
from tools.bench_configs import maps
from pypbr_amd import functional as F
s5 = maps(4, 4096, 4096, dtype=torch.float16, seed=5)
lights = [[math.cos(t), math.sin(t), 1.0] for t in [2 * math.pi * i / 16 for i in range(16)]]
p = F.plan_cook_torrance(*s5, view_dir=[0, 0, 1], light=lights, light_intensity=[[1.0 / 16] * 3] * 16, light_type="point", light_size=1.0)
print(p.kernel_name)
st = torch.cuda.current_stream().cuda_stream
for blk in range(8):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): p.launch(st)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print(f"launches {blk*50}-{blk*50+49}: {us:.1f} us  {4*4096*4096/us/1e3:.1f} Gpix/s", flush=True)
