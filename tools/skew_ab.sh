cd $GRAFT_REPO_ROOT
for i in 1 2; do for sk in 0 4352; do PBR_PLANE_SKEW_BYTES=$sk python3 bench.py --no-cpu-baseline --steps 300 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('skew $sk', d['value'], d['value_cold'], d['roofline']['kernel_us_steady'], d['roofline']['kernel_us_cold'])"; done; done
python3 - <<'PY'
import torch, sys, os
sys.path.insert(0,'.')
from bench import synth_material
from pypbr_amd import functional as F
dev=torch.device('cuda',0)
def timed(plan, iters=20):
    st=torch.cuda.current_stream().cuda_stream
    for _ in range(5): plan.launch(st)
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): plan.launch(st)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/iters*1e3
for S,B in ((2048,16),(2048,64),(4096,4)):
    one=synth_material(S,dev,3); maps=[torch.stack([t]*B) for t in one]
    kw=dict(view_dir=[0,0,1], light=[0.3,-0.2,1.0], light_intensity=[1,1,1], light_type='directional')
    base=min(timed(F.plan_cook_torrance(*maps, **kw)) for _ in range(3))
    for sk in (0,4352):
        F.PLANE_SKEW_BYTES=sk
        *pm,out=F.pack_maps(*maps, reserve_output=True)
        t=min(timed(F.plan_cook_torrance(*pm, out=out, **kw)) for _ in range(3))
        print(f'{B}x{S}^2 directional: plain tensors {base:.1f} us; pack_maps skew {sk}: {t:.1f} us  {44*B*S*S/t/1e3:.0f} GB/s')
        del pm,out
    del maps
PY
python3 -m pytest tests -q -m gpu -x 2>&1 | tail -3
