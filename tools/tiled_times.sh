#!/bin/bash
# In-process times of the tiled kernels (tools/repeat_bwd_probe.py: not clock-settled; tools/run_kernels.py tiled: settled) -- the A/B loop of DESIGN.md 3.2.
python tools/repeat_bwd_probe.py 2>/dev/null
python tools/run_kernels.py 20 tiled 200 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('%-44s %8.1f us' % (r['case'][:44], r['us_per_launch_hip_events']))
"
