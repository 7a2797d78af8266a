#!/bin/bash
# development loop on the GPU box (gpurun -- bash tools/quick_check.sh): solver parity subset, the headline line, config 2
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -x -q --timeout 300 -k "cgnr or fista or admm" 2>&1 | tail -4
timeout 300 python bench.py --no-extras --no-cpu-baseline 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('value', d['value'], 'us/iter', d['roofline']['iteration']['us_hip_events']); [print(k[:30], v['us_per_launch']) for k, v in d['roofline']['per_kernel'].items()]
"
timeout 120 python tools/bench_configs.py fista
