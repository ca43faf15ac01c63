#!/bin/bash
# Developer tool: bench lines of the other BASELINE configurations, the N > 1 code path on one rank, and the harness loop
# usage (GPU box): bash tools/workload_sweep.sh  -> gpurun_out/sweep.log
cd $GRAFT_REPO_ROOT
line() { python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$1', round(d['ms_per_step'], 4), 'ms/step', round(d['value']), d['unit'], d['dtype'])"; }
for wl in cfg1 cfg3 cfg4 cfg5 cfg2_b4096; do
  timeout -k 10 200 python3 bench.py --workload $wl --steps 200 --warmup 20 --no-extras 2>/dev/null | line $wl
done
timeout -k 10 200 python3 bench.py --steps 200 --warmup 20 --no-extras --force-collectives 2>/dev/null | line cfg2_force_collectives
timeout -k 10 200 python3 bench.py --steps 200 --warmup 20 --no-extras --packed-input 2>/dev/null | line cfg2_packed_input
timeout -k 10 300 python3 tools/harness_speed.py 2>/dev/null | tail -12
