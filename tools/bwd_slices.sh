#!/bin/bash
# Developer tool: batch-slice count of the fusion backward's weight-gradient jobs (EMB_BWD_S) against step time.
# usage (GPU box): bash tools/bwd_slices.sh  -> gpurun_out/bwd_slices.log
cd $GRAFT_REPO_ROOT
for wl in cfg2_b4096 cfg2 cfg5; do
  for S in 0 2 3 4 6 8; do
    EMB_BWD_S=$S timeout -k 10 120 python3 bench.py --workload $wl --steps 200 --warmup 20 --no-extras 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$wl S=$S', d['ms_per_step'], d['value'])"
  done
done
