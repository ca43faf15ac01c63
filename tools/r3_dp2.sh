#!/bin/bash
cd $GRAFT_REPO_ROOT
line() { python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$1', round(d['ms_per_step'], 4), 'ms/step', d['config'].get('final_loss'))"; }
for st in 5 50; do
timeout -k 10 200 python3 bench.py --steps $st --warmup 0 --windows 1 --no-extras 2>/dev/null | line single_$st
timeout -k 10 200 python3 bench.py --steps $st --warmup 0 --windows 1 --no-extras --force-collectives 2>/dev/null | line fc_$st
timeout -k 10 200 python3 bench.py --steps $st --warmup 0 --windows 1 --no-extras --force-collectives --no-fused-loss 2>/dev/null | line fc_nofused_$st
timeout -k 10 200 python3 bench.py --steps $st --warmup 0 --windows 1 --no-extras --no-fused-loss 2>/dev/null | line single_nofused_$st
done
