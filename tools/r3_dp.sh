#!/bin/bash
# Developer tool (GPU box): the N > 1 code path on one rank vs the single-process step, + per-step kernel list of the former
cd $GRAFT_REPO_ROOT
if [ "$1" = "tests" ]; then timeout -k 10 1000 python3 -m pytest tests -q -m gpu > gpurun_out/q_tests.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/q_tests.log; fi
line() { python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$1', round(d['ms_per_step'], 4), 'ms/step', round(d['value']), d['config'].get('graph_mode'), d['config'].get('final_loss'))"; }
timeout -k 10 200 python3 bench.py --steps 200 --warmup 20 --no-extras 2>/dev/null | line cfg2
timeout -k 10 200 python3 bench.py --steps 200 --warmup 20 --no-extras --force-collectives 2>gpurun_out/dp_fc.err | line cfg2_force_collectives
timeout -k 10 300 python3 bench.py --gpus 2 --backend gloo --steps 20 --warmup 5 --no-extras 2>gpurun_out/dp_gloo.err | line cfg2_gloo2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/dp_prof -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 10 --windows 1 --no-extras --force-collectives > $GRAFT_REPO_ROOT/gpurun_out/dp_prof.log 2>&1
cd $GRAFT_REPO_ROOT && python3 tools/per_step.py gpurun_out/dp_prof "first_stats" > gpurun_out/dp_per_step.txt; rm -f gpurun_out/dp_prof/*kernel_trace.csv gpurun_out/dp_prof/*/*kernel_trace.csv
cut -c1-150 gpurun_out/dp_per_step.txt
