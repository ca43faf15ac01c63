#!/bin/bash
# Developer tool (GPU box): GPU suite (optional filter) + bench lines without the CPU leg.  usage: bash tools/r3_quick.sh "<pytest -k expr or empty>" "<workloads>"
cd $GRAFT_REPO_ROOT
if [ -n "$1" ]; then K=(-k "$1"); else K=(); fi
timeout -k 10 1000 python3 -m pytest tests -q -m gpu "${K[@]}" > gpurun_out/q_tests.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/q_tests.log
for wl in ${2:-cfg2 cfg3 cfg4 cfg5}; do
timeout -k 10 300 python3 bench.py --workload $wl --no-extras --steps 100 --warmup 10 > gpurun_out/q_bench_$wl.log 2> gpurun_out/q_bench_$wl.err; echo "bench $wl rc=$?"
python3 - <<PY
import json
try:
    r=json.loads(open("gpurun_out/q_bench_$wl.log").read().strip().splitlines()[-1])
    print("$wl", round(r["ms_per_step"],4), "ms/step", round(r["value"]), r["extra"]["windows_ms_per_step"])
except Exception as e:
    print("parse failed", e)
PY
done
