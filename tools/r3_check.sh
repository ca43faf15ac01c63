#!/bin/bash
# Developer tool (GPU box): GPU suite, smoke, self-launched two-rank rehearsal (gloo on one GPU), default bench line
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python3 -m pytest tests -q -m gpu > gpurun_out/r3_tests.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r3_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 400 python3 bench.py --gpus 2 --backend gloo --steps 20 --warmup 5 > gpurun_out/r3_bench_gloo2.log 2> gpurun_out/r3_bench_gloo2.err; echo "self-launched gloo2 rc=$?"; tail -c 1500 gpurun_out/r3_bench_gloo2.log
timeout -k 10 600 python3 bench.py > gpurun_out/r3_bench.log 2> gpurun_out/r3_bench.err; echo "bench rc=$?"; tail -c 3000 gpurun_out/r3_bench.log
