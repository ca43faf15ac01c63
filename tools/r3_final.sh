#!/bin/bash
# Developer tool (GPU box): the round's evidence run with the final binary.  usage: bash tools/r3_final.sh <tag>
#   1. GPU suite + smoke  2. kernel-trace summaries + per-step lists of every BASELINE workload (-> profiles/<tag>_bench_<wl>_*)
#   3. HBM traffic of the isolated fusion kernels (cfg2, cfg4) -> profiles/pmc_traffic.json, of every kernel of the cfg2 step
#      -> <tag>_pmc_hbm_per_kernel.csv, SQ counters -> <tag>_pmc_sq_per_kernel.csv   4. sweep of bench lines
TAG=${1:-round3_c}
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 1000 python3 -m pytest tests -q -m gpu > gpurun_out/${TAG}_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/${TAG}_tests.log
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
for wl in cfg2 cfg3 cfg4 cfg5 cfg2_b4096; do
  timeout -k 10 300 bash tools/round_profile.sh $TAG $wl 0 > gpurun_out/${TAG}_${wl}_profile.log 2>&1 || { echo "profile $wl failed"; tail -5 gpurun_out/${TAG}_${wl}_profile.log; }
done
cd /tmp && export TMPDIR=/tmp
for wl in cfg2 cfg4; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmcr_${wl}_$C -o run -- python3 $R/bench.py --workload $wl --roofline-only > $R/gpurun_out/pmcr_${wl}_$C.log 2>&1
  done
  (cd $R && python3 tools/pmc_traffic.py gpurun_out/pmcr_${wl}_FETCH_SIZE gpurun_out/pmcr_${wl}_WRITE_SIZE $wl > gpurun_out/${TAG}_pmc_traffic_$wl.log 2>&1; tail -3 gpurun_out/${TAG}_pmc_traffic_$wl.log)
  rm -rf $R/gpurun_out/pmcr_${wl}_FETCH_SIZE $R/gpurun_out/pmcr_${wl}_WRITE_SIZE
done
cp $R/profiles/pmc_traffic.json $R/gpurun_out/${TAG}_pmc_traffic.json
cd $R
timeout -k 10 400 bash tools/run_pmc_hbm.sh $TAG > gpurun_out/${TAG}_pmc_hbm.log 2>&1; tail -3 gpurun_out/${TAG}_pmc_hbm.log
timeout -k 10 300 bash tools/run_pmc_sq.sh > gpurun_out/${TAG}_pmc_sq.log 2>&1; cp gpurun_out/pmc_sq_per_kernel.csv gpurun_out/${TAG}_pmc_sq_per_kernel.csv 2>/dev/null
line() { python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$1', round(d['ms_per_step'], 4), 'ms/step', round(d['value']), d['unit'], d['dtype'], 'windows', [round(w, 4) for w in d['extra']['windows_ms_per_step']])"; }
{
for wl in cfg1 cfg2 cfg3 cfg4 cfg5 cfg2_b4096; do
  timeout -k 10 200 python3 bench.py --workload $wl --steps 200 --warmup 20 --no-extras 2>/dev/null | line $wl
done
timeout -k 10 200 python3 bench.py --steps 200 --warmup 20 --no-extras --force-collectives 2>/dev/null | line cfg2_force_collectives
timeout -k 10 200 python3 bench.py --steps 200 --warmup 20 --no-extras --packed-input 2>/dev/null | line cfg2_packed_input
timeout -k 10 300 python3 bench.py --gpus 2 --backend gloo --steps 20 --warmup 5 --no-extras 2>/dev/null | line cfg2_self_launched_gloo2_on_one_gpu
} > gpurun_out/${TAG}_workload_sweep.txt 2>&1
cat gpurun_out/${TAG}_workload_sweep.txt
timeout -k 10 600 python3 bench.py > gpurun_out/${TAG}_bench.log 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"; tail -c 400 gpurun_out/${TAG}_bench.log
