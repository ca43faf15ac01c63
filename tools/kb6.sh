#!/bin/bash
# Developer tool (GPU box): per-launch durations of the fp32 tile GEMM (gemm_jobs.h) inside a cfg4 / cfg3 step for every variant forced
# (gemm_jobs.o rebuilt with -DGJ_DIAG_FORCE_MODE=<2|16>; the library as built runs first).  usage: bash tools/kb6.sh [workloads]
R=$GRAFT_REPO_ROOT
C=$(ls -d $R/prediction-*_amd/csrc)
cd /tmp && export TMPDIR=/tmp
for mode in default 2 16; do
  if [ "$mode" != "default" ]; then
    (cd $C && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -DGJ_DIAG_FORCE_MODE=$mode -c gemm_jobs.hip -o _build/gemm_jobs.o && make > /dev/null 2>&1) || exit 1
  fi
  for wl in ${@:-cfg4}; do
    OUT=$R/gpurun_out/kb6_${mode}_$wl
    rocprofv3 --kernel-trace --output-format csv -d $OUT -o run -- python3 $R/bench.py --workload $wl --steps 30 --warmup 5 --windows 1 --no-extras > $OUT.log 2>&1
    echo "== mode $mode $wl"
    python3 $R/tools/ring_launches.py $OUT
    rm -rf $OUT
  done
done
