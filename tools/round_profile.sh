#!/bin/bash
# Developer tool: the per-round evidence run (on the GPU box): default bench line, rocprofv3 kernel-trace summary of the same
# workload, per-step table.  usage: bash tools/round_profile.sh <tag>   -> gpurun_out/<tag>_*
set -e
TAG=${1:-round}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof -o run -- python3 $R/bench.py --steps 60 --warmup 10 --no-extras > $R/gpurun_out/${TAG}_prof.log 2>&1
python3 $R/tools/per_step.py $R/gpurun_out/${TAG}_prof "first_stats" > $R/gpurun_out/${TAG}_per_step.txt
rm -f $R/gpurun_out/${TAG}_prof/*kernel_trace.csv
cp $R/gpurun_out/${TAG}_prof/run_kernel_stats.csv $R/profiles/${TAG}_bench_cfg2_kernel_stats.csv
cd $R && python3 bench.py > gpurun_out/${TAG}_bench.log 2> gpurun_out/${TAG}_bench.err
tail -c 600 gpurun_out/${TAG}_bench.log
