#!/bin/bash
# Developer tool: the per-round evidence run (on the GPU box): rocprofv3 kernel-trace summary of a bench workload + per-step
# table, copied into profiles/ (the tracked copies the judge reads).
# usage: bash tools/round_profile.sh <tag> [workload=cfg2] [bench=1: also run the default bench line afterwards]
set -e
TAG=${1:-round}
WL=${2:-cfg2}
BENCH=${3:-1}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
case $WL in cfg1|cfg3|cfg4) MARK=ncl_to_nlc;; *) MARK=first_stats;; esac
OUT=$R/gpurun_out/${TAG}_${WL}_prof
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o run -- python3 $R/bench.py --workload $WL --steps 60 --warmup 10 --windows 1 --no-extras > $R/gpurun_out/${TAG}_${WL}_prof.log 2>&1
python3 $R/tools/per_step.py $OUT "$MARK" > $R/gpurun_out/${TAG}_${WL}_per_step.txt
rm -f $OUT/*kernel_trace.csv $OUT/*/*kernel_trace.csv
cp $(find $OUT -name 'run_kernel_stats.csv' | head -1) $R/profiles/${TAG}_bench_${WL}_kernel_stats.csv
cp $R/gpurun_out/${TAG}_${WL}_per_step.txt $R/profiles/${TAG}_bench_${WL}_per_step.txt
cp $R/profiles/${TAG}_bench_${WL}_kernel_stats.csv $R/profiles/${TAG}_bench_${WL}_per_step.txt $R/gpurun_out/
if [ "$BENCH" = "1" ]; then
  cd $R && python3 bench.py --workload $WL > gpurun_out/${TAG}_${WL}_bench.log 2> gpurun_out/${TAG}_${WL}_bench.err
  tail -c 600 gpurun_out/${TAG}_${WL}_bench.log
fi
head -8 $R/gpurun_out/${TAG}_${WL}_per_step.txt
