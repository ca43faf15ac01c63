#!/bin/bash
# Developer tool (GPU box): L2 requests / hits / misses per kernel of the whole cfg2 step -> gpurun_out/<tag>_pmc_l2_per_kernel.csv
set -e
TAG=${1:-round}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/pmc_l2 -o run -- python3 $R/bench.py --no-extras --steps 30 --warmup 5 --eager > $R/gpurun_out/pmc_l2.log 2>&1
python3 $R/tools/pmc_l2.py $R/gpurun_out/pmc_l2 > $R/gpurun_out/${TAG}_pmc_l2_per_kernel.csv
rm -rf $R/gpurun_out/pmc_l2
cat $R/gpurun_out/${TAG}_pmc_l2_per_kernel.csv
