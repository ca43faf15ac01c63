#!/bin/bash
# Developer tool (GPU box): HBM traffic per kernel of the whole cfg2 step, separate --pmc passes -> gpurun_out/<tag>_pmc_hbm_per_kernel.csv
set -e
TAG=${1:-round}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc_$C -o run -- python3 $R/bench.py --no-extras --steps 30 --warmup 5 --eager > $R/gpurun_out/pmc_$C.log 2>&1
done
cd $R/tools && python3 pmc_hbm_step.py $R/gpurun_out/pmc_FETCH_SIZE $R/gpurun_out/pmc_WRITE_SIZE > $R/gpurun_out/${TAG}_pmc_hbm_per_kernel.csv
rm -rf $R/gpurun_out/pmc_FETCH_SIZE $R/gpurun_out/pmc_WRITE_SIZE
cat $R/gpurun_out/${TAG}_pmc_hbm_per_kernel.csv
