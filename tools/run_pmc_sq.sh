#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/pmc_sq -o run -- python3 $R/bench.py --no-extras --steps 30 --warmup 5 --eager > $R/gpurun_out/pmc_sq.log 2>&1
python3 $R/tools/pmc_sq.py $R/gpurun_out/pmc_sq > $R/gpurun_out/pmc_sq_per_kernel.csv
rm -rf $R/gpurun_out/pmc_sq
cat $R/gpurun_out/pmc_sq_per_kernel.csv | cut -c1-60,110-400 | head -30
