#!/bin/bash
# Developer tool (GPU box): premasked-backward parity tests + phase stamps of the fp32 tile GEMM (gemm_jobs.h).  usage: bash tools/kb.sh
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python3 -m pytest tests/test_gpu_kernels.py -q -m gpu -k "premasked" 2>&1 | tail -6
for args in "1024 16 1856 768" "1024 16 1856 768 2" "1024 16 1856 768 8" "1024 64 1024 1024" "1024 64 1024 1024 2" "1024 64 1024 1024 8" "1024 16 1856 256" "4096 16 1856 256"; do
  timeout -k 5 60 ./tools/kbench/kbench gj $args 2>&1 | grep -v "^occupancy"
done
