#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python3 -m pytest tests/test_gpu_kernels.py -q -m gpu -k "premasked" 2>&1 | tail -3
for args in "1024 16 1856 256" "4096 16 1856 256" "512 32 3712 768"; do
  for k in bwd bwdpre; do timeout -k 5 60 ./tools/kbench/kbench $k $args 2>&1 | grep -v "^occupancy\|in-kernel"; done
done
