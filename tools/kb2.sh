#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python3 -m pytest tests/test_gpu_kernels.py -q -m gpu -k "premasked" 2>&1 | tail -3
for b in kbench; do
for args in "1024 16 1856 768" "1024 64 1024 1024" "1024 64 1024 1024 2" "1024 16 1856 256"; do
  echo "== $b"; timeout -k 5 60 ./tools/kbench/$b gj $args 2>&1 | grep -v "^occupancy"
done; done
timeout -k 5 60 ./tools/kbench/kbench bwd 1024 16 1856 256 2>&1 | grep -v "^occupancy"
