#!/bin/bash
# Developer tool (GPU box): the fp32 tile GEMM with its variant forced (kbench_s2: 512-thread workgroups, two per CU; kbench_s16:
# paired 1024-thread workgroups) and what each ingredient of the main loop costs (timing-only variants).
# usage: bash tools/kb3.sh [variants...]
cd $GRAFT_REPO_ROOT
for v in ${@:-kbench_s2 kbench_s16 kbench_nodma kbench_nobar kbench_noload kbench_nomma}; do
  echo "== $v"
  for args in "1024 64 1024 1024" "1024 16 1856 768" "1024 64 1024 1024 8" "1024 16 1856 768 8"; do
    timeout -k 5 60 ./tools/kbench/$v gj $args 2>&1 | grep -v "^occupancy"
  done
done
