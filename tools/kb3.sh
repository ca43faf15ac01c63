#!/bin/bash
# Developer tool (GPU box): the fp32 ring GEMM with its ring depth forced (kbench_s2: two slots, two workgroups per CU;
# kbench_s5: five slots, one workgroup per CU) and what each ingredient of the main loop costs (timing-only variants).
# usage: bash tools/kb3.sh [variants...]
cd $GRAFT_REPO_ROOT
for v in ${@:-kbench_s5 kbench_s2 kbench_nodma kbench_nobar kbench_noload kbench_nomma}; do
  echo "== $v"
  for args in "1024 64 1024 1024" "1024 16 1856 768" "1024 64 1024 1024 8" "1024 16 1856 768 8"; do
    timeout -k 5 60 ./tools/kbench/$v gj $args 2>&1 | grep -v "^occupancy"
  done
done
