#!/bin/bash
# Developer tool (GPU box): kernel-trace evidence for every BASELINE configuration.  usage: bash tools/profile_workloads.sh <tag>
TAG=${1:-round3_a}
cd $GRAFT_REPO_ROOT
for wl in cfg2 cfg3 cfg4 cfg5 cfg2_b4096; do
  echo "=== $wl"
  timeout -k 10 300 bash tools/round_profile.sh $TAG $wl 0 || exit 1
done
