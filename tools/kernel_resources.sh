#!/bin/bash
# Developer tool (GPU box): registers / LDS / grid of every kernel of a workload's step (from a rocprofv3 kernel trace).
# usage: bash tools/kernel_resources.sh <workload>
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/kres_$1
rocprofv3 --kernel-trace --output-format csv -d $OUT -o run -- python3 $R/bench.py --workload $1 --steps 12 --warmup 3 --windows 1 --no-extras > $OUT.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/*_kernel_trace.csv", recursive=True)[0]
seen = {}
for r in csv.DictReader(open(f)):
    k = (r["Kernel_Name"][:70], r.get("Grid_Size_X", r.get("Grid_Size")), r.get("Workgroup_Size_X", r.get("Workgroup_Size")))
    if k in seen: continue
    seen[k] = 1
    wg = int(k[2]); grid = int(k[1]) // max(1, wg)
    print(f"{k[0]:70s} wgs {grid:6d} x {wg:4d} thr  vgpr {r.get('VGPR_Count','?'):>4} agpr {r.get('Accum_VGPR_Count','?'):>4} lds {r.get('LDS_Block_Size','?'):>7} {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f} us")
PY
rm -rf $OUT
