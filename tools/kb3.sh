#!/bin/bash
# Developer tool (GPU box): SQ counters of the kbench kernels
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for b in kbench; do
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/kb3_$b -o run -- $R/tools/kbench/$b gj 1024 16 1856 256 > $R/gpurun_out/kb3_$b.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/kb3b_$b -o run -- $R/tools/kbench/$b gj 1024 16 1856 256 >> $R/gpurun_out/kb3_$b.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/kb3*_kbench")):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not f: print(d, "no csv"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"][:40]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in agg.items():
        print(d, k, {a: round(b) for a, b in v.items()})
PY
