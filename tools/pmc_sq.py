#!/usr/bin/env python3
"""Developer tool: per-kernel averages of SQ counters from a rocprofv3 --pmc run.
On the GPU box:
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY \\
      --output-format csv -d gpurun_out/pmc_sq -o run -- python3 bench.py --no-extras --steps 30 --warmup 5 --eager
then:  python tools/pmc_sq.py gpurun_out/pmc_sq > profiles/<tag>_pmc_sq_per_kernel.csv
(--eager: counters are collected per dispatch, graph replays are not instrumented.  SQ_ACTIVE_INST_* / SQ_WAIT_* count
quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles summed over SIMDs -- see MI355X_MICROARCH.md.)"""
import collections, csv, glob, os, sys

d = sys.argv[1]
f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[k].add(r["Dispatch_Id"])
names = sorted({c for v in agg.values() for c in v})
w = csv.writer(sys.stdout)
w.writerow(["kernel", "dispatches"] + names + ["lds_conflict_per_active_lds", "valu_per_wave"])
for k in sorted(agg, key=lambda k: -agg[k].get("SQ_BUSY_CYCLES", 0) / max(1, len(cnt[k]))):
    n = len(cnt[k])
    v = agg[k]
    if n < 5 or k.startswith("__amd") or "at::" in k:
        continue
    row = [k[:110], n] + [round(v.get(c, 0) / n) for c in names]
    act = v.get("SQ_ACTIVE_INST_LDS", 0)
    row.append(round(v.get("SQ_LDS_BANK_CONFLICT", 0) / act, 3) if act else "")
    row.append(round(v.get("SQ_INSTS_VALU", 0) / max(1.0, v.get("SQ_WAVES", 0))))
    w.writerow(row)
