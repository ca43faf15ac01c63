#!/usr/bin/env python3
"""Developer tool: mean L2 counters per launch and kernel from a rocprofv3 --pmc pass (tools/run_pmc_l2.sh)."""
import csv, glob, os, sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))
for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        acc[row["Kernel_Name"]][row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
out = csv.writer(sys.stdout)
out.writerow(["kernel", "launches", "TCP_TCC_READ_REQ", "TCC_HIT", "TCC_MISS", "l2_hit_rate"])
for k, cs in sorted(acc.items()):
    if "emb" not in k:
        continue
    m = {c: sum(v.values()) / len(v) for c, v in cs.items()}
    n = len(next(iter(cs.values())))
    h, mi = m.get("TCC_HIT_sum", 0.0), m.get("TCC_MISS_sum", 0.0)
    out.writerow([k.split("(")[0][:90], n, round(m.get("TCP_TCC_READ_REQ_sum", 0.0)), round(h), round(mi),
                  round(h / (h + mi), 3) if h + mi else ""])
