#!/usr/bin/env python3
"""Developer tool: per-step kernel breakdown from a rocprofv3 kernel trace of `bench.py --no-extras`.
usage: python tools/per_step.py <dir with *_kernel_trace.csv> [marker substring of the FIRST kernel of a step]
Averages the last ten complete steps (graph replays): launches per step, time per step and per launch for every kernel."""
import collections, csv, glob, os, sys

d = sys.argv[1] if len(sys.argv) > 1 else "."
marker = sys.argv[2] if len(sys.argv) > 2 else "ncl_to_nlc"
f = glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
out = []
for a, b in zip(idx[-12:-2], idx[-11:-1]):
    agg = collections.OrderedDict()
    for r in rows[a:b]:
        e = agg.setdefault(r["Kernel_Name"], [0, 0.0])
        e[0] += 1
        e[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    out.append(((int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3, agg))
print(f"step wall (profiled) mean us: {sum(o[0] for o in out) / len(out):.1f}")
tot = {}
for _, agg in out:
    for n, (c, t) in agg.items():
        e = tot.setdefault(n, [0, 0.0])
        e[0] += c
        e[1] += t
busy = 0.0
for n, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    busy += t / len(out)
    print(f"{t / len(out):9.1f} us/step  x{c / len(out):5.1f}  avg {t / c:8.2f} us  {n[:150]}")
print(f"busy us/step: {busy:.1f}   launches/step: {sum(c for c, _ in tot.values()) / len(out):.0f}")
