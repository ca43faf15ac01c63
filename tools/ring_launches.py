#!/usr/bin/env python3
"""Developer tool: the launches of the fp32 tile GEMM (gemm_jobs_kernel) of one step, in launch order, from a rocprofv3 kernel trace of `bench.py --no-extras`
(average duration of each over the last ten steps, with its grid size).
usage: python tools/ring_launches.py <dir with *_kernel_trace.csv> [marker substring of the FIRST kernel of a step]"""
import csv, glob, os, sys

d = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "ncl_to_nlc"
f = glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
steps = []
for a, b in zip(idx[-12:-2], idx[-11:-1]):
    steps.append([(r["Kernel_Name"][:48], int(r.get("Grid_Size_X", r.get("Grid_Size", 0))) // max(1, int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)))),
                   (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows[a:b] if "gemm_jobs" in r["Kernel_Name"]])
n = len(steps[0])
tot = 0.0
for i in range(n):
    t = sum(s[i][2] for s in steps) / len(steps)
    tot += t
    print(f"{i}: {steps[0][i][0]:48s} tiles {steps[0][i][1]:5d}  {t:8.2f} us")
print(f"tile GEMM launches per step: {n}, {tot:.1f} us")
