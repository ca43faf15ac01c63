#!/usr/bin/env python3
"""Developer tool: HBM bytes per launch of EVERY kernel of the training step from two rocprofv3 PMC passes
(FETCH_SIZE, WRITE_SIZE; corrections as tools/pmc_traffic.py: KiB units, FETCH_SIZE doubled on gfx950).
usage: python tools/pmc_hbm_step.py <fetch_dir> <write_dir>  -> csv on stdout (kernel, launches, fetch_MB, write_MB, total_MB)"""
import csv
import re
import sys

from pmc_traffic import mean_counter


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"_ZN3emb(\d+)(\w+)", name)
    if m:
        return "emb::" + m.group(2)[: int(m.group(1))]
    return name.split("(")[0][:60]


fetch, write = mean_counter(sys.argv[1], "FETCH_SIZE"), mean_counter(sys.argv[2], "WRITE_SIZE")
out = csv.writer(sys.stdout)          # kernel names contain commas: quoted
out.writerow(["kernel", "fetch_MB_per_launch", "write_MB_per_launch", "hbm_MB_per_launch"])
rows = []
for k in sorted(set(fetch) | set(write)):
    if "emb" not in k:
        continue
    f, w = 2 * fetch.get(k, 0.0) * 1024 / 1e6, write.get(k, 0.0) * 1024 / 1e6
    rows.append((f + w, short(k), f, w))
for t, k, f, w in sorted(rows, reverse=True):
    out.writerow([k, f"{f:.3f}", f"{w:.3f}", f"{t:.3f}"])
