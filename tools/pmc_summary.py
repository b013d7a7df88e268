#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output: per-kernel mean of every counter, and the
kernel-trace stats.  usage: pmc_summary.py DIR [DIR...]"""
import collections
import csv
import os
import sys

def find(d, name):
    for root, _, files in os.walk(d):
        if name in files:
            return os.path.join(root, name)
    return os.path.join(d, name)


for d in sys.argv[1:]:
    st = find(d, "bench_kernel_stats.csv")
    if os.path.exists(st):
        print("==", st)
        for r in csv.DictReader(open(st)):
            name = r["Name"].replace("(anonymous namespace)::", "")[:60]
            print("  %-62s calls %4s avg %10.1f ns  min %8s max %8s  %5s%%" % (
                name, r["Calls"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"], r["Percentage"][:5]))
    cc = find(d, "bench_counter_collection.csv")
    if os.path.exists(cc):
        print("==", cc)
        rows = list(csv.DictReader(open(cc)))
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        meta = {}
        for r in rows:
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:50]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[k] = (r["VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"], r["Grid_Size"])
        for k, v in agg.items():
            if not ("k_" in k):
                continue
            print("  %s  vgpr/sgpr/lds/scratch/grid=%s" % (k, meta[k]))
            for c, vals in sorted(v.items()):
                print("      %-24s %16.1f  (n=%d)" % (c, sum(vals) / len(vals), len(vals)))
