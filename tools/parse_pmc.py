#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs: python tools/parse_pmc.py <dir> <kernel-substring>"""
import csv, glob, sys, collections
d, sub = sys.argv[1], sys.argv[2]
for f in sorted(glob.glob(d + "/*/*/*counter_collection.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print("%-28s n=%d  last=%.6g  mean=%.6g" % (k, len(v), v[-1], sum(v) / len(v)))
for f in sorted(glob.glob(d + "/*/*/*kernel_trace.csv"))[:1]:
    ds = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if sub in r["Kernel_Name"]]
    if ds:
        print("duration us: last %.1f mean %.1f (n=%d)" % (ds[-1] / 1e3, sum(ds) / len(ds) / 1e3, len(ds)))
