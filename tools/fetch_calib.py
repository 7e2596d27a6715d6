#!/usr/bin/env python3
"""FETCH_SIZE calibration factors from a `rocprofv3 --pmc FETCH_SIZE` run of tools/fetch_calib.
usage: tools/fetch_calib.py <rocprof dir> <stdout of fetch_calib> <out.json>"""
import csv
import glob
import json
import sys

src, log, out = sys.argv[1:4]
known = {}
for line in open(log):
    p = line.split()
    if len(p) == 3 and p[1] == "bytes":
        known[p[0]] = int(p[2])
    if len(p) == 2 and p[0] == "table_bytes":
        table = int(p[1])
res = {"what": "rocprofv3 FETCH_SIZE (KiB) against exactly known read volumes, gfx950; factor = true bytes / (FETCH_SIZE x 1024)",
       "tool": "tools/fetch_calib.hip", "table_bytes": table, "patterns": {}}
for f in glob.glob(src + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != "FETCH_SIZE":
            continue
        for k in known:
            if k + "(" in r["Kernel_Name"] or r["Kernel_Name"].startswith(k):
                fetched = float(r["Counter_Value"]) * 1024
                res["patterns"][k] = {"true_bytes": known[k], "fetch_size_bytes": fetched, "factor": known[k] / fetched}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
