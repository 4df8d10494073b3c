#!/usr/bin/env python3
"""Copy the rocprofv3 summaries of a gpurun profiling call from gpurun_out/prof
(scratch) into profiles/ (tracked), and derive profiles/pmc_traffic.json.

    python tools/collect_profiles.py r01_baseline

Expects gpurun_out/prof/{stats,fetch,write}/<host>/*_{kernel_stats,counter_collection}.csv written by
    rocprofv3 --kernel-trace --stats --output-format csv -d .../stats -- python3 bench.py ...
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d .../fetch -- python3 bench.py ...
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d .../write -- python3 bench.py ...
(counters in their own passes: FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2).

HBM bytes per launch follow /opt/skills/guides/MI355X_MICROARCH.md "HBM": the
counters are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request, so the
read side is doubled; WRITE_SIZE is taken as is.  The guide calibrated this for
16-B-per-lane streams; these kernels read 4/8 B per lane, so the read figure is
an upper estimate.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", "prof")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)


def one(pattern):
    # gpurun MERGES into gpurun_out/, so files of earlier calls linger: take the newest
    f = sorted(glob.glob(os.path.join(src, pattern)), key=os.path.getmtime)
    return f[-1] if f else None


ks = one("stats/*/*_kernel_stats.csv")
if ks:
    shutil.copy(ks, os.path.join(dst, f"{tag}_kernel_stats.csv"))
    print("kernel stats ->", f"profiles/{tag}_kernel_stats.csv")


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0].split("<")[0]


traffic = collections.defaultdict(dict)
for kind, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    f = one(f"{kind}/*/*_counter_collection.csv")
    if not f:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == ctr:
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        traffic[k][ctr + "_KiB_per_launch"] = sum(v) / len(v)
        traffic[k][ctr + "_launches"] = len(v)
for k, d in traffic.items():
    rd = d.get("FETCH_SIZE_KiB_per_launch", 0.0) * 2 * 1024      # gfx950: FETCH_SIZE = 1/2 of the bytes
    wr = d.get("WRITE_SIZE_KiB_per_launch", 0.0) * 1024
    d["hbm_read_bytes_per_launch"] = rd
    d["hbm_write_bytes_per_launch"] = wr
    d["hbm_bytes_per_launch"] = rd + wr
if traffic:
    traffic["_source"] = {"tag": tag, "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes, "
                          "bench.py workload (10k samples); read side doubled per MI355X_MICROARCH.md"}
    json.dump(traffic, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1, sort_keys=True)
    json.dump(traffic, open(os.path.join(dst, f"{tag}_pmc_traffic.json"), "w"), indent=1, sort_keys=True)
    print("pmc traffic ->", "profiles/pmc_traffic.json")
