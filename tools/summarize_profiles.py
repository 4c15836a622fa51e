#!/usr/bin/env python3
"""Turns gpurun_out/<tag>/{stats,fetch,write} into profiles/<tag>_kernel_stats.csv and
profiles/<tag>_traffic.json (HBM bytes per launch per kernel, with the gfx950 corrections of
MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE counts wide coalesced
reads at half their size, so it is doubled)."""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = f"gpurun_out/{tag}"


def per_kernel(counter_dir, counter):
    files = sorted(glob.glob(f"{src}/{counter_dir}/*/*counter_collection.csv"), key=os.path.getmtime)[-1:]   # newest run only
    agg = collections.defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            m = re.search(r"(k_[a-z0-9_]+)", r["Kernel_Name"])
            if m:
                agg[m.group(1)].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


try:
    build_id = open(f"{src}/build_id.txt").read().strip()   # flacgpu_build_id() of the library the passes ran
except FileNotFoundError:
    build_id = None
fetch = per_kernel("fetch", "FETCH_SIZE")
write = per_kernel("write", "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write)):
    fb = fetch.get(k, 0.0) * 1024 * 2   # KiB -> bytes, x2 gfx950 wide-read correction
    wb = write.get(k, 0.0) * 1024
    out[k] = {"fetch_bytes": round(fb), "write_bytes": round(wb), "hbm_bytes": round(fb + wb),
              "raw_FETCH_SIZE_KiB": fetch.get(k), "raw_WRITE_SIZE_KiB": write.get(k)}
json.dump({"_build_id": build_id, **out}, open(f"profiles/{tag}_traffic.json", "w"), indent=1)
# VALU issue counters (per launch, summed over the chip).  SQ_ACTIVE_INST_VALU and SQ_WAVE_CYCLES
# count quad-cycles (MI355X_MICROARCH.md), SQ_INSTS_VALU counts wave-instructions.
valu = {}
for c in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_WAVES"):
    for k, v in per_kernel("valu", c).items():
        valu.setdefault(k, {})[c] = v
for k, v in valu.items():
    if v.get("SQ_WAVE_CYCLES"):
        v["valu_active_per_wave_cycle"] = round(v.get("SQ_ACTIVE_INST_VALU", 0.0) / v["SQ_WAVE_CYCLES"], 4)
    if v.get("SQ_INSTS_VALU") and v.get("SQ_WAVES"):
        v["valu_insts_per_wave"] = round(v["SQ_INSTS_VALU"] / v["SQ_WAVES"], 1)
if valu:
    json.dump({"_build_id": build_id, **valu}, open(f"profiles/{tag}_valu.json", "w"), indent=1)
stats = sorted(glob.glob(f"{src}/stats/*/*kernel_stats.csv"), key=os.path.getmtime)[-1:]   # newest run only
if stats:
    shutil.copy(stats[0], f"profiles/{tag}_kernel_stats.csv")
for name in ("bench.json", "bench_line.json"):     # the full record and the compact line the driver parses
    try:
        shutil.copy(f"{src}/{name}", f"profiles/{tag}_{name}")
    except FileNotFoundError:
        pass
for k, v in out.items():
    print(k.ljust(18), f"{v['hbm_bytes']/1e6:10.1f} MB  (fetch {v['fetch_bytes']/1e6:.1f}, write {v['write_bytes']/1e6:.1f})")
