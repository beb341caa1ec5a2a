#!/usr/bin/env python
"""Summarise rocprofv3 output directories into small tracked files under profiles/.

    python tools/summarize_prof.py stats  <dir with *_kernel_stats.csv>  profiles/rNN_kernel_stats.md
    python tools/summarize_prof.py pmc    <dir of pmc passes>            profiles/rNN_pmc.json

`pmc` aggregates per kernel name: launches, average duration and every counter's per-launch average; FETCH_SIZE is
also reported corrected (x2: gfx950 tallies 128-byte requests at 64 B - MI355X_MICROARCH.md §HBM) in bytes."""
import collections
import csv
import glob
import json
import os
import sys


def short(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", "")
    return name[:110]


def stats(src, dst):
    f = sorted(glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True))[0]
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(dst, "w") as o:
        o.write(f"# rocprofv3 --kernel-trace --stats summary ({os.path.basename(f)})\n\n")
        o.write(f"total kernel time {tot / 1e6:.2f} ms\n\n| kernel | calls | total ms | avg us | % |\n|---|---:|---:|---:|---:|\n")
        for r in rows:
            o.write(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.3f} | "
                    f"{float(r['AverageNs']) / 1e3:.2f} | {float(r['Percentage']):.2f} |\n")
    print("wrote", dst)


def pmc(src, dst):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    out = {}
    for k, cs in agg.items():
        e = {"launches_per_pass": len(next(iter(cs.values()))), "avg_us": sum(dur[k]) / max(len(dur[k]), 1)}
        for c, v in cs.items():
            e[c] = sum(v) / len(v)
        if "FETCH_SIZE" in e:
            e["hbm_read_bytes_corrected"] = e["FETCH_SIZE"] * 1024 * 2
        if "WRITE_SIZE" in e:
            e["hbm_write_bytes"] = e["WRITE_SIZE"] * 1024
        if e.get("SQ_LDS_IDX_ACTIVE"):                # share of the LDS array's cycles that are bank-conflict replays
            e["lds_conflict_share"] = e.get("SQ_LDS_BANK_CONFLICT", 0.0) / e["SQ_LDS_IDX_ACTIVE"]
        out[k] = e
    json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
    print("wrote", dst, len(out), "kernels")


if __name__ == "__main__":
    {"stats": stats, "pmc": pmc}[sys.argv[1]](sys.argv[2], sys.argv[3])
