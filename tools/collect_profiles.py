#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel trace stats + PMC passes) into small text/JSON files.
usage: collect_profiles.py <dir with trace/ pmc_*/ subdirs> [out.json]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    n = name.split("(")[0]
    for pre in ("void conp::", "conp::"):
        if n.startswith(pre):
            n = n[len(pre):]
    return n.split("<")[0][:60]


def main():
    root = sys.argv[1]
    out = {}
    # kernel stats
    for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
        rows = list(csv.DictReader(open(f)))
        ks = []
        for r in rows:
            ks.append(dict(kernel=short(r.get("Name", "")), calls=int(r.get("Calls", 0)),
                           total_ns=float(r.get("TotalDurationNs", 0)), avg_ns=float(r.get("AverageNs", 0)),
                           pct=float(r.get("Percentage", 0))))
        ks.sort(key=lambda k: -k["total_ns"])
        out["kernel_stats"] = ks
        print("== kernel trace stats (%s)" % os.path.relpath(f, root))
        for k in ks[:20]:
            print("%-44s calls %6d  avg %10.1f us  total %10.3f ms  %5.1f%%" % (k["kernel"], k["calls"], k["avg_ns"] / 1e3,
                                                                            k["total_ns"] / 1e6, k["pct"]))
    # per-launch durations of the update's kernels in launch order (kernel trace): the chip's clock governor needs tens of
    # launches after the setup phase to settle, so the all-calls average above sits above the steady state -- both are printed
    for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_trace.csv"), recursive=True):
        per = defaultdict(list)
        for r in csv.DictReader(open(f)):
            per[short(r["Kernel_Name"])].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
        out["kernel_trace_steady"] = {}
        print("== per-launch durations in launch order (%s): first 10 / last half" % os.path.relpath(f, root))
        for k, v in per.items():
            if len(v) < 20:
                continue
            v.sort()
            d = [x[1] / 1e3 for x in v]
            half = d[len(d) // 2:]
            out["kernel_trace_steady"][k] = dict(calls=len(d), first10_avg_us=sum(d[:10]) / 10, last_half_avg_us=sum(half) / len(half),
                                                 all_avg_us=sum(d) / len(d))
            print("%-44s calls %6d  first 10 avg %8.1f us  last half avg %8.1f us  all %8.1f us" %
                  (k, len(d), sum(d[:10]) / 10, sum(half) / len(half), sum(d) / len(d)))
    # PMC
    pmc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            pmc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out["pmc"] = {}
    print("== PMC averages per dispatch")
    for k, cs in pmc.items():
        # every kernel of the library that ran at least ten times (the update's kernels; round 4's name filter left out the
        # dominant kernels of two BASELINE configs).  torch's own kernels (fills, copies of the harness) are not the product's.
        if k.startswith(("at::", "void at::", "__amd_rocclr")) or max(len(v) for v in cs.values()) < 10:
            continue
        d = {c: sum(v) / len(v) for c, v in cs.items()}
        d["dispatches"] = max(len(v) for v in cs.values())
        out["pmc"][k] = d
        print(k)
        for c, v in sorted(d.items()):
            print("    %-28s %16.1f" % (c, v))
    if len(sys.argv) > 2:
        json.dump(out, open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
