#!/usr/bin/env python3
"""Folds the rocprofv3 passes of one bench.py configuration into profiles/<tag>_counters.json, the file bench.py
reads for `stages.*.valu_issue_frac / hbm_traffic_frac` and `roofline`.

Inputs (each from its own `rocprofv3 --kernel-trace --pmc <group>` run of the same command, as
/opt/skills/guides/MI355X_MICROARCH.md prescribes -- counters never share a pass with other trace domains):
  fetch  : FETCH_SIZE   (KiB; reports half the bytes of wide coalesced reads on gfx950 -> x2)
  write  : WRITE_SIZE   (KiB; exact for 16-byte-per-lane stores)
  sq     : SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE
  stats  : the *_kernel_stats.csv of a --kernel-trace --stats run (average duration per kernel)
Values are means per launch over the launches sampled.

usage: fold_counters.py --fetch F.csv --write W.csv --sq S.csv --stats K.csv --out OUT.json
                        --res 4096 --flush swap --float-mode strict --commit <sha> [--note "..."]
"""
import argparse
import collections
import csv
import hashlib
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    m = re.search(r"(\w+(?:<[^>(]*>)?)\(", name)
    return m.group(1) if m else name


def agg(path):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    if path and os.path.exists(path):
        for r in csv.DictReader(open(path)):
            d[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return d


def kernel_sources_sha():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "noize_job_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".cpp", ".hpp")) or name == "Makefile":
            with open(os.path.join(d, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def main():
    ap = argparse.ArgumentParser()
    for k in ("fetch", "write", "sq", "stats", "out", "commit", "note", "flush"):
        ap.add_argument("--" + k, default=None)
    ap.add_argument("--float-mode", default="strict")
    ap.add_argument("--res", type=int, default=4096)
    a = ap.parse_args()
    mean = lambda v: sum(v) / len(v)  # noqa: E731
    f, w, s = agg(a.fetch), agg(a.write), agg(a.sq)
    stats = {}
    if a.stats and os.path.exists(a.stats):
        for r in csv.DictReader(open(a.stats)):
            stats[short(r["Name"])] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                                       "pct": float(r["Percentage"])}
    kernels = {}
    for k in sorted(set(f) | set(w) | set(s)):
        e = {}
        if "FETCH_SIZE" in f.get(k, {}):
            e["fetch_bytes_per_launch"] = 2.0 * 1024.0 * mean(f[k]["FETCH_SIZE"])
        if "WRITE_SIZE" in w.get(k, {}):
            e["write_bytes_per_launch"] = 1024.0 * mean(w[k]["WRITE_SIZE"])
        e["hbm_bytes_per_launch"] = e.get("fetch_bytes_per_launch", 0.0) + e.get("write_bytes_per_launch", 0.0)
        for c, v in s.get(k, {}).items():
            e[c] = mean(v)
        e["launches_sampled"] = max([len(v) for v in list(f.get(k, {}).values()) + list(s.get(k, {}).values())] + [0])
        if e.get("GRBM_GUI_ACTIVE") and e.get("SQ_INSTS_VALU"):
            # GRBM_GUI_ACTIVE comes back summed over the 8 XCDs; 128 SIMDs per XCD; a wave64 VALU instruction
            # occupies its SIMD for 2 cycles
            e["valu_issue_utilisation_in_profiler"] = round(e["SQ_INSTS_VALU"] * 2.0 / (e["GRBM_GUI_ACTIVE"] * 128), 4)
        if k in stats:
            e["avg_ns_in_stats_run"] = stats[k]["avg_ns"]
            e["pct_of_kernel_time"] = stats[k]["pct"]
        kernels[k] = e
    out = {"config": {"res": a.res, "flush": a.flush, "float_mode": a.float_mode, "sharded": False, "commit": a.commit,
                      "kernel_sources_sha": kernel_sources_sha(),
                      "command": "python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --res %d --flush %s --float-mode %s "
                                 "(counter passes); --steps 60 --warmup 20 (stats pass)" % (a.res, a.flush, a.float_mode),
                      "note": a.note or ""},
           "corrections": "FETCH_SIZE KiB x2 (gfx950 under-report of wide reads), WRITE_SIZE KiB x1; means per launch",
           "kernels": kernels}
    json.dump(out, open(a.out, "w"), indent=1)
    for k, e in kernels.items():
        print("%-40s n=%4d  valu %.3e  hbm %7.1f MB  avg %8.1f us" % (k, e["launches_sampled"], e.get("SQ_INSTS_VALU", 0),
              e["hbm_bytes_per_launch"] / 1e6, e.get("avg_ns_in_stats_run", 0) / 1e3))


if __name__ == "__main__":
    main()
