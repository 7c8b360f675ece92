#!/usr/bin/env python3
"""Folds two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE, collected in separate runs as
/opt/skills/guides/MI355X_MICROARCH.md prescribes) into profiles/<tag>_pmc.json.

Units and gfx950 corrections (same guide, section HBM): FETCH_SIZE / WRITE_SIZE are in KiB;
FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read on gfx950, so it is
doubled; WRITE_SIZE is exact for 16-byte-per-lane stores.  Values are per launch (mean over the
launches of a kernel in the run).

usage: pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> [note]
"""
import collections
import csv
import json
import re
import sys

STAGE_OF = [("fractal_simplex_tab_kernel", "noise"), ("fractal_tab2_kernel", "noise"), ("fractal_kernel", "noise"),
            ("conv_reg_kernel", "gauss"), ("conv_pass", "gauss_pass"), ("erosion_reg_kernel", "erosion"),
            ("min_pass_kernel", "erosion_pass"), ("flow_fused_kernel", "flow"), ("velocity_kernel", "flow_velocity")]


def short(name):
    m = re.search(r"(\w+(?:<[^>(]*>)?)\(", name.replace("(anonymous namespace)::", ""))
    return m.group(1) if m else name


def agg(path, counter):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            d[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return d


def main():
    fetch, write, out = sys.argv[1:4]
    note = sys.argv[4] if len(sys.argv) > 4 else ""
    f, w = agg(fetch, "FETCH_SIZE"), agg(write, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(f) | set(w)):
        fb = 2.0 * 1024.0 * sum(f[k]) / len(f[k]) if f.get(k) else None
        wb = 1024.0 * sum(w[k]) / len(w[k]) if w.get(k) else None
        e = {"launches_sampled": len(f.get(k, [])), "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb,
             "hbm_bytes_per_launch": (fb or 0) + (wb or 0)}
        kernels[k] = e
        for pat, stage in STAGE_OF:
            if k.startswith(pat.split("<")[0]) and (("<" not in pat) or pat in k):
                kernels.setdefault(stage, e)
    json.dump({"note": note, "corrections": "FETCH_SIZE KiB x2 (gfx950 under-report), WRITE_SIZE KiB x1",
               "kernels": kernels}, open(out, "w"), indent=1)
    for k, e in kernels.items():
        print("%-34s n=%3d fetch %8.1f MB  write %8.1f MB" % (k, e["launches_sampled"],
              (e["fetch_bytes_per_launch"] or 0) / 1e6, (e["write_bytes_per_launch"] or 0) / 1e6))


if __name__ == "__main__":
    main()
