#!/usr/bin/env python3
"""Per-kernel bound of the stages outside the metric pipeline: reads a counter summary folded by tools/fold_counters.py from
rocprofv3 passes over tools/bench_next.py (tools/collect_next.sh) and prints, per kernel, the launch time of the stats pass,
the fp32 VALU issue fraction (SQ_INSTS_VALU x 2 cycles / (1024 SIMDs x 2.4 GHz x launch time)) and the HBM traffic fraction
((FETCH_SIZE x 2 + WRITE_SIZE) / launch time / 8 TB/s).
usage: next_counters_table.py profiles/r05_next_counters.json"""
import json
import sys

N_SIMD, CLK, HBM = 1024, 2.4e9, 8e12
d = json.load(open(sys.argv[1]))
print("%-58s %9s %12s %10s %10s %9s  %s" % ("kernel", "avg us", "VALU insts", "issue", "HBM MB", "HBM frac", "bound"))
for name, e in sorted(d["kernels"].items(), key=lambda kv: -kv[1].get("avg_ns_in_stats_run", 0)):
    t = e.get("avg_ns_in_stats_run")
    if not t or name.startswith("__amd"):
        continue
    t *= 1e-9
    valu = e.get("SQ_INSTS_VALU", 0.0) * 2.0 / (N_SIMD * CLK) / t
    hbm = e.get("hbm_bytes_per_launch", 0.0) / t / HBM
    print("%-58s %9.1f %12.4g %10.3f %10.1f %9.3f  %s" % (name[:58], t * 1e6, e.get("SQ_INSTS_VALU", 0.0), valu,
                                                     e.get("hbm_bytes_per_launch", 0.0) / 1e6, hbm,
                                                     "valu-fp32" if valu >= hbm else "hbm"))
