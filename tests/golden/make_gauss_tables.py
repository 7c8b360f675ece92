#!/usr/bin/env python3
"""Extracts the Gaussian / fixed kernel coefficient literals the reference holds for the separable
filter path into tests/golden/gauss_tables.json (data only: numbers, no source text).

Sources (read as text, in the build container only):
  /root/reference/Filter/Kernel/KernelJob.cs:97-136          gauss{9,7,5,3}_s{1,2}, smooth3, sobel3, prewitt3
  /root/reference/Filter/Kernel/Blur/BlurKernels.cs:59-318   kernels[GaussSigma][12 widths]
These literals are the only reference-held known answers on the hot path; the oracle's table
generator (exp(-i^2/2s^2)/sum in double, rounded to fp32) is pinned against them in
tests/test_oracle_tables.py.
"""
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
NUM = r"[-+]?(?:\d+\.?\d*|\.\d+)(?:[eE][-+]?\d+)?"


def floats(body):
    return [float(m) for m in re.findall(r"(%s)f" % NUM, body)]


def main():
    out = {"source": "xshazwar/noize-job Filter/Kernel/KernelJob.cs:97-136, Filter/Kernel/Blur/BlurKernels.cs:59-318",
           "fixed": {}, "by_sigma": {}}
    kj = open(os.path.join(REF, "Filter/Kernel/KernelJob.cs")).read()
    for name, body in re.findall(r"public static float\[\] (\w+) = \{(.*?)\};", kj, re.S):
        out["fixed"][name] = floats(body)
    for name, expr in re.findall(r"public static float (\w+Factor)\s*=\s*(.*?);", kj):
        vals = floats(expr + " ")
        out["fixed"][name] = vals  # e.g. smooth3Factor = [1.0, 3.0] meaning 1f / 3f
    bk = open(os.path.join(REF, "Filter/Kernel/Blur/BlurKernels.cs")).read()
    for sigma, block in re.findall(r"GaussSigma\.(s\dd\d\d)\s*, new List<float\[\]>\(\) \{(.*?)\n\s*\}\s*\n\s*\}", bk, re.S):
        rows = [floats(b) for b in re.findall(r"new float\[\] \{(.*?)\}", block, re.S)]
        out["by_sigma"][sigma] = rows
    assert len(out["by_sigma"]) == 16, sorted(out["by_sigma"])
    for s, rows in out["by_sigma"].items():
        assert [len(r) for r in rows] == list(range(3, 26, 2)), (s, [len(r) for r in rows])
    with open(os.path.join(HERE, "gauss_tables.json"), "w") as f:
        json.dump(out, f, indent=0, separators=(",", ":"))
    print("wrote gauss_tables.json:", len(out["fixed"]), "fixed tables,", len(out["by_sigma"]), "sigmas")


if __name__ == "__main__":
    main()
