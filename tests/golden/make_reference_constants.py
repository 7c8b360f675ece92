#!/usr/bin/env python3
"""Extracts the constants the reference holds beside the hot path -- enum orders, stage defaults, ErosionSettings' field
order and Reset() values, the demo assets' parameters -- into tests/golden/reference_constants.json (data only: names
and numbers, no source text).  tests/test_reference_constants.py checks the product's enums, defaults and the demo
parameters the tests use against it, so that none of them is a hand-typed expectation.

Sources (read as text, in the build container only):
  Noise/NoiseStage.cs:15-54                                   FractalNoise order, NoiseStage defaults
  Filter/Kernel/KernelJob.cs:79-94                            KernelFilterType order
  Filter/Kernel/Blur/BlurKernels.cs:8-27                      GaussSigma order, BlurHelper.max_width
  Geologic/Stage/FlowMapStage.cs:18-23                        FlowMapStage defaults
  Geologic/ParticleErosion/ScriptableObject/ErosionSettings.cs:8-92   field order, FLOW_LOSS_RATE initialiser, Reset()
  Geologic/ParticleErosion/LiveErosionDataTypes.cs            ErosionMode order, ErosionParameters field order
  BasicDemo~/*.asset, BasicDemo~/DynamicNoise.unity           the demo pipelines' stage parameters
"""
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def read(rel):
    return open(os.path.join(REF, rel)).read()


def enum(text, name):
    body = re.search(r"enum\s+%s\s*(?::\s*\w+\s*)?\{(.*?)\}" % name, text, re.S).group(1)
    body = re.sub(r"//.*", "", body)
    return [m.split("=")[0].strip() for m in body.split(",") if m.strip()]


def number(tok):
    tok = tok.strip().rstrip("f")
    if tok in ("true", "false"):
        return tok == "true"
    if tok.startswith("."):
        tok = "0" + tok
    if tok.startswith("-."):
        tok = "-0" + tok[1:]
    return float(tok) if re.search(r"[.eE]", tok) else int(tok)


def public_fields(text, cls):
    """public fields of class / struct `cls` in declaration order, with their initialisers (None if absent)."""
    rest = re.search(r"(?:class|struct)\s+%s\b[^{]*\{(.*)" % cls, text, re.S).group(1)
    depth, end = 1, 0
    for end, ch in enumerate(rest):  # the type's own body: up to its closing brace
        depth += (ch == "{") - (ch == "}")
        if depth == 0:
            break
    body = rest[:end]
    out = []
    for ty, name, init in re.findall(r"public\s+(int|float|bool|\w+)\s+(\w+)\s*(?:=\s*([^;]+))?;", body):
        if "(" in name:
            continue
        out.append([name, ty, number(init) if init and re.fullmatch(r"\s*-?[\d.]+f?\s*|\s*(?:true|false)\s*", init) else None])
    return out


def asset(rel):
    out = {}
    for line in read(rel).splitlines():
        m = re.match(r"\s+(\w+):\s+(-?[\d.]+(?:e-?\d+)?)\s*$", line)
        if m and not m.group(1).startswith("m_"):
            out[m.group(1)] = number(m.group(2))
    return out


def main():
    ns = read("Noise/NoiseStage.cs")
    kj = read("Filter/Kernel/KernelJob.cs")
    bk = read("Filter/Kernel/Blur/BlurKernels.cs")
    fm = read("Geologic/Stage/FlowMapStage.cs")
    es = read("Geologic/ParticleErosion/ScriptableObject/ErosionSettings.cs")
    ld = read("Geologic/ParticleErosion/LiveErosionDataTypes.cs")
    out = {"source": "xshazwar/noize-job: enum orders, stage defaults, ErosionSettings, BasicDemo~ assets (see the script's header)"}
    out["enums"] = {"FractalNoise": enum(ns, "FractalNoise"), "KernelFilterType": enum(kj, "KernelFilterType"),
                    "GaussSigma": enum(bk, "GaussSigma"), "ErosionMode": enum(ld, "ErosionMode")}
    out["BlurHelper.max_width"] = int(re.search(r"max_width\s*=\s*(\d+)", bk).group(1))
    out["NoiseStage.defaults"] = {n: v for n, _, v in public_fields(ns, "NoiseStage") if v is not None}
    out["FlowMapStage.defaults"] = {n: v for n, _, v in public_fields(fm, "FlowMapStage") if v is not None}
    fields = [f for f in public_fields(es, "ErosionSettings")]
    out["ErosionSettings.fields"] = [n for n, _, _ in fields]
    out["ErosionSettings.initialisers"] = {n: v for n, _, v in fields if v is not None}
    reset = re.search(r"void\s+Reset\s*\(\s*\)\s*\{(.*?)\n\s{8}\}", es, re.S).group(1)
    out["ErosionSettings.Reset"] = {}
    for n, v in re.findall(r"(\w+)\s*=\s*([^;]+);", reset):
        v = v.strip()
        out["ErosionSettings.Reset"][n] = v.split(".")[-1] if v.startswith("ErosionMode.") else number(v)
    out["ErosionParameters.fields"] = [n for n, _, _ in public_fields(ld, "ErosionParameters")]
    out["demo_assets"] = {name: asset("BasicDemo~/%s.asset" % name)
                          for name in ("Simplex", "Perl", "Sin", "GaussLF", "GaussHF", "Sobel2D", "FlowMapStage", "MeshTileStage")}
    scene = read("BasicDemo~/DynamicNoise.unity")
    out["demo_generator"] = {k: number(re.search(r"^\s+%s:\s+(-?[\d.]+)\s*$" % k, scene, re.M).group(1))
                             for k in ("tileHeight", "generatorResolution", "tileResolution", "margin")}
    out["demo_generator"]["tileSize"] = [number(v) for v in re.findall(r"^\s+tileSize:\s+(-?[\d.]+)\s*$", scene, re.M)]
    with open(os.path.join(HERE, "reference_constants.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write("\n")
    print("wrote reference_constants.json: %d enums, %d ErosionSettings fields, %d demo assets" % (
        len(out["enums"]), len(out["ErosionSettings.fields"]), len(out["demo_assets"])))


if __name__ == "__main__":
    main()
