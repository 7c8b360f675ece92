"""host-cs/ -- the C# host (source only: no .NET toolchain in the image) -- is checked against the C ABI by parsing:
Native.cs is what tools/gen_native_cs.py generates from include/noize_hip.h; every prototype matches its [DllImport] in
name, arity, order and type; every Native.nz_* call in the hand-written files names a declared entry with the right
argument count; the sequential structs carry their C counterparts' fields in order."""
import importlib.util
import os
import re

import pytest

from conftest import ROOT

spec = importlib.util.spec_from_file_location("gen_native_cs", os.path.join(ROOT, "tools", "gen_native_cs.py"))
gen = importlib.util.module_from_spec(spec)
spec.loader.exec_module(gen)
CS = os.path.join(ROOT, "host-cs")


def _cs_sources():
    out = {}
    for d, _, files in os.walk(CS):
        for f in files:
            if f.endswith(".cs"):
                out[os.path.relpath(os.path.join(d, f), CS)] = open(os.path.join(d, f)).read()
    return out


def test_native_cs_is_generated_from_the_header():
    assert open(os.path.join(CS, "Native.cs")).read() == gen.generate(), "run tools/gen_native_cs.py"


def test_every_prototype_has_its_dllimport_with_the_same_shape():
    header = gen.parse_header()
    cs = gen.parse_cs()
    assert len(header) >= 85 and {n for n, _, _ in header} == set(cs)
    for name, ret, params in header:
        cs_ret, cs_params = cs[name]
        assert cs_ret == gen.cs_return(ret), name
        assert len(cs_params) == len(params), name
        for (ctype, pname), got in zip(params, cs_params):
            assert got == gen.cs_type(ctype, pname, name), (name, pname)
        # the reference's delegates end in (JobHandle dependency) and return a JobHandle: (dep, out) close the list
        if params and params[-1] == ("nz_handle*", "out") and len(params) >= 2 and params[-2][0] == "nz_handle":
            assert cs_params[-2:] == ["ulong", "out ulong"], name
    # ... and every entry that ends in `nz_handle *out` has ONE overload taking the pointer itself (IntPtr.Zero = no handle,
    # no event record): the same parameters up to it
    over = gen.parse_cs_overloads()
    with_out = {n for n, _, ps in header if ps and ps[-1] == ("nz_handle*", "out")}
    assert set(over) == with_out and len(with_out) >= 60
    for name in with_out:
        assert over[name] == cs[name][1][:-1] + ["IntPtr"], name
    # scalar order of three delegates, spelled out against the C# reference signatures
    f = dict((n, [p for _, p in ps]) for n, _, ps in header)
    assert f["nz_fractal"] == ["ctx", "noiseType", "src", "resolution", "hurst", "startingAmplitude", "stepdown", "detuneRate",
                               "octaves", "xpos", "zpos", "noiseSize", "dep", "out"]            # Fractal.cs:76-88
    assert f["nz_heightmap_mesh"] == ["ctx", "meshType", "vertices", "indices", "resolution", "inputResolution", "marginPix",
                                      "tileHeight", "tileSize", "heights", "dep", "out"]       # HeightMapMeshJob.cs:55-65
    assert f["nz_flowmap_update_water"] == ["ctx", "waterMap", "waterMap__buff", "flowMapN", "flowMapS", "flowMapE", "flowMapW",
                                            "resolution", "dep", "out"]                        # FlowMapJob.cs:154-165


def _split_args(s):
    args, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            args.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        args.append(cur.strip())
    return args


def test_hand_written_files_call_declared_entries_with_the_right_arity():
    cs = gen.parse_cs()
    over = gen.parse_cs_overloads()
    calls = 0
    for path, text in _cs_sources().items():
        if path == "Native.cs":
            continue
        for m in re.finditer(r"Native\.(nz_\w+)\s*\(", text):
            name = m.group(1)
            assert name in cs, "%s calls undeclared %s" % (path, name)
            i, depth = m.end(), 1
            while depth:
                depth += {"(": 1, ")": -1}.get(text[i], 0)
                i += 1
            args = _split_args(text[m.end():i - 1])
            assert len(args) == len(cs[name][1]), "%s: %s takes %d arguments, called with %d" % (path, name, len(cs[name][1]), len(args))
            for k, (a, t) in enumerate(zip(args, cs[name][1])):   # out / ref at the call site where the declaration has them
                if t.startswith("out "):
                    # the last argument may pick the overload that leaves the handle out
                    null_handle = k == len(args) - 1 and t == "out ulong" and a == "IntPtr.Zero" and name in over
                    assert a.startswith("out ") or null_handle, (path, name, a)
                if t.startswith("ref "):
                    assert a.startswith("ref "), (path, name, a)
            calls += 1
    assert calls >= 25
    stages = _cs_sources()["Stages/Stages.cs"]
    for cls in ("NoiseStage", "KernelFilterStage", "StageGaussianBlur", "StageSmoothBlur", "ErosionStage", "FlowMapStage",
                "MeshTileStage", "ConstantStage", "ReduceStage", "CurveStage", "CropStage", "StageThermalErosion"):
        assert re.search(r"class %s\s*:" % cls, stages), cls
    # the other halves of the operator API: the fan-in pipeline and the live-erosion driver with every job of its cycle
    assert re.search(r"class ReducePipeline\s*:\s*BasePipeline", _cs_sources()["Pipeline/ReducePipeline.cs"])
    # the multi-GPU half: communicator + the sharded stage list, every entry a host needs to run BASELINE config 5
    sharded = _cs_sources()["Pipeline/ShardedPipeline.cs"]
    for entry in ("nz_comm_unique_id", "nz_comm_init", "nz_comm_destroy", "nz_sharded_create", "nz_sharded_pipeline",
                  "nz_sharded_stripe", "nz_sharded_destroy", "nz_sharded_map_range", "nz_sharded_normalize"):
        assert "Native.%s(" % entry in sharded, entry
    live = _cs_sources()["LiveErosion/LiveErosion.cs"]
    for job in ("nz_thermal_erosion", "nz_fill_beyer_queue", "nz_queued_beyer_cycle", "nz_process_beyer_erosive_events",
                "nz_clear_particle_queue", "nz_erode_height_maps", "nz_update_flow_from_track", "nz_erode_height_maps_and_flow",
                "nz_pool_automata_job",
                "nz_set_rgba32", "nz_curviture_map"):
        assert "Native.%s(" % job in live, job


def test_sequential_structs_match_the_c_structs():
    header = gen.strip_comments(open(gen.HEADER).read())
    runtime = open(os.path.join(CS, "Runtime.cs")).read()

    def c_fields(name):
        body = re.search(r"typedef\s+struct\s+%s\s*\{(.*?)\}\s*%s\s*;" % (name, name), header, re.S).group(1)
        out = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            ctype, rest = decl.split(" ", 1)
            for n in rest.split(","):
                n = n.strip()
                ct = ctype + "*" if n.startswith("*") else ctype   # a pointer member
                n = n.lstrip("*")
                m = re.match(r"(\w+)\[(\d+)\]", n)
                out += [(ct, "%s_%s" % (m.group(1), "xy"[k])) for k in range(int(m.group(2)))] if m else [(ct, n)]
        return out

    def cs_fields(name):
        body = re.search(r"public struct %s\s*\{(.*?)\}" % name, runtime, re.S).group(1)
        out = []
        for decl in re.findall(r"public\s+(\w+)\s+([^;]+);", body):
            out += [(decl[0], n.strip()) for n in decl[1].split(",")]
        return out

    kind = {"int32_t": "int", "float": "float", "uint32_t": "uint"}
    for cname, csname in (("nz_stripe", "NzStripe"), ("nz_rw_tile", "NzRwTile"), ("nz_erosion_params", "NzErosionParams"),
                          ("nz_tile_set_meta", "NzTileSetMeta"), ("nz_terrain_params", "NzTerrainParams"),
                          ("nz_sharded_desc", "NzShardedDesc")):
        cf, sf = c_fields(cname), cs_fields(csname)
        assert [n for _, n in cf] == [n for _, n in sf], cname
        for (ct, n), (st, _) in zip(cf, sf):
            assert st == ("IntPtr" if ct not in kind else kind[ct]), (cname, n)


def test_the_cs_host_has_every_class_the_python_host_exports():
    # host-cs/ is level with the two hosts the test suite drives: every public class / enum of noize_job_amd (the operator
    # API: payloads, stages, pipelines, the state manager with its context stages, persistence, the live erosion) exists in
    # the C# sources under the same name -- or under the name the .NET side gives it
    import ast
    pkg = os.path.join(ROOT, "noize_job_amd")
    exported = set()
    for mod in ("pipeline.py", "pipeline_state.py", "persistence.py", "live_erosion.py", "runtime.py", "sharded.py"):
        tree = ast.parse(open(os.path.join(pkg, mod)).read())
        exported |= {n.name for n in tree.body if isinstance(n, ast.ClassDef) and not n.name.startswith("_")}
    renamed = {"Context": "GpuContext", "JobHandle": "GpuJobHandle", "NativeComm": "GpuComm", "ShardedGrid": "ShardedPipeline"}
    # Python-side plumbing with no counterpart in a compiled host: the mesh buffers are two members of MeshStageData, the
    # Python schedule of the sharded path (sharded.py) is nz_sharded_* behind the C ABI for a compiled host
    python_only = {"MeshBuffers", "StripePlan", "PipelineParams", "HipStripeOps", "TorchComm", "NoComm"}
    text = "\n".join(_cs_sources().values())
    missing = []
    for name in sorted(exported - python_only):
        cs_name = renamed.get(name, name)
        if not re.search(r"\b(class|struct|enum)\s+%s\b" % re.escape(cs_name), text):
            missing.append(cs_name)
    assert not missing, missing
    # the members the context stages and the batch payload rest on
    state = _cs_sources()["PipelineState/PipelineState.cs"]
    for member in ("GetBuffer", "GetBufferNoLoad", "SaveBufferToDisk", "BufferExists", "ReleaseBuffer", "IsLocked", "TrySetLock",
                   "RegisterCallback", "RemoveCallback", "TriggerUpdateCallbacks", "SetSavePath", "OnDestroy"):
        assert re.search(r"public\s+[\w<>\[\]]+\s+%s\s*\(" % member, state), member
    serde = _cs_sources()["PipelineState/PipelineSerialization.cs"]
    for member in ("WriteData", "ReadData", "CachedSize", "GetFQN", "SetCount", "GetCount", "FlushToDisk", "FromFile"):
        assert re.search(r"\b%s\s*\(" % member, serde), member
    assert '"NativeArray`1"' in serde and "save__" in serde and "files.json" in serde
    stages = _cs_sources()["Stages/Stages.cs"]
    for entry in ("nz_fractal_batch", "nz_kernel_filter_stage_batch", "nz_gauss_blur_stage_batch", "nz_smooth_blur_stage_batch",
                  "nz_erosion_stage_batch", "nz_flowmap_stage_batch", "nz_heightmap_mesh_batch"):
        assert "Native.%s(" % entry in stages, entry
    ctxs = _cs_sources()["Stages/ContextStages.cs"]
    assert ctxs.count("Native.nz_flush_write_slice(") == 2 and "Native.nz_handle_record(" in ctxs
    assert "stageManager" in _cs_sources()["Pipeline/Pipeline.cs"] and "nz_ctx_set_float_mode" in _cs_sources()["Runtime.cs"]


def test_the_cs_index_reader_skips_unknown_fields_and_decodes_every_escape():
    # the reference reads files.json with JsonUtility.FromJson (unknown fields ignored, full JSON escapes); the hand-rolled
    # C# reader must not be stricter (tests/test_persistence.py holds the same fixture for the Python host)
    src = open(os.path.join(ROOT, "host-cs", "PipelineState", "PipelineSerialization.cs")).read()
    assert "unknown field" not in src and src.count("SkipValue(text, ref i)") == 2
    body = src[src.index("static void SkipValue"):src.index("static string ReadString")]
    for needle in ("ReadString(t, ref i)", "c == '{' || c == '['", "SkipValue(t, ref i)", "Expect(t, ref i, close)"):
        assert needle in body, needle
    rs = src[src.index("static string ReadString"):]
    for esc in ("'n'", "'t'", "'r'", "'b'", "'f'", "'u'"):
        assert "== " + esc in rs, esc


def _cs_bracket_depths(src):
    """A C# lexer as far as nesting goes: line / block comments, char literals, regular, verbatim (@) and interpolated ($) strings
    are skipped; returns the final depth of { ( [ and raises when a closer comes before its opener or a literal does not end."""
    i, n = 0, len(src)
    depth = {"{": 0, "(": 0, "[": 0}
    closers = {"}": "{", ")": "(", "]": "["}
    while i < n:
        c = src[i]
        if src.startswith("//", i):
            j = src.find("\n", i)
            i = n if j < 0 else j
        elif src.startswith("/*", i):
            j = src.find("*/", i)
            assert j >= 0, "block comment without an end"
            i = j + 2
        elif c == "'":
            j = i + (3 if src[i + 1] == "\\" else 2)
            assert src[j] == "'", "char literal without an end: %r" % src[i:i + 8]
            i = j + 1
        elif c == '"' or (c in "$@" and (src[i + 1:i + 2] == '"' or (src[i + 1:i + 2] in ("$", "@") and src[i + 2:i + 3] == '"'))):
            verbatim = False
            while src[i] in "$@":
                verbatim |= src[i] == "@"
                i += 1
            i += 1
            while True:
                assert i < n, "string without an end"
                if verbatim and src[i] == '"' and src[i + 1:i + 2] == '"':
                    i += 2
                elif not verbatim and src[i] == "\\":
                    i += 2
                elif src[i] == '"':
                    i += 1
                    break
                else:
                    i += 1
        else:
            if c in depth:
                depth[c] += 1
            elif c in closers:
                depth[closers[c]] -= 1
                assert depth[closers[c]] >= 0, "a %s before its opener near %r" % (c, src[max(0, i - 40):i + 1])
            i += 1
    return depth


def test_every_cs_file_lexes_and_nests():
    # no compiler has seen host-cs/ (no .NET in the image): the least a source file owes is literals that end and brackets that
    # pair up -- checked with a small lexer, so that an edit which breaks a file's structure fails here
    seen = 0
    for d, _, files in os.walk(os.path.join(ROOT, "host-cs")):
        for f in files:
            if f.endswith(".cs"):
                with open(os.path.join(d, f)) as fh:
                    assert _cs_bracket_depths(fh.read()) == {"{": 0, "(": 0, "[": 0}, f
                seen += 1
    assert seen >= 10
    assert _cs_bracket_depths("class A { void f() { if (x) { } }")["{"] == 1      # (the checker itself has power)
    with pytest.raises(AssertionError):
        _cs_bracket_depths("class A { } }")
    assert _cs_bracket_depths('class A { string s = $"{{"; char c = \'}\'; /* { */ }') == {"{": 0, "(": 0, "[": 0}
