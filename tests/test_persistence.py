"""The reference's tile persistence format (Pipeline/PipelineState/PipelineSerialization.cs), host side."""
import json
import os

import numpy as np
import pytest


def test_layout_and_round_trip(tmp_path):
    from noize_job_amd.persistence import PipelineSerdeManager, clean_file_name
    m = PipelineSerdeManager(str(tmp_path), "terrain", "v1")
    plane = np.random.default_rng(3).random((32, 32), dtype=np.float32)
    m.WriteData(plane, "tile_0_0")
    m.WriteData(np.arange(7, dtype=np.int32), "ids")
    base = tmp_path / "save__terrain"
    assert sorted(os.listdir(base)) == ["data", "files.json"]
    assert sorted(os.listdir(base / "data")) == ["ids.data", "tile_0_0.data"]
    assert (base / "data" / "tile_0_0.data").stat().st_size == 32 * 32 * 4
    assert (base / "data" / "tile_0_0.data").read_bytes() == plane.astype("<f4").tobytes()  # raw little-endian
    idx = json.loads((base / "files.json").read_text())
    assert list(idx) == ["alias", "version", "files"] and idx["alias"] == "terrain" and idx["version"] == "v1"
    # typeof(T).Name with T = NativeArray<float> / NativeArray<int> (PipelineStateManager.cs:64,111): the container's CLR name
    assert idx["files"] == [{"id": "tile_0_0", "type": "NativeArray`1", "size": 1024},
                            {"id": "ids", "type": "NativeArray`1", "size": 7}]
    m2 = PipelineSerdeManager(str(tmp_path), "terrain", "v1")  # a fresh manager finds the index
    assert m2.CachedSize("tile_0_0") == 1024 and m2.CachedSize("missing") == -1
    assert np.array_equal(m2.ReadData("tile_0_0").reshape(32, 32), plane)
    assert m2.ReadData("missing") is None
    m2.WriteData(plane[:16], "tile_0_0")  # SetCount updates in place
    assert PipelineSerdeManager(str(tmp_path), "terrain", "v1").CachedSize("tile_0_0") == 512
    assert clean_file_name("a/b//c..") == "a_b_c" and clean_file_name("plain") == "plain"


def test_reads_an_index_written_by_the_reference_and_by_version_1(tmp_path):
    from noize_job_amd.persistence import PipelineSerdeManager
    base = tmp_path / "save__terrain"
    (base / "data").mkdir(parents=True)
    plane = np.arange(16, dtype=np.float32)
    plane.astype("<f4").tofile(base / "data" / "a.data")
    plane.astype("<f4").tofile(base / "data" / "b.data")
    # JsonUtility.ToJson(FileDirectory): what the reference's SaveBufferToDisk<float, NativeArray<float>> leaves behind,
    # next to an entry in this package's first-version spelling
    (base / "files.json").write_text('{"alias":"terrain","version":"v1","files":[{"id":"a","type":"NativeArray`1","size":16},'
                                     '{"id":"b","type":"Single","size":16}]}')
    m = PipelineSerdeManager(str(tmp_path), "terrain", "v1")
    assert m.CachedSize("a") == 16 and m.CachedSize("b") == 16 and m.CachedSize("c") == -1
    assert np.array_equal(m.ReadData("a"), plane)


def test_an_index_touched_by_another_tool_still_loads(tmp_path):
    # JsonUtility.FromJson ignores fields it does not know and decodes every JSON escape; so do the hosts (the C# reader is
    # hand-rolled: host-cs/PipelineState/PipelineSerialization.cs SkipValue / ReadString, checked in tests/test_host_cs.py)
    from noize_job_amd.persistence import PipelineSerdeManager
    base = tmp_path / "save__terrain"
    (base / "data").mkdir(parents=True)
    np.arange(4, dtype="<f4").tofile(base / "data" / "a.data")
    (base / "files.json").write_text(
        '{ "alias": "terrain", "tool": {"name": "x", "tags": [1, 2, {"k": null}]}, "version": "v1",\n'
        '  "files": [ {"id": "a", "checksum": "ab\\/cd", "type": "NativeArray`1", "size": 4, "dirty": false},\n'
        '             {"id": "tab\\there\\r\\b\\f\\u0041", "type": "NativeArray`1", "size": 9} ], "saved": 1.5e3 }')
    m = PipelineSerdeManager(str(tmp_path), "terrain", "v1")
    assert m.CachedSize("a") == 4 and np.array_equal(m.ReadData("a"), np.arange(4, dtype=np.float32))
    assert m.CachedSize("tab\there\r\b\fA") == 9


@pytest.mark.gpu
def test_device_tile_round_trip(nj, ctx, tmp_path):
    from noize_job_amd.persistence import PipelineSerdeManager
    res = 64
    st = nj.NoiseStage(ctx, nj.FractalNoise.Cellular, 0.4, 1.0, 5, 2.0, 0.0, 100)
    d = nj.GeneratorData("t", ctx.alloc(res * res), res, 0, 0)
    st.ReceiveHandledInput(nj.PipelineWorkItem(d), nj.JobHandle())
    st.jobHandle.Complete()
    m = PipelineSerdeManager(str(tmp_path), "gpu", "1")
    m.WriteData(d.data, "heights")
    back = ctx.alloc(res * res)
    m.ReadData("heights", target=back)
    assert np.array_equal(back.ToArray(), d.data.ToArray())
