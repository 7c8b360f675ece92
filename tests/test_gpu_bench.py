"""bench.py keeps its contract: ONE JSON line on stdout with the metric, the roofline of the dominant
kernel (measured with HIP events inside the timed steps) and the CPU baseline."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(*args):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True,
                         timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "stdout must carry exactly one line, got %d" % len(lines)
    return json.loads(lines[0])


def test_single_gpu_line():
    d = _run("--steps", "6", "--warmup", "2", "--res", "1024", "--grid", "2048")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "verified", "cold_ms"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["unit"] == "Mcells/s"
    assert d["higher_is_better"] is True and d["scaling"] == "strong" and d["vs_baseline"] is None and d["dtype"] == "f32"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] == pytest.approx(d["config"]["cells"] / (d["ms_per_step"] * 1e-3) / 1e6, rel=1e-3)
    assert d["verified"] is True   # the plane of the last timed step equals the oracle's, bit for bit
    assert set(d["stages"]) == {"noise", "gauss", "flow", "erosion"}
    r = d["roofline"]
    # the dominant kernel, under the bound that limits it: a fraction of a peak, never above it
    assert r["stage"] in d["stages"] and r["avg_launch_ms"] * r["launches_per_step"] <= d["ms_per_step"] * 1.05
    assert r["stage"] == max(d["stages"], key=lambda n: d["stages"][n]["avg_launch_ms"] * d["stages"][n]["launches"])
    if d["counters_source"] is not None:   # a counter summary of these kernel sources is committed
        assert r["bound"] in ("hbm", "valu-fp32") and 0.0 < r["frac"] <= 1.0
        assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=2e-3)
        if r["bound"] == "valu-fp32":   # ... and beside the executed-instruction rate, the algorithmic one (never above it)
            assert 0.0 < r["algorithmic"]["frac"] <= r["frac"] + 1e-3
        for s in d["stages"].values():
            assert 0.0 < s["valu_issue_frac"] <= 1.0 and 0.0 < s["hbm_traffic_frac"] <= 1.0
            assert s["bound"] == ("valu-fp32" if s["valu_issue_frac"] >= s["hbm_traffic_frac"] else "hbm")
            assert 0.0 < s["useful_valu_frac"] <= 1.0  # algorithmic lane-operations over executed ones
        assert 0.0 < d["step_valu"]["frac"] <= 1.0
    else:
        assert r["bound"] == "unknown" and r["frac"] is None
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "Mcells/s" and c["cores"] >= 1 and c["value"] > 0 and "-O" in c["sample"]
    assert d["in_place_entries"]["ms_per_step"] > 0 and d["grid_2048"]["recompute"]["Mcells/s"] > 0
    # the warm-up that ran is what the line says ran; the command taken literally is reported beside the steady state
    assert d["warmup_effective"] == 50 and d["config"]["preheat_steps"] == 48
    assert d["literal_command"]["ms_per_step"] > 0 and d["literal_command"]["Mcells/s"] > 0
    # the sharded grid's rows are compared with the oracle's after its timed passes
    assert d["grid_2048"]["verified"] is True and d["grid_2048"]["recompute"]["verified"] is True
    assert d["grid_2048"]["recompute"]["rows_checked"] == 2048
    # the float modes side by side: strict is bit-equal end to end, the tolerance modes are faster and are not
    fm = d["float_modes"]
    assert set(fm) >= {"strict", "fast", "relaxed"} and d["config"]["float_mode"] == "strict"
    assert fm["strict"]["end_to_end_vs_oracle"]["bit_equal"] is True
    assert fm["fast"]["end_to_end_vs_oracle"]["bit_equal"] is False and fm["fast"]["end_to_end_vs_oracle"]["max_abs"] < 2e-3
    assert fm["fast"]["ms_per_step"] > 0 and set(fm["relaxed"]["stages_ms"]) == {"noise", "gauss", "flow", "erosion"}
    # ... and the tolerance mode verifies itself the way the contract is stated: per stage, each fed the oracle's input
    assert fm["strict"]["stages_within_1e-5"] is True and fm["fast"]["stages_within_1e-5"] is True
    assert set(fm["fast"]["stages_vs_oracle"]) == {"noise", "gauss", "flow", "erosion"}
    assert all(v["cells_outside"] == 0 for v in fm["fast"]["stages_vs_oracle"].values())


def test_a_wrong_plane_fails_the_run():
    # debug hook: one cell of the downloaded plane is changed before it is compared with the oracle's -> the line is still
    # printed (verified: false, with the count), and the process does NOT exit 0
    env = dict(os.environ, NZ_BENCH_FLIP_CELL="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--res", "512",
                          "--grid", "0", "--no-extras"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["verified"] is False and d["verified_detail"]["cells_outside_1e-5_rel"] == 1
    assert out.returncode not in (0, None) and "FAILED" in out.stderr
    # the same command without the hook is clean
    ok = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--res", "512",
                         "--grid", "0", "--no-extras"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert ok.returncode == 0 and json.loads(ok.stdout.strip())["verified"] is True


def test_stripe_rehearsal_line():
    d = _run("--steps", "4", "--warmup", "1", "--as-rank", "1", "4", "--stripe-rows", "256", "--cols", "1024")
    assert d["n_gpus"] == 1 and "cpu_baseline" not in d and "roofline" in d
    assert "rehearsal of rank 1 of 4" in d["config"]["parallelism"] and d["config"]["cells"] == 256 * 1024
    # the rank's owned rows (global rows 256 .. 511 of the 1024-row grid) equal the oracle's rows of the monolithic grid
    assert d["verified"] is True and d["verified_detail"]["rows_checked"] == 256
    assert d["comm"]["ranks"][0]["owned_rows"] == [256, 512]


@pytest.mark.timeout(400, method="thread")
@pytest.mark.parametrize("halo", ["exchange", "exchange_once"])
def test_launched_like_the_driver_at_n_gt_1(halo):
    # the launcher the driver uses for N > 1 (torch.distributed.run, one rank per GPU, backend nccl), with the one rank a
    # one-GPU box has: process group on RCCL, the context on torch's stream, the stripe schedule with asynchronous
    # exchanges, the strong-scaling grid beside it, ONE line on rank 0's stdout
    import socket
    for attempt in range(3):  # (a port probed as free can be gone when the launcher's store binds it: EADDRINUSE, try another)
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                              "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                              "--gpus", "1", "--sharded", "--halo", halo, "--steps", "4", "--warmup", "1",
                              "--stripe-rows", "384", "--cols", "1024", "--grid", "1024", "--no-cpu-baseline"],
                             capture_output=True, text=True, timeout=380, cwd=ROOT)
        if out.returncode == 0 or "EADDRINUSE" not in out.stderr:
            break
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["config"]["cells"] == 384 * 1024 and d["value"] > 0
    assert "row-stripe" in d["config"]["parallelism"] and d["grid_1024"]["recompute"]["Mcells/s"] > 0


def _self_launched(*args):
    # plain `python bench.py --gpus N ...`, no launcher around it, exactly as the driver runs N = 1: bench.py starts the
    # ranks itself (torch.distributed.run) before it touches the GPU and relays rank 0's line
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True,
                         timeout=380, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "stdout must carry exactly one line, got %d" % len(lines)
    assert "starting" in out.stderr and "torch.distributed.run" in out.stderr
    return json.loads(lines[0])


@pytest.mark.timeout(400, method="thread")
def test_bench_starts_its_own_ranks_one_rank():
    # the one-GPU box drives the self-launch path with one rank: RCCL process group, stripe schedule, exchanges with
    # the rank's own neighbours absent (first and last stripe at once), `comm` in the line
    d = _self_launched("--gpus", "1", "--sharded", "--self-launch", "--steps", "4", "--warmup", "1", "--stripe-rows", "384",
                       "--cols", "1024", "--grid", "1024", "--halo", "exchange")
    assert d["n_gpus"] == 1 and d["config"]["cells"] == 384 * 1024 and d["value"] > 0
    assert d["verified"] is True and d["grid_1024"]["verified"] is True
    c = d["comm"]
    assert c["ranks"][0]["native_comm"] == {"rank": 0, "world": 1} and c["ranks"][0]["owned_rows"] == [0, 384]
    assert c["backend"] == "nccl" and c["world"] == 1 and c["halo"] == "exchange" and c["overlapped"] is False
    # one exchange per stencil launch: 4 filter launches + flow + erosion unless a fusion-depth knob regroups them
    regrouped = any(os.environ.get(k) for k in ("NZ_FLOW_NMAX", "NZ_EROSION_EMAX"))
    assert c["impl"].startswith("native") and (c["exchanges_per_step"] == 6 or regrouped) and c["host_enqueue_ms_idle_queue"] > 0.0
    assert c["exchange_ms_per_step"] is not None and c["exchange_ms_per_step"] >= 0.0 and c["rccl_version"]
    assert d["grid_1024"]["recompute"]["Mcells/s"] > 0


@pytest.mark.timeout(400, method="thread")
def test_bench_gpus_2_without_a_launcher():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible GPUs")
    # N = 2 exactly as the driver would type it: the 2048^2 grid split over two ranks, ghost rows exchanged over RCCL
    d = _self_launched("--gpus", "2", "--steps", "4", "--warmup", "1", "--grid", "2048", "--cols", "2048")
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["cells"] == 2048 * 2048
    assert d["verified"] is True and d["grid_2048"]["verified"] is True
    assert sorted(r["device"] for r in d["comm"]["ranks"]) == [0, 1] and all(r["native_comm"]["world"] == 2 for r in d["comm"]["ranks"])
    c = d["comm"]
    assert c["backend"] == "nccl" and c["world"] == 2 and c["halo"] == "exchange" and c["exchange_ms_per_step"] > 0.0
    assert set(d["grid_2048"]) >= {"recompute", "exchange", "exchange_interior_first", "exchange_border_first", "exchange_once"}
    assert c["impl"].startswith("native") and c["bytes_sent_per_step"] > 0
