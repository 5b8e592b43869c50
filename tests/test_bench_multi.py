"""The `bench.py --gpus N` path the driver launches on a multi-GPU node, exercised on ONE GPU (-m gpu): N ranks oversubscribed
on the device (every exchange staged through host memory over the authenticated control plane -- RCCL refuses two ranks on
one device), the N = 1 run of the same cloud for the 64-bit fingerprint of all descriptor rows, and one emulated rank.
The programs are started by tests/_launcher.py, a process that never touches the GPU."""
import json
import os
import sys

import pytest

from conftest import run_program

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

COMMON = ["--points-per-gpu", "60000", "--radius", "0.08", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--sustained-seconds", "0",
          "--no-density", "--no-defaults", "--no-dropin", "--no-normals", "--no-ransac", "--parity-rows", "60"]


LINE_KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
             "dtype", "data", "config", "roofline", "cpu_baseline", "detail"}


def record_of(r):
    """stdout carries exactly ONE line, short enough for the driver to recover (round 4's 23.6 KB line came back unparsed);
    every other block of the run is in the side file the line names.  Returns the side file's record with the line under
    `_line`."""
    assert r["rc"] == 0, r["stderr"][-2000:]
    lines = [ln for ln in r["stdout"].splitlines() if ln.strip()]
    assert len(lines) == 1, r["stdout"][-2000:]
    assert len(lines[0]) < 4096, len(lines[0])
    line = json.loads(lines[0])
    assert LINE_KEYS <= set(line), sorted(LINE_KEYS - set(line))
    assert set(line["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    with open(os.path.join(ROOT, line["detail"])) as f:
        d = json.load(f)
    for k in ("value", "n_gpus", "steps", "warmup"):
        assert line[k] == d[k] or abs(line[k] - d[k]) <= 1e-5 * abs(d[k]), (k, line[k], d[k])
    d["_line"] = line
    return d


def bench(*flags, timeout=900):
    return record_of(run_program([sys.executable, "bench.py", *COMMON, *flags], timeout=timeout))


@pytest.fixture(scope="module")
def one_rank():
    return bench("--gpus", "1", "--checksum", "--no-match")


def test_one_rank_record_is_scale_ready(one_rank):
    d = one_rank
    assert d["n_gpus"] == 1 and d["emulated"] is False and d["rccl_ranks"] == 1
    assert d["parity"]["ok"] and d["value"] > 0 and len(d["per_rank_ms_per_step"]) == 1
    assert d["roofline"]["frac"] > 0 and "value_includes" in d


def test_two_ranks_on_one_device_reproduce_the_one_rank_rows(one_rank):
    """`bench.py --gpus 2 --oversubscribe` as the driver would start it (it spawns its ranks itself): rendezvous, the
    descriptor pass with the neighbour exchange of SPFH rows, strong-scaling pass of the N = 1 cloud, config 5's tail.  The
    strong-scaling pass runs the SAME 60000-point cloud as the N = 1 run: equal 64-bit fingerprints = every FPFH and SHOT
    row bit-identical."""
    d = bench("--gpus", "2", "--oversubscribe", "--checksum", "--match-rows", "20000", "--match-steps", "1")
    assert d["n_gpus"] == 2 and d["emulated"] is False
    assert d["parity"]["ok"] and all(p["ok"] for p in d["parity"]["per_rank"]) and len(d["parity"]["per_rank"]) == 2
    assert len(d["per_rank_ms_per_step"]) == 2
    assert "staged through host memory" in d["config"]["exchange"] and d["rccl_ranks"] == 0
    ss = d["strong_scaling"]
    assert ss["parity_ok"] and ss["checksum"] == one_rank["checksum"], (ss["checksum"], one_rank["checksum"])
    em = d["exchange_match"]
    assert em["matches"] > 0 and em["matches_recovering_true_correspondence"] > 0.8


def test_three_ranks_oversubscribed(one_rank):
    d = bench("--gpus", "3", "--oversubscribe", "--checksum", "--no-match")
    assert d["parity"]["ok"] and len(d["per_rank_ms_per_step"]) == 3
    assert d["strong_scaling"]["checksum"] == one_rank["checksum"]


def test_emulated_rank_record_cannot_be_read_as_a_multi_gpu_result():
    d = bench("--gpus", "8", "--emulate-rank", "3", "--no-match")
    assert d["emulated"] is True and d["emulated_rank"] == 3 and d["emulated_world"] == 8
    assert d["n_gpus"] == 1 and d["value"] is None and d["projected_value_upper_bound"] > 0
    assert d["parity"]["ok"]


def test_secondary_keys_of_the_one_rank_record():
    """The keys the default run adds behind the timed steps, at a small size: the sustained window, the slot-capacity line and
    the two-stream step (same rows, the two chains side by side) -- and the parity block, read AFTER them, still holds."""
    r = run_program([sys.executable, "bench.py", "--points-per-gpu", "60000", "--radius", "0.08", "--steps", "3", "--warmup", "1",
                     "--no-cpu-baseline", "--sustained-seconds", "0.05", "--sustained-steps", "20", "--no-density", "--no-defaults",
                     "--no-dropin", "--no-normals", "--no-match", "--parity-rows", "60"], timeout=900)
    d = record_of(r)
    assert d["_line"]["sustained"]["steps"] >= 20
    assert d["sustained"]["steps"] >= 20 and d["k2_slot_capacity"]["ms_per_step_with_a_sample_counted_in_every_search"] > 0
    assert d["two_streams"]["ms_per_step"] > 0 and 0.5 < d["two_streams"]["vs_timed_steps"] < 1.5
    assert d["parity"]["ok"]
