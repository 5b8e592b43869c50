"""Host-side pieces of the drop-in (no GPU): filters, Kabsch solver / RigidTransform,
histogram edges, shard planning."""
import numpy as np
import pytest

from conftest import load_golden, synth_cloud


def test_kabsch_and_rigid_transform_match_golden_draws():
    from shot_fpfh_amd.core import RigidTransform, solver_point_to_point

    g = load_golden("ransac_500.npz")
    a, b = g["scan_kp"][g["scan_idx"]], g["ref_kp"][g["ref_idx"]]
    for d, rt in list(zip(g["draws"], g["draw_rt"]))[:50]:
        tf = solver_point_to_point(a[d], b[d])
        assert np.array_equal(tf.as_row12(), rt)
        assert abs(np.linalg.det(tf.rotation) - 1.0) < 1e-9
    tf = RigidTransform(g["rotation"], g["translation"])
    assert np.allclose(tf[a[:5]], a[:5] @ g["rotation"].T + g["translation"])
    inv = ~tf
    assert np.array_equal(inv.rotation, g["rotation"].T) and np.array_equal(inv.translation, -g["translation"])
    assert "\n" in repr(tf)


def test_ransac_draw_stream_is_the_reference_stream():
    g = load_golden("ransac_500.npz")
    gen = np.random.default_rng(seed=72)
    mine = np.stack([gen.choice(500, 4, replace=False, shuffle=False) for _ in range(len(g["draws"]))])
    assert np.array_equal(mine, g["draws"])


def test_vectorised_draw_stream_equals_the_per_draw_generator_calls():
    """matching.ransac.draw_stream produces, from the generator's raw output, what n_draws successive
    rng.choice(n, size, replace=False, shuffle=False) return (ransac.py:50-55) and leaves the generator in the state those calls
    leave it in: the reference's recorded draws, populations where Floyd's algorithm meets repeats, where the bounded-integer
    sampler rejects, a pending half-word in the generator, and a second call continuing the stream."""
    import shot_fpfh_amd.matching.ransac as R

    g = load_golden("ransac_500.npz")
    assert R._replica_matches_this_numpy()  # (else every call below is the loop itself)
    assert np.array_equal(R.draw_stream(np.random.default_rng(seed=72), 500, 4, len(g["draws"])), g["draws"])
    for seed, n, size, k, pre in ((72, 1_000_000, 4, 10_000, 0), (1, 9, 4, 4_000, 1), (2, 60_000_000, 4, 6_000, 0), (3, 100_000, 7, 1_500, 1),
                                  (4, 5, 4, 300, 0), (5, 2_000_000_000, 4, 500, 0), (6, 30, 2, 40, 1)):
        a, b = np.random.default_rng(seed), np.random.default_rng(seed)
        for gen in (a, b):
            for _ in range(pre):
                gen.integers(0, 10, dtype=np.uint32)  # leaves a buffered 32-bit half-word behind
        first = R.draw_stream(a, n, size, k)
        assert np.array_equal(first, R._draws_by_loop(b, n, size, k)), (seed, n, size)
        assert (np.sort(first, axis=1)[:, 1:] != np.sort(first, axis=1)[:, :-1]).all() and first.min() >= 0 and first.max() < n
        assert a.bit_generator.state == b.bit_generator.state
        assert np.array_equal(R.draw_stream(a, n, size, 77), R._draws_by_loop(b, n, size, 77))
        assert np.array_equal(a.random(4), b.random(4))


def test_filters():
    from shot_fpfh_amd.matching import left_median_filter, quantile_filter, threshold_filter

    d = np.array([0.0, 0.5, 1.0, 2.0, 8.0, 0.25])
    assert threshold_filter(d, 4).tolist() == [True, True, True, False, False, True]
    assert quantile_filter(d, (0.2, 0.8)).tolist() == [False, True, True, True, False, True]
    # (median + index of the first non-zero distance) / 2 -- the reference's own formula
    assert left_median_filter(d).tolist() == [(x <= 0.75) and (x >= (0.75 + 1) / 2) for x in d]


def test_fpfh_edges_are_histogramdd_edges():
    from shot_fpfh_amd.engine import fpfh_edges

    rng = np.random.default_rng(0)
    sample = rng.random((50, 3)) * 2 - 1
    for nb in (3, 4, 5, 8):
        _, edges = np.histogramdd(sample, bins=nb, range=[(-1, 1), (-1, 1), (-np.pi / 2, np.pi / 2)])
        e = fpfh_edges(nb)
        assert e.shape == (3, nb + 1)
        for a in range(3):
            assert np.array_equal(e[a], edges[a])


def test_shard_plan_partitions_exactly():
    from shot_fpfh_amd.sharding import ShardPlan

    for n in (0, 1, 7, 1000, 1_000_003):
        for world in (1, 2, 3, 8):
            blocks = [ShardPlan(n, world, r).block() for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            assert all(e - b <= ShardPlan(n, world, 0).rows_per_rank for b, e in blocks)
    with pytest.raises(ValueError):
        ShardPlan(10, 2, 2)


def test_api_surface_matches_reference_signatures():
    import inspect

    from shot_fpfh_amd.descriptors import ShotMultiprocessor, compute_fpfh_descriptor, compute_normals
    from shot_fpfh_amd.matching import basic_matching, match_descriptors, ransac_on_matches

    p = list(inspect.signature(compute_fpfh_descriptor).parameters)
    assert p[:8] == ["keypoints_indices", "cloud_points", "normals", "radius", "n_bins", "decorrelated", "verbose",
                     "disable_progress_bars"]
    p = inspect.signature(compute_normals).parameters
    assert list(p)[:2] == ["query_points", "cloud_points"] and p["k"].kind is inspect.Parameter.KEYWORD_ONLY
    sm = ShotMultiprocessor()
    assert (sm.normalize, sm.share_local_rfs, sm.min_neighborhood_size, sm.n_procs) == (True, True, 100, 8)
    p = list(inspect.signature(sm.compute_descriptor_single_scale).parameters)
    assert p == ["point_cloud", "normals", "keypoints", "radius", "subsampling_voxel_size"]
    p = list(inspect.signature(match_descriptors).parameters)
    assert p[:6] == ["scan_descriptors", "ref_descriptors", "filter_callback", "filter_nonreciprocal", "verbose", "n_min_matches"]
    p = inspect.signature(ransac_on_matches).parameters
    assert (p["n_draws"].default, p["draw_size"].default, p["distance_threshold"].default) == (10000, 4, 1)
    assert list(inspect.signature(basic_matching).parameters)[:2] == ["scan_descriptors", "ref_descriptors"]


def test_keypoint_selection_host_branches_match_reference_golden():
    """The branches of keypoint_selection.py that need no device work (the seeded random draws); the voxel and
    radius-search branches are GPU tests."""
    import importlib

    import shot_fpfh_amd.keypoint_selection as ks

    g = load_golden("keypoints_6k.npz")
    p = g["cloud"]
    importlib.reload(ks)  # fresh module-level default_rng(1), as in a fresh process of the reference
    assert np.array_equal(ks.select_keypoints_randomly(p, 50), g["random_points"])
    idx = ks.select_query_indices_randomly(100, 10)
    assert idx.shape == (10,) and np.unique(idx).size == 10 and idx.max() < 100


def test_point_to_plane_solver_recovers_a_small_motion():
    from scipy.spatial.transform import Rotation

    from shot_fpfh_amd.core import solver_point_to_plane

    rng = np.random.default_rng(3)
    ref = rng.random((400, 3))
    nrm = rng.standard_normal((400, 3))
    nrm /= np.linalg.norm(nrm, axis=1)[:, None]
    rot = Rotation.from_euler("xyz", [2e-3, -1e-3, 1.5e-3]).as_matrix()
    t = np.array([1e-3, -2e-3, 5e-4])
    scan = (ref - t) @ rot  # ref = scan @ rot.T + t
    tf = solver_point_to_plane(scan, ref, nrm)
    assert np.abs(tf.rotation - rot).max() < 1e-5 and np.abs(tf.translation - t).max() < 1e-5


def test_ply_reader_and_writer_against_a_file_the_reference_wrote(tmp_path):
    import os

    from conftest import GOLDEN as GOLDEN_DIR
    from shot_fpfh_amd.helpers import get_data, read_ply, write_ply

    g = load_golden("ply_40.npz")
    ref_file = os.path.join(GOLDEN_DIR, "ref_written_40.ply")
    data = read_ply(ref_file)
    assert list(data.dtype.names) == list(g["names"])
    assert np.array_equal(np.vstack((data["x"], data["y"], data["z"])).T, g["points"])
    assert np.array_equal(np.vstack((data["nx"], data["ny"], data["nz"])).T, g["normals"])
    assert np.array_equal(data["label"], g["labels"]) and np.array_equal(data["blue"], g["colors"][:, 2])
    out = str(tmp_path / "mine")  # extension appended
    assert write_ply(out, [g["points"], g["normals"], g["colors"], g["labels"]], list(g["names"]))
    assert open(out + ".ply", "rb").read() == open(ref_file, "rb").read()  # byte for byte the reference's file
    assert write_ply(out, [g["points"], g["labels"][:5]], ["x", "y", "z", "l"]) is False
    assert write_ply(out, [g["points"]], ["x", "y"]) is False
    # get_data: stored normals orient the recomputed ones; duplicates are dropped on request
    calls = {}

    def fake_normals(query_points, cloud_points, *, k=None, radius=None, pre_computed_normals=None):
        calls["args"] = (query_points.shape, k, radius, pre_computed_normals is not None)
        return -pre_computed_normals if pre_computed_normals is not None else np.ones_like(query_points)

    pts, nrm = get_data(ref_file, k=7, normals_computation_callback=fake_normals)
    assert calls["args"] == ((40, 3), 7, None, True) and np.array_equal(nrm, -g["normals"]) and pts.shape == (40, 3)
    pts2, nrm2 = get_data(ref_file, recompute_normals=False, remove_duplicates=True)
    assert pts2.shape[0] == 40 and np.array_equal(np.sort(pts2, axis=0), np.sort(g["points"].astype(pts2.dtype), axis=0))
    bare = str(tmp_path / "bare.ply")
    write_ply(bare, g["points"], ["x", "y", "z"])
    with pytest.raises(ValueError):
        get_data(bare)
    with open(str(tmp_path / "ascii.ply"), "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex 0\nend_header\n")
    with pytest.raises(ValueError):
        read_ply(str(tmp_path / "ascii.ply"))


def test_every_mirrored_function_keeps_the_reference_signature():
    """tests/golden/api_signatures.json (tools/gen_api_signatures.py) lists the reference's parameters, kinds and
    defaults; the drop-in may only ADD optional keyword-only parameters (e.g. `engine=`)."""
    import importlib
    import inspect
    import json
    import os

    from conftest import GOLDEN

    recorded = json.load(open(os.path.join(GOLDEN, "api_signatures.json")))
    rename = {"shot_fpfh.descriptors.pca_based_descriptors": "shot_fpfh_amd.descriptors",
              "shot_fpfh.helpers.io_ply": "shot_fpfh_amd.helpers", "shot_fpfh.core": "shot_fpfh_amd.core"}
    relocated = {"shot_fpfh.core:compute_point_to_point_error": "shot_fpfh_amd.icp"}
    assert len(recorded) >= 40
    for key, params in recorded.items():
        module, name = key.split(":")
        mine = relocated.get(key) or rename.get(module) or module.replace("shot_fpfh", "shot_fpfh_amd", 1)
        obj = importlib.import_module(mine)
        for part in name.split("."):
            obj = getattr(obj, part)
        got = [[n, p.kind.name, None if p.default is inspect.Parameter.empty else repr(p.default)]
               for n, p in inspect.signature(inspect.unwrap(obj)).parameters.items()]
        names = {n for n, _, _ in params}
        assert [g for g in got if g[0] in names] == params, f"{key}: {got} != {params}"
        for n, kind, default in got:
            if n not in names:
                assert default is not None and kind == "KEYWORD_ONLY", f"{key}: extra parameter {n} must be an optional keyword"


def test_double_matching_with_rejects_raises_what_the_reference_raises():
    """matching.py:172-221 is broken in the reference: every call ends in an exception.  Which one -- IndexError from the
    indexing at :202 / :220, ValueError from np.argpartition(kth=2) or np.divide(out=) -- for a table of shapes the reference
    was called with (tools/gen_golden_r6.py -> tests/golden/double_matching_errors.json)."""
    import json
    import os

    from conftest import GOLDEN
    from shot_fpfh_amd.matching import double_matching_with_rejects

    with open(os.path.join(GOLDEN, "double_matching_errors.json")) as fh:
        table = json.load(fh)["cases"]
    assert len(table) >= 12 and {c["raises"] for c in table} == {"IndexError", "ValueError"}
    rng = np.random.default_rng(0)
    for c in table:
        a, b = rng.random((c["scan_rows"], 12)) + 0.1, rng.random((c["ref_rows"], 12)) + 0.1
        a[c["scan_nonempty"]:], b[c["ref_nonempty"]:] = 0.0, 0.0
        with pytest.raises(Exception) as info:
            double_matching_with_rejects(a, b, 0.8, verbose=False)
        assert type(info.value).__name__ == c["raises"], (c, info.value)


def test_stacked_kabsch_is_bit_identical_to_the_per_draw_solver():
    """ransac_on_matches solves all draws as one stack; every transform must equal solver_point_to_point's bit for
    bit (plain, mirrored = every draw a reflection, noisy, planar = rank-deficient covariances)."""
    from shot_fpfh_amd.core import solver_point_to_point
    from shot_fpfh_amd.core.geometry import solver_point_to_point_batched

    rng = np.random.default_rng(7)
    for mirror, noise, planar in ((False, 0.01, False), (True, 0.01, False), (False, 0.3, False), (False, 0.0, True)):
        a = rng.random((5000, 3))
        if planar:
            a[:, 2] = 0.5
        q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
        q *= np.sign(np.linalg.det(q))
        if mirror:
            q[:, 0] *= -1
        b = a @ q.T + 0.3 + noise * rng.standard_normal((5000, 3))
        for k in (3, 4, 7):
            draws = np.array([rng.choice(5000, k, replace=False, shuffle=False) for _ in range(1500)])
            rot, tr = solver_point_to_point_batched(a[draws], b[draws])
            ref = [solver_point_to_point(a[d], b[d]) for d in draws]
            assert np.array_equal(rot, np.array([t.rotation for t in ref]))
            assert np.array_equal(tr, np.array([t.translation for t in ref]))


def test_pinned_output_pool_reuses_blocks_after_garbage_collection():
    """Engine.host_empty's pool (engine._PinnedPool) with malloc standing in for hipHostMalloc: arrays are writable and
    C-contiguous, a block returns to the pool only when the array AND its views are gone, and is then reused."""
    import ctypes
    import gc

    from shot_fpfh_amd.engine import _PinnedPool

    libc = ctypes.CDLL(None)
    libc.malloc.restype, libc.malloc.argtypes = ctypes.c_void_p, [ctypes.c_size_t]
    libc.free.argtypes = [ctypes.c_void_p]

    class Lib:
        allocs, frees = [], []

        def sf_host_alloc(self, ctx, size):
            p = libc.malloc(size)
            self.allocs.append(p)
            return p

        def sf_host_free(self, ctx, p):
            self.frees.append(p)
            libc.free(p)
            return 0

    lib = Lib()
    pool = _PinnedPool(lib, ctx=object())
    a = pool.array((1000, 352), np.dtype(np.float64), 352000, 352000 * 8)
    assert a.flags.c_contiguous and a.flags.writeable and a.shape == (1000, 352)
    a[:] = 3.0
    view = a[10:20]
    del a
    gc.collect()
    assert not pool.free  # the view keeps the block alive
    assert view[0, 0] == 3.0
    del view
    gc.collect()
    assert len(pool.free) + len(pool.returned) == 1 and len(lib.allocs) == 1  # (handed back, sorted in by the next request)
    b = pool.array((999, 352), np.dtype(np.float64), 999 * 352, 999 * 352 * 8)  # fits the cached block
    assert len(lib.allocs) == 1 and not pool.free
    c = pool.array((10, 352), np.dtype(np.float64), 3520, 3520 * 8)  # far smaller: its own block
    assert len(lib.allocs) == 2
    pool.drain()
    del b, c
    gc.collect()
    assert sorted(lib.frees) == sorted(lib.allocs)  # after drain(), returning blocks are unpinned at once


# ---- bench.py's control plane (sockets + pickle; no torch) ---------------------------------------------------------------
_CTL_WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import bench
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
c = bench.Control(rank, world, timeout=60)
got = c.allgather({"rank": rank, "blob": bytes([rank]) * (300000 + rank)})
assert [g["rank"] for g in got] == list(range(world)) and all(len(g["blob"]) == 300000 + r for r, g in enumerate(got))
assert c.max(float(rank)) == world - 1
assert c.bcast("id" if rank == 0 else None) == "id"
c.barrier()
c.close()
"""


@pytest.mark.parametrize("under_launcher", [False, True])
def test_bench_control_plane_rendezvous_allgather_max_bcast(tmp_path, under_launcher):
    """Three ranks over bench.Control.  under_launcher: MASTER_PORT is taken by somebody else's listener (torchrun's store
    in real life) that never sends the greeting -- the ranks must meet on a neighbouring port."""
    import os
    import socket
    import subprocess
    import sys

    from conftest import ROOT

    blocker = socket.socket()
    blocker.bind(("127.0.0.1", 0))
    port = blocker.getsockname()[1]
    if under_launcher:
        blocker.listen(8)
    else:
        blocker.close()
    script = tmp_path / "w.py"
    script.write_text(_CTL_WORKER)
    procs = []
    for rank in range(3):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="3", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        if not under_launcher:
            env["SF_BENCH_SELF_SPAWNED"] = "1"
        else:
            env.pop("SF_BENCH_SELF_SPAWNED", None)
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env))
    try:
        for p in procs:
            assert p.wait(timeout=120) == 0
    finally:
        blocker.close()


def test_bench_control_plane_survives_a_stale_token_file(tmp_path):
    """Advisor (round 5): a run that crashed leaves its token file behind (right owner, mode 0600, 32 bytes); ranks started
    before rank 0 has replaced it read the OLD token.  Rank 0 drops their hello, they read the file again and retry: the job
    meets.  Here ranks 1 and 2 start two seconds before rank 0, with the stale file in place and no SF_BENCH_TOKEN."""
    import os
    import socket
    import subprocess
    import sys
    import tempfile
    import time

    from conftest import ROOT

    s_ = socket.socket()
    s_.bind(("127.0.0.1", 0))
    port = s_.getsockname()[1]
    s_.close()
    path = os.path.join(tempfile.gettempdir(), f"sfbench-{os.getuid()}-{port}.token")
    fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o600)
    os.write(fd, b"0" * 32)  # (what a crashed run left)
    os.close(fd)
    script = tmp_path / "w.py"
    script.write_text(_CTL_WORKER)
    procs = {}
    try:
        for rank in (1, 2, 0):
            env = dict(os.environ, RANK=str(rank), WORLD_SIZE="3", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SF_BENCH_SELF_SPAWNED="1")
            env.pop("SF_BENCH_TOKEN", None)
            procs[rank] = subprocess.Popen([sys.executable, str(script), ROOT], env=env)
            if rank == 2:
                time.sleep(2.0)
        for p in procs.values():
            assert p.wait(timeout=120) == 0
    finally:
        for p in procs.values():
            if p.poll() is None:
                p.kill()
        if os.path.exists(path):
            os.unlink(path)


def test_bench_control_plane_drops_a_client_without_the_job_secret(tmp_path):
    """No SF_BENCH_TOKEN and no launcher secret: rank 0 writes 16 random bytes to a file only this user can read and the other
    ranks read it back; a local process that connects to rank 0's port and sends a well-formed message under a guessed key (the
    public run id "none", the port, the world size) is dropped unread, and the job's own ranks still meet."""
    import hashlib
    import hmac
    import os
    import pickle
    import socket
    import struct
    import subprocess
    import sys
    import time

    from conftest import ROOT

    s0 = socket.socket()
    s0.bind(("127.0.0.1", 0))
    port = s0.getsockname()[1]
    s0.close()
    script = tmp_path / "w.py"
    script.write_text(_CTL_WORKER)
    env = dict(os.environ, WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SF_BENCH_SELF_SPAWNED="1", TORCHELASTIC_RUN_ID="none")
    env.pop("SF_BENCH_TOKEN", None)
    r0 = subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK="0"))
    rogue = None
    for _ in range(200):  # rank 0 listens on MASTER_PORT itself when the bench spawned its ranks
        try:
            rogue = socket.create_connection(("127.0.0.1", port), timeout=2.0)
            break
        except OSError:
            time.sleep(0.05)
    assert rogue is not None
    rogue.settimeout(10.0)
    assert rogue.recv(64).startswith(b"SFBENCH")
    data = pickle.dumps(1)
    guessed = ("sfbench|none|" + str(port) + "|2").encode()
    rogue.sendall(struct.pack("<Q", len(data)) + hmac.new(guessed, data, hashlib.sha256).digest() + data)
    assert rogue.recv(16) == b""  # dropped: rank 0 closed the connection without unpickling
    rogue.close()
    r1 = subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK="1"))
    assert r0.wait(timeout=120) == 0 and r1.wait(timeout=120) == 0
    import tempfile

    assert not os.path.exists(os.path.join(tempfile.gettempdir(), f"sfbench-{os.getuid()}-{port}.token"))  # (removed at close)


def test_pinned_pool_finalizer_takes_no_lock_and_trim_releases():
    """engine._PinnedPool: a block returns through a weakref finalizer that may run at any allocation point, also inside
    array() while the pool's lock is held -- so the finalizer only appends to a deque.  Views keep a block alive; a
    returned block is re-used for the next fitting request; trim() unpins what is cached."""
    import ctypes
    import gc

    from shot_fpfh_amd.engine import _PinnedPool

    class FakeLib:
        def __init__(self):
            self.live = {}

        def sf_host_alloc(self, ctx, size):
            b = ctypes.create_string_buffer(size)
            self.live[ctypes.addressof(b)] = b
            return ctypes.addressof(b)

        def sf_host_free(self, ctx, addr):
            del self.live[addr]

    lib = FakeLib()
    pool = _PinnedPool(lib, 1)
    a = pool.array((1000,), np.float64, 1000, 8000)
    view = a[10:20]
    del a
    gc.collect()
    assert not pool.returned and len(lib.live) == 1  # the view keeps the block
    del view
    gc.collect()
    assert len(pool.returned) == 1
    with pool.lock:  # a finalizer firing while the lock is held must not block
        c = pool.array.__self__.returned  # (the deque is what finalizers touch)
        c.append(c.popleft())
    b = pool.array((900,), np.float64, 900, 7200)  # fits the cached block
    assert len(lib.live) == 1 and pool.cached == 0
    del b
    gc.collect()
    assert pool.trim() == 8192 and not lib.live


# ---- bench.py's record: one short line for the driver, everything else in a side file ------------------------------------------
def test_bench_line_is_compact_and_keeps_the_contract_keys(tmp_path, monkeypatch):
    """Round 4's single 23.6 KB JSON line came back from the driver unparsed.  The line is now a summary of < 4 KB with the
    contract's keys, `roofline` and `cpu_baseline`; the full record of a round-4 run (the largest shape there is) goes to the
    side file."""
    import importlib.util
    import json
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    full = json.load(open(os.path.join(root, "profiles", "r04_bench.json")))
    assert len(json.dumps(full)) > 20000
    monkeypatch.setenv("SF_BENCH_DETAIL_DIR", str(tmp_path))
    r, w = os.pipe()
    bench.emit_record(w, full)
    os.close(w)
    text = os.read(r, 1 << 16).decode()
    os.close(r)
    assert text.endswith("\n") and text.count("\n") == 1 and len(text) < bench.LINE_LIMIT
    line = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "sustained", "parity", "rccl_ranks", "per_rank_ms_per_step", "detail"):
        assert k in line, k
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(line["roofline"])
    assert {"value", "unit", "cores", "kind", "sample"} <= set(line["cpu_baseline"])
    assert "workload" in line["config"] and "model" not in line["config"]
    assert abs(line["value"] - full["value"]) <= 1e-5 * full["value"]
    assert json.load(open(tmp_path / "bench_detail.json")) == full
    # the worst case the contract allows for -- 8 ranks, strong-scaling and exchange blocks present -- still fits
    full.update(n_gpus=8, per_rank_ms_per_step=[3.123456] * 8, exchange_ms={"c_exchange": 0.123456, "c_allgather": 1.234567},
                strong_scaling={"ms_per_step": 0.5, "value": 4e9, "parity_ok": True, "checksum": {"x": 1}}, end_to_end_config5_ms=12.5)
    assert len(json.dumps(bench.compact_record(full, "bench_detail.json"), separators=(",", ":"))) < bench.LINE_LIMIT
