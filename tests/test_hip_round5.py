"""GPU parity tests added in round 5 (all through the C ABI).

  * K7 on the matrix cores for lists of MORE than 255 points (k_fpfh_mcl): against the oracle, sparse-block form == full form bit for bit, consistent normals (high bytes really used);
  * K2's lists leave through an LDS ring (whole 256-byte runs): neighbour sets bit-exact at slot sizes around the ring's size.
"""
import os

import numpy as np
import pytest

from conftest import family, run_program, synth_cloud

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import shot_fpfh_amd as s

    return s.default_engine()


@pytest.fixture(scope="module")
def O():
    from oracle import oracle

    return oracle


def dense_cloud(n, seed, consistent_normals=False):
    p, nr, _ = synth_cloud(n, seed)
    if consistent_normals:  # smooth surface: nearly all pairs of a point fall into a few bins -> counts above 255
        rng = np.random.default_rng(seed)
        p[:, 2] = (0.5 + 0.01 * rng.standard_normal(n)).astype(np.float32)
        nr = np.tile(np.array([[0.0, 0.0, 1.0]]), (n, 1)) + 0.02 * rng.standard_normal((n, 3))
        nr /= np.linalg.norm(nr, axis=1)[:, None]
    return p, nr


@pytest.mark.parametrize("consistent", [False, True])
@pytest.mark.parametrize("n_bins", [5, 4, 3])
def test_k7_matrix_core_form_for_long_lists(eng, O, monkeypatch, n_bins, consistent):
    """Lists of 300 .. 1500 points (one, two and three super-chunks of the long form; a list of exactly 256, 512 and 513 is
    looked for among the keypoints): rows against the oracle.  (The vector-ALU form with exact sums this form replaced was its
    cross-check for a round -- 1e-12 apart -- and was removed in round 6.)"""
    import shot_fpfh_amd as s

    p, nr = dense_cloud(30000, 17 + n_bins, consistent)
    r = 0.2 if not consistent else 0.09
    cloud = eng.cloud(p)
    nb = cloud.radius_search_self(r)
    cnt = nb.counts()[np.argsort(cloud.perm())]
    nb.free()
    cloud.free()
    assert cnt.max() > 600 and (cnt > 255).sum() > 1000, (cnt.max(), (cnt > 255).sum())
    order = np.argsort(cnt)
    picks = [order[-60:], order[:10]]
    for edge in (255, 256, 257, 511, 512, 513, 1023, 1024, 1025):
        lo = np.searchsorted(cnt[order], edge - 2)
        picks.append(order[lo:lo + 12])
    kp = np.unique(np.concatenate(picks + [np.random.default_rng(3).choice(p.shape[0], 200, replace=False)]))
    got = s.compute_fpfh_descriptor(kp, p, nr, r, n_bins, verbose=False)
    sub = kp[:: max(1, kp.size // 120)]
    rows = np.searchsorted(kp, sub)
    want = O.compute_fpfh_descriptor(sub, p, nr, r, n_bins)
    assert np.abs(got[rows] - want).max() < 1e-9, np.abs(got[rows] - want).max()
    # all points keypoints (the launch over the selection of long lists) gives the same rows as keypoints by index
    full = s.compute_fpfh_descriptor(np.arange(p.shape[0]), p, nr, r, n_bins, verbose=False)
    assert np.array_equal(full[kp], got)
    # ... and the full form (every block live) the same bits as the sparse-block form
    monkeypatch.setenv("SF_FPFH_DENSE", "1")
    dense = s.compute_fpfh_descriptor(kp, p, nr, r, n_bins, verbose=False)
    monkeypatch.delenv("SF_FPFH_DENSE", raising=False)
    assert np.array_equal(dense, got)


@pytest.mark.parametrize("cap", [64, 96, 256, 288])
def test_k2_lists_through_the_lds_ring(eng, O, monkeypatch, cap):
    """The single sweep stores a list 64 positions at a time from a 256-entry ring: slot sizes below, at and above the ring's
    size, lists that overflow their slot (re-done exactly) and lists that end anywhere inside a 64-block."""
    p, _, _ = synth_cloud(40000, 23)
    monkeypatch.setenv("SF_K2_CAP", str(cap))
    cloud = eng.cloud(p)
    try:
        for r in (0.05, 0.09, 0.125):  # ~ 20, 120 and 320 neighbours per ball
            q = p[:: 2]  # (20 000 queries: the single sweep into slots, not the exact two-pass scheme of small query sets)
            off, idx = cloud.radius_search(q, r).export()
            sub = np.arange(0, q.shape[0], 50)
            oo, oi = O.radius_search(p, q[sub], r)
            assert np.array_equal(np.diff(off)[sub], np.diff(oo)), (r, cap)
            for t, row in enumerate(sub):
                assert np.array_equal(idx[off[row]:off[row + 1]], oi[oo[t]:oo[t + 1]]), (r, cap, row)
    finally:
        cloud.free()


# ---- k-NN on K2's mapping (k_knn4) ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("k", [1, 8, 30, 64])
@pytest.mark.parametrize("kind", ["uniform", "blob", "duplicates", "surface"])
def test_knn_on_the_radius_search_mapping(eng, O, monkeypatch, kind, k):
    """k <= 64: four queries per wave, the points within R ranked in ONE pass (k_knn4), against brute force; normals computed from
    the lists against the oracle's.  Dense blob: queries with more than 256 points within R go to the one-wave-per-query kernel
    at the same R; duplicated points: exact distance ties, broken by position; queries far outside the cloud: radius doublings."""
    from conftest import config1_cloud
    from shot_fpfh_amd.descriptors import compute_normals

    n, m = 30000, 4000
    p, _, rng = synth_cloud(n, 61 + k)
    if kind == "blob":
        p[: n // 2] = 0.5 + 0.02 * (p[: n // 2] - 0.5)
    elif kind == "duplicates":
        p[1::3] = p[0:-1:3][: p[1::3].shape[0]]  # every third point sits on its predecessor
    elif kind == "surface":
        p, _ = config1_cloud(n, 61 + k)
    q = np.vstack([p[rng.choice(n, m - 200, replace=False)], rng.random((200, 3)) * 3.0 - 1.0])
    cloud = eng.cloud(p)
    try:
        nb = cloud.knn_search(q, k)
        off, idx = nb.export()
    finally:
        cloud.free()
    assert np.array_equal(off, np.arange(m + 1) * k)
    sub = np.arange(0, m, 8)
    off_o, idx_o = O.knn_lists(p, q[sub], k)
    got, want = idx.reshape(m, k)[sub], np.sort(idx_o.reshape(sub.size, k), axis=1)
    for i in np.flatnonzero(~(got == want).all(axis=1)):  # rows may differ only where the k-th and (k+1)-th are equidistant
        d = np.sort(((p - q[sub[i]]) ** 2).sum(axis=1))
        assert d[k - 1] == d[k], f"query {sub[i]}: different neighbour set without a distance tie"
    if k >= 8 and kind in ("uniform", "surface"):  # normals from these lists against the oracle's (k-NN branch of compute_normals)
        pre = np.tile(np.array([[0.3, -0.5, 0.8]]), (400, 1))
        a = compute_normals(q[:400], p, k=k, pre_computed_normals=pre)
        b = O.compute_normals(q[:400], p, k=k, pre_computed_normals=pre)
        assert np.abs(a - b).max() < 1e-9


# ---- 3-D matching ("minimum over scales") through the matrix-core matcher ---------------------------------------------------------
@pytest.mark.parametrize("case", ["normalised", "empty_rows", "far_apart"])
def test_multiscale_matching_through_the_prefilter_equals_the_exact_kernel(eng, O, monkeypatch, case):
    """match_descriptors on (n_scales, n, 352) input: per scale the FP16 pre-filter + exact float64 re-rank, folded over the
    scales, against the exact tile kernel (SF_MATCH_EXACT=1) and the NumPy restatement of matching.py:77-136.  Empty descriptors
    at one scale or at all of them (a pair with an empty side counts as 1000); descriptors farther apart than 1000 (the fold
    hands the whole problem back to the exact kernel)."""
    from shot_fpfh_amd.matching import match_descriptors

    rng = np.random.default_rng(77)
    n1, n2, d, ns = 2600, 2300, 352, 3
    a = rng.random((ns, n1, d)) * (rng.random((ns, n1, d)) < 0.3)
    perm = rng.permutation(n1)[:n2]
    b = a[:, perm] + 0.01 * rng.standard_normal((ns, n2, d)) * (a[:, perm] != 0)
    if case != "far_apart":
        a /= np.maximum(np.linalg.norm(a, axis=2, keepdims=True), 1e-300)
        b /= np.maximum(np.linalg.norm(b, axis=2, keepdims=True), 1e-300)
    else:
        a[:, :40] *= 4000.0  # these rows are more than 1000 away from every reference row
    if case != "normalised":
        a[0, 5:60] = 0.0      # empty at one scale
        a[:, 100:130] = 0.0   # empty at every scale
        b[1, 10:200] = 0.0
        b[:, 300:320] = 0.0
    monkeypatch.delenv("SF_MATCH_EXACT", raising=False)
    got = match_descriptors(a, b, verbose=False, engine=eng)
    monkeypatch.setenv("SF_MATCH_EXACT", "1")
    exact = match_descriptors(a, b, verbose=False, engine=eng)
    monkeypatch.delenv("SF_MATCH_EXACT", raising=False)
    assert np.array_equal(got[0], exact[0]) and np.array_equal(got[1], exact[1])
    want = O.match_descriptors_multiscale(a, b)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    if case == "normalised":
        assert (perm[got[1]] == got[0]).mean() > 0.5  # (and it does find the partners)


# ---- FPFH with 6, 7, 8 bins on the byte table (a window of the bins) and the matrix cores ---------------------------------------
@pytest.mark.parametrize("n_bins", [6, 7, 8, 9, 11])
def test_fpfh_six_to_eight_bins_keep_a_window_of_the_bins_in_the_byte_table(eng, O, monkeypatch, n_bins):
    """(... and 9 and 11 bins: 81 of 729, 121 of 1331 -- the odd counts whose central alpha bin fits the 128-column row.)
    radius * max|n|^2 inside the central alpha bin(s): the table keeps 72 / 49 / 128 of the 216 / 343 / 512 bins as bytes
    (sf_spfh_create_for_radius) and K7 runs on the matrix cores.  Against the oracle (SPFH bit-exact), against the 16-bit table
    + vector K7 these bin counts took until round 5 (SF_FPFH_NO_WINDOW=1), sparse-block form == full form, lists above 255
    points (high bytes, the long-list form), keypoints by index and all points, and the sharded job in two blocks."""
    import shot_fpfh_amd as s
    from shot_fpfh_amd.engine import Spfh
    from shot_fpfh_amd.sharding import DescriptorJob

    p, nr, rng = synth_cloud(20000, 40 + n_bins)
    cloud = eng.cloud(p, nr)
    try:
        wide = 2 if n_bins <= 8 else 4  # (without a window: 16-bit counts up to 8 bins, the generic kernels' 32-bit beyond)
        assert Spfh(cloud, n_bins, 100, 0.05).elem_bytes == 1 and Spfh(cloud, n_bins, 100).elem_bytes == wide
        assert Spfh(cloud, n_bins, 100, 0.9).elem_bytes == wide  # (a radius that reaches other alpha bins: no window)
    finally:
        cloud.free()
    for r, nkp in ((0.05, 1500), (0.17, 400)):  # ~10 and ~400 neighbours per ball (lists above 255 points at the second radius)
        kp = np.sort(rng.choice(p.shape[0], nkp, replace=False))
        monkeypatch.delenv("SF_FPFH_NO_WINDOW", raising=False)
        f, spfh = s.compute_fpfh_descriptor(kp, p, nr, r, n_bins, verbose=False, return_spfh=True)
        monkeypatch.setenv("SF_FPFH_NO_WINDOW", "1")
        g = s.compute_fpfh_descriptor(kp, p, nr, r, n_bins, verbose=False)
        monkeypatch.delenv("SF_FPFH_NO_WINDOW", raising=False)
        assert np.abs(f - g).max() <= 1e-12 * max(1.0, np.abs(g).max())
        sub = kp[:: max(1, nkp // 100)]
        fo, spfh_o = O.compute_fpfh_descriptor(sub, p, nr, r, n_bins, return_spfh=True)
        assert np.array_equal(spfh, spfh_o)
        assert np.abs(f[np.searchsorted(kp, sub)] - fo).max() < 1e-9
        full = s.compute_fpfh_descriptor(np.arange(p.shape[0]), p, nr, r, n_bins, verbose=False)
        assert np.array_equal(full[kp], f)
        monkeypatch.setenv("SF_FPFH_DENSE", "1")
        dense = s.compute_fpfh_descriptor(kp, p, nr, r, n_bins, verbose=False)
        monkeypatch.delenv("SF_FPFH_DENSE", raising=False)
        assert np.array_equal(dense, f)
    # two blocks of a sharded job (neighbour exchange of table rows) == the unsharded rows, bit for bit
    q, qn, _ = synth_cloud(6000, 50 + n_bins)
    one = s.compute_fpfh_descriptor(np.arange(6000), q, qn, 0.09, n_bins, verbose=False)
    got = np.full((6000, n_bins**3), np.nan)
    for rank in range(2):
        job = DescriptorJob(eng, q, qn, 0.09, n_bins=n_bins, min_neighborhood_size=5, world=2, rank=rank, spfh_exchange="halo")
        job.step()
        assert job.spfh.elem_bytes == 1
        got[job.block_original_indices()] = job.fpfh_out.to_host()
        job.close()
    assert np.array_equal(got, one)


# ---- a self search of a range that was searched before: planned from its record, no read-back --------------------------------------
@pytest.mark.parametrize("kind", ["uniform", "clustered"])
def test_repeated_self_search_is_planned_from_its_record(eng, O, monkeypatch, kind):
    """The lists of a self search are a function of (cloud, radius, range): the second search of a range launches its sweep and
    every selection from the record of the first -- no statistics kernel, no read-back -- and leaves the same lists; a block of
    the cloud has a record of its own; SF_K2_CHECK_RECORD=1 re-counts and compares; descriptors of a repeated DescriptorJob
    step are bit-identical to the first step's.  Clustered cloud: lists that overflow their slot and lists above 255 points."""
    from shot_fpfh_amd.sharding import DescriptorJob

    if kind == "uniform":
        p, nr, _ = synth_cloud(40000, 71)
        r = 0.06
    else:
        p, nr, _, _ = family("clustered", 40000, np.random.default_rng(71))
        r = 0.05
    cloud = eng.cloud(p, nr)
    try:
        cloud.build_grid(r)

        def search(b=0, e=None):
            eng.sync(); eng.profile_reset(); eng.profile(True)
            nb = cloud.radius_search_self(r, b, e)
            eng.sync(); eng.profile(False)
            names = {k for k, v in eng.profile_report().items() if v[0]}
            out = (nb.total, nb.max_count, nb.export())
            nb.free()
            return out, names

        first, n1 = search()
        again, n2 = search()
        assert "k2_reduce" in n1 and "k2_reduce" not in n2 and "k2_sample" not in n2
        assert first[0] == again[0] and first[1] == again[1]
        assert np.array_equal(first[2][0], again[2][0]) and np.array_equal(first[2][1], again[2][1])
        blk, nb1 = search(5000, 30000)  # (at least 16 384 queries: the single sweep into slots, which is what keeps a record)
        blk2, nb2 = search(5000, 30000)
        assert "k2_reduce" in nb1 and "k2_reduce" not in nb2 and np.array_equal(blk[2][1], blk2[2][1]) and blk[0] == blk2[0]
        monkeypatch.setenv("SF_K2_CHECK_RECORD", "1")
        chk, n3 = search()
        monkeypatch.delenv("SF_K2_CHECK_RECORD", raising=False)
        assert "k2_reduce" in n3 and chk[0] == first[0]
    finally:
        cloud.free()
    job = DescriptorJob(eng, p, nr, r, n_bins=5, normalize=True, min_neighborhood_size=5)
    try:
        job.step()
        f1, s1 = job.fpfh_out.to_host(), job.shot_out.to_host()
        eng.sync(); eng.profile_reset(); eng.profile(True)
        job.step()
        eng.sync(); eng.profile(False)
        assert "k2_reduce" not in {k for k, v in eng.profile_report().items() if v[0]}
        assert np.array_equal(job.fpfh_out.to_host(), f1) and np.array_equal(job.shot_out.to_host(), s1)
        rows = job.block_original_indices()
        pick = np.arange(0, p.shape[0], 400)
        fo = O.compute_fpfh_descriptor(rows[pick], p, nr, r, 5)
        assert np.abs(f1[pick] - fo).max() < 1e-9
    finally:
        job.close()


# ---- a repeated step as one launch (HIP graph) ----------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", ["whole", "rank3_of_8", "clustered"])
def test_step_replayed_as_a_hip_graph_gives_the_same_rows(eng, mode):
    """DescriptorJob.step_replay(): two eager steps, the third captured (Engine.capture: sf_graph_begin / sf_graph_end), later
    calls replay the graph with one launch -- rows bit-identical to step()'s, also for one rank's share of an 8-rank job (block
    build, halo rows of the emulated peers) and on a clustered cloud (second launches over selections, lists above 255 points);
    buffers written between replays are overwritten again; close() gives the graph's blocks back."""
    from shot_fpfh_amd.sharding import DescriptorJob

    if mode == "clustered":
        p, nr, _, _ = family("clustered", 60000, np.random.default_rng(81))
        kw, r = {}, 0.04
    else:
        p, nr, _ = synth_cloud(60000, 81)
        kw, r = ({} if mode == "whole" else dict(world=8, rank=3, emulate_peers=True)), 0.05
    job = DescriptorJob(eng, p, nr, r, n_bins=5, normalize=True, min_neighborhood_size=5, **kw)
    try:
        job.step()
        f0, s0 = job.fpfh_out.to_host(), job.shot_out.to_host()
        for i in range(5):
            if i == 3:  # scribble over the outputs: the next replay must write every row again
                job.fpfh_out.from_host(np.full(job.fpfh_out.shape, np.nan))
                job.shot_out.from_host(np.full(job.shot_out.shape, np.nan))
            job.step_replay()
        eng.sync()
        assert getattr(job, "_graph_failed", None) is None and getattr(job, "_graph", None) is not None
        assert np.array_equal(job.fpfh_out.to_host(), f0) and np.array_equal(job.shot_out.to_host(), s0)
        job.step()  # (an eager step after replays: the pool still hands out sound blocks)
        eng.sync()
        assert np.array_equal(job.fpfh_out.to_host(), f0) and np.array_equal(job.shot_out.to_host(), s0)
    finally:
        job.close()


def _shot_launches(eng, fn):
    eng.profile_reset()
    eng.profile(True)
    try:
        out = fn()
    finally:
        eng.profile(False)
    return out, {k: v[0] for k, v in eng.profile_report().items() if v[0]}


@pytest.mark.parametrize("mean_k", [380, 800, 1500, 3300])
def test_k5_team_of_waves_for_long_lists(eng, O, monkeypatch, mean_k):
    """K5 for lists above 255 points: a team of waves per keypoint holds the list in registers (k_shot_team: three chunks per
    wave, as many waves as the list needs out of a workgroup of 4 / 8 / 16 -- lists up to 768 / 1 536 / 3 072 points), the
    streaming form takes what is longer.  Rows against the oracle at every boundary of the wave count, bit-identical from run
    to run (one writer per slot, fixed order of the five tables).  (The streaming form for EVERY long list was this form's
    cross-check for a round -- 1e-13 apart -- behind a switch that is gone; beyond 3 072 points it still serves.)"""
    from shot_fpfh_amd.descriptors import ShotMultiprocessor

    n = 14000
    p, nr, _ = synth_cloud(n, 60 + mean_k)
    p = p.astype(np.float64)
    ext = p.max(0) - p.min(0)
    r = float((mean_k * ext.prod() / n / (4.0 / 3.0 * np.pi)) ** (1.0 / 3.0)) * 1.08  # (+ the share of a ball outside the cloud)
    cloud = eng.cloud(p)
    nb = cloud.radius_search_self(r)
    cnt = nb.counts()[np.argsort(cloud.perm())]
    nb.free()
    cloud.free()
    assert cnt.max() > mean_k and cnt.max() > 256, cnt.max()
    order = np.argsort(cnt)
    picks = [order[-25:], order[:5], np.random.default_rng(mean_k).choice(n, 40, replace=False)]
    for edge in (255, 256, 257, 320, 384, 385, 576, 577, 767, 768, 769, 960, 1152, 1153, 1535, 1536, 1537, 2048, 3071, 3072, 3073):
        lo = np.searchsorted(cnt[order], edge)
        picks.append(order[max(lo - 2, 0):lo + 3])
    kp = np.unique(np.concatenate(picks))
    rows = {}
    for normalize, min_nb in ((True, 100), (False, 300)):
        with ShotMultiprocessor(min_neighborhood_size=min_nb, normalize=normalize, verbose=False) as sm:
            # every point a keypoint: the search plans its launches, the long lists are a selection of their own
            full, rep = _shot_launches(eng, lambda: sm.compute_descriptor_single_scale(p, nr, p, r))
            again = sm.compute_descriptor_single_scale(p, nr, p, r)
        assert np.array_equal(full, again)
        assert rep.get("k5_shot_tail") == 1, rep
        assert ("k5_shot_tail_stream" in rep) == (cnt.max() > 3072), (rep, cnt.max())
        want = O.shot_single_scale(p, nr, p[kp], r, normalize, min_nb)
        assert np.abs(full[kp] - want).max() < 1e-9, (normalize, np.abs(full[kp] - want).max())
        assert (np.abs(want).sum(1) > 0).sum() > kp.size // 2  # (the gate lets the long lists through)
        rows[normalize] = full
    # frames computed first, descriptors from given frames (sf_shot_lrf + sf_shot: the team form without the fused sign votes)
    cloud = eng.cloud(p, nr)
    nb = cloud.radius_search(p[kp], r)
    lrf = nb.shot_lrf()
    unfused = nb.shot(lrf, True, 100)
    nb.free()
    cloud.free()
    assert np.abs(unfused - rows[True][kp]).max() < 1e-13, np.abs(unfused - rows[True][kp]).max()

@pytest.mark.parametrize("kind", ["uniform", "clustered"])
def test_point_order_leaves_every_row_unchanged_at_full_size(eng, kind):
    """BASELINE config 3 at full size, all 2 x 10^6 rows (the oracle checks 300 of each, test_hip_round2.py): the same cloud
    handed over with its points in another order.  The cell-sorted positions of the points of a cell, every neighbour list's
    order and which lane holds which neighbour all change; no descriptor may.  FPFH rows are sums of integers and of
    fixed-point weights: bit-identical (the clustered cloud's long neighbours add their high bytes in float64, in list order:
    1e-12 there).  SHOT rows add their (at most five) contributions per bin in list order, and the frame's moments are float64
    sums in list order: 1e-12.  The clustered cloud (the bench's: 13 % of its lists above 255 points) takes every second
    launch -- K5's team form, K7's long form, the high-byte rows."""
    from shot_fpfh_amd.sharding import DescriptorJob

    n = 1_000_000
    if kind == "uniform":
        r = 0.03
        p, nr, rng = synth_cloud(n, 3)
    else:
        r = 0.004454284480470688  # (tools/bench_density.py: the radius that gives this cloud the headline's 110 neighbours)
        rng = np.random.default_rng(3)
        p, nr, _, _ = family("clustered", n, rng)
    shuffle = rng.permutation(n)
    a = DescriptorJob(eng, p, nr, r, n_bins=5, normalize=True, min_neighborhood_size=10)
    b = DescriptorJob(eng, p[shuffle], nr[shuffle], r, n_bins=5, normalize=True, min_neighborhood_size=10)
    try:
        a.step()
        b.step()
        assert a.last_pairs == b.last_pairs
        where_a = np.empty(n, np.int64)  # row of job a's outputs that belongs to point i
        where_a[a.block_original_indices()] = np.arange(n)
        point_b = shuffle[b.block_original_indices()]  # point (numbering of the cloud as given) of job b's row j
        fa = a.fpfh_out.to_host()
        worst_f = worst_s = 0.0
        nonzero = 0
        for j0 in range(0, n, 125_000):
            rows = where_a[point_b[j0:j0 + 125_000]]
            fb = b.fpfh_out.rows_to_host(j0, min(125_000, n - j0))
            worst_f = max(worst_f, float(np.abs(fb - fa[rows]).max()))
        del fa
        sa = a.shot_out.to_host()
        for j0 in range(0, n, 125_000):
            rows = where_a[point_b[j0:j0 + 125_000]]
            sb = b.shot_out.rows_to_host(j0, min(125_000, n - j0))
            worst_s = max(worst_s, float(np.abs(sb - sa[rows]).max()))
            nonzero += int(np.any(sb, axis=1).sum())
        assert worst_f == 0.0 if kind == "uniform" else worst_f < 1e-12, worst_f
        assert worst_s < 1e-12, worst_s
        assert nonzero > (0.99 if kind == "uniform" else 0.5) * n
    finally:
        a.close()
        b.close()


def _rows_by_point(job, arr, n):
    where = np.empty(n, np.int64)
    where[job.block_original_indices()] = np.arange(n)
    return arr.to_host()[where]


@pytest.mark.parametrize("with_fpfh", [True, False])
def test_a_power_of_two_scale_changes_no_bit_of_a_shot_row_at_full_size(eng, with_fpfh):
    """BASELINE config 3 at full size, every row: coordinates and radius multiplied by 4 (exact in floating point).  Every
    decision of SHOT is a comparison of quantities that scale together -- distances against the radius and its half, the
    frame's votes, the bins from ratios -- and every weight is a ratio, so all 10^6 frames and rows must come out bit for bit:
    no absolute threshold anywhere on the path.  (FPFH is not scale-invariant in the reference either: its weights are 1 / d
    and alpha = ((p_j - p_i) x n_i) . n_j is not normalised, fpfh.py:60.)"""
    from shot_fpfh_amd.sharding import DescriptorJob

    n, r = 1_000_000, 0.03
    p, nr, _ = synth_cloud(n, 3)
    out = []
    for scale in (1.0, 4.0):
        # (with FPFH the frames' moments come out of K6's sweep and K5 is fused; without, k_shot_lrf forms them)
        job = DescriptorJob(eng, scale * p, nr, scale * r, n_bins=5, normalize=True, min_neighborhood_size=10, do_fpfh=with_fpfh)
        job.step()
        out.append((_rows_by_point(job, job.lrf_out, n), _rows_by_point(job, job.shot_out, n), job.last_pairs))
        job.close()
    assert out[0][2] == out[1][2]
    assert np.array_equal(out[0][0], out[1][0])
    assert np.array_equal(out[0][1], out[1][1])
    assert np.any(out[0][1], axis=1).mean() > 0.99


def test_rigid_motion_of_the_cloud_at_full_size(eng):
    """BASELINE config 3 at full size, every row: the cloud and its normals rotated and translated.  FPFH is built from angles
    and distances: all 10^6 rows within 1e-9.  SHOT's frame is defined up to the signs its votes settle (shot.py:40-45) --
    a tied vote, 7 % of the lists of 110 points per axis, keeps the sign the eigen-solver returned, in the reference too --
    so: every frame equals the rotated frame up to the signs of its columns, and wherever the signs agree the rows agree."""
    from shot_fpfh_amd.sharding import DescriptorJob

    n, r = 1_000_000, 0.03
    p, nr, rng = synth_cloud(n, 3)
    q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    t = rng.standard_normal(3)
    out = []
    for pp, nn in ((p, nr), (p @ q.T + t, nr @ q.T)):
        job = DescriptorJob(eng, pp, nn, r, n_bins=5, normalize=True, min_neighborhood_size=10)
        job.step()
        out.append((_rows_by_point(job, job.fpfh_out, n), _rows_by_point(job, job.lrf_out, n).reshape(n, 3, 3),
                    _rows_by_point(job, job.shot_out, n), job.last_pairs))
        job.close()
    (fa, ea, sa, pa), (fb, eb, sb, pb) = out
    assert pa == pb  # (no pair at a distance within rounding of the radius)
    assert np.abs(fb - fa).max() < 1e-9, np.abs(fb - fa).max()
    # columns of a frame are its axes x, y, z: (Q E_a)^T E_b must be diag(+-1)
    c = np.einsum("nij,nik->njk", q @ ea, eb)
    diag = np.stack([c[:, 0, 0], c[:, 1, 1], c[:, 2, 2]], axis=1)
    assert (np.abs(np.abs(diag) - 1.0) < 1e-6).all(axis=1).mean() > 0.9999  # (the rest: two eigenvalues within rounding of each other)
    same = (diag > 0.999999).all(axis=1)
    assert same.mean() > 0.8, same.mean()
    err = np.abs(sb - sa).max(axis=1)
    # (a neighbour within rounding of a bin boundary may change bins under the rotation: a handful of rows in 10^6)
    assert (err[same] > 1e-9).sum() <= 20, ((err[same] > 1e-9).sum(), err[same].max())


def test_radius_search_is_symmetric_at_full_size(eng):
    """K2 at BASELINE config 3's size, all 110 M pairs: j is in the list of i exactly when i is in the list of j (the squared
    distance is the same float64 either way, so an asymmetric pair is a candidate a sweep missed or one it invented), every
    point is in its own list, lists are ascending and duplicate-free."""
    n, r = 1_000_000, 0.03
    p, _, _ = synth_cloud(n, 3)
    cloud = eng.cloud(p)
    nb = cloud.radius_search(p, r)
    off, idx = nb.export()
    nb.free()
    cloud.free()
    cnt = np.diff(off)
    assert off[-1] == idx.size and 100 < idx.size / n < 120
    rows = np.repeat(np.arange(n, dtype=np.int64), cnt)
    fwd = rows * n + idx
    assert (np.diff(fwd) > 0).all()  # ascending inside a list, no duplicates
    assert int((rows == idx).sum()) == n
    bwd = idx.astype(np.int64) * n + rows
    del rows, idx
    bwd.sort()
    assert np.array_equal(fwd, bwd)


def test_knn_lists_agree_with_the_radius_search_at_full_size(eng):
    """k-NN (k = 30, the CLI's default for normals) for all 10^6 points of config 3's cloud against K2, an independent kernel:
    the k nearest points hold every point within r0 when there are at most k of them, and only such points otherwise --
    #{entries with d <= r0} == min(#{points within r0}, k) for every query; the point itself is in its list."""
    n, k, r0 = 1_000_000, 30, 0.02
    p, _, _ = synth_cloud(n, 3)
    cloud = eng.cloud(p)
    nbk = cloud.knn_search(p, k)
    off, idx, dist = nbk.export(True)
    nbk.free()
    nb = cloud.radius_search(p, r0)
    within = nb.counts()
    nb.free()
    cloud.free()
    assert np.array_equal(np.diff(off), np.full(n, k))
    dist, idx = dist.reshape(n, k), idx.reshape(n, k)
    assert (np.diff(idx, axis=1) > 0).all()  # (the export's canonical form: ascending indices, no duplicates)
    assert ((idx == np.arange(n)[:, None]) & (dist == 0)).any(axis=1).all()  # the point itself, at distance 0
    assert 0.2 < (within < k).mean() < 0.8  # (both cases of the minimum are exercised)
    assert np.array_equal((dist <= r0).sum(axis=1), np.minimum(within, k))


def test_ransac_scores_at_full_size(eng, O):
    """K9 at BASELINE config 4's size -- 10^4 candidate transforms x 10^6 matches in one launch: the inlier counts of 32 draws
    spread over the launch (and of the best one) equal the oracle's, count for count; an infinite threshold counts every
    match for every draw, a negative one none."""
    from shot_fpfh_amd.core.geometry import solver_point_to_point_batched

    n, draws = 1_000_000, 10_000
    rng = np.random.default_rng(11)
    scan = rng.random((n, 3))
    q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    ref = scan @ q.T + np.array([0.3, -0.2, 0.1]) + 2e-3 * rng.standard_normal((n, 3))
    wrong = rng.random(n) < 0.3  # (30 % of the matches are wrong)
    ref[wrong] = rng.random((int(wrong.sum()), 3))
    pick = rng.integers(0, n, size=(draws, 4))
    rot, tr = solver_point_to_point_batched(scan[pick], ref[pick])
    rt = np.concatenate([rot.reshape(draws, 9), tr], axis=1)
    thr = 0.01
    got = eng.ransac_score(scan, ref, rt, thr)
    assert got.max() > 0.5 * n  # (a draw of four correct matches exists among 10^4)
    check = np.unique(np.concatenate([np.linspace(0, draws - 1, 32).astype(int), [int(np.argmax(got))]]))
    want = O.ransac_score(scan, ref, rt[check], thr)
    assert np.array_equal(got[check], want), (got[check], want)
    assert np.array_equal(eng.ransac_score(scan, ref, rt[check], np.inf), np.full(check.size, n))
    assert not eng.ransac_score(scan, ref, rt[check], -1.0).any()


def test_normals_under_scale_and_rigid_motion_at_full_size(eng):
    """compute_normals(radius) (K3, the fused sweep) for all 10^6 points of config 3's cloud: unit vectors; coordinates and radius
    x 4 change no bit; under a rigid motion every normal is the rotated normal up to its sign (the eigen-solver's, with no
    pre-computed normals to orient it) -- the direction within 1e-9 for all but the points whose two smallest eigenvalues meet."""
    n, r = 1_000_000, 0.03
    p, _, rng = synth_cloud(n, 3)
    q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    t = rng.standard_normal(3)
    res = []
    for pts, rad in ((p, r), (4.0 * p, 4.0 * r), (p @ q.T + t, r)):
        cloud = eng.cloud(pts)
        res.append(cloud.normals_radius(pts, rad))
        cloud.free()
    na, ns, nr = res
    assert np.abs(np.linalg.norm(na, axis=1) - 1.0).max() < 1e-12
    assert np.array_equal(na, ns)
    cosine = np.abs(np.einsum("ij,ij->i", nr, na @ q.T))
    assert (cosine < 1.0 - 1e-9).mean() < 1e-4, ((cosine < 1.0 - 1e-9).sum(), cosine.min())


def test_a_c_program_gets_the_rows_the_python_drop_ins_return(eng, tmp_path):
    """examples/c_abi_demo.c, compiled with gcc against include/shotfpfh.h: FPFH and SHOT rows of 1 500 keypoints of a 30 000-point
    cloud computed by a C caller equal the rows of compute_fpfh_descriptor / compute_descriptor_single_scale bit for bit -- the
    Python layer adds nothing to the boundary."""
    import shutil
    import subprocess

    import shot_fpfh_amd as s
    from shot_fpfh_amd.descriptors import ShotMultiprocessor

    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib_dir = os.path.join(root, "shot_fpfh_amd")
    exe = str(tmp_path / "c_abi_demo")
    r = subprocess.run(["gcc", "-std=c99", "-O1", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "c_abi_demo.c"),
                        "-L", lib_dir, "-lshotfpfh", "-lm", "-Wl,-rpath," + lib_dir, "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    n, radius = 30_000, 0.05
    p, nr, rng = synth_cloud(n, 77)
    kp = np.sort(rng.choice(n, 1_500, replace=False)).astype(np.int64)
    with open(tmp_path / "cloud.bin", "wb") as f:
        np.array([n, kp.size], np.int64).tofile(f)
        np.ascontiguousarray(p, np.float64).tofile(f)
        np.ascontiguousarray(nr, np.float64).tofile(f)
        kp.tofile(f)
    # (the helper process of conftest starts programs that use the GPU: this process already holds a context)
    res = run_program([exe, str(tmp_path / "cloud.bin"), str(tmp_path / "rows.bin"), repr(radius)], timeout=300)
    assert res["rc"] == 0, res
    rows = np.fromfile(tmp_path / "rows.bin", dtype=np.float64)
    fpfh_c, shot_c = rows[: kp.size * 125].reshape(kp.size, 125), rows[kp.size * 125:].reshape(kp.size, 352)
    fpfh_py = s.compute_fpfh_descriptor(kp, p, nr, radius, 5, verbose=False)
    with ShotMultiprocessor(normalize=True, min_neighborhood_size=10, verbose=False) as sm:
        shot_py = sm.compute_descriptor_single_scale(p, nr, p[kp], radius)
    assert np.array_equal(fpfh_c, fpfh_py)
    assert np.array_equal(shot_c, shot_py)
    assert np.any(shot_c, axis=1).mean() > 0.5 and np.any(fpfh_c, axis=1).all()  # (a corner point's list may fail SHOT's gate)


def test_reciprocity_filter_at_scale_keeps_exactly_the_true_pairs(eng):
    """match_descriptors(filter_nonreciprocal=True) on 200 000 x 150 000 x 352 (the FP16 pre-filter's size): the reference set is
    a noisy copy of 150 000 of the scan rows in another order.  A scan row whose copy is there must keep it; one whose copy is
    not finds some other row's copy, which is closer to its own original -- the column arg-min rejects it.  What is left is
    exactly the set of true pairs."""
    from shot_fpfh_amd.matching import match_descriptors

    rng = np.random.default_rng(505)
    m, mr, d = 200_000, 150_000, 352
    a = rng.random((m, d), dtype=np.float32).astype(np.float64)
    a *= rng.random((m, d), dtype=np.float32) < 0.3  # SHOT-like sparsity
    a /= np.maximum(np.linalg.norm(a, axis=1)[:, None], 1e-300)
    src = rng.permutation(m)[:mr]  # ref row j is a copy of scan row src[j]
    b = a[src] + 1e-6 * rng.standard_normal((mr, d)).astype(np.float32)
    si, ri = match_descriptors(a, b, filter_nonreciprocal=True, verbose=False, engine=eng)
    order = np.argsort(si)
    want_scan = np.sort(src)
    assert np.array_equal(si[order], want_scan)
    back = np.empty(m, np.int64)
    back[src] = np.arange(mr)
    assert np.array_equal(ri[order], back[want_scan])


def test_icp_recovers_a_planted_motion_on_a_large_scan(eng):
    """Point-to-point ICP at a size its golden does not reach: 150 000 points of a 600 000-point surface scan moved
    by a known small rigid motion.  Every scan point has its exact partner in the reference, so the planted motion is a fixed
    point of the iteration with zero residual: the inverse motion must come back.  (Point-to-plane on this cloud -- a sphere
    with radial normals -- cannot see a rotation about the centre; its golden is a plane-rich scene, test_hip_parity.py.)"""
    from conftest import config1_cloud
    from shot_fpfh_amd.core import RigidTransform
    from shot_fpfh_amd.icp import icp_point_to_point

    n, m = 600_000, 150_000
    ref, _ = config1_cloud(n, 21)
    rng = np.random.default_rng(22)
    axis = rng.standard_normal(3)
    axis /= np.linalg.norm(axis)
    ang = np.deg2rad(0.6)
    kx = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    rot = np.eye(3) + np.sin(ang) * kx + (1 - np.cos(ang)) * kx @ kx
    t = np.array([0.002, -0.001, 0.0015])
    scan = ref[rng.choice(n, m, replace=False)] @ rot.T + t
    tf, rms, ok = icp_point_to_point(scan, ref, RigidTransform(np.eye(3), np.zeros(3)), d_max=0.02, voxel_size=0.002, max_iter=80,
                                     rms_threshold=1e-7, disable_progress_bar=True)
    assert ok, rms
    assert np.abs(tf.rotation @ rot - np.eye(3)).max() < 1e-7, np.abs(tf.rotation @ rot - np.eye(3)).max()
    assert np.abs(tf.rotation @ t + tf.translation).max() < 1e-7
