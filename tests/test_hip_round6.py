"""GPU parity tests added in round 6 (all through the C ABI).

  * caller-supplied neighbourhoods (sf_nbrs_import): ShotMultiprocessor.compute_local_rf / compute_descriptor on KDTree.query lists,
    on lists of another radius and on lists of the radius itself, against rows the reference produced for exactly those lists;
  * float32 clouds against rows the REFERENCE computed from the float32 arrays (not against the build's own float64 call);
  * compute_metrics_post_icp against the reference's numbers;
  * the per-range search record across grid rebuilds (advisor, round 5) and the grid stamp on list sets.
"""
import os

import numpy as np
import pytest

from conftest import load_golden, synth_cloud

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import shot_fpfh_amd as s

    return s.default_engine()


@pytest.fixture(scope="module")
def O():
    from oracle import oracle

    return oracle


def _object_lists(off, idx):
    arr = np.empty(off.size - 1, dtype=object)
    for i in range(off.size - 1):
        arr[i] = idx[off[i]:off[i + 1]]
    return arr


# ---- caller-supplied neighbourhoods ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind", ["knn", "wide", "same"])
def test_shot_pieces_use_the_lists_they_are_handed(kind):
    """shot_parallelization.py:46-133: frames and descriptors over support[neighborhoods[i]] -- k-NN lists (40 points whatever the
    radius), lists of 1.5 x the radius (the points beyond it weigh negatively in the frame, as in the reference) and the radius's
    own lists.  Expected rows: the reference's, for these very lists (tools/gen_golden_r6.py)."""
    from shot_fpfh_amd.descriptors import ShotMultiprocessor

    g = load_golden("shot_lists.npz")
    off, idx, r = g[f"{kind}_offsets"], g[f"{kind}_idx"], float(g["radius"])
    lists = _object_lists(off, idx)
    with ShotMultiprocessor(normalize=True, min_neighborhood_size=10, verbose=False) as sm:
        lrf = sm.compute_local_rf(g["keypoints"], lists, g["cloud"], r)
        desc = sm.compute_descriptor(g["keypoints"], g["normals"], lists, g[f"{kind}_lrf"], g["cloud"], r)
        assert lrf.shape == (off.size - 1, 3, 3) and desc.shape == (off.size - 1, 352)
        assert np.abs(lrf - g[f"{kind}_lrf"]).max() < 1e-9
        assert np.abs(desc - g[f"{kind}_desc"]).max() < 1e-9
        if kind == "knn":  # KDTree.query's 2-D array, and lists in another order / with negative indices, name the same sets
            two_d = idx.reshape(off.size - 1, -1)
            assert np.array_equal(sm.compute_descriptor(g["keypoints"], g["normals"], two_d, g[f"{kind}_lrf"], g["cloud"], r), desc)
            n = g["cloud"].shape[0]
            flipped = _object_lists(off, idx)
            for i in range(flipped.size):
                flipped[i] = flipped[i][::-1] - (n if i % 2 else 0)
            d2 = sm.compute_descriptor(g["keypoints"], g["normals"], flipped, g[f"{kind}_lrf"], g["cloud"], r)
            assert np.abs(d2 - desc).max() < 1e-12
        if kind == "wide":  # the lists matter: the radius's own lists give other rows for most keypoints
            own = sm.compute_descriptor(g["keypoints"], g["normals"], None, g[f"{kind}_lrf"], g["cloud"], r)
            assert (np.abs(own - desc).max(axis=1) > 1e-3).sum() > 60


def test_imported_lists_refuse_what_numpy_indexing_refuses(eng):
    from shot_fpfh_amd._ffi import ShotFpfhError

    p, nr, rng = synth_cloud(500, 71)
    cloud = eng.cloud(p, nr)
    try:
        kp = p[:4]
        ok = [np.array([0, 1, 2]), np.array([], dtype=np.int64), np.array([499]), np.array([-500, 3])]
        nb = cloud.import_neighbors(kp, ok, 0.2)
        assert (nb.m, nb.total, nb.max_count) == (4, 6, 3)
        lrf = nb.shot_lrf()
        assert np.array_equal(lrf[1], np.eye(3))  # (an empty neighbourhood: the identity, shot.py:24-25)
        # the lists are bound to the grid they were imported on: a search with a much larger radius rebuilds it
        other = cloud.radius_search(kp, 0.9)
        with pytest.raises(ShotFpfhError, match="another grid"):
            nb.shot_lrf()
        other.free()
        nb.free()
        for bad in ([np.array([0]), np.array([500]), np.array([1]), np.array([2])],
                    [np.array([0]), np.array([-501]), np.array([1]), np.array([2])],
                    [np.array([0.0]), np.array([1]), np.array([1]), np.array([2])]):
            with pytest.raises(IndexError):
                cloud.import_neighbors(kp, bad, 0.2)
        with pytest.raises(ValueError):
            cloud.import_neighbors(kp, ok[:3], 0.2)
    finally:
        cloud.free()


# ---- float32 clouds -----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["unit", "offset"])
@pytest.mark.parametrize("ntag", ["n64", "n32"])
def test_float32_clouds_against_the_reference_run_on_float32_arrays(tag, ntag):
    """get_data hands the PLY's float32 columns on (io_ply.py:269) and the reference then subtracts and takes norms in float32
    (shot.py:211-214, fpfh.py:45-48).  The drop-ins cast to float64 at the boundary; held against the rows the reference itself
    computed from the float32 arrays -- a unit-cube cloud and the same cloud at (40, -25, 12), float64 and float32 normals.
    Tolerance (north_star): abs(a - b) <= 1e-5 (SHOT), <= 1e-5 max(1, abs(b)) (FPFH); the COUNT of rows outside it must be 0."""
    import shot_fpfh_amd as s
    from shot_fpfh_amd.descriptors import ShotMultiprocessor

    g = load_golden("shot_f32.npz")
    p = g[f"{tag}_cloud"]
    assert p.dtype == np.float32
    nrm = g["normals64"] if ntag == "n64" else g["normals64"].astype(np.float32)
    kp = p[g["keypoints_indices"]][g["rows"]]
    with ShotMultiprocessor(normalize=True, min_neighborhood_size=int(g["min_neighborhood_size"]), verbose=False) as sm:
        d = sm.compute_descriptor_single_scale(p, nrm, kp, float(g["radius"]))
    gap = np.abs(d - g[f"{tag}_{ntag}_desc"]).max(axis=1)
    print(f"SHOT {tag}/{ntag}: {int((gap > 1e-5).sum())} of {gap.size} rows outside 1e-5, largest gap {gap.max():.3g}")
    assert int((gap > 1e-5).sum()) == 0
    f = load_golden("fpfh_f32.npz")
    pf = f[f"{tag}_cloud"]
    nf = f["normals64"] if ntag == "n64" else f["normals64"].astype(np.float32)
    got = s.compute_fpfh_descriptor(f["keypoints_indices"], pf, nf, float(f["radius"]), int(f["n_bins"]), verbose=False)
    want = f[f"{tag}_{ntag}_desc"]
    bad = (np.abs(got - want) > 1e-5 * np.maximum(1.0, np.abs(want))).any(axis=1)
    print(f"FPFH {tag}/{ntag}: {int(bad.sum())} of {bad.size} rows outside tolerance, largest gap {np.abs(got - want).max():.3g}")
    assert int(bad.sum()) == 0


# ---- post-ICP metrics ---------------------------------------------------------------------------------------------------------
def test_metrics_post_icp_equal_the_references():
    """pipeline.py:544-587: share of aligned scan points with a ref point within the threshold, and the same for the keypoints."""
    from shot_fpfh_amd.core import RigidTransform
    from shot_fpfh_amd.pipeline import RegistrationPipeline

    g, icp = load_golden("metrics_post_icp.npz"), load_golden("icp_3500.npz")
    pipe = RegistrationPipeline(scan=icp["scan"], scan_normals=np.zeros_like(icp["scan"]), ref=icp["ref"], ref_normals=icp["ref_normals"])
    pipe.scan_keypoints, pipe.ref_keypoints = g["scan_keypoints"], g["ref_keypoints"]
    for i in range(int(g["n_cases"])):
        tf = RigidTransform(g[f"case{i}_rotation"], g[f"case{i}_translation"])
        overlap, ratio = pipe.compute_metrics_post_icp(tf, float(g[f"case{i}_threshold"]))
        assert (overlap, ratio) == tuple(g[f"case{i}_metrics"]), (i, overlap, ratio, g[f"case{i}_metrics"])


# ---- the search record and the grid it belongs to -----------------------------------------------------------------------------
def _self_lists(cloud, O, p, radius, begin, end):
    nb = cloud.radius_search_self(radius, begin, end)
    off, idx = nb.export()
    nb.free()
    perm = cloud.perm()
    q = p[perm[begin:end]]
    eo, ei = O.radius_search(p, q, radius)
    return (off, idx), (eo, ei.astype(np.int32))


def test_search_record_of_a_sub_range_does_not_outlive_its_grid(eng, O, monkeypatch):
    """Advisor (round 5, high): records were keyed by (radius, first position, count); ensure_grid serves radius r from any grid
    with cell in [r, 2r], and positions [b, e) name other points on another grid.  Sequence: grid of 0.05 -> self search of a
    sub-range at 0.03 (twice: the second from the record) -> k-NN search (rebuilds) -> the same sub-range at 0.03 on the new
    grid -> a 0.1 search -> again.  Every search's lists against the oracle's, in both the slot scheme and the exact scheme."""
    monkeypatch.delenv("SF_K2_CHECK_RECORD", raising=False)
    for n, b, e in ((60000, 7000, 41000), (3000, 500, 2500)):  # (slots: m >= 16 384; exact: below)
        p, nr, rng = synth_cloud(n, 81)
        r = 0.03 if n > 10000 else 0.08
        cloud = eng.cloud(p, nr)
        try:
            cloud.build_grid(r * 5 / 3)
            for step in range(7):
                got, want = _self_lists(cloud, O, p, r, b, e)
                assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), (n, step)
                if step == 1:
                    cloud.knn_search(p[:200], 12).free()
                elif step == 3:
                    cloud.radius_search(p[:50], 0.1 if n > 10000 else 0.3).free()
                elif step == 5:
                    cloud.build_grid(r * 1.9)
        finally:
            cloud.free()


def test_lists_of_a_rebuilt_grid_are_refused_not_misread(eng):
    from shot_fpfh_amd._ffi import ShotFpfhError

    p, nr, rng = synth_cloud(20000, 82)
    cloud = eng.cloud(p, nr)
    try:
        nb = cloud.radius_search_self(0.04)
        first = nb.shot_single_scale(True, 10)
        cloud.radius_search(p[:10], 0.03).free()  # (the 0.04 grid serves 0.03 too: no rebuild, the lists stay valid)
        assert np.array_equal(nb.shot_single_scale(True, 10), first)
        cloud.radius_search(p[:10], 0.2).free()  # another grid
        for call in (lambda: nb.shot_single_scale(True, 10), lambda: nb.shot_lrf(), lambda: nb.normals(), lambda: nb.export()):
            with pytest.raises(ShotFpfhError, match="another grid"):
                call()
        nb.free()
        nb = cloud.radius_search_self(0.04)
        again = nb.shot_single_scale(True, 10)
        nb.free()
        assert np.array_equal(again, first)  # (rebuilt with the first grid's cells: the same order, the same rows)
        # a rebuild with the SAME cells re-allocates the sorted arrays a self search's queries point into: refused as well
        nb = cloud.radius_search_self(0.04)
        cloud.build_grid(0.04)
        with pytest.raises(ShotFpfhError, match="another grid"):
            nb.shot_single_scale(True, 10)
        nb.free()
    finally:
        cloud.free()


# ---- RANSAC: device gather, draws from the raw stream, fits in stacks while K9 scores ---------------------------------------------
def test_ransac_pipeline_equals_the_per_draw_formulation(eng):
    """ransac_on_matches now gathers keypoints[indices] on the device, takes all draws from the generator's raw stream and fits
    chunks of draws while K9 scores the previous chunk.  Against the plain formulation -- rng.choice per draw, solver_point_to_point
    per draw, one sf_ransac_score over host-gathered points -- ratio and transform bit for bit, with negative indices among the
    matches (keypoints[-1] is the last keypoint)."""
    import shot_fpfh_amd.matching.ransac as R
    from shot_fpfh_amd.core import solver_point_to_point

    rng = np.random.default_rng(91)
    n_kp, n_m, n_draws, thr = 50_000, 120_000, 5_000, 0.01
    scan_kp = rng.random((n_kp, 3))
    rot = np.array([[0.8, -0.6, 0.0], [0.6, 0.8, 0.0], [0.0, 0.0, 1.0]])
    ref_kp = (scan_kp @ rot.T + np.array([0.2, -0.1, 0.3]))[rng.permutation(n_kp)]
    si = rng.integers(0, n_kp, n_m)
    ri = rng.integers(0, n_kp, n_m)
    good = rng.random(n_m) < 0.3  # a third of the matches are true correspondences
    back = np.empty(n_kp, np.int64)
    # (ref_kp = moved[perm]: the ref row that holds scan row s is inv_perm[s]; rebuilt from the points themselves)
    order = np.lexsort(ref_kp.T[::-1])
    moved = scan_kp @ rot.T + np.array([0.2, -0.1, 0.3])
    back[np.lexsort(moved.T[::-1])] = order
    ri[good] = back[si[good]]
    si[:50] -= n_kp  # negative indices
    R.rng = np.random.default_rng(seed=72)
    ratio, tf = R.ransac_on_matches(si, ri, scan_kp, ref_kp, n_draws=n_draws, draw_size=4, distance_threshold=thr, disable_progress_bar=True)
    gen = np.random.default_rng(seed=72)
    a, b = scan_kp[si], ref_kp[ri]
    records = np.empty((n_draws, 12))
    for d in range(n_draws):
        pick = gen.choice(n_m, 4, replace=False, shuffle=False)
        records[d] = solver_point_to_point(a[pick], b[pick]).as_row12()
    inl = eng.ransac_score(a, b, records, thr)
    best = int(np.argmax(inl))
    assert ratio == inl[best] / n_m and ratio > 0.25
    want = R.RigidTransform(records[best, :9].reshape(3, 3).copy(), records[best, 9:].copy())
    want.normalize_rotation()
    assert np.array_equal(tf.rotation, want.rotation) and np.array_equal(tf.translation, want.translation)
    assert R.rng.bit_generator.state == gen.bit_generator.state
    with pytest.raises(IndexError):
        R.ransac_on_matches(np.array([0, 1, 2, 3, n_kp]), np.arange(5), scan_kp, ref_kp, n_draws=40, disable_progress_bar=True)
    with pytest.raises(AttributeError):
        R.ransac_on_matches(si, ri, scan_kp, ref_kp, n_draws=0)


# ---- K8: the integer pre-filter (match_i8.hip) ------------------------------------------------------------------------------------
@pytest.mark.parametrize("splits", [None, 3, 11])
@pytest.mark.parametrize("case", ["shot_like", "fpfh_like", "wide_range", "descending"])
def test_match_i8_prefilter_equals_exact(eng, O, monkeypatch, case, splits):
    """The int8 matrix-core pre-filter only prunes: index AND distance equal scipy's / the exact kernel's bit for bit on the cases
    the FP16 pre-filter is held to (exact ties and duplicates, an equidistant pair, zero rows, 125- and 33-column rows, norms over
    twelve orders of magnitude, an order in which every later column is nearer) -- with the column range in one split, and cut
    into 3 and 11 (a row's minimum and its ties then sit in different splits: several live pairs per row)."""
    from test_hip_parity import _half_cases

    a, b = next((a, b) for name, a, b in _half_cases() if name == case)
    monkeypatch.setenv("SF_MATCH_I8", "1")
    if splits:
        monkeypatch.setenv("SF_MATCH_I8_SPLITS", str(splits))
    eng.profile_reset()
    eng.profile(True)
    idx, dist, col = eng.match_argmin(a, b, want_col=True)
    eng.profile(False)
    rep = eng.profile_report()
    io, do, co = O.match_argmin(a, b, want_col=True)
    assert np.array_equal(idx, io) and np.array_equal(dist, do) and np.array_equal(col, co)
    assert rep.get("k8_match_i8", (0, 0))[0] == 2, "the integer pre-filter was expected to run for both directions"
    if case == "shot_like":  # rows with a clear nearest descriptor are served without the FP16 pass ... mostly
        assert rep.get("k8_i8_collect", (0, 0))[0] == 2


def test_match_i8_masked_rows_and_rows_without_a_clear_minimum(eng, monkeypatch):
    """Resident, masked form through the integer pre-filter; a block of near-identical reference rows puts hundreds of columns
    inside the integer window of the scan rows that match them (candidate lists overflow -> the FP16 pass decides those rows)."""
    from shot_fpfh_amd.matching import basic_matching
    from shot_fpfh_amd.sharding import MatchJob

    rng = np.random.default_rng(833)
    a = rng.random((3000, 352)) * (rng.random((3000, 352)) < 0.3)
    b = a[rng.permutation(3000)][:2800] + 0.01 * rng.standard_normal((2800, 352))
    b[1000:1400] = b[1000] + 1e-5 * rng.standard_normal((400, 352))
    a[:60] = b[1000] + 1e-5 * rng.standard_normal((60, 352))
    a[[3, 99, 500]] = 0.0
    b[[7, 640]] = 0.0
    monkeypatch.setenv("SF_MATCH_I8", "1")
    job = MatchJob(eng, 352, 3000, 2800)
    eng.profile_reset()
    eng.profile(True)
    job.run(eng.empty((3000, 352)).from_host(a), eng.empty((2800, 352)).from_host(b))
    eng.profile(False)
    s1, r1 = job.matches()
    dist = job.dist.to_host()
    s2, r2 = basic_matching(a, b)
    assert np.array_equal(s1, s2) and np.array_equal(r1, r2)
    assert np.isinf(dist[[3, 99, 500]]).all() and np.isfinite(np.delete(dist, [3, 99, 500])).all()
    rep = eng.profile_report()
    assert rep.get("k8_match_i8", (0, 0))[0] >= 1 and rep.get("k8_match_half", (0, 0))[0] >= 1
    job.close()


def test_match_i8_pilot_hands_unclear_problems_to_the_fp16_pass(eng, O, monkeypatch):
    """Dense random rows have no nearest descriptor that stands clear of the rest: the pilot slab sees nearly every row flagged and
    the whole problem takes the FP16 pass; with planted copies the pilot passes and the rows beyond it are served.  Results equal
    the exact kernel's either way."""
    rng = np.random.default_rng(834)
    m = 6000
    b = rng.random((m, 352))
    b /= np.linalg.norm(b, axis=1)[:, None]
    a_clear = b[rng.permutation(m)] + 1e-4 * rng.standard_normal((m, 352))
    a_unclear = rng.random((m, 352))
    a_unclear /= np.linalg.norm(a_unclear, axis=1)[:, None]
    monkeypatch.setenv("SF_MATCH_I8", "1")
    monkeypatch.setenv("SF_I8_PILOT_ROWS", "1024")
    for a, pilot_passes in ((a_clear, True), (a_unclear, False)):
        eng.profile_reset()
        eng.profile(True)
        idx, dist = eng.match_argmin(a, b)[:2]
        eng.profile(False)
        rep = eng.profile_report()
        io, do = O.match_argmin(a, b)[:2]
        assert np.array_equal(idx, io) and np.array_equal(dist, do)
        assert rep.get("k8_match_i8", (0, 0))[0] == (2 if pilot_passes else 1), rep.get("k8_match_i8")
        assert rep.get("k8_i8_collect", (0, 0))[0] == (2 if pilot_passes else 1)  # (the pilot runs all steps on its slab)
        other = rep.get("k8_match_half", (0, 0))[0] + rep.get("k8_match_gemm", (0, 0))[0]  # (this size: the FP64 GEMM form)
        assert (other >= 1) == (not pilot_passes)


# ---- the descriptor exchange under K8 (streamed K8) ---------------------------------------------------------------------------
@pytest.mark.parametrize("chunks", [2, 5])
def test_streamed_k8_gives_the_one_shot_matches(O, monkeypatch, chunks):
    """MatchJob(chunks=C): the reference rows become available in C pieces (between ranks: grouped ncclSend / ncclRecv on the side
    stream) and every piece gets its int8 image and its share of the integer pass as it lands (sf_match_stream_feed); the decision
    steps run once over all pieces' minima (sf_match_stream_end).  Index and distance vectors equal the one-shot job's bit for
    bit -- exact ties across pieces (duplicated reference rows far apart in the set), masked (zero) rows on both sides, a piece
    that holds nothing but zero rows -- and basic_matching's pairs; the reciprocity test (column arg-min) agrees as well."""
    import shot_fpfh_amd as s
    from shot_fpfh_amd.sharding import MatchJob

    monkeypatch.setenv("SF_MATCH_I8", "1")  # (this size would take the FP64 GEMM form by itself)
    e2 = s.Engine(0)
    e2.comm_init(e2.comm_unique_id(), 1, 0)
    try:
        rng = np.random.default_rng(97)
        m1, m2, d = 2300, 2112, 352  # (2112 = 33 tiles of 64 rows)
        b = rng.random((m2, d)) * (rng.random((m2, d)) < 0.3)
        a = b[rng.integers(0, m2, m1)] + 1e-3 * rng.standard_normal((m1, d))
        b[1900] = b[30]      # exact duplicates in different pieces: the first (row 30) must win
        b[1050] = b[30]
        a[9] = b[30]
        a[[4, 700]] = 0.0
        b[[8, 2099]] = 0.0
        if chunks == 5:
            b[m2 - 400:] = 0.0  # the last piece is all zero rows
        da, db = e2.empty((m1, d)).from_host(a), e2.empty((m2, d)).from_host(b)
        plain = MatchJob(e2, d, m1, m2)
        plain.run(da, db)
        e2.profile_reset()
        e2.profile(True)
        job = MatchJob(e2, d, m1, m2, chunks=chunks)
        assert job.chunks == chunks
        job.run(da, db)
        e2.sync()
        e2.profile(False)
        rep = e2.profile_report()
        assert rep["k8_match_i8"][0] == chunks and rep["k8_i8_collect"][0] == 1  # a first pass per piece, ONE decision
        assert np.array_equal(job.idx.to_host(), plain.idx.to_host()) and np.array_equal(job.dist.to_host(), plain.dist.to_host())
        s1, r1 = job.matches()
        s2, r2 = O.basic_matching(a, b)
        assert np.array_equal(s1, s2) and np.array_equal(r1, r2)
        assert r1[np.flatnonzero(s1 == 9)[0]] == 30
        for kw in (dict(filter_nonreciprocal=True, n_min_matches=10), dict(filter_nonreciprocal=True, n_min_matches=10**6)):
            x, y = job.matches(**kw), plain.matches(**kw)
            assert np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1])
        # the integer pass switched off: the feeds cost nothing and the end runs the one-shot paths -- same vectors
        monkeypatch.setenv("SF_MATCH_I8", "0")
        job.run(da, db)
        assert np.array_equal(job.idx.to_host(), plain.idx.to_host()) and np.array_equal(job.dist.to_host(), plain.dist.to_host())
        job.close()
        plain.close()
    finally:
        e2.close()


# ---- K8: every path on random problems -----------------------------------------------------------------------------------------
def _random_match_problem(seed):
    """A problem drawn at random: sizes, row length, the kind of rows, how the scan rows relate to the reference rows, ties,
    zero rows, a block of near-identical reference rows, the scale of the numbers."""
    rng = np.random.default_rng(10_000 + seed)
    d = int(rng.choice([352, 352, 352, 125, 125, 33, 7, 64, 1, 351, 97]))
    if rng.random() < 0.15:  # small and degenerate shapes (the exact tile kernel serves these whatever is forced)
        m1, m2 = int(rng.integers(1, 300)), int(rng.integers(1, 300))
    else:  # at least 6e8 pair-dimensions where the row length allows: the matrix-core forms engage
        m1 = int(rng.integers(400, 4500))
        m2 = int(min(max(rng.integers(400, 6000), -(-600_000_000 // (m1 * d))), 60_000))
    kind = rng.choice(["sparse", "dense", "percent", "signed", "lattice"])
    if kind == "sparse":
        b = rng.random((m2, d)) * (rng.random((m2, d)) < rng.choice([0.05, 0.3, 0.8]))
    elif kind == "dense":
        b = rng.random((m2, d))
    elif kind == "percent":
        b = rng.random((m2, d)) ** 4 * 100.0
    elif kind == "signed":
        b = rng.standard_normal((m2, d))
    else:  # few distinct values per entry: exact ties between distances everywhere
        b = rng.integers(0, 3, (m2, d)).astype(np.float64) * 0.25
    if rng.random() < 0.5 and kind != "lattice":
        b /= np.maximum(np.linalg.norm(b, axis=1)[:, None], 1e-300)
    rel = rng.choice(["copies", "near", "unrelated", "mixed"])
    if rel == "unrelated":
        a = b[rng.integers(0, m2, m1)][:, rng.permutation(d)]
    else:
        noise = {"copies": 0.0, "near": 1e-3, "mixed": 0.05}[rel] * (b.std() + 1e-300)
        a = b[rng.integers(0, m2, m1)] + noise * rng.standard_normal((m1, d)) * (rng.random((m1, 1)) < 0.8)
    if m2 > 40 and rng.random() < 0.4:  # near-identical reference rows: long candidate lists
        w = int(rng.integers(2, min(m2 // 2, 700)))
        s = int(rng.integers(0, m2 - w))
        b[s:s + w] = b[s] + rng.choice([0.0, 1e-9, 1e-5]) * rng.standard_normal((w, d))
    if rng.random() < 0.5:  # duplicated reference rows far apart: the lower column must win
        for _ in range(int(rng.integers(1, 6))):
            i, j = rng.integers(0, m2, 2)
            b[j] = b[i]
    if rng.random() < 0.5:  # empty descriptors on both sides
        a[rng.integers(0, m1, max(1, m1 // 50))] = 0.0
        b[rng.integers(0, m2, max(1, m2 // 50))] = 0.0
    scale = float(rng.choice([1.0, 1.0, 1e-6, 1e5, 3.0]))
    return a * scale, b * scale


_STRESS_SEEDS = int(os.environ.get("SF_STRESS_SEEDS", "24"))


@pytest.mark.parametrize("seed", range(_STRESS_SEEDS))
def test_every_matching_path_agrees_on_random_problems(eng, O, monkeypatch, seed):
    """Integer pre-filter (forced, with a random number of column splits), FP16 pre-filter (forced), the dispatcher's own choice
    and the resident masked form: index and distance vectors equal to each other bit for bit and, for the first seeds, to
    SciPy's order of arithmetic in the oracle.  (SF_STRESS_SEEDS=n runs n problems.)"""
    from shot_fpfh_amd.sharding import MatchJob

    a, b = _random_match_problem(seed)
    rng = np.random.default_rng(seed)
    results = {}
    for name, env in (("i8", {"SF_MATCH_I8": "1", "SF_MATCH_I8_SPLITS": str(int(rng.integers(1, 12)))}),
                      ("half", {"SF_MATCH_I8": "0", "SF_MATCH_HALF": "1", "SF_MATCH_HALF_SPLITS": str(int(rng.integers(1, 7)))}),
                      ("auto", {})):
        with monkeypatch.context() as mp:
            for k in ("SF_MATCH_I8", "SF_MATCH_I8_SPLITS", "SF_MATCH_HALF", "SF_MATCH_HALF_SPLITS"):
                mp.delenv(k, raising=False)
            for k, v in env.items():
                mp.setenv(k, v)
            results[name] = eng.match_argmin(a, b, want_col=True)
    for name in ("half", "auto"):
        for x, y in zip(results["i8"], results[name]):
            assert np.array_equal(x, y), (seed, name)
    if a.shape[0] * b.shape[0] * a.shape[1] < (3e9 if seed < 6 else 4e8):
        for x, y in zip(results["i8"], O.match_argmin(a, b, want_col=True)):
            assert np.array_equal(x, y), (seed, "oracle")
    # the resident, masked form (zero rows never match and are never matched), streamed in a random number of pieces
    with monkeypatch.context() as mp:
        mp.setenv("SF_MATCH_I8", "1")
        m1, m2, d = a.shape[0], b.shape[0], a.shape[1]
        plain = MatchJob(eng, d, m1, m2)
        da, db = eng.empty((m1, d)).from_host(a), eng.empty((m2, d)).from_host(b)
        plain.run(da, db)
        mp.setenv("SF_MATCH_I8", "0")
        mp.setenv("SF_MATCH_HALF", "0")
        other = MatchJob(eng, d, m1, m2)
        other.run(da, db)
        assert np.array_equal(plain.idx.to_host(), other.idx.to_host()) and np.array_equal(plain.dist.to_host(), other.dist.to_host()), seed
        plain.close()
        other.close()
