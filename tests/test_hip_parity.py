"""GPU parity tests: the HIP path (through the C ABI, via the shot_fpfh_amd wrappers) against
  (1) the golden vectors the reference itself produced (tests/golden, tools/gen_golden.py) and
  (2) the CPU oracle on seeded random inputs, including BASELINE config C2 (100k points, 10k keypoints).

Tolerance (BASELINE.json north_star): neighbour index sets bit-exact; SHOT / FPFH values within
|a-b| <= 1e-5 * max(1, |b|); match indices equal.
"""
import numpy as np
import pytest

from conftest import load_golden, synth_cloud

pytestmark = pytest.mark.gpu

TOL = 1e-5


def close(a, b, tol=TOL):
    return np.abs(a - b) <= tol * np.maximum(1.0, np.abs(b))


@pytest.fixture(scope="module")
def eng():
    import shot_fpfh_amd as s

    return s.default_engine()


@pytest.fixture(scope="module")
def O():
    from oracle import oracle

    return oracle


# ---- K1 + K2 -------------------------------------------------------------------------------------
def test_radius_search_golden_bit_exact(eng):
    g = load_golden("nbrs_2k.npz")
    cloud = eng.cloud(g["cloud"])
    nb = cloud.radius_search(g["cloud"], float(g["radius"]))
    off, idx, dist = nb.export(return_distance=True)
    assert np.array_equal(off, g["offsets"])
    assert np.array_equal(idx, g["idx"])
    assert np.array_equal(dist, g["dist"])
    nb2 = cloud.radius_search(g["queries"], float(g["radius"]))
    off2, idx2 = nb2.export()
    assert np.array_equal(off2, g["q_offsets"]) and np.array_equal(idx2, g["q_idx"])


@pytest.mark.parametrize("n,m,r,seed", [(50000, 3000, 0.05, 21), (20000, 500, 0.31, 22), (1000, 64, 2.0, 23)])
def test_radius_search_vs_oracle(eng, O, n, m, r, seed):
    p, _, rng = synth_cloud(n, seed)
    q = np.vstack([p[rng.choice(n, m // 2, replace=False)], rng.random((m - m // 2, 3)) * 1.6 - 0.3])
    cloud = eng.cloud(p)
    off, idx, dist = cloud.radius_search(q, r).export(return_distance=True)
    off_o, idx_o, dist_o = O.radius_search(p, q, r, return_distance=True)
    assert np.array_equal(off, off_o) and np.array_equal(idx, idx_o) and np.array_equal(dist, dist_o)


def test_radius_search_self_matches_coordinate_queries(eng):
    p, _, _ = synth_cloud(30000, 24)
    cloud = eng.cloud(p)
    nb_self = cloud.radius_search_self(0.06)
    nb_q = cloud.radius_search(p, 0.06)
    assert nb_self.total == nb_q.total and nb_self.max_count == nb_q.max_count
    # ragged / degenerate inputs
    empty = cloud.radius_search(np.zeros((0, 3)), 0.1)
    assert empty.m == 0 and empty.total == 0
    one = eng.cloud(p[:1]).radius_search(p[:3], 0.5)
    off, idx = one.export()
    assert off.tolist()[0] == 0 and idx.tolist().count(0) == off[-1]


# ---- K3 ----------------------------------------------------------------------------------------------
def test_normals_golden(eng):
    import shot_fpfh_amd as s

    g = load_golden("normals_2k.npz")
    n1 = s.compute_normals(g["queries"], g["cloud"], radius=float(g["radius"]))
    assert np.abs(n1 - g["n_radius"]).max() < 1e-9  # sign as LAPACK returns it
    n2 = s.compute_normals(g["queries"], g["cloud"], radius=float(g["radius"]), pre_computed_normals=g["pre"])
    assert np.abs(n2 - g["n_radius_pre"]).max() < 1e-9
    n3 = s.compute_normals(g["queries"], g["cloud"], k=int(g["k"]))
    assert np.abs(n3 - g["n_knn"]).max() < 1e-9
    n4 = s.compute_normals(g["queries"], g["cloud"], k=int(g["k"]), pre_computed_normals=g["pre"])
    assert np.abs(n4 - g["n_knn_pre"]).max() < 1e-9
    with pytest.raises(ValueError):
        s.compute_normals(g["queries"], g["cloud"][:10], k=30)  # k > number of points, as sklearn


@pytest.mark.parametrize("n,m,k,seed", [(20000, 1500, 30, 71), (5000, 400, 100, 72), (3000, 300, 300, 73), (50, 20, 50, 74)])
def test_knn_search_vs_brute_force(eng, O, n, m, k, seed):
    """k-NN lists (KDTree.query) incl. queries far outside the cloud, which need several radius doublings."""
    p, _, rng = synth_cloud(n, seed)
    if seed == 72:  # strongly non-uniform density: a dense blob inside a sparse cloud
        p[: n // 2] = 0.5 + 0.02 * (p[: n // 2] - 0.5)
    q = np.vstack([p[rng.choice(n, m // 2, replace=False)], rng.random((m - m // 2, 3)) * 3.0 - 1.0])
    off, idx = eng.cloud(p).knn_search(q, k).export()
    assert np.array_equal(np.diff(off), np.full(m, k))
    off_o, idx_o = O.knn_lists(p, q, k)
    got = idx.reshape(m, k)
    want = np.sort(idx_o.reshape(m, k), axis=1)
    same = (got == want).all(axis=1)
    # rows may differ only where the k-th and (k+1)-th neighbours are exactly equidistant
    for i in np.flatnonzero(~same):
        d = np.sort(((p - q[i]) ** 2).sum(axis=1))
        assert d[k - 1] == d[k], f"query {i}: different neighbour set without a distance tie"
    assert same.mean() > 0.99


# ---- K4 + K5 ---------------------------------------------------------------------------------------------
def test_local_rf_golden(eng, O):
    g = load_golden("shot_150.npz")
    cloud = eng.cloud(g["cloud"], g["normals"])
    nb = cloud.radius_search(g["keypoints"], float(g["radius"]))
    lrf = nb.shot_lrf()
    off, _ = nb.export()
    ok = np.diff(off) >= 4  # rank-deficient supports have no defined frame (see test_oracle_golden)
    assert np.abs(lrf - g["lrf"])[ok].max() < 1e-9
    assert np.array_equal(lrf[-2], np.eye(3))


@pytest.mark.parametrize("norm", [True, False])
@pytest.mark.parametrize("mn", [10, 100])
def test_shot_single_scale_golden(norm, mn):
    from shot_fpfh_amd.descriptors import ShotMultiprocessor

    g = load_golden("shot_150.npz")
    with ShotMultiprocessor(normalize=norm, min_neighborhood_size=mn, verbose=False) as sm:
        d = sm.compute_descriptor_single_scale(g["cloud"], g["normals"], g["keypoints"], float(g["radius"]))
    ref = g[f"single_n{int(norm)}_m{mn}"]
    assert d.shape == ref.shape and d.dtype == np.float64
    assert close(d, ref).all(), f"max err {np.abs(d - ref).max()}"
    assert not d[-2].any() and not d[-1].any()


def test_shot_variants_golden():
    from shot_fpfh_amd.descriptors import ShotMultiprocessor

    g = load_golden("shot_150.npz")
    p, nr, kp, r = g["cloud"], g["normals"], g["keypoints"], float(g["radius"])
    with ShotMultiprocessor(normalize=True, min_neighborhood_size=10, verbose=False) as sm:
        d = sm.compute_descriptor_single_scale(p, nr, kp, r, subsampling_voxel_size=float(g["voxel"]))
        assert close(d, g["single_sub"]).all()
        d = sm.compute_descriptor_bi_scale(p, nr, kp[:60], local_rf_radius=0.08, shot_radius=r,
                                           subsampling_voxel_size=float(g["voxel"]))
        assert close(d, g["bi_scale_sub"]).all()
        d = sm.compute_descriptor_multiscale(p, nr, kp[:60], radii=[0.08, 0.12], weights=[1.0, 0.5])
        assert d.shape == (60, 704) and close(d, g["multi_shared"]).all()
        with pytest.raises(IndexError):
            sm.compute_descriptor_bi_scale(p, nr, kp[:5], 0.08, r)
    with ShotMultiprocessor(normalize=True, share_local_rfs=False, min_neighborhood_size=10, verbose=False) as sm:
        d = sm.compute_descriptor_multiscale(p, nr, kp[:60], radii=[0.08, 0.12], voxel_sizes=[0.008, 0.012])
        assert close(d, g["multi_unshared"]).all()
    with pytest.raises(AttributeError):
        ShotMultiprocessor().compute_descriptor_single_scale(p, nr, kp, r)


def test_shot_duplicates_edge(O):
    from shot_fpfh_amd.descriptors import ShotMultiprocessor

    e = load_golden("edge_dups.npz")
    with ShotMultiprocessor(normalize=True, min_neighborhood_size=5, verbose=False) as sm:
        d = sm.compute_descriptor_single_scale(e["cloud"], e["normals"], e["keypoints"], float(e["radius"]))
    off, idx, dist = O.radius_search(e["cloud"], e["keypoints"], float(e["radius"]), return_distance=True)
    tied = np.array([len(np.unique(dist[off[i]:off[i + 1]])) < off[i + 1] - off[i] for i in range(len(off) - 1)])
    assert close(d, e["shot_m5"])[~tied].all()  # rows with tied rho are undefined in the reference itself


# ---- K6 + K7 ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("fix,nb,key", [("fpfh_200.npz", 5, "fpfh5"), ("fpfh_200.npz", 4, "fpfh4"),
                                        ("fpfh_surface.npz", 5, "fpfh5"), ("fpfh_surface.npz", 3, "fpfh3")])
def test_fpfh_golden(fix, nb, key):
    import shot_fpfh_amd as s

    g = load_golden(fix)
    f = s.compute_fpfh_descriptor(g["kp_idx"], g["cloud"], g["normals"], float(g["radius"]), nb, verbose=False)
    assert f.shape == g[key].shape and f.dtype == np.float64
    assert close(f, g[key]).all(), f"max err {np.abs(f - g[key]).max()}"


def test_fpfh_duplicates_and_spfh(O):
    import shot_fpfh_amd as s

    e = load_golden("edge_dups.npz")
    f, spfh = s.compute_fpfh_descriptor(e["kp_idx"], e["cloud"], e["normals"], float(e["radius"]), 5, verbose=False,
                                        return_spfh=True)
    assert close(f, e["fpfh5"]).all()
    _, spfh_o = O.compute_fpfh_descriptor(e["kp_idx"], e["cloud"], e["normals"], float(e["radius"]), 5, return_spfh=True)
    assert np.array_equal(spfh, spfh_o)  # integer counts / k: bit-exact
    with pytest.raises(ValueError):
        s.compute_fpfh_descriptor(e["kp_idx"], e["cloud"], e["normals"], 0.1, 5, decorrelated=True)
    with pytest.raises(s.ShotFpfhError):
        s.compute_fpfh_descriptor(np.array([10**6]), e["cloud"], e["normals"], 0.1, 5, verbose=False)


# ---- config C2: 100k points, 10k keypoints, r = 0.05, SHOT + FPFH, vs the oracle ---------------------------------
def test_config_c2_shot_fpfh_vs_oracle(O):
    import shot_fpfh_amd as s
    from shot_fpfh_amd.descriptors import ShotMultiprocessor

    p, nr, rng = synth_cloud(100000, 2)
    kp = np.sort(rng.choice(100000, 10000, replace=False))
    r = 0.05
    f = s.compute_fpfh_descriptor(kp, p, nr, r, 5, verbose=False)
    fo = O.compute_fpfh_descriptor(kp, p, nr, r, 5)
    bad = (~close(f, fo)).any(axis=1).sum()
    assert bad == 0, f"{bad} FPFH rows outside tolerance, max err {np.abs(f - fo).max()}"
    with ShotMultiprocessor(normalize=True, min_neighborhood_size=10, verbose=False) as sm:
        d = sm.compute_descriptor_single_scale(p, nr, p[kp], r)
    do = O.shot_single_scale(p, nr, p[kp], r, True, 10)
    bad_rows = np.flatnonzero((~close(d, do)).any(axis=1))
    # a bin decision that sits on a rounding boundary could flip a row; the observed number on MI355X is 0 and the
    # test holds the kernel to it (a flipped row would be listed here, not hidden)
    assert bad_rows.size == 0, f"{bad_rows.size} SHOT rows outside tolerance: {bad_rows[:10]}"
    nz = d.any(axis=1)  # corner keypoints with <= 10 neighbours stay all-zero (shot.py:212)
    assert np.array_equal(nz, do.any(axis=1)) and nz.sum() > 9900
    assert np.abs(np.linalg.norm(d[nz], axis=1) - 1.0).max() < 1e-12


# ---- K8 ---------------------------------------------------------------------------------------------------------------
def test_matching_golden():
    from shot_fpfh_amd.matching import basic_matching, match_descriptors, threshold_filter

    g = load_golden("match_300.npz")
    s_, r_ = basic_matching(g["scan"], g["ref"])
    assert s_.dtype == np.int64 and np.array_equal(s_, g["basic_s"]) and np.array_equal(r_, g["basic_r"])
    s_, r_ = match_descriptors(g["scan"], g["ref"], verbose=False)
    assert np.array_equal(s_, g["md_s"]) and np.array_equal(r_, g["md_r"])
    s_, r_ = match_descriptors(g["scan"], g["ref"], threshold_filter, verbose=False, threshold_multiplier=10)
    assert np.array_equal(s_, g["thr_s"]) and np.array_equal(r_, g["thr_r"])
    s_, r_ = match_descriptors(g["scan"], g["ref"], filter_nonreciprocal=True, verbose=False, n_min_matches=100)
    assert np.array_equal(s_, g["rec_s"]) and np.array_equal(r_, g["rec_r"])
    s_, r_ = match_descriptors(g["scan"], g["ref"], filter_nonreciprocal=True, verbose=False, n_min_matches=10**6)
    assert np.array_equal(s_, g["recbig_s"]) and np.array_equal(r_, g["recbig_r"])


@pytest.mark.parametrize("m1,m2,d", [(1000, 777, 352), (65, 4100, 125), (3, 2, 7)])
def test_match_argmin_vs_oracle_bit_exact(eng, O, m1, m2, d):
    rng = np.random.default_rng(31)
    a, b = rng.random((m1, d)), rng.random((m2, d))
    b[min(5, m2 - 1)] = b[0]  # exact tie: the first minimum must win
    a[0] = b[0]
    idx, dist, col = eng.match_argmin(a, b, want_col=True)
    io, do, co = O.match_argmin(a, b, want_col=True)
    assert np.array_equal(idx, io) and np.array_equal(dist, do) and np.array_equal(col, co)
    assert idx[0] == 0


# ---- K9 ------------------------------------------------------------------------------------------------------------------
def test_ransac_golden(eng):
    import shot_fpfh_amd.matching.ransac as R

    g = load_golden("ransac_500.npz")
    a, b = g["scan_kp"][g["scan_idx"]], g["ref_kp"][g["ref_idx"]]
    inl = eng.ransac_score(a, b, g["draw_rt"], float(g["thr"]))
    assert np.array_equal(inl, g["draw_inliers"])
    R.rng = np.random.default_rng(seed=72)  # a fresh process, as when the golden was made
    ratio, tf = R.ransac_on_matches(g["scan_idx"], g["ref_idx"], g["scan_kp"], g["ref_kp"], n_draws=int(g["n_draws"]),
                                    draw_size=4, distance_threshold=float(g["thr"]), disable_progress_bar=True)
    assert ratio == float(g["ratio"])
    assert np.array_equal(tf.rotation, g["rotation"]) and np.array_equal(tf.translation, g["translation"])


def test_ransac_score_threshold_band_and_tiling(eng):
    """K9 decides `norm <= thr` on the squared residual and takes the square root only inside the one-ulp band around
    thr^2: residuals sitting exactly on / next to the threshold, a ragged pair count (tile tail), more draws than one
    workgroup's LDS counters, thresholds 0 / inf / NaN -- all against the NumPy expression of ransac.py:60-67."""
    rng = np.random.default_rng(77)
    thr = 0.0123
    xs = [thr]
    for _ in range(4):
        xs.append(np.nextafter(xs[-1], np.inf))
    x = thr
    for _ in range(4):
        x = np.nextafter(x, -np.inf)
        xs.append(x)
    xs = np.array(xs)
    m = 2048 * 3 + 37
    a = rng.random((m, 3))
    b = a + 0.02 * rng.standard_normal((m, 3))
    k = xs.shape[0]
    a[:k] = 0.0
    b[:k] = 0.0
    b[:k, 0] = xs  # identity draw: residual norm is exactly xs
    b[k:2 * k, 1] = a[k:2 * k, 1]  # keep the rest generic
    n_draws = 8192 + 301
    rt = np.tile(np.concatenate((np.eye(3).ravel(), np.zeros(3))), (n_draws, 1))
    rt[1:, 9:] = 0.02 * rng.standard_normal((n_draws - 1, 3))
    rt[5:, :9] += 1e-3 * rng.standard_normal((n_draws - 5, 9))

    def ref(thr_):
        with np.errstate(invalid="ignore"):
            return np.array([(np.linalg.norm(a @ r[:9].reshape(3, 3).T + r[9:] - b, axis=1) <= thr_).sum() for r in rt])

    for t in (thr, 0.0, np.inf, np.nan, -1.0):
        assert np.array_equal(eng.ransac_score(a, b, rt, t), ref(t)), t
    assert eng.ransac_score(a[:0], b[:0], rt[:7], thr).tolist() == [0] * 7


# ---- sharding on one device: G shards run one after the other must reproduce the single-shard result -------------
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_blocks_bit_identical_to_single(eng, O, world):
    from shot_fpfh_amd.sharding import DescriptorJob

    p, nr, _ = synth_cloud(20000, 51)
    r = 0.08

    def run(w):
        f, s = np.zeros((20000, 125)), np.zeros((20000, 352))
        seen = np.zeros(20000, dtype=int)
        for rank in range(w):
            job = DescriptorJob(eng, p, nr, r, n_bins=5, normalize=True, min_neighborhood_size=10, world=w, rank=rank, spfh_exchange="halo")
            job.step()
            rows = job.block_original_indices()
            f[rows], s[rows] = job.fpfh_out.to_host(), job.shot_out.to_host()
            seen[rows] += 1
            job.close()
        assert (seen == 1).all()
        return f, s

    f1, s1 = run(1)
    fw, sw = run(world)
    assert np.array_equal(f1, fw) and np.array_equal(s1, sw)
    fo = O.compute_fpfh_descriptor(np.arange(20000), p, nr, r, 5)
    assert close(f1, fo).all()


def test_shot_large_neighbourhoods_streaming_kernel(O):
    """k > 256 neighbours per keypoint takes the streaming K5 kernel instead of the register-cached one."""
    from shot_fpfh_amd.descriptors import ShotMultiprocessor

    p, nr, rng = synth_cloud(4000, 61)
    kp = p[rng.choice(4000, 200, replace=False)]
    for r in (0.3, 0.17):  # ~450 and ~80 neighbours
        with ShotMultiprocessor(normalize=True, min_neighborhood_size=10, verbose=False) as sm:
            d = sm.compute_descriptor_single_scale(p, nr, kp, r)
        do = O.shot_single_scale(p, nr, kp, r, True, 10)
        assert close(d, do).all(), f"r={r}: max err {np.abs(d - do).max()}"


# ---- config C4 in small: two clouds related by a rigid motion -> SHOT -> basic_matching -> RANSAC -------------
def test_config_c4_registration_chain(O):
    from scipy.spatial.transform import Rotation

    import shot_fpfh_amd.matching.ransac as R
    from shot_fpfh_amd.descriptors import ShotMultiprocessor
    from shot_fpfh_amd.matching import basic_matching

    n = 20000
    scan, nrm, rng = synth_cloud(n, 4)
    rot = Rotation.from_euler("xyz", [0.3, -0.2, 0.5]).as_matrix()
    t = np.array([0.1, -0.3, 0.2])
    perm = rng.permutation(n)
    ref, ref_nrm = (scan @ rot.T + t)[perm], (nrm @ rot.T)[perm]
    kp_s = np.sort(rng.choice(n, 3000, replace=False))
    kp_r = np.argsort(perm)[kp_s]  # the same physical points in the reference cloud
    with ShotMultiprocessor(normalize=True, min_neighborhood_size=10, verbose=False) as sm:
        ds = sm.compute_descriptor_single_scale(scan, nrm, scan[kp_s], 0.08)
        dr = sm.compute_descriptor_single_scale(ref, ref_nrm, ref[kp_r], 0.08)
    si, ri = basic_matching(ds, dr)
    io, _ = O.match_argmin(ds[np.any(ds, axis=1)], dr[np.any(dr, axis=1)])
    assert np.array_equal(ri, np.flatnonzero(np.any(dr, axis=1))[io])  # same arg-min as the oracle, bit for bit
    # frames whose sign vote is within +-1 of a tie keep LAPACK's arbitrary sign (shot.py:40-45), so a rigid copy
    # does not reproduce every descriptor -- the reference behaves the same; most matches are still right
    correct = (si == ri).mean()
    assert correct > 0.6, correct
    R.rng = np.random.default_rng(seed=72)
    ratio, tf = R.ransac_on_matches(si, ri, scan[kp_s], ref[kp_r], n_draws=300, draw_size=4, distance_threshold=0.01,
                                    disable_progress_bar=True)
    assert ratio >= correct - 1e-9
    assert np.abs(tf.rotation - rot).max() < 1e-6 and np.abs(tf.translation - t).max() < 1e-6


def test_matching_multiscale_golden():
    from shot_fpfh_amd.matching import match_descriptors, threshold_filter

    g = load_golden("match3d_200.npz")
    s_, r_ = match_descriptors(g["scan"], g["ref"], verbose=False)
    assert np.array_equal(s_, g["md_s"]) and np.array_equal(r_, g["md_r"])
    s_, r_ = match_descriptors(g["scan"], g["ref"], threshold_filter, verbose=False, threshold_multiplier=3)
    assert np.array_equal(s_, g["thr_s"]) and np.array_equal(r_, g["thr_r"])
    s_, r_ = match_descriptors(g["scan"], g["ref"], filter_nonreciprocal=True, verbose=False, n_min_matches=10**6)
    assert np.array_equal(s_, g["rec_s"]) and np.array_equal(r_, g["rec_r"])


def test_match_job_resident_equals_basic_matching(eng):
    """The sharded matching step (device-resident, zero rows masked in place) on one rank == basic_matching."""
    from shot_fpfh_amd.matching import basic_matching
    from shot_fpfh_amd.sharding import MatchJob

    rng = np.random.default_rng(81)
    a = rng.random((700, 352)) * (rng.random((700, 352)) < 0.3)
    b = a[rng.permutation(700)][:650] + 0.01 * rng.standard_normal((650, 352))
    a[[3, 99, 500]] = 0.0
    b[[7, 640]] = 0.0
    job = MatchJob(eng, 352, 700, 650)
    job.run(eng.empty((700, 352)).from_host(a), eng.empty((650, 352)).from_host(b))
    s1, r1 = job.matches()
    s2, r2 = basic_matching(a, b)
    assert np.array_equal(s1, s2) and np.array_equal(r1, r2)
    job.close()


def test_match_gemm_fast_path_equals_exact(eng, O):
    """Large problems take the FP64 matrix-core path (match_gemm.hip); the result must equal the exact
    kernel's / scipy's bit for bit, including duplicated rows (exact ties -> first index) and near ties."""
    rng = np.random.default_rng(83)
    m1, m2, d = 2100, 2300, 352
    b = rng.random((m2, d)) * (rng.random((m2, d)) < 0.3)
    b /= np.maximum(np.linalg.norm(b, axis=1)[:, None], 1e-300)
    a = b[rng.integers(0, m2, m1)] + 1e-3 * rng.standard_normal((m1, d))
    b[1200] = b[17]  # exact duplicate reference rows: every scan row near them has a two-way exact tie
    b[1201] = b[17]
    a[5] = b[17]
    a[6] = 0.5 * (b[40] + b[41])  # equidistant (up to rounding) from two reference rows
    idx, dist, col = eng.match_argmin(a, b, want_col=True)
    io, do, co = O.match_argmin(a, b, want_col=True)
    assert np.array_equal(idx, io) and np.array_equal(dist, do) and np.array_equal(col, co)
    assert idx[5] == 17
    rep = eng.profile_report()
    assert rep.get("k8_match_gemm", (0, 0))[0] >= 1, "the GEMM path was expected to run for this size"


def _half_cases():
    rng = np.random.default_rng(831)
    # (a) SHOT-like: sparse unit rows, scan = perturbed reference rows, exact duplicates, an equidistant pair, zero rows
    m1, m2, d = 1500, 1700, 352
    b = rng.random((m2, d)) * (rng.random((m2, d)) < 0.3)
    b /= np.maximum(np.linalg.norm(b, axis=1)[:, None], 1e-300)
    a = b[rng.integers(0, m2, m1)] + 1e-3 * rng.standard_normal((m1, d))
    b[1200] = b[17]; b[1201] = b[17]; a[5] = b[17]
    a[6] = 0.5 * (b[40] + b[41])
    a[7] = 0.0
    b[9] = 0.0
    yield "shot_like", a, b
    # (b) FPFH-like: 125 columns of percentages (short K: the 8-step instantiation), many near-equal rows
    m1, m2, d = 2500, 2600, 125
    b = rng.random((m2, d)) ** 4 * 100.0
    b[100:400] = b[100] + 1e-4 * rng.standard_normal((300, d))  # 300 columns within the FP16 window of each other
    a = b[rng.integers(0, m2, m1)] + 0.05 * rng.standard_normal((m1, d))
    yield "fpfh_like", a, b
    # (c) odd length, rows spanning twelve orders of magnitude in norm
    m1, m2, d = 4200, 4100, 33
    b = rng.standard_normal((m2, d)) * 10.0 ** rng.integers(-6, 6, (m2, 1))
    a = rng.standard_normal((m1, d)) * 10.0 ** rng.integers(-6, 6, (m1, 1))
    a[:500] = b[rng.integers(0, m2, 500)] * (1 + 1e-9)
    yield "wide_range", a, b
    # (d) adversarial order: every later column is nearer than all earlier ones for every scan row, so each
    # column tile lowers every threshold and the candidate lists overflow -> float64 path for those rows
    m1, m2, d = 1300, 1500, 352
    u = rng.random(d); u /= np.linalg.norm(u)
    b = np.linspace(0.2, 0.9, m2)[:, None] * u[None, :] + 1e-7 * rng.standard_normal((m2, d))
    a = (1.0 + 0.1 * rng.random((m1, 1))) * u[None, :] + 1e-7 * rng.standard_normal((m1, d))
    yield "descending", a, b


@pytest.mark.parametrize("splits", [None, 5])
@pytest.mark.parametrize("case", ["shot_like", "fpfh_like", "wide_range", "descending"])
def test_match_half_prefilter_equals_exact(eng, O, monkeypatch, case, splits):
    """The FP16 matrix-core pre-filter (match_half.hip) only prunes: index AND distance must equal scipy's / the exact
    kernel's bit for bit on ties, near ties, zero rows, short / odd descriptor lengths, extreme norms and an
    order that overflows every candidate list."""
    a, b = next((a, b) for name, a, b in _half_cases() if name == case)
    monkeypatch.setenv("SF_MATCH_HALF", "1")
    if splits:  # column splits (normally chosen for problems with few 256-row blocks): per-split thresholds, merged at the end
        monkeypatch.setenv("SF_MATCH_HALF_SPLITS", str(splits))
    eng.profile_reset()
    eng.profile(True)
    idx, dist, col = eng.match_argmin(a, b, want_col=True)
    eng.profile(False)
    rep = eng.profile_report()
    io, do, co = O.match_argmin(a, b, want_col=True)
    assert np.array_equal(idx, io) and np.array_equal(dist, do) and np.array_equal(col, co)
    assert rep.get("k8_match_half", (0, 0))[0] == 2, "the FP16 pre-filter was expected to run for both directions"
    overflow = rep.get("k8_match_gemm_overflow", (0, 0))[0]
    if case == "descending":
        assert overflow >= 1, "the adversarial order should have overflowed candidate lists"
    if case == "shot_like":  # (with splits too: a split full of ties far above the final minimum is ignored)
        assert overflow == 0, "no candidate list should overflow on ordinary data"


def test_match_half_prefilter_masked_rows(eng, monkeypatch):
    """Resident, masked form (zero descriptors never match and are never matched) through the pre-filter."""
    from shot_fpfh_amd.matching import basic_matching
    from shot_fpfh_amd.sharding import MatchJob

    rng = np.random.default_rng(832)
    a = rng.random((1500, 352)) * (rng.random((1500, 352)) < 0.3)
    b = a[rng.permutation(1500)][:1400] + 0.01 * rng.standard_normal((1400, 352))
    a[[3, 99, 500]] = 0.0
    b[[7, 640]] = 0.0
    monkeypatch.setenv("SF_MATCH_HALF", "1")
    job = MatchJob(eng, 352, 1500, 1400)
    eng.profile_reset()
    eng.profile(True)
    job.run(eng.empty((1500, 352)).from_host(a), eng.empty((1400, 352)).from_host(b))
    eng.profile(False)
    s1, r1 = job.matches()
    dist = job.dist.to_host()
    s2, r2 = basic_matching(a, b)
    assert np.array_equal(s1, s2) and np.array_equal(r1, r2)
    assert np.isinf(dist[[3, 99, 500]]).all() and np.isfinite(np.delete(dist, [3, 99, 500])).all()
    assert eng.profile_report().get("k8_match_half", (0, 0))[0] >= 1
    job.close()


# ---- less travelled paths ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("nb", [1, 2, 7, 8])
def test_fpfh_other_bin_counts_vs_oracle(O, nb):
    import shot_fpfh_amd as s

    p, nr, rng = synth_cloud(3000, 95)
    kp = np.sort(rng.choice(3000, 150, replace=False))
    f = s.compute_fpfh_descriptor(kp, p, nr, 0.13, nb, verbose=False)
    fo = O.compute_fpfh_descriptor(kp, p, nr, 0.13, nb)
    assert f.shape == (150, nb**3) and close(f, fo).all()
    with pytest.raises(ValueError):
        s.compute_fpfh_descriptor(kp, p, nr, 0.13, 0, verbose=False)  # (n_bins >= 1; above 32: tests/test_hip_round3.py)


def test_fpfh_wide_count_table(eng, O):
    """Neighbourhoods above 65 535 points switch the SPFH table to 32-bit counts; force that layout on a small cloud."""
    from shot_fpfh_amd.engine import Spfh

    p, nr, rng = synth_cloud(2500, 96)
    cloud = eng.cloud(p, nr)
    nb = cloud.radius_search_self(0.14)
    sp = Spfh(cloud, 5, max_count=70000).compute(nb)
    kp = np.sort(rng.choice(2500, 200, replace=False))
    f = sp.fpfh(nb, kp)
    fo, spo = O.compute_fpfh_descriptor(kp, p, nr, 0.14, 5, return_spfh=True)
    assert close(f, fo).all() and np.array_equal(sp.export(), spo)


def test_degenerate_inputs(eng):
    import shot_fpfh_amd as s
    from shot_fpfh_amd.descriptors import ShotMultiprocessor

    p, nr, _ = synth_cloud(500, 97)
    # empty keypoint sets
    assert s.compute_fpfh_descriptor(np.zeros(0, dtype=np.int64), p, nr, 0.2, 5, verbose=False).shape == (0, 125)
    with ShotMultiprocessor(min_neighborhood_size=5, verbose=False) as sm:
        assert sm.compute_descriptor_single_scale(p, nr, np.zeros((0, 3)), 0.2).shape == (0, 352)
        # all points identical: every rho is 0 -> all-zero descriptors, identity-less but finite frames
        same = np.tile(p[:1], (50, 1))
        d = sm.compute_descriptor_single_scale(same, nr[:50], same[:5], 0.1)
        assert d.shape == (5, 352) and not d.any()
    f = s.compute_fpfh_descriptor(np.arange(5), same, nr[:50], 0.1, 5, verbose=False)
    assert f.shape == (5, 125) and not f.any()  # no pair at non-zero distance -> empty histograms
    # non-finite coordinates are rejected loudly
    bad = p.copy()
    bad[7, 1] = np.nan
    with pytest.raises(s.ShotFpfhError, match="non-finite"):
        s.compute_fpfh_descriptor(np.arange(5), bad, nr, 0.2, 5, verbose=False)
    with pytest.raises(s.ShotFpfhError):
        eng.cloud(p).radius_search(p[:3], -1.0)
    # one wave spans several tiny clouds' worth of points: n smaller than a wave
    tiny = eng.cloud(p[:3]).radius_search(p[:3], 10.0).export()
    assert tiny[0].tolist() == [0, 3, 6, 9]


def test_two_stream_overlap_gives_identical_results(eng):
    """FPFH and SHOT chains on the context's two HIP streams (fork / switch / join) == single-stream results."""
    from shot_fpfh_amd.sharding import DescriptorJob

    p, nr, _ = synth_cloud(30000, 99)
    outs = []
    for overlap in (False, True):
        job = DescriptorJob(eng, p, nr, 0.07, overlap_chains=overlap)
        job.step()
        job.step()
        outs.append((job.fpfh_out.to_host(), job.shot_out.to_host(), job.lrf_out.to_host()))
        job.close()
    for a, b in zip(*outs):
        assert np.array_equal(a, b)


# ---- keypoint selection, PCA features (kernels K2 + K3) ---------------------------
def test_keypoint_selection_search_branches_match_reference_golden(eng):
    import shot_fpfh_amd.keypoint_selection as ks

    g = load_golden("keypoints_6k.npz")
    p = g["cloud"]
    assert np.array_equal(ks.select_keypoints_iteratively(p, float(g["radius"])), g["iterative"])
    got = ks.select_keypoints_with_density_threshold(p, float(g["voxel"]), 25, float(g["density_radius"]))
    assert np.array_equal(got, g["density_radius_sel"])
    assert ks.select_keypoints_iteratively(np.zeros((0, 3)), 0.1).size == 0


def test_local_pca_and_features_match_reference_golden(eng):
    from shot_fpfh_amd.descriptors import (
        compute_local_pca_with_moments,
        compute_pca_based_basic_features,
        compute_pca_based_features,
        compute_sphericity,
    )

    g = load_golden("pca_features_300.npz")
    q, p, r = g["queries"], g["cloud"], float(g["radius"])
    w, v, mo, sizes = compute_local_pca_with_moments(q, p, radius=r)
    assert sizes == list(g["sizes"])
    assert np.abs(w - g["eigenvalues"]).max() < 1e-12 and np.abs(mo - g["moments"]).max() < 1e-12
    assert np.abs(v - g["eigenvectors"]).max() < 1e-9  # eigenvector signs as LAPACK returns them
    wk, vk, mok, sk = compute_local_pca_with_moments(q, p, nghbrd_search="knn", k=int(g["k"]))
    assert sk == list(g["sizes_knn"])
    assert np.abs(wk - g["eigenvalues_knn"]).max() < 1e-12 and np.abs(mok - g["moments_knn"]).max() < 1e-12
    assert np.abs(vk - g["eigenvectors_knn"]).max() < 1e-9
    assert close(compute_pca_based_features(q, p, r), g["features"], 1e-9).all()
    for got, key in zip(compute_pca_based_basic_features(q, p, r), ("verticality", "linearity", "planarity", "basic_sphericity")):
        assert close(got, g[key], 1e-9).all()
    assert close(compute_sphericity(q, p, r), g["sphericity"], 1e-9).all()


def test_local_pca_vs_oracle_on_a_seeded_cloud(eng, O):
    p, _, rng = synth_cloud(40000, 41)
    q = np.vstack([p[rng.choice(40000, 1500, replace=False)], rng.random((37, 3))])  # 1537 queries: ragged last wave
    cloud = eng.cloud(p)
    nb = cloud.radius_search(q, 0.07)
    w, v, mo = nb.pca(moments=True)
    wo, vo, moo, sizes = O.local_pca(q, p, radius=0.07, moments=True)
    ok = sizes >= 4  # rank-deficient neighbourhoods have LAPACK-arbitrary eigenvectors
    assert np.array_equal(nb.counts(), sizes)
    assert np.abs(w - wo)[ok].max() < 1e-12 and np.abs(mo - moo)[ok].max() < 1e-12
    gap = np.minimum(wo[:, 1] - wo[:, 0], wo[:, 2] - wo[:, 1]) / wo[:, 2]
    good = ok & (gap > 1e-3)
    assert good.sum() > 1400 and np.abs(v - vo)[good].max() < 1e-9
    w2, v2 = nb.pca()
    assert np.array_equal(w2, w) and np.array_equal(v2, v)


def test_icp_matches_reference_golden(eng):
    """ICP with the nearest-neighbour query on the device (k-NN kernel, k = 1)."""
    import shot_fpfh_amd.icp as icp
    from shot_fpfh_amd.core import RigidTransform

    g = load_golden("icp_3500.npz")
    tf, rms, ok = icp.icp_point_to_plane(g["scan"], g["ref"], g["ref_normals"], RigidTransform(), d_max=float(g["d_max"]),
                                         voxel_size=float(g["voxel"]), max_iter=int(g["plane_max_iter"]),
                                         rms_threshold=float(g["plane_rms_threshold"]), disable_progress_bar=True)
    assert np.abs(tf.rotation - g["plane_rotation"]).max() < 1e-9 and np.abs(tf.translation - g["plane_translation"]).max() < 1e-9
    assert abs(rms - float(g["plane_rms"])) < 1e-9 and bool(ok) == bool(g["plane_converged"])
    err, moved = icp.compute_point_to_point_error(g["scan"], g["ref"], tf)
    assert abs(err - float(g["p2p_error"])) < 1e-9 and np.abs(moved[:50] - g["moved_head"]).max() < 1e-9
    np.random.seed(int(g["sampling_seed"]))
    aligned, rms_s, ok_s = icp.icp_point_to_point_with_sampling(
        g["scan"], g["ref"], d_max=float(g["d_max"]), max_iter=int(g["sampling_max_iter"]),
        rms_threshold=float(g["sampling_rms_threshold"]), sampling_limit=int(g["sampling_limit"]), disable_progress_bar=True)
    assert np.abs(aligned[:200] - g["sampling_aligned_head"]).max() < 1e-9 and abs(rms_s - float(g["sampling_rms"])) < 1e-9
    tf2, rms2, ok2 = icp.icp_point_to_point(g["scan"], g["ref"], RigidTransform(), d_max=float(g["d_max"]),
                                            voxel_size=float(g["voxel"]), max_iter=30, rms_threshold=1e-9)
    assert np.abs(tf2.rotation - g["true_rotation"]).max() < 5e-3


def test_register_point_clouds_script_end_to_end(eng, tmp_path):
    """PLY in -> normals -> keypoints -> SHOT -> matching -> RANSAC -> ICP -> metrics, through
    scripts/register_point_clouds.py, recovers the motion that generated the scan."""
    import importlib.util
    import os

    from conftest import ROOT
    from shot_fpfh_amd.helpers import read_ply, write_ply

    g = load_golden("icp_3500.npz")
    scan_file, ref_file = str(tmp_path / "scan.ply"), str(tmp_path / "ref.ply")
    write_ply(scan_file, [g["scan"]], ["x", "y", "z"])
    write_ply(ref_file, [g["ref"], g["ref_normals"]], ["x", "y", "z", "nx", "ny", "nz"])
    spec = importlib.util.spec_from_file_location("register_point_clouds", os.path.join(ROOT, "scripts", "register_point_clouds.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = str(tmp_path / "aligned")
    rc = mod.main([scan_file, ref_file, "--radius", "0.2", "--keypoints", "subsampling", "--keypoint-size", "0.05",
                   "--min-neighborhood-size", "10", "--ransac-draws", "2000", "--ransac-threshold", "0.02",
                   "--icp", "point_to_plane", "--icp-dmax", "0.05", "--icp-voxel", "0.04", "--icp-rms", "1e-9",
                   "--metric-threshold", "0.01", "--write", out])
    assert rc == 0
    merged = read_ply(out + "_icp.ply")
    n_scan = g["scan"].shape[0]
    assert merged.shape[0] == n_scan + g["ref"].shape[0] and merged["is_scan"][:n_scan].all() and not merged["is_scan"][n_scan:].any()
    moved = np.vstack((merged["x"], merged["y"], merged["z"])).T[:n_scan]
    truth = g["scan"] @ g["true_rotation"].T + g["true_translation"]
    assert np.abs(moved - truth).max() < 2e-3  # the script found the generating motion


def test_block_grid_build_keeps_global_numbering(eng):
    """sf_cloud_build_grid_block: only the slab a block needs is sorted, yet positions / lists / results are those
    of the whole-cloud build; a search outside the populated slab falls back to a whole build."""
    p, nr, _ = synth_cloud(30000, 71)
    r = 0.05
    full = eng.cloud(p, nr)
    full.build_grid(r)
    perm_full = full.perm()
    b, e = 11000, 17000
    hb, he = full.halo_range(b, e)
    off_f, idx_f = full.radius_search_self(r, hb, he).export()
    part = eng.cloud(p, nr)
    pb, pe = part.build_grid(r, block=(b, e), reach=2)
    assert 0 < pb <= hb and he <= pe < 30000 and pe - pb < 20000  # a slab, not the whole cloud
    assert np.array_equal(part.perm()[pb:pe], perm_full[pb:pe])
    assert part.halo_range(b, e) == (hb, he)
    off_p, idx_p = part.radius_search_self(r, hb, he).export()
    assert np.array_equal(off_p, off_f) and np.array_equal(idx_p, idx_f)
    # outside the slab: the grid is rebuilt for the whole cloud and the answer is still right
    off_o, idx_o = part.radius_search_self(r, 0, 500).export()
    off_g, idx_g = full.radius_search_self(r, 0, 500).export()
    assert np.array_equal(off_o, off_g) and np.array_equal(idx_o, idx_g)
    empty = eng.cloud(p, nr)
    assert empty.build_grid(r, block=(100, 100)) == (0, 0)


def test_config1_plumbing_case_through_a_ply_file(eng, tmp_path):
    """BASELINE config 1 end to end on the device: write the stand-in cloud as binary PLY, get_data (k = 30 normals
    re-oriented by the stored ones), 500 random keypoints, FPFH -- against the reference's outputs."""
    from conftest import config1_cloud
    from shot_fpfh_amd import compute_fpfh_descriptor, compute_normals
    from shot_fpfh_amd.helpers import get_data, write_ply

    g = load_golden("config1_fpfh_500.npz")
    p, d = config1_cloud(int(g["n"]), int(g["seed"]))
    path = str(tmp_path / "config1.ply")
    write_ply(path, [p, d], ["x", "y", "z", "nx", "ny", "nz"])
    points, normals = get_data(path, k=30, normals_computation_callback=compute_normals)
    assert np.array_equal(points, p) and np.abs(normals[:200] - g["normals_head"]).max() < 1e-9
    f = compute_fpfh_descriptor(g["kp_idx"], points, normals, radius=float(g["radius"]), n_bins=5, verbose=False)
    assert close(f, g["fpfh"]).all() and np.abs(f - g["fpfh"]).max() < 1e-9


def test_pipeline_stage_methods_other_choices(eng):
    """RegistrationPipeline with the non-default choices: density keypoints, FPFH descriptors, threshold matching,
    RANSAC, point-to-point ICP, metrics; stage caching and the error paths of the dispatchers."""
    from shot_fpfh_amd import compute_normals
    from shot_fpfh_amd.core import RigidTransform
    from shot_fpfh_amd.pipeline import RegistrationPipeline

    import shot_fpfh_amd.matching.ransac as ransac_module

    ransac_module.rng = np.random.default_rng(seed=72)  # the draws below must not depend on which tests ran before
    g = load_golden("icp_3500.npz")
    scan, ref = g["scan"], g["ref"]
    scan_normals = compute_normals(scan, scan, k=20)
    pipe = RegistrationPipeline(scan=scan, scan_normals=scan_normals, ref=ref, ref_normals=g["ref_normals"])
    with pytest.raises(ValueError):
        pipe.select_keypoints("nope")
    pipe.select_keypoints("subsampling_with_density", neighborhood_size=0.06, min_n_neighbors=3)
    kept = pipe.scan_keypoints.copy()
    pipe.select_keypoints("random", proportion_picked=0.1)  # cached: a second call changes nothing ...
    assert np.array_equal(pipe.scan_keypoints, kept)
    pipe.select_keypoints("iterative", neighborhood_size=0.05, force_recompute=True)  # ... unless forced
    assert pipe.scan_keypoints.shape[0] > 50 and pipe.ref_keypoints.shape[0] > 50
    with pytest.raises(ValueError):
        pipe.compute_descriptors(0.2, descriptor_choice="nope")
    pipe.compute_descriptors(0.2, descriptor_choice="fpfh", fpfh_n_bins=4, disable_progress_bars=True, verbose=False)
    assert pipe.scan_descriptors.shape == (pipe.scan_keypoints.shape[0], 64)
    with pytest.raises(ValueError):
        pipe.find_descriptors_matches("nope", reject_threshold=0.8, threshold_multiplier=10)
    pipe.find_descriptors_matches("threshold", reject_threshold=0.8, threshold_multiplier=10, debug_mode=True)
    assert pipe.matches[0].shape == pipe.matches[1].shape and pipe.matches[0].shape[0] > 20
    tf, ratio = pipe.run_ransac(n_draws=1500, draw_size=4, max_inliers_distance=0.03, disable_progress_bar=True,
                                exact_transformation=RigidTransform(g["true_rotation"], g["true_translation"]))
    assert 0.0 < ratio <= 1.0
    with pytest.raises(ValueError):
        pipe.run_icp("nope", tf, d_max=0.05)
    tf_icp, rms, converged = pipe.run_icp("point_to_point", tf, d_max=0.05, voxel_size=0.04, max_iter=40, rms_threshold=1e-9)
    # a smooth height field lets point-to-point ICP slide a little: it must land near the generating motion and
    # must not be worse than the RANSAC start it was given
    from shot_fpfh_amd.icp import compute_point_to_point_error

    assert np.abs(tf_icp.rotation - g["true_rotation"]).max() < 5e-2 and np.abs(tf_icp.translation - g["true_translation"]).max() < 5e-2
    assert compute_point_to_point_error(scan, ref, tf_icp)[0] <= compute_point_to_point_error(scan, ref, tf)[0] + 1e-12
    overlap, kp_ratio = pipe.compute_metrics_post_icp(tf_icp, 0.02)
    assert overlap > 0.9 and 0.0 <= kp_ratio <= 1.0
    # SHOT variants through the dispatcher (bi-scale needs the subsampled support, as in the reference)
    pipe.compute_descriptors(0.1, descriptor_choice="shot_bi_scale", phi=2.0, rho=10.0, min_neighborhood_size=10,
                             disable_progress_bars=True, verbose=False, force_recompute=True)
    assert pipe.scan_descriptors.shape[1] == 352
    pipe.compute_descriptors(0.1, descriptor_choice="shot_multiscale", phi=2.0, n_scales=2, min_neighborhood_size=10,
                             disable_progress_bars=True, verbose=False, force_recompute=True)
    assert pipe.scan_descriptors.shape[1] == 704


def test_config_c3_full_size_properties(eng, O):
    """BASELINE config 3 at its full size (1M points, every point a keypoint, r = 0.03, FPFH + SHOT), checked
    through properties that do not need a CPU run of the whole cloud:
      * a 1/50 shard computed through the block grid build equals the same rows of the unsharded run, bit for bit;
      * every SHOT row has unit norm (all neighbourhoods exceed min_neighborhood_size here);
      * a sample of SHOT rows of that shard equals the oracle's (which searches the full 1M cloud for them);
      * the neighbour relation is symmetric: sum_i sum_{j in N(i)} j == sum_i i |N(i)| over the shard's halo-free
        interior is replaced by the stronger per-list check on a sample: j in N(i) => i in N(j)."""
    from shot_fpfh_amd.sharding import DescriptorJob

    n, r = 1_000_000, 0.03
    p, nr, rng = synth_cloud(n, 3)
    full = DescriptorJob(eng, p, nr, r, n_bins=5, normalize=True, min_neighborhood_size=10)
    full.step()
    world, rank = 50, 17
    part = DescriptorJob(eng, p, nr, r, n_bins=5, normalize=True, min_neighborhood_size=10, world=world, rank=rank, spfh_exchange="halo")
    part.step()
    b, e = part.plan.block()
    assert e - b == n // world
    shot_part, fpfh_part = part.shot_out.to_host(), part.fpfh_out.to_host()
    assert np.array_equal(shot_part, full.shot_out.rows_to_host(b, e - b))
    assert np.array_equal(fpfh_part, full.fpfh_out.rows_to_host(b, e - b))
    assert np.abs(np.linalg.norm(shot_part, axis=1) - 1.0).max() < 1e-12
    assert np.isfinite(fpfh_part).all() and (fpfh_part >= 0).all() and fpfh_part.sum(axis=1).min() > 0
    rows = part.block_original_indices()
    assert np.array_equal(rows, full.block_original_indices()[b:e])
    pick = rng.choice(e - b, 300, replace=False)
    shot_o = O.shot_single_scale(p, nr, p[rows[pick]], r, normalize=True, min_neighborhood_size=10)
    assert close(shot_part[pick], shot_o).all() and np.abs(shot_part[pick] - shot_o).max() < 1e-9
    # symmetry of the radius search at full size, on lists exported for a sample of the shard
    cloud = eng.cloud(p)
    q_idx = rows[pick[:60]]
    off, idx = cloud.radius_search(p[q_idx], r).export()
    nbr = np.unique(idx)
    off2, idx2 = cloud.radius_search(p[nbr], r).export()
    back = {int(j): set(idx2[off2[t]:off2[t + 1]].tolist()) for t, j in enumerate(nbr)}
    for t, i in enumerate(q_idx):
        assert all(int(i) in back[int(j)] for j in idx[off[t]:off[t + 1]])
    part.close()
    full.close()


def test_clustered_cloud_overflows_the_optimistic_slots(eng, O):
    """A cloud whose density varies by two orders of magnitude: the mean-density slot capacity of the single-pass
    search is exceeded inside the cluster, so the exact count -> scan -> fill scheme runs; lists, normals, SHOT
    and FPFH must still be the oracle's (K5 / K6 / K7 take their streaming variants for the long lists)."""
    from shot_fpfh_amd import ShotMultiprocessor, compute_fpfh_descriptor

    rng = np.random.default_rng(81)
    sparse = rng.random((6000, 3), dtype=np.float32).astype(np.float64)
    dense = (0.5 + 0.02 * rng.standard_normal((3000, 3))).astype(np.float32).astype(np.float64)
    p = np.vstack([sparse, dense])
    nr = rng.standard_normal(p.shape)
    nr /= np.linalg.norm(nr, axis=1)[:, None]
    r = 0.06
    cloud = eng.cloud(p)
    q = np.vstack([p[rng.choice(9000, 300, replace=False)], p[6000:6050]])
    nb = cloud.radius_search(q, r)
    off, idx = nb.export()
    off_o, idx_o = O.radius_search(p, q, r)
    assert np.array_equal(off, off_o) and np.array_equal(idx, idx_o)
    counts = np.diff(off)
    assert counts.max() > 1500 and np.median(counts) < 50  # slots sized for the mean cannot hold the cluster's lists
    kp = np.concatenate([rng.choice(6000, 80, replace=False), 6000 + rng.choice(3000, 40, replace=False)])
    with ShotMultiprocessor(normalize=True, min_neighborhood_size=5, verbose=False, disable_progress_bar=True) as sm:
        shot = sm.compute_descriptor_single_scale(point_cloud=p, keypoints=p[kp], normals=nr, radius=r)
    shot_o = O.shot_single_scale(p, nr, p[kp], r, normalize=True, min_neighborhood_size=5)
    assert close(shot, shot_o).all()
    f = compute_fpfh_descriptor(kp, p, nr, radius=r, n_bins=5, verbose=False)
    f_o = O.compute_fpfh_descriptor(kp, p, nr, r, 5)
    assert close(f, f_o).all()


def test_rccl_communicator_single_rank(eng):
    """The exchange layer on the one GPU a test box has: RCCL initialises a 1-rank communicator through the C ABI,
    the in-place all-gather of a 1-rank job leaves the buffer as it is, and a second init on the same context is
    refused.  (The N-rank data path is covered on CPU by tests/test_multi_gpu_gloo.py.)"""
    import shot_fpfh_amd as s

    e2 = s.Engine(0)
    uid = e2.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    e2.comm_init(uid, 1, 0)
    a = e2.empty((4, 8)).from_host(np.arange(32.0).reshape(4, 8))
    e2.allgather(a, a.nbytes)
    assert np.array_equal(a.to_host(), np.arange(32.0).reshape(4, 8))
    with pytest.raises(s.ShotFpfhError):
        e2.comm_init(uid, 1, 0)
    a.free()


@pytest.mark.parametrize("n,r,kmin", [(6000, 0.105, 192), (4000, 0.12, 128), (3000, 0.1, 64)])
def test_fpfh_matrix_core_path_with_peaked_histograms(eng, O, n, r, kmin):
    """The uint8 SPFH table / int8 matrix-core K7 at its limits: neighbourhoods of 130-250 points (three and four
    64-neighbour steps) on a nearly flat patch with nearly parallel normals, so that single bins collect more than
    127 pairs -- the stored byte count ^ 128 is then positive as int8 and the padding-bin bias correction has to
    hold.  FPFH and the exported SPFH against the oracle."""
    from shot_fpfh_amd import compute_fpfh_descriptor

    rng = np.random.default_rng(91)
    xy = rng.random((n, 2), dtype=np.float32).astype(np.float64)
    p = np.column_stack([xy, (0.002 * rng.standard_normal(n)).astype(np.float32).astype(np.float64)])
    nr = np.column_stack([0.02 * rng.standard_normal((n, 2)), np.ones(n)])
    nr /= np.linalg.norm(nr, axis=1)[:, None]
    cloud = eng.cloud(p, nr)
    nb = cloud.radius_search_self(r)
    assert kmin < nb.max_count <= 255  # two, three and four 64-neighbour steps
    sp = eng.spfh(cloud, 5, nb.max_count)
    sp.compute(nb)
    spfh = sp.export()
    kp = np.sort(rng.choice(n, 400, replace=False))
    f_o, spfh_o = O.compute_fpfh_descriptor(kp, p, nr, r, 5, return_spfh=True)
    assert np.array_equal(spfh, spfh_o)  # integer counts / k: bit-exact
    peak = (spfh * nb.counts()[np.argsort(cloud.perm())][:, None]).max()
    assert peak > 127.5 or kmin < 128  # some bin really holds > 127 pairs
    f = compute_fpfh_descriptor(kp, p, nr, radius=r, n_bins=5, verbose=False)
    assert close(f, f_o).all() and np.abs(f - f_o).max() < 1e-9 * max(1.0, np.abs(f_o).max())


@pytest.mark.parametrize("prefilter", ["int8", "fp16"])
def test_match_large_permutation_property(eng, O, monkeypatch, prefilter):
    """BASELINE config 4's matching step at a size the CPU cannot check pair by pair (400k x 400k x 352: the integer
    pre-filter by default since round 6, the FP16 pre-filter with SF_MATCH_I8=0), through properties: the reference set is a row permutation of the scan set plus noise far
    below the spacing of the descriptors, so the arg-min must invert the permutation exactly; the distances must be the
    sequential float64 ones of those pairs; and a sample of rows must equal the exact kernel / oracle bit for bit."""
    rng = np.random.default_rng(404)
    m, d = 400_000, 352
    a = rng.random((m, d), dtype=np.float32).astype(np.float64)
    a *= rng.random((m, d), dtype=np.float32) < 0.3  # SHOT-like sparsity
    a /= np.maximum(np.linalg.norm(a, axis=1)[:, None], 1e-300)
    perm = rng.permutation(m)
    b = a[perm] + 1e-6 * rng.standard_normal((m, d)).astype(np.float32)
    inv = np.empty(m, dtype=np.int64)
    inv[perm] = np.arange(m)
    da, db = eng.empty((m, d)).from_host(a), eng.empty((m, d)).from_host(b)
    idx, dist = eng.empty((m,), np.int64), eng.empty((m,), np.float64)
    if prefilter == "fp16":
        monkeypatch.setenv("SF_MATCH_I8", "0")
    eng.profile_reset()
    eng.profile(True)
    eng.match_argmin_device(da, db, idx, dist)
    eng.sync()
    eng.profile(False)
    rep = eng.profile_report()
    if prefilter == "int8":
        assert rep.get("k8_match_i8", (0, 0))[0] >= 1, "this size was expected to take the integer pre-filter"
    else:
        assert rep.get("k8_match_half", (0, 0))[0] >= 1 and "k8_match_i8" not in rep, "this size was expected to take the FP16 pre-filter"
    i_h, d_h = idx.to_host(), dist.to_host()
    assert np.array_equal(i_h, inv)
    pick = rng.choice(m, 2000, replace=False)
    diff = a[pick] - b[inv[pick]]
    seq = np.zeros(pick.shape[0])
    for t in range(d):  # scipy's left-to-right sum
        seq += diff[:, t] * diff[:, t]
    assert np.array_equal(d_h[pick], np.sqrt(seq))
    io, do = O.match_argmin(a[pick[:40]], b)
    assert np.array_equal(io, i_h[pick[:40]]) and np.array_equal(do, d_h[pick[:40]])
    for x in (da, db, idx, dist):
        x.free()


def test_shared_sweep_frames_equal_the_two_kernel_form(eng, O):
    """With both descriptors wanted, K6 also accumulates the SHOT frame moments and K4 is reduced to its eigen-solves.
    The frames and descriptors must equal the separate K4 + K5 ones up to the rounding of a different summation
    order (and the oracle within the usual tolerance); FPFH must not change at all.  Also through a block view."""
    from shot_fpfh_amd.sharding import DescriptorJob

    p, nr, rng = synth_cloud(30000, 61)
    r = 0.07
    runs = {}
    for share in (True, False):
        job = DescriptorJob(eng, p, nr, r, n_bins=5, normalize=True, min_neighborhood_size=10, share_sweep=share)
        job.step()
        runs[share] = (job.shot_out.to_host(), job.lrf_out.to_host(), job.fpfh_out.to_host(), job.block_original_indices())
        job.close()
    (s1, l1, f1, rows), (s0, l0, f0, _) = runs[True], runs[False]
    assert np.array_equal(f1, f0)
    assert np.abs(l1 - l0).max() < 1e-11 and np.abs(s1 - s0).max() < 1e-10
    pick = rng.choice(30000, 200, replace=False)
    so = O.shot_single_scale(p, nr, p[rows[pick]], r, normalize=True, min_neighborhood_size=10)
    assert close(s1[pick], so).all()
    # a shard (block view of the lists, moments offset by the halo) gives the same rows as the full run, bit for bit
    part = DescriptorJob(eng, p, nr, r, n_bins=5, normalize=True, min_neighborhood_size=10, world=3, rank=1, spfh_exchange="halo")
    part.step()
    b, e = part.plan.block()
    assert np.array_equal(part.shot_out.to_host(), s1[b:e]) and np.array_equal(part.fpfh_out.to_host(), f1[b:e])
    part.close()
