"""GPU parity tests added in round 2 (all through the C ABI):

  * a8  serial compute_shot_descriptor against the reference's golden and the oracle;
  * a4  get_azimuth_idx on the device against the reference's boundary table; a SHOT row whose neighbours all sit ON bin
        boundaries;
  * C3  FPFH at full size (1M points, k ~ 110) against the oracle on a sample;  C5: one rank's block of the 8M-point
        cloud (r = 0.015) on one device against the oracle on a sample;
  * C4  at full size: 2 x 1M SHOT -> basic_matching -> RANSAC, with a sample of match rows re-derived on the CPU;
  * the exchange layer: ncclAllGather really executed (one-rank communicator), SubsetMatchJob on the device;
  * regressions for the advisor's findings (k-NN queries far outside the cloud, two-stream overlap with unshared sweeps).
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


# ---- a8 --------------------------------------------------------------------------------------------------------------
def test_serial_shot_descriptor_golden(O):
    """compute_shot_descriptor (shot.py:310-499): frame WITHOUT the keypoint in its support, always normalised."""
    from shot_fpfh_amd.descriptors.shot import compute_shot_descriptor

    g = load_golden("shot_150.npz")
    kp, r = g["keypoints"][:60], float(g["radius"])
    d = compute_shot_descriptor(kp, g["cloud"], g["normals"], r, min_neighborhood_size=10)
    assert d.shape == g["serial"].shape and d.dtype == np.float64
    assert close(d, g["serial"]).all(), np.abs(d - g["serial"]).max()
    assert np.abs(d - O.compute_shot_descriptor(kp, g["cloud"], g["normals"], r, 10)).max() < 1e-9
    with pytest.raises(AssertionError):
        compute_shot_descriptor(kp, g["cloud"], g["normals"], r, n_azimuth_bins=4)


def test_serial_shot_descriptor_vs_oracle_with_duplicates_and_sparse_rows(O):
    """Duplicated keypoints (several zero-distance neighbours leave the support), keypoints off the cloud (the keypoint
    is not a cloud point: nothing to drop), sparse corners (gate fails -> zero row) and min_neighborhood_size extremes."""
    from shot_fpfh_amd.descriptors.shot import compute_shot_descriptor

    p, nr, rng = synth_cloud(5000, 61)
    p = np.vstack([p, p[:40], p[:15]])  # duplicates (and triplicates) of cloud points
    nr = np.vstack([nr, nr[:40], nr[:15]])
    kp = np.vstack([p[:80], p[2000:2100], rng.random((50, 3)), [[5.0, 5.0, 5.0]]])
    for mn in (10, 60, 0):
        d = compute_shot_descriptor(kp, p, nr, 0.13, min_neighborhood_size=mn)
        do = O.compute_shot_descriptor(kp, p, nr, 0.13, mn)
        assert np.array_equal(d.any(axis=1), do.any(axis=1))
        assert np.abs(d - do).max() < 1e-9
    assert not d[-1].any()  # the far keypoint has an empty neighbourhood


# ---- a4 --------------------------------------------------------------------------------------------------------------
def test_azimuth_idx_device_function_matches_the_reference_table():
    from shot_fpfh_amd.descriptors.shot import get_azimuth_idx

    g = load_golden("azimuth_table.npz")
    got = get_azimuth_idx(g["x"], g["y"])
    assert got.dtype == np.int64 and np.array_equal(got, g["idx"])
    named = {(1, 0): 3, (0, 1): 5, (-1, 0): 7, (0, -1): 1, (1, 1): 4, (-1, 1): 6, (-1, -1): 0, (1, -1): 2, (0, 0): 0}
    xy = np.array(list(named), dtype=np.float64)
    assert get_azimuth_idx(xy[:, 0], xy[:, 1]).tolist() == list(named.values())


def test_shot_row_with_every_neighbour_on_a_bin_boundary(eng):
    """One keypoint at the origin, identity frame, neighbours exactly ON octant / elevation / radial boundaries and
    cosines on the half-way points of the cosine bins (reference output: tests/golden/shot_boundary.npz)."""
    g = load_golden("shot_boundary.npz")
    cloud = eng.cloud(g["neighbors"], g["normals"])
    nb = cloud.radius_search(g["point"][None, :], float(g["radius"]))
    assert nb.total == g["neighbors"].shape[0]
    for key, normalize in (("desc_n1", True), ("desc_n0", False)):
        d = nb.shot(np.eye(3)[None], normalize, 5)[0]
        assert np.abs(d - g[key]).max() < 1e-9, np.abs(d - g[key]).max()
    nb.free()
    cloud.free()


# ---- C2 tightened: DESIGN.md claims 0 rows outside tolerance; hold it to that -------------------------------------------
def test_config_c2_has_no_row_outside_tolerance(O):
    import shot_fpfh_amd as s
    from shot_fpfh_amd.descriptors import ShotMultiprocessor

    p, nr, rng = synth_cloud(100000, 2)
    kp = np.sort(rng.choice(100000, 10000, replace=False))
    with ShotMultiprocessor(normalize=True, min_neighborhood_size=10, verbose=False) as sm:
        d = sm.compute_descriptor_single_scale(p, nr, p[kp], 0.05)
    do = O.shot_single_scale(p, nr, p[kp], 0.05, True, 10)
    bad = np.flatnonzero((~close(d, do)).any(axis=1))
    assert bad.size == 0, f"SHOT rows outside tolerance: {bad.tolist()}"
    assert np.abs(d - do).max() < 1e-9
    # no row is excluded from the comparison above.  (Rows with two neighbours at EXACTLY the same distance are undefined in
    # the reference -- unstable argsort, shot.py:218 -- and would have to be: counted here, expected and asserted 0 on this
    # float32-grid cloud; printed with -s.)
    off, _, dist = O.radius_search(p, p[kp], 0.05, return_distance=True)
    tied = sum(len(np.unique(dist[off[i]:off[i + 1]])) < off[i + 1] - off[i] for i in range(kp.size))
    print(f"config C2: {tied} of {kp.size} keypoints have tied rho (rows excluded from comparison: 0)")
    assert tied == 0
    f = s.compute_fpfh_descriptor(kp, p, nr, 0.05, 5, verbose=False)
    fo = O.compute_fpfh_descriptor(kp, p, nr, 0.05, 5)
    assert np.abs(f - fo).max() < 1e-9


# ---- C3 / C5 at full size against the oracle ------------------------------------------------------------------------------
def test_config_c3_fpfh_and_shot_rows_vs_oracle_at_full_size(eng, O):
    """BASELINE config 3 (1M points, all keypoints, r = 0.03, k ~ 110): 300 FPFH rows and 300 SHOT rows of the
    resident outputs against the oracle, which evaluates SPFH only for the sample's neighbours (bit-identical rows)."""
    from shot_fpfh_amd.sharding import DescriptorJob

    n, r = 1_000_000, 0.03
    p, nr, rng = synth_cloud(n, 3)
    job = DescriptorJob(eng, p, nr, r, n_bins=5, normalize=True, min_neighborhood_size=10)
    job.step()
    assert 100 < job.last_pairs / n < 120
    orig = job.block_original_indices()
    pick = np.sort(rng.choice(n, 300, replace=False))
    f = np.stack([job.fpfh_out.rows_to_host(int(i), 1)[0] for i in pick])
    fo = O.compute_fpfh_descriptor_sample(orig[pick], p, nr, r, 5)
    assert close(f, fo).all() and np.abs(f - fo).max() < 1e-9, np.abs(f - fo).max()
    d = np.stack([job.shot_out.rows_to_host(int(i), 1)[0] for i in pick])
    do = O.shot_single_scale(p, nr, p[orig[pick]], r, True, 10)
    assert close(d, do).all() and np.abs(d - do).max() < 1e-9
    off, _, dist = O.radius_search(p, p[orig[pick]], r, return_distance=True)
    tied = sum(len(np.unique(dist[off[i]:off[i + 1]])) < off[i + 1] - off[i] for i in range(pick.size))
    print(f"config C3: {tied} of {pick.size} sampled keypoints have tied rho (rows excluded from comparison: 0)")
    assert tied == 0
    # the drop-in call returns the same rows (original numbering) as the resident job
    import shot_fpfh_amd as s

    fd = s.compute_fpfh_descriptor(orig[pick], p, nr, r, 5, verbose=False)
    assert np.array_equal(fd, f)
    job.close()


# (config C5 at full size: tests/test_hip_round3.py, ranks 0, 3 and 7, both SPFH exchange modes)


# ---- C4 at full size ---------------------------------------------------------------------------------------------------------
def _exact_first_argmin(O, a_rows, b, chunk=65536):
    """Row arg-min of cdist(a_rows, b) with scipy's first-minimum rule, affordable for a few hundred rows against 10^6:
    float64 BLAS keys ||b||^2 - 2 a.b locate every column within 1e-7 of the minimum (the GEMM's rounding is ~1e-13),
    the oracle's sequential sum then decides among those candidates in index order."""
    bn = np.einsum("ij,ij->i", b, b)
    best = np.full(a_rows.shape[0], np.inf)
    keys = []
    for c0 in range(0, b.shape[0], chunk):
        k = bn[None, c0:c0 + chunk] - 2.0 * (a_rows @ b[c0:c0 + chunk].T)
        best = np.minimum(best, k.min(axis=1))
        keys.append(k)
    keys = np.concatenate(keys, axis=1)
    idx, dist = np.zeros(a_rows.shape[0], np.int64), np.zeros(a_rows.shape[0])
    for i in range(a_rows.shape[0]):
        cand = np.flatnonzero(keys[i] <= best[i] + 1e-7)
        j, dd = O.match_argmin(a_rows[i:i + 1], b[cand])
        idx[i], dist[i] = cand[j[0]], dd[0]
    return idx, dist


def test_config_c4_full_size_chain_with_cpu_checked_match_rows(eng, O):
    """BASELINE config 4: two 1M-point clouds (ref = scan[perm] R^T + t), SHOT r = 0.03 on both, basic_matching
    scan -> ref, RANSAC with 10 000 draws at threshold 0.01.
      * a sample of 256 match rows equals the exact first-minimum arg-min over all 10^6 reference rows (and its distance),
        re-derived on the CPU;
      * the chain recovers R, t; the share of matches equal to the true correspondence is reported against the measured
        level (SHOT rows are only rotation-invariant up to the sign votes of get_local_rf, shot.py:40-45)."""
    import shot_fpfh_amd.matching.ransac as R
    from scipy.spatial.transform import Rotation
    from shot_fpfh_amd.descriptors import ShotMultiprocessor
    from shot_fpfh_amd.matching import basic_matching

    n, r = 1_000_000, 0.03
    scan, nrm, rng = synth_cloud(n, 4)
    perm = rng.permutation(n)
    rot = Rotation.from_euler("xyz", [0.3, -0.2, 0.5]).as_matrix()
    t = np.array([0.1, -0.3, 0.2])
    ref, ref_n = scan[perm] @ rot.T + t, nrm[perm] @ rot.T
    with ShotMultiprocessor(normalize=True, min_neighborhood_size=10, verbose=False) as sm:
        ds = sm.compute_descriptor_single_scale(scan, nrm, scan, r)
        dr = sm.compute_descriptor_single_scale(ref, ref_n, ref, r)
    si, ri = basic_matching(ds, dr)
    assert si.dtype == np.int64 and ri.dtype == np.int64
    assert np.array_equal(si, np.flatnonzero(ds.any(axis=1)))
    pick = np.sort(rng.choice(si.size, 256, replace=False))
    nz_r = np.flatnonzero(dr.any(axis=1))
    want, _ = _exact_first_argmin(O, ds[si[pick]], dr[nz_r])
    assert np.array_equal(ri[pick], nz_r[want])
    inv = np.empty(n, np.int64)
    inv[perm] = np.arange(n)
    correct = float((ri == inv[si]).mean())
    # the seeded, measured level (profiles/r02_config4.txt: 92.5 %; the REFERENCE itself recovers 93.9 % of a rigidly moved
    # 6 000-point copy, tests/golden/c4_pair_6k.npz: frames whose sign votes tie keep LAPACK's sign, shot.py:40-45)
    assert abs(correct - 0.925) <= 0.005, correct
    R.rng = np.random.default_rng(seed=72)
    ratio, tf = R.ransac_on_matches(si, ri, scan, ref, n_draws=10000, draw_size=4, distance_threshold=0.01,
                                    disable_progress_bar=True)
    assert abs(ratio - correct) < 0.02  # inliers are the correct matches (wrong ones land far away)
    # ransac_on_matches keeps the FIRST draw that reaches the best inlier count and does not refit (ransac.py:60-72): a draw
    # of three true matches and one near miss moves every true match by less than the 0.01 threshold, scores as many
    # inliers as an exact draw and wins if it comes first -- so the motion is recovered to the threshold's scale (1e-3
    # here; 5e-16 with the bench's 2 000-draw sequence, where an all-true draw happens to come first), not to rounding
    assert np.abs(tf.rotation - rot).max() < 5e-3 and np.abs(tf.translation - t).max() < 5e-3
    assert ratio >= correct - 1e-12  # every true match is an inlier of the winning transform
    # ... and, exactly: replay the seeded draw stream on the host -- the transform returned is, to the last bit, the Kabsch fit
    # of ONE of the seeded draws, that draw's NumPy inlier count (ransac.py:60-67) is the count the device reported (up to the
    # matches within an ulp of the threshold, where BLAS's fused `a @ R.T` may round differently from the reference's order),
    # and no earlier draw beats it
    from shot_fpfh_amd.core import solver_point_to_point

    replay = np.random.default_rng(seed=72)
    sp, rp = scan[si], ref[ri]
    best_count = int(round(ratio * si.size))
    winner, earlier_best = None, 0
    for d in range(10000):
        pick4 = replay.choice(si.size, 4, replace=False, shuffle=False)
        cand = solver_point_to_point(sp[pick4], rp[pick4])
        if np.abs(cand.rotation - rot).max() > 0.05:  # (0.05 off the true rotation: the true matches move by several thresholds)
            continue
        cnt_d = int((np.linalg.norm((sp @ cand.rotation.T + cand.translation) - rp, axis=1) <= 0.01).sum())
        cand.normalize_rotation()
        if np.array_equal(cand.rotation, tf.rotation) and np.array_equal(cand.translation, tf.translation):
            winner = (d, cnt_d)
            break
        earlier_best = max(earlier_best, cnt_d)
    assert winner is not None, "the returned transform is not the fit of any seeded draw"
    assert abs(winner[1] - best_count) <= 2 and earlier_best <= winner[1] + 2, (winner, best_count, earlier_best)
    # refitting on the winning transform's inliers (this test only; the reference does not refit) recovers the motion to rounding
    # (three passes: the 0.01 inliers still hold a few WRONG matches -- a wrong match often lands on a near neighbour of the
    # right point, 0.006 away on average -- which bias the first fit at the 1e-5 level; at 1e-3 around that fit only true
    # matches remain)
    inl = np.linalg.norm((sp @ tf.rotation.T + tf.translation) - rp, axis=1) <= 0.01
    refit = solver_point_to_point(sp[inl], rp[inl])
    for thr in (1e-3, 1e-6):  # (at 1e-3 a handful of wrong matches that land within a millimetre of the right point still bias the
        inl = np.linalg.norm((sp @ refit.rotation.T + refit.translation) - rp, axis=1) <= thr  # fit at the 1e-8 level)
        assert inl.mean() > 0.9
        refit = solver_point_to_point(sp[inl], rp[inl])
    assert np.abs(refit.rotation - rot).max() < 1e-9 and np.abs(refit.translation - t).max() < 1e-9
    # the winning draw's inlier count equals the NumPy expression of ransac.py:60-67 for that transform
    best_inl = (np.linalg.norm((scan[si] @ tf.rotation.T + tf.translation) - ref[ri], axis=1) <= 0.01).sum()
    assert abs(best_inl / si.size - ratio) < 1e-3


# ---- exchange layer -----------------------------------------------------------------------------------------------------
def test_nccl_allgather_is_really_executed_on_a_one_rank_communicator(O):
    """Once sf_comm_init has run, sf_comm_allgather goes through ncclAllGather -- also with ONE rank, which is what a
    single-GPU box can execute of the N-rank exchange: in-place form (send inside recv), out-of-place form through
    MatchJob / SubsetMatchJob, and the HIP-event timer 'c_allgather' counts the launches."""
    import shot_fpfh_amd as s
    from shot_fpfh_amd.sharding import SubsetMatchJob

    e2 = s.Engine(0)
    e2.comm_init(e2.comm_unique_id(), 1, 0)
    e2.profile_reset()
    e2.profile(True)
    a = e2.empty((1000, 352)).from_host(np.arange(352000.0).reshape(1000, 352))
    e2.allgather(a, a.nbytes)
    assert np.array_equal(a.to_host(), np.arange(352000.0).reshape(1000, 352))
    rng = np.random.default_rng(3)
    scan = rng.random((900, 352)) * (rng.random((900, 352)) < 0.3)
    perm = rng.permutation(900)
    ref = scan[perm] + 1e-3 * rng.standard_normal((900, 352))
    scan[[5, 6]] = 0.0
    sub = SubsetMatchJob(e2, 352, 512, 1, 0)
    s_sel = np.flatnonzero(np.arange(900) % 2 == 0)
    r_sel = np.flatnonzero(perm % 2 == 0)
    sub.select(e2.empty((900, 352)).from_host(scan), s_sel, s_sel, e2.empty((900, 352)).from_host(ref), r_sel, perm[r_sel])
    sub.run()
    s_lab, r_lab = sub.matches()
    si, ri = O.basic_matching(scan[s_sel], ref[r_sel])
    assert np.array_equal(s_lab, s_sel[si]) and np.array_equal(r_lab, perm[r_sel][ri])
    assert (s_lab == r_lab).mean() > 0.99
    e2.profile(False)
    rep = e2.profile_report()
    assert rep["c_allgather"][0] >= 3 and rep["c_allgather"][1] > 0.0  # rows, rows, labels: ncclAllGather launches
    sub.close()
    a.free()
    e2.close()


# ---- advisor regressions -----------------------------------------------------------------------------------------------------
def test_knn_answers_queries_far_outside_the_cloud(eng, O):
    """KDTree.query returns the nearest points for ANY query; ICP feeds it scans that are not aligned yet.  Queries 10
    and 1000 bounding-box diagonals away (and a mix with inside queries) against brute force."""
    p, _, rng = synth_cloud(3000, 12)
    inside = rng.random((50, 3))
    far = np.array([[11.0, 12.0, -9.0], [1000.0, 0.5, 0.5], [-1e3, -1e3, -1e3], [0.5, 0.5, 40.0]])
    q = np.vstack([inside, far, inside + 25.0])
    cloud = eng.cloud(p)
    for k in (1, 7):
        nb = cloud.knn_search(q, k)
        off, idx = nb.export()
        d2 = ((p[None, :, :] - q[:, None, :]) ** 2).sum(axis=2)
        want = np.sort(np.argsort(d2, axis=1, kind="stable")[:, :k], axis=1)
        got = np.sort(idx.reshape(q.shape[0], k), axis=1)
        assert np.array_equal(got, want)
        nb.free()
    cloud.free()


@pytest.mark.parametrize("share", [False, True])
def test_two_stream_overlap_with_unshared_sweep_and_long_lists(eng, share):
    """The --overlap path when K6 does NOT run before the fork (share_sweep off, or a list longer than 256 points):
    the normals are then sorted by whichever chain gets there first and the other stream must wait for that gather."""
    from shot_fpfh_amd.sharding import DescriptorJob

    rng = np.random.default_rng(8)
    sparse = rng.random((20000, 3), dtype=np.float32).astype(np.float64)
    dense = (0.5 + 0.01 * rng.standard_normal((600, 3))).astype(np.float32).astype(np.float64)  # lists > 256 points
    p = np.vstack([sparse, dense]) if not share else sparse
    nr = rng.standard_normal(p.shape)
    nr /= np.linalg.norm(nr, axis=1)[:, None]
    outs = []
    for overlap in (False, True):
        job = DescriptorJob(eng, p, nr, 0.06, overlap_chains=overlap, share_sweep=share)
        for _ in range(3):  # every step rebuilds the grid, which clears the sorted-normals flag
            job.step()
        outs.append((job.fpfh_out.to_host(), job.shot_out.to_host(), job.lrf_out.to_host()))
        job.close()
    for a, b in zip(*outs):
        assert np.array_equal(a, b)


# ---- voxel subsampling on the device -------------------------------------------------------------------
def test_grid_subsampling_device_matches_reference_goldens():
    """grid_subsampling / select_keypoints_subsampling / the voxel-count density threshold against the reference's
    outputs (20k uniform cloud at voxel 0.05: 5 694 multi-point voxels, 2 042 exact two-point ties; the SHOT supports)."""
    import shot_fpfh_amd.keypoint_selection as ks
    from shot_fpfh_amd.core import grid_subsampling, voxel_closest_to_barycentre

    g = load_golden("grid_sub_20k.npz")
    p, _, _ = synth_cloud(int(g["n"]), int(g["seed"]))
    got = grid_subsampling(p, float(g["voxel"]))
    assert got.dtype == np.int64 and np.array_equal(got, g["idx"])
    s = load_golden("shot_150.npz")
    assert np.array_equal(grid_subsampling(s["cloud"], float(s["voxel"])), s["support"])
    assert np.array_equal(grid_subsampling(s["cloud"], 0.008), s["support_008"])
    k = load_golden("keypoints_6k.npz")
    assert np.array_equal(ks.select_keypoints_subsampling(k["cloud"], float(k["voxel"])), k["subsampling"])
    assert np.array_equal(ks.select_keypoints_with_density_threshold(k["cloud"], float(k["voxel"]), int(k["density_value"])),
                          k["density_voxel"])
    # the platform-independent order: same voxels, same populations; a different representative only where the
    # reference's own choice hangs on its unstable sort (equal distances to the barycentre, or last-bit barycentres)
    pick_i, cnt_i = voxel_closest_to_barycentre(p, float(g["voxel"]), within_voxel_order="index")
    pick_n, cnt_n = voxel_closest_to_barycentre(p, float(g["voxel"]))
    assert np.array_equal(cnt_i, cnt_n) and pick_i.shape == pick_n.shape
    keys = lambda idx: ((p[idx] - p.min(axis=0)) // float(g["voxel"])).astype(int)
    assert np.array_equal(keys(pick_i), keys(pick_n))
    diff = np.flatnonzero(pick_i != pick_n)
    assert 0 < diff.size < cnt_n.size
    # every differing voxel is a tie within rounding: both candidates are equally close to the barycentre
    allk = keys(np.arange(p.shape[0]))
    for v in diff[:100]:
        members = np.flatnonzero((allk == allk[pick_n[v]]).all(axis=1))
        bary = p[members].mean(axis=0)
        da, db = np.linalg.norm(p[pick_i[v]] - bary), np.linalg.norm(p[pick_n[v]] - bary)
        assert abs(da - db) <= 1e-12 * max(da, 1e-300) + 1e-15, (v, da, db)
    with pytest.raises(Exception):
        grid_subsampling(p, 0.0)
    assert grid_subsampling(np.zeros((0, 3)), 0.1).size == 0
    one = grid_subsampling(p[:1], 0.1)
    assert one.tolist() == [0]


def test_grid_subsampling_device_at_one_million_points():
    """1M points, voxel = radius / 10 of config 3 (0.003): mostly one-point voxels; and a coarse voxel (0.02, ~8 points
    each).  Against the NumPy expression of subsampling.py:12-37 evaluated with the same visiting order."""
    from shot_fpfh_amd.core import voxel_closest_to_barycentre

    p, _, _ = synth_cloud(1_000_000, 3)
    for voxel in (0.003, 0.02):
        picked, counts = voxel_closest_to_barycentre(p, voxel)
        keys = ((p - np.min(p, axis=0)) // voxel).astype(int)
        _, inverse, cnt = np.unique(keys, axis=0, return_inverse=True, return_counts=True)
        inverse = np.asarray(inverse).reshape(-1)
        assert np.array_equal(counts, cnt)
        order = np.argsort(inverse)
        starts = np.concatenate(([0], np.cumsum(cnt)[:-1]))
        grouped = p[order]
        bary = np.add.reduceat(grouped, starts, axis=0) / cnt[:, None]
        seg = np.repeat(np.arange(cnt.shape[0]), cnt)
        dist = np.linalg.norm(grouped - bary[seg], axis=1)
        seg_min = np.minimum.reduceat(dist, starts)
        hit = np.flatnonzero(dist == seg_min[seg])
        first = hit[np.unique(seg[hit], return_index=True)[1]]
        assert np.array_equal(picked, order[first])


# ---- limits the reference does not have (VERDICT r1 "missing" 6) -----------------------------------------------------------
@pytest.mark.parametrize("nb", [9, 11, 16])
def test_fpfh_bin_counts_above_eight_take_the_generic_kernels(O, nb):
    """compute_fpfh_descriptor accepts any n_bins (fpfh.py:16): above 8 the 32-bit table and the generic K6 / K7 run.
    Uniform and surface clouds, keypoint subset and all points, SPFH bit-exact (integer counts / k)."""
    import shot_fpfh_amd as s
    from conftest import config1_cloud

    p, nr, rng = synth_cloud(2500, 90 + nb)
    kp = np.sort(rng.choice(2500, 300, replace=False))
    f, spfh = s.compute_fpfh_descriptor(kp, p, nr, 0.14, nb, verbose=False, return_spfh=True)
    fo, spfh_o = O.compute_fpfh_descriptor(kp, p, nr, 0.14, nb, return_spfh=True)
    assert f.shape == (300, nb**3) and np.array_equal(spfh, spfh_o)
    assert close(f, fo).all() and np.abs(f - fo).max() < 1e-9
    ps, ns = config1_cloud(3000, 7)
    fa = s.compute_fpfh_descriptor(np.arange(3000), ps, ns, 0.08, nb, verbose=False)
    assert np.abs(fa - O.compute_fpfh_descriptor(np.arange(3000), ps, ns, 0.08, nb)).max() < 1e-9


def test_fpfh_generic_bins_through_the_sharded_job(eng, O):
    """DescriptorJob with n_bins = 10 (no shared sweep, generic table) in two blocks == the oracle."""
    from shot_fpfh_amd.sharding import DescriptorJob

    p, nr, _ = synth_cloud(4000, 55)
    fo = O.compute_fpfh_descriptor(np.arange(4000), p, nr, 0.1, 10)
    got = np.full((4000, 1000), np.nan)
    for rank in range(2):
        job = DescriptorJob(eng, p, nr, 0.1, n_bins=10, min_neighborhood_size=5, world=2, rank=rank, spfh_exchange="halo")
        job.step()
        got[job.block_original_indices()] = job.fpfh_out.to_host()
        job.close()
    assert np.abs(got - fo).max() < 1e-9


@pytest.mark.parametrize("n,m,k", [(6000, 200, 500), (3000, 64, 1200), (2500, 40, 1984)])
def test_knn_with_more_than_448_neighbours(eng, n, m, k):
    p, _, rng = synth_cloud(n, k)
    q = np.vstack([rng.random((m - 2, 3)), [[2.0, 2.0, 2.0]], p[:1]])
    cloud = eng.cloud(p)
    nb = cloud.knn_search(q, k)
    _, idx = nb.export()
    d2 = ((p[None, :, :] - q[:, None, :]) ** 2).sum(axis=2)
    want = np.sort(np.argsort(d2, axis=1, kind="stable")[:, :k], axis=1)
    assert np.array_equal(np.sort(idx.reshape(m, k), axis=1), want)
    nb.free()
    cloud.free()


# ---- K7's sparse-block form and the packed copy of the SPFH table --------------------------------------------------
def test_fpfh_live_block_mask_grows_between_computes(eng, O):
    """One SPFH table filled by TWO computes whose rows have different non-empty 16-bin blocks: a plane with normals along
    its own (phi = 0 for every pair: one block of the 27-bin table) sorts before a blob (both blocks).  After the first
    compute the matrix-core K7 runs its one-block form on the packed copy; the second compute grows the table-wide mask,
    so the plane's rows have to be re-packed under the new pair of blocks.  Every stage against the oracle."""
    rng = np.random.default_rng(4)
    plane = np.column_stack([rng.random((6000, 2)), np.zeros(6000)])
    blob = rng.random((6000, 3)) * np.array([1.0, 1.0, 0.3]) + np.array([0.0, 0.0, 2.0])
    p = np.ascontiguousarray(np.vstack([plane, blob]).astype(np.float32).astype(np.float64))
    nr = np.vstack([np.tile([0.0, 0.0, 1.0], (6000, 1)), rng.standard_normal((6000, 3))])
    nr[6000:] /= np.linalg.norm(nr[6000:], axis=1)[:, None]
    r, nb = 0.06, 3
    want = O.compute_fpfh_descriptor(np.arange(12000), p, nr, r, nb)
    cloud = eng.cloud(p, nr)
    cloud.build_grid(r)
    perm = cloud.perm()
    assert set(perm[:6000].tolist()) == set(range(6000))  # z-major cell order: the plane's points come first
    full = cloud.radius_search_self(r)
    lo, hi = cloud.radius_search_self(r, 0, 6000), cloud.radius_search_self(r, 6000, 12000)
    sp = eng.spfh(cloud, nb, full.max_count)
    sp.compute(lo)
    got_lo = sp.fpfh(lo)
    assert np.abs(got_lo - want[perm[:6000]]).max() < 1e-9
    assert not got_lo[:, 16:].any() and got_lo[:, 12:15].any()  # only bins 12..14 (alpha, phi central): block 0
    sp.compute(hi)
    got = sp.fpfh(full)
    assert np.abs(got - want[perm]).max() < 1e-9
    assert got[6000:, 16:].any()  # the blob does reach the second block
    # ... and the same table computed in one go, and again (steady state), gives the same bits
    sp2 = eng.spfh(cloud, nb, full.max_count)
    for _ in range(2):
        sp2.compute(full)
        assert np.array_equal(sp2.fpfh(full), got)
    kp = rng.choice(12000, 500, replace=False)
    assert np.array_equal(sp.fpfh(full, kp), got[np.argsort(perm)][kp])


def test_spfh_alpha_bin_shortcut_only_with_short_normals(eng, O):
    """K6 skips alpha when |alpha| <= radius * max|n|^2 cannot reach an edge of its histogram.  Unit normals and a small
    radius take the shortcut; the same cloud with normals ten times too long (alpha up to 100 x larger: other bins, and
    samples beyond +-1 that the reference drops) must not -- both against the oracle, on the integer SPFH table itself."""
    import shot_fpfh_amd as s

    p, nr, rng = synth_cloud(20000, 123)
    kp = np.arange(20000)
    r = 0.05
    for scale in (1.0, 10.0, 0.5):
        nn = nr * scale
        got = s.compute_fpfh_descriptor(kp, p, nn, r, 5)
        want = O.compute_fpfh_descriptor(kp, p, nn, r, 5)
        assert np.abs(got - want).max() < 1e-9, scale
        live_alpha = sorted(set((np.flatnonzero(want.any(axis=0)) // 25).tolist()))
        assert (live_alpha == [2]) == (scale <= 1.0), (scale, live_alpha)
    # even bin count: 0 is an edge of the alpha histogram, no shortcut possible; radius above the inner edge neither
    for nb, rr in ((4, 0.05), (5, 0.25)):
        got = s.compute_fpfh_descriptor(kp[:3000], p, nr, rr, nb)
        assert np.abs(got - O.compute_fpfh_descriptor(kp[:3000], p, nr, rr, nb)).max() < 1e-9


def test_fpfh_sparse_block_form_equals_full_form_bit_for_bit(monkeypatch):
    """SF_FPFH_DENSE=1 marks every 16-bin block of a new SPFH table live, which forces K7's full form; without it tables
    with one or two live blocks (small radius, any bin count) take the sparse-block form on the packed copy.  Same bits."""
    import shot_fpfh_amd as s

    rng = np.random.default_rng(5)
    for n, r, scale in ((60000, 0.04, 1.0), (20000, 0.45, 1.0), (40000, 40.0, 1000.0)):
        p = rng.random((n, 3), dtype=np.float32).astype(np.float64) * scale
        nr = rng.standard_normal((n, 3))
        nr /= np.linalg.norm(nr, axis=1)[:, None]
        kp = np.sort(rng.choice(n, 5000, replace=False))
        for nb in (5, 4, 3, 2):
            monkeypatch.delenv("SF_FPFH_DENSE", raising=False)
            a = s.compute_fpfh_descriptor(kp, p, nr, r, nb)
            monkeypatch.setenv("SF_FPFH_DENSE", "1")
            b = s.compute_fpfh_descriptor(kp, p, nr, r, nb)
            assert np.array_equal(a, b), (n, r, nb)
    monkeypatch.delenv("SF_FPFH_DENSE", raising=False)
