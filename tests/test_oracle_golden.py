"""The CPU oracle against the golden vectors produced by the reference itself (tools/gen_golden.py).

This is what pins the oracle: every hot-path function of oracle/shot_fpfh_oracle.c must reproduce
the reference's outputs on the reference's inputs before it may judge the HIP kernels.
Runs without a GPU.
"""
import numpy as np
import pytest

from conftest import load_golden
from oracle import oracle as O

TOL = 1e-12  # oracle vs reference: same arithmetic up to summation order


def test_radius_search_matches_kdtree_bit_exact():
    g = load_golden("nbrs_2k.npz")
    off, idx, dist = O.radius_search(g["cloud"], g["cloud"], float(g["radius"]), return_distance=True)
    assert np.array_equal(off, g["offsets"])
    assert np.array_equal(idx, g["idx"])
    assert np.array_equal(dist, g["dist"])  # sqrt(((dx^2+dy^2)+dz^2)), bit for bit
    off2, idx2 = O.radius_search(g["cloud"], g["queries"], float(g["radius"]))
    assert np.array_equal(off2, g["q_offsets"]) and np.array_equal(idx2, g["q_idx"])
    offb, idxb = O.radius_search(g["cloud"], g["queries"], float(g["radius"]), brute=True)
    assert np.array_equal(offb, g["q_offsets"]) and np.array_equal(idxb, g["q_idx"])


def test_eigh3_matches_lapack_including_signs():
    rng = np.random.default_rng(5)
    for _ in range(3000):
        k = rng.integers(4, 80)
        c = rng.standard_normal((k, 3)) * (rng.random(3) + 0.01) * 0.05
        w = rng.random(k)
        a = c.T @ (c * w[:, None]) / w.sum()
        w1, v1 = np.linalg.eigh(a)
        w2, v2 = O.eigh3(a)
        gap = min(w1[1] - w1[0], w1[2] - w1[1]) / w1[2]
        assert np.abs(w1 - w2).max() <= 1e-14 * np.abs(w1).max()
        assert np.abs(v1 - v2).max() * gap <= 1e-13
    for a in (np.zeros((3, 3)), np.eye(3), np.diag([3.0, 1.0, 2.0])):
        w1, v1 = np.linalg.eigh(a)
        w2, v2 = O.eigh3(a)
        assert np.array_equal(w1, w2) and np.array_equal(v1, v2)


def test_normals_radius():
    g = load_golden("normals_2k.npz")
    n1 = O.compute_normals(g["queries"], g["cloud"], radius=float(g["radius"]))
    assert np.abs(n1 - g["n_radius"]).max() < 1e-9  # signs as LAPACK returns them
    n2 = O.compute_normals(g["queries"], g["cloud"], radius=float(g["radius"]), pre_computed_normals=g["pre"])
    assert np.abs(n2 - g["n_radius_pre"]).max() < 1e-9


def test_normals_knn():
    g = load_golden("normals_2k.npz")
    n1 = O.compute_normals(g["queries"], g["cloud"], k=int(g["k"]))
    assert np.abs(n1 - g["n_knn"]).max() < 1e-9
    n2 = O.compute_normals(g["queries"], g["cloud"], k=int(g["k"]), pre_computed_normals=g["pre"])
    assert np.abs(n2 - g["n_knn_pre"]).max() < 1e-9


def test_local_rf():
    g = load_golden("shot_150.npz")
    lrf = O.shot_lrf(g["cloud"], g["keypoints"], float(g["radius"]))
    # a rank-deficient support (k < 4 here: the sparse off-cloud keypoint has ONE neighbour) has a
    # repeated zero eigenvalue: LAPACK's basis of that null space is rounding noise, not a target
    off, _ = O.radius_search(g["cloud"], g["keypoints"], float(g["radius"]))
    ok = np.diff(off) >= 4
    assert ok.sum() >= 150
    assert np.abs(lrf - g["lrf"])[ok].max() < 1e-9
    assert np.array_equal(lrf[-2], np.eye(3))  # empty neighbourhood (shot.py:24-25)
    e = load_golden("edge_dups.npz")
    lrf = O.shot_lrf(e["cloud"], e["keypoints"], float(e["radius"]))
    assert np.abs(lrf - e["lrf"]).max() < 1e-9


@pytest.mark.parametrize("norm", [True, False])
@pytest.mark.parametrize("mn", [10, 100])
def test_shot_single_scale(norm, mn):
    g = load_golden("shot_150.npz")
    d = O.shot_single_scale(g["cloud"], g["normals"], g["keypoints"], float(g["radius"]), norm, mn)
    ref = g[f"single_n{int(norm)}_m{mn}"]
    assert d.shape == ref.shape
    assert np.abs(d - ref).max() < TOL
    assert not d[-2].any()  # off-cloud keypoint with an empty neighbourhood -> zero row


def test_shot_with_subsampled_support_and_dups():
    g = load_golden("shot_150.npz")
    d = O.shot_single_scale(g["cloud"], g["normals"], g["keypoints"], float(g["radius"]), True, 10, support=g["support"])
    assert np.abs(d - g["single_sub"]).max() < TOL
    e = load_golden("edge_dups.npz")
    d = O.shot_single_scale(e["cloud"], e["normals"], e["keypoints"], float(e["radius"]), True, 5)
    # exact duplicate points give tied rho; the reference sorts with an UNSTABLE argsort
    # (shot.py:218), so which duplicate writes last is undefined there.  Rows without ties must
    # match; rows that differ must all contain a tie.
    off, idx, dist = O.radius_search(e["cloud"], e["keypoints"], float(e["radius"]), return_distance=True)
    tied = np.array([len(np.unique(dist[off[i]:off[i + 1]])) < off[i + 1] - off[i] for i in range(len(off) - 1)])
    row_err = np.abs(d - e["shot_m5"]).max(axis=1)
    assert (~tied).sum() >= 5 and tied.sum() >= 5
    assert row_err[~tied].max() < TOL


@pytest.mark.parametrize("nb,key", [(5, "fpfh5"), (4, "fpfh4")])
def test_fpfh(nb, key):
    g = load_golden("fpfh_200.npz")
    f = O.compute_fpfh_descriptor(g["kp_idx"], g["cloud"], g["normals"], float(g["radius"]), nb)
    assert np.abs(f - g[key]).max() < 1e-10 * max(1.0, np.abs(g[key]).max())


@pytest.mark.parametrize("nb,key", [(5, "fpfh5"), (3, "fpfh3")])
def test_fpfh_surface(nb, key):
    g = load_golden("fpfh_surface.npz")
    f = O.compute_fpfh_descriptor(g["kp_idx"], g["cloud"], g["normals"], float(g["radius"]), nb)
    assert np.abs(f - g[key]).max() < 1e-10 * max(1.0, np.abs(g[key]).max())


def test_fpfh_duplicates():
    e = load_golden("edge_dups.npz")
    f = O.compute_fpfh_descriptor(e["kp_idx"], e["cloud"], e["normals"], float(e["radius"]), 5)
    assert np.abs(f - e["fpfh5"]).max() < 1e-10 * max(1.0, np.abs(e["fpfh5"]).max())


def test_matching():
    from shot_fpfh_amd.matching.filters import threshold_filter

    g = load_golden("match_300.npz")
    s, r = O.basic_matching(g["scan"], g["ref"])
    assert np.array_equal(s, g["basic_s"]) and np.array_equal(r, g["basic_r"])
    s, r = O.match_descriptors(g["scan"], g["ref"])
    assert np.array_equal(s, g["md_s"]) and np.array_equal(r, g["md_r"])
    s, r = O.match_descriptors(g["scan"], g["ref"], threshold_filter, threshold_multiplier=10)
    assert np.array_equal(s, g["thr_s"]) and np.array_equal(r, g["thr_r"])
    s, r = O.match_descriptors(g["scan"], g["ref"], filter_nonreciprocal=True, n_min_matches=100)
    assert np.array_equal(s, g["rec_s"]) and np.array_equal(r, g["rec_r"])
    s, r = O.match_descriptors(g["scan"], g["ref"], filter_nonreciprocal=True, n_min_matches=10**6)
    assert np.array_equal(s, g["recbig_s"]) and np.array_equal(r, g["recbig_r"])


def test_ransac_scoring():
    g = load_golden("ransac_500.npz")
    a = g["scan_kp"][g["scan_idx"]]
    b = g["ref_kp"][g["ref_idx"]]
    inl = O.ransac_score(a, b, g["draw_rt"], float(g["thr"]))
    assert np.array_equal(inl, g["draw_inliers"])
    assert inl.max() / a.shape[0] == float(g["ratio"])


def test_matching_multiscale():
    from shot_fpfh_amd.matching.filters import threshold_filter

    g = load_golden("match3d_200.npz")
    s, r = O.match_descriptors_multiscale(g["scan"], g["ref"])
    assert np.array_equal(s, g["md_s"]) and np.array_equal(r, g["md_r"])
    assert 77 not in s  # empty at every scale -> stays at max_val -> dropped
    s, r = O.match_descriptors_multiscale(g["scan"], g["ref"], threshold_filter, threshold_multiplier=3)
    assert np.array_equal(s, g["thr_s"]) and np.array_equal(r, g["thr_r"])


def test_local_pca_and_features_match_reference_golden():
    """pca() / compute_local_pca_with_moments and the feature functions on top."""
    g = load_golden("pca_features_300.npz")
    r = float(g["radius"])
    w, v, mo, sizes = O.local_pca(g["queries"], g["cloud"], radius=r, moments=True)
    assert np.array_equal(sizes, g["sizes"])
    assert np.abs(w - g["eigenvalues"]).max() < TOL
    assert np.abs(v - g["eigenvectors"]).max() < 1e-9  # signs as LAPACK returns them
    assert np.abs(mo - g["moments"]).max() < TOL
    wk, vk, mok, sk = O.local_pca(g["queries"], g["cloud"], k=int(g["k"]), moments=True)
    assert np.array_equal(sk, g["sizes_knn"])
    assert np.abs(wk - g["eigenvalues_knn"]).max() < TOL and np.abs(mok - g["moments_knn"]).max() < TOL
    assert np.abs(vk - g["eigenvectors_knn"]).max() < 1e-9
    assert np.abs(O.compute_pca_based_features(g["queries"], g["cloud"], r) - g["features"]).max() < 1e-9
    basic = O.compute_pca_based_basic_features(g["queries"], g["cloud"], r)
    for got, key in zip(basic, ("verticality", "linearity", "planarity", "basic_sphericity")):
        assert np.abs(got - g[key]).max() < 1e-9
    assert np.abs(O.compute_sphericity(g["queries"], g["cloud"], r) - g["sphericity"]).max() < 1e-9


def test_config1_plumbing_case_fpfh_matches_reference_golden():
    """BASELINE config 1: ~35k-point surface cloud, k = 30 PCA normals oriented by the stored ones, 500 random
    keypoints, FPFH 5 bins at radius 0.05 x the bounding-box extent (reference outputs: config1_fpfh_500.npz)."""
    from conftest import config1_cloud

    g = load_golden("config1_fpfh_500.npz")
    p, d = config1_cloud(int(g["n"]), int(g["seed"]))
    normals = O.compute_normals(p, p, k=30, pre_computed_normals=d, knn_radius_hint=0.05)
    assert np.abs(normals[:200] - g["normals_head"]).max() < 1e-9
    f = O.compute_fpfh_descriptor(g["kp_idx"], p, normals, float(g["radius"]), 5)
    assert np.abs(f - g["fpfh"]).max() <= 1e-10 * max(1.0, np.abs(g["fpfh"]).max())


def test_azimuth_idx_boundary_table():
    """get_azimuth_idx on the reference's own outputs for every exact octant boundary, the doubles
    one ulp either side of each, signed zeros, subnormal / huge magnitudes and 20 000 random points."""
    g = load_golden("azimuth_table.npz")
    with np.errstate(over="ignore"):
        got = O.azimuth_idx(g["x"], g["y"])
    assert np.array_equal(got, g["idx"])
    # the table SURVEY 8a quotes: boundaries belong to the LOWER octant
    named = {(1, 0): 3, (0, 1): 5, (-1, 0): 7, (0, -1): 1, (1, 1): 4, (-1, 1): 6, (-1, -1): 0, (1, -1): 2, (0, 0): 0}
    for (x, y), k in named.items():
        assert O.azimuth_idx([float(x)], [float(y)])[0] == k


def test_shot_row_with_every_neighbour_on_a_bin_boundary():
    """compute_single_shot_descriptor with an identity frame and neighbours exactly ON the octant, elevation and radial
    boundaries (rho == r/2, r/4, 3r/4; z == 0; axes and diagonals) and cosines on the half-way points of the cosine bins."""
    g = load_golden("shot_boundary.npz")
    for key, normalize in (("desc_n1", True), ("desc_n0", False)):
        d = O.shot_single(g["point"], g["neighbors"], g["normals"], float(g["radius"]), np.eye(3), normalize, 5)
        assert np.abs(d - g[key]).max() <= TOL, np.abs(d - g[key]).max()


def test_serial_shot_descriptor():
    """compute_shot_descriptor (shot.py:310-499): the frame is computed without the keypoint in its support, so rows
    differ from ShotMultiprocessor's wherever a sign vote was within one of a tie."""
    g = load_golden("shot_150.npz")
    kp = g["keypoints"][:60]
    d = O.compute_shot_descriptor(kp, g["cloud"], g["normals"], float(g["radius"]), 10)
    assert np.abs(d - g["serial"]).max() <= TOL
    par = O.shot_single_scale(g["cloud"], g["normals"], kp, float(g["radius"]), True, 10)
    assert (np.abs(par - g["serial"]).max(axis=1) > 1e-6).sum() >= 1  # the fixture does tell the two variants apart


def test_fpfh_sample_helper_equals_the_full_function():
    from conftest import synth_cloud

    p, nr, rng = synth_cloud(6000, 77)
    kp = np.sort(rng.choice(6000, 150, replace=False))
    full = O.compute_fpfh_descriptor(kp, p, nr, 0.09, 5)
    assert np.array_equal(O.compute_fpfh_descriptor_sample(kp, p, nr, 0.09, 5), full)
    g = load_golden("fpfh_200.npz")
    f = O.compute_fpfh_descriptor_sample(g["kp_idx"], g["cloud"], g["normals"], float(g["radius"]), 5)
    assert np.abs(f - g["fpfh5"]).max() <= 1e-10 * max(1.0, np.abs(g["fpfh5"]).max())


def test_numpy_shaped_cpu_baseline_reproduces_the_reference():
    """oracle/numpy_shaped.py (the reference-shaped CPU baseline bench.py times) against the reference's goldens."""
    from oracle import numpy_shaped as NS

    g = load_golden("fpfh_200.npz")
    f = NS.fpfh_numpy_shaped(g["kp_idx"], g["cloud"], g["normals"], float(g["radius"]), 5)
    assert np.abs(f - g["fpfh5"]).max() <= TOL
    s = load_golden("shot_150.npz")
    d = NS.shot_numpy_shaped(s["cloud"], s["normals"], s["keypoints"], float(s["radius"]), True, 10, n_procs=2)
    assert np.abs(d - s["single_n1_m10"]).max() <= TOL


# ---- degenerate cloud families against the reference's own outputs (tools/gen_golden_r2b.py) ---------------------------
@pytest.mark.parametrize("name", ["lattice", "plane", "rough_plane", "duplicates", "far_origin"])
def test_degenerate_families_golden(name):
    """Lattice (neighbours exactly ON the radius, distance ties), exact plane with normals along the plane's (alpha = 0,
    theta = 0 on histogram edges), rough plane, duplicated points, a cloud 4096 units from the origin: neighbour lists bit
    for bit, FPFH (4 and 5 bins), local frames and SHOT rows as the reference produced them."""
    from oracle import oracle as O

    g = load_golden("degenerate_families.npz")
    p, nr, r = g[f"{name}_cloud"], g[f"{name}_normals"], float(g[f"{name}_radius"])
    off, idx, dist = O.radius_search(p, p, r, return_distance=True)
    assert np.array_equal(off, g[f"{name}_offsets"]) and np.array_equal(idx, g[f"{name}_idx"])
    assert np.array_equal(dist, g[f"{name}_dist"])
    if name == "lattice":
        assert (dist == np.sqrt(5.0 / 256.0)).any()  # the shell at exactly the radius is in the lists
    for nb in (4, 5):
        got = O.compute_fpfh_descriptor(g[f"{name}_kp"], p, nr, r, nb)
        assert np.abs(got - g[f"{name}_fpfh{nb}"]).max() < 1e-12
    if f"{name}_shot" in g.files:
        kq = g[f"{name}_shot_kp"]
        framed = np.diff(O.radius_search(p, kq, r)[0]) >= 5  # fewer points span no frame (two zero eigenvalues)
        assert np.abs(O.shot_lrf(p, kq, r) - g[f"{name}_lrf"])[framed].max() < 1e-12
        d = O.shot_single_scale(p, nr, kq, r, normalize=True, min_neighborhood_size=5)
        assert np.abs(d - g[f"{name}_shot"])[framed].max() < 1e-12


def test_config_c4_shaped_pair_match_vector():
    """The reference on a config-4-shaped pair (tools/gen_golden_r3.py): the oracle chain SHOT -> basic_matching gives the
    reference's match vector exactly, hence also its 93.9 % share of matches that recover the true correspondence."""
    g = load_golden("c4_pair_6k.npz")
    r = float(g["radius"])
    ds = O.shot_single_scale(g["scan"], g["normals"], g["scan"], r, True, 10)
    dr = O.shot_single_scale(g["ref"], g["ref_normals"], g["ref"], r, True, 10)
    assert np.abs(ds[g["sample_rows"]] - g["scan_desc_sample"]).max() < 1e-12
    si, ri = O.basic_matching(ds, dr)
    assert np.array_equal(si, g["match_scan"]) and np.array_equal(ri, g["match_ref"])


# ---- round 6: caller-supplied lists, float32 inputs, post-ICP metrics ---------------------------------------------------------
@pytest.mark.parametrize("kind", ["knn", "wide", "same"])
def test_shot_on_caller_supplied_lists(kind):
    """compute_local_rf / compute_descriptor on lists that did not come from a search of `radius` (shot_parallelization.py:46-133):
    KDTree.query lists, lists of 1.5 x the radius (neighbours beyond the radius keep their negative weights), the radius itself."""
    g = load_golden("shot_lists.npz")
    off, idx, r = g[f"{kind}_offsets"], g[f"{kind}_idx"], float(g["radius"])
    lrf = O.shot_lrf_lists(g["cloud"], g["keypoints"], off, idx, r)
    assert np.abs(lrf - g[f"{kind}_lrf"]).max() < 1e-9
    d = O.shot_lists(g["cloud"], g["normals"], g["keypoints"], off, idx, r, g[f"{kind}_lrf"], True, 10)
    assert np.abs(d - g[f"{kind}_desc"]).max() < TOL * 10


@pytest.mark.parametrize("tag", ["unit", "offset"])
@pytest.mark.parametrize("ntag", ["n64", "n32"])
def test_float32_clouds_stay_within_the_tolerance_of_the_float64_arithmetic(tag, ntag):
    """The reference run on float32 arrays (get_data hands the PLY's float32 columns on, io_ply.py:269; `neighbors - point` and
    `np.linalg.norm` then round to float32, shot.py:211-214, fpfh.py:45-48) against the float64 arithmetic on the same values --
    which is what the build computes after its cast at the boundary: every row within 1e-5 (the count is asserted to be 0)."""
    g = load_golden("shot_f32.npz")
    p = g[f"{tag}_cloud"]
    assert p.dtype == np.float32
    nrm = g["normals64"] if ntag == "n64" else g["normals64"].astype(np.float32)
    kp = p[g["keypoints_indices"]][g["rows"]]
    d = O.shot_single_scale(p.astype(np.float64), nrm.astype(np.float64), kp.astype(np.float64), float(g["radius"]), True,
                            int(g["min_neighborhood_size"]))
    gap = np.abs(d - g[f"{tag}_{ntag}_desc"]).max(axis=1)
    assert int((gap > 1e-5).sum()) == 0, (int((gap > 1e-5).sum()), gap.max())
    assert int(g[f"{tag}_{ntag}_rows_beyond_1e5"]) == 0  # (reference on float32 arrays vs reference on float64 arrays, all 2 000 rows)
    f = load_golden("fpfh_f32.npz")
    pf = f[f"{tag}_cloud"]
    nf = f["normals64"] if ntag == "n64" else f["normals64"].astype(np.float32)
    got = O.compute_fpfh_descriptor(f["keypoints_indices"], pf.astype(np.float64), nf.astype(np.float64), float(f["radius"]), int(f["n_bins"]))
    want = f[f"{tag}_{ntag}_desc"]
    bad = (np.abs(got - want) > 1e-5 * np.maximum(1.0, np.abs(want))).any(axis=1)
    assert int(bad.sum()) == 0, int(bad.sum())
