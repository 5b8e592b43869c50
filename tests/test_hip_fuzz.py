"""Differential tests of the HIP path against the CPU oracle on cloud FAMILIES the goldens do not hold: lattices (exact
distance ties, neighbours exactly ON the search radius), planes and lines (rank-deficient supports), clustered densities,
duplicated points, clouds far from the origin or squeezed flat, and every n_bins / min_neighborhood_size / radius drawn at
random from a fixed seed.  Sizes are what the scalar oracle does in a second or two.  All through the C ABI.

What is compared, and how strictly:
  * neighbour lists (K1 + K2): offsets, indices and distances bit for bit -- the inclusion test d2 <= r2 is decided on
    the same float64 expression as the reference's, so a lattice point exactly r away is in or out identically;
  * SPFH counts are integers and FPFH is a sum of them: <= 1e-9 (observed ~1e-14);
  * SHOT rows: <= 1e-9, on clouds where the reference itself is defined -- i.e. without exactly equidistant neighbours
    (last-writer-wins on an unstable argsort, shot.py:218) and without supports so flat that the frame's third axis is
    rounding noise.  Lattices and exact planes therefore check neighbour search, normals, FPFH and PCA only.
"""
import os

import numpy as np
import pytest

from conftest import FAMILIES, _unit, family, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import shot_fpfh_amd as s

    return s.default_engine()


@pytest.fixture(scope="module")
def O():
    from oracle import oracle

    return oracle


def _radius_for(p, rng, target):
    """a radius that gives about `target` neighbours on average, jittered"""
    n = p.shape[0]
    ext = np.maximum(p.max(0) - p.min(0), 1e-3)
    dims = ext > 0.02 * ext.max()
    vol = np.prod(ext[dims])
    d = int(dims.sum())
    unit_ball = {1: 2.0, 2: np.pi, 3: 4.18879}[d]
    return float((target * vol / (n * unit_ball)) ** (1.0 / d) * rng.uniform(0.8, 1.3))


# three seeds by default; SF_FUZZ_SEEDS="a:b" runs seeds a .. b-1 (a long sweep on an idle GPU: tools/fuzz_sweep.sh)
_sweep = os.environ.get("SF_FUZZ_SEEDS")
SEEDS = list(range(*(int(v) for v in _sweep.split(":")))) if _sweep else [0, 1, 2]


@pytest.mark.parametrize("seed", SEEDS)
@pytest.mark.parametrize("name", FAMILIES)
def test_neighbour_lists_bit_exact(eng, O, name, seed):
    rng = np.random.default_rng(FAMILIES.index(name) + 500 + 7919 * seed)
    p, _, _, _ = family(name, 4096, rng)
    n = p.shape[0]
    for target in (8, 60):
        r = _radius_for(p, rng, target)
        if name == "lattice":  # EXACTLY the distance of a lattice shell (nudged up when r * r rounds below the shell's d2)
            shell = float(rng.choice([3, 5, 6, 9, 14])) / 256.0
            r = float(np.sqrt(shell))
            if r * r < shell:
                r = float(np.nextafter(r, np.inf))
        q = np.vstack([p[rng.choice(n, 300, replace=False)], p.min(0) + (p.max(0) - p.min(0)) * (rng.random((100, 3)) * 1.4 - 0.2)])
        cloud = eng.cloud(p)
        off, idx, dist = cloud.radius_search(q, r).export(return_distance=True)
        off_o, idx_o, dist_o = O.radius_search(p, q, r, return_distance=True)
        assert np.array_equal(off, off_o), f"{name}: list lengths differ at r={r}"
        assert np.array_equal(idx, idx_o) and np.array_equal(dist, dist_o)
        if name == "lattice":  # and the shell really is included: some neighbour sits exactly on it
            assert (dist == np.sqrt(shell)).any()
        nb = cloud.radius_search_self(r)
        off_s, idx_s = nb.export()
        off_b, idx_b = O.radius_search(p, p, r)
        # self-search rows are by cell-sorted position (row i <-> point perm[i]); the entries are original indices
        perm = cloud.perm()
        assert np.array_equal(np.diff(off_s), np.diff(off_b)[perm])
        for pos in rng.choice(n, 200, replace=False):
            i = int(perm[pos])
            assert np.array_equal(idx_s[off_s[pos]:off_s[pos + 1]], idx_b[off_b[i]:off_b[i + 1]])


@pytest.mark.parametrize("seed", SEEDS)
@pytest.mark.parametrize("name", FAMILIES)
def test_fpfh_vs_oracle(eng, O, name, seed):
    import shot_fpfh_amd as s

    rng = np.random.default_rng(FAMILIES.index(name) + 600 + 7919 * seed)
    p, nr, _, _ = family(name, 3000, rng)
    n = p.shape[0]
    for trial in range(3):
        n_bins = int(rng.choice([2, 3, 4, 5, 6, 7, 8, 9, 11]))
        r = _radius_for(p, rng, int(rng.choice([12, 40, 120])))
        if name == "lattice":
            r = float(np.sqrt(rng.choice([3, 5, 6])) / 16.0)
        kp = np.sort(rng.choice(n, 400, replace=False))
        got = s.compute_fpfh_descriptor(kp, p, nr, r, n_bins)
        want = O.compute_fpfh_descriptor(kp, p, nr, r, n_bins)
        assert got.shape == want.shape == (400, n_bins**3)
        err = np.abs(got - want).max()
        assert err < 1e-9, f"{name} n_bins={n_bins} r={r}: {err}"


@pytest.mark.parametrize("seed", SEEDS)
@pytest.mark.parametrize("name", FAMILIES)
def test_normals_and_pca_vs_oracle(eng, O, name, seed):
    import shot_fpfh_amd as s

    rng = np.random.default_rng(FAMILIES.index(name) + 700 + 7919 * seed)
    p, nr, ties, flat = family(name, 3000, rng)
    n = p.shape[0]
    q = p[rng.choice(n, 300, replace=False)]
    r = _radius_for(p, rng, 40)
    if name == "lattice":
        r = float(np.sqrt(5) / 16.0)
    off, _ = O.radius_search(p, q, r)
    sizes = np.diff(off)
    w, v, mom, cnt = s.descriptors.compute_local_pca_with_moments(q, p, radius=r)
    wo, vo, mo, co = O.local_pca(q, p, radius=r, moments=True)
    assert np.array_equal(np.asarray(cnt), np.asarray(co)) and np.array_equal(np.asarray(cnt), sizes)
    scale = max(1.0, float(np.abs(p).max())) ** 2
    assert np.abs(w - wo).max() < 1e-9 * scale
    # eigenvectors, and the moments (projections onto them), only where the spectrum is simple: a repeated eigenvalue
    # (lattice symmetry, exact planes with an isotropic in-plane spread) has no defined basis, in LAPACK either
    gap = np.minimum(w[:, 1] - w[:, 0], w[:, 2] - w[:, 1])
    simple = (gap > 1e-6 * np.maximum(w[:, 2], 1e-300)) & (sizes >= 4)
    if simple.any() and name != "lattice":
        assert np.abs(v - vo)[simple].max() < 1e-6
        assert np.abs(mom - mo)[simple].max() < 1e-6 * scale * max(1.0, float(np.abs(p).max()))
    elif simple.any():
        # axis-aligned lattice: covariance entries are exact zeros whose SIGN (+0 / -0, i.e. the order the products were
        # summed in) steers LAPACK's Householder / Givens signs -- compare the axes up to orientation
        same = np.minimum(np.abs(v - vo), np.abs(v + vo)).max(axis=1)  # per column
        assert same[simple].max() < 1e-6
    nrm = s.compute_normals(q, p, radius=r, pre_computed_normals=nr[:300])
    nrm_o = O.compute_normals(q, p, radius=r, pre_computed_normals=nr[:300])
    if simple.any():
        assert np.abs(nrm - nrm_o)[simple].max() < 1e-6
    if not ties:  # k-NN sets are defined only without equidistant candidates
        k = int(rng.choice([5, 17, 30]))
        nk = s.compute_normals(q, p, k=k, pre_computed_normals=nr[:300])
        nk_o = O.compute_normals(q, p, k=k, pre_computed_normals=nr[:300])
        wk = O.local_pca(q, p, k=k)[0]
        simple_k = np.minimum(wk[:, 1] - wk[:, 0], wk[:, 2] - wk[:, 1]) > 1e-6 * np.maximum(wk[:, 2], 1e-300)
        if not flat and simple_k.any():
            assert np.abs(nk - nk_o)[simple_k].max() < 1e-6


@pytest.mark.parametrize("seed", SEEDS)
@pytest.mark.parametrize("name", ["uniform", "clustered", "rough_plane", "far_origin"])
def test_shot_vs_oracle(eng, O, name, seed):
    """parallel (frame from the full support) and serial (frame without the keypoint) SHOT, random settings"""
    from shot_fpfh_amd.descriptors import ShotMultiprocessor
    from shot_fpfh_amd.descriptors.shot import compute_shot_descriptor

    rng = np.random.default_rng(FAMILIES.index(name) + 800 + 7919 * seed)
    p, nr, _, _ = family(name, 3000, rng)
    n = p.shape[0]
    for trial in range(3):
        r = _radius_for(p, rng, int(rng.choice([20, 70, 200, 330])))
        mn = int(rng.choice([0, 5, 10, 50]))
        norm = bool(rng.integers(0, 2))
        kp = np.vstack([p[rng.choice(n, 250, replace=False)], p.min(0) + (p.max(0) - p.min(0)) * rng.random((50, 3))])
        with ShotMultiprocessor(normalize=norm, min_neighborhood_size=mn, verbose=False) as sm:
            got = sm.compute_descriptor_single_scale(p, nr, kp, r)
        want = O.shot_single_scale(p, nr, kp, r, normalize=norm, min_neighborhood_size=mn)
        assert np.array_equal(got.any(axis=1), want.any(axis=1)), f"{name}: gate differs (r={r}, mn={mn})"
        # a support of fewer than 4 points spans no frame (two eigenvalues are zero: any basis of their plane is "the"
        # eigenvectors), so with a permissive gate such rows are whatever LAPACK's rounding makes them
        framed = np.diff(O.radius_search(p, kp, r)[0]) >= 5
        err = np.abs(got - want)[framed].max()
        assert err < 1e-9, f"{name} r={r} mn={mn} norm={norm}: {err}"
        got_s = compute_shot_descriptor(kp, p, nr, r, min_neighborhood_size=mn)
        want_s = O.compute_shot_descriptor(kp, p, nr, r, mn)
        assert np.array_equal(got_s.any(axis=1), want_s.any(axis=1))
        assert np.abs(got_s - want_s)[framed].max() < 1e-9


@pytest.mark.parametrize("seed", [901, 902, 903, 904])
def test_matching_and_ransac_scoring_vs_oracle(eng, O, seed):
    """K8 on descriptor-like rows with planted exact ties, duplicated rows and all-zero rows; K9 with matches sitting on
    the inlier threshold."""
    import shot_fpfh_amd as s

    rng = np.random.default_rng(seed)
    d = int(rng.choice([33, 125, 352]))
    m1, m2 = int(rng.integers(300, 1500)), int(rng.integers(300, 1500))
    a = np.abs(rng.standard_normal((m1, d))) * (rng.random((m1, d)) < 0.2)
    b = np.abs(rng.standard_normal((m2, d))) * (rng.random((m2, d)) < 0.2)
    b[rng.choice(m2, 40, replace=False)] = a[rng.choice(m1, 40, replace=False)]  # exact matches (distance 0)
    b[5] = b[900 % m2]                                                            # duplicated columns: first one wins
    a[7] = 0.0
    b[11] = 0.0
    idx_o, dist_o = O.match_argmin(a, b)
    idx_g, dist_g, _ = eng.match_argmin(a, b)
    assert np.array_equal(idx_g, idx_o) and np.array_equal(dist_g, dist_o)
    got_scan, got_ref = s.matching.basic_matching(a, b)
    want_scan, want_ref = O.basic_matching(a, b)
    assert np.array_equal(got_scan, want_scan) and np.array_equal(got_ref, want_ref)
    # K9
    n = 4000
    pa = rng.random((n, 3))
    rot = np.linalg.qr(rng.standard_normal((3, 3)))[0]
    rot *= np.sign(np.linalg.det(rot))
    t = rng.random(3)
    pb = pa @ rot.T + t
    thr = 0.01
    pb[:200] += _unit(rng.standard_normal((200, 3))) * thr            # on the threshold (up to rounding)
    pb[200:400] += _unit(rng.standard_normal((200, 3))) * thr * 1.01  # just outside
    rt = np.array([np.concatenate([rot.ravel(), t]), np.concatenate([np.eye(3).ravel(), np.zeros(3)])])  # R row-major, then t
    got = eng.ransac_score(pa, pb, rt, thr)
    want = O.ransac_score(pa, pb, rt, thr)
    assert np.array_equal(got, want) and n - 400 <= want[0] <= n - 200


@pytest.mark.parametrize("name", ["uniform", "clustered", "lattice", "duplicates", "far_origin", "slab"])
def test_grid_subsampling_partition_properties(eng, name):
    """voxel subsampling on the device: one representative per occupied voxel, the representative lies in its voxel and is
    the member closest to the voxel's barycentre (ties by the documented rule), for any voxel size."""
    from shot_fpfh_amd.core import grid_subsampling

    rng = np.random.default_rng(FAMILIES.index(name) + 1000)
    p, _, _, _ = family(name, 5000, rng)
    for vox in (0.013, 0.11, 0.7):
        sel = grid_subsampling(p, vox)
        keys = ((p - p.min(axis=0)) // vox).astype(np.int64)  # subsampling.py:13
        uniq, inv = np.unique(keys, axis=0, return_inverse=True)
        inv = inv.reshape(-1)
        assert sel.shape[0] == uniq.shape[0] and np.unique(inv[sel]).shape[0] == uniq.shape[0]
        # each representative attains the minimum distance to its voxel's barycentre
        sums = np.zeros((uniq.shape[0], 3))
        np.add.at(sums, inv, p)
        bary = sums / np.bincount(inv)[:, None]
        d2 = ((p - bary[inv]) ** 2).sum(1)
        best = np.full(uniq.shape[0], np.inf)
        np.minimum.at(best, inv, d2)
        assert np.all(d2[sel] <= best[inv[sel]] * (1 + 1e-12) + 1e-300)


# ---- the same families against the REFERENCE's own outputs (tests/golden/degenerate_families.npz, tools/gen_golden_r2b.py) ----
@pytest.mark.parametrize("name", ["lattice", "plane", "rough_plane", "duplicates", "far_origin"])
def test_degenerate_families_golden(eng, name):
    import shot_fpfh_amd as s
    from shot_fpfh_amd.descriptors import ShotMultiprocessor

    g = load_golden("degenerate_families.npz")
    p, nr, r = g[f"{name}_cloud"], g[f"{name}_normals"], float(g[f"{name}_radius"])
    off, idx, dist = eng.cloud(p).radius_search(p, r).export(return_distance=True)
    assert np.array_equal(off, g[f"{name}_offsets"]) and np.array_equal(idx, g[f"{name}_idx"])
    assert np.array_equal(dist, g[f"{name}_dist"])
    for nb in (4, 5):
        got = s.compute_fpfh_descriptor(g[f"{name}_kp"], p, nr, r, nb)
        assert np.abs(got - g[f"{name}_fpfh{nb}"]).max() < 1e-9, (name, nb)
    if f"{name}_shot" in g.files:
        kq = g[f"{name}_shot_kp"]
        cloud = eng.cloud(p, nr)
        nbq = cloud.radius_search(kq, r)
        framed = nbq.counts() >= 5
        assert np.abs(nbq.shot_lrf() - g[f"{name}_lrf"])[framed].max() < 1e-9
        with ShotMultiprocessor(normalize=True, min_neighborhood_size=5, verbose=False) as sm:
            d = sm.compute_descriptor_single_scale(p, nr, kq, r)
        assert np.abs(d - g[f"{name}_shot"])[framed].max() < 1e-9


# ---- the drop-in calls with the argument FORMS NumPy code hands them -------------------------------------------------
def test_drop_in_calls_accept_numpy_argument_forms(eng):
    """float32 / Fortran-ordered / strided / list inputs, index arrays of any integer type, NumPy scalars for radius and
    bin count: the reference's NumPy / sklearn code takes all of them; results must equal the canonical float64 C-order call."""
    import shot_fpfh_amd as s
    from shot_fpfh_amd.descriptors import ShotMultiprocessor
    from shot_fpfh_amd.matching import basic_matching

    rng = np.random.default_rng(77)
    p = rng.random((3000, 3), dtype=np.float32).astype(np.float64)  # float32-representable, so the float32 form is lossless
    nr = _unit(rng.standard_normal((3000, 3))).astype(np.float32).astype(np.float64)
    kp = np.sort(rng.choice(3000, 200, replace=False))
    r = 0.12
    base_f = s.compute_fpfh_descriptor(kp, p, nr, r, 5)
    base_n = s.compute_normals(p[kp], p, k=20)
    with ShotMultiprocessor(normalize=True, min_neighborhood_size=10, verbose=False) as sm:
        base_s = sm.compute_descriptor_single_scale(p, nr, p[kp], r)

    wide = np.zeros((3000, 7))
    wide[:, 1:4] = p
    forms = {
        "float32": (p.astype(np.float32), nr.astype(np.float32)),
        "fortran": (np.asfortranarray(p), np.asfortranarray(nr)),
        "strided view": (wide[:, 1:4], nr[::1]),
        "lists": (p.tolist(), nr.tolist()),
    }
    for name, (pp, nn) in forms.items():
        got = s.compute_fpfh_descriptor(kp, pp, nn, r, 5)
        assert got.dtype == np.float64 and np.array_equal(got, base_f), name
        assert np.array_equal(s.compute_normals(np.asarray(pp)[kp], pp, k=20), base_n), name
        with ShotMultiprocessor(normalize=True, min_neighborhood_size=10, verbose=False) as sm:
            assert np.array_equal(sm.compute_descriptor_single_scale(pp, nn, np.asarray(pp)[kp], r), base_s), name
    for kk in (kp.astype(np.int32), kp.astype(np.uint16), kp.tolist(), kp[::-1][::-1]):
        assert np.array_equal(s.compute_fpfh_descriptor(kk, p, nr, r, 5), base_f)
    assert np.array_equal(s.compute_fpfh_descriptor(kp, p, nr, np.float32(r).astype(np.float64), np.int64(5)),
                          s.compute_fpfh_descriptor(kp, p, nr, float(np.float32(r)), 5))
    # matching on float32 / non-contiguous descriptor matrices
    a, b = base_s[:150], base_s[50:]
    want = basic_matching(a, b)
    for aa, bb in ((np.asfortranarray(a), np.asfortranarray(b)), (base_s[:150:1], base_s[50::1])):
        got = basic_matching(aa, bb)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    # repeated and unsorted keypoint indices (fancy indexing semantics: one row per entry, in the order given)
    idx = np.array([5, 5, 1999, 0, 5, 42])
    rows = s.compute_fpfh_descriptor(idx, p, nr, r, 5)
    full = s.compute_fpfh_descriptor(np.arange(3000), p, nr, r, 5)
    assert np.array_equal(rows, full[idx])
    # empty keypoint set
    assert s.compute_fpfh_descriptor(np.zeros(0, dtype=np.int64), p, nr, r, 5).shape == (0, 125)
    with ShotMultiprocessor(normalize=True, min_neighborhood_size=10, verbose=False) as sm:
        assert sm.compute_descriptor_single_scale(p, nr, np.zeros((0, 3)), r).shape == (0, 352)


# ---- round 4: the paths a small cloud never reaches -------------------------------------------------------------------------
# 20 000 points (the single-sweep search with sampled slots, not the exact two-pass scheme of small query sets) at radii that
# put hundreds of points into many balls: second launches for long lists (K3, K5, K6, K7), the table of high bytes, the
# matrix-core form's high-byte term, lists overflowing their slots.  One seed by default (the oracle needs ~10 s per family).
LONG_SEEDS = SEEDS[:1] if not _sweep else SEEDS


@pytest.mark.parametrize("seed", LONG_SEEDS)
@pytest.mark.parametrize("name", ["uniform", "clustered", "rough_plane", "duplicates", "far_origin", "slab"])
def test_long_lists_vs_oracle(eng, O, name, seed):
    import shot_fpfh_amd as s
    from shot_fpfh_amd.descriptors import ShotMultiprocessor

    rng = np.random.default_rng(FAMILIES.index(name) + 900 + 7919 * seed)
    p, nr, ties, flat = family(name, 20000, rng)
    n = p.shape[0]
    if name in ("rough_plane", "slab"):  # consistent normals: bin counts close to the list length (high bytes really used)
        nr = np.tile(np.array([[0.0, 0.0, 1.0]]), (n, 1))
    r = _radius_for(p, rng, int(rng.choice([260, 320, 400])))
    cloud = eng.cloud(p)
    for _ in range(6):  # (the jittered radius may leave every list short: grow it until some list exceeds 255 points)
        nb = cloud.radius_search_self(r)
        cnt = nb.counts()[np.argsort(cloud.perm())]  # by original index
        nb.free()
        if cnt.max() > 300:
            break
        r *= 1.2
    cloud.free()
    assert cnt.max() > 255, (name, cnt.max())
    # keypoints of every kind: the longest lists, lists around the 255 / 256 boundary, short ones, random ones
    order = np.argsort(cnt)
    near = order[np.searchsorted(cnt[order], 250):np.searchsorted(cnt[order], 262)][:30]
    kp = np.unique(np.concatenate([order[-40:], near, order[:20], rng.choice(n, 60, replace=False)]))
    n_bins = int(rng.choice([3, 4, 5]))
    got = s.compute_fpfh_descriptor(kp, p, nr, r, n_bins, verbose=False)
    want = O.compute_fpfh_descriptor(kp, p, nr, r, n_bins)
    assert np.abs(got - want).max() < 1e-9, (name, n_bins, r, np.abs(got - want).max())
    pre = np.tile(np.array([[0.0, 0.0, 1.0]]), (kp.size, 1))
    a = s.compute_normals(p[kp], p, radius=r, pre_computed_normals=pre)
    b = O.compute_normals(p[kp], p, radius=r, pre_computed_normals=pre)
    if not flat:
        assert np.abs(a - b).max() < 1e-9, (name, np.abs(a - b).max())
    if not ties and not flat:
        with ShotMultiprocessor(min_neighborhood_size=int(rng.choice([5, 100])), normalize=bool(rng.integers(0, 2)), verbose=False) as sm:
            d = sm.compute_descriptor_single_scale(p, nr, p[kp], r)
            do = O.shot_single_scale(p, nr, p[kp], r, sm.normalize, sm.min_neighborhood_size)
        assert np.abs(d - do).max() < 1e-9, (name, np.abs(d - do).max())
