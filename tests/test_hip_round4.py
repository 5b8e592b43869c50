"""GPU parity tests added in round 4 (all through the C ABI): the hot path on clouds that are NOT uniform volumes.

  * K2 sizes its slots from a sample of the lists (surface scans take the single sweep) and re-does only the lists that
    overflow their slot (dense clusters); neighbour sets stay bit-exact;
  * K3 / K5 / K6 / K7 dispatch per keypoint by list length: the keypoints whose own list fits the register-cached / matrix-core
    form run it whatever the longest list of the cloud is, the others are served by a second launch; a point with more than
    255 neighbours keeps the high bytes of its SPFH counts in a side table -- all against the oracle, sharded == unsharded
    bit for bit, and the exchange's wire image carries the high bytes.
"""
import os

import numpy as np
import pytest

from conftest import config1_cloud, family, synth_cloud

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


def clustered(n, seed=4):
    p, nr, _, _ = family("clustered", n, np.random.default_rng(seed))
    return p, nr


def dense_cap(n, seed=4):
    """A noisy sphere with outward normals (as tests/conftest.py::config1_cloud) whose sampling is far from even: half of
    the points sit in a cap around +z.  On a smooth surface with consistent normals nearly all pairs of a point fall into
    the same few SPFH bins, so a point with k neighbours has bin counts close to k: above 255 the byte table's high bytes
    are really used (with random normals the counts spread over 25 bins and never get there)."""
    rng = np.random.default_rng(seed)
    d = rng.standard_normal((n, 3))
    d[: n // 2] = np.array([0.0, 0.0, 1.0]) + 0.2 * rng.standard_normal((n // 2, 3))
    d /= np.linalg.norm(d, axis=1)[:, None]
    p = (0.5 + 0.5 * d * (1.0 + 0.002 * rng.standard_normal((n, 1)))).astype(np.float32).astype(np.float64)
    keep = np.unique(p, axis=0, return_index=True)[1]  # (float32 rounding may merge points of the dense cap)
    keep.sort()
    return p[keep], d[keep]


def launches(eng, fn):
    eng.sync()
    eng.profile_reset()
    eng.profile(True)
    try:
        out = fn()
        eng.sync()
    finally:
        eng.profile(False)
    return out, {k: v[0] for k, v in eng.profile_report().items() if v[0]}


# ---- K2 ----------------------------------------------------------------------------------------------------------------
def test_surface_scan_takes_the_single_sweep(eng, O):
    """A noisy sphere (the stand-in for the Stanford scans): the bounding-box density is an order of magnitude below the
    density on the surface.  The slots come from a sample of the lists themselves: one sweep, no second pass, and the
    neighbour sets are the oracle's."""
    p, _ = config1_cloud(60000, 2)
    r = 0.03
    cloud = eng.cloud(p)
    nb, rep = launches(eng, lambda: cloud.radius_search_self(r))
    assert rep.get("k2_radius_slots") == 1 and "k2_radius_fill" not in rep and "k2_radius_refill" not in rep, rep
    assert rep.get("k2_sample", 0) >= 1  # (the first search of this cloud counted a sample)
    off, idx = nb.export()
    perm = cloud.perm()
    rows = np.random.default_rng(0).choice(60000, 400, replace=False)
    woff, widx = O.radius_search(p, p[perm[rows]], r)
    for i, row in enumerate(rows):
        assert np.array_equal(idx[off[row]:off[row + 1]], widx[woff[i]:woff[i + 1]])
    nb.free()
    # the next search with this radius sizes its slots from the first one's lists: no sample, same lists
    nb2, rep2 = launches(eng, lambda: cloud.radius_search_self(r))
    assert "k2_sample" not in rep2 and "k2_radius_fill" not in rep2, rep2
    off2, idx2 = nb2.export()
    assert np.array_equal(off, off2) and np.array_equal(idx, idx2)
    nb2.free()
    cloud.free()


def test_dense_clusters_redo_only_the_lists_that_overflow(eng, O):
    """Six tight blobs in a sparse background.  With slots sized for the bulk of the lists (forced here: the sample would
    have seen the blobs) the lists of the blobs' cores overflow -- and are re-done on their own (k2_radius_refill), the
    sweep's other lists stay where they are (no k2_radius_fill)."""
    p, _ = clustered(80000)
    r = 0.012
    cloud = eng.cloud(p)
    os.environ["SF_K2_CAP"] = "320"
    try:
        nb, rep = launches(eng, lambda: cloud.radius_search_self(r))
    finally:
        del os.environ["SF_K2_CAP"]
    cnt = nb.counts()
    assert 0 < (cnt > 320).mean() < 0.5, (cnt.max(), cnt.mean())
    assert rep.get("k2_radius_slots") == 1 and rep.get("k2_radius_refill") == 1 and "k2_radius_fill" not in rep, rep
    off, idx = nb.export()
    perm = cloud.perm()
    long_rows = np.argsort(cnt)[-150:]
    rows = np.concatenate([long_rows, np.random.default_rng(1).choice(80000, 250, replace=False)])
    woff, widx = O.radius_search(p, p[perm[rows]], r)
    for i, row in enumerate(rows):
        assert np.array_equal(idx[off[row]:off[row + 1]], widx[woff[i]:woff[i + 1]])
    nb.free()
    # the same search sized from the previous one's statistics (no sample, nothing overflows) gives the same lists
    nb2, rep2 = launches(eng, lambda: cloud.radius_search_self(r))
    assert "k2_sample" not in rep2 and "k2_radius_refill" not in rep2 and "k2_radius_fill" not in rep2, rep2
    off2, idx2 = nb2.export()
    assert np.array_equal(off, off2) and np.array_equal(idx, idx2)
    nb2.free()
    cloud.free()


def test_undersized_slots_are_corrected_list_by_list(eng, O):
    """A first search on a sparse sub-range leaves a hint far too small for the dense range searched next: every long list
    overflows its slot and is re-done; results are exact either way."""
    p, _ = clustered(60000, seed=9)
    r = 0.015
    cloud = eng.cloud(p)
    full = cloud.radius_search_self(r)
    cnt = full.counts()
    off, idx = full.export()
    full.free()
    # ranges of cell-sorted positions: the sparsest and the densest 20000 consecutive positions
    win = np.convolve(cnt, np.ones(20000), "valid")
    lo, hi = int(np.argmin(win)), int(np.argmax(win))
    a = cloud.radius_search_self(r, lo, lo + 20000)  # (leaves the hint of a sparse range)
    a.free()
    nb, rep = launches(eng, lambda: cloud.radius_search_self(r, hi, hi + 20000))
    assert "k2_sample" not in rep
    o2, i2 = nb.export()
    assert np.array_equal(np.diff(o2), cnt[hi:hi + 20000])
    assert np.array_equal(i2, idx[off[hi]:off[hi + 20000]])
    nb.free()
    cloud.free()


# ---- per-keypoint dispatch ---------------------------------------------------------------------------------------------
@pytest.fixture(scope="module", params=["blobs", "cap"])
def clustered_job(eng, request):
    from shot_fpfh_amd.sharding import DescriptorJob

    if request.param == "blobs":
        p, nr = clustered(70000)
        r = 0.0125
    else:
        p, nr = dense_cap(70000)
        r = 0.02
    job = DescriptorJob(eng, p, nr, r, n_bins=5, min_neighborhood_size=10)
    _, rep = launches(eng, job.step)
    yield job, p, nr, r, rep
    job.close()


def test_one_dense_cluster_does_not_move_the_other_keypoints_to_a_fallback_form(eng, clustered_job):
    job, p, nr, r, rep = clustered_job
    nb = job.cloud.radius_search_self(r)
    cnt = nb.counts()
    nb.free()
    assert (cnt > 255).sum() > 500 and (cnt <= 255).mean() > 0.5, (cnt.max(), (cnt > 255).mean())
    # the main launches are the register-cached / matrix-core forms; the long lists have launches of their own; the frame
    # moments still come out of K6's sweep (no k4_shot_lrf), K5 stays fused
    for name in ("k5_shot", "k5_shot_tail", "k6_spfh", "k6_spfh_tail", "k7_fpfh", "k7_fpfh_tail", "k4_lrf_from_cov"):
        assert rep.get(name) == 1, (name, rep)
    assert "k4_shot_lrf" not in rep, rep


def test_clustered_cloud_descriptors_against_the_oracle(eng, O, clustered_job):
    """Rows of every kind: keypoints with long lists (second launches, high bytes of their own counts), keypoints whose
    neighbours have long lists (the matrix-core form + the high-byte term), and the undisturbed rest."""
    job, p, nr, r, _ = clustered_job
    nb = job.cloud.radius_search_self(r)
    cnt = nb.counts()
    off, idx = nb.export()
    nb.free()
    orig = job.block_original_indices()
    inv = np.empty_like(orig)
    inv[orig] = np.arange(orig.size)
    long_pt = cnt > 255
    has_long_nb = np.array([long_pt[inv[idx[off[i]:off[i + 1]]]].any() for i in range(cnt.size)])
    rng = np.random.default_rng(3)
    kinds = {
        "long": np.flatnonzero(long_pt),
        "short_with_long_neighbours": np.flatnonzero(~long_pt & has_long_nb),
        "undisturbed": np.flatnonzero(~long_pt & ~has_long_nb),
    }
    for name, pool in kinds.items():
        assert pool.size > 100, name
        rows = np.sort(rng.choice(pool, 80, replace=False))
        got_f = np.stack([job.fpfh_out.rows_to_host(int(i), 1)[0] for i in rows])
        want_f = O.compute_fpfh_descriptor_sample(orig[rows], p, nr, r, 5)
        assert close(got_f, want_f).all(), (name, np.abs(got_f - want_f).max())
        got_s = np.stack([job.shot_out.rows_to_host(int(i), 1)[0] for i in rows])
        want_s = O.shot_single_scale(p, nr, p[orig[rows]], r, True, 10)
        assert close(got_s, want_s).all(), (name, np.abs(got_s - want_s).max())
        assert np.abs(got_s - want_s).max() < 1e-9 and np.abs(got_f - want_f).max() < 1e-9


def test_spfh_counts_above_255_are_exact(eng, O):
    """The integer SPFH table of a surface with a densely sampled cap, exported as the reference's float64 rows (count / k):
    equal to the oracle's bit for bit -- low bytes + 256 x high bytes, bins with more than 255 counts among them."""
    p, nr = dense_cap(30000, seed=6)
    r = 0.03
    cloud = eng.cloud(p, nr)
    nb = cloud.radius_search_self(r)
    assert nb.max_count > 600
    sp = eng.spfh(cloud, 5, nb.max_count)
    sp.compute(nb)
    got = sp.export()
    _, want = O.compute_fpfh_descriptor(np.arange(10), p, nr, r, 5, return_spfh=True)
    k = nb.counts()[np.argsort(cloud.perm())]  # by original index
    assert (want.max(axis=1) * k > 255).sum() > 100  # (bins that do not fit a byte)
    assert np.array_equal(got, want)
    for obj in (sp, nb, cloud):
        obj.free()


@pytest.mark.parametrize("n_bins", [5, 4, 3])
def test_drop_in_fpfh_by_keypoint_index_with_long_lists(eng, O, n_bins):
    """compute_fpfh_descriptor(keypoints_indices, ...) on a clustered cloud: keypoints by index go through the matrix-core
    launch and the second launch without a selection (every keypoint looked at); 4 bins: all eight 16-bin blocks live."""
    import shot_fpfh_amd as s

    p, nr = clustered(30000, seed=8)
    r = 0.02
    kp = np.sort(np.random.default_rng(2).choice(30000, 2500, replace=False))
    got = s.compute_fpfh_descriptor(kp, p, nr, r, n_bins, verbose=False)
    want = O.compute_fpfh_descriptor(kp, p, nr, r, n_bins)
    assert close(got, want).all() and np.abs(got - want).max() < 1e-9


def test_normals_and_shot_drop_ins_on_a_clustered_cloud(eng, O):
    import shot_fpfh_amd as s
    from shot_fpfh_amd.descriptors import ShotMultiprocessor

    p, nr = clustered(30000, seed=10)
    r = 0.02
    q = p[np.sort(np.random.default_rng(5).choice(30000, 3000, replace=False))]
    pre = np.tile(np.array([[0.0, 0.0, 1.0]]), (q.shape[0], 1))
    got = s.compute_normals(q, p, radius=r, pre_computed_normals=pre)
    want = O.compute_normals(q, p, radius=r, pre_computed_normals=pre)
    assert np.abs(got - want).max() < 1e-9
    with ShotMultiprocessor(min_neighborhood_size=10, verbose=False) as sm:
        d = sm.compute_descriptor_single_scale(p, nr, q, r)
    want_d = O.shot_single_scale(p, nr, q, r, True, 10)
    assert np.abs(d - want_d).max() < 1e-9


@pytest.mark.parametrize("world,kind", [(2, "blobs"), (3, "blobs"), (2, "cap"), (3, "cap")])
def test_neighbor_mode_equals_the_unsharded_pass_on_a_clustered_cloud(eng, world, kind):
    """Blocks of a sharded job on a cloud with long lists: tail launches on slices, high-byte rows among the borrowed
    rows, every rank on the byte table with high bytes -- stitched together == one rank, bit for bit."""
    from shot_fpfh_amd.sharding import DescriptorJob

    p, nr = clustered(40000, seed=12) if kind == "blobs" else dense_cap(40000, seed=12)
    r = 0.016 if kind == "blobs" else 0.025
    one = DescriptorJob(eng, p, nr, r, n_bins=5, min_neighborhood_size=5)
    one.step()
    f1, s1, l1, rows1 = one.fpfh_out.to_host(), one.shot_out.to_host(), one.lrf_out.to_host(), one.block_original_indices()
    one.close()
    for rank in range(world):
        job = DescriptorJob(eng, p, nr, r, n_bins=5, min_neighborhood_size=5, world=world, rank=rank, spfh_exchange="neighbor",
                            emulate_peers=True)
        for _ in range(2):
            job.step()
        b, e = job.plan.block()
        assert np.array_equal(job.block_original_indices(), rows1[b:e])
        assert np.array_equal(job.fpfh_out.to_host(), f1[b:e])
        assert np.array_equal(job.shot_out.to_host(), s1[b:e])
        assert np.array_equal(job.lrf_out.to_host(), l1[b:e])
        job.close()


def test_wire_image_carries_the_high_bytes(eng):
    """A table with long lists: 32-byte packed row + 32-byte record + 128 high bytes per row travel; the image written
    into other rows reads back the same (host-staged path; the RCCL path moves the same three arrays)."""
    p, nr = dense_cap(30000, seed=6)
    r = 0.03
    cloud = eng.cloud(p, nr)
    nb = cloud.radius_search_self(r)
    sp = eng.spfh(cloud, 5, nb.max_count)
    sp.compute(nb)
    k = nb.counts()
    first = int(np.argmax(np.convolve(k > 255, np.ones(2000), "valid")))  # 2000 consecutive rows with many long lists
    img = sp.rows_image(first, first + 2000)
    assert img.size == 2000 * (32 + 32 + 128)
    high = img[2000 * 64:].reshape(2000, 128)
    long_rows = k[first:first + 2000] > 255
    assert long_rows.sum() > 200 and high[long_rows].any()
    dst = 0 if first > 4000 else first + 4000
    sp.set_rows_image(dst, dst + 2000, img)
    assert np.array_equal(sp.rows_image(dst, dst + 2000), img)
    assert np.array_equal(sp.rows_image(first, first + 2000), img)
    for obj in (sp, nb, cloud):
        obj.free()


def test_empty_block_takes_part_in_the_collective_statistics():
    """A rank whose block is empty still joins the all-reduce of the list statistics its peers issue (one-rank communicator:
    the call must neither hang nor fail)."""
    import shot_fpfh_amd as s

    e = s.Engine(0)
    try:
        e.comm_init(e.comm_unique_id(), 1, 0)
        p, nr, _ = synth_cloud(5000, 1)
        cloud = e.cloud(p, nr)
        cloud.build_grid(0.1)
        e.collective_stats(True)
        try:
            e.profile_reset()
            nb = cloud.radius_search_self(0.1, 1234, 1234)
            assert nb.m == 0 and nb.max_count == 0 and nb.max_count_all == 0
            assert e.profile_report()["c_allreduce"][0] == 1
            nb.free()
        finally:
            e.collective_stats(False)
        cloud.free()
    finally:
        e.close()


# ---- compute_normals(radius) in one sweep --------------------------------------------------------------------------------
@pytest.mark.parametrize("kind", ["uniform", "surface", "blobs", "cap"])
def test_fused_normals_sweep_is_bit_identical_to_search_plus_normals(eng, O, kind):
    """sf_normals_radius (hits reduced to the covariance in LDS, no lists) == sf_radius_search + sf_normals, bit for bit: the
    cloud's own points and coordinate queries, with and without pre_computed_normals, lists of up to 256 points (LDS list)
    and longer ones (the ring: two more sweeps); and against the oracle."""
    if kind == "uniform":
        p, _, _ = synth_cloud(40000, 21)
        r = 0.06
    elif kind == "surface":
        p, _ = config1_cloud(40000, 21)
        r = 0.04
    elif kind == "blobs":
        p, _ = clustered(40000, seed=21)
        r = 0.016
    else:
        p, _ = dense_cap(40000, seed=21)
        r = 0.025
    cloud = eng.cloud(p)
    nb = cloud.radius_search_self(r)
    cnt = nb.counts()
    if kind in ("blobs", "cap"):
        assert (cnt > 256).sum() > 100 and (cnt <= 256).sum() > 100
    want = eng.empty((cloud.n, 3))
    nb.normals(out=want)
    got = eng.empty((cloud.n, 3))
    cloud.normals_radius_self(r, got)
    assert np.array_equal(got.to_host(), want.to_host())
    nb.free()
    # a block of positions
    part = eng.empty((5000, 3))
    cloud.normals_radius_self(r, part, 12000, 17000)
    assert np.array_equal(part.to_host(), want.to_host()[12000:17000])
    # coordinate queries (off-cloud points among them), with pre_computed_normals
    rng = np.random.default_rng(2)
    q = np.vstack([p[rng.choice(p.shape[0], 3000, replace=False)], rng.random((500, 3))])
    pre = np.tile(np.array([[0.0, 0.0, 1.0]]), (q.shape[0], 1))
    nbq = cloud.radius_search(q, r)
    a = nbq.normals(pre)
    b = cloud.normals_radius(q, r, pre)
    has = nbq.counts() > 0
    nbq.free()
    assert np.array_equal(a[has], b[has])  # (queries without any neighbour: 0 / 0 on both paths)
    ok = nbq_ok = has & (np.arange(q.shape[0]) < 3000)
    wo = O.compute_normals(q[ok][:400], p, radius=r, pre_computed_normals=pre[ok][:400])
    assert np.abs(b[ok][:400] - wo).max() < 1e-9
    for obj in (want, got, part, cloud):
        obj.free()


# ---- the few lists that need more chunks than the bulk ---------------------------------------------------------------------
def test_a_launch_of_its_own_for_the_lists_that_need_more_chunks_changes_no_bit(eng, O):
    """A uniform cloud at ~40 neighbours: 99 % of the lists fit one 64-neighbour chunk, a fraction of a per cent needs two.  The
    bulk runs the 1-chunk instantiations, the rest a 4-chunk launch of the SAME forms over a selection (k*_mid) -- and every
    row equals, bit for bit, the run that serves everybody with the 2-chunk instantiations (SF_NO_MID_LAUNCH=1)."""
    from shot_fpfh_amd.sharding import DescriptorJob

    p, nr, _ = synth_cloud(60000, 31)
    r = 0.057
    outs = []
    for env in (None, "1"):
        if env:
            os.environ["SF_NO_MID_LAUNCH"] = env
        try:
            job = DescriptorJob(eng, p, nr, r, n_bins=5, min_neighborhood_size=5)
            _, rep = launches(eng, job.step)
            nrm = eng.empty((job.cloud.n, 3))
            nb = job.cloud.radius_search_self(r)
            cnt = nb.counts()
            nb.normals(out=nrm)
            nb.free()
            orig = job.block_original_indices()
            outs.append((job.fpfh_out.to_host(), job.shot_out.to_host(), job.lrf_out.to_host(), nrm.to_host(), rep))
            nrm.free()
            job.close()
        finally:
            os.environ.pop("SF_NO_MID_LAUNCH", None)
    share = (cnt > 64).mean()
    assert 0 < share <= 0.02 and cnt.max() <= 128, (share, cnt.max())
    a, b = outs
    for name in ("k5_shot_mid", "k6_spfh_mid", "k7_fpfh_mid"):
        assert a[4].get(name) == 1 and name not in b[4], (name, a[4], b[4])
    for x, y in zip(a[:4], b[:4]):
        assert np.array_equal(x, y)
    rows = np.sort(np.concatenate([np.flatnonzero(cnt > 64)[:60], np.arange(0, 60000, 1500)]))
    want = O.compute_fpfh_descriptor_sample(orig[rows], p, nr, r, 5)
    assert np.abs(a[0][rows] - want).max() < 1e-9
    want_s = O.shot_single_scale(p, nr, p[orig[rows]], r, True, 5)
    assert np.abs(a[1][rows] - want_s).max() < 1e-9


# ---- K1: the counting build ----------------------------------------------------------------------------------------------
def _grid_and_lists(eng, p, r, block, launch_names):
    cloud = eng.cloud(p)
    try:
        def go():
            if block is None:
                cloud.build_grid(r)
                return cloud.radius_search_self(r)
            cloud.build_grid(r, block=block, reach=1)
            return cloud.radius_search_self(r, *block)

        nb, rep = launches(eng, go)
        launch_names.append(set(rep))
        off, idx = nb.export()
        lo, hi = (0, cloud.n) if block is None else block
        perm = cloud.perm()[lo:hi]
        nb.free()
        return perm, off, idx
    finally:
        cloud.free()


@pytest.mark.parametrize("kind", ["uniform", "surface", "duplicates", "clustered"])
@pytest.mark.parametrize("block", [None, (9000, 23000)])
def test_counting_build_gives_the_order_of_the_stable_sort(eng, kind, block):
    """K1 orders a dense-enough grid through its own cell table (count, scan, place, settle) instead of a radix sort of the
    cell ids: the cell-sorted order (ties: ascending internal index), hence every list downstream, is the same bit for bit.
    Duplicated points put many points into one cell (the rank-within-the-cell loop); a clustered cloud's grid is sparse and
    stays on the radix sort."""
    if kind == "uniform":
        p, _, _ = synth_cloud(40000, 5)
        r = 0.05
    elif kind == "surface":
        p, _ = config1_cloud(40000, 5)
        r = 0.04
    elif kind == "duplicates":
        q, _, _ = synth_cloud(5000, 5)
        p = np.concatenate([q] * 8)[np.random.default_rng(1).permutation(40000)]
        r = 0.06
    else:
        p, _ = clustered(40000)
        r = 0.02
    names = []
    os.environ.pop("SF_K1_RADIX", None)
    a = _grid_and_lists(eng, p, r, block, names)
    os.environ["SF_K1_RADIX"] = "1"
    try:
        b = _grid_and_lists(eng, p, r, block, names)
    finally:
        del os.environ["SF_K1_RADIX"]
    assert "k1_radix_sort" in names[1] and "k1_cell_settle" not in names[1], names[1]
    if kind != "clustered":
        assert "k1_cell_settle" in names[0] and "k1_radix_sort" not in names[0], names[0]
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_layer_bounds_on_the_host_equal_the_device_search(eng):
    """A block build looks the z-layers' first points up in a host copy of the sorted z coordinates (no launch, no
    read-back); the wave-parallel device search it replaces stays for very large clouds: same table, same slab, same lists."""
    p, _ = config1_cloud(50000, 8)
    r, block = 0.035, (12000, 31000)
    out = []
    for device in (False, True):
        if device:
            os.environ["SF_K1_DEVICE_BOUNDS"] = "1"
        try:
            cloud = eng.cloud(p)
            nb, rep = launches(eng, lambda: (cloud.build_grid(r, block=block, reach=1), cloud.radius_search_self(r, *block))[1])
            off, idx = nb.export()
            out.append((cloud.layer_table().copy(), cloud.perm()[block[0]:block[1]].copy(), off, idx, set(rep)))
            nb.free()
            cloud.free()
        finally:
            os.environ.pop("SF_K1_DEVICE_BOUNDS", None)
    assert "k1_layer_bounds" not in out[0][4] and "k1_layer_bounds" in out[1][4]
    for a, b in zip(out[0][:4], out[1][:4]):
        assert np.array_equal(a, b)


def test_full_form_of_k7_stays_within_reach_of_the_sparse_form(eng, monkeypatch):
    """K7's full form (tables with more than two live 16-bin blocks: even bin counts, large radii, clouds in millimetres) and
    its sparse-block form give the same rows; this holds the full form to a TIME as well: one of its instantiations, held to 64
    registers, spilled 360 bytes and ran 24 ms per 1M keypoints instead of 1.5 -- for a whole round, with every row correct."""
    p, nr, _ = synth_cloud(300000, 6)
    r = 0.044  # ~ 105 neighbours, lists up to ~160: two chunks for most, the three-chunk instantiation for the launch
    out = {}
    for dense in (False, True):
        if dense:
            monkeypatch.setenv("SF_FPFH_DENSE", "1")
        else:
            monkeypatch.delenv("SF_FPFH_DENSE", raising=False)
        cloud = eng.cloud(p, nr)
        nb = cloud.radius_search_self(r)
        spfh = eng.spfh(cloud, 5, nb.max_count)
        spfh.compute(nb)
        dst = eng.empty((cloud.n, 125))
        spfh.fpfh(nb, None, out=dst)  # (first call: reads the block mask back)
        _, rep = launches(eng, lambda: spfh.fpfh(nb, None, out=dst))
        eng.sync()
        eng.profile_reset()
        eng.profile(True)
        for _ in range(3):
            spfh.fpfh(nb, None, out=dst)
        eng.sync()
        eng.profile(False)
        ms = sum(v[1] for k, v in eng.profile_report().items() if k.startswith("k7_")) / 3
        out[dense] = (ms, dst.rows_to_host(0, 2000).copy(), nb.max_count)
        for o in (dst, spfh, nb, cloud):
            o.free()
    monkeypatch.delenv("SF_FPFH_DENSE", raising=False)
    assert out[False][2] > 128  # (the launch is the three-chunk instantiation)
    assert np.array_equal(out[False][1], out[True][1])
    assert out[True][0] < 3.0 * out[False][0], (out[True][0], out[False][0])


def test_cloud_of_a_subset_equals_the_cloud_of_the_gathered_rows(eng):
    """Cloud(points, normals, subset=keep) -- the rows gathered on the device -- is the cloud of points[keep], normals[keep]:
    same lists, same descriptors, bit for bit; an empty subset is an empty cloud; indices outside the array are refused."""
    from shot_fpfh_amd.engine import Cloud

    p, nr, _ = synth_cloud(30000, 9)
    keep = np.random.default_rng(2).permutation(30000)[:17000].astype(np.int64)  # (any order, as grid_subsampling's)
    kp = p[keep[:1500]]
    outs = []
    for cloud in (Cloud(eng, p[keep], nr[keep]), Cloud(eng, p, nr, subset=keep)):
        nb = cloud.radius_search(kp, 0.06)
        off, idx = nb.export()
        outs.append((off, idx, nb.shot_single_scale(True, 5).copy()))
        nb.free()
        cloud.free()
    for a, b in zip(*outs):
        assert np.array_equal(a, b)
    empty = Cloud(eng, p, nr, subset=np.zeros(0, dtype=np.int64))
    assert empty.n == 0
    empty.free()
    with pytest.raises(ValueError):
        Cloud(eng, p, nr, subset=np.array([0, 30000]))
