"""GPU parity tests added in round 5 (all through the C ABI).

  * K7 on the matrix cores for lists of MORE than 255 points (k_fpfh_mcl): against the oracle, against the vector-ALU form it
    replaces (SF_FPFH_TAIL_VECTOR=1), sparse-block form == full form bit for bit, consistent normals (high bytes really used);
  * K2's lists leave through an LDS ring (whole 256-byte runs): neighbour sets bit-exact at slot sizes around the ring's size.
"""
import os

import numpy as np
import pytest

from conftest import family, synth_cloud

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
    looked for among the keypoints): rows against the oracle and against the vector-ALU form with exact sums."""
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
    monkeypatch.delenv("SF_FPFH_TAIL_VECTOR", raising=False)
    got = s.compute_fpfh_descriptor(kp, p, nr, r, n_bins, verbose=False)
    monkeypatch.setenv("SF_FPFH_TAIL_VECTOR", "1")
    vec = s.compute_fpfh_descriptor(kp, p, nr, r, n_bins, verbose=False)
    monkeypatch.delenv("SF_FPFH_TAIL_VECTOR", raising=False)
    assert np.abs(got - vec).max() <= 1e-12 * max(1.0, np.abs(vec).max()), np.abs(got - vec).max()
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


# ---- K5 for lists of 256 .. 512 points held in registers (k_shot_wide) ---------------------------------------------------------
@pytest.mark.parametrize("kind", ["uniform", "surface"])
def test_k5_register_held_form_equals_the_streaming_form_bit_for_bit(eng, O, monkeypatch, kind):
    """Lists of 256 .. 512 points: the form that fetches the list once and runs its passes from registers (k_shot_wide) against
    the streaming form (SF_SHOT_NO_WIDE=1, k_shot_long) -- same functions, same LDS operations in the same order: equal bits --
    and against the oracle; fused frame (single scale: votes in the kernel) and given frames (sf_shot); lists on both sides of
    the 255 / 256 and 512 / 513 boundaries in one launch; a gate that zeroes some rows."""
    from conftest import config1_cloud
    from shot_fpfh_amd.descriptors import ShotMultiprocessor

    if kind == "uniform":
        p, nr, _ = synth_cloud(40000, 31)
        r = 0.145
    else:
        p, nr = config1_cloud(40000, 31)
        r = 0.11
    cloud = eng.cloud(p)
    nb = cloud.radius_search_self(r)
    cnt = nb.counts()[np.argsort(cloud.perm())]
    nb.free()
    cloud.free()
    assert cnt.min() < 256 and ((cnt > 255) & (cnt <= 512)).sum() > 2000 and cnt.max() > 512, (cnt.min(), cnt.max())
    order = np.argsort(cnt)
    picks = [order[:20], order[-40:]]
    for edge in (255, 256, 257, 320, 384, 385, 448, 511, 512, 513):
        lo = np.searchsorted(cnt[order], edge - 1)
        picks.append(order[lo:lo + 10])
    kp = np.unique(np.concatenate(picks + [np.random.default_rng(5).choice(p.shape[0], 300, replace=False)]))
    res = {}
    for mode in ("wide", "stream"):
        if mode == "stream":
            monkeypatch.setenv("SF_SHOT_NO_WIDE", "1")
        else:
            monkeypatch.delenv("SF_SHOT_NO_WIDE", raising=False)
        with ShotMultiprocessor(normalize=True, min_neighborhood_size=300, verbose=False) as sm:
            d = sm.compute_descriptor_single_scale(p, nr, p[kp], r)
            lrf = sm.compute_local_rf(p[kp], None, p, r)
            d2 = sm.compute_descriptor(p[kp], nr, None, lrf, p, r)
            full = sm.compute_descriptor_single_scale(p, nr, p, r)
        res[mode] = (d, d2, full)
    monkeypatch.delenv("SF_SHOT_NO_WIDE", raising=False)
    for a, b in zip(res["wide"], res["stream"]):
        assert np.array_equal(a, b)
    d, d2, full = res["wide"]
    assert np.array_equal(full[kp], d)  # all points keypoints (launch over the selection of long lists) == keypoint subset
    assert (np.abs(d).sum(axis=1) == 0).any() and (np.abs(d).sum(axis=1) > 0).any()  # (the gate at 300 zeroes the short lists)
    sub = np.arange(0, kp.size, max(1, kp.size // 100))
    want = O.shot_single_scale(p, nr, p[kp[sub]], r, True, 300)
    assert np.abs(d[sub] - want).max() < 1e-9, np.abs(d[sub] - want).max()
    assert np.abs(d2 - d).max() < 1e-12
