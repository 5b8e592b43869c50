"""The N>1 path on CPU: two gloo ranks run shot_fpfh_amd.sharding.DescriptorJob (block planning, halo
range, list slicing, SPFH exchange) over an oracle-backed engine stand-in; the stitched result must
equal the single-rank result and the oracle, for both SPFH exchange modes.  The GPU kernels themselves
are covered by `-m gpu` tests; this covers the sharding logic the driver's 2/4/8-GPU runs depend on."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, synth_cloud


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("mode,world", [("neighbor", 2), ("neighbor", 3), ("neighbor", 5), ("halo", 2), ("allgather", 2)])
def test_two_rank_gloo_equals_single_rank_and_oracle(tmp_path, mode, world):
    from fake_engine import FakeEngine
    from oracle import oracle as O
    from shot_fpfh_amd.sharding import DescriptorJob

    out = str(tmp_path / "stitched.npz")
    port = free_port()
    procs = []
    for rank in range(world):  # (three ranks: the middle one borrows from and lends to both sides; five: blocks thinner than a z-layer,
                               #  a halo that reaches past the adjacent rank)
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   LOCAL_RANK=str(rank), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_gloo_worker.py"), out, mode], env=env))
    for p in procs:
        assert p.wait(timeout=600) == 0
    got = np.load(out)
    assert (got["seen"] == 1).all()  # every point in exactly one block
    assert not np.isnan(got["fpfh"]).any() and not np.isnan(got["shot"]).any()  # no poisoned SPFH row was read

    p, nr, _ = synth_cloud(1500, 41)
    single = DescriptorJob(FakeEngine(), p, nr, 0.15, n_bins=5, normalize=True, min_neighborhood_size=5)
    single.step()
    rows = single.block_original_indices()
    f1, s1 = np.zeros((1500, 125)), np.zeros((1500, 352))
    f1[rows], s1[rows] = single.fpfh_out.to_host(), single.shot_out.to_host()
    assert np.array_equal(got["fpfh"], f1) and np.array_equal(got["shot"], s1)  # bit-identical to 1 rank

    fo = O.compute_fpfh_descriptor(np.arange(1500), p, nr, 0.15, 5)
    so = O.shot_single_scale(p, nr, p, 0.15, True, 5)
    assert np.abs(got["fpfh"] - fo).max() < 1e-9 and np.abs(got["shot"] - so).max() < 1e-12


@pytest.mark.parametrize("mode", ["allgather_long", "neighbor_long"])
def test_ranks_size_their_spfh_tables_by_the_longest_list_of_any_rank(tmp_path, mode):
    """Only rank 0's slab holds lists of more than 255 points.  Ranks that exchange table rows must all hold the same storage
    (bytes + high-byte rows here): DescriptorJob folds the longest list over the ranks before it creates the table -- in
    step() (all-gather mode) as in _step_neighbor -- and the all-gather fails loudly when the storages differ (the stand-in's
    check mirrors sf_spfh_allgather's format word)."""
    from conftest import long_list_cloud
    from fake_engine import FakeEngine
    from shot_fpfh_amd.sharding import DescriptorJob

    out = str(tmp_path / "stitched.npz")
    port = free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK=str(rank),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_gloo_worker.py"), out, mode], env=env))
    for p in procs:
        assert p.wait(timeout=600) == 0
    got = np.load(out)
    p, nr = long_list_cloud()
    single = DescriptorJob(FakeEngine(), p, nr, 0.15, n_bins=5, normalize=True, min_neighborhood_size=5)
    single.step()
    assert single.spfh.storage == 3  # (the cloud does have lists above 255 points)
    rows = single.block_original_indices()
    f1, s1 = np.zeros((p.shape[0], 125)), np.zeros((p.shape[0], 352))
    f1[rows], s1[rows] = single.fpfh_out.to_host(), single.shot_out.to_host()
    assert (got["seen"] == 1).all() and np.array_equal(got["fpfh"], f1) and np.array_equal(got["shot"], s1)


@pytest.mark.parametrize("world", [2, 3, 5, 8])
def test_exchange_plan_covers_every_halo_and_pairs_up(world):
    """sharding.exchange_plan from a layer table alone: for every rank, block + received rows = its halo exactly; what
    r sends to p is what p receives from r; interior keypoints reach no foreign row.  Unaligned blocks, blocks thinner
    than a layer, empty layers and empty blocks included."""
    from shot_fpfh_amd.sharding import ShardPlan, exchange_plan

    rng = np.random.default_rng(world)
    for case in range(40):
        nl = int(rng.integers(1, 30))
        pop = rng.integers(0, 50, nl) * (rng.random(nl) < 0.8)
        if case % 7 == 0:
            pop[:] = 0
            pop[rng.integers(0, nl)] = rng.integers(1, 9)  # almost every block empty
        if pop.sum() == 0:
            pop[0] = 1
        first = np.concatenate([[0], np.cumsum(pop)]).astype(np.int64)
        n = int(first[-1])
        layer = np.repeat(np.arange(nl), pop)  # layer of every position
        plans = [exchange_plan(first, n, world, r) for r in range(world)]
        for r, xp in enumerate(plans):
            b, e = ShardPlan(n, world, r).block()
            need = np.zeros(n, bool)
            for q in range(b, e):
                need |= np.abs(layer - layer[q]) <= 1
            hb, he = xp.halo
            assert need[hb:he].all() and not need[:hb].any() and not need[he:].any() or b == e
            have = np.zeros(n, bool)
            have[b:e] = True
            for p, sb, se, rb, re in xp.ops:
                assert p != r and not have[rb:re].any()
                have[rb:re] = True
                pb, pe = ShardPlan(n, world, p).block()
                assert pb <= rb and re <= pe and b <= sb and se <= e
                back = [o for o in plans[p].ops if o[0] == r]
                assert len(back) == 1 and back[0][1:3] == (rb, re) and back[0][3:5] == (sb, se)
            assert np.array_equal(have, need | have) and (have[hb:he].all() if b < e else True)
            i0, i1 = xp.interior
            assert b <= i0 <= i1 <= e
            for q in range(i0, i1):
                reach = np.flatnonzero(np.abs(layer - layer[q]) <= 1)
                assert reach[0] >= b and reach[-1] < e
            # the interior is maximal: the keypoints just outside it do reach a foreign row (or are outside the block)
            for q in ([i0 - 1] if i0 > b and i1 > i0 else []) + ([i1] if i1 < e and i1 > i0 else []):
                reach = np.flatnonzero(np.abs(layer - layer[q]) <= 1)
                assert reach[0] < b or reach[-1] >= e


@pytest.mark.parametrize("chunks", [1, 3])
def test_two_rank_gloo_sharded_matching_equals_basic_matching(tmp_path, chunks):
    """MatchJob over two gloo ranks: reference rows all-gathered -- at once, or in three chunks with K8 on chunk c while chunk
    c + 1 is gathered and the chunks' arg-mins folded (MatchJob(chunks=3)) -- each rank matches its scan block; the union of
    the per-rank results must equal basic_matching on the whole sets (oracle restatement)."""
    from oracle import oracle as O

    out = str(tmp_path / "match.npz")
    port = free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   LOCAL_RANK=str(rank), OMP_NUM_THREADS="1", SF_TEST_CHUNKS=str(chunks))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_gloo_worker.py"), out, "match"], env=env))
    for p in procs:
        assert p.wait(timeout=600) == 0
    got = np.load(out)
    s, r = O.basic_matching(got["a"], got["b"])
    assert np.array_equal(got["s"], s) and np.array_equal(got["r"], r)


@pytest.mark.parametrize("chunks", [1, 4])
def test_two_rank_gloo_subset_matching_by_label(tmp_path, chunks):
    """SubsetMatchJob (the tail of BASELINE config 5) over two gloo ranks: a keypoint subset picked out of every rank's
    blocks, reference subset + labels all-gathered, sharded K8; the label pairs must be those of basic_matching on the
    subset rows taken in label order."""
    from oracle import oracle as O

    out = str(tmp_path / "subset.npz")
    port = free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   LOCAL_RANK=str(rank), OMP_NUM_THREADS="1", SF_TEST_CHUNKS=str(chunks))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_gloo_worker.py"), out, "subset"], env=env))
    for p in procs:
        assert p.wait(timeout=600) == 0
    got = np.load(out)
    scan, ref, perm, sub = got["scan"], got["ref"], got["perm"], got["in_subset"]
    s_lab = np.flatnonzero(sub)                       # scan subset, label = scan row
    r_rows = np.flatnonzero(sub[perm])                # reference rows whose label is in the subset
    si, ri = O.basic_matching(scan[s_lab], ref[r_rows])
    want = dict(zip(s_lab[si].tolist(), perm[r_rows[ri]].tolist()))
    have = dict(zip(got["s"].tolist(), got["r"].tolist()))
    assert have == want
    assert np.mean([k == v for k, v in have.items()]) > 0.9  # and the matches do recover the correspondence


@pytest.mark.parametrize("world,chunks", [(2, 1), (3, 1), (3, 5), (4, 2)])
def test_gloo_sharded_match_descriptors_with_filters_and_reciprocity(tmp_path, world, chunks):
    """MatchJob.matches(filter_callback, filter_nonreciprocal, n_min_matches) over gloo ranks on the reference's golden
    inputs (match_300.npz): the union of the per-rank results, mapped back through the non-empty-row numbering, equals
    the reference's match_descriptors outputs -- the reciprocity fallback (n_min_matches = 10^6) and a filter that needs
    every rank's distances (quantile) included."""
    from conftest import load_golden
    from oracle import oracle as O
    from shot_fpfh_amd.matching.filters import quantile_filter

    out = str(tmp_path / "recip.npz")
    port = free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   LOCAL_RANK=str(rank), OMP_NUM_THREADS="1", SF_TEST_CHUNKS=str(chunks))  # (chunks > 1: the column arg-min re-assembles the set)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_gloo_worker.py"), out, "reciprocal"], env=env))
    for p in procs:
        assert p.wait(timeout=600) == 0
    got, g = np.load(out), load_golden("match_300.npz")
    for name in ("rec", "recbig", "thr"):  # the reference's own outputs
        assert np.array_equal(got[name + "_s"], g[name + "_s"]) and np.array_equal(got[name + "_r"], g[name + "_r"]), name
    s, r = O.match_descriptors(g["scan"], g["ref"], quantile_filter, quantiles=(0.2, 0.7))
    assert np.array_equal(got["quant_s"], s) and np.array_equal(got["quant_r"], r)
    s, r = O.match_descriptors(g["scan"], g["ref"], quantile_filter, filter_nonreciprocal=True, n_min_matches=50, quantiles=(0.2, 0.7))
    assert np.array_equal(got["quantrec_s"], s) and np.array_equal(got["quantrec_r"], r)
