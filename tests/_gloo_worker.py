"""Worker of tests/test_multi_gpu_gloo.py: one rank of a world_size-2 gloo job running DescriptorJob on
the oracle-backed FakeEngine, then handing its block to rank 0."""
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from conftest import synth_cloud  # noqa: E402
from fake_engine import FakeEngine  # noqa: E402
from shot_fpfh_amd.sharding import DescriptorJob  # noqa: E402

CHUNKS = int(os.environ.get("SF_TEST_CHUNKS", "1"))  # pieces of the reference all-gather (MatchJob(chunks=...))


def match_main(out_path, rank, world):
    """Sharded basic_matching: each rank holds a block of scan and of ref rows; ref rows are all-gathered."""
    from shot_fpfh_amd.sharding import MatchJob, ShardPlan

    rng = np.random.default_rng(91)
    n_ref = 277 if CHUNKS == 1 else 64 * world * 3 - 1  # (the streamed exchange needs whole 64-row tiles per rank: 192 rows each, the last rank one short)
    a = rng.random((301, 40)) * (rng.random((301, 40)) < 0.4)
    b = a[rng.permutation(301)][:n_ref] + 0.01 * rng.standard_normal((n_ref, 40)) if n_ref <= 301 else None
    if b is None:
        b = np.vstack([a[rng.permutation(301)], a[rng.permutation(301)]])[:n_ref] + 0.01 * rng.standard_normal((n_ref, 40))
    a[[0, 150, 300]] = 0.0
    b[[5, n_ref - 1]] = 0.0
    eng = FakeEngine()
    job = MatchJob(eng, 40, 301, n_ref, world, rank, chunks=CHUNKS)
    assert (job.chunks > 1) == (CHUNKS > 1), (job.chunks, CHUNKS)
    sb, se = ShardPlan(301, world, rank).block()
    rb, re = ShardPlan(n_ref, world, rank).block()
    job.run(eng.empty((se - sb, 40)).from_host(a[sb:se]), eng.empty((max(re - rb, 1), 40)).from_host(b[rb:re] if re > rb else 0.0))
    gathered = [None] * world
    dist.all_gather_object(gathered, job.matches())
    if rank == 0:
        np.savez(out_path, s=np.concatenate([g[0] for g in gathered]), r=np.concatenate([g[1] for g in gathered]), a=a, b=b)
    dist.barrier()
    dist.destroy_process_group()


def reciprocal_main(out_path, rank, world):
    """match_descriptors with filters and the reciprocity test on sharded rows: the reference's own golden inputs."""
    from conftest import load_golden
    from shot_fpfh_amd.matching.filters import quantile_filter, threshold_filter
    from shot_fpfh_amd.sharding import MatchJob, ShardPlan

    g = load_golden("match_300.npz")
    a, b = g["scan"], g["ref"]
    eng = FakeEngine()
    job = MatchJob(eng, a.shape[1], a.shape[0], b.shape[0], world, rank, chunks=CHUNKS)
    sb, se = ShardPlan(a.shape[0], world, rank).block()
    rb, re = ShardPlan(b.shape[0], world, rank).block()
    job.run(eng.empty((se - sb, a.shape[1])).from_host(a[sb:se]), eng.empty((max(re - rb, 1), b.shape[1])).from_host(b[rb:re] if re > rb else 0.0))
    res = {}
    for name, kw in (("rec", dict(filter_nonreciprocal=True, n_min_matches=100)),
                     ("recbig", dict(filter_nonreciprocal=True, n_min_matches=10**6)),
                     ("thr", dict(filter_callback=threshold_filter, threshold_multiplier=10)),
                     ("quant", dict(filter_callback=quantile_filter, quantiles=(0.2, 0.7))),
                     ("quantrec", dict(filter_callback=quantile_filter, filter_nonreciprocal=True, n_min_matches=50, quantiles=(0.2, 0.7)))):
        gathered = [None] * world
        dist.all_gather_object(gathered, job.matches(**kw))
        res[name + "_s"] = np.concatenate([x[0] for x in gathered])
        res[name + "_r"] = np.concatenate([x[1] for x in gathered])
    if rank == 0:
        np.savez(out_path, **res)
    dist.barrier()
    dist.destroy_process_group()


def subset_main(out_path, rank, world):
    """Config-5 tail on two ranks: every rank owns a block of scan and of reference descriptors; a keypoint subset
    (chosen by label) is gathered out of both, the reference part all-gathered with its labels, and matched."""
    from shot_fpfh_amd.sharding import ShardPlan, SubsetMatchJob

    rng = np.random.default_rng(17)
    n, d = 400, 24
    scan = rng.random((n, d)) * (rng.random((n, d)) < 0.5)
    perm = rng.permutation(n)
    ref = scan[perm] + 0.004 * rng.standard_normal((n, d))  # reference row j corresponds to scan row perm[j]
    scan[[3, 77]] = 0.0
    ref[[10]] = 0.0
    in_subset = rng.random(n) < 0.3  # by SCAN label
    eng = FakeEngine()
    sb, se = ShardPlan(n, world, rank).block()
    s_sel = np.flatnonzero(in_subset[sb:se])
    r_lab_block = perm[sb:se]
    r_sel = np.flatnonzero(in_subset[r_lab_block])
    rows = int(in_subset.sum())  # generous per-rank capacity: exercises the zero-row padding
    if CHUNKS > 1:
        rows = -(-rows // 64) * 64  # (whole 64-row tiles per rank: the streamed exchange)
    job = SubsetMatchJob(eng, d, rows, world, rank, chunks=CHUNKS)
    assert (job.job.chunks > 1) == (CHUNKS > 1)
    job.select(eng.empty((se - sb, d)).from_host(scan[sb:se]), s_sel, s_sel + sb,
               eng.empty((se - sb, d)).from_host(ref[sb:se]), r_sel, r_lab_block[r_sel])
    job.run()
    gathered = [None] * world
    dist.all_gather_object(gathered, job.matches())
    all_s, all_r = job.gather_matches()  # the same pairs through the engine's own all-gather, on every rank
    assert np.array_equal(all_s, np.concatenate([g[0] for g in gathered]))
    assert np.array_equal(all_r, np.concatenate([g[1] for g in gathered]))
    if rank == 0:
        np.savez(out_path, s=np.concatenate([g[0] for g in gathered]), r=np.concatenate([g[1] for g in gathered]),
                 scan=scan, ref=ref, perm=perm, in_subset=in_subset)
    dist.barrier()
    dist.destroy_process_group()


def main():
    out_path, mode = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if mode == "match":
        return match_main(out_path, rank, world)
    if mode == "subset":
        return subset_main(out_path, rank, world)
    if mode == "reciprocal":
        return reciprocal_main(out_path, rank, world)
    p, nr, _ = synth_cloud(1500, 41)
    if mode.endswith("_long"):  # one dense blob at low z: only the first rank sees lists of more than 255 points
        from conftest import long_list_cloud

        p, nr = long_list_cloud()
        mode = mode[: -len("_long")]
    job = DescriptorJob(FakeEngine(), p, nr, 0.15, n_bins=5, normalize=True, min_neighborhood_size=5, world=world,
                        rank=rank, spfh_exchange=mode)
    job.step()
    mine = (job.block_original_indices(), job.fpfh_out.to_host(), job.shot_out.to_host())
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    if rank == 0:
        n = p.shape[0]
        fpfh, shot, seen = np.full((n, 125), np.nan), np.full((n, 352), np.nan), np.zeros(n, dtype=int)
        for rows, f, s in gathered:
            fpfh[rows], shot[rows] = f, s
            seen[rows] += 1
        np.savez(out_path, fpfh=fpfh, shot=shot, seen=seen)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
