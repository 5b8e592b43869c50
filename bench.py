#!/usr/bin/env python3
"""bench.py -- descriptors/s (SHOT + FPFH) of the hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json metric "descriptors/sec (SHOT+FPFH) on 1M-pt cloud"): a synthetic uniform
cloud of 1M points PER GPU (seed 3, float32-grid coordinates, random unit normals), every point a
keypoint, radius 0.03 at N=1 (k ~ 106-113 neighbours); for N>1 the cloud has N*1M points and the radius
shrinks by N^(-1/3) so the per-GPU work is fixed (weak scaling, BASELINE config 5 at N=8).  One step =
one pass of the path with inputs resident in HBM: K1 grid build, K2 radius search, K6 SPFH (which also accumulates
the SHOT frame moments from the neighbours it gathers), K7 FPFH (1M x 125 float64 out), K4 frame eigen-solves, K5 SHOT
(1M x 352 float64 out); outputs stay in HBM.
value = 2 * (N*1M) descriptors / step time (max over ranks).

Prints ONE JSON line (rank 0) with the contract fields plus `roofline` (dominant kernel, algorithmic
bytes / HIP-event kernel time, measured live on the engine's own stream) and `cpu_baseline` (the CPU
oracle, single thread, on a bounded sample of the same workload; N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
# algorithmic bytes per unit at float64 API widths (DESIGN.md "Measurement")
ALG_BYTES = {
    "k6_spfh": 48 + 1000,  # per cloud point: xyz+normal in, 125 x 8 B SPFH row out
    "k7_fpfh": 1000 + 1000,  # per descriptor: own SPFH row in, 125 x 8 B FPFH row out
    "k5_shot": 2816 + 24 + 48,  # per descriptor: 352 x 8 B row out, keypoint, cloud share (all points keypoints)
    "k4_shot_lrf": 24 + 72,
    "k4_lrf_from_cov": 48 + 72,
    "k2_radius_count": 24 + 4,
    "k2_radius_fill": 24,  # + 4 B per pair, added below
}


def make_cloud(n: int, seed: int):
    rng = np.random.default_rng(seed)
    p = rng.random((n, 3), dtype=np.float32).astype(np.float64)
    nr = rng.standard_normal((n, 3))
    nr /= np.linalg.norm(nr, axis=1)[:, None]
    return p, nr


def cpu_baseline(points_per_gpu: int, radius: float) -> dict:
    """The CPU oracle (scalar C port of the reference algorithm, one thread) on a bounded sample:
    a 150k-point cloud at the SAME point density per radius-ball (radius scaled by (n/150k)^(1/3)), all
    points keypoints, FPFH + SHOT.  ~10 s of CPU work."""
    from oracle import oracle as O

    ns = min(150000, points_per_gpu)
    r = radius * (points_per_gpu / ns) ** (1.0 / 3.0)
    p, nr = make_cloud(ns, 33)
    O.lib()
    t0 = time.perf_counter()
    O.compute_fpfh_descriptor(np.arange(ns), p, nr, r, 5)
    t1 = time.perf_counter()
    O.shot_single_scale(p, nr, p, r, True, 10)
    t2 = time.perf_counter()
    return {
        "value": 2 * ns / (t2 - t0),
        "unit": "descriptors/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{ns}-pt uniform cloud, all points keypoints, r={r:.4f} (same neighbours per ball as the GPU "
        f"workload), FPFH 5 bins {t1 - t0:.1f}s + SHOT {t2 - t1:.1f}s, oracle/shot_fpfh_oracle.c single thread",
        "host_cores_available": os.cpu_count(),
    }


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--points-per-gpu", type=int, default=1_000_000)
    ap.add_argument("--radius", type=float, default=0.03)
    ap.add_argument("--spfh-exchange", choices=["halo", "allgather"], default="halo")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--only", choices=["both", "fpfh", "shot"], default="both")
    ap.add_argument("--overlap", action="store_true",
                    help="run the FPFH and SHOT chains on two HIP streams (faster; per-kernel times then overlap)")
    ap.add_argument("--emulate-rank", type=int, default=None, metavar="R",
                    help="single process, no rendezvous: run rank R's share of a --gpus N job on this GPU (what one "
                         "GPU of an N-GPU node does per step; for sizing the sharded path on a 1-GPU box)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    emulated = args.emulate_rank is not None
    if emulated:
        world, rank, local_rank = args.gpus, args.emulate_rank, 0
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dist = None
    if world > 1 and not emulated:
        import torch.distributed as dist  # control plane only: rendezvous, barrier, max-reduce of the timing

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)

    import shot_fpfh_amd as s
    from shot_fpfh_amd.sharding import DescriptorJob

    from shot_fpfh_amd import _ffi

    n_dev = max(_ffi.load().sf_device_count(), 1)
    if local_rank >= n_dev and rank == 0:
        print(f"# warning: {world} ranks share {n_dev} GPU(s) (functional test only, timings are not a scaling result)", file=sys.stderr)
    eng = s.Engine(local_rank % n_dev)
    if world > 1 and args.spfh_exchange == "allgather" and not emulated:
        ids = [eng.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        eng.comm_init(ids[0], world, rank)

    n_total = args.points_per_gpu * world
    radius = args.radius * world ** (-1.0 / 3.0)
    points, normals = make_cloud(n_total, 3)
    job = DescriptorJob(eng, points, normals, radius, n_bins=5, normalize=True, min_neighborhood_size=10, world=world,
                        rank=rank, spfh_exchange=args.spfh_exchange, do_fpfh=args.only in ("both", "fpfh"),
                        do_shot=args.only in ("both", "shot"), overlap_chains=args.overlap)
    del points, normals

    def barrier():
        eng.sync()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        job.step()
    barrier()
    eng.profile_reset()
    eng.profile(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        job.step()
    barrier()
    elapsed = time.perf_counter() - t0
    eng.profile(False)
    if dist is not None:
        import torch

        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    kinds = (1 if job.do_fpfh else 0) + (1 if job.do_shot else 0)
    n_desc = kinds * n_total
    ms_per_step = 1000.0 * elapsed / args.steps
    value = n_desc / (elapsed / args.steps)

    if rank == 0 or emulated:
        rep = eng.profile_report()
        kern = {k: (v[0], v[1] / max(v[0], 1)) for k, v in rep.items() if v[0] > 0 and v[1] > 0}
        per_step_ms = {k: rep[k][1] / args.steps for k in kern}
        dom = max((k for k in kern if k in ALG_BYTES), key=lambda k: rep[k][1])
        launches, avg_ms = kern[dom]
        units = job.m  # descriptors of this rank's block per launch (halo SPFH rows are extra work, not counted)
        alg_bytes = ALG_BYTES[dom] * units + (4 * job.last_pairs if dom == "k2_radius_fill" else 0)
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(dom)
            except Exception:
                traffic = None
        out = {
            "metric": "descriptors/sec (SHOT+FPFH) on 1M-pt cloud",
            "value": value,
            "unit": "descriptors/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"uniform cloud, {args.points_per_gpu} points per GPU ({n_total} total), all points "
                f"keypoints, radius {radius:.5f}, {'FPFH(5 bins)' if job.do_fpfh else ''}"
                f"{'+' if kinds == 2 else ''}{'SHOT(352, normalize, min_nb 10)' if job.do_shot else ''}, "
                f"mean neighbourhood {job.last_pairs / max(job.plan.end - job.plan.begin, 1):.1f}",
                "sharding": f"query blocks over {world} GPU(s), cloud replicated, SPFH {args.spfh_exchange}",
                "points_per_gpu": args.points_per_gpu,
            },
            "kernels_ms_per_step": {k: round(v, 4) for k, v in sorted(per_step_ms.items())},
            "roofline": {
                "kernel": dom,
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "avg_launch_ms": avg_ms,
                "algorithmic_bytes_per_launch": alg_bytes,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.points_per_gpu, args.radius)
            out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    job.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
