#!/usr/bin/env python3
"""bench.py -- descriptors/s (SHOT + FPFH) of the hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Run directly with N > 1 (no WORLD_SIZE in the environment) it starts N child processes itself, one per GPU, BEFORE
touching the GPU; under a launcher (torchrun: RANK / LOCAL_RANK / WORLD_SIZE set) it is one of the ranks.

Workload (BASELINE.json metric "descriptors/sec (SHOT+FPFH) on 1M-pt cloud"): a synthetic uniform cloud of 1M points
PER GPU (seed 3, float32-grid coordinates, random unit normals), every point a keypoint, radius 0.03 at N = 1 (k ~ 110
neighbours); for N > 1 the cloud has N x 1M points and the radius shrinks by N^(-1/3), so the per-GPU work is fixed
(weak scaling; BASELINE config 5 at N = 8).

Timed region (`value`): K steps, each ONE pass of the descriptor path over this rank's block with inputs resident in
HBM -- K1 grid build, K2 radius search, K6 SPFH (+ the SHOT frame moments), K7 FPFH (rows x 125 float64 out), K4 frame
eigen-solves, K5 SHOT (rows x 352 float64 out); outputs stay in HBM.  value = 2 x (N x 1M) descriptors / step time
(max over ranks).  Same definition at every N, so the driver's scaling efficiency compares like with like.

After the timed region the SAME process measures the rest of the path north_star names and reports it under extra keys:
  * `exchange_match`  -- BASELINE config 5's tail: the C4 partner cloud's SHOT rows (one more descriptor pass), a
    keypoint subset of both descriptor sets gathered per rank, ONE RCCL all-gather of the reference subset rows over
    xGMI (ncclAllGather is executed at N = 1 too, on a one-rank communicator), the sharded K8 brute-force L2 matching,
    and the fraction of matches that recover the true correspondence;
  * `ransac`          -- (N = 1) K9 at 10^4 draws x 10^6 matches through ransac_on_matches, recovered transform checked;
  * `dropin_host_to_host` -- (N = 1) the reference-signature Python calls on the same cloud, NumPy in / NumPy out;
  * `parity`          -- 300 rows of the TIMED outputs against the CPU oracle (the timed path is the checked path);
  * `cpu_baseline`    -- (N = 1) the reference-shaped NumPy restatement (oracle/numpy_shaped.py, calibrated against
    the reference in the build container) and the scalar C port, on bounded samples, on this box's host cores.
Prints ONE JSON line (rank 0).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
FP16_MFMA_PEAK_TF = 2500.0  # dense FP16 matrix peak (ibid.)
FP64_VALU_PEAK_TF = 78.6    # FP64 vector peak (ibid.)
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
C4_EULER, C4_T = (0.3, -0.2, 0.5), (0.1, -0.3, 0.2)  # SURVEY 8d, config C4's rigid motion


def make_cloud(n: int, seed: int):
    rng = np.random.default_rng(seed)
    p = rng.random((n, 3), dtype=np.float32).astype(np.float64)
    nr = rng.standard_normal((n, 3))
    nr /= np.linalg.norm(nr, axis=1)[:, None]
    return p, nr


def c4_partner(points, normals, seed: int):
    """ref = scan[perm] R^T + t, normals rotated, perm from the given seed."""
    from scipy.spatial.transform import Rotation

    rot = Rotation.from_euler("xyz", C4_EULER).as_matrix()
    perm = np.random.default_rng(seed).permutation(points.shape[0])
    return points[perm] @ rot.T + np.asarray(C4_T), normals[perm] @ rot.T, perm, rot


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: N fresh children, one per GPU, started before this process has
    made any GPU call (a process that has initialised HIP must never fork + exec).  Rank 0 prints the JSON line."""
    port = free_port()
    procs = []
    for rank in range(n):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


# ---- CPU baselines (N = 1 only) -----------------------------------------------------------------------------------
def start_numpy_shaped_baseline(points_per_gpu: int, radius: float, sample: int):
    """The reference-shaped baseline forks a multiprocessing.Pool, so it runs in a child started BEFORE HIP is
    initialised in this process, and is waited for before any GPU timing starts."""
    n_procs = min(8, os.cpu_count() or 1)  # the reference's default n_procs = 8 (shot_parallelization.py:26)
    cmd = [sys.executable, os.path.join(ROOT, "oracle", "numpy_shaped.py"), str(points_per_gpu), repr(radius), str(sample), str(n_procs)]
    return subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=dict(os.environ, OMP_NUM_THREADS="1"))


def finish_numpy_shaped_baseline(proc) -> dict:
    out, _ = proc.communicate(timeout=900)
    if proc.returncode != 0:
        raise RuntimeError("oracle/numpy_shaped.py failed")
    r = json.loads(out.strip().splitlines()[-1])
    cal = None
    try:
        cal = json.load(open(os.path.join(ROOT, "profiles", "r02_cpu_calibration.json")))
    except Exception:
        pass
    return {
        "value": r["desc_per_s"],
        "unit": "descriptors/s",
        "cores": r["n_procs"],
        "kind": "port",
        "sample": f"{r['points']}-pt uniform cloud, all points keypoints, r={r['radius']:.4f} (same expected neighbours per ball as "
        f"the GPU workload); reference-shaped NumPy restatement oracle/numpy_shaped.py: sklearn KDTree + per-point NumPy loop, "
        f"FPFH single process {r['fpfh_s']:.1f}s (as the reference runs it), SHOT through a fork Pool of {r['n_procs']} "
        f"{r['shot_s']:.1f}s (the reference's default n_procs)",
        "host_cores_available": os.cpu_count(),
        "calibration_vs_reference": None if cal is None else {
            "ratio_time_restatement_over_reference": cal["ratio_total"], "reference_desc_per_s_build_container": cal["ref_desc_per_s"],
            "max_abs_diff_vs_reference": max(cal["fpfh_max_abs_diff"], cal["shot_max_abs_diff"]),
            "where": "build container, 8 vCPU (tools/calibrate_cpu_baseline.py; the reference cannot travel to the GPU box)"},
    }


def c_port_baseline(points_per_gpu: int, radius: float) -> dict:
    """The scalar C oracle (one thread) on a 150k-point sample at the same neighbours per ball: a stricter baseline."""
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
    return {"value": 2 * ns / (t2 - t0), "unit": "descriptors/s", "cores": 1, "kind": "port",
            "sample": f"{ns}-pt uniform cloud, all points keypoints, r={r:.4f}, FPFH {t1 - t0:.1f}s + SHOT {t2 - t1:.1f}s, "
            f"oracle/shot_fpfh_oracle.c single thread"}


# ---- parity of the timed outputs ---------------------------------------------------------------------------------------
def parity_sample(job, points, normals, radius, rows: int = 300) -> dict:
    """`rows` rows of the outputs the TIMED steps left in HBM against the CPU oracle (BASELINE tolerance)."""
    from oracle import oracle as O

    rng = np.random.default_rng(5)
    orig = job.block_original_indices()
    pick = np.sort(rng.choice(job.m, min(rows, job.m), replace=False))
    out = {"rows": int(pick.size), "tolerance": "|a-b| <= 1e-5*max(1,|b|)"}
    ok = True
    if job.do_fpfh:
        got = np.stack([job.fpfh_out.rows_to_host(int(i), 1)[0] for i in pick])
        want = O.compute_fpfh_descriptor_sample(orig[pick], points, normals, radius, job.n_bins)
        err = np.abs(got - want)
        out["fpfh_max_abs_err"] = float(err.max())
        ok &= bool((err <= 1e-5 * np.maximum(1.0, np.abs(want))).all())
    if job.do_shot:
        got = np.stack([job.shot_out.rows_to_host(int(i), 1)[0] for i in pick])
        want = O.shot_single_scale(points, normals, points[orig[pick]], radius, job.normalize, job.min_nb)
        err = np.abs(got - want)
        out["shot_max_abs_err"] = float(err.max())
        ok &= bool((err <= 1e-5 * np.maximum(1.0, np.abs(want))).all())
    out["ok"] = ok
    return out


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--points-per-gpu", type=int, default=1_000_000)
    ap.add_argument("--radius", type=float, default=0.03)
    ap.add_argument("--spfh-exchange", choices=["neighbor", "halo", "allgather"], default="neighbor",
                    help="N > 1: how a rank gets the SPFH rows of its block's halo (sharding.py)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=20000, help="points of the reference-shaped CPU baseline's sample")
    ap.add_argument("--only", choices=["both", "fpfh", "shot"], default="both")
    ap.add_argument("--overlap", action="store_true",
                    help="run the FPFH and SHOT chains on two HIP streams (faster; per-kernel times then overlap)")
    ap.add_argument("--no-match", action="store_true", help="skip the exchange + matching phase (config 5's tail)")
    ap.add_argument("--match-rows", type=int, default=None,
                    help="keypoints of the matched subset, whole job (default 262144 at N = 1, 131072 x N otherwise)")
    ap.add_argument("--match-steps", type=int, default=2)
    ap.add_argument("--no-ransac", action="store_true")
    ap.add_argument("--no-dropin", action="store_true", help="skip the host-to-host drop-in timing (N = 1)")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-kernel-timers", action="store_true",
                    help="diagnostic: time the steps without the per-kernel HIP events (no roofline / kernels_ms_per_step then)")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="FUNCTIONAL TEST ONLY: allow more ranks than GPUs; the exchange is then staged through host memory "
                         "and gloo (RCCL refuses two ranks on one device) and no timing is a scaling result")
    ap.add_argument("--emulate-rank", type=int, default=None, metavar="R",
                    help="single process, no rendezvous: run rank R's share of a --gpus N descriptor pass on this GPU")
    args = ap.parse_args()

    emulated = args.emulate_rank is not None
    if "WORLD_SIZE" not in os.environ and args.gpus > 1 and not emulated:
        return spawn_ranks(args.gpus)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if emulated:
        world, rank, local_rank = args.gpus, args.emulate_rank, 0
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    lead = rank == 0 or emulated
    single = world == 1
    # stdout carries exactly ONE line, the JSON record: libraries that print there (RCCL's version banner) are sent to
    # stderr for the rest of the run, and the record goes to the saved descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    shaped = None
    if single and not args.no_cpu_baseline:  # (forks a Pool: must start before this process touches the GPU)
        shaped = finish_numpy_shaped_baseline(start_numpy_shaped_baseline(args.points_per_gpu, args.radius, args.cpu_sample))

    dist = None
    if world > 1 and not emulated:
        import torch.distributed as dist  # control plane only: rendezvous, barrier, max-reduce of the timing

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)

    import shot_fpfh_amd as s
    from shot_fpfh_amd import _ffi
    from shot_fpfh_amd.sharding import DescriptorJob, SubsetMatchJob

    n_dev = max(_ffi.load().sf_device_count(), 1)
    oversub = world > n_dev and not emulated
    if oversub and not args.oversubscribe:
        raise SystemExit(f"{world} ranks but {n_dev} GPU(s): one process per GPU is the only measured configuration "
                         f"(--oversubscribe runs a functional test with a host-staged exchange)")
    eng = s.Engine(local_rank % n_dev)
    exchange = "none (descriptor pass only)"
    rccl_ranks = 0

    def make_host_staged_allgather():
        import torch

        def host_staged_allgather(buf, bytes_per_rank):
            flat = buf.to_host().reshape(-1).view(np.uint8)
            mine = torch.from_numpy(flat[rank * bytes_per_rank:(rank + 1) * bytes_per_rank].copy())
            parts = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(parts, mine)
            flat[: world * bytes_per_rank] = torch.cat(parts).numpy()
            buf.from_host(flat.view(buf.dtype).reshape(buf.shape))

        return host_staged_allgather

    if not emulated:
        if oversub:
            exchange = "gloo, staged through host memory (oversubscribed functional test, NOT RCCL)"
            eng.allgather = make_host_staged_allgather()
            if args.spfh_exchange == "allgather":
                raise SystemExit("--oversubscribe supports the halo SPFH exchange only")
        else:
            # RCCL communicator -- one rank too, so that ncclAllGather really executes.  The descriptor pass (`value`)
            # needs no collective; if the communicator cannot be built on this node the exchange phase still runs,
            # staged through host memory, and the record says so instead of the whole bench dying.
            err = ""
            try:
                ids = [eng.comm_unique_id() if rank == 0 else None]
                if dist is not None:
                    dist.broadcast_object_list(ids, src=0)
                eng.comm_init(ids[0], world, rank)
            except Exception as exc:  # noqa: BLE001 -- reported in the record
                err = f"{type(exc).__name__}: {exc}"
            errs = [err]
            if dist is not None:
                errs = [None] * world
                dist.all_gather_object(errs, err)
            if any(errs):
                first = next(e for e in errs if e)
                if dist is None:
                    raise SystemExit(f"RCCL communicator: {first}")
                exchange = f"gloo, staged through host memory -- RCCL communicator failed ({first[:200]})"
                eng.allgather = make_host_staged_allgather()
                if args.spfh_exchange == "allgather":
                    raise SystemExit("SPFH all-gather needs RCCL: " + first)
            else:
                exchange = f"RCCL ncclAllGather over {world} rank(s)"
                rccl_ranks = world

    n_total = args.points_per_gpu * world
    radius = args.radius * world ** (-1.0 / 3.0)
    points, normals = make_cloud(n_total, 3)
    job = DescriptorJob(eng, points, normals, radius, n_bins=5, normalize=True, min_neighborhood_size=10, world=world,
                        rank=rank, spfh_exchange=args.spfh_exchange, do_fpfh=args.only in ("both", "fpfh"),
                        do_shot=args.only in ("both", "shot"), overlap_chains=args.overlap, emulate_peers=emulated)

    def barrier():
        eng.sync()
        if dist is not None:
            dist.barrier()

    def max_over_ranks(x: float) -> float:
        if dist is None:
            return x
        import torch

        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # ---- timed region: K descriptor passes ---------------------------------------------------------------------------
    # Two HIP event records per launch cost the step 1.6 % (4.55 against 4.48 ms), so the timed region brackets the roofline
    # kernel only -- roofline.achieved is measured live over the timed launches, as the contract asks; which kernel that is
    # comes from the warm-up steps (every launch bracketed; the first step, which grows the pools, left out when there are
    # two or more), and the breakdown of the other kernels from a few untimed steps after the timed ones.
    warm_rep, warm_steps = {}, 0
    for w in range(args.warmup):
        if w == min(1, args.warmup - 1):
            eng.sync()
            eng.profile_reset()
            eng.profile(True)
        job.step()
        warm_steps += 1 if w >= min(1, args.warmup - 1) else 0
    if args.warmup:
        eng.sync()
        eng.profile(False)
        warm_rep = eng.profile_report()
    timed_only = None
    if warm_rep and not args.no_kernel_timers:
        cand = [k for k, v in warm_rep.items() if k in ALG_BYTES and v[0] > 0 and v[1] > 0]
        if cand:
            timed_only = max(cand, key=lambda k: warm_rep[k][1])
    barrier()
    eng.profile_reset()
    eng.profile_only(timed_only)
    eng.profile(not args.no_kernel_timers)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        job.step()
    barrier()
    elapsed = time.perf_counter() - t0
    eng.profile(False)
    eng.profile_only(None)
    elapsed = max_over_ranks(elapsed)
    rep = eng.profile_report()
    if timed_only is not None:
        # the breakdown of the other kernels: a few more steps, untimed, every launch bracketed, with the clocks where the
        # timed steps left them (the warm-up steps run 5-10 % slow and only chose the roofline kernel)
        warm_steps = min(args.steps, 5)
        eng.profile_reset()
        eng.profile(True)
        for _ in range(warm_steps):
            job.step()
        eng.sync()
        eng.profile(False)
        warm_rep = eng.profile_report()
        barrier()

    kinds = (1 if job.do_fpfh else 0) + (1 if job.do_shot else 0)
    n_desc = kinds * n_total
    ms_per_step = 1000.0 * elapsed / args.steps
    value = n_desc / (elapsed / args.steps)

    if args.no_kernel_timers:
        if lead:
            os.write(json_fd, (json.dumps({"diagnostic": "steps timed without per-kernel HIP events", "ms_per_step": ms_per_step,
                                           "value": value, "n_gpus": world, "steps": args.steps}) + "\n").encode())
        return 0
    out = {}
    if lead:
        kern = {k: (v[0], v[1] / max(v[0], 1)) for k, v in rep.items() if v[0] > 0 and v[1] > 0}
        per_step_ms = {k: rep[k][1] / args.steps for k in kern}
        if timed_only is not None:  # the other kernels: from the instrumented warm-up steps
            for k, v in warm_rep.items():
                if k not in per_step_ms and v[0] > 0 and v[1] > 0:
                    per_step_ms[k] = v[1] / max(warm_steps, 1)
        dom = max((k for k in kern if k in ALG_BYTES), key=lambda k: rep[k][1])
        launches, avg_ms = kern[dom]
        units = job.m  # descriptors of this rank's block per launch (halo SPFH rows are extra work, not counted)
        alg_bytes = ALG_BYTES[dom] * units + (4 * job.last_pairs if dom == "k2_radius_fill" else 0)
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic, traffic_src = tj.get(dom), tj.get("_source", "profiles/traffic.json")
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
                "exchange": exchange,
            },
            "kernels_ms_per_step": {k: round(v, 4) for k, v in sorted(per_step_ms.items())},
            "kernels_ms_per_step_source": "HIP events around every launch of the timed steps" if timed_only is None else
            f"{timed_only}: HIP events around its launches in the timed steps; the others: {warm_steps} more, untimed steps with every "
            "launch bracketed (two event records per launch cost the step 1.6 %, so the timed steps bracket the roofline kernel only)",
            "roofline": {
                "kernel": dom,
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": None if traffic is None else
                f"NOT measured in this run: per-launch HBM bytes from separate rocprofv3 --pmc passes of this command, {traffic_src}",
                "avg_launch_ms": avg_ms,
                "algorithmic_bytes_per_launch": alg_bytes,
                "note": ("the kernel is float64-VALU-issue bound, not HBM bound: 783 vector instructions per keypoint at ~95 % "
                         "issue (SQ counters, profiles/r02_k5.md); the HBM fraction is reported because the contract asks for it")
                if dom == "k5_shot" else None,
            },
            "whole_step_hbm": {"algorithmic_bytes": sum(ALG_BYTES[k] for k in ("k6_spfh", "k7_fpfh", "k5_shot")) * units,
                               "achieved_gbs": sum(ALG_BYTES[k] for k in ("k6_spfh", "k7_fpfh", "k5_shot")) * units / (ms_per_step * 1e-3) / 1e9}
            if kinds == 2 else None,
        }

    # ---- parity of what the timed steps left in HBM ------------------------------------------------------------------
    if not args.no_parity:
        par = parity_sample(job, points, normals, radius)
        if dist is not None:
            allp = [None] * world
            dist.all_gather_object(allp, par)
            par = {"per_rank": allp, "ok": all(p["ok"] for p in allp)}
        if lead:
            out["parity"] = par

    # ---- config 5's tail: partner cloud, subset, RCCL all-gather, sharded K8 ------------------------------------------
    if not args.no_match and job.do_shot and not emulated:
        total_rows = args.match_rows or (262144 if single else 131072 * world)
        total_rows = min(total_rows, n_total)
        ref_pts, ref_nrm, perm, rot = c4_partner(points, normals, 4)
        t_prep = time.perf_counter()
        ref_job = DescriptorJob(eng, ref_pts, ref_nrm, radius, n_bins=5, normalize=True, min_neighborhood_size=10, world=world,
                                rank=rank, spfh_exchange="halo", do_fpfh=False, do_shot=True)
        ref_job.step()
        eng.sync()
        t_prep = time.perf_counter() - t_prep
        scan_orig = job.block_original_indices()            # scan rows of this rank -> scan point
        ref_label = perm[ref_job.block_original_indices()]  # ref rows of this rank -> the scan point they came from
        s_sel, r_sel = np.flatnonzero(scan_orig < total_rows), np.flatnonzero(ref_label < total_rows)
        cap = max_over_ranks(float(max(s_sel.size, r_sel.size, 1)))
        cap = int(-(-int(cap) // 256) * 256)
        sub = SubsetMatchJob(eng, 352, cap, world, rank)
        sub.select(job.shot_out, s_sel, scan_orig[s_sel], ref_job.shot_out, r_sel, ref_label[r_sel])
        sub.run()  # warm-up (RCCL channel setup, pool growth)
        barrier()
        eng.profile_reset()
        eng.profile(True)
        t0 = time.perf_counter()
        for _ in range(args.match_steps):
            sub.run()
        barrier()
        t_match = max_over_ranks(time.perf_counter() - t0) / max(args.match_steps, 1)
        eng.profile(False)
        mrep = eng.profile_report()
        s_lab, r_lab = sub.matches()
        stats = np.array([float((s_lab == r_lab).sum()), float(s_lab.size)])
        if dist is not None:
            import torch

            tt = torch.from_numpy(stats)
            dist.all_reduce(tt)
            stats = tt.numpy()
        # the (scan, reference) pairs of every rank on every rank (one more small all-gather), then the registration
        # they are for: RANSAC over the WHOLE match set, scored on this rank's GPU (K9)
        t0 = time.perf_counter()
        all_s, all_r = sub.gather_matches()
        t_pairs = time.perf_counter() - t0
        reg = None
        if lead and all_s.size >= 4:
            import shot_fpfh_amd.matching.ransac as R
            from shot_fpfh_amd.matching import ransac_on_matches

            inv = np.empty_like(perm)
            inv[perm] = np.arange(perm.size)
            R.rng = np.random.default_rng(seed=72)
            t0 = time.perf_counter()
            ratio_m, tf_m = ransac_on_matches(all_s, inv[all_r], points, ref_pts, n_draws=2000, draw_size=4,
                                              distance_threshold=0.01, disable_progress_bar=True, engine=eng)
            reg = {"matches": int(all_s.size), "draws": 2000, "seconds": time.perf_counter() - t0, "inlier_ratio": float(ratio_m),
                   "rotation_err": float(np.abs(tf_m.rotation - rot).max()),
                   "translation_err": float(np.abs(tf_m.translation - np.asarray(C4_T)).max())}
        if lead:
            k8 = {k: v[1] / max(args.match_steps, 1) for k, v in mrep.items() if v[1] > 0}
            k8_ms = sum(v for k, v in k8.items() if k.startswith("k8_"))
            gathered = cap * world
            flop = 2.0 * cap * gathered * 352
            out["exchange_match"] = {
                "what": "BASELINE config 5 tail on SHOT rows: subset gather, all-gather of reference rows + labels, sharded K8, "
                        "all-gather of the match pairs, RANSAC over all of them",
                "rccl_ranks": rccl_ranks,
                "exchange": exchange,
                "subset_keypoints_total": int(total_rows),
                "rows_per_rank_padded": cap,
                "allgather_bytes_per_rank": cap * 352 * 8,
                "ms_per_pass": 1000.0 * t_match,
                "kernels_ms_per_pass": {k: round(v, 4) for k, v in sorted(k8.items())},
                "k8_pair_dists_per_s": cap * gathered / (k8_ms * 1e-3) if k8_ms > 0 else None,
                "k8_roofline": {"bound": "mfma", "achieved": flop / (k8_ms * 1e-3) / 1e12 if k8_ms > 0 else None,
                                "peak": FP16_MFMA_PEAK_TF, "unit": "TFLOP/s",
                                "frac": flop / (k8_ms * 1e-3) / 1e12 / FP16_MFMA_PEAK_TF if k8_ms > 0 else None,
                                "note": "2*m1*m2*d flop of the FP16 pre-filter pass; exact FP64 re-ranking of the survivors included in the time"},
                "matches": int(stats[1]),
                "matches_recovering_true_correspondence": float(stats[0] / max(stats[1], 1.0)),
                "match_pairs_allgather_s": t_pairs,
                "registration_from_all_matches": reg,
                "partner_cloud_descriptor_pass_s": t_prep,
            }
        sub.close()
        ref_job.close()

        # ---- K9: 10^4 draws x 10^6 matches (N = 1) --------------------------------------------------------------------
        if single and not args.no_ransac:
            from shot_fpfh_amd.matching import ransac_on_matches
            import shot_fpfh_amd.matching.ransac as R

            m = min(1_000_000, n_total)
            rng = np.random.default_rng(9)
            inv = np.empty_like(perm)
            inv[perm] = np.arange(perm.size)
            si = np.arange(m)
            ri = inv[:m].copy()
            bad = rng.random(m) < 1.0 / 3.0
            ri[bad] = rng.integers(0, n_total, int(bad.sum()))  # a third of the matches are wrong
            R.rng = np.random.default_rng(seed=72)
            eng.profile_reset()
            eng.profile(True)
            t0 = time.perf_counter()
            ratio, tf = ransac_on_matches(si, ri, points, ref_pts, n_draws=10000, draw_size=4, distance_threshold=0.01,
                                          disable_progress_bar=True, engine=eng)
            t_r = time.perf_counter() - t0
            eng.profile(False)
            k9 = eng.profile_report().get("k9_ransac_score", (0, 0.0))
            flop9 = 26.0 * 1e4 * m
            out["ransac"] = {
                "draws": 10000, "matches": m, "host_to_host_s": t_r, "k9_ms": k9[1],
                "k9_pair_draws_per_s": 1e4 * m / (k9[1] * 1e-3) if k9[1] > 0 else None,
                "k9_roofline": {"bound": "valu_fp64", "achieved": flop9 / (k9[1] * 1e-3) / 1e12 if k9[1] > 0 else None,
                                "peak": FP64_VALU_PEAK_TF, "unit": "TFLOP/s",
                                "frac": flop9 / (k9[1] * 1e-3) / 1e12 / FP64_VALU_PEAK_TF if k9[1] > 0 else None},
                "inlier_ratio": float(ratio),
                "rotation_err": float(np.abs(tf.rotation - rot).max()),
                "translation_err": float(np.abs(tf.translation - np.asarray(C4_T)).max()),
            }

    # ---- the drop-in calls, host to host (N = 1) ---------------------------------------------------------------------------
    if single and not args.no_dropin and not emulated:
        from shot_fpfh_amd.descriptors import ShotMultiprocessor, compute_fpfh_descriptor

        job.close()
        kp = np.arange(n_total)
        res = {}
        for name, call in (
            ("compute_fpfh_descriptor", lambda: compute_fpfh_descriptor(kp, points, normals, radius, 5, verbose=False)),
            ("ShotMultiprocessor.compute_descriptor_single_scale", None),
        ):
            if call is None:
                def call():
                    with ShotMultiprocessor(normalize=True, min_neighborhood_size=10, verbose=False) as sm:
                        return sm.compute_descriptor_single_scale(points, normals, points, radius)
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                arr = call()
                ts.append(time.perf_counter() - t0)
                shape = arr.shape
                del arr
            res[name] = {"first_call_s": ts[0], "best_s": min(ts), "out_shape": list(shape),
                         "desc_per_s_best": n_total / min(ts), "out_gb": shape[0] * shape[1] * 8 / 1e9}
        best = sum(v["best_s"] for v in res.values())
        out["dropin_host_to_host"] = {
            "what": "reference-signature Python calls, NumPy arrays in, fresh NumPy arrays out (H2D + kernels + D2H), same cloud",
            "calls": res, "desc_per_s_both": 2 * n_total / best,
        }
    else:
        job.close()

    if lead:
        if single and not args.no_cpu_baseline:
            out["cpu_baseline"] = shaped
            out["cpu_baseline_c_port"] = c_port_baseline(args.points_per_gpu, args.radius)
            out["speedup_vs_cpu_baseline"] = value / shaped["value"]
            out["speedup_vs_c_port"] = value / out["cpu_baseline_c_port"]["value"]
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    os.close(json_fd)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
