#!/usr/bin/env python3
"""bench.py -- descriptors/s (SHOT + FPFH) of the hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Run directly with N > 1 (no WORLD_SIZE in the environment) it starts N child processes itself, one per GPU, BEFORE
touching the GPU; under a launcher (torchrun: RANK / LOCAL_RANK / WORLD_SIZE set) it is one of the ranks.  The ranks'
control plane (rendezvous, barrier, max of the timings, the RCCL unique id) is a few lines of sockets + pickle on
127.0.0.1 (class Control): the product has no PyTorch, and neither has its bench.

Workload (BASELINE.json metric "descriptors/sec (SHOT+FPFH) on 1M-pt cloud"): a synthetic uniform cloud of 1M points
PER GPU (seed 3, float32-grid coordinates, random unit normals), every point a keypoint, radius 0.03 at N = 1 (k ~ 110
neighbours); for N > 1 the cloud has N x 1M points and the radius shrinks by N^(-1/3), so the per-GPU work is fixed
(weak scaling; BASELINE config 5 at N = 8).

Timed region (`value`): K steps, each ONE pass of the descriptor path over this rank's block with inputs resident in
HBM -- K1 grid build, K2 radius search, K6 SPFH (+ the SHOT frame moments), K7 FPFH (rows x 125 float64 out), K4 frame
eigen-solves, K5 SHOT (rows x 352 float64 out); outputs stay in HBM.  value = 2 x (N x 1M) descriptors / step time
(max over ranks).  Same definition at every N, so the driver's scaling efficiency compares like with like.

After the timed region the SAME process measures the rest of the path north_star names and reports it under extra keys:
  * `roofline_all`    -- every kernel of the step: algorithmic bytes, average launch time, fraction of the 8.0 TB/s spec
    peak and of the 6.29 TB/s measured-copy peak; for K5 also the float64 issue ceiling its SQ counters give;
  * `normals`         -- (N = 1) compute_normals(radius) on the same cloud: K2 + K3, resident and host to host;
  * `strong_scaling`  -- (N > 1) the FIXED 1M-point cloud of the metric's name cut over the N ranks (`value` stays weak);
  * `exchange_match`  -- BASELINE config 5's tail: the C4 partner cloud's SHOT rows (one more descriptor pass), a
    keypoint subset of both descriptor sets gathered per rank, ONE RCCL all-gather of the reference subset rows over
    xGMI (ncclAllGather is executed at N = 1 too, on a one-rank communicator), the sharded K8 brute-force L2 matching,
    and the fraction of matches that recover the true correspondence;
  * `ransac`          -- (N = 1) K9 at 10^4 draws x 10^6 matches through ransac_on_matches, recovered transform checked;
  * `dropin_host_to_host` -- (N = 1) the reference-signature Python calls on the same cloud, NumPy in / NumPy out;
  * `parity`          -- 300 rows of the TIMED outputs against the CPU oracle (the timed path is the checked path);
  * `cpu_baseline`    -- (N = 1) the reference-shaped NumPy restatement (oracle/numpy_shaped.py, calibrated against
    the reference in the build container) and the scalar C port, on bounded samples, on this box's host cores.
Prints ONE compact JSON line (rank 0, < 4 KB: the contract's keys, `roofline`, `cpu_baseline`, `sustained`, `parity`); every
other block named above goes to the side file the line's `detail` key names (bench_detail.json) and to stderr.
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
HBM_COPY_GBS = 6290.0       # measured copy peak (ibid. :36) -- the ceiling a streaming kernel can actually reach
FP16_MFMA_PEAK_TF = 2500.0  # dense FP16 matrix peak (ibid.)
INT8_MFMA_PEAK_TOP = 5000.0  # dense INT8 matrix peak (ibid.: 2 x BF16 per clock)


def _k8_roofline(k8: dict, k8_ms: float, ops: float) -> dict:
    """K8 against the matrix-core peak of the operand type its dominant pass ran in: int8 (match_i8.hip, round 6) when
    `k8_match_i8` is among the kernels, FP16 (match_half.hip) otherwise.  `ops` = 2 m1 m2 d of ONE pass over all pairs; the conversion
    passes, the re-scan of the live (row, split) pairs, the FP16 pass of the rows without a clear nearest descriptor and the float64
    decision are all in the time."""
    int8 = k8.get("k8_match_i8", 0.0) > 0.0
    peak = INT8_MFMA_PEAK_TOP if int8 else FP16_MFMA_PEAK_TF
    ach = ops / (k8_ms * 1e-3) / 1e12 if k8_ms > 0 else None
    main = k8.get("k8_match_i8" if int8 else "k8_match_half", 0.0)
    return {"bound": "mfma", "operand": "int8" if int8 else "fp16", "achieved": ach, "peak": peak, "unit": "Top/s" if int8 else "TFLOP/s",
            "frac": ach / peak if ach else None,
            "main_pass_ms": main, "main_pass_frac": ops / (main * 1e-3) / 1e12 / peak if main > 0 else None,
            "note": "`peak` is the spec figure at 2.4 GHz; under a dense matrix stream this chip holds a lower clock -- bare loops on random "
                    "operands sustain 1 715 TFLOP/s (v_mfma_f32_32x32x16_f16) and 3 804 Top/s (v_mfma_i32_32x32x32_i8), "
                    "tools/ubench/mfma_rates.hip -- which is what `main_pass_frac` x peak is to be read against"}
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
    "k2_radius_slots": 24,  # + 4 B per pair (the optimistic single pass), added below
    "k3_normals": 24 + 24,  # query in, normal out
    "k1_gather_sorted": 48 + 48 + 4,  # xyz + normal in, 48-byte record (+ SoA copy of xyz) out, perm
    "k1_radix_sort": 16,  # (cell id, index) pairs in and out, per pass of the sort -- one pass counted
    "k1_cell_settle": 8 + 48 + 48 + 4 + 4,  # (index, cell) slot in, xyz + normal in, 48-byte record (+ SoA copy of xyz) out, both permutations
}
PER_PAIR = ("k2_radius_fill", "k2_radius_slots")
PROFILE_TAG = "r06"  # profiles/<tag>_{traffic,k5_sq,sustained_clock}.json: counter files of THIS round's build (stamped)
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


def dry_run_n(args) -> int:
    """`--dry-run-n N`: everything a `--gpus N` run would plan AND compute, rank after rank, on this ONE GPU (no RCCL: what a
    collective would deliver is put in place by device copies) -- so that a first real N-GPU launch cannot fail on something a
    single GPU can find out: index widths, buffer sizes, list capacities, the exchange plan of every rank, the chunked matching
    phase at its true shapes.  Nothing here is a scaling measurement; the times are one GPU's, per rank's share."""
    import shot_fpfh_amd as s
    from shot_fpfh_amd.sharding import DescriptorJob, MatchJob, ShardPlan, exchange_plan

    world = int(args.dry_run_n)
    n_total = args.points_per_gpu * world
    radius = args.radius * world ** (-1.0 / 3.0)
    total_rows = min(args.match_rows or 131072 * world, n_total)
    chunks = args.match_chunks or 4
    problems = []
    if n_total >= 2**31 - 1024:
        problems.append(f"{n_total} points exceed the 32-bit cell-sorted positions")
    points, normals = make_cloud(n_total, 3)
    ref_pts, ref_nrm, perm, rot = c4_partner(points, normals, 4)
    eng = s.Engine(0)
    out = {"dry_run_ranks": world, "points_total": n_total, "radius": radius, "spfh_exchange": args.spfh_exchange,
           "host_bytes_per_rank_process": int(2 * (points.nbytes + normals.nbytes) + perm.nbytes),
           "what": "every rank's descriptor pass (emulate_peers: the adjacent ranks' SPFH rows computed here) and every rank's share of the "
                   "chunked matching phase, one after the other on ONE GPU; collectives replaced by device copies"}
    ranks, scan_sub, ref_sub, scan_lab, ref_lab = [], [], [], [], []
    first = None
    cap = 0
    picks = []
    for r in range(world):
        row = {"rank": r}
        job = DescriptorJob(eng, points, normals, radius, n_bins=5, normalize=True, min_neighborhood_size=10, world=world, rank=r,
                            spfh_exchange=args.spfh_exchange, emulate_peers=True)
        ref_job = DescriptorJob(eng, ref_pts, ref_nrm, radius, n_bins=5, normalize=True, min_neighborhood_size=10, world=world, rank=r,
                                spfh_exchange="halo", do_fpfh=False, do_shot=True)
        try:
            job.step()
            eng.sync()
            t0 = time.perf_counter()
            job.step()
            eng.sync()
            row["descriptor_step_ms"] = 1000.0 * (time.perf_counter() - t0)
            ref_job.step()
            eng.sync()
            if first is None:
                first = job.cloud.layer_table()
                out["z_layers"] = int(first.size - 1)
            xp = exchange_plan(first, n_total, world, r)
            b, e = job.plan.block()
            row.update({"block": [int(b), int(e)], "halo": [int(xp.halo[0]), int(xp.halo[1])], "interior": [int(xp.interior[0]), int(xp.interior[1])],
                        "pairs": int(job.last_pairs), "mean_list": job.last_pairs / max(e - b, 1),
                        "exchange_ops": [{"peer": int(p_), "send_rows": int(se - sb), "recv_rows": int(re - rb)} for (p_, sb, se, rb, re) in xp.ops],
                        "output_bytes": int((e - b) * (125 + 352 + 9) * 8)})
            if len(xp.ops) > 2:
                problems.append(f"rank {r} exchanges with {len(xp.ops)} peers (a block thinner than a z-layer)")
            scan_orig = job.block_original_indices()
            ref_label = perm[ref_job.block_original_indices()]
            picks.append((np.flatnonzero(scan_orig < total_rows), scan_orig, np.flatnonzero(ref_label < total_rows), ref_label))
            cap = max(cap, picks[-1][0].size, picks[-1][2].size)
            # this rank's subset rows stay on the device (what select() would gather), compacted to their count for now
            for sel_rows, rows_dev, store in ((picks[-1][0], job.shot_out, scan_sub), (picks[-1][2], ref_job.shot_out, ref_sub)):
                sel = eng.empty((max(sel_rows.size, 1),), np.int64).from_host(sel_rows if sel_rows.size else np.zeros(1, np.int64))
                dst = eng.empty((max(sel_rows.size, 1), 352))
                eng.rows_gather_device(rows_dev, sel, dst)
                eng.sync()
                sel.free()
                store.append(dst)
        finally:
            job.close()
            ref_job.close()
        ranks.append(row)
    cap = int(-(-cap // 256) * 256)
    out["match"] = {"subset_keypoints_total": int(total_rows), "rows_per_rank_padded": cap, "chunks": chunks,
                    "allgather_bytes_per_rank": cap * 352 * 8, "gathered_set_bytes": cap * world * 352 * 8}
    # the gathered reference set as ncclAllGather would leave it: rank-major blocks of `cap` rows, zero rows behind each rank's own
    gathered = eng.empty((cap * world, 352)).from_host(np.zeros((cap * world, 352)))
    labels_all = np.full(cap * world, -1, np.int64)
    for r in range(world):
        n_r = picks[r][2].size
        if n_r:
            gathered.copy_from_device(ref_sub[r], dst_byte_offset=r * cap * 352 * 8, nbytes=n_r * 352 * 8)
        labels_all[r * cap:r * cap + n_r] = picks[r][3][picks[r][2]]
    eng.sync()
    good = total = 0
    for r in range(world):
        n_s = picks[r][0].size
        blk = eng.empty((cap, 352)).from_host(np.zeros((cap, 352)))
        if n_s:
            blk.copy_from_device(scan_sub[r], nbytes=n_s * 352 * 8)
        res = {}
        for c in (1, chunks):
            mj = MatchJob(eng, 352, cap, cap * world, 1, 0, chunks=c)
            try:
                t0 = time.perf_counter()
                mj.run(blk, gathered)
                eng.sync()
                res[c] = (mj.idx.to_host().copy(), mj.dist.to_host().copy(), 1000.0 * (time.perf_counter() - t0))
                if c == 1:
                    rows_m, idx_m = mj.matches()
            finally:
                mj.close()
        same = bool(np.array_equal(res[1][0], res[chunks][0]) and np.array_equal(res[1][1], res[chunks][1]))
        if not same:
            problems.append(f"rank {r}: the chunked matching differs from the one-shot one")
        ranks[r]["k8_ms_one_shot"], ranks[r]["k8_ms_chunked"], ranks[r]["chunked_equals_one_shot"] = res[1][2], res[chunks][2], same
        s_lab = picks[r][1][picks[r][0]][rows_m] if n_s else np.zeros(0, np.int64)
        r_lab = labels_all[idx_m]
        good += int((s_lab == r_lab).sum())
        total += int(s_lab.size)
        blk.free()
    out["match"]["matches"] = total
    out["match"]["matches_recovering_true_correspondence"] = good / max(total, 1)
    if total and good / total < 0.85:
        problems.append(f"only {good / total:.3f} of the matches recover the true correspondence")
    for a in scan_sub + ref_sub + [gathered]:
        a.free()
    out["ranks"] = ranks
    out["problems"] = problems
    out["ok"] = not problems
    eng.close()
    path = os.path.join(os.environ.get("SF_BENCH_DETAIL_DIR", ROOT), f"bench_detail_dry_run_n{world}.json")
    try:
        with open(path, "w") as f:
            json.dump(out, f, indent=1)
    except OSError as exc:
        sys.stderr.write(f"bench.py: could not write {path}: {exc}\n")
    brief = {"dry_run_ranks": world, "ok": out["ok"], "problems": problems, "points_total": n_total, "z_layers": out.get("z_layers"),
             "descriptor_step_ms_per_rank": [round(x["descriptor_step_ms"], 3) for x in ranks],
             "exchange_peers_per_rank": [len(x["exchange_ops"]) for x in ranks],
             "k8_ms_per_rank_chunked": [round(x["k8_ms_chunked"], 2) for x in ranks], "match": out["match"],
             "note": "one GPU stood in for every rank in turn: a functional dry run, no scaling curve exists", "detail": os.path.relpath(path, ROOT)}
    print(json.dumps(brief, separators=(",", ":")))
    return 0 if out["ok"] else 4


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
    token = os.urandom(16).hex()  # (authenticates the control plane's messages: Control._key)
    procs = []
    for rank in range(n):
        # HSA_ENABLE_IPC_MODE_LEGACY=0: this pool's host driver only supports dmabuf IPC, and without the variable RCCL's
        # peer-to-peer set-up between processes fails (hipIpcGetMemHandle: invalid argument).  The image exports it already;
        # whatever the caller has set is passed through untouched, "0" is only the default for a bare environment.
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), SF_BENCH_SELF_SPAWNED="1", SF_BENCH_TOKEN=token,
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


class Control:
    """The ranks' control plane: rank 0 listens on 127.0.0.1, the others connect; every operation is an all-gather of a
    pickled object through rank 0 (barrier, max, broadcast are special cases).  A few small messages per run.

    Port: MASTER_PORT when this script spawned the ranks itself; under torchrun that port belongs to the launcher's own
    store, so the candidates are MASTER_PORT + 1 ... + 32 -- rank 0 takes the first it can bind and greets every
    connection with a line naming MASTER_PORT and the world size, a client keeps trying the candidates until one answers
    with that greeting."""

    def __init__(self, rank: int, world: int, timeout: float = 180.0):
        self.rank, self.world = rank, world
        Control.establish_secret(rank, timeout)
        base = int(os.environ.get("MASTER_PORT", "29500"))
        first = base if os.environ.get("SF_BENCH_SELF_SPAWNED") else base + 1
        cands = [first + i for i in range(32)]
        hello = f"SFBENCH {base} {world}\n".encode()
        self.peers = []
        if rank == 0:
            srv = None
            for port in cands:
                try:
                    srv = socket.socket()
                    srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                    srv.bind(("127.0.0.1", port))
                    break
                except OSError:
                    srv.close()
                    srv = None
            if srv is None:
                raise SystemExit("bench control plane: no free port next to MASTER_PORT")
            srv.listen(world)
            srv.settimeout(timeout)
            got = {}
            while len(got) < world - 1:
                c, _ = srv.accept()
                c.settimeout(timeout)
                c.sendall(hello)
                try:
                    r = int(self._recv(c))
                    if not 1 <= r < world or r in got:
                        raise ValueError(f"rank {r} out of range or already connected")
                    self._send(c, ("ok", r))  # (authenticated: the client knows its key is this job's)
                    got[r] = c
                except (ConnectionError, ValueError, TypeError, OSError):
                    c.close()  # (not one of this job's ranks -- or one that read a stale token file: it retries)
            srv.close()
            self.peers = [got[r] for r in range(1, world)]
        else:
            # A hello is only complete when rank 0 has ANSWERED it under the job's key: a rank that picked up the token file a
            # crashed earlier run left behind (rank 0 replaces it, but may not have yet) is dropped by rank 0, reads the file again
            # and retries until the deadline (advisor, round 5).
            deadline = time.time() + timeout
            self.sock = None
            while self.sock is None and time.time() < deadline:
                c = None
                for port in cands:
                    try:
                        c = socket.create_connection(("127.0.0.1", port), timeout=2.0)
                        c.settimeout(2.0)
                        if self._readline(c) == hello:
                            break
                        c.close()
                        c = None
                    except OSError:
                        c = None
                if c is None:
                    time.sleep(0.2)
                    continue
                try:
                    c.settimeout(min(timeout, 20.0))
                    self._send(c, rank)
                    ack = self._recv(c)
                    if ack == ("ok", rank) or ack == ["ok", rank]:
                        c.settimeout(timeout)
                        self.sock = c
                        break
                    raise ValueError("unexpected answer to the hello")
                except (ConnectionError, ValueError, TypeError, OSError, EOFError):
                    c.close()
                    if not os.environ.get("SF_BENCH_TOKEN"):
                        Control._token = None
                        Control.establish_secret(rank, max(deadline - time.time(), 1.0))  # (the file may have been replaced)
                    time.sleep(0.2)
            if self.sock is None:
                raise SystemExit("bench control plane: rank 0 did not answer (or refused this rank's key: set SF_BENCH_TOKEN to the "
                                 "same 16+ random characters on every rank)")

    @staticmethod
    def _readline(c) -> bytes:
        buf = b""
        while not buf.endswith(b"\n") and len(buf) < 64:
            chunk = c.recv(1)
            if not chunk:
                break
            buf += chunk
        return buf

    _token = None  # the job's secret, once known
    _token_file = None

    @staticmethod
    def _token_path() -> str:
        import tempfile

        return os.path.join(tempfile.gettempdir(), f"sfbench-{os.getuid()}-{os.environ.get('MASTER_PORT', '29500')}.token")

    @classmethod
    def establish_secret(cls, rank: int, timeout: float) -> None:
        """Every message carries an HMAC-SHA256 under a key only the ranks of THIS job know, and a message that does not verify is
        never unpickled.  The key: the random token spawn_ranks() put into its children's environment (SF_BENCH_TOKEN; a launcher
        may set it too) -- or, under a launcher that sets none (torchrun's run id defaults to the public string "none"), 16
        random bytes that rank 0 writes to a file only this user can read (mode 0600, created exclusively) and the other ranks
        of the node read back.  Never an empty or guessable key: without a secret the control plane does not start."""
        tok = os.environ.get("SF_BENCH_TOKEN")
        if not tok:
            path = cls._token_path()
            if rank == 0:
                try:
                    os.unlink(path)  # (a stale file of an earlier run of this user on this port)
                except FileNotFoundError:
                    pass
                except OSError as exc:  # (somebody else's file squats the path)
                    raise SystemExit(f"bench control plane: cannot replace {path} ({exc}); set SF_BENCH_TOKEN to 16+ random characters, "
                                     "the same on every rank") from None
                try:
                    fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
                except OSError as exc:
                    raise SystemExit(f"bench control plane: cannot create {path} ({exc}); set SF_BENCH_TOKEN") from None
                tok = os.urandom(16).hex()
                os.write(fd, tok.encode())
                os.close(fd)
                cls._token_file = path
            else:
                deadline = time.time() + timeout
                while time.time() < deadline:
                    try:
                        st = os.stat(path)
                        if st.st_uid == os.getuid() and (st.st_mode & 0o077) == 0 and st.st_size == 32:
                            with open(path) as f:
                                tok = f.read().strip()
                            break
                    except FileNotFoundError:
                        pass
                    time.sleep(0.05)
        if not tok or len(tok) < 16:
            raise SystemExit("bench control plane: no job secret (set SF_BENCH_TOKEN to 16+ random characters, the same on every rank)")
        cls._token = tok

    @staticmethod
    def _key() -> bytes:
        if not Control._token:
            raise SystemExit("bench control plane: used before its secret was established")
        return ("sfbench|" + Control._token + "|" + os.environ.get("MASTER_PORT", "") + "|" + os.environ.get("WORLD_SIZE", "")).encode()

    @staticmethod
    def _send(c, obj) -> None:
        import hashlib
        import hmac
        import pickle
        import struct

        data = pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL)
        mac = hmac.new(Control._key(), data, hashlib.sha256).digest()
        c.sendall(struct.pack("<Q", len(data)) + mac + data)

    @staticmethod
    def _recv(c):
        import hashlib
        import hmac
        import pickle
        import struct

        def exactly(n):
            parts, left = [], n
            while left:
                chunk = c.recv(min(left, 1 << 20))
                if not chunk:
                    raise ConnectionError("bench control plane: peer closed")
                parts.append(chunk)
                left -= len(chunk)
            return b"".join(parts)

        (n,) = struct.unpack("<Q", exactly(8))
        if n > (1 << 34):
            raise ConnectionError("bench control plane: implausible message length")
        mac, data = exactly(32), exactly(n)
        if not hmac.compare_digest(mac, hmac.new(Control._key(), data, hashlib.sha256).digest()):
            raise ConnectionError("bench control plane: message failed authentication (not from a rank of this job)")
        return pickle.loads(data)

    def allgather(self, obj) -> list:
        if self.world == 1:
            return [obj]
        if self.rank == 0:
            out = [obj] + [self._recv(c) for c in self.peers]
            for c in self.peers:
                self._send(c, out)
            return out
        self._send(self.sock, obj)
        return self._recv(self.sock)

    def barrier(self) -> None:
        self.allgather(None)

    def max(self, x: float) -> float:
        return max(self.allgather(float(x)))

    def bcast(self, obj):
        return self.allgather(obj)[0]

    def close(self) -> None:
        if Control._token_file:
            try:
                os.unlink(Control._token_file)
            except OSError:
                pass
            Control._token_file = None
        for c in self.peers + ([self.sock] if self.rank else []):
            try:
                c.close()
            except OSError:
                pass


# ---- the record: ONE compact line on stdout, everything else in a side file -------------------------------------------------
LINE_LIMIT = 4096  # bytes; the driver recovers the last stdout line as JSON -- round 4's 23.6 KB line came back unparsed
DETAIL_FILE = "bench_detail.json"


def _r(x, digits=6):
    """Floats to `digits` significant digits (the line is a summary; bench_detail.json keeps full precision)."""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        return float(f"{x:.{digits}g}")
    if isinstance(x, dict):
        return {k: _r(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, digits) for v in x]
    return x


def _pick(d, keys):
    return None if d is None else {k: d[k] for k in keys if k in d}


def compact_record(out: dict, detail_path: str | None) -> dict:
    """The line the driver parses: the contract's keys + roofline + cpu_baseline + a few scalars, < LINE_LIMIT bytes.
    Every other block of the run (`roofline_all`, `surface_cloud`, `reference_defaults`, `exchange_match`, ...) lives in the
    side file `detail_path` names."""
    cfg = out.get("config") or {}
    line = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                    "scaling", "vs_baseline", "dtype", "data")}
    line["config"] = {"workload": cfg.get("workload"), "sharding": cfg.get("sharding"), "library": cfg.get("library"),
                      "exchange": (cfg.get("exchange") or "")[:80]}
    line["roofline"] = _pick(out.get("roofline"), ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic",
                                                   "avg_launch_ms", "algorithmic_bytes_per_launch", "valu_issue_frac"))
    cb = out.get("cpu_baseline")
    if cb is not None:
        cb = _pick(cb, ("value", "unit", "cores", "kind", "sample", "sample_wall_seconds"))
        cb["sample"] = (cb.get("sample") or "")[:200]
    line["cpu_baseline"] = cb
    line["sustained"] = _pick(out.get("sustained"), ("ms_per_step_median", "ms_per_step_p95", "steps", "clock_mhz"))
    par = out.get("parity")
    if par is not None:
        par = _pick(par, ("ok", "rows", "fpfh_max_abs_err", "shot_max_abs_err", "tolerance"))
    line["parity"] = par
    for k in ("rccl_ranks", "per_rank_ms_per_step", "exchange_ms", "emulated", "emulated_rank", "emulated_world",
              "projected_value_upper_bound", "kernels_ms_per_step"):
        if out.get(k) is not None:
            line[k] = out[k]
    ss = out.get("strong_scaling")
    if ss is not None:
        line["strong_scaling"] = _pick(ss, ("ms_per_step", "graph_ms_per_step", "value", "parity_ok"))
    em = out.get("exchange_match")
    if em is not None:
        line["end_to_end_config5_ms"] = out.get("end_to_end_config5_ms")
        ov = em.get("allgather_under_k8")
        if isinstance(ov, dict) and "allgather_exposed_ms" in ov:  # the descriptor all-gather in chunks under K8: what stays exposed
            line["exchange_ms"] = dict(line.get("exchange_ms") or {}, allgather_chunks=ov["chunks"], allgather_exposed=ov["allgather_exposed_ms"],
                                       allgather_hidden=ov["allgather_hidden_ms"])
        if out.get("n_gpus", 1) == 1:
            line["scaling_note"] = "no scaling curve exists (one GPU); --dry-run-n 8 plans and runs every rank's share on this GPU"
    line["detail"] = detail_path
    line = _r(line)
    text = json.dumps(line, separators=(",", ":"))
    if len(text) >= LINE_LIMIT:  # never the driver's problem: shed the optional blocks, longest first
        for k in ("kernels_ms_per_step", "per_rank_ms_per_step", "strong_scaling", "sustained", "exchange_ms"):
            line.pop(k, None)
            if len(json.dumps(line, separators=(",", ":"))) < LINE_LIMIT:
                break
    return line


def emit_record(json_fd: int, out: dict, detail_name: str = DETAIL_FILE) -> None:
    """Full record -> side file (and stderr, for the log); compact line -> the saved stdout descriptor, LAST."""
    path = None
    try:
        path = os.path.join(os.environ.get("SF_BENCH_DETAIL_DIR", ROOT), detail_name)
        with open(path, "w") as f:
            json.dump(out, f, indent=1)
        path = os.path.relpath(path, ROOT)
    except OSError as exc:
        sys.stderr.write(f"bench.py: could not write {detail_name}: {exc}\n")
        path = None
    sys.stderr.write("bench.py detail record: " + json.dumps(out) + "\n")
    sys.stderr.flush()
    text = json.dumps(compact_record(out, path), separators=(",", ":"))
    assert len(text) < LINE_LIMIT, len(text)
    os.write(json_fd, (text + "\n").encode())


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
    # the calibration against the reference itself, and the reference's own times at config 2 / a 200 000-point slice of config 3
    # (tools/cpu_reference_r6.py, build container: the reference cannot travel to the GPU box)
    cal, ref_runs = None, None
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "r06_cpu_reference.json")))
        cal = dict(rec["calibration_of_numpy_shaped"])
        cal["ratio_total"] = cal["ratio_total"]
        ref_runs = {"config2_as_written": {k: rec["config2_as_written"][k] for k in ("fpfh_s", "shot_s", "descriptors_per_s")},
                    "config3_200k_slice": {k: rec["config3_200k_slice"][k] for k in ("fpfh_s", "shot_s", "descriptors_per_s", "implied_seconds_for_config3_full")},
                    "host_cores": rec["host"]["cores"], "versions": rec["versions"], "source": "profiles/r06_cpu_reference.json"}
    except Exception:
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
        "sample_wall_seconds": r["fpfh_s"] + r["shot_s"],
        "host_cores_available": os.cpu_count(),
        "calibration_vs_reference": None if cal is None else {
            "ratio_time_restatement_over_reference": cal["ratio_total"], "reference_desc_per_s_build_container": cal["ref_desc_per_s"],
            "max_abs_diff_vs_reference": max(cal["fpfh_max_abs_diff"], cal["shot_max_abs_diff"]),
            "where": "build container, 8 vCPU (tools/cpu_reference_r6.py; the reference cannot travel to the GPU box)"},
        "reference_itself_build_container": ref_runs,
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


def rows_checksum(job) -> dict:
    """A partition-independent fingerprint of the outputs a job left in HBM: per row a 64-bit hash of its bit pattern,
    multiplied by an odd function of the row's ORIGINAL point index, summed modulo 2^64.  Equal for any number of ranks
    exactly when every descriptor row is bit-identical (sum the per-rank values modulo 2^64)."""
    orig = job.block_original_indices().astype(np.uint64)
    key = 2 * orig + 1
    out = {}
    for name, arr in (("fpfh", job.fpfh_out), ("shot", job.shot_out)):
        if arr is None:
            continue
        mult = np.random.default_rng(1234).integers(1, 2**63, arr.shape[1], dtype=np.uint64) | np.uint64(1)
        total = np.uint64(0)
        with np.errstate(over="ignore"):
            for r0 in range(0, job.m, 65536):
                rows = arr.rows_to_host(r0, min(65536, job.m - r0)).view(np.uint64)
                h = (rows * mult[None, :]).sum(axis=1, dtype=np.uint64)
                total = total + (h * key[r0:r0 + rows.shape[0]]).sum(dtype=np.uint64)
        out[name] = int(total)
    return out


def fold_checksums(parts: list) -> dict:
    return {k: format(sum(p[k] for p in parts) % 2**64, "016x") for k in parts[0]}


COUNTERS_APPLY = True  # (main(): False for a workload other than the one the counter passes ran -- 1M points per GPU, radius 0.03)


def load_stamped(name: str, build: str):
    """A measurement kept under profiles/ (HBM traffic per launch, SQ counters) -- but only if it was taken on the very
    build that is loaded now (tools/parse_rocprof.py stamps the file with sf_version()) and on this workload."""
    if not COUNTERS_APPLY:
        return None, f"profiles/{name} holds counters of the default workload (1M points per GPU, radius 0.03): not quoted for this one"
    path = os.path.join(ROOT, "profiles", name)
    try:
        d = json.load(open(path))
    except Exception:  # noqa: BLE001 -- absent or unreadable: nothing to quote
        return None, f"profiles/{name} absent"
    if d.get("_build") != build:
        return None, f"profiles/{name} was measured on build {d.get('_build')!r}, this is {build!r}: not quoted"
    return d, d.get("_source", f"profiles/{name}")


def time_steps(job, eng, steps: int, warmup: int, barrier, max_over_ranks, with_timers: bool):
    """W untimed steps, then exactly K steps between barrier + device sync on both sides; max over ranks.
    Returns (seconds for the K steps, profile of the timed steps, profile of a few fully bracketed steps after them)."""
    # Two HIP event records per launch cost the step 1.6 % (4.55 against 4.48 ms), so the timed region brackets the roofline
    # kernel only -- roofline.achieved is measured live over the timed launches, as the contract asks; which kernel that is
    # comes from the warm-up steps (every launch bracketed; the first step, which grows the pools, left out when there are
    # two or more), and the breakdown of the other kernels from a few untimed steps after the timed ones.
    warm_rep = {}
    for w in range(warmup):
        if w == min(1, warmup - 1):
            eng.sync()
            eng.profile_reset()
            eng.profile(True)
        job.step()
    if warmup:
        eng.sync()
        eng.profile(False)
        warm_rep = eng.profile_report()
    timed_only = None
    if warm_rep and with_timers:
        cand = [k for k, v in warm_rep.items() if k in ALG_BYTES and k.startswith(("k5_", "k6_", "k7_", "k2_")) and v[0] > 0 and v[1] > 0]
        if cand:
            timed_only = max(cand, key=lambda k: warm_rep[k][1])
    barrier()
    eng.profile_reset()
    eng.profile_only(timed_only)
    eng.profile(with_timers)
    t0 = time.perf_counter()
    for _ in range(steps):
        job.step()
    barrier()
    elapsed = time.perf_counter() - t0
    eng.profile(False)
    eng.profile_only(None)
    time_steps.own_elapsed = elapsed  # (this rank's own time: the record lists every rank's)
    elapsed = max_over_ranks(elapsed)
    rep = eng.profile_report()
    extra_rep, extra_steps = {}, 0
    if timed_only is not None:
        # the breakdown of the other kernels: a few more steps, untimed, every launch bracketed, with the clocks where the
        # timed steps left them (the warm-up steps run 5-10 % slow and only chose the roofline kernel)
        extra_steps = min(steps, 5)
        eng.profile_reset()
        eng.profile(True)
        for _ in range(extra_steps):
            job.step()
        eng.sync()
        eng.profile(False)
        extra_rep = eng.profile_report()
        barrier()
    return elapsed, rep, timed_only, extra_rep, extra_steps


def sustained_window(job, eng, ms_per_step_short: float, min_seconds: float, min_steps: int, barrier, max_over_ranks) -> dict:
    """A long window of the SAME step right after the driver's K steps: at least `min_seconds` and `min_steps` steps, timed in
    chunks of 10 steps (one host synchronisation per chunk: < 0.1 % of a chunk) so that a median and a 95th percentile exist.
    On a chip that lowers its clock under load a 20-step window is not a sustained figure; this one is."""
    chunk = 10
    steps = max(min_steps, int(np.ceil(min_seconds * 1000.0 / max(ms_per_step_short, 1e-3))))
    steps = -(-steps // chunk) * chunk
    per_chunk = []
    barrier()
    t_all = time.perf_counter()
    for c in range(steps // chunk):
        t0 = time.perf_counter()
        for _ in range(chunk):
            job.step()
        eng.sync()
        per_chunk.append((time.perf_counter() - t0) / chunk * 1e3)
    barrier()
    window = max_over_ranks(time.perf_counter() - t_all)
    a = np.asarray(per_chunk)
    return {"steps": steps, "window_s": window, "chunk_steps": chunk, "ms_per_step_mean": 1000.0 * window / steps,
            "ms_per_step_median": float(np.median(a)), "ms_per_step_p95": float(np.percentile(a, 95)),
            "ms_per_step_min": float(a.min()), "ms_per_step_max": float(a.max()),
            "first_tenth_over_last_tenth": float(a[: max(len(a) // 10, 1)].mean() / a[-max(len(a) // 10, 1):].mean()),
            }


def density_line(eng, name: str, n: int, kbar: float, steps: int, parity_rows: int) -> dict:
    """The same FPFH + SHOT pass on a cloud that is not a uniform volume (tools/bench_density.py): a surface, the data the
    reference is run on (scripts/parse_args.py:6-22: Stanford scans), or clusters two orders of magnitude denser than
    their background -- radius chosen for the same mean neighbourhood size as the headline cloud."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_density

    r = bench_density.run(eng, name, n, kbar, steps, parity_rows=parity_rows)
    k = r["kernels_ms_per_step"]
    r["fallback_launches"] = {x: r["launches_per_step"][x] for x in ("k2_radius_fill", "k4_shot_lrf") if x in r["launches_per_step"]}
    r["second_launches_ms"] = {x: k[x] for x in k if x.endswith("_tail") or x in ("k2_radius_refill", "k2_select")}
    return r


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
    ap.add_argument("--cpu-sample", type=int, default=50000, help="points of the reference-shaped CPU baseline's sample")
    ap.add_argument("--only", choices=["both", "fpfh", "shot"], default="both")
    ap.add_argument("--overlap", action="store_true",
                    help="run the FPFH and SHOT chains on two HIP streams (faster; per-kernel times then overlap)")
    ap.add_argument("--no-match", action="store_true", help="skip the exchange + matching phase (config 5's tail)")
    ap.add_argument("--match-rows", type=int, default=None,
                    help="keypoints of the matched subset, whole job (default 262144 at N = 1, 131072 x N otherwise)")
    ap.add_argument("--match-steps", type=int, default=2)
    ap.add_argument("--match-chunks", type=int, default=None,
                    help="pieces of the descriptor all-gather, gathered on the side stream while K8 works on the piece that has "
                         "landed (default: 4 at N > 1, 1 at N = 1 -- where a second, chunked pass is measured beside the timed one)")
    ap.add_argument("--dry-run-n", type=int, default=None, metavar="N",
                    help="plan a --gpus N run on THIS one GPU and exit: the N x points-per-gpu cloud's layer table, every rank's block / "
                         "halo / interior / exchange operations and buffer sizes, the matching phase's buffers -- nothing is timed")
    ap.add_argument("--dry-run-n8", dest="dry_run_n", action="store_const", const=8, help="the same for N = 8 (--dry-run-n 8)")
    ap.add_argument("--no-ransac", action="store_true")
    ap.add_argument("--no-dropin", action="store_true", help="skip the host-to-host drop-in timing (N = 1)")
    ap.add_argument("--no-normals", action="store_true", help="skip the compute_normals line (N = 1)")
    ap.add_argument("--no-strong", action="store_true", help="skip the strong-scaling line (N > 1)")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--parity-rows", type=int, default=300)
    ap.add_argument("--checksum", action="store_true",
                    help="also report a partition-independent 64-bit fingerprint of the descriptor rows (N = 1: of the timed "
                         "outputs; N > 1: of the strong-scaling outputs, the SAME cloud): equal values = bit-identical rows")
    ap.add_argument("--no-kernel-timers", action="store_true",
                    help="diagnostic: time the steps without the per-kernel HIP events (no roofline / kernels_ms_per_step then)")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="FUNCTIONAL TEST ONLY: allow more ranks than GPUs; every exchange is then staged through host memory "
                         "and the control plane (RCCL refuses two ranks on one device) and no timing is a scaling result")
    ap.add_argument("--allow-staged", action="store_true",
                    help="N > 1: if the RCCL communicator cannot be built, run anyway with every exchange staged through host memory "
                         "(a functional test; without this flag the run exits with code 3 and prints no record)")
    ap.add_argument("--sustained-seconds", type=float, default=2.0,
                    help="length of the sustained window run after the K timed steps (0: skip)")
    ap.add_argument("--sustained-steps", type=int, default=500, help="... and its minimum number of steps")
    ap.add_argument("--no-density", action="store_true", help="skip the surface_cloud / clustered_cloud lines (N = 1)")
    ap.add_argument("--no-defaults", action="store_true", help="skip the reference_defaults lines (N = 1)")
    ap.add_argument("--emulate-rank", type=int, default=None, metavar="R",
                    help="single process, no rendezvous: run rank R's share of a --gpus N descriptor pass on this GPU (the rows "
                         "the adjacent ranks would send are computed once beforehand; the exchange call itself is skipped)")
    args = ap.parse_args()
    global COUNTERS_APPLY
    COUNTERS_APPLY = args.points_per_gpu == 1_000_000 and abs(args.radius - 0.03) < 1e-12

    emulated = args.emulate_rank is not None
    if args.dry_run_n:
        return dry_run_n(args)
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

    ctl = Control(rank, world) if world > 1 and not emulated else None

    import shot_fpfh_amd as s
    from shot_fpfh_amd import _ffi
    from shot_fpfh_amd.engine import Spfh
    from shot_fpfh_amd.sharding import DescriptorJob, SubsetMatchJob

    lib = _ffi.load()
    build = lib.sf_version().decode()
    n_dev = max(lib.sf_device_count(), 1)
    oversub = world > n_dev and not emulated
    if oversub and not args.oversubscribe:
        raise SystemExit(f"{world} ranks but {n_dev} GPU(s): one process per GPU is the only measured configuration "
                         f"(--oversubscribe runs a functional test with a host-staged exchange)")
    eng = s.Engine(local_rank % n_dev)
    exchange = "none (descriptor pass only)"
    rccl_ranks = 0

    def stage_through_host(why: str) -> str:
        """Every exchange through host memory and the control plane -- a functional stand-in, never a measurement."""

        def host_staged_allgather(buf, bytes_per_rank):
            flat = buf.to_host().reshape(-1).view(np.uint8)
            parts = ctl.allgather(flat[rank * bytes_per_rank:(rank + 1) * bytes_per_rank].tobytes())
            flat[: world * bytes_per_rank] = np.frombuffer(b"".join(parts), dtype=np.uint8)
            buf.from_host(flat.view(buf.dtype).reshape(buf.shape))

        def host_staged_rows(self, ops):
            ops = list(ops)
            mail = ctl.allgather({peer: self.rows_image(sb, se) for peer, sb, se, _, _ in ops if se > sb})
            for peer, _, _, rb, re in ops:
                if re > rb:
                    self.set_rows_image(rb, re, mail[peer][rank])

        def host_staged_min(buf, n=None):
            flat = buf.to_host().reshape(-1)
            n = flat.size if n is None else n
            flat[:n] = np.min(np.stack(ctl.allgather(flat[:n].copy())), axis=0)
            buf.from_host(flat.reshape(buf.shape))

        def host_staged_exchange(ops):
            """Engine.exchange (grouped ncclSend / ncclRecv of byte ranges of device arrays) over the control plane"""
            ops = list(ops)
            mail = ctl.allgather({peer: sbuf.to_host().reshape(-1).view(np.uint8)[soff:soff + sbytes].tobytes()
                                  for peer, sbuf, soff, sbytes, _, _, _ in ops if sbytes})
            for peer, _, _, _, rbuf, roff, rbytes in ops:
                if rbytes:
                    flat = rbuf.to_host().reshape(-1).view(np.uint8)
                    flat[roff:roff + rbytes] = np.frombuffer(mail[peer][rank], dtype=np.uint8)
                    rbuf.from_host(flat.view(rbuf.dtype).reshape(rbuf.shape))

        eng.allgather = host_staged_allgather
        eng.exchange = host_staged_exchange
        eng.allreduce_min_u64 = host_staged_min
        # the ranks size their SPFH tables by the longest list of ANY rank (same storage and wire format everywhere): without
        # RCCL that maximum goes over the control plane -- DescriptorJob asks for it through this hook (sharding.py)
        eng.collective_stats = lambda on: None
        eng.fold_max_count = lambda v: int(max(ctl.allgather(int(v))))
        Spfh.exchange_rows = host_staged_rows
        if args.spfh_exchange == "allgather":
            raise SystemExit("the staged exchange supports the neighbor and halo SPFH modes only")
        return f"control plane, staged through host memory -- {why}"

    if not emulated:
        if oversub:
            exchange = stage_through_host("oversubscribed functional test, NOT RCCL")
        else:
            # RCCL communicator -- one rank too, so that the collectives really execute.  If it cannot be built on this node
            # the run still completes, staged through host memory, and the record says so instead of the whole bench dying.
            err = ""
            try:
                uid = ctl.bcast(eng.comm_unique_id() if rank == 0 else None) if ctl else eng.comm_unique_id()
                eng.comm_init(uid, world, rank)
                # pre-flight: the three RCCL call shapes of the run, on a few bytes, checked -- a node where point-to-point
                # between two GPUs does not work should say so here, not inside the timed loop
                probe = eng.empty((2 * world + 2,), np.uint64).from_host(np.full(2 * world + 2, rank + 1, dtype=np.uint64))
                up, down = (rank + 1) % world, (rank - 1) % world
                ops = [(up, probe, 0, 8, probe, 8 * world, 8)]
                if down != up:  # (two ranks: both neighbours are the same peer -- one operation, as in the descriptor pass)
                    ops.append((down, probe, 0, 8, probe, 8 * (world + 1), 8))
                eng.exchange(ops)
                eng.allgather(probe, 8)
                eng.sync()
                got = probe.to_host()
                want_gather = np.arange(1, world + 1, dtype=np.uint64)
                if world > 1 and not (np.array_equal(got[:world], want_gather) and got[world] == up + 1 and (down == up or got[world + 1] == down + 1)):
                    raise RuntimeError(f"RCCL pre-flight returned wrong data on rank {rank}: {got.tolist()}")
                eng.allreduce_min_u64(probe, 1)
                eng.sync()
                probe.free()
            except Exception as exc:  # noqa: BLE001 -- reported in the record
                err = f"{type(exc).__name__}: {exc}"
            errs = ctl.allgather(err) if ctl else [err]
            if any(errs):
                first = next(e for e in errs if e)
                if ctl is None or not args.allow_staged:
                    # a multi-GPU line measured over host memory would be read as an xGMI result: fail, loudly, on every rank
                    # (all of them reach this point: the error strings were all-gathered over the control plane)
                    if lead:
                        sys.stderr.write(f"bench.py: the RCCL communicator of {world} rank(s) could not be built or failed its pre-flight "
                                         f"({first[:400]}); no record is printed.  --allow-staged runs the same job with every exchange "
                                         f"staged through host memory (a functional test, never a measurement).\n")
                    if ctl is not None:
                        ctl.close()
                    return 3
                exchange = stage_through_host(f"RCCL communicator failed ({first[:200]})")
            else:
                exchange = f"RCCL over {world} rank(s): ncclSend/ncclRecv (SPFH rows), ncclAllGather (descriptor rows)"
                rccl_ranks = world

    n_total = args.points_per_gpu * world
    radius = args.radius * world ** (-1.0 / 3.0)
    points, normals = make_cloud(n_total, 3)
    job = DescriptorJob(eng, points, normals, radius, n_bins=5, normalize=True, min_neighborhood_size=10, world=world,
                        rank=rank, spfh_exchange=args.spfh_exchange, do_fpfh=args.only in ("both", "fpfh"),
                        do_shot=args.only in ("both", "shot"), overlap_chains=args.overlap, emulate_peers=emulated)

    def barrier():
        eng.sync()
        if ctl is not None:
            ctl.barrier()

    def max_over_ranks(x: float) -> float:
        return x if ctl is None else ctl.max(x)

    # ---- timed region: K descriptor passes ---------------------------------------------------------------------------
    elapsed, rep, timed_only, warm_rep, warm_steps = time_steps(job, eng, args.steps, args.warmup, barrier, max_over_ranks,
                                                                not args.no_kernel_timers)
    kinds = (1 if job.do_fpfh else 0) + (1 if job.do_shot else 0)
    n_desc = kinds * n_total
    ms_per_step = 1000.0 * elapsed / args.steps
    value = n_desc / (elapsed / args.steps)
    own_ms = 1000.0 * time_steps.own_elapsed / args.steps
    per_rank_ms = [own_ms] if ctl is None else ctl.allgather(own_ms)
    # ---- sustained: the same step for >= 2 s (>= 500 steps), no per-kernel events; if it disagrees with the K-step figure by
    #      more than 2 % the sustained figure is the one `value` reports (and the record says so) ------------------------------
    sustained = None
    value_source = (f"the {args.steps} timed steps (every step repeats the search of the same range: planned from the previous "
                    f"search's record, no read-back -- `k2_slot_capacity` in the detail file has the step without it)")
    if args.sustained_seconds > 0 and not args.no_kernel_timers:
        sustained = sustained_window(job, eng, ms_per_step, args.sustained_seconds, args.sustained_steps, barrier, max_over_ranks)
        sustained["vs_timed_steps"] = sustained["ms_per_step_median"] / ms_per_step
        ck, ck_src = load_stamped(f"{PROFILE_TAG}_sustained_clock.json", build)
        sustained["clock_mhz"] = None if ck is None else ck.get("clock_mhz_time_weighted_hot_kernels")
        sustained["clock_source"] = ck_src if ck is None else ck.get("_source")
        if abs(sustained["vs_timed_steps"] - 1.0) > 0.02:
            value = n_desc / (sustained["ms_per_step_median"] * 1e-3)
            value_source = (f"the sustained window's median step ({sustained['ms_per_step_median']:.4f} ms over {sustained['steps']} steps): "
                            f"it differs from the {args.steps} timed steps ({ms_per_step:.4f} ms) by more than 2 %")

    if args.no_kernel_timers:
        if lead:
            os.write(json_fd, (json.dumps({"diagnostic": "steps timed without per-kernel HIP events", "ms_per_step": ms_per_step,
                                           "value": None if emulated else value, "n_gpus": 1 if emulated else world,
                                           "emulated": emulated, "emulated_rank": rank if emulated else None,
                                           "emulated_world": world if emulated else None,
                                           "projected_value_upper_bound": value if emulated else None,
                                           "steps": args.steps}) + "\n").encode())
        return 0
    out = {}
    if lead:
        kern = {k: (v[0], v[1] / max(v[0], 1)) for k, v in rep.items() if v[0] > 0 and v[1] > 0}
        per_step_ms = {k: rep[k][1] / args.steps for k in kern}
        launches_per_step = {k: rep[k][0] / args.steps for k in kern}
        if timed_only is not None:  # the other kernels: from the instrumented steps after the timed ones
            for k, v in warm_rep.items():
                if k not in per_step_ms and v[0] > 0 and v[1] > 0:
                    per_step_ms[k] = v[1] / max(warm_steps, 1)
                    launches_per_step[k] = v[0] / max(warm_steps, 1)
        dom = max((k for k in kern if k in ALG_BYTES), key=lambda k: rep[k][1])
        launches, avg_ms = kern[dom]
        units = job.m  # descriptors of this rank's block per launch (halo SPFH rows are extra work, not counted)

        def alg_bytes_of(k):
            return ALG_BYTES[k] * units + (4 * job.last_pairs if k in PER_PAIR else 0)

        alg_bytes = alg_bytes_of(dom)
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
        tj, traffic_src = load_stamped(f"{PROFILE_TAG}_traffic.json", build)
        traffic = None if tj is None else tj.get(dom)
        # ---- every kernel of the step against both HBM ceilings (per STEP: a kernel launched in pieces counts once) ------------
        roof_all = {}
        for k, ms in sorted(per_step_ms.items()):
            if k not in ALG_BYTES or ms <= 0:
                continue
            gbs = alg_bytes_of(k) / (ms * 1e-3) / 1e9
            roof_all[k] = {"algorithmic_bytes_per_step": alg_bytes_of(k), "ms_per_step": round(ms, 4),
                           "launches_per_step": round(launches_per_step.get(k, 0), 2), "achieved_gbs": round(gbs, 1),
                           "frac_of_8000": round(gbs / HBM_PEAK_GBS, 4), "frac_of_6290": round(gbs / HBM_COPY_GBS, 4),
                           "hbm_bytes_measured": None if tj is None else tj.get(k)}
        sq, sq_src = load_stamped(f"{PROFILE_TAG}_k5_sq.json", build)
        if sq is not None and "k5_shot" in roof_all:
            # float64 issue ceiling: a wave's vector instructions occupy its SIMD for SQ_ACTIVE_INST_VALU quad-cycles (x 4
            # cycles); one keypoint = one wave; 256 CUs x 4 SIMDs share the keypoints
            cyc = sq["SQ_ACTIVE_INST_VALU_per_wave"] * 4.0
            ceiling_ms = units * cyc / (256 * 4) / (sq["clock_mhz"] * 1e3)
            roof_all["k5_shot"]["valu_issue"] = {
                "vector_instructions_per_keypoint": sq["SQ_INSTS_VALU_per_wave"], "busy_cycles_per_keypoint": cyc,
                "simds": 1024, "clock_mhz": sq["clock_mhz"], "ceiling_ms": round(ceiling_ms, 4),
                "ceiling_over_measured": round(ceiling_ms / roof_all["k5_shot"]["ms_per_step"], 4),
                "note": "counters and clock were taken under the profiler (the kernel runs ~8 % slower there, at a lower clock): a "
                        "ratio within a few per cent of 1 means the kernel's time IS its vector-issue time",
                "source": sq_src}
        out = {
            "metric": "descriptors/sec (SHOT+FPFH) on 1M-pt cloud",
            # an EMULATED record (--emulate-rank) is ONE GPU running one rank's share with the exchange call skipped: it has no
            # multi-GPU value; what its step time would project to is named as what it is
            "value": None if emulated else value,
            "unit": "descriptors/s",
            "n_gpus": 1 if emulated else world,
            "emulated": emulated,
            "emulated_rank": rank if emulated else None,
            "emulated_world": world if emulated else None,
            "projected_value_upper_bound": value if emulated else None,
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
                "sharding": f"query blocks over {world} GPU(s), cloud replicated, SPFH {args.spfh_exchange}"
                + (f" -- EMULATED: rank {rank} of {world} alone on one GPU, the exchange call skipped (c_exchange not in the step)" if emulated else ""),
                "points_per_gpu": args.points_per_gpu,
                "exchange": exchange,
                "library": build,
            },
            "value_source": value_source,
            "sustained": sustained,
            "per_rank_ms_per_step": [round(x, 4) for x in per_rank_ms],
            "rccl_ranks": rccl_ranks,
            "value_includes": "the descriptor pass of every rank (K1 K2 K6 K7 K4 K5) INCLUDING the neighbour exchange of SPFH rows it "
                              "contains at N > 1 (sf_spfh_exchange_rows on the side stream; `exchange_ms` below is its own launch "
                              "bracket); EXCLUDING the all-gather of descriptor rows and the matching, which are measured under "
                              "`exchange_match`" if not emulated else "one rank's share on one GPU, exchange call skipped",
            "exchange_ms": {k: round(v, 4) for k, v in sorted(per_step_ms.items()) if k.startswith("c_")},
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
                "frac_of_measured_copy_peak": achieved / HBM_COPY_GBS,
                "traffic": traffic,
                "traffic_source": traffic_src if traffic is None else
                f"per-launch HBM bytes from separate rocprofv3 --pmc passes on THIS build ({build}), {traffic_src}",
                "avg_launch_ms": avg_ms,
                "algorithmic_bytes_per_launch": alg_bytes,
                # share of the kernel's time its SIMDs spend issuing float64 vector instructions (counters of this build,
                # roofline_all.k5_shot.valu_issue): what actually bounds the kernel
                "valu_issue_frac": (roof_all.get(dom) or {}).get("valu_issue", {}).get("ceiling_over_measured"),
                "note": ("the kernel is float64-VALU-issue bound, not HBM bound (roofline_all.k5_shot.valu_issue); the HBM fraction is "
                         "reported because the contract asks for it") if dom == "k5_shot" else None,
            },
            "roofline_all": roof_all,
            "whole_step_hbm": {"algorithmic_bytes": sum(ALG_BYTES[k] for k in ("k6_spfh", "k7_fpfh", "k5_shot")) * units,
                               "achieved_gbs": sum(ALG_BYTES[k] for k in ("k6_spfh", "k7_fpfh", "k5_shot")) * units / (ms_per_step * 1e-3) / 1e9}
            if kinds == 2 else None,
        }

    # ---- K2's slot capacity: the timed steps size it from the statistics of the previous search with this radius on the cloud
    #      (a capacity hint, never an output: a list that outgrows its slot is re-done exactly).  What the step costs when
    #      every search counts a fresh 2 048-query sample instead (SF_K2_NO_HINT=1, what a FIRST search on a cloud does): ------
    if lead is not None and not args.no_kernel_timers and args.sustained_seconds > 0:
        os.environ["SF_K2_NO_HINT"] = "1"
        try:
            job.step()
            barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                job.step()
            barrier()
            t_nohint = max_over_ranks(time.perf_counter() - t0) / args.steps
        finally:
            del os.environ["SF_K2_NO_HINT"]
        if lead:
            out["k2_slot_capacity"] = {
                "timed_steps_use": "the record of the previous search of this range with this radius on this cloud (sf_cloud::search_records: "
                                   "total, longest list, histogram of the lengths, slot size) -- the lists of a self search are a function "
                                   "of (cloud, radius, range), so a repeated step launches its sweep and plans every later launch without a "
                                   "statistics pass or a read-back",
                "ms_per_step_with_a_sample_counted_in_every_search": 1000.0 * t_nohint,
                "ms_per_step_timed": ms_per_step,
                "what": "SF_K2_NO_HINT=1: every radius search first counts the lists of 2 048 sampled queries (one small launch, "
                        "8 KB read back) before it sizes its slots, and ends with the statistics pass and its read-back -- the cost "
                        "of a FIRST search of a range (what a one-off drop-in call pays)"}

    # ---- the same step with the FPFH chain (K6, K7) and the SHOT chain (K4, K5) on two HIP streams: K7 waits on memory where
    #      K5 waits on its vector pipe, side by side they fill each other's gaps.  NOT what `value` reports: the per-kernel
    #      durations of the timed steps (the roofline's denominators) would then overlap ------------------------------------
    if single and kinds == 2 and not args.overlap and not args.no_kernel_timers and args.sustained_seconds > 0:
        job.overlap = True
        try:
            for _ in range(3):
                job.step()
            barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                job.step()
            barrier()
            t_two = (time.perf_counter() - t0) / args.steps
        finally:
            job.overlap = False
        job.step()  # (what the parity block below reads is a step of the timed configuration)
        eng.sync()
        out["two_streams"] = {"ms_per_step": 1000.0 * t_two, "value": n_desc / t_two, "unit": "descriptors/s", "vs_timed_steps": 1000.0 * t_two / ms_per_step,
                              "what": "DescriptorJob(overlap_chains=True) / bench.py --overlap: the FPFH and SHOT chains on the context's two HIP "
                                      "streams; same rows bit for bit (tests/test_hip_parity.py::test_two_stream_overlap_gives_identical_results), kernels overlap in time"}

    # ---- the same step replayed as ONE launch (HIP graph: DescriptorJob.step_replay).  Same launches, same rows; what it saves
    #      is host time, so it shows on small steps (strong scaling below), not on this one.  Secondary: `value` is the eager step ----
    if (single or emulated) and not args.no_kernel_timers and args.sustained_seconds > 0:
        try:
            for _ in range(4):  # (two eager steps, the capture, one replay)
                job.step_replay()
            barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                job.step_replay()
            barrier()
            t_g = (time.perf_counter() - t0) / args.steps
            out["graph_replay"] = {"ms_per_step": 1000.0 * t_g, "vs_timed_steps": 1000.0 * t_g / ms_per_step,
                                   "captured": getattr(job, "_graph", None) is not None, "not_captured_because": getattr(job, "_graph_failed", None),
                                   "what": "DescriptorJob.step_replay(): the step captured once into a HIP graph (sf_graph_begin / sf_graph_end) "
                                           "and replayed with one call per step; rows bit-identical (tests/test_hip_round5.py)"}
        except Exception as exc:  # noqa: BLE001 -- a secondary line must not take the record down
            out["graph_replay"] = {"error": f"{type(exc).__name__}: {exc}"}

    # ---- parity of what the timed steps left in HBM ------------------------------------------------------------------
    if not args.no_parity:
        par = parity_sample(job, points, normals, radius, args.parity_rows)
        if ctl is not None:
            allp = ctl.allgather(par)
            par = {"per_rank": allp, "ok": all(p["ok"] for p in allp)}
        if lead:
            out["parity"] = par
    if args.checksum and single:
        out["checksum"] = fold_checksums([rows_checksum(job)])

    # ---- strong scaling: the metric's own 1M-point cloud cut over the N ranks (secondary; `value` stays weak) ---------------
    if world > 1 and not args.no_strong and kinds == 2:
        sp_pts, sp_nrm = make_cloud(args.points_per_gpu, 3)
        sjob = DescriptorJob(eng, sp_pts, sp_nrm, args.radius, n_bins=5, normalize=True, min_neighborhood_size=10, world=world,
                             rank=rank, spfh_exchange=args.spfh_exchange, emulate_peers=emulated)
        # (a block is 1 / world of the N = 1 step: `world` times the steps and warm-up steps -- the same device time and the same
        # number of descriptors as the weak line, so the two are timed at the same clocks)
        s_steps, s_warm = args.steps * world, max(args.warmup, 1) * world
        s_elapsed, _, _, _, _ = time_steps(sjob, eng, s_steps, s_warm, barrier, max_over_ranks, False)
        s_graph = None
        if emulated or ctl is None:  # (with real peers the exchange's RCCL calls would have to be captured: step_replay runs step())
            try:
                for _ in range(4):
                    sjob.step_replay()
                barrier()
                t0 = time.perf_counter()
                for _ in range(s_steps):
                    sjob.step_replay()
                barrier()
                s_graph = (time.perf_counter() - t0) / s_steps if getattr(sjob, "_graph", None) is not None else None
            except Exception:  # noqa: BLE001
                s_graph = None
        spar = None if args.no_parity else parity_sample(sjob, sp_pts, sp_nrm, args.radius, min(args.parity_rows, 100))
        if ctl is not None and spar is not None:
            spar = {"ok": all(p["ok"] for p in ctl.allgather(spar))}
        ssum = None
        if args.checksum and not emulated:
            mine = rows_checksum(sjob)
            ssum = fold_checksums(ctl.allgather(mine) if ctl is not None else [mine])
        sjob.close()
        if lead:
            out["strong_scaling"] = {
                "what": f"the {args.points_per_gpu}-point cloud of N = 1 (radius {args.radius}) cut into {world} blocks of "
                        f"{-(-args.points_per_gpu // world)} keypoints: total work fixed",
                "ms_per_step": 1000.0 * s_elapsed / s_steps, "steps": s_steps, "warmup": s_warm,
                "graph_ms_per_step": None if s_graph is None else 1000.0 * s_graph,
                "value": 2 * args.points_per_gpu / (s_elapsed / s_steps), "unit": "descriptors/s",
                "speedup_vs_n1_needs": "the N = 1 line of the same build (ms_per_step there / ms_per_step here)",
                "parity_ok": None if spar is None else spar["ok"],
                "checksum": ssum,
            }

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
        ref_job.step()  # (once more untimed: the pool's blocks settle over the first passes -- a 30 ms pass among 3 ms ones otherwise)
        eng.sync()
        barrier()
        t0 = time.perf_counter()
        ref_job.step()  # (the partner cloud's SHOT pass once more, warm: a term of end_to_end_config5_ms)
        barrier()
        t_ref_step = max_over_ranks(time.perf_counter() - t0)
        scan_orig = job.block_original_indices()            # scan rows of this rank -> scan point
        ref_label = perm[ref_job.block_original_indices()]  # ref rows of this rank -> the scan point they came from
        s_sel, r_sel = np.flatnonzero(scan_orig < total_rows), np.flatnonzero(ref_label < total_rows)
        cap = max_over_ranks(float(max(s_sel.size, r_sel.size, 1)))
        cap = int(-(-int(cap) // 256) * 256)
        match_chunks = args.match_chunks or (1 if single else 4)
        sub = SubsetMatchJob(eng, 352, cap, world, rank, chunks=match_chunks)
        sub.select(job.shot_out, s_sel, scan_orig[s_sel], ref_job.shot_out, r_sel, ref_label[r_sel])
        for _ in range(3):  # warm-up: RCCL channel setup, and the pool's blocks settling (its best-fit reuse hands a pass's large
            sub.run()       # blocks to other requests of the next pass until it holds one of every size: a 50 ms pass among 26 ms ones)
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

        # ---- how much of the descriptor all-gather K8 hides: the chunked job's pass, its K8 alone (on the rows the previous pass
        #      gathered) and its gathers alone, each between barriers.  At N = 1 a second job with 4 chunks is measured beside the
        #      timed one-shot job (the gather is then a local copy through the one-rank communicator: the mechanics, not xGMI) ----
        def exposure(j, n_chunks):
            def timed(**kw):
                j.run(**kw)
                barrier()
                t0 = time.perf_counter()
                j.run(**kw)
                barrier()
                return max_over_ranks(time.perf_counter() - t0)
            t_all, t_k8, t_ag = timed(), timed(gather=False), timed(match=False)
            exposed = max(t_all - t_k8, 0.0)
            return {"chunks": n_chunks, "pass_ms": 1000.0 * t_all, "k8_alone_ms": 1000.0 * t_k8, "gathers_alone_ms": 1000.0 * t_ag,
                    "allgather_exposed_ms": 1000.0 * exposed, "allgather_hidden_ms": 1000.0 * max(t_ag - exposed, 0.0)}
        overlap_rec = None
        try:
            if match_chunks > 1:
                overlap_rec = exposure(sub, match_chunks)
            elif single:
                sub4 = SubsetMatchJob(eng, 352, cap, world, rank, chunks=4)
                try:
                    sub4.select(job.shot_out, s_sel, scan_orig[s_sel], ref_job.shot_out, r_sel, ref_label[r_sel])
                    overlap_rec = exposure(sub4, 4)
                    a4, b4 = sub4.matches()
                    overlap_rec["same_matches_as_the_one_shot_gather"] = bool(np.array_equal(a4, s_lab) and np.array_equal(b4, r_lab))
                finally:
                    sub4.close()
        except Exception as exc:  # noqa: BLE001 -- a side measurement
            overlap_rec = {"error": f"{type(exc).__name__}: {exc}"}
        stats = np.array([float((s_lab == r_lab).sum()), float(s_lab.size)])
        if ctl is not None:
            stats = np.sum(np.stack(ctl.allgather(stats)), axis=0)
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
                "allgather_chunks": match_chunks,
                "allgather_under_k8": overlap_rec,
                "kernels_ms_per_pass": {k: round(v, 4) for k, v in sorted(k8.items())},
                "k8_pair_dists_per_s": cap * gathered / (k8_ms * 1e-3) if k8_ms > 0 else None,
                "k8_roofline": _k8_roofline(k8, k8_ms, flop),
                "matches": int(stats[1]),
                "matches_recovering_true_correspondence": float(stats[0] / max(stats[1], 1.0)),
                "match_pairs_allgather_s": t_pairs,
                "registration_from_all_matches": reg,
                "partner_cloud_descriptor_pass_s": t_prep,
                "partner_cloud_shot_pass_ms": 1000.0 * t_ref_step,
            }
            # BASELINE config 5 end to end on resident clouds: this cloud's FPFH + SHOT pass (the timed step), the partner cloud's
            # SHOT pass, and the exchange + matching pass (subset gather, all-gather of rows and labels, sharded K8) -- a sum of three
            # separately timed phases, each the maximum over the ranks
            out["end_to_end_config5_ms"] = ms_per_step + 1000.0 * t_ref_step + 1000.0 * t_match
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

    # ---- compute_normals(radius) on the same cloud (N = 1): K2 + K3, SURVEY 8d "timed as its own line" -------------------------
    if single and not args.no_normals and not emulated:
        from shot_fpfh_amd.descriptors import compute_normals

        ncloud = eng.cloud(points)  # (no normals: that is what is being computed)
        nout = eng.empty((n_total, 3))
        reps = 5

        def normals_pass():  # K1 + the fused sweep (no neighbour lists) + the eigen-solves
            ncloud.build_grid(radius)
            ncloud.normals_radius_self(radius, nout)

        normals_pass()
        eng.sync()
        eng.profile_reset()
        eng.profile(True)
        t0 = time.perf_counter()
        for _ in range(reps):
            normals_pass()
        eng.sync()
        t_n = (time.perf_counter() - t0) / reps
        eng.profile(False)
        nrep = {k: v[1] / reps for k, v in eng.profile_report().items() if v[1] > 0}
        k3 = nrep.get("k3_normals", 0.0) + nrep.get("k23_radius_cov", 0.0)
        # parity of the resident result: a sample against the oracle, up to the sign LAPACK leaves open
        from oracle import oracle as O

        pick = np.sort(np.random.default_rng(6).choice(n_total, 200, replace=False))
        # (rows are in cell-sorted order: map the sample through the inverse permutation)
        perm = ncloud.perm().astype(np.int64)
        inv = np.empty_like(perm)
        inv[perm] = np.arange(perm.size)
        got = np.stack([nout.rows_to_host(int(i), 1)[0] for i in inv[pick]])
        want = O.compute_normals(points[pick], points, radius=radius)
        nerr = float(np.minimum(np.abs(got - want).max(axis=1), np.abs(got + want).max(axis=1)).max())
        ncloud.free()
        nout.free()
        ts = []
        for _ in range(2):
            t0 = time.perf_counter()
            compute_normals(points, points, radius=radius)
            ts.append(time.perf_counter() - t0)
        out["normals"] = {
            "what": f"compute_normals(query_points = cloud_points = the {n_total}-point cloud, radius={radius}) "
                    "(pca_based_descriptors.py:29-59): K1 + one fused K2/K3 sweep (hits reduced to the covariance in LDS, no lists) + "
                    "eigen-solves, result resident / host to host",
            "resident_ms_per_pass": 1000.0 * t_n, "normals_per_s_resident": n_total / t_n,
            "kernels_ms_per_pass": {k: round(v, 4) for k, v in sorted(nrep.items())},
            "k3_roofline": {"bound": "hbm", "algorithmic_bytes_per_launch": ALG_BYTES["k3_normals"] * n_total, "avg_launch_ms": k3,
                            "achieved_gbs": ALG_BYTES["k3_normals"] * n_total / (k3 * 1e-3) / 1e9 if k3 > 0 else None,
                            "frac_of_8000": ALG_BYTES["k3_normals"] * n_total / (k3 * 1e-3) / 1e9 / HBM_PEAK_GBS if k3 > 0 else None,
                            "frac_of_6290": ALG_BYTES["k3_normals"] * n_total / (k3 * 1e-3) / 1e9 / HBM_COPY_GBS if k3 > 0 else None,
                            "note": "48 B per query are compulsory; the fused sweep tests ~3 candidates per neighbour (K2's own work) and "
                                    "solves a 3x3 eigenproblem per query: VALU-issue bound like K2 (DESIGN K3)"},
            "host_to_host_s": min(ts), "normals_per_s_host_to_host": n_total / min(ts),
            "parity_max_abs_err_up_to_sign": nerr, "parity_rows": int(pick.size), "parity_ok": bool(nerr <= 1e-5),
        }

    # ---- clouds that are not uniform volumes (N = 1): same pass, same mean neighbourhood size ------------------------------------
    if single and not args.no_density and not emulated and kinds == 2:
        kbar = job.last_pairs / max(job.m, 1)
        for key, name in (("surface_cloud", "surface"), ("clustered_cloud", "clustered")):
            d = density_line(eng, name, args.points_per_gpu, kbar, steps=10, parity_rows=0 if args.no_parity else args.parity_rows)
            d["ratio_to_uniform_step"] = d["ms_per_step"] / ms_per_step
            out[key] = d

    # ---- the reference's DEFAULT paths (N = 1): subsampled SHOT support, k-NN normals, bi- / multi-scale SHOT, 3-D matching ------
    if single and not args.no_defaults and not emulated:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_defaults

        out["reference_defaults"] = bench_defaults.run(eng, points, normals, radius, parity=not args.no_parity)

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
            # (a ratio against a CPU says nothing about kernel quality -- the roofline fractions do; what it means in seconds:)
            out["cpu_baseline"]["implied_seconds_for_this_workload"] = n_desc / shaped["value"]
            out["cpu_baseline"]["gpu_seconds_for_this_workload"] = n_desc / value
        emit_record(json_fd, out, DETAIL_FILE if not emulated else f"bench_detail_rank{rank}_of_{world}.json")
    os.close(json_fd)
    if ctl is not None:
        ctl.barrier()
        ctl.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
