"""RANSAC over descriptor matches -- drop-in for shot_fpfh.matching.ransac_on_matches
(ransac.py:17-82).

Host: the draws (same module-level np.random.default_rng(seed=72) stream as ransac.py:14, so draw
k of a process equals the reference's draw k -- produced for all draws at once from the generator's raw
output, `draw_stream`) and one Kabsch fit per draw -- solved in stacks through the same BLAS / LAPACK
routines (bit-identical transforms, checked against the per-draw solver on a sample of every stack).
Device: the gather keypoints[indices] of all matches and (K9) the O(n_draws x n_matches) inlier count of
every candidate transform, a chunk of draws per launch while the host fits the next chunk.
"""
from __future__ import annotations

import logging
from typing import Optional

import numpy as np
import numpy.typing as npt

from ..core import RigidTransform, solver_point_to_point
from ..core.geometry import solver_point_to_point_batched
from ..engine import Engine, default_engine

__all__ = ["ransac_on_matches", "rng"]

# same seed and same persistence across calls as the reference module's generator
rng = np.random.default_rng(seed=72)


# ---- the reference's draw stream, n_draws at a time -------------------------------------------------------------------------
# ransac.py:50-55 calls rng.choice(n, draw_size, replace=False, shuffle=False) once per draw: 10^4 generator calls, 60-80 ms of
# interpreter time around 4 x 10^4 random words.  What one such call does with the generator is fixed by NumPy
# (Generator.choice, Floyd's algorithm whenever draw_size <= n / 20 or n <= 10 000: for t = 0 .. size-1 one Lemire-bounded
# integer in [0, n - size + t] from the bit generator's 32-bit stream -- PCG64 hands out the low half of a 64-bit output, then
# the high half -- replaced by the bound itself when it repeats an earlier pick), so the same numbers can be produced for all
# draws at once from the raw 64-bit outputs and the generator left in exactly the state the loop would have left it in.  The
# replica is checked against Generator.choice itself the first time it is used (another NumPy could implement choice
# otherwise: the per-draw loop then stays) and a rejection of the bounded-integer sampler -- probability (2^32 mod bound) /
# 2^32 per word, some ten per 10^4 draws of 10^6 matches -- is replayed word by word.
def _draws_by_loop(gen: np.random.Generator, n: int, size: int, n_draws: int) -> np.ndarray:
    draws = np.empty((n_draws, size), dtype=np.int64)
    for d in range(n_draws):  # one generator call per draw, as ransac.py:50-55
        draws[d] = gen.choice(n, size, replace=False, shuffle=False)
    return draws


def _draws_from_raw_stream(gen: np.random.Generator, n: int, size: int, n_draws: int) -> Optional[np.ndarray]:
    """The draws of `n_draws` calls gen.choice(n, size, replace=False, shuffle=False), with `gen` advanced exactly as those
    calls advance it; None (generator untouched) where the replica does not apply."""
    bg = gen.bit_generator
    if type(bg).__name__ != "PCG64" or n_draws <= 0 or size < 1 or n - size < 1 or n > 0x7FFFFFFF:
        return None
    if n > 10000 and size > n // 20:  # (Generator.choice shuffles a tail of arange(n) there instead)
        return None
    st = bg.state
    words_needed = n_draws * size
    bounds = (np.arange(size, dtype=np.uint64) + np.uint64(n - size)) + np.uint64(1)  # rng_excl of pick t: j_t + 1
    thresh = (np.uint64(1 << 32) - bounds) % bounds  # Lemire's rejection threshold (UINT32_MAX - j) % (j + 1)
    draws = np.empty((n_draws, size), dtype=np.int64)

    def stream(n_words: int):
        """the next n_words 32-bit words of a COPY of the generator (incl. the buffered half-word) + how many were buffered"""
        clone = np.random.PCG64()
        clone.state = st
        raw = clone.random_raw((n_words + 1) // 2 + 1)
        u = np.empty(2 * raw.size, dtype=np.uint64)
        u[0::2] = raw & np.uint64(0xFFFFFFFF)
        u[1::2] = raw >> np.uint64(32)
        if st["has_uint32"]:
            u = np.concatenate([np.array([st["uinteger"]], dtype=np.uint64), u])
        return u

    extra = 64 + words_needed // 1000
    u = stream(words_needed + extra)
    pos, done, restarts = 0, 0, 0
    if float(thresh.sum()) / 4294967296.0 > 0.02:  # (a rejection in more than one draw of fifty: the loop is as fast)
        return None
    while done < n_draws:
        left = min(n_draws - done, 2048)  # (a window at a time: a rejection re-aligns only what follows it in the window)
        if pos + left * size > u.size:
            extra *= 4
            u = stream(pos + left * size + extra)
        w = u[pos:pos + left * size].reshape(left, size)
        m = w * bounds
        rejected = ((m & np.uint64(0xFFFFFFFF)) < thresh).any(axis=1)
        good = int(np.argmax(rejected)) if rejected.any() else left
        if good:
            vals = (m[:good] >> np.uint64(32)).astype(np.int64)
            out = draws[done:done + good]
            out[:, 0] = vals[:, 0]
            for t in range(1, size):  # Floyd: a repeated value is replaced by the bound j_t (larger than every earlier pick)
                dup = (out[:, :t] == vals[:, t:t + 1]).any(axis=1)
                out[:, t] = np.where(dup, n - size + t, vals[:, t])
            done += good
            pos += good * size
        if done < n_draws and good < left:  # this draw meets a rejection: word by word
            restarts += 1
            if restarts > 4096:
                break
            picked = []
            for t in range(size):
                while True:
                    if pos >= u.size:
                        extra *= 4
                        u = stream(pos + (n_draws - done) * size + extra)
                    mm = int(u[pos]) * int(bounds[t])
                    pos += 1
                    if (mm & 0xFFFFFFFF) >= int(thresh[t]):
                        break
                v = mm >> 32
                picked.append(n - size + t if v in picked else v)
            draws[done] = picked
            done += 1
    # leave the generator where the calls would have left it: `pos` words taken, the first of them the buffered one if any
    from_raw = pos - (1 if st["has_uint32"] and pos > 0 else 0)
    clone = np.random.PCG64()
    clone.state = st
    if from_raw > 0:
        tail = clone.random_raw((from_raw + 1) // 2)
        new = clone.state
        new["has_uint32"] = int(from_raw % 2 == 1)
        new["uinteger"] = int(tail[-1] >> np.uint64(32))  # (the half-word next_uint32 buffered last: pending, or stale and unused)
    else:
        new = dict(st)
        if pos > 0:
            new["has_uint32"] = 0
    bg.state = new
    if done < n_draws:
        draws[done:] = _draws_by_loop(gen, n, size, n_draws - done)
    return draws


_REPLICA_OK: Optional[bool] = None


def _replica_matches_this_numpy() -> bool:
    """Generator.choice against the replica on scratch generators: small and large populations, a buffered half-word, populations
    where repeats and rejections happen; values AND the state left behind."""
    global _REPLICA_OK
    if _REPLICA_OK is None:
        ok = True
        try:
            for seed, n, size, k, pre in ((1, 500, 4, 300, 0), (2, 1_000_000, 4, 300, 1), (3, 9, 4, 200, 0), (4, 20_000_000, 3, 900, 1),
                                          (5, 70, 3, 100, 0)):
                a, b = np.random.default_rng(seed), np.random.default_rng(seed)
                for g_ in (a, b):
                    for _ in range(pre):
                        g_.integers(0, 10, dtype=np.uint32)  # (leaves a buffered half-word)
                mine = _draws_from_raw_stream(a, n, size, k)
                ref = _draws_by_loop(b, n, size, k)
                ok &= mine is not None and np.array_equal(mine, ref) and a.bit_generator.state == b.bit_generator.state
        except Exception:  # noqa: BLE001 -- any surprise means: keep the loop
            ok = False
        _REPLICA_OK = bool(ok)
    return _REPLICA_OK


def draw_stream(gen: np.random.Generator, n: int, size: int, n_draws: int) -> np.ndarray:
    """(n_draws, size) int64: what n_draws successive gen.choice(n, size, replace=False, shuffle=False) return."""
    if n_draws > 32 and _replica_matches_this_numpy():
        draws = _draws_from_raw_stream(gen, n, size, n_draws)
        if draws is not None:
            return draws
    return _draws_by_loop(gen, n, size, n_draws)


def ransac_on_matches(
    scan_descriptors_indices: npt.NDArray[np.integer],
    ref_descriptors_indices: npt.NDArray[np.integer],
    scan_keypoints: npt.NDArray[np.float64],
    ref_keypoints: npt.NDArray[np.float64],
    n_draws: int = 10000,
    draw_size: int = 4,
    distance_threshold: float = 1,
    verbose: bool = False,
    disable_progress_bar: bool = False,
    *,
    engine: Optional[Engine] = None,
) -> tuple[float, RigidTransform]:
    """Returns (inlier ratio of the best draw, its RigidTransform with a re-normalised rotation).

    The best draw is the FIRST one reaching the maximal inlier count (strict `>` at ransac.py:68).
    """
    eng = engine or default_engine()
    n_matches = scan_descriptors_indices.shape[0]
    scan_idx = np.asarray(scan_descriptors_indices)
    ref_idx = np.asarray(ref_descriptors_indices)
    scan_kp, ref_kp = np.asarray(scan_keypoints), np.asarray(ref_keypoints)

    if n_draws <= 0:  # (no draw, no transform: what ransac.py:80 runs into)
        raise AttributeError("'NoneType' object has no attribute 'normalize_rotation'")
    draws = draw_stream(rng, n_matches, draw_size, n_draws)
    # The matched points of all 10^6 matches are only ever read by K9: keypoints and index vectors go up as they are
    # (2 x 24 MB + 2 x 8 MB at the link's rate) and the gather keypoints[indices] runs on the device -- on the host it is two
    # scattered passes over 10^6 rows, 40-300 ms.  The host needs the 4 matched pairs of each DRAW only.
    matched = _matched_points_on_device(eng, scan_kp, scan_idx, ref_kp, ref_idx)
    try:
        scan_d = np.ascontiguousarray(scan_kp[scan_idx[draws]], dtype=np.float64)
        ref_d = np.ascontiguousarray(ref_kp[ref_idx[draws]], dtype=np.float64)
        records = np.empty((n_draws, 12), dtype=np.float64)
        inliers = np.empty(n_draws, dtype=np.int64)
        # chunks of draws: the Kabsch fits of chunk c + 1 (host: LAPACK, ~2.5 us per draw) run while K9 scores chunk c
        chunk = max(1024, -(-n_draws // 4))
        pending = []
        for c0 in range(0, n_draws, chunk):
            c1 = min(c0 + chunk, n_draws)
            _solve_chunk(scan_d[c0:c1], ref_d[c0:c1], records[c0:c1])
            pending.append(matched.score_async(records[c0:c1], distance_threshold))
        for (c0, job) in zip(range(0, n_draws, chunk), pending):
            inliers[c0:c0 + job.n] = job.result()
    finally:
        matched.free()
    best = int(np.argmax(inliers))  # first maximum == reference's strict-greater update rule
    if verbose:
        logging.info(f"Best draw {best}: {int(inliers[best])} inliers out of {n_matches}")
    best_transform = RigidTransform(records[best, :9].reshape(3, 3).copy(), records[best, 9:].copy())
    best_transform.normalize_rotation()
    return inliers[best] / n_matches, best_transform


def _solve_chunk(scan_d: np.ndarray, ref_d: np.ndarray, records: np.ndarray) -> None:
    """One Kabsch fit per draw into records[:, :12] -- solved as a stack through the same BLAS / LAPACK routines as the
    per-draw solver and held to it bit for bit on a sample of the chunk (its first draws, an evenly spread few, and reflected
    draws: det < 0 is rare and the likeliest to differ); on any difference the whole chunk takes the per-draw loop."""
    n = scan_d.shape[0]
    rot, tr, reflected = solver_point_to_point_batched(scan_d, ref_d, return_reflected=True)
    records[:, :9] = rot.reshape(n, 9)
    records[:, 9:] = tr
    check = set(range(min(n, 4))) | set(np.linspace(0, n - 1, min(n, 12)).astype(int).tolist()) | set(reflected[:8].tolist())
    for d in sorted(check):
        if not np.array_equal(records[d], solver_point_to_point(scan_d[d], ref_d[d]).as_row12()):
            logging.warning("stacked Kabsch differs from the per-draw solver on this NumPy build: using the per-draw loop")
            for e in range(n):
                records[e] = solver_point_to_point(scan_d[e], ref_d[e]).as_row12()
            break


class _ScoreJob:
    def __init__(self, eng, rt_dev, out_dev, n):
        self.eng, self.rt_dev, self.out_dev, self.n = eng, rt_dev, out_dev, n

    def result(self) -> np.ndarray:
        try:
            return self.out_dev.to_host()[: self.n]
        finally:
            self.rt_dev.free()
            self.out_dev.free()


class _matched_points_on_device:
    """scan_keypoints[scan_idx], ref_keypoints[ref_idx] resident in HBM (gathered there), scored against chunks of transforms."""

    def __init__(self, eng: Engine, scan_kp, scan_idx, ref_kp, ref_idx):
        self.eng, self.m = eng, int(scan_idx.shape[0])
        self._held = []
        try:
            self.a = self._gather(scan_kp, scan_idx)
            self.b = self._gather(ref_kp, ref_idx)
        except Exception:
            self.free()
            raise

    def _gather(self, kp, idx):
        kp = np.ascontiguousarray(kp, dtype=np.float64)
        if kp.ndim != 2 or kp.shape[1] != 3:
            raise ValueError(f"expected (n, 3) keypoints, got shape {kp.shape}")
        sel = np.ascontiguousarray(idx, dtype=np.int64)
        n = kp.shape[0]
        if sel.size:
            lo, hi = int(sel.min()), int(sel.max())
            if lo < -n or hi >= n:  # (what keypoints[indices] raises)
                raise IndexError(f"index {lo if lo < -n else hi} is out of bounds for axis 0 with size {n}")
            if lo < 0:
                sel = np.where(sel < 0, sel + n, sel)
        rows = self.eng.empty((max(n, 1), 3))
        self._held.append(rows)
        dsel = self.eng.empty((max(self.m, 1),), np.int64)
        self._held.append(dsel)
        out = self.eng.empty((max(self.m, 1), 3))
        self._held.append(out)
        if n:
            rows.from_host(kp)
        if self.m:
            dsel.from_host(sel)
            self.eng.rows_gather_device(rows, dsel, out)
        return out

    def score_async(self, records: np.ndarray, thr: float) -> _ScoreJob:
        n = records.shape[0]
        rt = self.eng.empty((max(n, 1), 12))
        out = self.eng.empty((max(n, 1),), np.int64)
        if n:
            rt.from_host(np.ascontiguousarray(records))
            self.eng.ransac_score_device(self.a, self.b, self.m, rt, n, thr, out)
        return _ScoreJob(self.eng, rt, out, n)

    def free(self) -> None:
        for h in self._held:
            h.free()
        self._held = []
