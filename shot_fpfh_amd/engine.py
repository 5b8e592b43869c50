"""Object layer over the C ABI: Engine (one GPU), Cloud (resident points + grid), Neighbors (CSR lists).

Everything that computes runs in HIP kernels inside libshotfpfh.so; this module only moves NumPy
arrays across the boundary and keeps device handles alive.  NumPy inputs are cast to C-contiguous
float64 at the boundary (the reference computes in float64 throughout).
"""
from __future__ import annotations

import collections
import ctypes as C
import os
import threading
import weakref
from typing import Optional

import numpy as np

from . import _ffi
from ._ffi import SF_HOST, SF_IN_DEVICE, SF_OUT_DEVICE, ShotFpfhError

__all__ = ["Engine", "Cloud", "Neighbors", "Spfh", "DeviceArray", "default_engine", "ShotFpfhError", "fpfh_edges"]


def _f64(a, cols: Optional[int] = None) -> np.ndarray:
    arr = np.ascontiguousarray(a, dtype=np.float64)
    if cols is not None and (arr.ndim != 2 or arr.shape[1] != cols):
        raise ValueError(f"expected an (N, {cols}) array, got shape {arr.shape}")
    return arr


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def fpfh_edges(n_bins: int) -> np.ndarray:
    """The 3 x (n_bins+1) bin edges np.histogramdd derives from `bins=n_bins` and the ranges of
    fpfh.py:82-87.  Computed with np.linspace on the host so the device compares against the very
    same float64 edge values (they are not ideal fractions)."""
    return np.ascontiguousarray(
        np.stack(
            [
                np.linspace(-1, 1, n_bins + 1),
                np.linspace(-1, 1, n_bins + 1),
                np.linspace(-np.pi / 2, np.pi / 2, n_bins + 1),
            ]
        ),
        dtype=np.float64,
    )


class DeviceArray:
    """A typed, shaped view of device memory owned by an Engine (results kept resident in HBM)."""

    def __init__(self, engine: "Engine", shape, dtype):
        self.engine = engine
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        self.ptr = _ffi.check_handle(engine.lib.sf_dev_alloc(engine.h, max(self.nbytes, 8)), "sf_dev_alloc")

    def to_host(self) -> np.ndarray:
        out = np.empty(self.shape, dtype=self.dtype)
        _ffi.check(self.engine.lib.sf_d2h(self.engine.h, _ptr(out), self.ptr, self.nbytes), "sf_d2h")
        return out

    def rows_to_host(self, first: int, count: int) -> np.ndarray:
        """Rows [first, first + count) of a 2-D (or longer) array, without copying the rest."""
        if not (0 <= first and count >= 0 and first + count <= self.shape[0]):
            raise ValueError("row range outside the array")
        row_bytes = self.nbytes // max(self.shape[0], 1)
        out = np.empty((count,) + self.shape[1:], dtype=self.dtype)
        if count:
            _ffi.check(self.engine.lib.sf_d2h(self.engine.h, _ptr(out), self.offset_ptr(first * row_bytes), count * row_bytes), "sf_d2h")
        return out

    def from_host(self, a: np.ndarray) -> "DeviceArray":
        a = np.ascontiguousarray(a, dtype=self.dtype)
        if a.nbytes != self.nbytes:
            raise ValueError("size mismatch")
        _ffi.check(self.engine.lib.sf_h2d(self.engine.h, self.ptr, _ptr(a), self.nbytes), "sf_h2d")
        return self

    def copy_from_device(self, src: "DeviceArray", dst_byte_offset: int = 0, nbytes: Optional[int] = None) -> "DeviceArray":
        """Stream-ordered device-to-device copy of `src` into this array at a byte offset."""
        nbytes = src.nbytes if nbytes is None else nbytes
        if dst_byte_offset + nbytes > self.nbytes:
            raise ValueError("copy exceeds the destination")
        _ffi.check(self.engine.lib.sf_d2d(self.engine.h, self.offset_ptr(dst_byte_offset), src.ptr, nbytes), "sf_d2d")
        return self

    def offset_ptr(self, byte_offset: int):
        return C.c_void_p(self.ptr + byte_offset)

    def free(self) -> None:
        if self.ptr:
            self.engine.lib.sf_dev_free(self.engine.h, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class _PinnedPool:
    """Page-locked host blocks (sf_host_alloc) behind the NumPy arrays `Engine.host_empty` hands out.

    A block comes back through a weakref finalizer, which the garbage collector may run at ANY allocation point --
    including inside `array()` of the same thread.  The finalizer therefore takes no lock at all: it appends to a deque
    (atomic in CPython), and `array()` / `trim()` move the returned blocks into the free list under the lock."""

    THRESHOLD = int(float(os.environ.get("SF_PINNED_THRESHOLD_MB", "32")) * (1 << 20))  # smaller results use ordinary NumPy memory
    KEEP_BYTES = 4 << 30      # cached (unused) blocks beyond this are unpinned; Engine.trim_host_cache() drops them all

    def __init__(self, lib, ctx):
        self.lib, self.ctx = lib, ctx
        self.free: list[tuple[int, int]] = []  # (nbytes, address)
        self.cached = 0
        self.lock = threading.Lock()
        self.returned: collections.deque = collections.deque()  # blocks handed back by finalizers, not yet sorted in

    def _absorb(self) -> list:
        """(under the lock) returned blocks -> free list; those beyond KEEP_BYTES are given to the caller to unpin."""
        drop = []
        while True:
            try:
                blk = self.returned.popleft()
            except IndexError:
                break
            if self.ctx is not None and self.cached + blk[0] <= self.KEEP_BYTES:
                self.free.append(blk)
                self.cached += blk[0]
            else:
                drop.append(blk)
        return drop

    def array(self, shape, dtype, count, nbytes) -> np.ndarray:
        with self.lock:
            drop = self._absorb()
            blk = None
            for b in self.free:  # (a plain loop: no allocation-heavy comprehension while the lock is held)
                if nbytes <= b[0] <= nbytes + nbytes // 2 + (1 << 20) and (blk is None or b < blk):
                    blk = b
            if blk is not None:
                self.free.remove(blk)
                self.cached -= blk[0]
        for _, addr in drop:
            self.lib.sf_host_free(None, addr)
        if blk is None:
            size = (nbytes + 4095) & ~4095
            addr = self.lib.sf_host_alloc(self.ctx, size)
            if not addr:
                return np.empty(shape, dtype=dtype)  # pinning refused (ulimit, memory pressure): plain memory works too
            blk = (size, addr)
        buf = (C.c_char * blk[0]).from_address(blk[1])
        weakref.finalize(buf, self._give_back, blk)  # `buf` lives exactly as long as the array and its views
        return np.frombuffer(buf, dtype=dtype, count=count).reshape(shape)

    def _give_back(self, blk) -> None:
        # no lock here (see the class note).  After drain() nobody sorts the deque in any more: unpin at once.
        if self.ctx is None:
            self.lib.sf_host_free(None, blk[1])
        else:
            self.returned.append(blk)

    def trim(self, keep_bytes: int = 0) -> int:
        """Unpin cached blocks until at most `keep_bytes` stay; returns the bytes released."""
        with self.lock:
            drop = self._absorb()
            self.free.sort()
            while self.free and self.cached > keep_bytes:
                blk = self.free.pop()
                self.cached -= blk[0]
                drop.append(blk)
        for _, addr in drop:
            self.lib.sf_host_free(None, addr)
        return sum(b[0] for b in drop)

    def drain(self) -> None:
        with self.lock:
            blocks, self.free, self.cached, self.ctx = self.free + self._absorb(), [], 0, None
        for _, addr in blocks:
            self.lib.sf_host_free(None, addr)


class StepGraph:
    """An executable HIP graph of a captured step (Engine.capture): launch() replays it on the engine's main stream."""

    def __init__(self, engine: "Engine", handle):
        self.engine, self.h = engine, handle

    def launch(self) -> None:
        _ffi.check(self.engine.lib.sf_graph_launch(self.engine.h, self.h), "sf_graph_launch")

    def free(self) -> None:
        if getattr(self, "h", None) and self.engine.h:
            self.engine.lib.sf_graph_free(self.engine.h, self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class MatchStream:
    """Handle of a streamed K8 (Engine.match_stream)."""

    def __init__(self, engine: "Engine", handle):
        self.engine, self.h = engine, handle

    def feed(self, row_begin: int, row_end: int) -> None:
        _ffi.check(self.engine.lib.sf_match_stream_feed(self.engine.h, self.h, int(row_begin), int(row_end)), "sf_match_stream_feed")

    def end(self, idx: DeviceArray, dist: Optional[DeviceArray] = None) -> None:
        h, self.h = self.h, None  # (released by the call, whatever it returns)
        _ffi.check(self.engine.lib.sf_match_stream_end(self.engine.h, h, idx.ptr, None if dist is None else dist.ptr), "sf_match_stream_end")

    def abort(self) -> None:
        if self.h is not None and self.engine.h:
            self.engine.lib.sf_match_stream_abort(self.engine.h, self.h)
        self.h = None

    def __del__(self):
        try:
            self.abort()
        except Exception:
            pass


class Engine:
    """One GPU: a libshotfpfh context with its own HIP stream.  Fails loudly when the native library
    or the GPU is missing -- there is no CPU path."""

    def __init__(self, device: Optional[int] = None):
        self.lib = _ffi.load()
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0"))
            n = self.lib.sf_device_count()
            if n > 0:
                device %= n
        self.device = device
        self.h = _ffi.check_handle(self.lib.sf_create(device), f"sf_create({device})")
        self.pid = os.getpid()
        self.nranks, self.rank = 1, 0
        self._pinned = _PinnedPool(self.lib, self.h)

    # ---- lifetime -----------------------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "h", None) and os.getpid() == self.pid:
            self._pinned.drain()
            self.lib.sf_destroy(self.h)
        self.h = None

    def host_empty(self, shape, dtype=np.float64) -> np.ndarray:
        """A fresh, writable, C-contiguous NumPy array for a result coming back from the GPU.  Large ones (the
        (M, 352) / (M, n_bins^3) descriptor matrices) live in page-locked memory, so the device-to-host copy that
        fills them is ONE DMA at PCIe speed; the block goes back to a per-engine pool when the array (and every view
        of it) has been garbage-collected, and is pinned again only if a later result does not fit a cached one."""
        dtype = np.dtype(dtype)
        shape = tuple(int(v) for v in np.atleast_1d(shape)) if not isinstance(shape, tuple) else tuple(int(v) for v in shape)
        count = int(np.prod(shape, dtype=np.int64))
        nbytes = count * dtype.itemsize
        if nbytes < _PinnedPool.THRESHOLD:
            return np.empty(shape, dtype=dtype)
        return self._pinned.array(shape, dtype, count, nbytes)

    def trim_host_cache(self, keep_bytes: int = 0) -> int:
        """Release the page-locked blocks the engine keeps for re-use by later results (at most 4 GiB are kept; results of
        32 MiB and more live in such blocks for as long as the returned array or any view of it is alive)."""
        return self._pinned.trim(keep_bytes)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self) -> None:
        _ffi.check(self.lib.sf_sync(self.h), "sf_sync")

    def fork(self) -> None:
        """Following calls go to the side stream (ordered after the work issued so far)."""
        _ffi.check(self.lib.sf_fork(self.h), "sf_fork")

    def switch(self, side: int) -> None:
        _ffi.check(self.lib.sf_switch(self.h, int(side)), "sf_switch")

    def mark(self) -> None:
        """Remember the point reached on the current stream (see wait_mark)."""
        _ffi.check(self.lib.sf_mark(self.h), "sf_mark")

    def wait_mark(self) -> None:
        """The current stream waits for the marked point -- not for what was queued on the marked stream after it."""
        _ffi.check(self.lib.sf_wait_mark(self.h), "sf_wait_mark")

    def join(self) -> None:
        """Back on the main stream, ordered after the side stream."""
        _ffi.check(self.lib.sf_join(self.h), "sf_join")

    # ---- a repeated step as one launch (HIP graph; include/shotfpfh.h "a repeated step as ONE launch") -----------------------
    def capture(self, fn) -> "StepGraph":
        """Run fn() with the context's streams in capture mode: nothing executes, every device operation fn issues becomes a
        node of the returned graph.  fn must not synchronise with the device (a repeated DescriptorJob.step() does not; a first
        one does) -- the capture is then abandoned and ShotFpfhError raised."""
        _ffi.check(self.lib.sf_graph_begin(self.h), "sf_graph_begin")
        try:
            fn()
        except BaseException:
            g = self.lib.sf_graph_end(self.h)  # (leave capture mode whatever happened)
            if g:
                self.lib.sf_graph_free(self.h, g)
            raise
        return StepGraph(self, _ffi.check_handle(self.lib.sf_graph_end(self.h), "sf_graph_end"))

    def empty(self, shape, dtype=np.float64) -> DeviceArray:
        return DeviceArray(self, shape, dtype)

    # ---- cloud / search ---------------------------------------------------------------------------
    def cloud(self, points, normals=None) -> "Cloud":
        return Cloud(self, points, normals)

    def spfh(self, cloud: "Cloud", n_bins: int, max_count: int, radius: Optional[float] = None) -> "Spfh":
        return Spfh(cloud, n_bins, max_count, radius)

    def azimuth_idx(self, x, y) -> np.ndarray:
        """get_azimuth_idx (shot.py:51-70) evaluated by the device function K5 bins with."""
        x, y = np.ascontiguousarray(x, dtype=np.float64), np.ascontiguousarray(y, dtype=np.float64)
        if x.shape != y.shape:
            raise ValueError("x and y must have the same shape")
        out = np.zeros(x.shape, dtype=np.int64)
        _ffi.check(self.lib.sf_azimuth_idx(self.h, _ptr(x), _ptr(y), x.size, _ptr(out), SF_HOST), "sf_azimuth_idx")
        return out

    # ---- matching (K8) / RANSAC scoring (K9) ------------------------------------------------------
    def match_argmin(self, a, b, want_dist=True, want_col=False):
        """Row arg-min of cdist(a, b) (first minimum), winners' distances, optional column arg-min."""
        a, b = _f64(a), _f64(b)
        if a.ndim != 2 or b.ndim != 2 or a.shape[1] != b.shape[1]:
            raise ValueError("descriptor matrices must be 2-D with equal width")
        m1, m2 = a.shape[0], b.shape[0]
        idx = np.zeros(m1, dtype=np.int64)
        dist = np.zeros(m1, dtype=np.float64) if want_dist else None
        col = np.zeros(m2, dtype=np.int64) if want_col else None
        if m1 and not m2:
            raise ValueError("attempt to get argmin of an empty sequence")
        if m1:
            _ffi.check(
                self.lib.sf_match_argmin(self.h, _ptr(a), m1, _ptr(b), m2, a.shape[1], _ptr(idx), _ptr(dist), _ptr(col), SF_HOST),
                "sf_match_argmin",
            )
        return idx, dist, col

    def match_argmin_device(self, a: DeviceArray, b: DeviceArray, idx: DeviceArray, dist: Optional[DeviceArray] = None,
                            col: Optional[DeviceArray] = None, rows: Optional[int] = None, row_offset: int = 0) -> None:
        """Resident variant: a (rows starting at row_offset) vs all of b, outputs stay on the device."""
        d = a.shape[1]
        m1 = a.shape[0] - row_offset if rows is None else rows
        _ffi.check(
            self.lib.sf_match_argmin(self.h, a.offset_ptr(row_offset * d * 8), m1, b.ptr, b.shape[0], d, idx.ptr,
                                     None if dist is None else dist.ptr, None if col is None else col.ptr,
                                     SF_IN_DEVICE | SF_OUT_DEVICE),
            "sf_match_argmin",
        )

    def match_argmin_multiscale(self, a, b, max_val: float = 1000.0):
        """a: (S, M1, D), b: (S, M2, D).  Arg-min over j of min over scales of the per-scale distance, with
        max_val wherever either descriptor is all-zero at that scale (matching.py:77-136)."""
        a, b = _f64(a), _f64(b)
        if a.ndim != 3 or b.ndim != 3 or a.shape[0] != b.shape[0] or a.shape[2] != b.shape[2]:
            raise ValueError("multi-scale descriptor stacks must be (n_scales, n_points, length) with equal scales/length")
        a_ok = np.ascontiguousarray(np.any(a, axis=2), dtype=np.uint8)
        b_ok = np.ascontiguousarray(np.any(b, axis=2), dtype=np.uint8)
        m1, m2 = a.shape[1], b.shape[1]
        idx, dist = np.zeros(m1, dtype=np.int64), np.zeros(m1, dtype=np.float64)
        if m1 and not m2:
            raise ValueError("attempt to get argmin of an empty sequence")
        if m1:
            _ffi.check(
                self.lib.sf_match_argmin_multiscale(self.h, _ptr(a), _ptr(b), a.shape[0], m1, m2, a.shape[2], _ptr(a_ok),
                                                    _ptr(b_ok), float(max_val), _ptr(idx), _ptr(dist), SF_HOST),
                "sf_match_argmin_multiscale",
            )
        return idx, dist

    def rows_nonzero_device(self, rows: DeviceArray, out: Optional[DeviceArray] = None, n_rows: Optional[int] = None,
                            first_row: int = 0) -> DeviceArray:
        """uint8 mask on the device: 1 where a descriptor row has a non-zero entry (rows first_row .. first_row + n_rows - 1)."""
        m = rows.shape[0] - first_row if n_rows is None else n_rows
        out = out if out is not None else self.empty((rows.shape[0],), np.uint8)
        _ffi.check(self.lib.sf_rows_nonzero(self.h, rows.offset_ptr(first_row * rows.shape[1] * 8), m, rows.shape[1],
                                            out.offset_ptr(first_row)), "sf_rows_nonzero")
        return out

    def rows_gather_device(self, rows: DeviceArray, sel: DeviceArray, out: DeviceArray) -> DeviceArray:
        """out[i] = rows[sel[i]] (a zero row where sel[i] < 0), all resident: a keypoint subset of a descriptor
        matrix, padded to the equal per-rank block an all-gather needs."""
        m = sel.shape[0]
        if out.shape[0] < m or out.shape[1] != rows.shape[1] or sel.dtype != np.int64:
            raise ValueError("rows_gather_device: int64 selection and an (>= len(sel), d) output expected")
        _ffi.check(self.lib.sf_rows_gather(self.h, rows.ptr, rows.shape[0], sel.ptr, m, rows.shape[1], out.ptr), "sf_rows_gather")
        return out

    def match_masked_device(self, a: DeviceArray, a_ok: DeviceArray, b: DeviceArray, b_ok: DeviceArray, idx: DeviceArray,
                            dist: Optional[DeviceArray] = None, a_rows: Optional[int] = None, b_rows: Optional[int] = None) -> None:
        """Resident arg-min of a's rows over b's rows, skipping rows whose mask is 0 (they get distance +inf)."""
        m1 = a.shape[0] if a_rows is None else a_rows
        m2 = b.shape[0] if b_rows is None else b_rows
        _ffi.check(
            self.lib.sf_match_argmin_multiscale(self.h, a.ptr, b.ptr, 1, m1, m2, a.shape[1], a_ok.ptr, b_ok.ptr,
                                                float("inf"), idx.ptr, None if dist is None else dist.ptr,
                                                SF_IN_DEVICE | SF_OUT_DEVICE),
            "sf_match_argmin_multiscale",
        )

    def col_candidates_device(self, local_dist: DeviceArray, global_dist: DeviceArray, local_idx: DeviceArray, row_offset: int,
                              out: DeviceArray, m: Optional[int] = None) -> DeviceArray:
        """out[j] = row_offset + local_idx[j] where local_dist[j] equals the (all-reduced) global_dist[j], else 2^64 - 1."""
        m = local_dist.shape[0] if m is None else m
        if out.dtype != np.uint64:
            raise ValueError("col_candidates_device writes uint64 candidates")
        _ffi.check(self.lib.sf_match_col_candidates(self.h, local_dist.ptr, global_dist.ptr, local_idx.ptr, int(row_offset), int(m),
                                                    out.ptr), "sf_match_col_candidates")
        return out

    def ransac_score(self, a, b, rt, thr: float) -> np.ndarray:
        a, b = _f64(a, 3), _f64(b, 3)
        rt = _f64(rt).reshape(-1, 12)
        out = np.zeros(rt.shape[0], dtype=np.int64)
        _ffi.check(
            self.lib.sf_ransac_score(self.h, _ptr(a), _ptr(b), a.shape[0], _ptr(rt), rt.shape[0], float(thr), _ptr(out), SF_HOST),
            "sf_ransac_score",
        )
        return out

    def ransac_score_device(self, a: DeviceArray, b: DeviceArray, m: int, rt: DeviceArray, n_draws: int, thr: float,
                            out: DeviceArray) -> DeviceArray:
        """K9 on resident operands, asynchronous: a, b (m, 3) matched points, rt (n_draws, 12), out (n_draws,) int64."""
        if out.dtype != np.int64 or out.shape[0] < n_draws or rt.shape[0] < n_draws or a.shape[0] < m or b.shape[0] < m:
            raise ValueError("ransac_score_device: operands smaller than the counts given")
        _ffi.check(self.lib.sf_ransac_score(self.h, a.ptr, b.ptr, int(m), rt.ptr, int(n_draws), float(thr), out.ptr,
                                            SF_IN_DEVICE | SF_OUT_DEVICE), "sf_ransac_score")
        return out

    # ---- multi-GPU (RCCL) -------------------------------------------------------------------------
    def comm_unique_id(self) -> bytes:
        buf = C.create_string_buffer(128)
        _ffi.check(self.lib.sf_comm_unique_id(buf), "sf_comm_unique_id")
        return buf.raw

    def comm_init(self, unique_id: bytes, nranks: int, rank: int) -> None:
        buf = C.create_string_buffer(unique_id, 128)
        _ffi.check(self.lib.sf_comm_init(self.h, buf, nranks, rank), "sf_comm_init")
        self.nranks, self.rank = nranks, rank

    def allgather(self, buf: DeviceArray, bytes_per_rank: int) -> None:
        """In-place all-gather: rank r's block lives at byte offset r * bytes_per_rank of `buf`."""
        if bytes_per_rank * self.nranks > buf.nbytes:
            raise ValueError("buffer too small for the all-gather")
        _ffi.check(
            self.lib.sf_comm_allgather(self.h, buf.offset_ptr(self.rank * bytes_per_rank), buf.ptr, bytes_per_rank),
            "sf_comm_allgather",
        )

    def rows_abs_max(self, rows: DeviceArray, n_rows: Optional[int] = None) -> float:
        """Largest |entry| of the first n_rows resident rows (the scale of an int8 image: sf_rows_abs_max)."""
        m = rows.shape[0] if n_rows is None else n_rows
        out = C.c_double(0.0)
        _ffi.check(self.lib.sf_rows_abs_max(self.h, rows.ptr, int(m), int(rows.shape[1]), C.byref(out)), "sf_rows_abs_max")
        return float(out.value)

    def match_stream(self, a: DeviceArray, a_ok: DeviceArray, m1: int, b: DeviceArray, b_ok: DeviceArray, m2: int, b_entry_max: float,
                     max_ranges: int) -> "MatchStream":
        """K8 while the reference rows are still arriving (sf_match_stream_*): feed(begin, end) as ranges of `b` land, end(idx, dist)."""
        h = _ffi.check_handle(self.lib.sf_match_stream_begin(self.h, a.ptr, a_ok.ptr, int(m1), b.ptr, b_ok.ptr, int(m2), int(a.shape[1]),
                                                             float(b_entry_max), int(max_ranges)), "sf_match_stream_begin")
        return MatchStream(self, h)

    def exchange(self, ops) -> None:
        """Grouped point-to-point exchange (ncclSend / ncclRecv in ONE group).  ops: iterable of
        (peer, send DeviceArray | None, send_byte_offset, send_bytes, recv DeviceArray | None, recv_byte_offset, recv_bytes)."""
        ops = list(ops)
        n = len(ops)
        peers = (C.c_int * max(n, 1))(*[int(o[0]) for o in ops])
        sp = (C.c_void_p * max(n, 1))(*[None if o[1] is None else o[1].ptr + int(o[2]) for o in ops])
        sb = (C.c_size_t * max(n, 1))(*[int(o[3]) for o in ops])
        rp = (C.c_void_p * max(n, 1))(*[None if o[4] is None else o[4].ptr + int(o[5]) for o in ops])
        rb = (C.c_size_t * max(n, 1))(*[int(o[6]) for o in ops])
        _ffi.check(self.lib.sf_comm_exchange(self.h, n, peers, sp, sb, rp, rb), "sf_comm_exchange")

    def allreduce_min_u64(self, buf: DeviceArray, n: Optional[int] = None) -> None:
        """In-place element-wise minimum over the ranks of a uint64 device array (packed (distance, index) keys)."""
        if buf.dtype != np.uint64:
            raise ValueError("allreduce_min_u64 needs a uint64 array")
        count = int(np.prod(buf.shape)) if n is None else int(n)
        _ffi.check(self.lib.sf_comm_allreduce_min_u64(self.h, buf.ptr, buf.ptr, count), "sf_comm_allreduce_min_u64")

    def collective_stats(self, on: bool) -> None:
        """While on, radius searches fold their longest-list statistic over all ranks (Neighbors.max_count_all): every
        rank must then run the same searches in the same order."""
        _ffi.check(self.lib.sf_comm_collective_stats(self.h, int(bool(on))), "sf_comm_collective_stats")

    # ---- profiling ----------------------------------------------------------------------------------
    def profile(self, on: bool) -> None:
        _ffi.check(self.lib.sf_profile_enable(self.h, int(on)), "sf_profile_enable")

    def profile_only(self, name: Optional[str]) -> None:
        """Time launches of this kernel name only (None = all)."""
        _ffi.check(self.lib.sf_profile_only(self.h, None if name is None else name.encode()), "sf_profile_only")

    def profile_reset(self) -> None:
        _ffi.check(self.lib.sf_profile_reset(self.h), "sf_profile_reset")

    def profile_report(self) -> dict:
        """{kernel name: (launches, total milliseconds measured with HIP events on the engine's stream)}"""
        need = self.lib.sf_profile_report(self.h, None, 0)
        buf = C.create_string_buffer(int(need) + 16)
        self.lib.sf_profile_report(self.h, buf, len(buf))
        rep = {}
        for line in buf.value.decode().splitlines():
            name, launches, ms = line.split()
            rep[name] = (int(launches), float(ms))
        return rep


class Cloud:
    """Point cloud resident in HBM with its uniform grid -- the stand-in for sklearn's KDTree(X)."""

    def __init__(self, engine: Engine, points, normals=None, subset=None):
        """`subset` (int64 indices): the cloud is points[subset] (normals[subset]) in that order -- gathered on the device
        from one streaming upload of the whole arrays (a scattered `points[subset]` of 10^6 rows on the host is ~40 ms per
        array; the subsampled support of the reference's default SHOT configuration is exactly that)."""
        self.engine = engine
        pts = _f64(points, 3)
        nrm = None if normals is None else _f64(normals, 3)
        if nrm is not None and nrm.shape != pts.shape:
            raise ValueError("normals must have the same shape as points")
        if subset is not None:
            keep = np.ascontiguousarray(subset, dtype=np.int64)
            if keep.ndim != 1 or (keep.size and (keep.min() < 0 or keep.max() >= pts.shape[0])):
                raise ValueError("subset must be a 1-D array of row indices")
            self.n = int(keep.shape[0])
            tmp = []
            try:
                sel = engine.empty((max(self.n, 1),), np.int64)
                tmp.append(sel)
                if self.n:
                    sel.from_host(keep)
                dev = []
                for a in (pts, nrm):
                    if a is None:
                        dev.append(None)
                        continue
                    full = engine.empty(a.shape).from_host(a)
                    tmp.append(full)
                    out = engine.empty((max(self.n, 1), 3))
                    tmp.append(out)
                    if self.n:
                        engine.rows_gather_device(full, sel, out)
                    dev.append(out)
                self.h = _ffi.check_handle(
                    engine.lib.sf_cloud_upload(engine.h, dev[0].ptr, None if dev[1] is None else dev[1].ptr, self.n, SF_IN_DEVICE),
                    "sf_cloud_upload",
                )
            finally:
                for t in tmp:
                    t.free()
            return
        self.n = pts.shape[0]
        self.h = _ffi.check_handle(
            engine.lib.sf_cloud_upload(engine.h, _ptr(pts), _ptr(nrm), self.n, SF_HOST), "sf_cloud_upload"
        )

    def set_normals(self, normals) -> None:
        nrm = _f64(normals, 3)
        if nrm.shape[0] != self.n:
            raise ValueError("normals must have one row per point")
        _ffi.check(self.engine.lib.sf_cloud_set_normals(self.engine.h, self.h, _ptr(nrm), SF_HOST), "sf_cloud_set_normals")

    def build_grid(self, cell: float, block: Optional[tuple[int, int]] = None, reach: int = 2) -> tuple[int, int]:
        """K1.  With `block=(begin, end)` only the z-layers that block's queries can reach within `reach` cells are
        sorted and gathered (global position numbering kept): what one rank of a sharded job needs.  Returns the
        populated range of cell-sorted positions."""
        if block is None:
            _ffi.check(self.engine.lib.sf_cloud_build_grid(self.engine.h, self.h, float(cell)), "sf_cloud_build_grid")
            return 0, self.n
        pb, pe = C.c_int64(0), C.c_int64(0)
        _ffi.check(
            self.engine.lib.sf_cloud_build_grid_block(self.engine.h, self.h, float(cell), int(block[0]), int(block[1]),
                                                      int(reach), C.byref(pb), C.byref(pe)),
            "sf_cloud_build_grid_block",
        )
        return pb.value, pe.value

    def perm(self) -> np.ndarray:
        """cell-sorted position -> original point index (valid after a grid build / search)."""
        out = np.zeros(self.n, dtype=np.int32)
        _ffi.check(self.engine.lib.sf_cloud_perm(self.engine.h, self.h, _ptr(out)), "sf_cloud_perm")
        return out

    def layer_table(self) -> np.ndarray:
        """first[z] = first cell-sorted position of z-layer z of the grid (len = layers + 1, last entry = n): what the
        sharding plan needs to know every rank's halo (valid after a grid build)."""
        nl = C.c_int64(0)
        _ffi.check(self.engine.lib.sf_cloud_layer_table(self.engine.h, self.h, None, 0, C.byref(nl)), "sf_cloud_layer_table")
        first = np.zeros(nl.value + 1, dtype=np.int64)
        _ffi.check(self.engine.lib.sf_cloud_layer_table(self.engine.h, self.h, _ptr(first), first.size, C.byref(nl)),
                   "sf_cloud_layer_table")
        return first

    def halo_range(self, begin: int, end: int) -> tuple[int, int]:
        """Range of cell-sorted positions holding every point within one grid cell of block [begin, end)."""
        hb, he = C.c_int64(0), C.c_int64(0)
        _ffi.check(
            self.engine.lib.sf_cloud_halo_range(self.engine.h, self.h, begin, end, C.byref(hb), C.byref(he)),
            "sf_cloud_halo_range",
        )
        return hb.value, he.value

    def radius_search(self, queries, radius: float) -> "Neighbors":
        q = _f64(queries, 3)
        h = _ffi.check_handle(
            self.engine.lib.sf_radius_search(self.engine.h, self.h, _ptr(q), q.shape[0], float(radius), SF_HOST),
            "sf_radius_search",
        )
        return Neighbors(self, h)

    def knn_search(self, queries, k: int) -> "Neighbors":
        """Lists of the k nearest cloud points of each query (KDTree.query semantics)."""
        q = _f64(queries, 3)
        if not 1 <= int(k) <= self.n:
            raise ValueError(f"k={k} must be between 1 and the number of cloud points ({self.n})")
        h = _ffi.check_handle(
            self.engine.lib.sf_knn_search(self.engine.h, self.h, _ptr(q), q.shape[0], int(k), SF_HOST), "sf_knn_search"
        )
        return Neighbors(self, h)

    def import_neighbors(self, queries, neighborhoods, radius: float) -> "Neighbors":
        """The caller's own lists as a list set (sf_nbrs_import): `neighborhoods[i]` = integer indices into this cloud's points
        that make up keypoint i's neighbourhood -- the object array KDTree.query_radius returns, the 2-D array of KDTree.query,
        or any sequence of index sequences (shot_parallelization.py:46-84: `support[neighborhoods[i]]`).  `radius` is what the
        frame / descriptor formulas are evaluated with; it selects nothing."""
        q = _f64(queries, 3)
        m = q.shape[0]
        if len(neighborhoods) != m:
            raise ValueError(f"one neighbourhood per keypoint expected ({len(neighborhoods)} lists for {m} keypoints)")
        nb_arr = neighborhoods if isinstance(neighborhoods, np.ndarray) and neighborhoods.dtype != object else None
        if nb_arr is not None and nb_arr.ndim == 2:  # (KDTree.query: every list of one length)
            offsets = np.arange(m + 1, dtype=np.int64) * nb_arr.shape[1]
            idx = np.ascontiguousarray(nb_arr, dtype=np.int64).reshape(-1)
        else:
            lists = [np.asarray(a).reshape(-1) for a in neighborhoods]
            for a in lists:
                if a.size and a.dtype.kind not in "iu":  # (support[float_array] raises IndexError in NumPy)
                    raise IndexError("arrays used as indices must be of integer (or boolean) type")
            offsets = np.zeros(m + 1, dtype=np.int64)
            if m:
                np.cumsum([a.size for a in lists], out=offsets[1:])
            idx = np.concatenate(lists).astype(np.int64, copy=False) if offsets[-1] else np.zeros(0, dtype=np.int64)
            idx = np.ascontiguousarray(idx)
        if idx.size:
            # NumPy's fancy indexing accepts -n .. n-1 and raises IndexError beyond: the same here
            lo, hi = int(idx.min()), int(idx.max())
            if lo < -self.n or hi >= self.n:
                bad = lo if lo < -self.n else hi
                raise IndexError(f"index {bad} is out of bounds for axis 0 with size {self.n}")
            if lo < 0:
                idx = np.where(idx < 0, idx + self.n, idx)
        h = _ffi.check_handle(
            self.engine.lib.sf_nbrs_import(self.engine.h, self.h, _ptr(q), m, _ptr(offsets), _ptr(idx), float(radius), SF_HOST),
            "sf_nbrs_import",
        )
        return Neighbors(self, h)

    def normals_radius(self, queries, radius: float, pre_computed_normals=None) -> np.ndarray:
        """compute_normals(queries, cloud, radius=...) in one sweep: the neighbour lists are never materialised
        (sf_normals_radius); bit-identical to radius_search(queries, radius).normals(pre_computed_normals)."""
        q = _f64(queries, 3)
        pre = None if pre_computed_normals is None else _f64(pre_computed_normals, 3)
        if pre is not None and pre.shape[0] != q.shape[0]:
            raise ValueError("pre_computed_normals must have one row per query point")
        out = np.zeros((q.shape[0], 3))
        _ffi.check(self.engine.lib.sf_normals_radius(self.engine.h, self.h, _ptr(q), q.shape[0], 0, 0, float(radius), _ptr(pre),
                                                    _ptr(out), SF_HOST), "sf_normals_radius")
        return out

    def normals_radius_self(self, radius: float, out: DeviceArray, begin: int = 0, end: Optional[int] = None) -> DeviceArray:
        """The same for the cloud's own points at cell-sorted positions [begin, end), result resident (row i = position begin + i)."""
        end = self.n if end is None else end
        _ffi.check(self.engine.lib.sf_normals_radius(self.engine.h, self.h, None, 0, begin, end, float(radius), None, out.ptr,
                                                    SF_IN_DEVICE | SF_OUT_DEVICE), "sf_normals_radius")
        return out

    def radius_search_self(self, radius: float, begin: int = 0, end: Optional[int] = None) -> "Neighbors":
        end = self.n if end is None else end
        h = _ffi.check_handle(
            self.engine.lib.sf_radius_search_self(self.engine.h, self.h, float(radius), begin, end), "sf_radius_search_self"
        )
        return Neighbors(self, h)

    def free(self) -> None:
        if getattr(self, "h", None) and self.engine.h:
            self.engine.lib.sf_cloud_free(self.engine.h, self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Neighbors:
    """Device-resident CSR radius-neighbour lists of one query set (result of kernels K2)."""

    def __init__(self, cloud: Cloud, handle):
        self.cloud, self.engine, self.h = cloud, cloud.engine, handle
        lib = self.engine.lib
        self.m = lib.sf_nbrs_num_queries(handle)
        self.total = lib.sf_nbrs_total(handle)
        self.max_count = lib.sf_nbrs_max_count(handle)
        self.max_count_all = lib.sf_nbrs_max_count_all(handle)  # over every rank, when Engine.collective_stats is on

    def slice(self, first: int, count: int) -> "Neighbors":
        """Non-owning view of queries [first, first+count); keep the parent alive while it is used."""
        h = _ffi.check_handle(self.engine.lib.sf_nbrs_slice(self.engine.h, self.h, first, count), "sf_nbrs_slice")
        v = Neighbors(self.cloud, h)
        v._parent = self
        return v

    def export(self, return_distance: bool = False):
        """(offsets[m+1], idx[total]) with ascending original indices inside each list
        [+ dist[total] = sqrt(d2)], the canonical form of KDTree.query_radius's output."""
        off = np.zeros(self.m + 1, dtype=np.int64)
        idx = np.zeros(max(self.total, 1), dtype=np.int32)
        dist = np.zeros(max(self.total, 1), dtype=np.float64) if return_distance else None
        _ffi.check(
            self.engine.lib.sf_nbrs_export(self.engine.h, self.cloud.h, self.h, _ptr(off), _ptr(idx), _ptr(dist)),
            "sf_nbrs_export",
        )
        if return_distance:
            return off, idx[: self.total], dist[: self.total]
        return off, idx[: self.total]

    def counts(self) -> np.ndarray:
        """Neighbourhood size of every query (len of each KDTree.query_radius list), without the lists."""
        off = np.zeros(self.m + 1, dtype=np.int64)
        _ffi.check(self.engine.lib.sf_nbrs_export(self.engine.h, self.cloud.h, self.h, _ptr(off), None, None), "sf_nbrs_export")
        return np.diff(off)

    # ---- descriptors on these lists ---------------------------------------------------------------
    def normals(self, pre_computed_normals=None, out: Optional[DeviceArray] = None):
        """K3.  out ((m, 3) device array): the normals stay in HBM (no pre_computed_normals then)."""
        if out is not None:
            if pre_computed_normals is not None:
                raise ValueError("device-resident output takes no pre_computed_normals")
            _ffi.check(self.engine.lib.sf_normals(self.engine.h, self.cloud.h, self.h, None, out.ptr, SF_OUT_DEVICE), "sf_normals")
            return out
        pre = None if pre_computed_normals is None else _f64(pre_computed_normals, 3)
        if pre is not None and pre.shape[0] != self.m:
            raise ValueError("pre_computed_normals must have one row per query point")
        out = np.zeros((self.m, 3))
        _ffi.check(self.engine.lib.sf_normals(self.engine.h, self.cloud.h, self.h, _ptr(pre), _ptr(out), SF_HOST), "sf_normals")
        return out

    def pca(self, moments: bool = False):
        """Local PCA of every neighbourhood: (eigenvalues (m,3) ascending, eigenvectors (m,3,3) as np.linalg.eigh
        returns them[, moments (m,8)]) -- pca() / compute_local_pca_with_moments of the reference."""
        w, v = np.zeros((self.m, 3)), np.zeros((self.m, 3, 3))
        mo = np.zeros((self.m, 8)) if moments else None
        _ffi.check(self.engine.lib.sf_pca(self.engine.h, self.cloud.h, self.h, _ptr(w), _ptr(v), _ptr(mo), SF_HOST), "sf_pca")
        return (w, v, mo) if moments else (w, v)

    def shot_lrf(self, out: Optional[DeviceArray] = None):
        if out is not None:
            _ffi.check(self.engine.lib.sf_shot_lrf(self.engine.h, self.cloud.h, self.h, out.ptr, SF_OUT_DEVICE), "sf_shot_lrf")
            return out
        lrf = np.zeros((self.m, 3, 3))
        _ffi.check(self.engine.lib.sf_shot_lrf(self.engine.h, self.cloud.h, self.h, _ptr(lrf), SF_HOST), "sf_shot_lrf")
        return lrf

    def shot(self, lrf, normalize: bool = True, min_neighborhood_size: int = 100, out: Optional[DeviceArray] = None,
             host_out: Optional[np.ndarray] = None):
        """host_out: a C-contiguous float64 array of (m, 352) that receives the rows (a slice of a larger result, e.g. one
        scale of compute_descriptor_multiscale's stack: the device-to-host copy lands where the rows are wanted)."""
        if isinstance(lrf, DeviceArray):
            if out is None:
                raise ValueError("device-resident LRFs need a device output")
            _ffi.check(
                self.engine.lib.sf_shot(self.engine.h, self.cloud.h, self.h, lrf.ptr, int(bool(normalize)),
                                        int(min_neighborhood_size), out.ptr, SF_IN_DEVICE | SF_OUT_DEVICE),
                "sf_shot",
            )
            return out
        lrf = _f64(lrf).reshape(-1, 9)
        if lrf.shape[0] != self.m:
            raise ValueError("one local reference frame per keypoint expected")
        if host_out is not None:
            if host_out.shape != (self.m, _ffi.SHOT_LEN) or host_out.dtype != np.float64 or not host_out.flags.c_contiguous:
                raise ValueError("host_out must be a C-contiguous float64 array of shape (m, 352)")
            res = host_out
        else:
            res = self.engine.host_empty((self.m, _ffi.SHOT_LEN))
        _ffi.check(
            self.engine.lib.sf_shot(self.engine.h, self.cloud.h, self.h, _ptr(lrf), int(bool(normalize)),
                                    int(min_neighborhood_size), _ptr(res), SF_HOST),
            "sf_shot",
        )
        return res

    def shot_single_scale(self, normalize: bool = True, min_neighborhood_size: int = 100, out: Optional[DeviceArray] = None,
                          lrf_out: Optional[DeviceArray] = None):
        """Frames + descriptors from these lists in one call (the frame's sign votes fused into the SHOT kernel)."""
        if out is not None:
            _ffi.check(
                self.engine.lib.sf_shot_single_scale(self.engine.h, self.cloud.h, self.h, int(bool(normalize)),
                                                     int(min_neighborhood_size), None if lrf_out is None else lrf_out.ptr,
                                                     out.ptr, SF_OUT_DEVICE),
                "sf_shot_single_scale",
            )
            return out
        res = self.engine.host_empty((self.m, _ffi.SHOT_LEN))
        _ffi.check(
            self.engine.lib.sf_shot_single_scale(self.engine.h, self.cloud.h, self.h, int(bool(normalize)),
                                                 int(min_neighborhood_size), None, _ptr(res), SF_HOST),
            "sf_shot_single_scale",
        )
        return res

    def shot_serial(self, min_neighborhood_size: int = 10) -> np.ndarray:
        """compute_shot_descriptor (shot.py:310-499): frames from the neighbours at non-zero distance, rows normalised."""
        res = self.engine.host_empty((self.m, _ffi.SHOT_LEN))
        _ffi.check(
            self.engine.lib.sf_shot_serial(self.engine.h, self.cloud.h, self.h, int(min_neighborhood_size), _ptr(res), SF_HOST),
            "sf_shot_serial",
        )
        return res

    def shot_from_moments(self, moments: DeviceArray, first_row: int, normalize: bool, min_neighborhood_size: int,
                          out: DeviceArray, lrf_out: Optional[DeviceArray] = None) -> DeviceArray:
        """shot_single_scale from frame moments Spfh.compute(..., moments_out=) left for these lists; `first_row` = row
        of `moments` that belongs to this object's first query (non-zero for a slice view)."""
        _ffi.check(
            self.engine.lib.sf_shot_from_moments(self.engine.h, self.cloud.h, self.h, moments.offset_ptr(first_row * 48),
                                                 int(bool(normalize)), int(min_neighborhood_size),
                                                 None if lrf_out is None else lrf_out.ptr, out.ptr, SF_OUT_DEVICE),
            "sf_shot_from_moments",
        )
        return out

    def lrf_raw_from_moments(self, moments: DeviceArray, first_row: int, lrf_out: DeviceArray) -> None:
        """First half of shot_from_moments: the eigen-solves (raw axes into lrf_out)."""
        _ffi.check(
            self.engine.lib.sf_lrf_raw_from_moments(self.engine.h, self.cloud.h, self.h, moments.offset_ptr(first_row * 48),
                                                    lrf_out.ptr),
            "sf_lrf_raw_from_moments",
        )

    def shot_from_raw_lrf(self, lrf: DeviceArray, normalize: bool, min_neighborhood_size: int, out: DeviceArray) -> DeviceArray:
        """Second half: the fused K5 completes the frames in `lrf` and writes the descriptors."""
        _ffi.check(
            self.engine.lib.sf_shot_from_raw_lrf(self.engine.h, self.cloud.h, self.h, lrf.ptr, int(bool(normalize)),
                                                 int(min_neighborhood_size), out.ptr),
            "sf_shot_from_raw_lrf",
        )
        return out

    def free(self) -> None:
        if getattr(self, "h", None) and self.engine.h:
            self.engine.lib.sf_nbrs_free(self.engine.h, self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Spfh:
    """Device-resident SPFH table (integer bin counts + neighbourhood sizes) of a whole cloud."""

    def __init__(self, cloud: Cloud, n_bins: int, max_count: int, radius: Optional[float] = None):
        """radius: the search radius the table will be computed with -- lets bin counts 6, 7 and 8 keep the window of bins a
        neighbourhood of that radius can fill in the byte table of the matrix-core path (sf_spfh_create_for_radius)."""
        self.cloud, self.engine, self.n_bins = cloud, cloud.engine, int(n_bins)
        self.edges = fpfh_edges(self.n_bins)
        if radius is not None and radius > 0:
            self.h = _ffi.check_handle(
                self.engine.lib.sf_spfh_create_for_radius(self.engine.h, cloud.h, self.n_bins, int(max_count), float(radius)),
                "sf_spfh_create_for_radius")
        else:
            self.h = _ffi.check_handle(
                self.engine.lib.sf_spfh_create(self.engine.h, cloud.h, self.n_bins, int(max_count)), "sf_spfh_create"
            )
        self.elem_bytes = int(self.engine.lib.sf_spfh_elem_bytes(self.h))  # 1: byte table (matrix-core K7), 2 / 4: wider counts

    def compute(self, self_nbrs: Neighbors, moments_out: Optional[DeviceArray] = None) -> "Spfh":
        """K6 for the queries of `self_nbrs`.  moments_out ((m, 6) device array): also leave the weighted covariance of
        the SHOT frame of every query there (Neighbors.shot_from_moments consumes it), from the same neighbour sweep."""
        if moments_out is not None:
            _ffi.check(
                self.engine.lib.sf_spfh_compute_moments(self.engine.h, self.cloud.h, self_nbrs.h, self.h, _ptr(self.edges),
                                                        moments_out.ptr),
                "sf_spfh_compute_moments",
            )
            return self
        _ffi.check(
            self.engine.lib.sf_spfh_compute(self.engine.h, self.cloud.h, self_nbrs.h, self.h, _ptr(self.edges)), "sf_spfh_compute"
        )
        return self

    def allgather(self, rows_per_rank: int) -> None:
        _ffi.check(self.engine.lib.sf_spfh_allgather(self.engine.h, self.h, int(rows_per_rank)), "sf_spfh_allgather")

    def exchange_rows(self, ops) -> None:
        """Neighbour-to-neighbour exchange of table rows.  ops: iterable of (peer, send_begin, send_end, recv_begin,
        recv_end) in cell-sorted positions (sharding.exchange_plan)."""
        ops = list(ops)
        n = len(ops)
        peers = (C.c_int * max(n, 1))(*[int(o[0]) for o in ops])
        cols = [(C.c_int64 * max(n, 1))(*[int(o[j]) for o in ops]) for j in (1, 2, 3, 4)]
        _ffi.check(self.engine.lib.sf_spfh_exchange_rows(self.engine.h, self.h, n, peers, *cols), "sf_spfh_exchange_rows")

    def rows_image(self, begin: int, end: int) -> np.ndarray:
        """Wire image (bytes) of table rows [begin, end): what exchange_rows would send for them."""
        need = C.c_size_t(0)
        lib, eh = self.engine.lib, self.engine.h
        _ffi.check(lib.sf_spfh_rows_image(eh, self.h, int(begin), int(end), None, 0, 0, C.byref(need)), "sf_spfh_rows_image")
        img = np.zeros(need.value, dtype=np.uint8)
        _ffi.check(lib.sf_spfh_rows_image(eh, self.h, int(begin), int(end), _ptr(img), img.size, 0, None), "sf_spfh_rows_image")
        return img

    def set_rows_image(self, begin: int, end: int, image: np.ndarray) -> None:
        """Write a wire image (rows_image of the owning rank) into rows [begin, end): the host-staged exchange."""
        img = np.ascontiguousarray(image, dtype=np.uint8)
        _ffi.check(self.engine.lib.sf_spfh_rows_image(self.engine.h, self.h, int(begin), int(end), _ptr(img), img.size, 1, None),
                   "sf_spfh_rows_image")

    def export(self) -> np.ndarray:
        out = np.zeros((self.cloud.n, self.n_bins**3))
        _ffi.check(self.engine.lib.sf_spfh_export(self.engine.h, self.cloud.h, self.h, _ptr(out), SF_HOST), "sf_spfh_export")
        return out

    def fpfh(self, self_nbrs: Neighbors, keypoints_indices=None, out: Optional[DeviceArray] = None, out_row: int = 0):
        """K7.  out (device): rows [out_row, out_row + m) receive the descriptors."""
        nb3 = self.n_bins**3
        if keypoints_indices is None:
            m, kp = self_nbrs.m, None
        else:
            kp = np.ascontiguousarray(keypoints_indices, dtype=np.int64)
            m = kp.shape[0]
        if out is not None:
            if out_row < 0 or out_row + m > out.shape[0]:
                raise ValueError("fpfh: output rows outside the array")
            _ffi.check(
                self.engine.lib.sf_fpfh(self.engine.h, self.cloud.h, self_nbrs.h, self.h, _ptr(kp), m,
                                        out.offset_ptr(out_row * nb3 * 8), SF_OUT_DEVICE),
                "sf_fpfh",
            )
            return out
        res = self.engine.host_empty((m, nb3))
        _ffi.check(
            self.engine.lib.sf_fpfh(self.engine.h, self.cloud.h, self_nbrs.h, self.h, _ptr(kp), m, _ptr(res), SF_HOST), "sf_fpfh"
        )
        return res

    def free(self) -> None:
        if getattr(self, "h", None) and self.engine.h:
            self.engine.lib.sf_spfh_free(self.engine.h, self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


_default: Optional[Engine] = None
_default_lock = threading.Lock()


def default_engine() -> Engine:
    """Process-wide engine on GPU LOCAL_RANK (or 0), created on first use.  A forked child must not
    reuse its parent's HIP context ('HIP + fork'), so the engine is re-created per pid."""
    global _default
    with _default_lock:
        if _default is None or _default.pid != os.getpid() or _default.h is None:
            _default = Engine()
        return _default
