"""Sharding of the descriptor path over the GPUs of one node (one process per GPU).

The reference has no distributed path (its only parallelism is a multiprocessing.Pool,
shot_parallelization.py:30-44).  Here the cloud is replicated on every GPU (an 8M-point cloud is
384 MB of float64 -- nothing next to 288 GB of HBM) and the QUERY points are partitioned: the grid
orders cells z-slowest, so rank g's contiguous block of cell-sorted positions is a z-slab.

* SHOT, normals and SPFH of a block need only read access to the cloud: no exchange.
* FPFH of a block needs the SPFH rows of the block's neighbours, some of which belong to the two
  adjacent slabs.  Two ways to get them (`spfh_exchange`):
    "neighbor"   compute only the block and borrow the halo rows from the ranks that own them: one grouped
                 ncclSend / ncclRecv with (normally) the two adjacent slabs, 64 bytes per row -- what K7 reads per
                 neighbour -- issued on the side stream under the FPFH reduction of the block's INTERIOR keypoints
                 (which need no foreign row); only the boundary layers wait for it;
    "halo"       recompute SPFH for the one-cell-thick halo on each side of the block -- a
                 contiguous run of positions -- so the descriptor pass has NO data-path collective
                 (24 % more K2 / K6 rows per interior rank of eight);
    "allgather"  compute only the block and all-gather the integer SPFH table over RCCL/xGMI.
* Matching needs every reference descriptor on every rank: one RCCL all-gather of descriptor rows
  (`gather_rows`) before K8, whose row arg-min is then local to each rank's scan block.

Results are bit-identical for any number of ranks: every kernel's arithmetic for a query depends only
on the cloud and the query, never on the block boundaries.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np

from .engine import Cloud, DeviceArray, Engine, Neighbors, Spfh

__all__ = ["ShardPlan", "ExchangePlan", "exchange_plan", "DescriptorJob", "MatchJob", "SubsetMatchJob"]


@dataclass(frozen=True)
class ShardPlan:
    """Equal blocks of ceil(n / world) cell-sorted positions; trailing ranks may get a short or empty block."""

    n: int
    world: int = 1
    rank: int = 0

    def __post_init__(self):
        if self.world < 1 or not 0 <= self.rank < self.world or self.n < 0:
            raise ValueError(f"bad shard plan n={self.n} world={self.world} rank={self.rank}")

    @property
    def rows_per_rank(self) -> int:
        return -(-self.n // self.world) if self.n else 0

    def block(self, rank: Optional[int] = None) -> tuple[int, int]:
        r = self.rank if rank is None else rank
        b = min(r * self.rows_per_rank, self.n)
        return b, min(b + self.rows_per_rank, self.n)

    @property
    def begin(self) -> int:
        return self.block()[0]

    @property
    def end(self) -> int:
        return self.block()[1]


@dataclass(frozen=True)
class ExchangePlan:
    """What one rank's FPFH pass borrows and lends (cell-sorted positions, identical numbering on every rank)."""

    halo: tuple[int, int]        # [hb, he): the block plus every row its keypoints' neighbours can be
    interior: tuple[int, int]    # [i0, i1): keypoints of the block all of whose neighbours the block itself holds
    ops: tuple                   # (peer, send_begin, send_end, recv_begin, recv_end), one per peer with anything to move


def _layer_of(first: np.ndarray, pos: int) -> int:
    """z-layer holding cell-sorted position `pos` (first[z] <= pos < first[z + 1]; empty layers are skipped)."""
    return int(np.searchsorted(first, pos, side="right")) - 1


def exchange_plan(layer_first, n: int, world: int, rank: int) -> ExchangePlan:
    """Halo, interior and the rows to exchange for `rank`, from the grid's layer table (Cloud.layer_table) alone.

    A keypoint in z-layer L reads rows of layers L - 1 .. L + 1 (the grid's cells are at least one radius thick), i.e.
    positions [first[L - 1], first[L + 2]).  Rank r therefore needs [first[zb - 1], first[ze + 2]) with zb / ze the layers
    of its first / last position; what of that lies in rank p's block is what p sends to r.  Blocks need not be aligned
    to layers, need not be a layer thick (a halo may then reach beyond the adjacent rank) and may be empty."""
    first = np.asarray(layer_first, dtype=np.int64)
    nl = first.size - 1
    if nl < 1 or first[0] != 0 or first[-1] != n:
        raise ValueError("layer table does not describe a cloud of this size")

    def halo_of(r: int):
        b, e = ShardPlan(n, world, r).block()
        if b == e:
            return b, e, b, e
        zb, ze = _layer_of(first, b), _layer_of(first, e - 1)
        return b, e, min(int(first[max(zb - 1, 0)]), b), max(int(first[min(ze + 2, nl)]), e)

    b, e, hb, he = halo_of(rank)
    # interior layers: the whole 3-layer reach inside [b, e)
    i0 = i1 = b
    if b < e:
        zb, ze = _layer_of(first, b), _layer_of(first, e - 1)
        ok = [z for z in range(zb, ze + 1) if first[max(z - 1, 0)] >= b and first[min(z + 2, nl)] <= e]
        if ok:
            i0, i1 = max(int(first[ok[0]]), b), min(int(first[ok[-1] + 1]), e)
    ops = []
    for p in range(world):
        if p == rank:
            continue
        pb, pe, phb, phe = halo_of(p)
        # what I receive: my halo outside my block, inside p's block (p is entirely below or entirely above me)
        rb, re = (max(hb, pb), min(b, pe)) if p < rank else (max(e, pb), min(he, pe))
        # what I send: p's halo outside p's block, inside my block
        sb, se = (max(phb, b), min(pb, e)) if rank < p else (max(pe, b), min(phe, e))
        rb, re = (rb, re) if re > rb else (0, 0)
        sb, se = (sb, se) if se > sb else (0, 0)
        if re > rb or se > sb:
            ops.append((p, sb, se, rb, re))
    return ExchangePlan((hb, he), (i0, i1), tuple(ops))


class DescriptorJob:
    """FPFH + SHOT for this rank's block of a resident cloud, outputs kept in HBM.

    One `step()` is one pass of the hot path: K1 grid build, K2 radius search (shared by both
    descriptors, their radii being equal), K6 SPFH, K7 FPFH, K4 local frames, K5 SHOT.
    When both descriptors are wanted, K6 -- which gathers every neighbour of every point anyway -- also accumulates
    the weighted covariance of the SHOT frame, and K4 shrinks to its eigen-solves (`share_sweep`).
    Rows of the outputs follow cell-sorted order; `block_original_indices()` maps them back.
    """

    def __init__(self, engine: Engine, points, normals, radius: float, n_bins: int = 5, normalize: bool = True,
                 min_neighborhood_size: int = 10, world: int = 1, rank: int = 0, spfh_exchange: str = "neighbor",
                 do_fpfh: bool = True, do_shot: bool = True, overlap_chains: bool = False, share_sweep: bool = True,
                 emulate_peers: bool = False):
        if spfh_exchange not in ("neighbor", "halo", "allgather"):
            raise ValueError("spfh_exchange must be 'neighbor', 'halo' or 'allgather'")
        # emulate_peers: ONE process stands in for rank `rank` of `world` (bench.py --emulate-rank): the rows the peers would
        # send are computed here once, before the first pass, and the passes themselves skip the exchange call
        self.emulate_peers = bool(emulate_peers)
        self._halo_filled = False
        self.engine, self.radius, self.n_bins = engine, float(radius), int(n_bins)
        self.normalize, self.min_nb = bool(normalize), int(min_neighborhood_size)
        self.exchange, self.do_fpfh, self.do_shot = spfh_exchange, do_fpfh, do_shot
        # overlap_chains: run the FPFH chain (K6, K7) and the SHOT chain (K4, K5) on the context's two HIP streams;
        # ~4 % faster at C3, but per-kernel durations then overlap, so the bench keeps it off by default
        self.overlap = bool(overlap_chains)
        self.share_sweep = bool(share_sweep) and do_fpfh and do_shot and self.n_bins <= 8  # (the fast K6 only)
        self.moments: Optional[DeviceArray] = None
        self.cloud: Cloud = engine.cloud(points, normals)
        self.plan = ShardPlan(self.cloud.n, world, rank)
        m = self.plan.end - self.plan.begin
        self.m = m
        self.fpfh_out: Optional[DeviceArray] = engine.empty((m, self.n_bins**3)) if do_fpfh else None
        self.lrf_out: Optional[DeviceArray] = engine.empty((m, 9)) if do_shot else None
        self.shot_out: Optional[DeviceArray] = engine.empty((m, 352)) if do_shot else None
        self.spfh: Optional[Spfh] = None
        self._spfh_wide = -1
        self.last_pairs = 0

    def _spfh_table(self, max_count: int) -> Spfh:
        # table kind: bytes (3: with the high-byte rows of the points that have more than 255 neighbours), 16 or 32 bits.  Bytes
        # for at most 128 bins -- and for 6, 7, 8 bins when the radius keeps alpha in its central bins (the table then holds that
        # window of the bins, Spfh(radius=...)): known from the table once one has been created
        byte_capable = self.n_bins**3 <= 128 or getattr(self, "_spfh_bytes", False)
        if max_count <= 65535 and byte_capable:
            wide = 0 if max_count <= 255 else 3
        else:
            wide = 2 if max_count > 65535 else 1
        if wide == 0 and self._spfh_wide == 3:
            wide = 3  # (a byte table that has its high-byte rows serves short lists as it is)
        if self.spfh is None or wide != self._spfh_wide:
            if self.spfh is not None:
                self.spfh.free()
            self.spfh = self.engine.spfh(self.cloud, self.n_bins, max_count, self.radius)
            self._spfh_bytes = getattr(self.spfh, "elem_bytes", 0) == 1
            if self._spfh_bytes and max_count <= 65535:
                wide = 3 if (max_count > 255 or wide == 3) else 0
            self._spfh_wide = wide
        return self.spfh

    # ---- world > 1, spfh_exchange="neighbor" -----------------------------------------------------------------------
    def _prefill_halo(self, hb: int, he: int) -> None:
        """emulate_peers: the SPFH rows of the halo, as the adjacent ranks would have sent them."""
        cloud, (b, e) = self.cloud, self.plan.block()
        cloud.build_grid(self.radius, block=(b, e), reach=2)
        nb = cloud.radius_search_self(self.radius, hb, he)
        try:
            self._spfh_table(nb.max_count).compute(nb)
        finally:
            nb.free()
        self._halo_filled = True

    def _exchange_plan(self) -> ExchangePlan:
        """(the plan is a function of the layer table, which a resident cloud reproduces pass after pass: planned once)"""
        first = self.cloud.layer_table()
        key = first.tobytes()
        if getattr(self, "_xp_key", None) != key:
            self._xp_key, self._xp = key, exchange_plan(first, self.cloud.n, self.plan.world, self.plan.rank)
        return self._xp

    def _step_neighbor(self) -> None:
        eng, cloud, (b, e) = self.engine, self.cloud, self.plan.block()
        cloud.build_grid(self.radius, block=(b, e), reach=1)  # the block's own neighbours: one cell
        xp = self._exchange_plan() if self.do_fpfh else None
        if self.do_fpfh and self.emulate_peers and not self._halo_filled:
            self._prefill_halo(*xp.halo)
            cloud.build_grid(self.radius, block=(b, e), reach=1)
        fold = self.do_fpfh and not self.emulate_peers
        if fold:
            eng.collective_stats(True)  # every rank sizes its table by the longest list of ANY rank (same row format)
        try:
            nb: Neighbors = cloud.radius_search_self(self.radius, b, e)
        finally:
            if fold:
                eng.collective_stats(False)
        try:
            self.last_pairs = nb.total
            if not self.do_fpfh:
                nb.shot_single_scale(self.normalize, self.min_nb, out=self.shot_out, lrf_out=self.lrf_out)
                return
            kind = self._spfh_wide
            longest = max(nb.max_count_all, nb.max_count)
            if getattr(eng, "fold_max_count", None) is not None and not self.emulate_peers:
                longest = eng.fold_max_count(longest)  # (a transport without RCCL folds the maximum itself: bench.py, staged)
            spfh = self._spfh_table(longest)
            if self.emulate_peers and self._spfh_wide != kind and kind != -1:
                raise RuntimeError("emulate_peers: the table changed its storage after the halo rows were filled")
            shared = self.share_sweep  # (any list length: the kernels dispatch per keypoint, sf_nbrs_dispatch)
            if shared:
                if self.moments is None or self.moments.shape[0] < nb.m:
                    if self.moments is not None:
                        self.moments.free()
                    self.moments = eng.empty((nb.m, 6))
                spfh.compute(nb, moments_out=self.moments)
            else:
                spfh.compute(nb)
            i0, i1 = xp.interior
            pieces = [(i0, i1), (b, i0), (i1, e)]  # interior first: it is what the exchange hides under
            views = [(lo, nb.slice(lo - b, hi - lo)) if (lo, hi) != (b, e) else (lo, nb) for lo, hi in pieces if hi > lo]
            forked = False
            try:
                eng.fork()  # side stream: the exchange, then the frame eigen-solves
                forked = True
                if not self.emulate_peers:
                    spfh.exchange_rows(xp.ops)
                eng.mark()  # the borrowed rows are in place from here on
                if shared:
                    nb.lrf_raw_from_moments(self.moments, 0, self.lrf_out)
                eng.switch(0)
                for lo, view in views:
                    if (lo, lo + view.m) != (i0, i1):
                        eng.wait_mark()  # boundary keypoints read the borrowed rows (not the eigen-solves queued after them)
                    spfh.fpfh(view, None, out=self.fpfh_out, out_row=lo - b)
                eng.join()  # K5 needs the frames
                forked = False
                if shared:
                    nb.shot_from_raw_lrf(self.lrf_out, self.normalize, self.min_nb, self.shot_out)
                elif self.do_shot:
                    nb.shot_single_scale(self.normalize, self.min_nb, out=self.shot_out, lrf_out=self.lrf_out)
            finally:
                if forked:
                    eng.join()
                for _, view in views:
                    if view is not nb:
                        view.free()
        finally:
            nb.free()

    def step(self) -> None:
        cloud, (b, e) = self.cloud, self.plan.block()
        if self.plan.world > 1 and self.exchange == "neighbor":
            return self._step_neighbor()
        if self.plan.world > 1:
            # only this rank's slab of the replicated cloud is sorted: the block's queries reach one cell, the SPFH
            # rows of the block's halo (recomputed here) one more
            cloud.build_grid(self.radius, block=(b, e), reach=2 if self.do_fpfh and self.exchange == "halo" else 1)
        else:
            cloud.build_grid(self.radius)
        if self.do_fpfh and self.exchange == "halo" and self.plan.world > 1:
            hb, he = cloud.halo_range(b, e)
        else:
            hb, he = b, e
        # spfh_exchange="allgather": the table rows of EVERY rank land in every rank's table, so all of them must hold the same
        # storage (bytes / bytes + high-byte rows / 16 / 32 bits) -- sized by the longest list of ANY rank, as in
        # _step_neighbor; "halo" recomputes what it needs and exchanges nothing, a rank-local size is right there
        fold = self.plan.world > 1 and self.do_fpfh and self.exchange == "allgather"
        if fold:
            self.engine.collective_stats(True)
        try:
            nb: Neighbors = cloud.radius_search_self(self.radius, hb, he)
        finally:
            if fold:
                self.engine.collective_stats(False)
        try:
            self.last_pairs = nb.total
            blk = nb if (hb, he) == (b, e) else nb.slice(b - hb, e - b)
            try:
                two_streams = self.overlap and self.do_fpfh and self.do_shot
                if self.do_fpfh:
                    longest = nb.max_count
                    if fold:
                        longest = max(nb.max_count_all, longest)
                        if getattr(self.engine, "fold_max_count", None) is not None:
                            longest = self.engine.fold_max_count(longest)  # (a transport without RCCL folds it itself)
                    spfh = self._spfh_table(longest)  # (allocates on first use: before forking)
                shared = self.share_sweep  # (any list length: the kernels dispatch per keypoint, sf_nbrs_dispatch)
                if shared:
                    if self.moments is None or self.moments.shape[0] < nb.m:
                        if self.moments is not None:
                            self.moments.free()
                        self.moments = self.engine.empty((nb.m, 6))
                    spfh.compute(nb, moments_out=self.moments)  # K6 + frame moments, before the chains part
                # the eigen-solves of the frames need only K6's moments and K7 needs only K6's table: side by side (the
                # small, long-latency eigen kernel disappears under K7); K5 follows both
                side_eig = shared and not two_streams
                forked = False
                try:
                    if side_eig:
                        self.engine.fork()
                        forked = True
                        blk.lrf_raw_from_moments(self.moments, b - hb, self.lrf_out)
                        self.engine.switch(0)
                    if two_streams:
                        self.engine.fork()  # FPFH chain on the side stream ...
                        forked = True
                    if self.do_fpfh:
                        if not shared:
                            spfh.compute(nb)
                        if self.exchange == "allgather" and self.plan.world > 1:
                            spfh.allgather(self.plan.rows_per_rank)
                        spfh.fpfh(blk, None, out=self.fpfh_out)
                    if two_streams:
                        self.engine.switch(0)  # ... the SHOT chain on the main one, side by side
                    if side_eig:
                        self.engine.join()
                        forked = False
                        blk.shot_from_raw_lrf(self.lrf_out, self.normalize, self.min_nb, self.shot_out)
                    elif self.do_shot:
                        if shared:
                            blk.shot_from_moments(self.moments, b - hb, self.normalize, self.min_nb, out=self.shot_out,
                                                  lrf_out=self.lrf_out)
                        else:
                            blk.shot_single_scale(self.normalize, self.min_nb, out=self.shot_out, lrf_out=self.lrf_out)
                finally:
                    if forked:  # also after an error: never leave the context on its side stream
                        self.engine.join()
            finally:
                if blk is not nb:
                    blk.free()
        finally:
            nb.free()

    def step_replay(self) -> None:
        """step() as ONE launch: the first calls run step() as it is (the second one is planned from the first's search record
        and has no read-back left), the third is captured into a HIP graph (Engine.capture) and every later call replays that
        graph -- same launches, same arguments, same rows bit for bit, without the thirty library calls and dozen launches
        behind a step.  What it buys is host time: nothing at 3.7 ms per step, a tenth of a rank's 0.5 ms share of a
        strong-scaled 1M-point cloud.  Valid while the job's buffers live (they do until close()).  With real peers
        (world > 1, not emulated) the exchange's RCCL calls would have to be captured too: not attempted, step() runs."""
        if self.plan.world > 1 and not self.emulate_peers:
            return self.step()
        g = getattr(self, "_graph", None)
        if g is not None:
            return g.launch()
        self._eager_steps = getattr(self, "_eager_steps", 0) + 1
        if getattr(self, "_graph_failed", None):
            return self.step()
        if self._eager_steps <= 2 or not getattr(self, "_sync_free", False):
            # an eager step, watched: only a step that never waited for the device can be captured
            before = self.engine.lib.sf_sync_count()
            self.step()
            self._sync_free = self.engine.lib.sf_sync_count() == before
            if self._eager_steps > 8 and not self._sync_free:
                self._graph_failed = "the step waits for the device (a host read-back in it): not capturable"
            return
        try:
            self._graph = self.engine.capture(self.step)
        except Exception as exc:  # noqa: BLE001 -- a step that cannot be captured keeps running eagerly; the reason is kept
            self._graph_failed = f"{type(exc).__name__}: {exc}"
            return self.step()
        return self._graph.launch()

    def block_original_indices(self) -> np.ndarray:
        b, e = self.plan.block()
        return self.cloud.perm()[b:e].astype(np.int64)

    def close(self) -> None:
        if getattr(self, "_graph", None) is not None:
            self._graph.free()
            self._graph = None
        for obj in (self.spfh, self.fpfh_out, self.lrf_out, self.shot_out, self.moments, self.cloud):
            if obj is not None:
                obj.free()
        self.spfh = self.fpfh_out = self.lrf_out = self.shot_out = self.moments = None


class MatchJob:
    """basic_matching (matching.py:149-169) of two descriptor sets that live SHARDED in HBM: rank g holds its block
    of the scan descriptors and its block of the reference descriptors (rows in the cell-sorted order of their
    clouds, as DescriptorJob leaves them).

    Exchange step: ONE all-gather of the reference rows over RCCL/xGMI (every scan row must see every reference
    row).  Then each rank runs K8 on its own scan block against the gathered reference set; all-zero
    descriptors (SHOT rows of too sparse neighbourhoods) are masked on the device instead of being compacted on
    the host.  The row arg-min is local to the rank -- no second collective on the data path; the small
    (index, flag) vectors are what a caller gathers afterwards.
    """

    def __init__(self, engine: Engine, length: int, n_scan: int, n_ref: int, world: int = 1, rank: int = 0, chunks: int = 1):
        """chunks > 1: the reference rows of the other ranks arrive in that many pieces per rank (grouped ncclSend / ncclRecv on
        the context's side stream, every piece landing in place) and K8's integer pass works on the pieces that have landed
        while the next ones travel (`run`, sf_match_stream_*); same matches bit for bit.  Needs blocks of a multiple of 64 rows
        per rank (SubsetMatchJob's are); otherwise the rows are gathered at once."""
        self.engine, self.d = engine, int(length)
        self.scan_plan, self.ref_plan = ShardPlan(n_scan, world, rank), ShardPlan(n_ref, world, rank)
        rpr = max(self.ref_plan.rows_per_rank, 1)
        self.ref_all: DeviceArray = engine.empty((rpr * world, self.d))
        self.ref_ok: DeviceArray = engine.empty((rpr * world,), np.uint8)
        m = self.scan_plan.end - self.scan_plan.begin
        self.scan_ok: DeviceArray = engine.empty((max(m, 1),), np.uint8)
        self.idx: DeviceArray = engine.empty((max(m, 1),), np.int64)
        self.dist: DeviceArray = engine.empty((max(m, 1),), np.float64)
        self.m = m
        self.chunks = 1
        if chunks > 1 and rpr % 64 == 0 and m > 0:
            self._piece = -(-(-(-rpr // int(chunks))) // 64) * 64  # rows of a piece: ceil(rpr / chunks), up to whole 64-row tiles
            self.chunks = -(-rpr // self._piece)

    def _pieces(self, c: int):
        """(rank, first row, end row) of chunk c's piece of every rank's block, in the gathered set's numbering, clipped to n."""
        plan, rpr = self.ref_plan, max(self.ref_plan.rows_per_rank, 1)
        out = []
        for r in range(plan.world):
            rb = r * rpr + c * self._piece
            re = min(r * rpr + min((c + 1) * self._piece, rpr), plan.n)
            if re > rb:
                out.append((r, rb, re))
        return out

    def run(self, scan_block: DeviceArray, ref_block: DeviceArray, gather: bool = True, match: bool = True) -> None:
        """scan_block: (m, d) rows of this rank's scan block; ref_block: this rank's reference block.
        gather=False / match=False (chunked form, measurements): K8 on the rows the previous run gathered / the exchange alone."""
        eng, plan = self.engine, self.ref_plan
        row_bytes = self.d * 8
        b, e = plan.block()
        self.ref_all.copy_from_device(ref_block, dst_byte_offset=b * row_bytes, nbytes=(e - b) * row_bytes)
        self._scan_block, self._ref_block = scan_block, ref_block
        if self.chunks > 1:
            return self._run_streamed(scan_block, gather, match)
        # (a lone context without a communicator has nothing to exchange and the call returns at once; with a
        # communicator -- of ONE rank too -- this is ncclAllGather)
        if gather:
            eng.allgather(self.ref_all, plan.rows_per_rank * row_bytes)
        if not match:
            return
        eng.rows_nonzero_device(self.ref_all, self.ref_ok, n_rows=plan.n)
        if self.m:
            eng.rows_nonzero_device(scan_block, self.scan_ok, n_rows=self.m)
            eng.match_masked_device(scan_block, self.scan_ok, self.ref_all, self.ref_ok, self.idx, self.dist,
                                    a_rows=self.m, b_rows=plan.n)

    def _bmax_over_ranks(self, local: float) -> float:
        """max over the ranks of a non-negative double, through the all-reduce(min) of 64-bit words the engine has: non-negative
        doubles order like their bit patterns, so the maximum is the complement of the minimum of the complements."""
        if self.ref_plan.world == 1:
            return local
        buf = self.engine.empty((1,), np.uint64)
        try:
            buf.from_host(~np.array([local], dtype=np.float64).view(np.uint64))
            self.engine.allreduce_min_u64(buf, 1)
            return float((~buf.to_host()).view(np.float64)[0])
        finally:
            buf.free()

    def _run_streamed(self, scan_block: DeviceArray, do_gather: bool = True, do_match: bool = True) -> None:
        """The exchange under K8 ("overlap C2 ... by chunking rows" -- here with K8 itself, the only consumer of the
        gathered rows).  Chunk c = rows [c piece, (c + 1) piece) of EVERY rank's block; the pieces of the other ranks land IN
        PLACE (one grouped ncclSend / ncclRecv per chunk, on the side stream), the main stream waits for chunk c only (sf_mark /
        sf_wait_mark), marks its empty rows, makes its int8 image and runs the integer pass over it (sf_match_stream_feed) while
        chunk c + 1 is in flight.  Cutting the COLUMNS must not cut the decision: a row's nearest descriptor sits in one chunk
        only, and seen from the other chunks the row has no clear minimum -- so only the integer minima are taken per chunk
        and the decision steps (live splits, candidates, float64) run once, over all chunks' minima (sf_match_stream_end)."""
        eng, plan, me = self.engine, self.ref_plan, self.ref_plan.rank
        row_bytes = self.d * 8
        b, e = plan.block()

        def exchange(c: int) -> None:
            """chunk c: one grouped exchange -- my piece to every peer, every peer's piece into its place in ref_all"""
            if not do_gather or plan.world == 1:
                return
            piece = {r: (rb, re) for (r, rb, re) in self._pieces(c)}
            sb, se = piece.get(me, (0, 0))
            ops = []
            for p in range(plan.world):
                if p == me:
                    continue
                rb, re = piece.get(p, (0, 0))
                if se > sb or re > rb:
                    ops.append((p, self.ref_all if se > sb else None, sb * row_bytes, (se - sb) * row_bytes,
                                self.ref_all if re > rb else None, rb * row_bytes, (re - rb) * row_bytes))
            eng.exchange(ops)

        stream = None
        if do_match:
            eng.rows_nonzero_device(scan_block, self.scan_ok, n_rows=self.m)
            bmax = self._bmax_over_ranks(eng.rows_abs_max(self._ref_block, e - b) if e > b else 0.0)
            stream = eng.match_stream(scan_block, self.scan_ok, self.m, self.ref_all, self.ref_ok, plan.n, bmax,
                                      self.chunks * plan.world)
        eng.fork()  # the side stream starts behind everything issued so far (the copy of this rank's block into place)
        try:
            exchange(0)
            eng.mark()
            for c in range(self.chunks):
                eng.switch(0)
                eng.wait_mark()  # chunk c has landed (and only that is waited for)
                if c + 1 < self.chunks:
                    eng.switch(1)
                    exchange(c + 1)
                    eng.mark()
                    eng.switch(0)
                if stream is not None:
                    for (_, rb, re) in self._pieces(c):
                        eng.rows_nonzero_device(self.ref_all, self.ref_ok, n_rows=re - rb, first_row=rb)
                        stream.feed(rb, re)
        except Exception:
            if stream is not None:
                stream.abort()
            raise
        finally:
            eng.join()
        if stream is not None:
            stream.end(self.idx, self.dist)

    # ---- the rest of match_descriptors (matching.py:54-74) on sharded rows ------------------------------------------------
    def _all_ranks(self, mine: np.ndarray, fill) -> np.ndarray:
        """A per-scan-row vector of this rank's block -> the vector over ALL scan rows, in global row order (one small
        all-gather of equal, padded blocks)."""
        eng, plan = self.engine, self.scan_plan
        rpr = max(plan.rows_per_rank, 1)
        buf = eng.empty((rpr * plan.world,), mine.dtype)
        try:
            block = np.full(rpr, fill, dtype=mine.dtype)
            block[: mine.shape[0]] = mine
            host = np.full(rpr * plan.world, fill, dtype=mine.dtype)
            host[plan.rank * rpr:(plan.rank + 1) * rpr] = block
            buf.from_host(host)
            eng.allgather(buf, rpr * mine.dtype.itemsize)
            flat = buf.to_host()
        finally:
            buf.free()
        return np.concatenate([flat[r * rpr: r * rpr + (plan.block(r)[1] - plan.block(r)[0])] for r in range(plan.world)])

    def _sum_over_ranks(self, value: int) -> int:
        eng, plan = self.engine, self.scan_plan
        if plan.world == 1:
            return value
        buf = eng.empty((plan.world,), np.int64)
        try:
            host = np.zeros(plan.world, dtype=np.int64)
            host[plan.rank] = value
            buf.from_host(host)
            eng.allgather(buf, 8)
            return int(buf.to_host().sum())
        finally:
            buf.free()

    def column_argmin(self) -> np.ndarray:
        """distance_matrix.argmin(axis=0) of matching.py:63 over ALL scan rows, for every reference row (global scan row
        numbers; 2^64 - 1 where no non-empty scan row exists): local column arg-min over this rank's block, all-reduce(min)
        of the column minima, then all-reduce(min) of the rows that attain them -- the first minimum, as NumPy takes it."""
        eng, n_ref = self.engine, self.ref_plan.n
        rows = max(self.ref_all.shape[0], 1)
        cd, ci = eng.empty((rows,), np.float64), eng.empty((rows,), np.int64)
        gd, cand = eng.empty((rows,), np.uint64), eng.empty((rows,), np.uint64)
        try:
            if self.m:
                eng.match_masked_device(self.ref_all, self.ref_ok, self._scan_block, self.scan_ok, ci, cd, a_rows=n_ref, b_rows=self.m)
            else:  # an empty block reaches no column
                cd.from_host(np.full(rows, np.inf))
                ci.from_host(np.zeros(rows, np.int64))
            gd.copy_from_device(cd)
            eng.allreduce_min_u64(gd, n_ref)       # non-negative doubles order like their bit patterns
            eng.col_candidates_device(cd, gd, ci, self.scan_plan.begin, cand, m=n_ref)
            eng.allreduce_min_u64(cand, n_ref)
            return cand.to_host()[:n_ref]
        finally:
            for a in (cd, ci, gd, cand):
                a.free()

    def matches(self, filter_callback=None, filter_nonreciprocal: bool = False, n_min_matches: int = 100,
                **kwargs) -> tuple[np.ndarray, np.ndarray]:
        """(scan rows of this rank that are matched, their reference rows) -- both in the global, cell-sorted row numbering.

        Without arguments: basic_matching, every non-empty scan row with its nearest non-empty reference row.  With a
        `filter_callback` and / or `filter_nonreciprocal` it is match_descriptors' 2-D branch (matching.py:54-74) on the
        sharded rows: the callback sees the winners' distances of ALL ranks' non-empty scan rows in global row order
        (quantile / median filters need them all; one small all-gather), the reciprocity test uses the column arg-min over
        all ranks (`column_argmin`), and "applied only if at least n_min_matches survive" counts survivors over all ranks.
        Every rank must call it with the same arguments (it contains collectives)."""
        if not self.m and filter_callback is None and not filter_nonreciprocal:
            return np.zeros(0, np.int64), np.zeros(0, np.int64)
        ok = self.scan_ok.to_host()[: self.m].astype(bool) if self.m else np.zeros(0, bool)
        idx = self.idx.to_host()[: self.m] if self.m else np.zeros(0, np.int64)
        rows = np.flatnonzero(ok)
        keep = np.ones(rows.shape[0], dtype=bool)
        if filter_callback is not None:
            dist = self.dist.to_host()[: self.m] if self.m else np.zeros(0)
            ok_all = self._all_ranks(ok.astype(np.uint8), 0).astype(bool)
            dist_all = self._all_ranks(dist, np.inf)
            mask_all = np.asarray(filter_callback(dist_all[ok_all], **kwargs), dtype=bool)
            before = int(ok_all[: self.scan_plan.begin].sum())  # non-empty scan rows of the ranks below this one
            keep = mask_all[before: before + rows.shape[0]]
        if filter_nonreciprocal:
            col = self.column_argmin()
            both = keep & (col[idx[rows]] == (rows + self.scan_plan.begin).astype(np.uint64))
            survivors = self._sum_over_ranks(int(both.sum()))
            if survivors >= n_min_matches:
                keep = both
        return rows[keep] + self.scan_plan.begin, idx[rows][keep]

    def close(self) -> None:
        for a in (self.ref_all, self.ref_ok, self.scan_ok, self.idx, self.dist):
            a.free()


class SubsetMatchJob:
    """The tail of BASELINE config 5: match a keypoint SUBSET of two sharded descriptor sets.

    Every rank holds its block of the scan descriptors and its block of the reference descriptors (`DescriptorJob`
    outputs, rows in the cell-sorted order of their clouds).  `select()` picks the rows of the subset out of both
    blocks on the device (`sf_rows_gather`) into blocks of exactly `rows_per_rank` rows -- zero rows pad a rank that
    owns fewer, and zero descriptors are skipped by the matching like the reference's empty ones (matching.py:162-163)
    -- together with a caller-chosen int64 label per row (the original point index, say).  `run()` is the exchange
    + K8: one RCCL all-gather of the reference subset rows, one of their labels, then the row arg-min of this rank's
    scan subset over the gathered set.  `matches()` returns label pairs, so ranks need no common row numbering.
    """

    def __init__(self, engine: Engine, length: int, rows_per_rank: int, world: int = 1, rank: int = 0, chunks: int = 1):
        if rows_per_rank < 1:
            raise ValueError("rows_per_rank must be positive")
        self.engine, self.d, self.rows, self.world, self.rank = engine, int(length), int(rows_per_rank), world, rank
        total = self.rows * world
        self.job = MatchJob(engine, length, total, total, world, rank, chunks=chunks)
        self.scan_sub: DeviceArray = engine.empty((self.rows, self.d))
        self.ref_sub: DeviceArray = engine.empty((self.rows, self.d))
        self.sel: DeviceArray = engine.empty((self.rows,), np.int64)
        self.ref_labels: DeviceArray = engine.empty((total,), np.int64)
        self.scan_labels = np.full(self.rows, -1, dtype=np.int64)
        self._pairs: Optional[DeviceArray] = None

    def _padded(self, values, what: str) -> np.ndarray:
        v = np.ascontiguousarray(values, dtype=np.int64)
        if v.ndim != 1 or v.shape[0] > self.rows:
            raise ValueError(f"{what}: at most {self.rows} rows per rank, got shape {v.shape}")
        out = np.full(self.rows, -1, dtype=np.int64)
        out[: v.shape[0]] = v
        return out

    def select(self, scan_rows: DeviceArray, scan_sel, scan_labels, ref_rows: DeviceArray, ref_sel, ref_labels) -> None:
        """scan_sel / ref_sel: local row numbers (into this rank's blocks) of the subset; *_labels: one int64 each."""
        eng = self.engine
        if len(scan_sel) != len(scan_labels) or len(ref_sel) != len(ref_labels):
            raise ValueError("one label per selected row expected")
        self.scan_labels = self._padded(scan_labels, "scan_labels")
        eng.rows_gather_device(scan_rows, self.sel.from_host(self._padded(scan_sel, "scan_sel")), self.scan_sub)
        eng.rows_gather_device(ref_rows, self.sel.from_host(self._padded(ref_sel, "ref_sel")), self.ref_sub)
        mine = eng.empty((self.rows,), np.int64).from_host(self._padded(ref_labels, "ref_labels"))
        try:
            self.ref_labels.copy_from_device(mine, dst_byte_offset=self.rank * self.rows * 8)
            eng.sync()
        finally:
            mine.free()

    def run(self, gather: bool = True, match: bool = True) -> None:
        self.job.run(self.scan_sub, self.ref_sub, gather, match)  # all-gather of descriptor rows (C2) + K8
        self.engine.allgather(self.ref_labels, self.rows * 8)     # all-gather of the small label vector (C3)

    def matches(self) -> tuple[np.ndarray, np.ndarray]:
        """(labels of this rank's non-empty scan subset rows, labels of the reference rows they matched)."""
        rows, idx = self.job.matches()
        labels = self.ref_labels.to_host()
        return self.scan_labels[rows - self.job.scan_plan.begin], labels[idx]

    def gather_matches(self) -> tuple[np.ndarray, np.ndarray]:
        """Every rank's matches on every rank: one more all-gather, of `rows_per_rank` (scan label, reference label)
        pairs per rank (-1 marks an empty scan row).  What a caller needs to run RANSAC on the whole match set."""
        eng = self.engine
        s_lab, r_lab = self.matches()
        mine = np.full((self.rows, 2), -1, dtype=np.int64)
        mine[: s_lab.shape[0], 0], mine[: s_lab.shape[0], 1] = s_lab, r_lab
        if self._pairs is None:
            self._pairs = eng.empty((self.rows * self.world, 2), np.int64)
        block = eng.empty((self.rows, 2), np.int64).from_host(mine)
        try:
            self._pairs.copy_from_device(block, dst_byte_offset=self.rank * self.rows * 16)
            eng.allgather(self._pairs, self.rows * 16)
            allp = self._pairs.to_host()
        finally:
            block.free()
        keep = allp[:, 0] >= 0
        return allp[keep, 0], allp[keep, 1]

    def close(self) -> None:
        self.job.close()
        for a in (self.scan_sub, self.ref_sub, self.sel, self.ref_labels, self._pairs):
            if a is not None:
                a.free()
