"""Oracle-backed stand-in for shot_fpfh_amd.engine.Engine -- TEST INFRASTRUCTURE ONLY.

It lets the multi-process sharding logic (shot_fpfh_amd/sharding.py: block planning, halo ranges, list
slicing, SPFH exchange) run under world_size-2 gloo on a machine without GPUs.  It implements exactly
the calls DescriptorJob makes, with NumPy + the CPU oracle, and deliberately poisons (NaN) every SPFH
row a rank did not compute so that a wrong halo or a missing exchange cannot go unnoticed.
"""
import numpy as np

from oracle import oracle as O


class FakeArray:
    def __init__(self, shape, dtype=np.float64):
        self.a = np.full(shape, np.nan) if np.dtype(dtype) == np.float64 else np.zeros(shape, dtype)
        self.dtype_ = np.dtype(dtype)
        self.shape = self.a.shape
        self.nbytes = self.a.nbytes

    def to_host(self):
        return self.a.copy()

    def from_host(self, x):
        self.a[...] = x
        return self

    @property
    def dtype(self):
        return self.a.dtype

    def copy_from_device(self, src, dst_byte_offset=0, nbytes=None):
        flat, sflat = self.a.reshape(-1).view(np.uint8), src.a.reshape(-1).view(np.uint8)
        nbytes = src.nbytes if nbytes is None else nbytes
        flat[dst_byte_offset:dst_byte_offset + nbytes] = sflat[:nbytes]
        return self

    def free(self):
        pass


class FakeNbrs:
    def __init__(self, cloud, radius, begin, off, idx_sorted):
        self.cloud, self.radius, self.begin = cloud, radius, begin
        self.off, self.idx = off, idx_sorted
        self.m = len(off) - 1
        self.total = int(off[-1] - off[0])
        self.max_count = int(np.diff(off).max()) if self.m else 0
        self.max_count_all = self.max_count
        eng = getattr(cloud, "engine", None)
        if eng is not None and eng.stats_on:  # sf_comm_collective_stats: the longest list of ANY rank
            import torch
            import torch.distributed as dist

            t = torch.tensor([self.max_count], dtype=torch.int64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            self.max_count_all = int(t[0])

    def slice(self, first, count):
        return FakeNbrs(self.cloud, self.radius, self.begin + first, self.off[first:first + count + 1], self.idx)

    def _queries(self):
        return self.cloud.ps[self.begin:self.begin + self.m]

    def shot_lrf(self, out=None):
        out.a[:] = O.shot_lrf(self.cloud.ps, self._queries(), self.radius).reshape(self.m, 9)
        return out

    def shot(self, lrf, normalize, min_nb, out=None):
        out.a[:] = O.shot(self.cloud.ps, self.cloud.ns, self._queries(), self.radius, lrf.a, normalize, min_nb)
        return out

    def shot_single_scale(self, normalize=True, min_neighborhood_size=100, out=None, lrf_out=None):
        self.shot_lrf(out=lrf_out)
        return self.shot(lrf_out, normalize, min_neighborhood_size, out=out)

    def lrf_raw_from_moments(self, moments, first_row, lrf_out):
        # first half of the shared-sweep form (the device engine leaves raw axes here; the stand-in the frames)
        assert moments.a.shape[0] >= first_row + self.m
        self.shot_lrf(out=lrf_out)

    def shot_from_raw_lrf(self, lrf, normalize, min_neighborhood_size, out):
        return self.shot(lrf, normalize, min_neighborhood_size, out=out)

    def shot_from_moments(self, moments, first_row, normalize, min_neighborhood_size, out, lrf_out=None):
        # the shared-sweep form of the device engine: same result as the two-kernel form (the oracle has no moments)
        assert moments.a.shape[0] >= first_row + self.m
        return self.shot_single_scale(normalize, min_neighborhood_size, out=out, lrf_out=lrf_out)

    def free(self):
        pass


class FakeSpfh:
    def __init__(self, cloud, n_bins, max_count=0):
        self.cloud, self.n_bins = cloud, n_bins
        # the storage the device table would have (engine.Engine.spfh / sf_spfh_create): bytes, bytes + high-byte rows, 16, 32 bits
        self.storage = (0 if max_count <= 255 else 3) if max_count <= 65535 and n_bins**3 <= 128 else (2 if max_count > 65535 else 1)
        self.table = np.full((cloud.n, n_bins**3), np.nan)
        self.k = np.zeros(cloud.n, dtype=np.int64)
        self._full = None

    def compute(self, nb, moments_out=None):
        if self._full is None:  # the oracle's SPFH of the whole cloud, by sorted position
            _, sp = O.compute_fpfh_descriptor(np.zeros(0, dtype=np.int64), self.cloud.ps, self.cloud.ns, nb.radius,
                                              self.n_bins, return_spfh=True)
            self._full = sp
        self.table[:] = np.nan
        rows = slice(nb.begin, nb.begin + nb.m)
        self.table[rows] = self._full[rows]
        self.k[rows] = np.diff(nb.off)
        return self

    def allgather(self, rows_per_rank):
        import torch
        import torch.distributed as dist

        world, rank = dist.get_world_size(), dist.get_rank()
        kinds = [None] * world
        dist.all_gather_object(kinds, self.storage)
        if len(set(kinds)) != 1:  # sf_spfh_allgather's format word: ranks that disagree fail loudly (on the device they would hang)
            raise RuntimeError(f"sf_spfh_allgather: the ranks hold SPFH tables of different storage {kinds}")
        pad = rows_per_rank * world
        buf = np.zeros((pad, self.table.shape[1]))
        buf[: self.cloud.n] = np.nan_to_num(self.table, nan=-7.0)
        mine = torch.from_numpy(buf[rank * rows_per_rank:(rank + 1) * rows_per_rank].copy())
        parts = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        full = torch.cat(parts).numpy()[: self.cloud.n]
        self.table = np.where(full == -7.0, np.nan, full)

    def exchange_rows(self, ops):
        """The neighbour-to-neighbour exchange over gloo point-to-point messages."""
        import torch
        import torch.distributed as dist

        reqs, landing = [], []
        for peer, sb, se, rb, re in ops:
            if se > sb:
                assert not np.isnan(self.table[sb:se]).any()  # a rank only lends rows it computed
                reqs.append(dist.isend(torch.from_numpy(self.table[sb:se].copy()), peer))
            if re > rb:
                buf = torch.zeros((re - rb, self.table.shape[1]), dtype=torch.float64)
                reqs.append(dist.irecv(buf, peer))
                landing.append((rb, re, buf))
        for r in reqs:
            r.wait()
        for rb, re, buf in landing:
            self.table[rb:re] = buf.numpy()

    def fpfh(self, blk, kp, out=None, out_row=0):
        ps = self.cloud.ps
        for q in range(blk.m):
            i = blk.begin + q
            js = blk.idx[blk.off[q]:blk.off[q + 1]]
            d = np.sqrt(((ps[js] - ps[i]) ** 2).sum(axis=1))
            keep = d > 0
            acc = (self.table[js[keep]] / d[keep, None]).sum(axis=0) if keep.any() else 0.0
            out.a[out_row + q] = self.table[i] + acc / len(js)
        return out

    def free(self):
        pass


class FakeCloud:
    def __init__(self, points, normals):
        self.p, self.nr = np.asarray(points, dtype=np.float64), np.asarray(normals, dtype=np.float64)
        self.n = self.p.shape[0]

    def build_grid(self, cell, block=None, reach=2):
        # (the stand-in always sorts the whole cloud; `block` only restricts what the device build populates)
        edge = cell * (1.0 + 2.0**-20)
        self.lo = self.p.min(axis=0)
        c = np.floor((self.p - self.lo) / edge).astype(np.int64)
        self.dim = c.max(axis=0) + 1
        cid = (c[:, 2] * self.dim[1] + c[:, 1]) * self.dim[0] + c[:, 0]
        self._perm = np.argsort(cid, kind="stable")
        self.ps, self.ns = np.ascontiguousarray(self.p[self._perm]), np.ascontiguousarray(self.nr[self._perm])
        self.cz = c[self._perm, 2]

    def layer_table(self):
        return np.searchsorted(self.cz, np.arange(int(self.dim[2]) + 1), side="left").astype(np.int64)

    def perm(self):
        return self._perm.astype(np.int32)

    def halo_range(self, b, e):
        if b == e:
            return b, e
        z0, z1 = self.cz[b], self.cz[e - 1]
        inside = np.flatnonzero((self.cz >= z0 - 1) & (self.cz <= z1 + 1))
        return int(min(inside[0], b)), int(max(inside[-1] + 1, e))

    def radius_search_self(self, radius, begin=0, end=None):
        end = self.n if end is None else end
        off, idx = O.radius_search(self.ps, self.ps[begin:end], radius)
        return FakeNbrs(self, radius, begin, off, idx.astype(np.int64))

    def free(self):
        pass


class FakeMatchStream:
    """Stand-in of the streamed K8: a feed is only allowed to look at the rows it names -- it records their masks AS THEY ARE
    when they are fed (rows fed before they have landed would be caught by the poison the workers put there) -- and the end runs
    the oracle's masked arg-min over exactly what was fed."""

    def __init__(self, eng, a, a_ok, m1, b, b_ok, m2, b_entry_max, max_ranges):
        assert b_entry_max > 0.0 or m1 == 0
        self.eng, self.a, self.a_ok, self.m1, self.b, self.b_ok, self.m2, self.max_ranges = eng, a, a_ok, m1, b, b_ok, m2, max_ranges
        self.fed = np.zeros(m2, dtype=np.int32)
        self.rows = np.full((m2, b.shape[1]), np.nan)
        self.ok = np.zeros(m2, dtype=np.uint8)
        self.n_feeds = 0

    def feed(self, rb, re):
        assert rb % 64 == 0 and (re % 64 == 0 or re == self.m2) and 0 <= rb < re <= self.m2, (rb, re, self.m2)
        self.n_feeds += 1
        assert self.n_feeds <= self.max_ranges
        self.fed[rb:re] += 1
        self.rows[rb:re] = self.b.a[rb:re]
        self.ok[rb:re] = self.b_ok.a[rb:re]

    def end(self, idx, dist=None):
        assert (self.fed == 1).all(), "every reference row must be fed exactly once"
        got = FakeArray((self.m2, self.rows.shape[1]))
        got.a[:] = self.rows
        okb = FakeArray((self.m2,), np.uint8)
        okb.a[:] = self.ok
        self.eng.match_masked_device(self.a, self.a_ok, got, okb, idx, dist, a_rows=self.m1, b_rows=self.m2)

    def abort(self):
        pass


class FakeEngine:
    stats_on = False

    # the two-stream interface of the device engine: the stand-in runs everything in program order
    def fork(self):
        pass

    def switch(self, side):
        pass

    def join(self):
        pass

    def mark(self):
        pass

    def wait_mark(self):
        pass

    def cloud(self, points, normals=None):
        c = FakeCloud(points, normals)
        c.engine = self
        return c

    def empty(self, shape, dtype=np.float64):
        return FakeArray(shape, dtype)

    def allgather(self, buf, bytes_per_rank):
        import torch
        import torch.distributed as dist

        world, rank = dist.get_world_size(), dist.get_rank()
        flat = buf.a.reshape(-1).view(np.uint8)
        mine = torch.from_numpy(flat[rank * bytes_per_rank:(rank + 1) * bytes_per_rank].copy())
        parts = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        flat[: world * bytes_per_rank] = torch.cat(parts).numpy()

    def exchange(self, ops):
        """grouped point-to-point exchange over gloo: every receive posted, every send posted, all waited for"""
        import torch
        import torch.distributed as dist

        reqs, landing = [], []
        for (peer, sbuf, soff, sbytes, rbuf, roff, rbytes) in ops:
            if rbytes:
                t = torch.zeros(rbytes, dtype=torch.uint8)
                reqs.append(dist.irecv(t, src=peer))
                landing.append((rbuf, roff, rbytes, t))
        for (peer, sbuf, soff, sbytes, rbuf, roff, rbytes) in ops:
            if sbytes:
                reqs.append(dist.isend(torch.from_numpy(sbuf.a.reshape(-1).view(np.uint8)[soff:soff + sbytes].copy()), dst=peer))
        for r in reqs:
            r.wait()
        for rbuf, roff, rbytes, t in landing:
            rbuf.a.reshape(-1).view(np.uint8)[roff:roff + rbytes] = t.numpy()

    def rows_abs_max(self, rows, n_rows=None):
        m = rows.shape[0] if n_rows is None else n_rows
        return float(np.abs(rows.a[:m]).max()) if m else 0.0

    def match_stream(self, a, a_ok, m1, b, b_ok, m2, b_entry_max, max_ranges):
        return FakeMatchStream(self, a, a_ok, m1, b, b_ok, m2, b_entry_max, max_ranges)

    def sync(self):
        pass

    def allreduce_min_u64(self, buf, n=None):
        import torch
        import torch.distributed as dist

        n = buf.a.size if n is None else n
        # (gloo has no uint64: gather the words and fold them here)
        mine = torch.from_numpy(buf.a.reshape(-1)[:n].view(np.int64).copy())
        parts = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(parts, mine)
        buf.a.reshape(-1)[:n] = np.min(np.stack([p.numpy().view(np.uint64) for p in parts]), axis=0)

    def col_candidates_device(self, local_dist, global_dist, local_idx, row_offset, out, m=None):
        m = local_dist.a.shape[0] if m is None else m
        l, g = local_dist.a[:m], global_dist.a[:m].view(np.float64)
        out.a[:m] = np.where((l == g) & np.isfinite(l), (row_offset + local_idx.a[:m]).astype(np.uint64), np.uint64(2**64 - 1))
        return out

    def collective_stats(self, on):
        self.stats_on = bool(on)

    def rows_gather_device(self, rows, sel, out):
        pick = sel.a.astype(np.int64)
        out.a[: pick.shape[0]] = np.where((pick >= 0)[:, None], np.nan_to_num(rows.a, nan=0.0)[np.maximum(pick, 0)], 0.0)
        return out

    def rows_nonzero_device(self, rows, out=None, n_rows=None, first_row=0):
        m = rows.shape[0] - first_row if n_rows is None else n_rows
        out.a[first_row:first_row + m] = np.any(np.nan_to_num(rows.a[first_row:first_row + m], nan=1.0), axis=1)
        return out

    def match_masked_device(self, a, a_ok, b, b_ok, idx, dist=None, a_rows=None, b_rows=None):
        m1 = a.shape[0] if a_rows is None else a_rows
        m2 = b.shape[0] if b_rows is None else b_rows
        ok_b = np.flatnonzero(b_ok.a[:m2])
        if ok_b.size == 0:  # every column masked (a chunk of padding rows): +inf from everything, first column -- as the device does
            idx.a[:m1] = 0
            if dist is not None:
                dist.a[:m1] = np.inf
            return
        i, d = O.match_argmin(a.a[:m1], b.a[:m2][ok_b])
        idx.a[:m1] = ok_b[i]
        if dist is not None:
            dist.a[:m1] = d

    def spfh(self, cloud, n_bins, max_count, radius=None):
        return FakeSpfh(cloud, n_bins, max_count)
