"""Sharded exact search: the multi-GPU form of the flat scan.

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in the CPU tests).  Rank g owns the contiguous row shard [g*N/G, (g+1)*N/G); a
query goes to every rank, each rank scans its shard for its local top-k, the k
(shard-local row, distance) pairs per rank are all-gathered in ONE collective (the scan
writes rows and distances into one [2][k] buffer that goes into the all-gather as it is:
k*8 bytes per rank — a latency, not a bandwidth, collective) and merged on every rank under
the same (distance, global row) order the single-GPU scan uses — the merge adds each shard's
base row — so the sharded result is identical to the unsharded one.

The reference has no counterpart (single process); this is the one
exchange step the path has.

The shard backend and the merge are injected so the orchestration can be exercised on
CPU with gloo: on the GPU they are `DeviceShard` (qv_index_search_device) and
`qv_merge_topk_shards_device`; the CPU tests plug in test-only stand-ins.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch
import torch.distributed as dist


def shard_bounds(n_rows: int, world: int, rank: int):
    """contiguous row blocks; global row = base + local row"""
    base = rank * n_rows // world
    return base, (rank + 1) * n_rows // world - base


class DeviceShard:
    """a qv_index holding this rank's rows; search writes device tensors on the current stream"""

    def __init__(self, index):
        self.index = index

    def search(self, d_query: torch.Tensor, k: int, rows_out: torch.Tensor, dist_out: torch.Tensor):
        """d_query [dim] -> rows_out / dist_out [k]; d_query [nq, dim] -> [nq, k] views (may be strided slices of a packed buffer)"""
        s = torch.cuda.current_stream().cuda_stream
        if d_query.dim() == 1:
            self.index.search_device(d_query.data_ptr(), 1, k, rows_out.data_ptr(), dist_out.data_ptr(), s)
            return
        nq = d_query.shape[0]
        st_r = torch.empty((nq, k), dtype=torch.int32, device=d_query.device)
        st_d = torch.empty((nq, k), dtype=torch.float32, device=d_query.device)
        done = False
        if nq >= 32:
            # big batches: fp32-MFMA filter + exact re-score (same results); queries it flags are redone by the exact scan
            from ._lib import QvError
            flags = torch.zeros(nq, dtype=torch.int32, device=d_query.device)
            try:
                self.index.search_batched_device(d_query.data_ptr(), nq, k, st_r.data_ptr(), st_d.data_ptr(), flags.data_ptr(), s)
                done = True
                for qi in flags.nonzero().flatten().tolist():
                    self.index.search_device(d_query[qi].data_ptr(), 1, k, st_r[qi].data_ptr(), st_d[qi].data_ptr(), s)
            except QvError as e:
                if e.code != -8:                          # QV_ERR_UNSUPPORTED: metric / corpus size / k -> exact scan below
                    raise
        if not done:
            self.index.search_device(d_query.data_ptr(), nq, k, st_r.data_ptr(), st_d.data_ptr(), s)   # exact multi-query scan
        rows_out.copy_(st_r); dist_out.copy_(st_d)


def device_merge(g_pack: torch.Tensor, bases: torch.Tensor, k: int, rows_out: torch.Tensor, dist_out: torch.Tensor):
    """g_pack [G, 2, k] (one query) or [G, nq, 2, k] int32: per shard (and query) k local rows then k distance bit patterns;
    bases [G] int32; outputs [k] or [nq, k]"""
    from .device_index import merge_topk_shards_device
    s = torch.cuda.current_stream().cuda_stream
    nq = 1 if g_pack.dim() == 3 else g_pack.shape[1]
    merge_topk_shards_device(g_pack.data_ptr(), bases.data_ptr(), g_pack.shape[0], nq, k, rows_out.data_ptr(), dist_out.data_ptr(), s)


class ShardedFlatSearch:
    def __init__(self, shard, base_row: int, k: int, device, group=None,
                 merge: Callable = device_merge, world: Optional[int] = None, ring: int = 4,
                 force_exchange: bool = False):
        self.shard, self.base, self.k, self.device, self.group = shard, int(base_row), k, device, group
        self.world = world if world is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        self.merge = merge
        self.force_exchange = force_exchange          # run the all-gather + merge even with one rank (test hook)
        # a ring of result/exchange buffers: slot i is reused by query i+ring; with submit/finish
        # pipelined one deep, any ring >= 3 is safe (stream order: merge(i) precedes scan(i+ring))
        self._ring = [self._buffers() for _ in range(max(3, ring))]
        self._n = 0
        # on the GPU the merge runs on a side stream (after the collective it depends on), so the scan stream never has a
        # merge between two scans; the caller's stream only waits for an event that fired long before it gets there
        self._side = torch.cuda.Stream(device=device) if torch.device(device).type == "cuda" else None
        # every shard's first global row, for the merge
        self.bases = torch.zeros(self.world, dtype=torch.int32, device=device)
        if self.world > 1:
            mine = torch.tensor([self.base], dtype=torch.int32, device=device)
            dist.all_gather_into_tensor(self.bases, mine, group=self.group)
        else:
            self.bases[0] = self.base

    def _buffers(self):
        k, G, dev = self.k, self.world, self.device
        pack = torch.empty((2, k), dtype=torch.int32, device=dev)        # [0] shard-local rows, [1] distance bits: one buffer, one collective
        return dict(pack=pack, rows=pack[0], dist=pack[1].view(torch.float32),
                    g_pack=torch.empty((G, 2, k), dtype=torch.int32, device=dev),
                    out_rows=torch.empty(k, dtype=torch.int32, device=dev), out_dist=torch.empty(k, dtype=torch.float32, device=dev))

    def submit(self, d_query: torch.Tensor):
        """local scan + start the exchange for this query; returns a ticket for finish().
        Calling submit() for query i+1 before finish() of query i overlaps the exchange of i
        with the scan of i+1 (the collectives run on the backend's own stream)."""
        b = self._ring[self._n % len(self._ring)]
        self._n += 1
        self.shard.search(d_query, self.k, b["rows"], b["dist"])
        if self.world == 1 and not self.force_exchange:
            return b, None
        w = dist.all_gather_into_tensor(b["g_pack"].view(-1), b["pack"].view(-1), group=self.group, async_op=True)
        return b, (w,)

    def finish(self, ticket):
        """-> (rows[k] int32 view of uint32 global rows, dist[k]); valid until the slot is reused"""
        b, works = ticket
        if works is None:
            if self.base:                                   # one rank, rows offset: local -> global (0xFFFFFFFF stays put)
                b["out_rows"].copy_(torch.where(b["rows"] != -1, b["rows"] + self.base, b["rows"]))
                return b["out_rows"], b["dist"]
            return b["rows"], b["dist"]
        if self._side is None:
            works[0].wait()
            self.merge(b["g_pack"], self.bases, self.k, b["out_rows"], b["out_dist"])
            return b["out_rows"], b["out_dist"]
        with torch.cuda.stream(self._side):
            works[0].wait()                                 # the side stream waits for the collective
            self.merge(b["g_pack"], self.bases, self.k, b["out_rows"], b["out_dist"])
            ev = b.get("ev") or torch.cuda.Event()
            ev.record(self._side)
            b["ev"] = ev
        torch.cuda.current_stream().wait_event(ev)          # results (and this slot's buffers) are ordered after the merge
        return b["out_rows"], b["out_dist"]

    def search(self, d_query: torch.Tensor):
        r, d = self.finish(self.submit(d_query))
        return r.clone(), d.clone()

    def search_batch(self, d_queries: torch.Tensor):
        """nq queries at once ([nq, dim]): every shard runs its multi-query scan, ONE all-gather carries all nq local top-k
        lists, one merge launch (a workgroup per query) -> (rows [nq, k], dist [nq, k]), identical to nq single searches"""
        nq, k, G, dev = d_queries.shape[0], self.k, self.world, self.device
        pack = torch.empty((nq, 2, k), dtype=torch.int32, device=dev)
        self.shard.search(d_queries, k, pack[:, 0, :], pack[:, 1, :].view(torch.float32))
        if self.world == 1 and not self.force_exchange:
            rows = pack[:, 0, :]
            if self.base:
                rows = torch.where(rows != -1, rows + self.base, rows)
            return rows.contiguous(), pack[:, 1, :].view(torch.float32).contiguous()
        g_pack = torch.empty((G, nq, 2, k), dtype=torch.int32, device=dev)
        dist.all_gather_into_tensor(g_pack.view(-1), pack.view(-1), group=self.group)
        out_rows = torch.empty((nq, k), dtype=torch.int32, device=dev); out_dist = torch.empty((nq, k), dtype=torch.float32, device=dev)
        self.merge(g_pack, self.bases, k, out_rows, out_dist)
        return out_rows, out_dist

    def search_stream(self, queries):
        """pipelined: exchange of query i overlaps the scan of query i+1"""
        out, pending = [], None
        for q in queries:
            t = self.submit(q)
            if pending is not None:
                r, d = self.finish(pending)
                out.append((r.clone(), d.clone()))
            pending = t
        if pending is not None:
            r, d = self.finish(pending)
            out.append((r.clone(), d.clone()))
        return out
