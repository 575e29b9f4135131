"""world_size-2 (and 3) gloo tests of the sharded search orchestration on CPU: shard
bounds, local->global row offset, all-gather layout, pipelined submit/finish order and
the (distance,row) merge must reproduce the unsharded oracle result exactly.  The shard
backend and the merge are TEST-ONLY stand-ins built on the oracle (on the GPU box they
are qv_index_search_device / qv_merge_topk_shards_device, covered by tests/test_gpu_flat.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

N, D, K, NQ = 5003, 24, 10, 7      # N not divisible by the world size: ragged shards


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class OracleShard:
    """test stand-in for DeviceShard: this rank's rows scanned by the CPU oracle"""

    def __init__(self, rows):
        self.rows = rows

    def search(self, q, k, rows_out, dist_out):
        from tests import _oracle as O
        qs = q.numpy() if q.dim() == 2 else q.numpy()[None, :]
        rr = np.full((qs.shape[0], k), 0xFFFFFFFF, np.uint32)
        dd = np.full((qs.shape[0], k), np.inf, np.float32)
        for i in range(qs.shape[0]):
            r, d = O.exact_search(0, self.rows, qs[i], k)
            rr[i, : len(r)], dd[i, : len(d)] = r, d
        if q.dim() == 1:
            rr, dd = rr[0], dd[0]
        rows_out.copy_(torch.from_numpy(rr.view(np.int32)))
        dist_out.copy_(torch.from_numpy(dd))


def oracle_merge(g_pack, bases, k, rows_out, dist_out):
    """test stand-in for qv_merge_topk_shards_device: g_pack [G, 2, k] (shard-local rows, distance bits), bases [G];
    k smallest by (distance, global row)"""
    gp = g_pack.numpy()
    if gp.ndim == 4:                                   # a batch: [G, nq, 2, k] -> one merge per query
        for qi in range(gp.shape[1]):
            oracle_merge(torch.from_numpy(np.ascontiguousarray(gp[:, qi])), bases, k, rows_out[qi], dist_out[qi])
        return
    local = gp[:, 0, :].copy().view(np.uint32)
    d = gp[:, 1, :].copy().view(np.float32).ravel()
    glob = (local.astype(np.uint64) + bases.numpy().astype(np.uint64)[:, None]).astype(np.uint32)
    keep = (local != 0xFFFFFFFF).ravel()
    r = glob.ravel()
    d, r = d[keep], r[keep]
    order = np.lexsort((r, d))[:k]
    rr = np.full(k, 0xFFFFFFFF, np.uint32)
    dd = np.full(k, np.inf, np.float32)
    rr[: len(order)], dd[: len(order)] = r[order], d[order]
    rows_out.copy_(torch.from_numpy(rr.view(np.int32)))
    dist_out.copy_(torch.from_numpy(dd))


def _worker(rank, world, port, n_rows, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from quiver_amd.sharded import ShardedFlatSearch, shard_bounds
        from tests import _oracle as O
        base, n_local = shard_bounds(n_rows, world, rank)
        rows = O.gen_rows(20260424, base, n_local, D) if n_local else np.zeros((0, D), np.float32)
        s = ShardedFlatSearch(OracleShard(rows), base, K, torch.device("cpu"), merge=oracle_merge)
        qs = [torch.from_numpy(q) for q in O.gen_rows(20260425, 0, NQ, D)]
        res = s.search_stream(qs)                     # pipelined path
        one = s.search(qs[0])                         # unpipelined path
        assert torch.equal(one[0], res[0][0]) and torch.equal(one[1], res[0][1])
        br, bd = s.search_batch(torch.stack(qs))       # all queries in one exchange
        for i in range(len(qs)):
            assert torch.equal(br[i], res[i][0]) and torch.equal(bd[i], res[i][1])
        out_r = np.stack([r.numpy().view(np.uint32) for r, _ in res])
        out_d = np.stack([d.numpy() for _, d in res])
        ret[rank] = (out_r, out_d)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_rows", [(2, N), (3, N), (2, 7)])
def test_sharded_equals_unsharded(world, n_rows):
    from tests import _oracle as O
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), n_rows, ret), nprocs=world, join=True)
    corpus = O.gen_rows(20260424, 0, n_rows, D)
    qs = O.gen_rows(20260425, 0, NQ, D)
    for rank in range(world):
        out_r, out_d = ret[rank]
        for i in range(NQ):
            er, ed = O.exact_search(0, corpus, qs[i], K)
            n = len(er)
            assert np.array_equal(out_r[i, :n], er), (world, rank, i)
            assert np.array_equal(out_d[i, :n].view(np.uint32), ed.view(np.uint32))
            assert np.all(out_r[i, n:] == 0xFFFFFFFF)


def test_shard_bounds_cover_and_are_contiguous():
    from quiver_amd.sharded import shard_bounds
    for n in (0, 1, 7, 1000, 10_000_000):
        for g in (1, 2, 3, 4, 8):
            pos = 0
            for r in range(g):
                b, m = shard_bounds(n, g, r)
                assert b == pos and m >= 0
                pos += m
            assert pos == n
            sizes = [shard_bounds(n, g, r)[1] for r in range(g)]
            assert max(sizes) - min(sizes) <= 1
