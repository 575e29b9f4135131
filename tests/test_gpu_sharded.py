"""Two ranks sharing the one GPU of the test box (gloo for the exchange, since RCCL refuses two
ranks on one device): every device piece of the multi-GPU path is the real one — per-rank
DeviceIndex shards, qv_index_search_device, the packed all-gather layout, qv_merge_topk_shards_device — and
the sharded result must equal the unsharded scan and the oracle."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, D, K, NQ = 200_003, 128, 10, 6


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import quiver_amd
        from quiver_amd.sharded import DeviceShard, ShardedFlatSearch, shard_bounds
        from tests import _oracle as O
        torch.cuda.set_device(0)
        base, n_local = shard_bounds(N, world, rank)
        idx = quiver_amd.DeviceIndex(D, "cosine", device=0)
        idx.add_synthetic(20260424, base, n_local)
        if rank == 1:
            idx.remove([5, 6])                        # tombstones in one shard (global rows base+5, base+6)
        s = ShardedFlatSearch(DeviceShard(idx), base, K, torch.device("cuda", 0), world=world)
        qs = [torch.from_numpy(q).cuda() for q in O.gen_rows(20260425, 0, NQ, D)]
        res = s.search_stream(qs)
        br, bd = s.search_batch(torch.stack(qs))       # the batch form: one all-gather, one merge launch
        torch.cuda.synchronize()
        for i in range(len(qs)):
            assert torch.equal(br[i], res[i][0]) and torch.equal(bd[i].view(torch.int32), res[i][1].view(torch.int32))
        ret[rank] = (np.stack([r.cpu().numpy().view(np.uint32) for r, _ in res]), np.stack([d.cpu().numpy() for _, d in res]), base)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_two_ranks_one_gpu_equal_unsharded(world):
    import torch.multiprocessing as mp
    from tests import _oracle as O
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    corpus = O.gen_rows(20260424, 0, N, D)
    qs = O.gen_rows(20260425, 0, NQ, D)
    alive = np.ones(N, bool)
    b1 = ret[1][2]
    alive[[b1 + 5, b1 + 6]] = False
    for rank in range(world):
        rr, dd, _ = ret[rank]
        for i in range(NQ):
            er, ed = O.exact_search(0, corpus, qs[i], K, alive=alive)
            assert np.array_equal(rr[i], er), (world, rank, i)
            assert np.array_equal(dd[i].view(np.uint32), ed.view(np.uint32))


def _worker_big_batch(rank, world, port, n, d, nq, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import quiver_amd
        from quiver_amd.sharded import DeviceShard, ShardedFlatSearch, shard_bounds
        from tests import _oracle as O
        torch.cuda.set_device(0)
        base, n_local = shard_bounds(n, world, rank)
        idx = quiver_amd.DeviceIndex(d, "cosine", device=0)
        idx.add_synthetic(20260424, base, n_local)
        s = ShardedFlatSearch(DeviceShard(idx), base, K, torch.device("cuda", 0), world=world)
        qs = torch.from_numpy(O.gen_rows(20260425, 0, nq, d)).cuda()
        br, bd = s.search_batch(qs)                      # 40 queries x 300k rows per shard (>= 8M query-rows): the fp32-MFMA path per shard
        torch.cuda.synchronize()
        ret[rank] = (br.cpu().numpy().view(np.uint32), bd.cpu().numpy())
    finally:
        dist.destroy_process_group()


def test_sharded_big_batch_goes_through_the_mfma_path_and_equals_the_oracle():
    import torch.multiprocessing as mp
    from tests import _oracle as O
    n, d, nq, world = 600_000, 64, 40, 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_big_batch, args=(world, _free_port(), n, d, nq, ret), nprocs=world, join=True)
    corpus = O.gen_rows(20260424, 0, n, d)
    qs = O.gen_rows(20260425, 0, nq, d)
    for rank in range(world):
        rr, dd = ret[rank]
        for i in range(0, nq, 3):
            er, ed = O.exact_search(0, corpus, qs[i], K)
            assert np.array_equal(rr[i], er), (rank, i)
            assert np.array_equal(dd[i].view(np.uint32), ed.view(np.uint32))
