"""CPU, world_size 2, gloo: the multi-GPU driver logic (pair sharding, weight broadcast, gather
of per-rank disparities, metric reduction).  The per-rank compute is replaced by the oracle on a
tiny op so that the N>1 path is exercised without a GPU."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_pairs, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import semstereo_amd.dist as sd
    from oracle import ops as oops
    r, w, _ = sd.init_from_env("gloo")
    assert (r, w) == (rank, world)
    # identical "weights" after broadcast
    lin = torch.nn.Linear(4, 4)
    torch.manual_seed(100 + rank)
    with torch.no_grad():
        lin.weight.normal_()
    sd.broadcast_module(lin, src=0)
    ref_w = torch.empty(4, 4)
    torch.manual_seed(100)                      # rank 0's draw, reproduced on every rank
    ref_w.normal_()
    # the full batch is defined by closed form on every rank; each rank computes its shard only
    g = torch.Generator().manual_seed(7)
    prob = torch.softmax(torch.randn(n_pairs, 8, 5, 6, generator=g), dim=1)
    (mine,) = sd.shard_batch([prob], rank, world)
    local = oops.disparity_regression(mine, 4)                       # stands in for the HIP forward
    full = sd.gather_batch(local, n_pairs)
    pairs, err, pix, tmax = sd.reduce_metrics(mine.shape[0], float(rank + 1), local.numel(), 0.5 + rank, "cpu")
    ok = torch.allclose(full, oops.disparity_regression(prob, 4)) and full.shape[0] == n_pairs
    ret[rank] = (ok, pairs, err, pix, tmax, torch.equal(lin.weight.detach(), ref_w),       # every rank holds rank 0's weights
                 float(lin.weight.detach().sum()))
    dist.barrier()
    dist.destroy_process_group()


def _run(n_pairs):
    world, port = 2, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, n_pairs, ret), nprocs=world, join=True)
    return dict(ret)


def test_two_ranks_shard_gather_reduce_even_and_ragged():
    for n_pairs in (4, 5):                     # 5: ragged split 3 + 2, padded all_gather
        ret = _run(n_pairs)
        assert all(v[0] for v in ret.values())
        assert ret[0][1] == n_pairs and ret[1][1] == n_pairs            # SUM of pairs
        assert ret[0][2] == 3.0 and ret[0][4] == 1.5                    # SUM of errors, MAX of time
        assert ret[0][5] and ret[1][5]                                  # both ranks hold exactly rank 0's weights
        assert abs(ret[0][6] - ret[1][6]) < 1e-6
