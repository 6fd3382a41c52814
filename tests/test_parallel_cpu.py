"""CPU, world_size 2 over gloo: the data-parallel pieces of the N>1 path (scene sharding, parameter
broadcast, single flat gradient all-reduce).  The HIP operators are per-scene and need no
collective; what is covered here is everything the multi-GPU bench adds around them."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmp):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from pointcloud_bridge_amd import parallel
    r, w, _ = parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)

    torch.manual_seed(100 + rank)  # deliberately different init per rank
    model = nn.Sequential(nn.Linear(6, 8), nn.BatchNorm1d(8), nn.ReLU(), nn.Linear(8, 3))
    parallel.broadcast_parameters(model)
    ref = [p.detach().clone() for p in model.parameters()]
    gathered = [torch.zeros_like(ref[0]) for _ in range(world)]
    dist.all_gather(gathered, ref[0])
    assert all(torch.equal(g, gathered[0]) for g in gathered)  # every rank holds rank 0's weights

    # global batch of 8 "scenes", sharded; mean gradient over shards == full-batch gradient / 1
    g = torch.Generator().manual_seed(5)
    X, Y = torch.randn(8, 6, generator=g), torch.randn(8, 3, generator=g)
    mine = list(parallel.shard_scenes(8, rank, world))
    assert len(mine) == 4 and mine[0] == rank * 4
    model.eval()  # no batch statistics: the sharded mean must equal the single-process gradient
    bucket = parallel.FlatGradAllReduce(model.parameters())
    bucket.zero()
    ((model(X[mine]) - Y[mine]) ** 2).mean().backward()
    bucket.reduce()
    got = [p.grad.clone() for p in model.parameters()]

    full = nn.Sequential(nn.Linear(6, 8), nn.BatchNorm1d(8), nn.ReLU(), nn.Linear(8, 3)).eval()
    full.load_state_dict(model.state_dict())
    ((full(X) - Y) ** 2).mean().backward()
    for a, p in zip(got, full.parameters()):
        torch.testing.assert_close(a, p.grad, rtol=1e-5, atol=1e-6)
    # after the reduce every gradient is a view into the one flat, averaged buffer
    lo, hi = bucket.flat.data_ptr(), bucket.flat.data_ptr() + bucket.flat.numel() * 4
    assert all(lo <= p.grad.data_ptr() < hi for p in model.parameters())
    # the bench's combination: gradients packed into the flat buffer (no view re-assignment), averaged,
    # and ONE flat fused Adam step on it -- against torch.optim.Adam on the full batch in one process
    model.zero_grad(set_to_none=True)
    params = [p for p in model.parameters()]
    bucket2 = parallel.FlatGradAllReduce(params, assign_views=False)
    opt = parallel.FlatAdam(params, lr=1e-2, betas=(0.9, 0.999), weight_decay=1e-4)
    full2 = nn.Sequential(nn.Linear(6, 8), nn.BatchNorm1d(8), nn.ReLU(), nn.Linear(8, 3)).eval()
    full2.load_state_dict(model.state_dict())
    ref_opt = torch.optim.Adam(full2.parameters(), lr=1e-2, betas=(0.9, 0.999), weight_decay=1e-4)
    for _ in range(3):
        bucket2.zero()
        ((model(X[mine]) - Y[mine]) ** 2).mean().backward()
        bucket2.reduce()
        opt.step(bucket2.flat)
        ref_opt.zero_grad()
        ((full2(X) - Y) ** 2).mean().backward()
        ref_opt.step()
    for a, b in zip(model.parameters(), full2.parameters()):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6)
    # the bench's overlapped form: gradients land in the flat buffer through hooks, every bucket's
    # all-reduce starts when its last gradient has arrived (during the backward pass); a parameter
    # without a gradient counts as zero.  Must give the same flat gradient as the one-shot reduce.
    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.a = nn.Linear(6, 8)
            self.unused = nn.Linear(4, 4)     # never called: no gradient on any rank
            self.b = nn.Sequential(nn.ReLU(), nn.Linear(8, 3))

        def forward(self, x):
            return self.b(self.a(x))

    torch.manual_seed(3)
    net = Net()
    parallel.broadcast_parameters(net)
    over = parallel.OverlappedGradAllReduce.by_children(net)
    assert len(over.ranges) == 3
    for step in range(2):
        over.zero()
        ((net(X[mine]) - Y[mine]) ** 2).mean().backward()
        flat = over.finish().clone()
        assert all(p.grad is None for p in net.parameters())   # moved into the flat buffer, bucket by bucket
    over.close()
    one = parallel.FlatGradAllReduce(net.parameters(), assign_views=False)
    one.zero()
    ((net(X[mine]) - Y[mine]) ** 2).mean().backward()
    one.reduce()
    torch.testing.assert_close(flat, one.flat, rtol=1e-6, atol=1e-7)
    assert float(flat[over.ranges[1][0]:over.ranges[1][1]].abs().max()) == 0.0   # the unused layer's slice
    assert over.sent_order == [2, 1, 0]   # last bucket first, whatever order the buckets completed in

    # a branch that receives a gradient on ONE rank only.  Its bucket completes during backward on that
    # rank and only in finish() on the other; the collectives must still be issued in one order everywhere (the
    # all-reduces have different sizes: a mismatch pairs slices of different buckets or hangs).
    class Branchy(nn.Module):
        def __init__(self):
            super().__init__()
            self.a = nn.Linear(6, 8)
            self.side = nn.Linear(8, 8)
            self.b = nn.Linear(8, 3)

        def forward(self, x, use_side):
            h = torch.relu(self.a(x))
            if use_side:
                h = h + self.side(h)
            return self.b(h)

    torch.manual_seed(4)
    br = Branchy()
    parallel.broadcast_parameters(br)
    ob = parallel.OverlappedGradAllReduce.by_children(br)
    ob.zero()
    ((br(X[mine], rank == 0) - Y[mine]) ** 2).mean().backward()
    got = ob.finish().clone()
    assert ob.sent_order == [2, 1, 0]
    ob.close()
    # reference: both ranks' gradients computed here, missing ones as zeros, averaged
    want = torch.zeros_like(got)
    for r in range(world):
        br.zero_grad(set_to_none=True)
        rows = list(parallel.shard_scenes(8, r, world))
        ((br(X[rows], r == 0) - Y[rows]) ** 2).mean().backward()
        want += torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in br.parameters()]) / world
    torch.testing.assert_close(got, want, rtol=1e-6, atol=1e-7)
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(tmp, f"ok{rank}"), "w").write("ok")


def test_flat_grad_allreduce_world2_gloo(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


def test_shard_scenes_covers_every_scene_once():
    from pointcloud_bridge_amd import parallel
    for n in (16, 17, 5, 1):
        for w in (1, 2, 3, 8):
            seen = [i for r in range(w) for i in parallel.shard_scenes(n, r, w)]
            assert seen == list(range(n))
