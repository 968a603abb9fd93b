"""World-size-2 tests of the N>1 path on CPU (gloo): graph-id sharding and the single flat
gradient all-reduce.  The HIP forward itself needs a GPU, so the per-shard forward here is the
CPU oracle acting as the checker of the sharding logic (shards must reproduce the full batch)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from graphnet_classifier_amd import synthetic as S
        from graphnet_classifier_amd.GNN import CombinedModel, GraphNet
        from graphnet_classifier_amd.sharding import FlatGradAllReduce, shard_ranges
        from oracle import graphnet_oracle as O

        # ---- one flat all-reduce averages every gradient ------------------------------------
        torch.manual_seed(0)
        model = CombinedModel(GraphNet(**S.graphnet_kwargs(16, 2)), num_nodes=12, classes=2).to("cpu")
        for i, p in enumerate(model.parameters()):
            p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
        reducer = FlatGradAllReduce(model.parameters())
        flat = reducer()
        assert flat.numel() == sum(p.numel() for p in model.parameters())
        mean_scale = sum(r + 1 for r in range(world)) / world
        for i, p in enumerate(model.parameters()):
            assert torch.allclose(p.grad, torch.full_like(p, mean_scale * (i + 1)))
        # identical Adam update on every rank afterwards
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        opt.step()
        digest = torch.stack([p.detach().double().sum() for p in model.parameters()]).sum().reshape(1)
        gathered = [torch.zeros_like(digest) for _ in range(world)]
        dist.all_gather(gathered, digest)
        assert all(torch.equal(g, gathered[0]) for g in gathered)

        # ---- graph-id sharding reproduces the unsharded forward --------------------------------
        batch = S.superpixel_like_graphs(7, seed=1000)
        sd = {k: v.detach() for k, v in model.graph_net.state_dict().items()}
        g0, g1 = shard_ranges(batch.edge_ptr, world)[rank]
        shard = batch.slice_graphs(g0, g1)
        y_local = O.graphnet_forward(sd, shard.x, shard.pos, shard.edge_index)
        sizes = [torch.zeros(1, dtype=torch.long) for _ in range(world)]
        dist.all_gather(sizes, torch.tensor([y_local.size(0)]))
        assert sum(int(s) for s in sizes) == batch.num_nodes
        parts = [torch.zeros(int(s), 1) for s in sizes]
        dist.all_gather(parts, y_local) if len({int(s) for s in sizes}) == 1 else None
        if rank == 0:
            y_full = O.graphnet_forward(sd, batch.x, batch.pos, batch.edge_index)
            n0 = int(batch.graph_ptr[g0])
            assert torch.allclose(y_full[n0:n0 + shard.num_nodes], y_local, atol=2e-6)
        # edges aggregated by all ranks == edges of the whole batch (what bench.py sums)
        tot = torch.tensor([float(shard.num_edges)])
        dist.all_reduce(tot)
        assert int(tot) == batch.num_edges
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_sharding_and_grad_allreduce(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))
