#!/usr/bin/env python3
"""Contract benchmark of the GraphNet forward hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2|c5] [--scaling strong|weak]

One "step" = one pass of the hot path over one synthetic batch that is already resident in
HBM: destination-CSR build for the batch's edge_index (topology cache cleared, so it is paid
every step) + GraphNet.forward (edge features, encoders, L GN blocks with the fused
gather/concat/MLP/LayerNorm/residual kernel and the CSR scatter-sum, decoder).

Default workload = BASELINE.json configs[2] "c3": 1M nodes / 10M edges, all widths 64, 2 GN
blocks -- the configuration the metric's HBM-roofline target is quoted on.  For N > 1 the
driver starts one process per GPU (torch.distributed.run) and the default is BASELINE.json
configs[3] "c4": the SAME c3 batch split by graph id into N contiguous ranges balanced by edges
("strong": total work fixed; `--scaling weak` gives every rank its own full-size batch instead).
Graphs are independent, so the timed forward has no data-path collective.

After the timed region the same JSON line gets a `train` object: a few training steps (forward + CE
loss + backward on the K8 kernels + ONE flat-gradient all-reduce over RCCL when N > 1 + fused Adam,
reference utils/train_model.py:37-42 with the collective inserted at :41-42), with the all-reduce timed
separately (`allreduce_ms`).  `--mode train` makes that step the timed metric instead.

Prints ONE JSON line on rank 0.  value = edges aggregated per second over the whole job
(= sum over ranks of E_rank * n_blocks * steps / max-over-ranks wall time).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from graphnet_classifier_amd import native, synthetic  # noqa: E402
from graphnet_classifier_amd.GNN import GraphNet  # noqa: E402
from graphnet_classifier_amd.sharding import shard_ranges  # noqa: E402
from graphnet_classifier_amd import topology  # noqa: E402
from graphnet_classifier_amd.topology import clear_topology_cache  # noqa: E402

HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense fp32 MFMA peak


def usable_cpus() -> int:
    """Host cores this process may really use: the affinity mask, capped by the cgroup CPU quota
    (a GPU box hands a 1-GPU job a share of the host, not all of its hardware threads)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]))))
            else:
                quota = int(parts[0])
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    period = int(f.read())
                if quota > 0:
                    n = min(n, max(1, quota // period))
            break
        except (OSError, ValueError, IndexError):
            continue
    env = os.environ.get("GNC_CPU_THREADS")
    return int(env) if env else n  # the cgroup's CPU share (16 on a 1-GPU box of this pool)


def cpu_baseline(batch, kw, n_blocks, target_seconds=12.0):
    """Times the CPU oracle (kind 'port': the repo's restatement of the reference forward, using
    the reference's own index_add_ scatter) on a bounded sample of the same workload."""
    from oracle import graphnet_oracle as O
    O.set_scatter_impl("index_add")
    threads = usable_cpus()
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    ref_model = GraphNet(**kw)  # same seed -> same weights as the GPU model
    sd = {k: v.detach().cpu() for k, v in ref_model.state_dict().items()}
    del ref_model

    def run(ngraphs):
        s = batch.slice_graphs(0, ngraphs)
        t0 = time.perf_counter()
        with torch.no_grad():
            y = O.graphnet_forward(sd, s.x, s.pos, s.edge_index)
        return time.perf_counter() - t0, s, y

    probe = max(1, batch.num_graphs // 100)
    run(probe)  # warm-up (thread pools, allocator)
    t_probe, s, _ = run(probe)
    rate = s.num_edges / max(t_probe, 1e-6)
    ngraphs = int(min(batch.num_graphs, max(probe, target_seconds * rate / (batch.num_edges / batch.num_graphs))))
    t, s, y = run(ngraphs)
    O.set_scatter_impl("sorted_loop")
    return {"value": s.num_edges * n_blocks / t, "unit": "edges aggregated/s", "cores": torch.get_num_threads(),
            "kind": "port", "sample": f"first {ngraphs} of {batch.num_graphs} graphs of the same batch "
            f"({s.num_nodes} nodes, {s.num_edges} edges), one forward, {t:.2f} s",
            "graphs_per_s": ngraphs / t}, (s, y), sd


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--preheat-ms", type=float, default=600.0, help="untimed device pre-heat before the warm-up steps")
    ap.add_argument("--workload", default="c3", choices=sorted(synthetic.WORKLOADS))
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong"],
                    help="strong (default): one workload split by graph id over the ranks (c3 over N GPUs = BASELINE config c4); "
                         "weak: every rank its own full-size batch")
    ap.add_argument("--validation", default="deferred", choices=["deferred", "sync"],
                    help="how the step reports an out-of-range edge_index: deferred (default) = the flags the topology build "
                         "computes stay on the device and are read once after the timed region (no host sync inside a step, the "
                         "host enqueues ahead of the GPU); sync = read back in every step, as the module API does by default")
    ap.add_argument("--train-steps", type=int, default=5,
                    help="training steps of the extra `train` leg after the timed forward region (0 = skip it)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--scale", type=float, default=1.0,
                    help="fraction of the workload's graphs (tests / rehearsals only; the contract number is --scale 1)")
    ap.add_argument("--graph", default="auto", choices=["auto", "on", "off"],
                    help="forward mode: replay the step (CSR build + forward) from ONE hipGraph captured after the warm-up instead "
                         "of enqueueing its ~40 launches from Python every step.  auto = on for N > 1 (a 1/8 shard of c3 is 1.3 ms "
                         "of kernels behind 1.5 ms of host enqueue: tools/host_profile.py), off at N = 1 (GPU-bound either way, and "
                         "the per-launch HIP events of the roofline objects then live inside the timed region)")
    ap.add_argument("--train-graph", default="auto", choices=["auto", "on", "off"],
                    help="training step: replay forward + CE + backward + gradient pack from ONE hipGraph (train.CapturedTrainStep) "
                         "and issue the all-reduce and the fused Adam launch directly after it (world > 1; at world 1 the Adam "
                         "launch is inside the graph).  auto = on for N > 1 (a 1/8 shard's training step is launch-bound)")
    ap.add_argument("--mode", default="forward", choices=["forward", "train"],
                    help="train: the timed step is forward + cross-entropy + backward (HIP K8 kernels) + one flat gradient "
                         "all-reduce (RCCL, world > 1) + fused Adam, graphs batched block-diagonally")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            sys.exit(f"--gpus {a.gpus} needs one process per GPU: launch with python -m torch.distributed.run "
                     f"--nnodes=1 --nproc-per-node {a.gpus} --master-addr 127.0.0.1 bench.py --gpus {a.gpus} ...")
    ndev = torch.cuda.device_count()
    dev_index = local_rank % max(ndev, 1)  # one rank per GPU; the modulo only matters for the gloo rehearsal below
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    backend = os.environ.get("GNC_BENCH_BACKEND", "nccl")  # "gloo": rehearse the N > 1 bookkeeping on a 1-GPU box
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # RCCL over xGMI
        else:
            dist.init_process_group(backend)

    native.load_library()
    topology.set_validation(a.validation)
    w = synthetic.WORKLOADS[a.workload]
    n_blocks = w["n_blocks"]
    if a.scaling == "weak":
        # every rank owns a full-size batch (different seed per rank: different graphs)
        synthetic.WORKLOADS[a.workload]["seed"] = w["seed"] + 100 * rank
        batch, kw = synthetic.make_workload(a.workload, a.scale)
    else:
        full, kw = synthetic.make_workload(a.workload, a.scale)
        g0, g1 = shard_ranges(full.edge_ptr, world)[rank]
        batch = full.slice_graphs(g0, g1)
    torch.manual_seed(0)  # identical weights on every rank
    model = GraphNet(**kw).to(dev)
    model.eval()
    x, pos, ei = batch.x.to(dev), batch.pos.to(dev), batch.edge_index.to(dev)

    global_graphs = batch.num_graphs
    if world > 1:
        gg = torch.tensor([batch.num_graphs], dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(gg)
        global_graphs = int(gg.item())

    def make_train_step():
        """utils/train_model.py:37-42 on the block-diagonal batch of this rank: forward (read-out per graph), CE loss,
        backward, ONE flat-gradient all-reduce (world > 1), fused Adam.  Ranks own unequal graph counts (the split
        balances edges), so every rank scales its SUM-reduced loss by 1 / global graph count and the all-reduce is a
        plain SUM: the result is exactly the gradient of the global-batch mean loss."""
        from graphnet_classifier_amd.GNN import CombinedModel
        from graphnet_classifier_amd.train import FlatParameters, FusedAdam
        sizes = (batch.graph_ptr[1:] - batch.graph_ptr[:-1])
        equal = bool((sizes == sizes[0]).all())
        num_nodes = int(sizes.max())
        torch.manual_seed(0)
        cmodel = CombinedModel(GraphNet(**kw), num_nodes=num_nodes, classes=2).to(dev)
        cmodel.train()
        labels = torch.randint(0, 2, (batch.num_graphs,), generator=torch.Generator().manual_seed(rank)).to(dev)
        flat = FlatParameters(cmodel, average=False)
        opt = FusedAdam(flat, lr=1e-3)                            # utils/train_model.py:9
        crit = torch.nn.CrossEntropyLoss(reduction="sum")         # :10, mean taken over the GLOBAL batch below
        gptr = None if equal else batch.graph_ptr.to(dev)
        ar = {"events": [], "launch": "eager launches from Python"}

        def batched(mod, xx, pp, ee):
            clear_topology_cache()  # the CSR build belongs to the step (captured with it)
            return mod.forward_batched(xx, pp, ee, batch.num_graphs) if equal else mod.forward_batched(xx, pp, ee, graph_ptr=gptr)

        if a.train_graph == "on" or (a.train_graph == "auto" and world > 1):
            from graphnet_classifier_amd.train import CapturedTrainStep
            try:
                loss_sum = torch.zeros((), dtype=torch.float64, device=dev)
                # thread_local: the process group's watchdog thread polls its events meanwhile
                cap = CapturedTrainStep(cmodel, opt, crit, (x, pos, ei), labels, loss_sum, forward=batched,
                                        loss_scale=1.0 / global_graphs, capture_error_mode="thread_local")

                def tstep_replay():
                    cap.graph.replay()
                    if cap.collective_outside:
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        flat.reducer.allreduce()   # ONE all-reduce of the packed flat buffer (RCCL)
                        e1.record()
                        ar["events"].append((e0, e1))
                        opt.step(reduce=False)     # one fused Adam launch over the flat buffers
                    return None
                ar["launch"] = ("one hipGraph replay per step (CSR build + forward + CE + backward + gradient pack"
                                + ("), then the all-reduce and the fused Adam launch" if cap.collective_outside else " + fused Adam)"))
                return tstep_replay, flat, ar
            except Exception as ex:  # noqa: BLE001 - whatever the runtime refuses: measure the eager step instead
                ar["launch"] = f"eager launches from Python (capture failed: {ex})"
                torch.cuda.synchronize()

        def tstep():
            clear_topology_cache()
            if equal:
                logits = cmodel.forward_batched(x, pos, ei, batch.num_graphs)
            else:
                logits = cmodel.forward_batched(x, pos, ei, graph_ptr=gptr)
            loss = crit(logits, labels) / global_graphs
            opt.zero_grad()
            loss.backward()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            flat.reducer()          # gradient pack + ONE all-reduce of the flat buffer (RCCL; nothing to reduce at world 1)
            e1.record()
            ar["events"].append((e0, e1))
            opt.step(reduce=False)  # one fused Adam launch over the flat buffers
            return logits
        return tstep, flat, ar

    train_ctx = None
    if a.mode == "train":
        step, flat_params, ar_events = make_train_step()
        train_ctx = (flat_params, ar_events)
    else:
        def step():
            clear_topology_cache()  # the CSR build belongs to the step
            with torch.no_grad():
                return model(x, pos, ei)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # warm-up steps run exactly like timed ones (per-launch HIP events included), so that nothing - event pools,
    # allocator blocks, lazily loaded code objects - is created for the first time inside the timed region
    warm_timers = native.KernelTimers()
    native.set_kernel_timers(warm_timers)
    # device pre-heat (untimed, before the W warm-up steps): clocks and caches settle, and the HIP runtime's
    # event / signal pools grow here instead of inside the K timed steps
    if a.preheat_ms > 0:
        t_heat = time.perf_counter()
        y = step()
        torch.cuda.synchronize()
        one = max((time.perf_counter() - t_heat) * 1e3, 1e-3)
        n_heat = torch.tensor([min(200, int(a.preheat_ms / one) + 1)], dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
        if world > 1:  # the training step contains a collective: every rank must run the same number of steps
            dist.all_reduce(n_heat, op=dist.ReduceOp.MAX)
        for _ in range(int(n_heat.item())):
            y = step()
        torch.cuda.synchronize()
    for _ in range(a.warmup):
        y = step()
        torch.cuda.Event(enable_timing=True).record()
    # Inside the timed region HIP events bracket ONLY the dominant fused-MLP launch (roofline_mlp) - an event pair costs
    # 5-10 us of GPU time per launch, 0.2 ms per c3 step when every launch carries one (kernel trace of the step) - the
    # full per-kernel table comes from a few eager steps directly after the region.
    wsum = warm_timers.summary()
    wmlp = [k for k in wsum if k.startswith("mlp_fused")]
    dominant = max(wmlp, key=lambda k: wsum[k]["avg_ms"] * wsum[k]["launches"]) if wmlp else None
    timers = native.KernelTimers(only=dominant or "<no launch is timed inside the region>")
    all_timers = native.KernelTimers()
    # every HIP event of the timed region exists before it starts (see KernelTimers.reserve)
    steps_seen = max(1, a.warmup + (int(n_heat.item()) + 1 if a.preheat_ms > 0 else 0))
    per_step_launches = warm_timers.num_launches() / steps_seen
    timers.reserve(int((wsum[dominant]["launches"] / steps_seen if dominant else 0) * (a.steps + 1) * 1.1) + 64)
    all_timers.reserve(int(per_step_launches * 12 * 1.1) + 64)
    del warm_timers
    # hipGraph replay of the step (see --graph): captured once, after the warm-up; any failure falls back to eager launches
    graph, graph_out, graph_note = None, None, None
    if a.mode == "forward" and (a.graph == "on" or (a.graph == "auto" and world > 1)):
        native.set_kernel_timers(None)  # HIP events per launch cannot be recorded into a capture
        torch.cuda.synchronize()
        try:
            graph = torch.cuda.CUDAGraph()
            # thread_local: what OTHER threads do meanwhile (the process group's watchdog polls its events) must not
            # invalidate this capture; only this thread's own unsafe calls may
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                graph_out = step()
            graph.replay()
            torch.cuda.synchronize()
        except Exception as ex:  # noqa: BLE001 - whatever the runtime refuses: measure the eager step instead
            graph, graph_out, graph_note = None, None, f"capture failed, eager launches used: {ex}"
            torch.cuda.synchronize()
    run_step = step
    if graph is not None:
        def run_step():
            graph.replay()
            return graph_out
    else:
        native.set_kernel_timers(timers)
    fence()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]  # per-step spread (diagnostic only)
    for ev in marks:
        ev.record()  # allocate them now (re-recorded below)
    allocs0 = torch.cuda.memory_stats(dev).get("num_device_alloc", 0)  # hipMalloc calls of the caching allocator so far
    coll0 = train_ctx[0].reducer.collectives if train_ctx else 0
    t0 = time.perf_counter()
    marks[0].record()
    for k in range(a.steps):
        y = run_step()
        marks[k + 1].record()
    fence()
    elapsed = time.perf_counter() - t0
    coll_timed = (train_ctx[0].reducer.collectives - coll0) if train_ctx else 0
    ar_timed = list(train_ctx[1]["events"][-a.steps:]) if train_ctx else []
    # the per-kernel table: the same step, eagerly, with HIP events around every launch, right after the timed region
    ksteps = min(a.steps, 10)  # steps behind the per-kernel HIP events
    native.set_kernel_timers(all_timers)
    for _ in range(ksteps):
        step()
    fence()
    native.set_kernel_timers(timers)
    topology.check_deferred()  # the out-of-range flags of every timed step, read with one sync (raises IndexError if set)
    y_fwd = y
    device_allocs_timed = torch.cuda.memory_stats(dev).get("num_device_alloc", 0) - allocs0
    per_step_order = [marks[k].elapsed_time(marks[k + 1]) for k in range(a.steps)]
    per_step = sorted(per_step_order)
    # (not part of `value`) the same step with the topology cached: SURVEY 8d asks for the throughput with and
    # without the CSR build
    cached_ms = sync_ms = None
    if a.mode == "forward" and a.validation == "deferred":  # the same step with the flags read back in every step
        native.set_kernel_timers(None)
        topology.set_validation("sync")
        step()
        fence()
        t1 = time.perf_counter()
        for _ in range(a.steps):
            step()
        fence()
        sync_ms = (time.perf_counter() - t1) / a.steps * 1e3
        topology.set_validation("deferred")
        native.set_kernel_timers(timers)
    if a.mode == "forward":
        native.set_kernel_timers(None)
        with torch.no_grad():
            model(x, pos, ei)
        fence()
        t1 = time.perf_counter()
        for _ in range(a.steps):
            with torch.no_grad():
                model(x, pos, ei)
        fence()
        cached_ms = (time.perf_counter() - t1) / a.steps * 1e3
        native.set_kernel_timers(timers)
    # K1 is the kernel north_star grades.  When the step's aggregation ran inside the edge kernel's epilogue
    # (fused) no K1 launch exists in the timed region: time it in isolation as SURVEY 8d prescribes
    # (N(0,1) messages [E, D] in CSR order, the batch's own row pointers), HIP events on the launch stream.
    k1_isolated = False
    if not any(k.startswith("scatter_sum_csr") for k in all_timers.summary()):
        from graphnet_classifier_amd.topology import get_topology
        topo = get_topology(ei, batch.num_nodes, dev)
        msgs = torch.randn(batch.num_edges, w["width"], device=dev, generator=torch.Generator(device=dev).manual_seed(7))
        agg_buf = torch.empty(batch.num_nodes, w["width"], device=dev)
        native.set_kernel_timers(None)
        for _ in range(3):
            native.scatter_sum_csr(msgs, topo.rowptr, None, batch.num_nodes, out=agg_buf)
        native.set_kernel_timers(all_timers)
        for _ in range(20):
            native.scatter_sum_csr(msgs, topo.rowptr, None, batch.num_nodes, out=agg_buf)
        torch.cuda.synchronize()
        del msgs, agg_buf
        k1_isolated = True
    native.set_kernel_timers(None)
    ksum = all_timers.summary()
    for k, v in timers.summary().items():  # the dominant MLP launch as timed INSIDE the region; K1's isolated launches
        ksum[k] = v
    timed_inside = set(timers.summary())

    # ---- training leg (not part of `value` unless --mode train): BASELINE config c4's "RCCL grad all-reduce" lives here
    t_train, ar_ms, train_steps, collectives = 0.0, 0.0, 0, 0
    if a.mode == "train":
        flat_params, ar_events = train_ctx
        evs = ar_timed
        torch.cuda.synchronize()
        t_train, train_steps = elapsed, a.steps
        ar_ms = sum(e0.elapsed_time(e1) for e0, e1 in evs) / max(1, len(evs)) if evs else 0.0
        collectives = coll_timed
    elif a.train_steps > 0:
        tstep, flat_params, ar_events = make_train_step()
        for _ in range(2):
            tstep()
        fence()
        ar_events["events"].clear()
        c0 = flat_params.reducer.collectives
        tt0 = time.perf_counter()
        for _ in range(a.train_steps):
            tstep()
        fence()
        t_train, train_steps = time.perf_counter() - tt0, a.train_steps
        topology.check_deferred()
        ar_ms = sum(e0.elapsed_time(e1) for e0, e1 in ar_events["events"]) / train_steps if ar_events["events"] else 0.0
        collectives = flat_params.reducer.collectives - c0

    stats = torch.tensor([elapsed, t_train, ar_ms, float(batch.num_edges), float(batch.num_graphs), float(batch.num_nodes)],
                         dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if world > 1:
        tmax = stats[:3].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(stats[3:], op=dist.ReduceOp.SUM)
        stats[:3] = tmax
    elapsed, t_train, ar_ms, tot_edges, tot_graphs, tot_nodes = (float(v) for v in stats.cpu())

    if rank == 0:
        if world > 1 and a.scaling == "strong":
            wl_name = (f"{'c4' if a.workload == 'c3' else a.workload + '/' + str(world)}: {w['desc']}, split by graph id into {world} "
                       f"contiguous ranges balanced by edges, one rank per GPU" + (" + RCCL all-reduce of the flat gradient buffer"
                                                                                    if a.mode == "train" else ""))
        else:
            wl_name = f"{a.workload}: {w['desc']}" + (f", one such batch per rank x {world}" if world > 1 else "")
        k1 = ksum.get("scatter_sum_csr_sorted") or ksum.get("scatter_sum_csr_perm")
        k1_gbps = k1["avg_work"] / (k1["avg_ms"] * 1e-3) / 1e9
        # HBM bytes per launch from PMC counters (FETCH_SIZE doubled + WRITE_SIZE, separate passes) of an EARLIER profiled
        # run of this workload, condensed by tools/pmc_summary.py into profiles/pmc_<workload>.json
        traffic = mlp_traffic = None
        pmc_file = os.path.join("profiles", f"pmc_{a.workload}.json")
        if a.scale == 1.0 and world == 1 and os.path.exists(os.path.join(ROOT, pmc_file)):
            with open(os.path.join(ROOT, pmc_file)) as f:
                pmc = json.load(f)
            traffic = pmc["kernels"].get(pmc.get("k1", ""), {}).get("hbm_bytes_per_launch")
            mlp_traffic = pmc["kernels"].get(pmc.get("dominant_mlp", ""), {}).get("hbm_bytes_per_launch")
        mlp_names = [k for k in ksum if k.startswith("mlp_fused")]  # none when the timed step was a hipGraph replay
        mlp_name = max(mlp_names, key=lambda k: ksum[k]["avg_ms"] * ksum[k]["launches"]) if mlp_names else None
        mlp = ksum[mlp_name] if mlp_name else None
        mlp_tflops = mlp["avg_work"] / (mlp["avg_ms"] * 1e-3) / 1e12 if mlp else None
        result = {
            "metric": "edges aggregated/sec (GraphNet forward)" if a.mode == "forward" else
                      "edges aggregated/sec (GraphNet+classifier training step: fwd+bwd+allreduce+Adam)",
            "value": tot_edges * n_blocks * a.steps / elapsed,
            "unit": "edges/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": a.scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl_name + ("" if a.scale == 1.0 else f" [scaled to {a.scale} of the graphs]"), "graphs": int(tot_graphs), "nodes": int(tot_nodes),
                       "edges": int(tot_edges), "n_blocks": n_blocks, "width": w["width"],
                       "step": "CSR build + GraphNet.forward, inputs resident in HBM" if a.mode == "forward" else
                               "CSR build + forward + CE loss + backward + flat grad all-reduce + Adam, inputs resident in HBM"},
            "graphs_per_sec": tot_graphs * a.steps / elapsed,
            "roofline": {"kernel": "scatter_sum_csr_vec4 (K1 scatter-sum aggregation, CSR-ordered messages)",
                         "measured": ("isolated: 20 launches on N(0,1) messages [E, D] with the batch's row pointers, after the "
                                      "timed region (in the step the aggregation runs in the edge kernel's epilogue)")
                         if k1_isolated else "every K1 launch of the timed region",
                         "bound": "hbm", "achieved": k1_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": k1_gbps / HBM_PEAK_GBPS, "traffic": traffic,
                         "traffic_source": f"{pmc_file}: PMC counters of an earlier profiled run of this workload "
                                           "(not measured in this run)" if traffic is not None else None,
                         "avg_launch_ms": k1["avg_ms"], "launches": k1["launches"],
                         "algorithmic_bytes_per_launch": k1["avg_work"]},
            "roofline_mlp": None if mlp is None else {"kernel": f"mlp_fused_kernel ({mlp_name}: fused gather+concat+MLP+LayerNorm+residual)",
                             "bound": "mfma", "achieved": mlp_tflops, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": mlp_tflops / MFMA_F32_PEAK_TFLOPS, "avg_launch_ms": mlp["avg_ms"],
                             "launches": mlp["launches"], "executed_flops_per_launch": mlp["avg_work"],
                             # HBM side of the same launch: rows in (4 D E), rows out (4 D E), and for the W-split edge launch the
                             # two gathered projections priced as if every gathered row came from memory (2 x 4 D E; the tables
                             # are 4 D N each and mostly L2 / MALL resident), three int32 ids per edge and the aggregate (4 D N)
                             "algorithmic_bytes_per_launch": (4.0 * w["width"] * (4 * batch.num_edges + batch.num_nodes) + 12.0 * batch.num_edges)
                             if "+2add" in mlp_name else None,
                             "traffic": mlp_traffic if "+2add" in mlp_name else None,
                             "traffic_over_algorithmic": (mlp_traffic / (4.0 * w["width"] * (4 * batch.num_edges + batch.num_nodes) + 12.0 * batch.num_edges))
                             if (mlp_traffic and "+2add" in mlp_name) else None,
                             "traffic_source": (f"{pmc_file} (earlier profiled run, not this one)" if mlp_traffic else None),
                             # the reference's concat form of the same launch (W-split removes 2 of the 3 first-Linear blocks)
                             "reference_form_flops_per_launch": mlp["avg_work"] + (4.0 * batch.num_edges * w["width"] ** 2
                                                                                    if "+2add" in mlp_name else 0.0)},
            "step_ms_spread": {"min": per_step[0], "median": per_step[len(per_step) // 2], "max": per_step[-1],
                               "slowest_step": per_step_order.index(per_step[-1]),
                               "device_allocs_in_timed_region": device_allocs_timed},
            "kernel_ms_per_step": {k: v["avg_ms"] * v["launches"] / (a.steps if (k in timed_inside and graph is None) else ksteps)
                                   for k, v in ksum.items() if not (k1_isolated and k.startswith("scatter_sum_csr"))},
            "kernel_ms_per_step_source": f"HIP events around every launch of {ksteps} eager steps directly after the timed region "
                                         "(inside it only the dominant fused-MLP launch carries events)",
        }
        if mlp_names and a.mode == "forward":  # the whole step against the same roof: every MLP launch's executed FLOPs over the step's wall time
            step_flops = sum(all_timers.summary()[k]["avg_work"] * all_timers.summary()[k]["launches"] for k in mlp_names
                             if k in all_timers.summary()) / ksteps
            result["whole_step"] = {"executed_flops_per_step": step_flops, "achieved": step_flops / (elapsed / a.steps) / 1e12,
                                    "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                    "frac": step_flops / (elapsed / a.steps) / 1e12 / MFMA_F32_PEAK_TFLOPS,
                                    "note": "all fused-MLP launches of one step (this rank) / ms_per_step; includes the CSR build and "
                                            "every non-MFMA launch in the denominator"}
        if a.mode == "forward":
            result["step_launch"] = (f"one hipGraph replay per step (captured after the warm-up; per-kernel HIP events from {ksteps} eager "
                                     "steps after the timed region)") if graph is not None else \
                                    ("eager launches from Python" + (f" ({graph_note})" if graph_note else ""))
        if train_steps:
            result["train"] = {
                "step": "CSR build + forward + CE loss + backward + gradient pack + ONE flat all-reduce (world > 1) + fused Adam",
                "steps": train_steps, "ms_per_step": t_train / train_steps * 1e3,
                "value": tot_edges * n_blocks * train_steps / t_train, "unit": "edges/s",
                "graphs_per_sec": tot_graphs * train_steps / t_train,
                "allreduce_ms": ar_ms if world > 1 else 0.0, "gradient_pack_ms": (ar_ms if ar_events["events"] else None) if world == 1 else None,
                "allreduce_bytes": flat_params.grad.numel() * 4, "collectives_per_step": collectives / train_steps,
                "backend": ("rccl" if backend == "nccl" else backend) if world > 1 else None,
                "step_launch": ar_events["launch"]}
        result["config"]["topology_validation"] = ("deferred: the out-of-range flags of each step's CSR build stay on the device and are "
                                                    "read once after the timed region" if a.validation == "deferred" else
                                                    "sync: flags read back (one host sync) inside every step")
        if sync_ms is not None:
            result["validated_every_step"] = {"ms_per_step": sync_ms, "value": tot_edges * n_blocks / (sync_ms * 1e-3),
                                              "note": "same step with --validation sync: one host sync per step (rank 0's clock)"}
        if cached_ms is not None:
            result["topology_cached"] = {"ms_per_step": cached_ms, "value": tot_edges * n_blocks / (cached_ms * 1e-3),
                                         "note": "same step without the CSR build (rank 0's clock)"}
        if world == 1 and not a.no_cpu_baseline and a.mode == "forward":
            base, (s, yref), sd = cpu_baseline(batch, kw, n_blocks)
            result["cpu_baseline"] = base
            # parity in the same run: the timed sample (a prefix of the batch) AND the first / middle / last graphs, so that
            # rows at the far end of the edge tables (beyond 4 GiB at c2 / c5) are looked at too
            from oracle import graphnet_oracle as O
            y_host = y_fwd.cpu()
            err = float((y_host[: s.num_nodes] - yref).abs().max())
            ng = batch.num_graphs
            for g0 in sorted({0, max(0, ng // 2 - 1), max(0, ng - 3)}):
                sg = batch.slice_graphs(g0, min(ng, g0 + 3))
                with torch.no_grad():
                    ref = O.graphnet_forward(sd, sg.x, sg.pos, sg.edge_index)
                n0 = int(batch.graph_ptr[g0])
                err = max(err, float((y_host[n0:n0 + sg.num_nodes] - ref).abs().max()))
            result["parity_max_abs_vs_oracle"] = err
            result["parity_checked"] = "timed CPU sample (prefix) + first / middle / last 3 graphs of the batch"
            result["speedup_vs_cpu_baseline"] = result["value"] / base["value"]
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
