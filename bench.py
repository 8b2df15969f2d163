#!/usr/bin/env python
"""bench.py -- graphs/s of one GINet training step (topology + forward + MSE + backward
+ gradient all-reduce + Adam) on synthetic residue-level interface graphs, batch 64 per GPU.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N > 1 is launched by torch.distributed.run, one rank per GPU (RCCL); weak scaling: every
  rank owns 64 graphs (ids 64r .. 64r+63).  Rank 0 prints ONE JSON line.

What a "step" is: everything the reference does per mini-batch inside NeuralNet._epoch
(NeuralNet.py:489-506) for data already on the device: zero_grad, model(batch) -- including
the per-batch index work the reference redoes every forward (cluster offsets,
consecutive_cluster, pool_edge; here: the topology kernel) -- MSE loss, backward, optimizer
step.  The step is captured once in a hipGraph and replayed (``--mode eager`` runs the same
Python step without capture).

Extra objects in the JSON line:
  roofline      HBM roofline of the dominant kernel: algorithmic bytes per launch
                (SURVEY.md §8(d) per-graph figures x 64 graphs) / its average duration measured
                here with HIP events over back-to-back launches on the launch stream.
  cpu_baseline  the CPU oracle (oracle/cpu_ref.py: the reference's algorithm, op for op, in
                PyTorch) timed on this box's host cores on a bounded sample, rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch
import torch.nn.functional as F

GRAPHS_PER_GPU = 64
N_FEAT = 32
# algorithmic HBM bytes per graph (derivation table): GINet
BYTES_FWD, BYTES_BWD = 67668, 90208
BYTES_TOPO = 4804 + 1800 + 5808 + 16 * 1000        # read edge_index (int64 [2,E]) + clusters, write CSR0 + pooled CSR
PMC_FILE = "r01_bench_native_v12_pmc.json"
SQ_FILE = "r01_bench_native_v12_sq.json"
HBM_PEAK_GBS = 8000.0                               # MI355X_MICROARCH.md: 8 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--mode", choices=["native", "native-eager", "graph", "eager"], default="native",
                    help="native: FusedTrainer step (our head/loss/Adam kernels) replayed from a hipGraph; "
                         "graph/eager: torch autograd + torch.optim.Adam around the fused body")
    ap.add_argument("--net", choices=["GINet", "sGAT", "FoutNet"], default="GINet")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="native mode: build each step's topology with its own launch at the start of the step "
                         "instead of inside the previous step's backward launch (double-buffered workspaces)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for "
                    "exercising the multi-process path on a single GPU)")
    ap.add_argument("--force-dp-path", action="store_true",
                    help="run the data-parallel code path (gradient graph, eager all-reduce, Adam graph) even "
                         "with one process -- for testing on a single GPU")
    ap.add_argument("--steps-per-replay", type=int, default=20,
                    help="training steps captured per hipGraph replay (pipelined native mode)")
    ap.add_argument("--graphs-per-gpu", type=int, default=GRAPHS_PER_GPU,
                    help="mini-batch per GPU; %d is BASELINE.json's configuration, other values are for the "
                         "batch-size sweep in DESIGN.md" % GRAPHS_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--epoch-graphs", type=int, default=4096,
                    help="size of the resident graph set of the secondary whole-epoch measurement (0: skip it)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def main():
    global GRAPHS_PER_GPU
    args = parse()
    # RCCL writes its version banner to STDOUT (NCCL_DEBUG unset or =VERSION, as this image exports it); stdout
    # carries the ONE json line.  Other NCCL_DEBUG levels are left as the user set them.
    if os.environ.get("NCCL_DEBUG", "VERSION").upper() == "VERSION":
        os.environ["NCCL_DEBUG"] = "NONE"
    GRAPHS_PER_GPU = args.graphs_per_gpu
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    import torch.distributed as dist
    n_dev = torch.cuda.device_count()
    dev_index = local_rank if args.backend == "nccl" else local_rank % max(n_dev, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)
    elif args.force_dp_path and args.backend == "nccl":
        # one-rank RCCL group: the DP schedule then issues a REAL (identity) all-reduce launch per step
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                                device_id=dev)

    import deeprank_gnn_amd.synthetic as synth
    from deeprank_gnn_amd import _lib
    from deeprank_gnn_amd.topology import Topology
    from deeprank_gnn_amd.parallel import FlatGradBucket
    from deeprank_gnn_amd.ginet import GINet
    from deeprank_gnn_amd.sGAT import sGAT
    from deeprank_gnn_amd.foutnet import FoutNet
    Net = {"GINet": GINet, "sGAT": sGAT, "FoutNet": FoutNet}[args.net]

    batch_cpu = synth.make_batch(rank * GRAPHS_PER_GPU, GRAPHS_PER_GPU)
    batch = batch_cpu.clone().to(dev)
    torch.manual_seed(0)
    net = Net(N_FEAT, 1, 1).to(dev)            # dropout stays 0.4 for GINet (training mode)
    net.train()
    native = args.mode.startswith("native")
    capture = args.mode in ("native", "graph")
    need_w = args.net == "sGAT"
    pipeline = native and not args.no_pipeline
    dp_path = world > 1 or args.force_dp_path
    state = {"k": 0}       # which of the two topology workspaces the next step trains from
    if native:
        from deeprank_gnn_amd.trainer import FusedTrainer
        trainer = FusedTrainer(net, lr=1e-3, task="reg", seed=1234 + rank)
        loss_out = trainer.loss
        # Two persistent topology workspaces.  Pipelined: while step t trains out of one, the
        # topology of step t+1 is built into the other INSIDE step t's backward launch (the builder
        # only depends on index tensors).  Every step still builds one topology and consumes one.
        topos = [Topology.from_batch(batch, need_weights=need_w),
                 Topology.from_batch(batch, need_weights=need_w)]

        def one_step(fn):
            k = state["k"]
            if pipeline:
                fn(batch, topo=topos[k], next_topo=topos[1 - k])
                state["k"] = 1 - k
            else:
                fn(batch, topo=topos[0].rebuild())

        if not dp_path:
            def grad_part():                    # fwd, bwd(+head+loss [+next topology]), reduce+Adam
                one_step(trainer.train_step)

            def all_reduce():
                pass

            def update_part():
                pass
        else:
            def grad_part():
                one_step(trainer.compute_gradients)

            def all_reduce():
                if world > 1:
                    trainer.all_reduce_gradients()
                elif dist.is_initialized():
                    dist.all_reduce(trainer.flat_g)
                elif args.force_dp_path:
                    trainer.flat_g.mul_(1.0)

            def update_part():
                trainer.apply_update()
    else:
        opt = torch.optim.Adam(net.parameters(), lr=1e-3, fused=True, capturable=capture)
        bucket = FlatGradBucket(net.parameters())
        loss_out = torch.zeros((), device=dev)

        def grad_part():
            bucket.zero()
            topo = Topology.from_batch(batch, need_weights=need_w)
            pred = net(batch, topo=topo)
            loss = F.mse_loss(pred.reshape(-1), batch.y)
            loss.backward()
            loss_out.copy_(loss.detach())

        def all_reduce():
            bucket.all_reduce()

        def update_part():
            opt.step()

    split = dp_path or (world > 1)            # eager collective between the gradient and the update part
    two_flavours = native and pipeline        # steps alternate between the two topology workspaces

    def eager_step():
        grad_part()
        all_reduce()
        update_part()

    pre_steps = 0          # training steps already run while validating recorded graphs; counted as warm-up
    dp_mode = "eager all-reduce between %s" % ("graph replays" if capture else "eager launches")
    if capture:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(4):                # even: the workspace parity is back to 0 afterwards
                eager_step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()

        def graph_of(fn, **kw):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, **kw):
                fn()
            return g

        state["k"] = 0
        if split:
            # Data parallel: the all-reduce (RCCL) stays an eager call between graph replays.  One replay per
            # step: the Adam launch of step t is recorded in FRONT of step t+1's gradient launches
            # ([Adam(t); grad(t+1)]), the first step of a run replays [grad] alone and the run ends with a
            # lone [Adam] -- n gradient passes, n all-reduces, n Adam updates per run_steps(n), n+1 replays.
            nfl = 2 if two_flavours else 1

            def adam_then_grad():
                update_part()
                grad_part()
            g_first, g_next = [], []
            for k in range(nfl):
                state["k"] = k
                g_first.append(graph_of(grad_part))
                state["k"] = k
                g_next.append(graph_of(adam_then_grad))
            g_upd = graph_of(update_part)
            state["k"] = 0

            def run_steps(n):
                for i in range(n):
                    k = state["k"] if two_flavours else 0
                    (g_first if i == 0 else g_next)[k].replay()
                    all_reduce()
                    if two_flavours:
                        state["k"] ^= 1
                if n > 0:
                    g_upd.replay()

            # With RCCL the collective is recorded INSIDE the hipGraph as well: one replay per DP_CHUNK whole
            # steps [grad; all-reduce; Adam] instead of one replay + one eager collective call per step (the
            # eager call costs ~18 us of host time per step, more than the collective itself for a 60 KB
            # buffer).  RCCL's watchdog thread polls events while we record, hence the thread-local capture
            # mode.  A failure while recording falls back to the scheme above; a first replay that does not
            # complete within 60 s aborts the run with a message instead of hanging.  DRGNN_DP_GRAPH=0 opts out.
            use_dp_graph = (dist.is_initialized() and dist.get_backend() == "nccl"
                            and os.environ.get("DRGNN_DP_GRAPH", "1") != "0")
            if use_dp_graph:
                try:
                    DP_CHUNK = 2 * max(1, args.steps_per_replay // 2)

                    def dp_chunk():
                        for _ in range(DP_CHUNK):
                            grad_part()
                            all_reduce()
                            update_part()

                    def dp_one():
                        grad_part()
                        all_reduce()
                        update_part()
                    state["k"] = 0
                    g_dp = graph_of(dp_chunk, capture_error_mode="thread_local")
                    g_dp1 = []
                    for k in range(nfl):
                        state["k"] = k
                        g_dp1.append(graph_of(dp_one, capture_error_mode="thread_local"))
                    state["k"] = 0
                    done = torch.cuda.Event()
                    for k in range(2):                       # an even count: parity back to 0 afterwards
                        g_dp1[k % nfl].replay()
                    done.record()
                    t_lim = time.time() + 60.0
                    while not done.query():
                        if time.time() > t_lim:
                            sys.stderr.write("rank %d: graph-recorded all-reduce did not complete; "
                                             "rerun with DRGNN_DP_GRAPH=0\n" % rank)
                            sys.stderr.flush()
                            os._exit(3)
                        time.sleep(0.001)
                    pre_steps = 2
                    dp_mode = "hipGraph of %d x [grad; RCCL all-reduce; Adam]" % DP_CHUNK

                    def run_steps(n):            # noqa: F811
                        if n > 0 and two_flavours and state["k"] == 1:
                            g_dp1[1].replay()
                            state["k"] = 0
                            n -= 1
                        for _ in range(n // DP_CHUNK):
                            g_dp.replay()
                        n %= DP_CHUNK
                        while n > 0:
                            k = state["k"] if two_flavours else 0
                            g_dp1[k].replay()
                            if two_flavours:
                                state["k"] ^= 1
                            n -= 1
                except Exception as exc:       # pragma: no cover - depends on the collective library
                    sys.stderr.write("rank %d: recording the collective failed (%s); eager all-reduce\n" % (rank, exc))
                    state["k"] = 0
        else:
            def whole_step():
                grad_part()
                update_part()

            if two_flavours:
                # one replay = CHUNK steps (an even number: the workspace parity returns to 0); the
                # remainder of any --steps / --warmup runs through single-step graphs of either parity
                CHUNK = 2 * max(1, args.steps_per_replay // 2)

                def chunk():
                    for _ in range(CHUNK):
                        whole_step()
                g_chunk = graph_of(chunk)                       # parity 0 -> 0
                g_one = [graph_of(whole_step), graph_of(whole_step)]   # parity 0 -> 1, then 1 -> 0
                state["k"] = 0

                def run_steps(n):
                    if n > 0 and state["k"] == 1:
                        g_one[1].replay()
                        state["k"] = 0
                        n -= 1
                    for _ in range(n // CHUNK):
                        g_chunk.replay()
                    n %= CHUNK
                    while n > 0:
                        g_one[state["k"]].replay()
                        state["k"] ^= 1
                        n -= 1
            else:
                g_step = graph_of(whole_step)

                def run_steps(n):
                    for _ in range(n):
                        g_step.replay()
    else:
        def run_steps(n):
            for _ in range(n):
                eager_step()

    run_steps(max(args.warmup - pre_steps, 0))
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(args.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    final_loss = float(loss_out.item())
    in_sync = None
    if world > 1:
        # data parallel invariant: every rank applied the same averaged gradients, so the parameters agree bit
        # for bit (checked on a checksum and on the extreme values, max == min over ranks)
        flat = torch.cat([q.detach().reshape(-1).double() for q in net.parameters()])
        probe = torch.stack([flat.sum(), flat.abs().max(), flat.min()])
        hi, lo = probe.clone(), probe.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        in_sync = bool(torch.equal(hi, lo))

    result = None
    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = GRAPHS_PER_GPU * world * args.steps / elapsed
        result = {
            "metric": "interface-graphs/sec (fwd+bwd) %s batch=64 per GPU" % args.net,
            "value": value, "unit": "graphs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s train step (topology + body fwd + FC head/MSE + bwd + grad all-reduce + Adam) on SYN graphs: "
                                   "200 nodes, ~1000 directed edges, 32 node feats, 50->16 clusters "
                                   "(BASELINE.json configs[1])" % args.net,
                       "graphs_per_gpu": GRAPHS_PER_GPU, "global_batch": GRAPHS_PER_GPU * world,
                       "parallelism": "dp%d" % world, "mode": args.mode,
                       "topology": ("rebuilt every step; the build of step t+1 shares step t's backward launch "
                                    "(double-buffered)" if pipeline else "rebuilt every step, own launch"),
                       "final_loss": final_loss},
        }
        if split:
            result["config"]["dp_exchange"] = dp_mode
            result["config"]["params_in_sync"] = in_sync
        if args.net == "GINet":
            result["roofline"] = measure_roofline(net, batch, dev, value)
        if world == 1 and native and args.epoch_graphs > 0:
            try:
                result["epoch_loop"] = measure_epoch_loop(Net, args.net, args.epoch_graphs, dev)
            except Exception as exc:                      # secondary figure: never lose the bench line over it
                result["epoch_loop"] = {"error": repr(exc)[:200]}
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(args.net, batch_cpu, args.cpu_seconds)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    elif dist.is_initialized():
        dist.destroy_process_group()
    if rank == 0:
        try:                                  # anything a native library left in C stdio goes out BEFORE the line
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(result), flush=True)


def measure_roofline(net, batch, dev, graphs_per_s, iters=200):
    """Average duration of each launch of the native step, measured with HIP events around
    `iters` back-to-back launches (replayed from a hipGraph on torch's current stream = the stream
    the kernels are launched on), and the dominant one against the HBM roofline.  Algorithmic bytes: SURVEY.md §8(d)
    per-graph figures x 64 graphs."""
    import copy
    from deeprank_gnn_amd import _lib
    from deeprank_gnn_amd.topology import Topology
    from deeprank_gnn_amd.trainer import FusedTrainer
    tr = FusedTrainer(copy.deepcopy(net), lr=1e-3, task="reg", seed=99)
    topo = Topology.from_batch(batch, need_weights=False)
    nxt = Topology.from_batch(batch, need_weights=False, build=False)
    assert tr._can_fuse(topo, batch.x.shape[1]), "SYN graphs must take the fused-step path"
    c = tr._fused_prepare(batch, topo)
    B = c["B"]

    def k_topo():
        topo.rebuild()

    def k_step_co():
        tr._fused_launch_step(c, nxt)

    def k_step():
        tr._fused_launch_step(c, None)

    def k_update():
        tr._fused_launch_update(c, True, lr=0.0)

    def run(fn):
        c["stream"] = _lib.current_stream(c["x"])      # the capture stream while capturing
        fn()

    upd_bytes = (c["partials"].numel() + c["hp"].numel() + c["readout"].numel() + 7 * tr.flat_p.numel()) * 4 / B
    # SURVEY figures of forward + backward (the fused launch moves less: xp/arg0/arg1 stay in LDS)
    step_bytes = BYTES_FWD - 5808 + BYTES_BWD
    out = {}
    for name, fn, nbytes in (
            ("k_step_co_topo<GINet> (fwd + head/loss + bwd, + topology of the next batch)", k_step_co,
             step_bytes + BYTES_TOPO),
            ("k_update (partials reduction + Adam)", k_update, upd_bytes),
            ("k_topo (own launch; not on the pipelined path)", k_topo, BYTES_TOPO),
            ("k_step_co_topo<GINet> without the co-launched topology (not on the pipelined path)", k_step,
             step_bytes)):
        # 20 back-to-back launches per hipGraph replay: the device-side duration, not the host's launch rate
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            run(fn)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(20):
                run(fn)
        for _ in range(3):
            gr.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(iters // 20):
            gr.replay()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (20 * (iters // 20))
        out[name] = {"avg_us": us, "alg_bytes_per_launch": nbytes * B,
                     "achieved_GBs": nbytes * B / (us * 1e-6) / 1e9}
    on_path = list(out)[:2]
    dom = max(on_path, key=lambda k: out[k]["avg_us"])
    ach = out[dom]["achieved_GBs"]
    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", PMC_FILE)
    if os.path.exists(pmc_path):          # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of this same command (offline passes)
        pmc = json.load(open(pmc_path))
        for k, v in pmc.items():
            if "k_step_co_topo" in k and "k_step_co_topo" in dom:
                traffic = v["hbm_bytes_per_launch"]
    mfma = None
    sq_path = os.path.join(ROOT, "profiles", SQ_FILE)
    if os.path.exists(sq_path):           # rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES of this same command (offline pass)
        for k, v in json.load(open(sq_path)).items():
            if "k_step_co_topo" in k and "k_step_co_topo" in dom:
                # busy cycles summed over SIMDs / (1024 SIMDs x kernel cycles at the measured 2.28 GHz shader clock)
                mfma = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (out[dom]["avg_us"] * 1e3 * 2.28 * 1024.0)
    return {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "mfma_util": mfma,
            "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
            "traffic_note": "bytes/launch from rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE (separate passes, profiles/"
                            "%s); FETCH doubled per MI355X_MICROARCH.md" % PMC_FILE,
            "whole_step_frac": graphs_per_s * (BYTES_FWD + BYTES_BWD) / 1e9 / HBM_PEAK_GBS,
            "kernels": out}


def measure_epoch_loop(Net, net_name, n_graphs, dev, epochs=3):
    """Secondary figure (not `value`): whole shuffled epochs over a graph set resident in HBM, driven by the native
    loop drgnn_train_epoch -- every mini-batch is a different random selection of graphs, read in place by the topology
    builder; includes the shuffle, the id upload, the outputs' copy back and one synchronisation per epoch."""
    import deeprank_gnn_amd.synthetic as synth
    from deeprank_gnn_amd.resident import ResidentGraphSet
    from deeprank_gnn_amd.trainer import FusedTrainer
    graphs = [synth.make_graph(GRAPHS_PER_GPU + i) for i in range(n_graphs)]
    torch.manual_seed(0)
    tr = FusedTrainer(Net(N_FEAT, 1, 1).to(dev), lr=1e-3, task="reg")
    rs = ResidentGraphSet(graphs, dev)
    gen = torch.Generator().manual_seed(0)

    def epoch():
        order = torch.randperm(n_graphs, generator=gen).tolist()
        done = tr.train_epoch(rs, order, GRAPHS_PER_GPU)
        if done is None:
            raise RuntimeError("the native epoch loop refused this configuration")
        losses, pred = done
        pred.cpu()
        return float(losses.sum())
    epoch()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sums = [epoch() for _ in range(epochs)]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nb = (n_graphs + GRAPHS_PER_GPU - 1) // GRAPHS_PER_GPU
    return {"graphs_per_s": n_graphs * epochs / dt, "us_per_batch": dt / (epochs * nb) * 1e6, "epochs": epochs,
            "resident_graphs": n_graphs, "batch": GRAPHS_PER_GPU, "net": net_name, "last_epoch_loss_sum": sums[-1],
            "what": "shuffled epochs via drgnn_train_epoch (native loop, mini-batches read in place from the resident set)"}


def cpu_baseline(net_name, batch_cpu, seconds):
    """The CPU oracle (reference algorithm restated op for op) on this box's host cores."""
    from oracle import cpu_ref
    params = {k: v.clone().requires_grad_(True) for k, v in cpu_ref.init_params(net_name, N_FEAT, 1, 1).items()}
    opt = torch.optim.Adam(list(params.values()), lr=1e-3)
    kw = {"dropout": 0.4, "training": True} if net_name == "GINet" else {}

    def step():
        opt.zero_grad()
        pred = cpu_ref.FORWARD[net_name](params, batch_cpu, **kw)
        loss = F.mse_loss(pred.reshape(-1), batch_cpu.y)
        loss.backward()
        opt.step()

    # these are small ops: more threads is not faster.  Try a few counts briefly, keep the best.
    host = os.cpu_count() or 1
    best_threads, best_t = 1, None
    for nt in sorted({1, 4, 8, 16, 32, min(64, host)}):
        if nt > host:
            continue
        torch.set_num_threads(nt)
        step()
        t0 = time.perf_counter()
        step()
        step()
        dt = (time.perf_counter() - t0) / 2
        if best_t is None or dt < best_t:
            best_threads, best_t = nt, dt
        if dt > 2.0:
            break
    torch.set_num_threads(best_threads)
    for _ in range(2):
        step()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        step()
        n += 1
    dt = time.perf_counter() - t0
    return {"value": GRAPHS_PER_GPU * n / dt, "unit": "graphs/s", "cores": torch.get_num_threads(),
            "kind": "port", "host_cores": host,
            "sample": "%d train steps of the same 64-graph batch in %.1f s (oracle/cpu_ref.py, torch %s CPU "
            "kernels, best of 1/4/8/16/32/64 threads)" % (n, dt, torch.__version__)}


if __name__ == "__main__":
    main()
