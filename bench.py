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

# DRGNN_BENCH_WATCHDOG=<seconds> (diagnosis): dump every thread's stack to stderr after that long and exit, instead of hanging
if os.environ.get("DRGNN_BENCH_WATCHDOG"):
    import faulthandler
    faulthandler.dump_traceback_later(float(os.environ["DRGNN_BENCH_WATCHDOG"]), exit=True)

import torch
import torch.nn.functional as F

GRAPHS_PER_GPU = 64
N_FEAT = 32
# ALGORITHMIC HBM bytes per graph, forward + backward (derivation table): the figure
# roofline.achieved / roofline.frac are computed from.  (fwd, bwd) per net.
ALG_BYTES = {"GINet": (67668, 90208), "sGAT": (62440, 49904), "FoutNet": (52840, 45104)}
# what the co-launched topology builder moves on top of that (NOT part of §8(d)'s figure; reported separately as
# roofline.with_builder): read edge_index (int64 [2,E]) + clusters, write CSR0 + pooled CSR (+ edge_attr for sGAT)
BYTES_TOPO = {"GINet": 4804 + 1800 + 5808 + 16 * 1000, "FoutNet": 4804 + 1800 + 5808 + 16 * 1000,
              "sGAT": 4804 + 1800 + 5808 + 16 * 1000 + 4000 + 4 * 200}
HBM_PEAK_GBS = 8000.0                               # MI355X_MICROARCH.md: 8 TB/s spec
# the K-step block is repeated until the timed region is this long: several SMI polling periods, so that an outside
# observer (the driver's gpu_busy sampler) sees the GPU busy DURING the measurement
MIN_TIMED_SECONDS = 7.0          # (sized from one untimed block, which runs a little slower than the timed ones: >= 6 s result)


def source_hash():
    """sha256 (16 hex digits) over the kernel sources: the key that ties a committed rocprofv3 counter summary
    (profiles/*_pmc.json, *_sq.json: collected offline, PMC passes cannot run inside this process) to the build it
    was taken from.  A summary of another build is NOT reported."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "deeprank-gnn_amd", "csrc")
    files = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".h", ".hip")))
    files.append(os.path.join(ROOT, "include", "drgnn.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def counter_summary(kind, net_name):
    """(values, note): the per-kernel counter averages of profiles/*_<kind>.json (kind = 'pmc' | 'sq') whose
    recorded source hash AND net match this build, else (None, reason)."""
    import glob
    want = source_hash()
    seen = []
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_%s.json" % kind)), reverse=True):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        meta = d.get("_meta") if isinstance(d, dict) else None
        if not meta:
            continue
        seen.append(os.path.basename(path))
        if meta.get("source_hash") == want and meta.get("net", "GINet") == net_name:
            return d, os.path.basename(path)
    return None, ("no counter summary under profiles/ was collected from this build (source hash %s, net %s; "
                  "collect with tools/collect_profiles.sh)" % (want, net_name))


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
    ap.add_argument("--topology", choices=["rebuilt", "cached"], default="rebuilt",
                    help="rebuilt (default, the headline): every step builds the topology of a mini-batch from its int64 "
                         "edge_index / cluster tensors, like the reference redoes its index work in every forward pass; "
                         "cached (declared second mode): per-graph topology built once at upload of the resident graph "
                         "set, a mini-batch is a list of graph numbers, no builder workgroups in the step")
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
    ap.add_argument("--step-layout", choices=["auto", "one", "two", "noclass", "af1"], default="auto",
                    help="GINet: workgroups per graph of the fused step -- auto (default): two while every workgroup of the "
                         "launch is resident, else one (both branches in sequence); one / two: forced, for A/B runs; noclass: "
                         "auto without the capacity-class kernels (compile-time LDS layout for batches inside 200 / 1024 / 52)")
    ap.add_argument("--dp-selftest", action="store_true",
                    help="data-parallel self-test BEFORE timing (dropout off for the whole run): 3 steps eagerly and 3 through "
                         "the recorded schedule on every rank, each checked for (a) the all-reduced gradient == rank 0's "
                         "recompute on the union of the shards (1e-5 relative, SURVEY 8(e)) and (b) parameters in sync "
                         "over the ranks; the result (ranks, backend, exchange, fallback taken) goes into config.dp_selftest "
                         "and to stderr.  Works with --gpus N (RCCL) and with --gpus N --backend gloo on one GPU")
    ap.add_argument("--n-feat", type=int, default=N_FEAT,
                    help="node features of the synthetic graphs (default 32 = BASELINE.json; 48 = the reference's shipped regression "
                         "models); the algorithmic bytes follow SURVEY 8(d)'s accounting: + 4 N bytes per feature and pass over x")
    ap.add_argument("--dp-distinct", action="store_true",
                    help="also time the same (data-parallel) schedule over a cycle of 32 DIFFERENT synthetic mini-batches per rank "
                         "(`dp_distinct` in the line): the figure without the L2 residency of a replayed mini-batch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dropin", action="store_true", help="skip the drop-in loop figures (dropin_loop)")
    ap.add_argument("--counter-pass", action="store_true",
                    help="for rocprofv3 --pmc passes (tools/collect_profiles.sh): skip the roofline leg that launches the step "
                         "kernel WITHOUT the co-launched builder, so that the per-kernel-name counter average is over one kind "
                         "of launch (the main loop's); with --topology cached the other way round")
    ap.add_argument("--dropin-only", action="store_true",
                    help="run ONLY the drop-in loop of --net at --graphs-per-gpu (profiling runs) and print its figures")
    ap.add_argument("--no-other-nets", action="store_true",
                    help="skip the secondary sGAT / FoutNet figures of the default GINet line (`other_nets`)")
    ap.add_argument("--epoch-graphs", type=int, default=4096,
                    help="size of the resident graph set of the secondary whole-epoch measurement (0: skip it)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--min-seconds", type=float, default=MIN_TIMED_SECONDS,
                    help="the K-step block is repeated until the timed region is at least this long (0: one block; "
                         "used by the rocprofv3 counter passes, which only need a few launches)")
    return ap.parse_args()


def main():
    global GRAPHS_PER_GPU, N_FEAT, COUNTER_PASS
    args = parse()
    COUNTER_PASS = bool(args.counter_pass)
    # RCCL writes its version banner to STDOUT (NCCL_DEBUG unset or =VERSION, as this image exports it); stdout
    # carries the ONE json line.  Other NCCL_DEBUG levels are left as the user set them.
    if os.environ.get("NCCL_DEBUG", "VERSION").upper() == "VERSION":
        os.environ["NCCL_DEBUG"] = "NONE"
    GRAPHS_PER_GPU = args.graphs_per_gpu
    N_FEAT = args.n_feat
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not launched by torch.distributed.run: spawn the N ranks ourselves (same command line) instead of
        # silently measuring one GPU
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
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
        if os.environ.get("TORCHELASTIC_USE_AGENT_STORE") and os.environ.get("MASTER_PORT"):
            # under torch.distributed.run the rendezvous store is the agent's: a private tcp:// address would be CONNECTED to,
            # not served, and never answer
            dist.init_process_group("nccl", device_id=dev)
        else:
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

    # same-box A/B runs of the whole program: the process default of the launch plan's overrides (drgnn_step_plan; the library
    # reads DRGNN_STEP_PLAN once, at its first plan query -- none has been made yet).  noclass: the run-time LDS layout also
    # where the capacity class applies; af1: sGAT / FoutNet with ONE workgroup per graph; one / two: workgroups per graph
    if args.step_layout != "auto":
        os.environ["DRGNN_STEP_PLAN"] = {"noclass": "noclass", "af1": "nosplit", "one": "one", "two": "two"}[args.step_layout]
    if args.dropin_only:
        print(json.dumps({"dropin_loop": measure_dropin_loop(dev, [(args.net, GRAPHS_PER_GPU)])}), flush=True)
        return
    batch_cpu = synth.make_batch(rank * GRAPHS_PER_GPU, GRAPHS_PER_GPU, n_feat=N_FEAT)
    batch = batch_cpu.clone().to(dev)
    torch.manual_seed(0)
    net = Net(N_FEAT, 1, 1).to(dev)            # dropout stays 0.4 for GINet (training mode)
    if args.dp_selftest and hasattr(net, "dropout"):
        net.dropout = 0.0                      # the union recompute cannot reproduce per-rank dropout streams
    net.train()
    native = args.mode.startswith("native")
    capture = args.mode in ("native", "graph")
    need_w = args.net == "sGAT"
    cached = native and args.topology == "cached"
    pipeline = native and not args.no_pipeline and not cached
    dp_path = world > 1 or args.force_dp_path
    state = {"k": 0}       # which of the two topology workspaces the next step trains from
    if native:
        from deeprank_gnn_amd.trainer import FusedTrainer
        trainer = FusedTrainer(net, lr=1e-3, task="reg", seed=1234 + rank)
        loss_out = trainer.loss
        oneshot = None
        if world > 1 and os.environ.get("DRGNN_DP_ONESHOT", "0") != "0":
            # OPT-IN (DRGNN_DP_ONESHOT=1): the one-shot peer-to-peer all-reduce (csrc/drgnn_p2p.h: one launch, one xGMI
            # round trip) instead of RCCL's ring for the 43 KB gradient -- adopted only if a verified trial exchange
            # succeeds on EVERY rank (bounded waits: no hang), else the run stays on RCCL.  Off by default: it has not
            # run on a multi-GPU node yet (none in the pool), RCCL recorded in the hipGraph is the measured path.
            oneshot = trainer.use_oneshot_allreduce(verify=True)
        # Two persistent topology workspaces.  Pipelined: while step t trains out of one, the
        # topology of step t+1 is built into the other INSIDE step t's backward launch (the builder
        # only depends on index tensors).  Every step still builds one topology and consumes one.
        cache = None
        if cached:
            from deeprank_gnn_amd.resident import ResidentGraphSet
            rs = ResidentGraphSet([synth.make_graph(rank * GRAPHS_PER_GPU + i, n_feat=N_FEAT) for i in range(GRAPHS_PER_GPU)], dev)
            cache = rs.topology_cache(need_weights=need_w)
            ids_host = list(range(GRAPHS_PER_GPU))
            ids_dev = rs.upload_ids(ids_host)
            topos = []
        else:
            topos = [Topology.from_batch(batch, need_weights=need_w),
                     Topology.from_batch(batch, need_weights=need_w)]

        def one_step(fn):
            k = state["k"]
            if cached:
                trainer.train_step_cached(cache, ids_host, ids_dev, apply_adam=(fn == trainer.train_step))
            elif pipeline:
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
                    DP_CHUNK = 2 * max(1, min(args.steps_per_replay, max(args.steps, 2)) // 2)

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
                CHUNK = 2 * max(1, min(args.steps_per_replay, max(args.steps, 2)) // 2)

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
                # one workspace (own-launch builder, or cached topology): CHUNK steps per replay + single steps
                CHUNK = max(1, min(args.steps_per_replay, args.steps))

                def chunk():
                    for _ in range(CHUNK):
                        whole_step()
                g_chunk = graph_of(chunk)
                g_step = graph_of(whole_step)

                def run_steps(n):
                    for _ in range(n // CHUNK):
                        g_chunk.replay()
                    for _ in range(n % CHUNK):
                        g_step.replay()
    else:
        def run_steps(n):
            for _ in range(n):
                eager_step()

    selftest = None
    if args.dp_selftest:
        if not native:
            raise SystemExit("--dp-selftest needs --mode native")
        selftest = dp_selftest(trainer, net, Net, dev, rank, world, eager_step, run_steps, state, two_flavours,
                               {"dp_exchange": dp_mode, "split_schedule": bool(split),
                                "backend": (dist.get_backend() if dist.is_initialized() else None),
                                "oneshot": oneshot is not None})
        if rank == 0:
            sys.stderr.write("dp_selftest: %s\n" % json.dumps(selftest))
            sys.stderr.flush()
        if not selftest["ok"]:
            if dist.is_initialized():
                dist.barrier()
                dist.destroy_process_group()
            raise SystemExit(4)
    # Warm-up: W steps as asked, rounded UP to an even count so that the two topology workspaces are back at
    # parity 0 and every timed block starts on the recorded chunk (a block of K steps = K // CHUNK chunk replays
    # + K % CHUNK single-step replays, whatever K and W are).
    warm = max(args.warmup - pre_steps, 0)
    if two_flavours_parity(state):
        warm += 1
    warm += warm & 1
    run_steps(warm)
    torch.cuda.synchronize()
    # One untimed block to size the timed region: the K-step block is repeated until the region is at least
    # MIN_TIMED_SECONDS long (a single 20-step block is ~0.5 ms: shorter than a host timer tick is accurate for
    # and invisible to an SMI poll).  Every rank uses the same repeat count.
    t0 = time.perf_counter()
    run_steps(args.steps)
    torch.cuda.synchronize()
    t_block = max(time.perf_counter() - t0, 1e-6)
    if args.steps & 1:
        run_steps(1)                          # parity back to 0
    repeats = int(min(max(1, -(-args.min_seconds // t_block)), 200000))
    if world > 1:
        t = torch.tensor([repeats], device=dev, dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        repeats = int(t.item())
    # HIP events only every `stride` blocks (>= 200 steps apart): an event between every two 20-step replays costs the device a
    # marker per replay (measured: 20.2 vs 19.8 us per step under --steps 20), and the median block time does not need them
    stride = max(1, 200 // max(args.steps, 1))
    n_ev = (repeats + stride - 1) // stride
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n_ev + 1)]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev[0].record()
    spans = []
    for r in range(repeats):
        run_steps(args.steps)
        if (r + 1) % stride == 0 or r + 1 == repeats:
            ev[len(spans) + 1].record()
            spans.append((r + 1) - sum(spans))
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    blocks_ms = sorted(ev[i].elapsed_time(ev[i + 1]) / spans[i] for i in range(len(spans)))
    block_med_ms = blocks_ms[len(blocks_ms) // 2]
    if world > 1:
        t = torch.tensor([elapsed, block_med_ms], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, block_med_ms = float(t[0].item()), float(t[1].item())
    timed_steps = args.steps * repeats
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

    dp_distinct = None
    if args.dp_distinct and native and not cached:
        try:
            dp_distinct = measure_dp_distinct(trainer, dev, rank, world, need_w, all_reduce, update_part, dp_path,
                                              dist.is_initialized() and dist.get_backend() == "nccl")
        except Exception as exc:                          # secondary figure
            dp_distinct = {"error": repr(exc)[:200]}
    result = None
    if rank == 0:
        # wall clock over `repeats` back-to-back blocks of K steps (barrier + synchronize on both sides, max over
        # ranks): K x repeats steps in `elapsed`
        ms = elapsed / timed_steps * 1e3
        value = GRAPHS_PER_GPU * world * timed_steps / elapsed
        result = {
            "metric": "interface-graphs/sec (fwd+bwd) %s batch=%d per GPU" % (args.net, GRAPHS_PER_GPU),
            "value": value, "unit": "graphs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "repeats": repeats, "timed_steps": timed_steps,
            "timed_seconds": elapsed, "block_ms_median": block_med_ms,
            "ms_per_step_median_block": block_med_ms / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s train step (topology + body fwd + FC head/MSE + bwd + grad all-reduce + Adam) on SYN graphs: "
                                   "200 nodes, ~1000 directed edges, %d node feats, 50->16 clusters "
                                   "(BASELINE.json configs[1]%s)" % (args.net, N_FEAT, "" if N_FEAT == 32 else " but for the feature count"),
                       "graphs_per_gpu": GRAPHS_PER_GPU, "global_batch": GRAPHS_PER_GPU * world,
                       "parallelism": "dp%d" % world, "mode": args.mode, "step_layout": args.step_layout,
                       "topology": ("cached per graph (declared): built once at upload of the resident set, the step "
                                    "reads it in place, no builder workgroups" if cached else
                                    "rebuilt every step; the build of step t+1 shares step t's backward launch "
                                    "(double-buffered)" if pipeline else "rebuilt every step, own launch"),
                       "input": "ONE synthetic mini-batch replayed every step (its graphs stay in the XCDs' L2 between steps: "
                                "`distinct_batches` and `epoch_loop` in this line are the figures without that)",
                       "final_loss": final_loss, "host_pool_threads": torch.get_num_threads()},
        }
        if split:
            if native and oneshot is not None:
                dp_mode = dp_mode.replace("RCCL all-reduce", "one-shot p2p all-reduce (drgnn_allreduce_oneshot)")
                dp_mode += " [opt-in DRGNN_DP_ONESHOT=1; verified trial exchange on every rank]"
                oneshot.check()
            result["config"]["dp_exchange"] = dp_mode
            result["config"]["params_in_sync"] = in_sync
            result["config"]["dp_backend"] = dist.get_backend() if dist.is_initialized() else None
            result["config"]["dp_ranks"] = dist.get_world_size() if dist.is_initialized() else 1
        if selftest is not None:
            result["config"]["dp_selftest"] = selftest
        if dp_distinct is not None:
            result["dp_distinct"] = dp_distinct
        if native:
            result["roofline"] = measure_roofline(net, args.net, batch, dev, value / world,
                                                  cache=(cache, ids_host, ids_dev) if cached else None)
        if world == 1 and native and args.epoch_graphs > 0:
            try:
                result["distinct_batches"] = measure_distinct_batches(Net, args.net, dev)
            except Exception as exc:                      # secondary figure: never lose the bench line over it
                result["distinct_batches"] = {"error": repr(exc)[:200]}
            try:
                result["epoch_loop"] = measure_epoch_loop(Net, args.net, args.epoch_graphs, dev)
            except Exception as exc:                      # secondary figure: never lose the bench line over it
                result["epoch_loop"] = {"error": repr(exc)[:200]}
            if not args.no_dropin:
                result["dropin_loop"] = measure_dropin_loop(dev, None if (args.net == "GINet" and GRAPHS_PER_GPU == 64)
                                                            else [(args.net, GRAPHS_PER_GPU)])
            try:
                result["neuralnet_train"] = measure_neuralnet_train(Net, args.net, dev)
            except Exception as exc:                      # secondary figure: never lose the bench line over it
                result["neuralnet_train"] = {"error": repr(exc)[:200]}
            try:
                result["inference_loop"] = measure_inference_loop(Net, args.net, args.epoch_graphs, dev)
            except Exception as exc:                      # secondary figure: never lose the bench line over it
                result["inference_loop"] = {"error": repr(exc)[:200]}
        if world == 1 and native and args.net == "GINet" and not args.no_other_nets and N_FEAT == 32:
            # BASELINE.json configs[2] / configs[3] in the same line (driver evidence for the single-branch nets)
            result["other_nets"] = {}
            for other in ("sGAT", "FoutNet"):
                try:
                    result["other_nets"][other] = measure_other_net(other, batch, dev)
                except Exception as exc:                  # secondary figure: never lose the bench line over it
                    result["other_nets"][other] = {"error": repr(exc)[:200]}
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


def measure_dp_distinct(trainer, dev, rank, world, need_w, all_reduce, update_part, dp_path, rccl, n_batches=32):
    """Secondary figure (not `value`): the run's own schedule -- [gradient launch (+ the next mini-batch's topology); all-reduce;
    Adam] data parallel, the fused step + update on one GPU -- over a cycle of `n_batches` DIFFERENT synthetic mini-batches per
    rank (every one with a topology workspace of its own, built by the previous step's launch), one hipGraph of `n_batches`
    steps where the collective can be recorded (RCCL / single rank), eager steps otherwise.  Max over the ranks."""
    import torch.distributed as dist
    import deeprank_gnn_amd.synthetic as synth
    from deeprank_gnn_amd.topology import Topology
    first = 100000 + rank * n_batches * GRAPHS_PER_GPU
    batches = [synth.make_batch(first + i * GRAPHS_PER_GPU, GRAPHS_PER_GPU, n_feat=N_FEAT).to(dev) for i in range(n_batches)]
    topos = [Topology.from_batch(b, need_weights=need_w, build=(i == 0)) for i, b in enumerate(batches)]
    n = n_batches

    def chunk():
        for k in range(n):
            if dp_path:
                trainer.compute_gradients(batches[k], topo=topos[k], next_topo=topos[(k + 1) % n])
                all_reduce()
                update_part()
            else:
                trainer.train_step(batches[k], topo=topos[k], next_topo=topos[(k + 1) % n])
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        chunk()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    run, how = chunk, "eager steps"
    if not dp_path or world == 1 or rccl:
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, **({"capture_error_mode": "thread_local"} if (dp_path and rccl) else {})):
                chunk()
            run, how = g.replay, "hipGraph of %d steps" % n
        except Exception:
            run, how = chunk, "eager steps (recording failed)"
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    reps = 60 if run is not chunk else 8
    t0 = time.perf_counter()
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    us = (time.perf_counter() - t0) / (reps * n) * 1e6
    if world > 1:
        t = torch.tensor([us], dtype=torch.float64, device=dev if rccl else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        us = float(t.item())
    return {"us_per_step": us, "graphs_per_s": GRAPHS_PER_GPU * world / (us * 1e-6), "distinct_batches_per_rank": n, "how": how}


def dp_selftest(trainer, net, Net, dev, rank, world, eager_step, run_steps, state, two_flavours, info):
    """3 eager steps + 3 steps through the recorded schedule; after every step the all-reduced gradient in
    ``trainer.flat_g`` is compared with rank 0's recompute on the UNION of the shards (same parameters: the snapshot taken
    before the step) and the parameters are compared over the ranks.  Collective-safe: every rank makes the same calls."""
    import copy
    import torch.distributed as dist
    import deeprank_gnn_amd.synthetic as synth
    from deeprank_gnn_amd.trainer import FusedTrainer
    multi = dist.is_initialized() and dist.get_world_size() > 1
    ref = None
    if rank == 0:
        ref = FusedTrainer(copy.deepcopy(net), lr=1e-3, task="reg", seed=1)
        union = synth.make_batch(0, GRAPHS_PER_GPU * world, n_feat=N_FEAT).to(dev)
    out = {"ranks": world, "rccl_ranks": (world if info["backend"] == "nccl" else 0), "backend": info["backend"],
           "dp_exchange": info["dp_exchange"] if info["split_schedule"] else "single process: reduce + Adam in one launch",
           "oneshot_allreduce": info["oneshot"], "graphs_per_rank": GRAPHS_PER_GPU, "steps": []}

    def check(kind, before):
        g = trainer.flat_g.detach().clone()
        want = torch.zeros_like(g)
        if rank == 0:
            ref.flat_p.copy_(before)
            ref.compute_gradients(union)
            want.copy_(ref.flat_g)
        if multi:
            dist.broadcast(want, src=0)
        scale = float(want.abs().max())
        err = float((g - want).abs().max()) / max(scale, 1e-30)
        flat = trainer.flat_p.detach().double()
        probe = torch.stack([flat.sum(), flat.abs().max(), flat.min(), torch.tensor(err, dtype=torch.float64, device=dev)])
        hi, lo = probe.clone(), probe.clone()
        if multi:
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        in_sync = bool(torch.equal(hi[:3], lo[:3]))
        worst = float(hi[3])
        out["steps"].append({"schedule": kind, "grad_max_rel_err": worst, "params_in_sync": in_sync})
        return in_sync and worst <= 1e-5

    ok = True
    for _ in range(3):
        before = trainer.flat_p.detach().clone()
        eager_step()
        torch.cuda.synchronize()
        ok = check("eager", before) and ok
    if state["k"] == 1 and two_flavours:           # 3 eager steps flipped the workspace parity: one more brings it back
        before = trainer.flat_p.detach().clone()
        eager_step()
        torch.cuda.synchronize()
        ok = check("eager", before) and ok
    for _ in range(4 if two_flavours else 3):      # (an even count keeps the recorded chunks' parity)
        before = trainer.flat_p.detach().clone()
        run_steps(1)
        torch.cuda.synchronize()
        ok = check("recorded", before) and ok
    out["ok"] = bool(ok)
    return out


def two_flavours_parity(state):
    return state["k"] == 1


COUNTER_PASS = False      # --counter-pass


def measure_roofline(net, net_name, batch, dev, graphs_per_s, iters=400, cache=None):
    """Average duration of each launch of the native step, measured LIVE with HIP events around `iters`
    back-to-back launches (20 per hipGraph replay, on torch's current stream = the stream the kernels are launched
    on), and the dominant one against the HBM roofline.

    roofline.achieved = ALGORITHMIC bytes per launch / that duration, algorithmic bytes = SURVEY.md §8(d)'s per-graph
    figure for forward + backward of this net (GINet 157 876 B, sGAT 112 344 B, FoutNet 97 944 B) x the graphs one
    launch processes.  The bytes the co-launched topology builder moves in the same launch are NOT part of §8(d)'s
    figure: the figure that includes them is reported separately (`with_builder`)."""
    import copy
    from deeprank_gnn_amd import _lib
    from deeprank_gnn_amd.topology import Topology
    from deeprank_gnn_amd.trainer import FusedTrainer
    need_w = net_name == "sGAT"
    tr = FusedTrainer(copy.deepcopy(net), lr=1e-3, task="reg", seed=99)
    topo = Topology.from_batch(batch, need_weights=need_w)
    nxt = Topology.from_batch(batch, need_weights=need_w, build=False)
    assert tr._can_fuse(topo, batch.x.shape[1]), "SYN graphs must take the fused-step path"
    variant = tr.api.step_is_specialised(tr.kind, batch.x, batch.x.shape[1], topo.max_nodes, topo.max_edges,
                                         topo.max_c0, tr.H, tr.O)
    c = tr._fused_prepare(batch, topo, True, nxt)
    B = c["B"]

    def k_topo():
        topo.rebuild()

    # the build the pipelined step launch actually carries (lean chains + aggregation tiles), as a launch of its own into a
    # third workspace
    lean_flags = tr._flags_for(nxt, batch.x.shape[1])
    lean_t = Topology.from_batch(batch, need_weights=need_w, build=False) if lean_flags & _lib.TOPO_LEAN else None

    def k_topo_lean():
        lean_t.rebuild(lean_flags)

    def k_step_co():
        tr._fused_launch_step(c, nxt)

    def k_step():
        tr._fused_launch_step(c, None)

    cc = tr._cached_prepare(cache[0], cache[1], cache[2]) if cache else None

    def k_step_cached():
        cc["stream"] = _lib.current_stream(c["x"])
        tr._cached_launch_step(cc, True)

    def k_update():
        tr._fused_launch_update(c, True, lr=0.0)

    def run(fn):
        c["stream"] = _lib.current_stream(c["x"])      # the capture stream while capturing
        fn()

    fwd_b, bwd_b = ALG_BYTES[net_name]
    # (SURVEY 8(d) counts x [N, F] once in the forward and once per branch in the backward: N = 200, 4-byte words)
    alg = fwd_b + bwd_b + (N_FEAT - 32) * 200 * 4 * (3 if net_name == "GINet" else 2)
    upd_bytes = (c["partials"].numel() + c["hp"].numel() + c["readout"].numel() + 7 * tr.flat_p.numel()) * 4 / B
    # which kernel the launch is: the plan the prepared step carries (the same decision procedure the launch runs)
    plan = c["plan"]
    wg_txt = "two workgroups per graph" if plan.wgs_per_graph == 2 else "one workgroup per graph"
    if plan.family == _lib.STEP_FAMILY_AGGREGATE and net_name != "GINet":
        kname = "k_step2_co_topo<%s,%d,%s>" % (net_name, plan.width, wg_txt)      # csrc/drgnn_step2.h
    elif plan.family == _lib.STEP_FAMILY_AGGREGATE:
        kname = "k_step3%s_co_topo<GINet,%d> (aggregation first, %s)" % ("" if plan.wgs_per_graph == 2 else "b", plan.width, wg_txt)
    else:
        raise RuntimeError("the benchmarked shape is stepped by the aggregation-first kernels; plan family %d" % plan.family)
    out = {}
    first = ((kname + " (fwd + head/loss + bwd, topology read from the per-graph cache)", k_step_cached, alg) if cache else
             (kname + " (fwd + head/loss + bwd, + topology of the next batch)", k_step_co, alg))
    for name, fn, nbytes in (
            first,
            ("k_update (partials reduction + Adam)", k_update, upd_bytes),
            ("k_topo (own launch; not on the pipelined path)", k_topo, BYTES_TOPO[net_name]),
            ("k_topo, lean chains + tiles: what the step launch co-builds (own launch here)", k_topo_lean, BYTES_TOPO[net_name]),
            (kname + " without the co-launched topology (not on the pipelined path)", k_step, alg)):
        if fn is k_topo_lean and lean_t is None:
            continue
        if COUNTER_PASS and (fn is k_step or (cache and fn is k_step_co)):
            continue
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
        # five timed segments, the median one reported: a single stall of the box inside one long region (seen: 9 ms in an 8 ms
        # region) once made k_update "the dominant kernel" of an sGAT line
        seg = max(1, (iters // 20) // 5)
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
        torch.cuda.synchronize()
        evs[0].record()
        for s_ in range(5):
            for _ in range(seg):
                gr.replay()
            evs[s_ + 1].record()
        torch.cuda.synchronize()
        us = sorted(evs[i].elapsed_time(evs[i + 1]) * 1e3 / (20 * seg) for i in range(5))[2]
        out[name] = {"avg_us": us, "alg_bytes_per_launch": nbytes * B,
                     "achieved_GBs": nbytes * B / (us * 1e-6) / 1e9}
    on_path = list(out)[:2]
    dom = max(on_path, key=lambda k: out[k]["avg_us"])
    ach = out[dom]["achieved_GBs"]
    step_us = out[on_path[0]]["avg_us"]
    extra = 0 if cache else BYTES_TOPO[net_name]
    with_builder = (alg + extra) * B / (step_us * 1e-6) / 1e9
    # HBM traffic and MFMA busy cycles come from rocprofv3 --pmc passes, which cannot run inside this process: they
    # are read from the summary under profiles/ that was collected from THIS build (matched by source hash and net),
    # and reported as null otherwise
    traffic, traffic_note, mfma, mfma_note = None, None, None, None
    traffic_cached = None     # the same counters of the launch WITHOUT a co-launched builder (cached topology), same summary

    def own_step_kernel(name):
        """is `name` (a kernel of a counter summary) the fused step kernel of THIS net?  (the default GINet run also launches
        sGAT's and FoutNet's for `other_nets`: their rows sit in the same summary)"""
        if "_co_topo" not in name or "k_step" not in name:
            return False
        kind = {"GINet": 0, "sGAT": 1, "FoutNet": 2}[net_name]
        if "k_step3" in name or "k_step1_co_topo" in name:
            return kind == 0
        return ("co_topo<%d," % kind) in name
    if "_co_topo" in dom:
        pmc, where = counter_summary("pmc", net_name)
        if pmc is None:
            traffic_note = where
        else:
            for k, v in pmc.items():
                if own_step_kernel(k):
                    traffic = v["hbm_bytes_per_launch"]
                    traffic_note = ("bytes/launch from rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE (separate passes, "
                                    "profiles/%s, same sources as this build); FETCH doubled per "
                                    "MI355X_MICROARCH.md; averaged over launches WITH the co-launched builder only "
                                    "(--counter-pass; rounds 4 - 5 averaged over both kinds of launch), "
                                    "traffic_cached_topology: the step launch alone" % where)
            for k, v in (pmc.get("_cached_topology") or {}).items():
                if own_step_kernel(k):
                    traffic_cached = v["hbm_bytes_per_launch"]
        sq, where = counter_summary("sq", net_name)
        if sq is None:
            mfma_note = where
        else:
            for k, v in sq.items():
                if own_step_kernel(k):
                    # busy cycles summed over SIMDs / (1024 SIMDs x kernel cycles at the 2.28 GHz shader clock)
                    mfma = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (step_us * 1e3 * 2.28 * 1024.0)
                    mfma_note = "SQ_VALU_MFMA_BUSY_CYCLES from profiles/%s (same sources as this build)" % where
    return {"bound": "hbm", "kernel": dom, "kernel_us": out[dom]["avg_us"], "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_note": traffic_note, "traffic_cached_topology": traffic_cached,
            "mfma_util": mfma, "mfma_note": mfma_note,
            "alg_bytes_per_graph": alg, "graphs_per_launch": B, "source_hash": source_hash(),
            "with_builder": {"achieved": with_builder, "frac": with_builder / HBM_PEAK_GBS,
                             "bytes_per_graph": alg + BYTES_TOPO[net_name],
                             "note": "adds what the co-launched topology builder of the same launch moves (int64 "
                                     "edge_index + clusters read, CSR0 + pooled CSR written); not SURVEY 8(d)'s figure"},
            "whole_step_frac": graphs_per_s * alg / 1e9 / HBM_PEAK_GBS,
            "kernels": out}


def measure_other_net(net_name, batch, dev, steps=20):
    """Secondary figure (not `value`): the SAME pipelined training step for another net of the path (BASELINE.json configs[2] /
    configs[3]: sGAT, FoutNet + community_pooling on the same synthetic graphs, batch 64) -- us per step of 20-step hipGraph
    replays (topology of step t+1 built inside step t's launch, update launch included) and the step launch against the HBM
    roofline by SURVEY 8(d)'s bytes of that net."""
    import time as _time
    from deeprank_gnn_amd.topology import Topology
    from deeprank_gnn_amd.trainer import FusedTrainer
    from deeprank_gnn_amd.sGAT import sGAT
    from deeprank_gnn_amd.foutnet import FoutNet
    Net = {"sGAT": sGAT, "FoutNet": FoutNet}[net_name]
    need_w = net_name == "sGAT"
    torch.manual_seed(0)
    net = Net(N_FEAT, 1, 1).to(dev)
    net.train()
    tr = FusedTrainer(net, lr=1e-3, task="reg", seed=1234)
    topos = [Topology.from_batch(batch, need_weights=need_w), Topology.from_batch(batch, need_weights=need_w)]

    def chunk():
        for k in range(steps):
            tr.train_step(batch, topo=topos[k & 1], next_topo=topos[1 - (k & 1)])
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        chunk()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        chunk()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = _time.perf_counter()
    reps = 0
    while _time.perf_counter() - t0 < 1.5:
        for _ in range(50):
            g.replay()
        reps += 50
        torch.cuda.synchronize()
    us = (_time.perf_counter() - t0) / (reps * steps) * 1e6
    rf = measure_roofline(net, net_name, batch, dev, GRAPHS_PER_GPU / (us * 1e-6), iters=200)
    return {"us_per_step": us, "graphs_per_s": GRAPHS_PER_GPU / (us * 1e-6), "kernel": rf["kernel"], "kernel_us": rf["kernel_us"],
            "frac": rf["frac"], "achieved": rf["achieved"], "alg_bytes_per_graph": rf["alg_bytes_per_graph"],
            "whole_step_frac": rf["whole_step_frac"], "final_loss": float(tr.loss.item())}


def measure_distinct_batches(Net, net_name, dev, n_batches=32, steps=32, graphs=None, batches=None):
    """Secondary figure (not `value`): the same pipelined step replayed from a hipGraph, but every step on a DIFFERENT
    host-collated synthetic mini-batch (a cycle of `n_batches`, every one with a topology workspace of its own built by the
    previous step's launch).  `value` replays ONE mini-batch, whose graphs stay in the L2 of the XCD that worked on them a step
    earlier; this is what that residency is worth."""
    import deeprank_gnn_amd.synthetic as synth
    from deeprank_gnn_amd.topology import Topology
    from deeprank_gnn_amd.trainer import FusedTrainer
    need_w = net_name == "sGAT"
    torch.manual_seed(0)
    tr = FusedTrainer(Net(N_FEAT, 1, 1).to(dev), lr=1e-3, task="reg")

    def timed(batches):
        n = len(batches)
        topos = [Topology.from_batch(b, need_weights=need_w, build=(i == 0)) for i, b in enumerate(batches)]

        def chunk():
            for k in range(steps):
                tr.train_step(batches[k % n], topo=topos[k % n], next_topo=topos[(k + 1) % n])
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            chunk()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            chunk()
        for _ in range(3):
            g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        reps = 200
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / (reps * steps)
    graphs = GRAPHS_PER_GPU if graphs is None else graphs
    if batches is not None:          # (the caller's cycle: only the distinct figure)
        us_distinct = timed(batches)
        return {"us_per_step": us_distinct, "graphs_per_s": graphs / (us_distinct * 1e-6), "distinct_batches": len(batches),
                "batch": graphs, "net": net_name}
    one = synth.make_batch(0, graphs, n_feat=N_FEAT).to(dev)
    us_same = timed([one, one])
    us_distinct = timed([synth.make_batch(graphs * (i + 1), graphs, n_feat=N_FEAT).to(dev) for i in range(n_batches)])
    return {"us_per_step_same_batch": us_same, "us_per_step": us_distinct, "graphs_per_s": graphs / (us_distinct * 1e-6),
            "distinct_batches": n_batches, "batch": graphs, "net": net_name,
            "what": "the pipelined step (topology of step t+1 built inside step t's launch) replayed from a hipGraph over a cycle "
                    "of %d different synthetic mini-batches, against the same replay of one mini-batch" % n_batches}


def measure_dropin_loop(dev, configs=None, n_batches=32):
    """Secondary figure (not `value`): the loop body an UNCHANGED reference trainer runs (reference NeuralNet.py:489-506)

        optimizer.zero_grad(); pred = model(batch); loss = MSELoss(pred.reshape(-1), y); loss.backward(); optimizer.step()

    with `model` this package's net, `optimizer` torch.optim.Adam over model.parameters() as NeuralNet constructs it
    (NeuralNet.py:183-184) and a cycle of `n_batches` different synthetic mini-batches -- the literal drop-in boundary
    (deeprank-gnn_amd/fused_autograd.py: model(batch) and loss.backward() are one launch each of the aggregation-first step
    kernels).  us per step, eager (host-bound) and recorded in a hipGraph (one graph over the cycle; Adam capturable=True, torch's
    default foreach implementation and fused=True), with the batch's topology workspace kept with the batch object (`kept`: what
    a DataLoader over pre-collated batches gives) and rebuilt by a builder launch in every call (`rebuilt`: fresh Batch objects
    every epoch, as the reference's DataLoader collates them); next to the native trainer's step over the same cycle."""
    import time as _time
    import deeprank_gnn_amd.synthetic as synth
    from deeprank_gnn_amd.fused_autograd import engine_for
    from deeprank_gnn_amd.ginet import GINet
    from deeprank_gnn_amd.sGAT import sGAT
    from deeprank_gnn_amd.foutnet import FoutNet
    nets = {"GINet": GINet, "sGAT": sGAT, "FoutNet": FoutNet}
    if configs is None:
        configs = [("GINet", 64), ("sGAT", 64), ("FoutNet", 64), ("GINet", 128)]
    out = {"what": "optimizer.zero_grad(); pred = model(batch); loss = mse(pred, y); loss.backward(); torch.optim.Adam.step() over a "
                   "cycle of %d distinct SYN mini-batches; us per step" % n_batches}
    for net_name, B in configs:
        key = net_name if B == 64 else "%s_b%d" % (net_name, B)
        try:
            batches = [synth.make_batch(B * (i + 1), B, n_feat=N_FEAT).to(dev) for i in range(n_batches)]
            res = {"batch": B, "distinct_batches": n_batches}

            def fresh(**adam):
                torch.manual_seed(0)
                net = nets[net_name](N_FEAT, 1, 1).to(dev)
                net.train()
                return net, torch.optim.Adam(net.parameters(), lr=1e-3, **adam)

            def body(net, opt, b):
                opt.zero_grad()
                pred = net(b)
                loss = F.mse_loss(pred.reshape(-1), b.y)
                loss.backward()
                opt.step()
                return loss

            # -- eager ---------------------------------------------------------------------------------------------------
            for label, keep in (("eager_kept_us", True), ("eager_rebuilt_us", False)):
                net, opt = fresh()
                eng = engine_for(net)
                eng.cache_topology = keep
                for b in batches:
                    body(net, opt, b)
                assert eng.last_path == "jacobian", eng.last_path
                torch.cuda.synchronize()
                t0 = _time.perf_counter()
                n = 0
                while _time.perf_counter() - t0 < 0.6:
                    for b in batches:
                        loss = body(net, opt, b)
                    n += len(batches)
                    torch.cuda.synchronize()
                res[label] = (_time.perf_counter() - t0) / n * 1e6
                res["final_loss_" + label[:-3]] = float(loss.item())
            res["plan"] = {"family": int(eng.last_plan.family), "wgs_per_graph": int(eng.last_plan.wgs_per_graph),
                           "width": int(eng.last_plan.width), "cls": int(eng.last_plan.cls)}
            # -- recorded ------------------------------------------------------------------------------------------------
            for label, keep, adam in (("graph_kept_us", True, {}), ("graph_rebuilt_us", False, {}),
                                      ("graph_kept_fused_adam_us", True, {"fused": True}),
                                      ("graph_rebuilt_fused_adam_us", False, {"fused": True})):
                net, opt = fresh(capturable=True, **adam)
                eng = engine_for(net)
                eng.cache_topology = keep
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for b in batches:       # (every batch once: with `kept` its workspace is built HERE, outside the recording)
                        body(net, opt, b)
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                opt.zero_grad(set_to_none=True)
                with torch.cuda.graph(g):
                    for b in batches:
                        loss = body(net, opt, b)
                for _ in range(3):
                    g.replay()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                reps = 60
                for _ in range(reps):
                    g.replay()
                e1.record()
                torch.cuda.synchronize()
                res[label] = e0.elapsed_time(e1) * 1e3 / (reps * len(batches))
                res["final_loss_" + label[:-3]] = float(loss.item())
                del g
            # -- the model's own share of the recorded step: pred = model(batch); pred.backward(ones) -- the step launch and the
            #    slab sum, none of torch's loss / optimiser launches (8 small kernels of ~4.4 us each in the figures above)
            net, opt = fresh()
            ones = torch.ones((B, 1), dtype=torch.float32, device=dev)

            def model_only(b):
                for p in net.parameters():
                    p.grad = None
                net(b).backward(ones)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for b in batches:
                    model_only(b)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for b in batches:
                    model_only(b)
            for _ in range(3):
                g.replay()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(60):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            res["graph_kept_model_only_us"] = e0.elapsed_time(e1) * 1e3 / (60 * len(batches))
            del g
            # -- the native trainer over the same cycle (topology of step t+1 built inside step t's launch) ---------------------
            nat = measure_distinct_batches(nets[net_name], net_name, dev, n_batches=n_batches, graphs=B, batches=batches)
            res["native_distinct_us"] = nat["us_per_step"]
            res["model_only_over_native"] = res["graph_kept_model_only_us"] / nat["us_per_step"]
            res["graph_kept_over_native"] = res["graph_kept_us"] / nat["us_per_step"]
            res["graph_kept_fused_adam_over_native"] = res["graph_kept_fused_adam_us"] / nat["us_per_step"]
            res["graph_rebuilt_fused_adam_over_native"] = res["graph_rebuilt_fused_adam_us"] / nat["us_per_step"]
            res["graphs_per_s_graph_kept_fused_adam"] = B / (res["graph_kept_fused_adam_us"] * 1e-6)
            out[key] = res
        except Exception as exc:                          # secondary figure: never lose the bench line over it
            out[key] = {"error": repr(exc)[:300]}
    return out


def measure_neuralnet_train(Net, net_name, dev, n_graphs=2048, epochs=20):
    """Secondary figure (not `value`): what a user of the trainer counterpart gets END TO END -- `deeprank_gnn_amd.NeuralNet(database,
    Net, ...).train(nepoch)` (the reference's NeuralNet.py:265-355 call) on a graph file of `n_graphs` synthetic graphs: upload
    once, shuffled epochs through the native loop (cached topology: the default), every host-side piece of an epoch included
    (shuffle, bookkeeping of outputs / targets / accuracy, the progress line).  us per mini-batch over whole train() calls,
    best of three."""
    import contextlib
    import io
    import shutil
    import tempfile
    import deeprank_gnn_amd.synthetic as synth
    from deeprank_gnn_amd.NeuralNet import NeuralNet
    tmp = tempfile.mkdtemp()
    try:
        db = synth.save_store(os.path.join(tmp, "syn.npz"), n_graphs, first_id=GRAPHS_PER_GPU, n_feat=N_FEAT)
        torch.manual_seed(0)
        quiet = io.StringIO()
        with contextlib.redirect_stdout(quiet):
            nn = NeuralNet(db, Net, node_feature=["feat"], edge_feature=["dist"], target="irmsd", batch_size=GRAPHS_PER_GPU,
                           percent=[1.0, 0.0], outdir=tmp, lr=1e-3)
            nn.train(nepoch=3, save_model=None, hdf5=None)      # upload, topology cache, allocations
        torch.cuda.synchronize()
        nb = (n_graphs + GRAPHS_PER_GPU - 1) // GRAPHS_PER_GPU
        runs = []
        for _ in range(3):
            with contextlib.redirect_stdout(quiet):
                t0 = time.perf_counter()
                nn.train(nepoch=epochs, save_model=None, hdf5=None)
                torch.cuda.synchronize()
                runs.append((time.perf_counter() - t0) / (epochs * nb) * 1e6)
        best = min(runs)
        return {"us_per_batch": best, "us_per_batch_runs": [round(r, 2) for r in runs], "graphs_per_s": n_graphs / (best * nb * 1e-6),
                "graphs": n_graphs, "batch": GRAPHS_PER_GPU, "batches_per_epoch": nb, "epochs_per_call": epochs, "net": net_name,
                "cached_topology": bool(nn._use_cache(nn._resident(nn.dataset))), "last_train_loss": float(nn.train_loss[-1]),
                "what": "deeprank_gnn_amd.NeuralNet(file of synthetic graphs, Net).train(%d) wall time per mini-batch, everything the "
                        "call does on the host included" % epochs}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def measure_epoch_loop(Net, net_name, n_graphs, dev, epochs=4):
    """Secondary figure (not `value`): whole shuffled epochs over a graph set resident in HBM, driven by the native
    loop drgnn_train_epoch -- every mini-batch is a different random selection of graphs, read in place by the topology
    builder.  The host never waits for an epoch before enqueuing the next one (shuffle, id upload and the launches of
    epoch e+1 are issued while epoch e runs; losses and predictions stay on the device and are read after the last
    epoch), which is how NeuralNet.train drives it.  Reported for 64 mini-batches per epoch (one pass over the set) and
    for 1024 (16 shuffled passes enqueued as one epoch)."""
    import deeprank_gnn_amd.synthetic as synth
    from deeprank_gnn_amd.resident import ResidentGraphSet
    from deeprank_gnn_amd.trainer import FusedTrainer
    graphs = [synth.make_graph(GRAPHS_PER_GPU + i, n_feat=N_FEAT) for i in range(n_graphs)]
    torch.manual_seed(0)
    tr = FusedTrainer(Net(N_FEAT, 1, 1).to(dev), lr=1e-3, task="reg")
    rs = ResidentGraphSet(graphs, dev)
    gen = torch.Generator().manual_seed(0)

    def run(cached, passes):
        def order():
            return torch.cat([torch.randperm(n_graphs, generator=gen) for _ in range(passes)])

        def enqueue():
            done = tr.train_epoch(rs, order(), GRAPHS_PER_GPU, cached=cached)
            if done is None:
                raise RuntimeError("the native epoch loop refused this configuration")
            return done
        enqueue()[0].sum().item()                      # warm-up epoch (allocations, topology cache)
        nb = passes * ((n_graphs + GRAPHS_PER_GPU - 1) // GRAPHS_PER_GPU)
        # The loop costs the host ~7 us per launch (two launches per mini-batch against ~21 us of device time): any hiccup
        # of the host shows.  Three timed repetitions, the fastest is reported (all three are kept in `us_per_batch_runs`).
        runs, sums = [], None
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pending = [enqueue() for _ in range(epochs)]   # no host synchronisation between epochs
            torch.cuda.synchronize()
            runs.append((time.perf_counter() - t0) / (epochs * nb) * 1e6)
            sums = [float(losses.sum()) for losses, _ in pending]
            del pending
        best = min(runs)
        return {"graphs_per_s": GRAPHS_PER_GPU / (best * 1e-6) * (n_graphs * passes / float(nb * GRAPHS_PER_GPU)),
                "us_per_batch": best, "us_per_batch_runs": [round(r, 2) for r in runs],
                "batches_per_epoch": nb, "last_epoch_loss_sum": sums[-1]}
    # the top-level entry is what NeuralNet.train runs by default (cached_topology = "auto": the per-graph topology built once
    # at upload whenever it fits the budget -- this set's does); `rebuilt_topology` = NeuralNet.cached_topology = False
    out = run(True, 1)
    out.update({"epochs": epochs, "resident_graphs": n_graphs, "batch": GRAPHS_PER_GPU, "net": net_name,
                "topology": "cached per graph, built once at upload (NeuralNet's default, cached_topology = 'auto')",
                "topology_cache_MiB": round(sum(t.numel() * t.element_size() for t in (
                    rs.topology_cache(need_weights=(net_name == "sGAT")).topo.ws_i32,
                    rs.topology_cache(need_weights=(net_name == "sGAT")).topo.tiles) if t is not None) / 2 ** 20, 1),
                "what": "shuffled epochs via drgnn_train_epoch (native loop; every mini-batch a different random selection of the "
                        "resident set's graphs, stepped straight out of the set's cached topology; epoch e+1 enqueued while epoch "
                        "e runs, outputs left on the device)"})
    for key, cached, passes, what in (
            ("rebuilt_topology", False, 1, "same loop, topology rebuilt for every mini-batch by the builder workgroups co-launched "
                                           "with the previous step (NeuralNet.cached_topology = False; round 5's top-level entry)"),
            ("long_epochs", False, 16, "1024 mini-batches per epoch (16 shuffled passes over the set enqueued as one epoch), rebuilt"),
            ("long_epochs_cached", True, 16, "1024 mini-batches per epoch, cached topology")):
        try:
            c = run(cached, passes)
            c["what"] = what
            out[key] = c
        except Exception as exc:
            out[key] = {"error": repr(exc)[:200]}
    return out


def measure_inference_loop(Net, net_name, n_graphs, dev, passes=4):
    """Secondary figure (not `value`, not the metric: forward + head only): predictions for every graph of a resident set
    through the native loop (FusedTrainer.predict_epoch -- what NeuralNet.test() runs; eval mode, one launch per mini-batch, the
    host never waits inside a pass), mini-batches of 64 and of 1024 graphs, topology rebuilt per mini-batch and cached."""
    import deeprank_gnn_amd.synthetic as synth
    from deeprank_gnn_amd.resident import ResidentGraphSet
    from deeprank_gnn_amd.trainer import FusedTrainer
    graphs = [synth.make_graph(GRAPHS_PER_GPU + i, n_feat=N_FEAT) for i in range(n_graphs)]
    torch.manual_seed(0)
    tr = FusedTrainer(Net(N_FEAT, 1, 1).to(dev), lr=1e-3, task="reg")
    rs = ResidentGraphSet(graphs, dev)
    order = torch.arange(n_graphs)
    out = {"resident_graphs": n_graphs, "net": net_name, "passes": passes,
           "what": "FusedTrainer.predict_epoch over the whole resident set (forward + head, eval mode), `passes` passes enqueued back "
                   "to back; best of three repetitions"}
    for bs in (GRAPHS_PER_GPU, 1024):
        for cached in (False, True):
            key = "batch%d_%s" % (bs, "cached" if cached else "rebuilt")
            try:
                pred = tr.predict_epoch(rs, order, bs, cached=cached)
                if pred is None:
                    raise RuntimeError("the native loop refused this configuration")
                check = float(pred.sum())
                runs = []
                for _ in range(3):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    pending = [tr.predict_epoch(rs, order, bs, cached=cached) for _ in range(passes)]
                    torch.cuda.synchronize()
                    runs.append(time.perf_counter() - t0)
                    del pending
                best = min(runs)
                nb = (n_graphs + bs - 1) // bs
                out[key] = {"graphs_per_s": n_graphs * passes / best, "us_per_batch": best / (passes * nb) * 1e6, "pred_sum": check}
            except Exception as exc:
                out[key] = {"error": repr(exc)[:200]}
    return out


def cpu_model():
    """model string of the host CPU (core count AND model are stated beside the CPU figure)"""
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def cpu_baseline(net_name, batch_cpu, seconds):
    """The CPU oracle (reference algorithm restated op for op) on this box's host cores: the best thread count of
    1/4/8/16/32/64 (`value`, `cores`), the single-thread figure beside it (`value_1thread`), and for FoutNet a second
    line with the vectorised FoutLayer (`value_vectorised`: the reference's per-node Python loop, foutnet.py:69-73, is
    what `value` times -- the vectorised form shows what the same arithmetic costs without it)."""
    from oracle import cpu_ref
    from deeprank_gnn_amd import hostcpu
    host = hostcpu.granted_cpus()                     # affinity mask and cgroup quota: threads beyond it are throttled, not run

    def make_step(**fw):
        params = {k: v.clone().requires_grad_(True) for k, v in cpu_ref.init_params(net_name, N_FEAT, 1, 1).items()}
        opt = torch.optim.Adam(list(params.values()), lr=1e-3)
        kw = {"dropout": 0.4, "training": True} if net_name == "GINet" else {}
        kw.update(fw)

        def step():
            opt.zero_grad()
            pred = cpu_ref.FORWARD[net_name](params, batch_cpu, **kw)
            loss = F.mse_loss(pred.reshape(-1), batch_cpu.y)
            loss.backward()
            opt.step()
        return step

    def rate(step, nthreads, budget):
        """graphs/s of `step` at `nthreads` over ~budget seconds (after 2 warm-up steps)"""
        torch.set_num_threads(nthreads)
        for _ in range(2):
            step()
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget or n < 3:
            step()
            n += 1
        dt = time.perf_counter() - t0
        return GRAPHS_PER_GPU * n / dt, n, dt

    step = make_step()
    # these are small ops: more threads is not faster.  Try a few counts briefly, keep the best.
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
    vectorised = net_name == "FoutNet"
    share = seconds / (3.0 if vectorised else 2.0)
    value, n, dt = rate(step, best_threads, share)
    value_1, n1, dt1 = (value, n, dt) if best_threads == 1 else rate(step, 1, share * 0.5)
    out = {"value": value, "unit": "graphs/s", "cores": best_threads, "kind": "port", "host_cores": host,
           "cpu_model": cpu_model(), "value_1thread": value_1,
           "sample": "%d train steps of the same 64-graph batch in %.1f s at %d threads + %d steps in %.1f s at 1 thread "
                     "(oracle/cpu_ref.py, torch %s CPU kernels, `value` = best of 1/4/8/16/32/64 threads)"
                     % (n, dt, best_threads, n1, dt1, torch.__version__)}
    if vectorised:
        vstep = make_step(looped=False)
        vbest, vt = 1, None
        for nt in sorted({1, 4, 8, 16, min(32, host)}):
            if nt > host:
                continue
            torch.set_num_threads(nt)
            vstep()
            t0 = time.perf_counter()
            vstep()
            dtv = time.perf_counter() - t0
            if vt is None or dtv < vt:
                vbest, vt = nt, dtv
        v, nv, dtv = rate(vstep, vbest, share)
        out["value_vectorised"] = v
        out["cores_vectorised"] = vbest
        out["sample"] += "; value_vectorised: FoutLayer without the reference's per-node loop (cpu_ref.fout_conv(looped=False)), " \
                         "%d steps in %.1f s at %d threads" % (nv, dtv, vbest)
    torch.set_num_threads(best_threads)
    return out


if __name__ == "__main__":
    main()
