#!/usr/bin/env python3
"""Headline benchmark: log-posterior evals/sec, 64 chains x (3x64 tanh MLP) x N=4096 per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scaling weak|strong] [--dtype f64|f32] [--kind logpost|grad]

One *step* = one lock-step evaluation of the log-posterior of all 64 chains' proposals on this
GPU (64 evals; `--kind grad`: log-posterior + parameter gradient, the HMC inner step), through
the C ABI, with weights / dataset resident in HBM.  Successive steps use different weight
batches (8 resident batches in rotation), nothing is cached between steps.  Chains shard over
ranks with no data-path collective (`--scaling weak`, the default: 64 chains per GPU; `--scaling
strong`: 64 chains in total, block-partitioned); one RCCL all_gather of the final log-posteriors
happens after the timed region.  Rank 0 prints ONE JSON line.

`--gpus N` with N > 1 and no WORLD_SIZE in the environment: this process starts the N ranks itself
(`quinn_amd.parallel.launch_ranks` = `python -m torch.distributed.run`, rendezvous on 127.0.0.1) BEFORE
it touches the GPU and forwards their output and exit code.  Launched by an outer torchrun
(WORLD_SIZE set) it is one of the ranks.  On a box with fewer than N GPUs the ranks share cuda:0 and
use gloo for the collectives (a rehearsal of the launch / shard / gather logic, marked in `config`).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from quinn_amd.ops import MLPArch, BatchedMLP, neg_log_post_from_sse  # noqa: E402
from quinn_amd import _lib  # noqa: E402

CHAINS, N, DIMS, SIGMA = 64, 4096, (1, 64, 64, 64, 1), 0.02
PEAK_TFLOPS = {"f64": 78.6, "f32": 157.3}      # MI355X dense matrix/vector peak (spec)
NBATCH = 8


def synthetic(n, d, noise=0.02, seed=0):
    """x ~ U[-pi,pi]^d, y = sum_j sin(x_j) + noise*randn (shape of ex_ufit.py:56-62)."""
    rs = np.random.RandomState(seed)
    x = rs.rand(n, d) * 2 * np.pi - np.pi
    y = noise * rs.randn(n, 1) + np.sum(np.sin(x), axis=1).reshape(-1, 1)
    return x, y


def cpu_baseline(arch, x, y, budget_s=20.0):
    """The oracle's sequential float64 path (one eval at a time: unflatten -> Linear/tanh ->
    NegLogPost -> .item(), as the reference does) on this host's cores; bounded sample.
    Timed with 1 thread and with min(16, os.cpu_count()) threads; the faster is `value`.  (More threads than 16
    only lose on this 64-wide network: the round-3 run with all 256 hardware threads of the box made 1.0 eval/s.)"""
    from oracle import mlp_ref
    mod = mlp_ref.build_module(mlp_ref.MLPSpec(arch.dims, arch.activ))
    yd = [v for v in y]
    ws = [0.1 * np.random.RandomState(1000 + c).randn(arch.nparams) for c in range(16)]
    ncpu = os.cpu_count() or 1
    counts = sorted({1, min(16, ncpu)})
    rates = {}
    old_threads = torch.get_num_threads()
    for nt in counts:
        torch.set_num_threads(nt)
        for i in range(10):
            mlp_ref.logpost(mod, ws[i % 16], x, yd, SIGMA)
        t0 = time.perf_counter()
        n = 0
        while True:
            mlp_ref.logpost(mod, ws[n % 16], x, yd, SIGMA)
            n += 1
            el = time.perf_counter() - t0
            if el > budget_s / len(counts) or n >= 20000:
                break
        rates[nt] = (n / el, n, el)
    torch.set_num_threads(old_threads)
    best = max(rates, key=lambda k: rates[k][0])
    r, n, el = rates[best]
    return {"value": r, "unit": "log-posterior evals/s", "cores": best, "kind": "port",
            "sample": f"{n} sequential float64 evals of the same workload (one chain at a time, N={x.shape[0]}, "
                      f"3x64 tanh MLP) in {el:.1f} s with {best} torch thread(s); "
                      + ", ".join(f"{k} thr: {v[0]:.1f}/s" for k, v in sorted(rates.items())),
            "by_threads": {str(k): v[0] for k, v in sorted(rates.items())},
            "host_cpus": ncpu}


def graph_rate(fn, dev, settle_ms=300.0, regions=5, min_region_ms=20.0):
    """Seconds per call of `fn` by the method of the headline number: the call captured in a HIP graph (as many consecutive
    calls as make >= 2 ms), replayed untimed for `settle_ms` (clock ramp of a fresh process), then `regions` timed regions of
    >= `min_region_ms` each between HIP events on the launch stream; returns (median, min, max) seconds per call."""
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        fn()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize(dev)
        once = max(time.perf_counter() - t0, 1e-6)
    torch.cuda.current_stream(dev).wait_stream(side)
    per = int(min(50, max(1, 2e-3 // once)))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(per):
            fn()
    g.replay()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < settle_ms:
        g.replay()
        torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    g.replay()
    torch.cuda.synchronize(dev)
    nrep = int(max(1, min_region_ms * 1e-3 // max(time.perf_counter() - t0, 1e-6) + 1))
    times = []
    for _ in range(regions):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(nrep):
            g.replay()
        b.record()
        torch.cuda.synchronize(dev)
        times.append(a.elapsed_time(b) * 1e-3 / (nrep * per))
    del g
    return float(np.median(times)), float(min(times)), float(max(times))


# the other BASELINE configurations at the sizes of tools/bench_configs.py (parity-test cases, not bench lines)
EXTRA_CFGS = (("cfg3_3x128_N8192_S128", (2, 128, 128, 128, 1), 8192, 128, "tanh"),
              ("cfg4_4x256_N16384_M64", (1, 256, 256, 256, 256, 1), 16384, 64, "tanh"),
              ("cfg5_4x256_N32768_C32", (1, 256, 256, 256, 256, 1), 32768, 32, "tanh"),
              # the reference's DEFAULT activation (quinn/nns/mlp.py:23) at the cfg2 / cfg3 shapes: int8 slices with per-row scales
              ("relu_3x64_N4096_C64", (1, 64, 64, 64, 1), 4096, 64, "relu"),
              ("relu_3x128_N8192_S128", (2, 128, 128, 128, 1), 8192, 128, "relu"))


def extras(op, arch, batches, args):
    """Secondary measurements (not part of `value`), every rate by `graph_rate` -- the method of the headline number (HIP
    graph, settled clock, HIP events, median of 5 regions): the other kind of eval on the resident workload, the headline step in
    plain float64 arithmetic (`exact_f64`: the float64-MFMA fused kernels), the end-to-end device-resident AMCMC loop, and
    the other BASELINE network shapes."""
    out = {"method": "each rate: call captured in a HIP graph, replayed 300 ms untimed, then the median of 5 regions (>= 20 ms each) "
                     "between HIP events; TFLOP/s = algorithmic float64 flops (SURVEY 8d) / time"}
    dev = op.device
    peak = PEAK_TFLOPS[args.dtype]
    other = "logpost" if args.kind == "grad" else "grad"
    k = [0]

    def rot(f):
        def call():
            k[0] += 1
            return f(batches[k[0] % NBATCH])
        return call
    fn = rot(op.sse) if other == "logpost" else rot(op.sse_grad)
    flops = arch.flops_fwd(N) if other == "logpost" else arch.flops_fwdbwd(N)
    nloc = batches[0].shape[0]
    t, tmin, tmax = graph_rate(fn, dev)
    out[other + "_evals_per_s"] = nloc / t
    out[other + "_tflops"] = nloc * flops / t / 1e12
    out[other + "_frac"] = nloc * flops / t / 1e12 / peak
    out[other + "_ms_per_step"] = {"median": 1e3 * t, "min": 1e3 * tmin, "max": 1e3 * tmax}
    if args.dtype == "f64" and args.path == "auto":
        # the same step in plain float64 arithmetic (kernels='float64': k_fused_fwd_f64 / k_fused_bwd_f64 on the float64 matrix cores)
        old = op.set_path(_lib.PATH_FUSED_DP)
        try:
            ex = {}
            for kind, f2, fl in (("logpost", rot(op.sse), arch.flops_fwd(N)), ("grad", rot(op.sse_grad), arch.flops_fwdbwd(N))):
                t, tmin, tmax = graph_rate(f2, dev)
                ex[kind] = {"evals_per_s": nloc / t, "tflops": nloc * fl / t / 1e12, "frac": nloc * fl / t / 1e12 / peak,
                            "ms_per_step": 1e3 * t}
            ex["kernels"] = "k_fused_fwd_f64 / k_fused_bwd_f64 (QN_PATH_FUSED_DP: float64 MFMA + float64 VALU, no operand rounding)"
            out["exact_f64"] = ex
        finally:
            op.set_path(old)
    if args.dtype == "f64":
        from quinn_amd.mcmc.device_amcmc import DeviceAMCMC
        ini = np.stack([np.random.RandomState(1000 + c).rand(arch.nparams) for c in range(nloc)])
        eng = DeviceAMCMC(op, SIGMA, gamma=0.01, t0=100, tadapt=1000, seed=1)
        eng.run(300, ini, store_chain=True)                                  # (also settles the clock on this launch pattern)
        torch.cuda.synchronize(dev)
        rates = []
        for _ in range(3):
            t0 = time.perf_counter()
            r = eng.run(900, ini, store_chain=True)                          # (900 < tadapt: initial proposal throughout)
            torch.cuda.synchronize(dev)
            rates.append(900 / (time.perf_counter() - t0))
        out["amcmc_end_to_end_steps_per_s"] = float(np.median(rates))
        out["amcmc_end_to_end_logpost_evals_per_s"] = float(np.median(rates)) * nloc
        out["amcmc_accrate"] = float(r["accrate"].mean())
        # with adaptation switched on early (t0=100, tadapt=200): adapted proposals are drawn in sample
        # space from the stored distinct states (qn_mcmc_propose_hist_block); cost grows with accepted moves
        eng = DeviceAMCMC(op, SIGMA, gamma=0.01, t0=100, tadapt=200, seed=1)
        eng.run(1000, ini, store_chain=True)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        r = eng.run(1000, ini, store_chain=True)
        torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0
        out["amcmc_adapting_steps_per_s"] = 1000 / el
        out["amcmc_adapting_accrate"] = float(r["accrate"].mean())
        del eng, r
        torch.cuda.empty_cache()
        for name, dims, n_rows, nb, act2 in EXTRA_CFGS:
            a2 = MLPArch(dims, act2)
            x2, y2 = synthetic(n_rows, dims[0])
            op2 = BatchedMLP(a2, x2, y2, device=dev)
            W2 = op2.weights(0.1 * np.random.RandomState(7).randn(nb, a2.nparams))
            res = {"arith_fwd": op2.arith(nb, n_rows, False), "arith_grad": op2.arith(nb, n_rows, True)}
            for kind, f2, fl in (("fwd", lambda: op2.sse(W2), a2.flops_fwd(n_rows)), ("grad", lambda: op2.sse_grad(W2), a2.flops_fwdbwd(n_rows))):
                t, tmin, tmax = graph_rate(f2, dev)
                res[kind + "_evals_per_s"] = nb / t
                res[kind + "_tflops"] = nb * fl / t / 1e12
                res[kind + "_frac"] = nb * fl / t / 1e12 / peak
                res[kind + "_ms"] = {"median": 1e3 * t, "min": 1e3 * tmin, "max": 1e3 * tmax}
            out[name] = res
            del op2, W2
            torch.cuda.empty_cache()
    return out


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: 64 chains per GPU; strong: 64 chains in total, block-partitioned over the GPUs")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--kind", default="logpost", choices=["logpost", "grad"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary measurements (N=1 only)")
    ap.add_argument("--path", default="auto", choices=["auto", "generic", "fused", "fused_dp"],
                    help="kernel family (fused_dp: the float64-MFMA fused kernels, without the sliced int8-product forward)")
    ap.add_argument("--graph", type=int, default=-1, help="capture this many consecutive steps in one HIP graph and "
                    "replay it (0 = direct launches, -1 = the largest divisor of --steps up to 50)")
    ap.add_argument("--regions", type=int, default=5, help="timed regions of EXACTLY --steps steps each (barrier + synchronize on both "
                    "sides, max over ranks); `value` / `ms_per_step` come from the MEDIAN region, all of them are listed in `config`")
    ap.add_argument("--settle-ms", type=float, default=500.0, help="untimed run of the step before the timed region until the GPU "
                    "clock has settled (it ramps up over the first ~500 launches of a fresh process: the same kernel takes 97 us "
                    "right after start-up and 80 us from then on, profiles/r03_clock_ramp_kernel_trace.txt); 0 = off")
    return ap.parse_args()


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # parent of the N ranks: starts them before anything here touches the GPU, forwards output and exit code
        from quinn_amd.parallel import launch_ranks
        env = dict(os.environ)
        if torch.cuda.device_count() < args.gpus:        # (counting devices does not initialise the GPU)
            env["QN_BENCH_BACKEND"] = "gloo"            # rehearsal: the ranks share cuda:0, CPU-side collectives
        sys.exit(launch_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus, env=env))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; measuring {world} rank(s)", file=sys.stderr)
    dist = None
    # rehearsal on a 1-GPU box: QN_BENCH_BACKEND=gloo runs all ranks on cuda:0 with CPU-side collectives
    backend = os.environ.get("QN_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = 0
    if local >= torch.cuda.device_count():
        print(f"bench.py: rank {rank}/{world}: LOCAL_RANK {local} but {torch.cuda.device_count()} GPU(s) visible -- one process per GPU "
              "(rehearsal of the launch on fewer GPUs: QN_BENCH_BACKEND=gloo, all ranks on cuda:0)", file=sys.stderr, flush=True)
        sys.exit(3)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cdev = dev if backend == "nccl" else torch.device("cpu")
    # QN_BENCH_FORCE_DIST=1: create the process group even for ONE rank (WORLD_SIZE=1 RANK=0 MASTER_PORT=... in the environment):
    # on a one-GPU box this runs RCCL's communicator set-up, barrier, all_reduce and all_gather and the graph capture beside a
    # live process group -- everything of the N > 1 path but the transport between ranks
    if world > 1 or os.environ.get("QN_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # One process per GPU, started by torchrun (or by this script's parent, before it touched the GPU).  A rank whose RCCL
        # set-up fails says so in ONE line and exits non-zero: torchrun then ends the other ranks and returns that code; nothing
        # here restarts or re-execs a process that has initialised the GPU.
        # RCCL announces itself on STDOUT when the first communicator is made ("RCCL version : ..." and four more lines, seen in the
        # one-rank rehearsal on the GPU box): while the communicator is set up, file descriptor 1 points at stderr, so that rank 0's
        # stdout carries the ONE JSON line and nothing else
        import ctypes
        sys.stdout.flush()
        saved_out = None
        try:
            saved_out = os.dup(1)
            os.dup2(2, 1)
        except OSError:                                      # (no usable fd 1 / fd 2: leave stdout alone)
            if saved_out is not None:
                os.close(saved_out)
            saved_out = None
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            else:
                dist.init_process_group(backend, rank=rank, world_size=world)
            probe = torch.ones(1, device=cdev, dtype=torch.float64)
            dist.all_reduce(probe)                               # the first collective creates the communicator: fail here, not mid-run
            torch.cuda.synchronize(dev)
            if float(probe.item()) != float(world):
                raise RuntimeError(f"all_reduce of ones over {world} ranks returned {float(probe.item())}")
        except Exception as e:  # noqa: BLE001
            print(f"bench.py: rank {rank}/{world}: {backend} (RCCL) initialisation failed on cuda:{local}: {type(e).__name__}: "
                  f"{str(e).splitlines()[0] if str(e) else ''} -- HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')}, "
                  f"MASTER_ADDR={os.environ.get('MASTER_ADDR')}, visible GPUs {torch.cuda.device_count()}", file=sys.stderr, flush=True)
            sys.exit(3)
        finally:
            sys.stdout.flush()
            try:
                ctypes.CDLL(None).fflush(None)                   # (the C library's buffered stdout, while it still goes to stderr)
            except Exception:  # noqa: BLE001
                pass
            if saved_out is not None:
                os.dup2(saved_out, 1)
                os.close(saved_out)

    from quinn_amd.parallel import shard_bounds
    # this rank's chains [lo, hi) of the job's `total`; global chain id c: W[c] = 0.1*RandomState(1000+c).randn(p)
    total = CHAINS * world if args.scaling == "weak" else CHAINS
    lo, hi = shard_bounds(total, rank, world)
    nloc = hi - lo
    if nloc < 1:
        raise SystemExit(f"rank {rank} owns no chain ({total} chains over {world} ranks)")

    tdt = "float64" if args.dtype == "f64" else "float32"
    arch = MLPArch(DIMS, "tanh")
    x, y = synthetic(N, DIMS[0])
    op = BatchedMLP(arch, x, y, device=dev, dtype=tdt)
    path_force = {"auto": _lib.PATH_AUTO, "generic": _lib.PATH_GENERIC, "fused": _lib.PATH_FUSED,
                  "fused_dp": _lib.PATH_FUSED_DP}[args.path]
    op.set_path(path_force)
    batches = []
    for k in range(NBATCH):
        Wk = np.stack([0.1 * np.random.RandomState(1000 + c + 100003 * k).randn(arch.nparams) for c in range(lo, hi)])
        batches.append(op.weights(Wk))
    want_grad = args.kind == "grad"
    run = (lambda W: op.sse_grad(W)) if want_grad else (lambda W: (op.sse(W), None))

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for i in range(args.warmup):
        out = run(batches[i % NBATCH])
    if args.graph < 0:
        # the largest divisor of --steps up to 50 that leaves at least two replays in the timed region
        cap = min(50, max(1, args.steps // 2))
        args.graph = max(g for g in range(1, cap + 1) if args.steps % g == 0)
        if args.graph == 1:
            args.graph = 0

    def capture(fn, nsteps):
        """nsteps consecutive steps (rotating over the resident weight batches) captured once in a HIP graph."""
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for i in range(NBATCH):
                res_ = fn(batches[i])
        torch.cuda.current_stream(dev).wait_stream(side)
        g = torch.cuda.CUDAGraph()
        # with a process group alive its watchdog / heartbeat threads may issue HIP calls of their own: relaxed (thread-local)
        # capture keeps them from invalidating this thread's capture; a single process keeps the strict default
        with torch.cuda.graph(g, capture_error_mode="thread_local" if dist is not None else "global"):
            for i in range(nsteps):
                res_ = fn(batches[i % NBATCH])
        g.replay()
        return g, res_

    if args.graph:
        assert args.steps % args.graph == 0
        graph, out = capture(run, args.graph)
    # the dominant kernel ALONE (what rocprofv3 reports for it): for the log-posterior the forward without the arrival
    # epilogue that adds the row-split partial sums of a chain (qn_mlp_sse_fwd_parts); replayed after EACH timed region,
    # i.e. interleaved with them at the same clock
    alone = (lambda W: (op.sse_parts(W), None)) if (not want_grad and op.sse_parts(batches[0]).shape[1] > 1) else run
    g2n = args.graph if args.graph else 1
    g2, _ = capture(alone, g2n)
    # untimed: the step itself, replayed until the clock of a freshly started GPU has settled
    settle_steps = 0
    if args.settle_ms > 0:
        t_s = time.perf_counter()
        while (time.perf_counter() - t_s) * 1e3 < args.settle_ms:
            if args.graph:
                graph.replay()
                settle_steps += args.graph
            else:
                for i in range(10):
                    out = run(batches[i % NBATCH])
                settle_steps += 10
            torch.cuda.synchronize(dev)
    nlaunch = args.steps // args.graph if args.graph else args.steps
    nrep2 = max(1, args.steps // g2n)
    regions = max(1, args.regions)
    el_r, dev_r, kern_r = [], [], []
    for r in range(regions):
        # ---- one timed region: EXACTLY --steps steps between barrier + synchronize on both sides
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        barrier()
        t0 = time.perf_counter()
        e0.record()
        if args.graph:
            for i in range(nlaunch):
                graph.replay()
        else:
            for i in range(args.steps):
                out = run(batches[i % NBATCH])
        e1.record()
        torch.cuda.synchronize(dev)
        el_r.append(time.perf_counter() - t0)                                 # this rank's K steps, done and synchronized;
        barrier()                                                            # the closing barrier; MAX over ranks below
        dev_r.append(e0.elapsed_time(e1) / args.steps)                       # device ms per step (HIP events, launch stream)
        # ---- the kernel alone, same number of steps, right behind the region
        k0, k1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        k0.record()
        for i in range(nrep2):
            g2.replay()
        k1.record()
        torch.cuda.synchronize(dev)
        kern_r.append(k0.elapsed_time(k1) / (nrep2 * g2n))
    lp_last = -neg_log_post_from_sse(out[0].cpu().numpy(), N, SIGMA)
    assert np.all(np.isfinite(lp_last))
    # region times: MAX over ranks per region, then the MEDIAN region is the one reported
    el_t = torch.tensor(el_r, dtype=torch.float64)
    if dist is not None:
        t = el_t.to(cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el_t = t.cpu()
    el = float(el_t.median())                                                # (lower median for an even count)
    step_dev_ms = float(np.median(dev_r))
    kms = np.array(kern_r)
    kern = {"kernel_ms": float(np.median(kms)), "kernel_ms_min": float(kms.min()), "kernel_ms_median": float(np.median(kms)),
            "kernel_ms_max": float(kms.max()), "kernel_ms_samples": int(regions), "kernel_steps_per_sample": int(nrep2 * g2n)}

    t_max = el
    if dist is not None:
        # the single end-of-run collective (RCCL): every rank's final log-posteriors, padded to the largest shard
        nmax = -(-total // world)
        send = torch.zeros(nmax, device=cdev, dtype=torch.float64)
        send[:nloc] = out[0].to(cdev)
        gathered = torch.empty(world * nmax, device=cdev, dtype=torch.float64)
        dist.all_gather_into_tensor(gathered, send)
        torch.cuda.synchronize(dev)
        assert torch.isfinite(gathered).all()

    if rank == 0:
        evals = total * args.steps
        value = evals / t_max
        flops = arch.flops_fwdbwd(N) if want_grad else arch.flops_fwd(N)
        ach = nloc * flops / (kern["kernel_ms"] * 1e-3) / 1e12
        peak = PEAK_TFLOPS[args.dtype]
        path = op.path(nloc, N, want_grad)
        # which kernel the step really dispatches, and what binds it
        fam = {1: "generic", 2: "fused"}.get(path, str(path))
        i8 = fam == "fused" and args.dtype == "f64" and args.path != "fused_dp"
        if fam == "fused":
            kernel = ("k_fused_bwd_i8" if i8 else "k_fused_bwd_f64") if want_grad else ("k_fused_fwd_i8" if i8 else "k_fused_fwd_f64")
        else:
            kernel = "layer-wise kernels (qn_generic.hip)"
        kpath = {"k_fused_fwd_i8": "fused_i8", "k_fused_bwd_i8": "fused_i8_bwd", "k_fused_fwd_f64": "fused_dp", "k_fused_bwd_f64": "fused_dp_bwd"}.get(kernel, fam)
        # int8 matrix work the sliced kernels execute: 26 kept digit products per 64 x 64 product and 16 rows
        # (v_mfma_i32_16x16x64_i8: 32768 ops in 16 cycles per SIMD -> 5.03 Pop/s dense on 256 CUs at 2.4 GHz)
        mfmas_per_group = 4 * 26 * (6 if want_grad else 2) + (2 * 6 if want_grad else 0)   # per 16 data rows: 64 x 64 products (+ bias row sums)
        i8_ops = nloc * (N / 16) * mfmas_per_group * 32768 if i8 else None
        traffic = traffic_src = None
        tfile = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        pfile = os.path.join(ROOT, "profiles", "pmc_busy.json")
        if os.path.exists(tfile) and nloc == CHAINS:
            # HBM bytes per launch from separate rocprofv3 --pmc passes (FETCH_SIZE doubled per the gfx950
            # correction, + WRITE_SIZE), recorded by tools/prof_traffic.sh for this kernel / config: NOT of this run
            key = f"{args.kind}_{args.dtype}_{kpath}"
            ent = json.load(open(tfile)).get(key, {})
            traffic = ent.get("hbm_bytes_per_launch")
            traffic_src = f"profiles/hbm_traffic.json[{key}] (separate rocprofv3 --pmc passes of this command, round {ent.get('round', '?')})" if traffic else None
        busy = {}
        if os.path.exists(pfile):
            busy = json.load(open(pfile)).get(kernel, {})
        per_gpu = f"{CHAINS} AMCMC chains/GPU" if args.scaling == "weak" else f"{CHAINS} AMCMC chains in total ({nloc} on rank 0)"
        cfg = {"workload": f"configs[1]: {per_gpu}, 3x64 tanh MLP (p=8513), N=4096 1-D regression; step = batched "
                           + ("log-posterior+gradient" if want_grad else "log-posterior") + " of all chains",
               "chains_total": total, "chains_rank0": nloc, "N": N, "dims": list(DIMS),
               "kind": args.kind, "kernel_path": kpath, "kernel": kernel, "settle_ms": args.settle_ms, "settle_steps": settle_steps,
               "launch": (f"HIP graph of {args.graph} steps, replayed {args.steps // args.graph}x" if args.graph
                          else "direct launches"),
               "timed_regions": int(regions), "value_from": f"median of {regions} timed regions of {args.steps} steps each (per region: barrier + synchronize, K steps, synchronize -> this rank's time, closing barrier; MAX over ranks)",
               "region_ms_per_step": [1e3 * float(v) / args.steps for v in el_t],
               "parallelism": f"chains sharded x{world}, no data-path collective, one all_gather at the end"}
        if backend != "nccl" and world > 1:
            cfg["rehearsal"] = f"{world} ranks share cuda:0 of a {torch.cuda.device_count()}-GPU box, {backend} collectives"
        if dist is not None and world == 1:
            cfg["rehearsal"] = f"one rank with a live {backend} process group (QN_BENCH_FORCE_DIST=1): communicator set-up, barriers, all_reduce, all_gather"
        res = {
            "metric": "log-posterior evals/sec (64 chains, 3x64 MLP, N=4096) at 1/2/4/8 GPU",
            "value": value, "unit": "log-posterior evals/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * t_max / args.steps, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": cfg,
            "roofline": dict({
                # what binds the dispatched kernel: the int8-slice kernels are bound by vector-instruction ISSUE (float64 VALU:
                # recombination, tanh, slicing) with the hidden GEMMs on the int8 matrix pipe beside it; the float64-MFMA
                # kernels by the float64 pipe that MFMA and VALU share
                "bound": "mfma", "bound_detail": "valu+i8mfma" if i8 else ("f64mfma" if fam == "fused" else "f64 valu/mfma + hbm (layer-wise)"),
                "arith": ("47-bit operand slices (2^-47 of the row / activation scale), exact int8 products, float64 accumulation"
                          if i8 else args.dtype),
                # yardstick shared by every kernel of this repo: ALGORITHMIC float64 flops of the evaluation against
                # the float64 MFMA peak -- for the int8-slice kernels a float64-EQUIVALENT rate, not work done on that pipe
                "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                "yardstick": "algorithmic float64 flops / kernel time vs the float64 MFMA peak (float64-equivalent for the int8-slice kernels)",
                "frac_step": nloc * flops / (t_max / args.steps) / 1e12 / peak,
                "int8_pipe": ({"ops_per_launch": i8_ops, "achieved_pops": i8_ops / (kern["kernel_ms"] * 1e-3) / 1e15, "peak_pops": 5.03,
                               "frac": i8_ops / (kern["kernel_ms"] * 1e-3) / 1e15 / 5.03} if i8 else None),
                "pmc": dict(busy, source="profiles/pmc_busy.json (rocprofv3 --pmc passes of this command, committed with the profile "
                                         "it was taken from; not of this run)") if busy else None,
                "traffic": traffic, "traffic_source": traffic_src, "flops_per_eval": flops, "evals_per_launch": nloc,
                "step_device_ms": step_dev_ms}, **kern),
        }
        if world == 1 and not args.no_extras:
            res["extras"] = extras(op, arch, batches, args)
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(arch, x, y)
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
