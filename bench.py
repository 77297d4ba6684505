"""GIM episodes/s on MI355X: one "step" = one training iteration of the hot path on one episode batch
(generator step + discriminator step, both Adam updates; training/gim_img_training.py:225-239) over synthetic
inputs already resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload vox64|om32|vox128] [--batch B] [--no-cpu-baseline] ...

``--gpus N`` with N > 1 launches itself: when WORLD_SIZE is not set the script starts N child processes (one rank per
GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT in their environment), relays rank 0's JSON line
and exits with the worst return code.  Under ``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`` it
uses the environment it is given.

Prints ONE JSON line on rank 0 (contract in the task statement): whole-job episodes/s; the MFMA roofline of the step -
the ALGORITHMIC figure of SURVEY.md 8(d) (FLOPs of the unfused reference ops) and, beside it, the EXECUTED figure (the pool /
sub-pixel folds run 16/36 resp. 100/324 of a 3x3 / 9x9 convolution's taps: that one is the matrix-pipe utilisation); the
dominant convolution's own roofline from a HIP-event microbenchmark through the C ABI; HBM traffic per step from rocprofv3
PMC passes of this same command; the CPU baseline (the oracle, a port of the reference's path, timed on the host cores on
a bounded sample).
"""
import argparse
import csv
import gc
import glob
import json
import os

# one hardware queue per stream (the step's four + RCCL's): see optimalstrategiesagainstgenerativeattacks_amd/__init__.py;
# set before anything touches the GPU, kept if the caller set it
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import shutil  # noqa: E402
import socket  # noqa: E402
import subprocess  # noqa: E402
import sys  # noqa: E402
import tempfile  # noqa: E402
import time  # noqa: E402

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_FP16_MFMA_TFLOPS = 2516.6  # 16x the fp32 MFMA rate (v_mfma_f32_32x32x16_f16: 32768 FLOP in 32 cycles per SIMD), ~2.5 PF dense
# forward unit costs in GFLOP (2*MAC; conv + linear + attention bmm), style_dim = 512
UNIT = {
    "om32": dict(S=32, C=1, E=1.2570, M=0.00210, Dec=0.2274, I2I=5.0189, FC=0.00944, H0=0.01468),
    "vox64": dict(S=64, C=3, E=1.7564, M=0.00210, Dec=0.9583, I2I=6.0770, FC=0.00944, H0=0.01468),
    # BASELINE config 5 shape (128x128x3, m=5 n=20 k=20), run here in fp32 (the reference has no fp16 path)
    "vox128": dict(S=128, C=3, E=7.6252, M=0.00210, Dec=3.9013, I2I=23.4687, FC=0.00944, H0=0.01468, mnk=(5, 20, 20), batch=2),
}


_T0 = time.time()


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench %7.1fs] %s" % (time.time() - _T0, msg), file=sys.stderr, flush=True)


def algo_gflop_per_episode(w, m, n, k):
    """ALGO = 3*F_im + F_auG + (2nE + H) + 3*F_auD  (SURVEY.md 8(d))."""
    u = UNIT[w]
    F_im = 2 * m * u["E"] + n * (u["M"] + u["Dec"] + u["I2I"])
    H = (n + k) * u["FC"] + u["H0"]
    F_auG = 2 * (n + k) * u["E"] + H
    F_auD = 2 * (k + 2 * n) * u["E"] + 2 * H
    return 3 * F_im + F_auG + (2 * n * u["E"] + H) + 3 * F_auD


def synthetic_batch(B, m, n, k, C, S, device, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    mk = lambda t: (torch.rand((B, t, C, S, S), generator=g) * 2 - 1).to(device)  # noqa: E731  dataset range is [-1, 1]
    return mk(m), mk(n), mk(k)


def build_trainer(S, C, n, m, k, device, style_dim=512, reg_param=0.0):
    import optimalstrategiesagainstgenerativeattacks_amd as G
    torch.manual_seed(1)  # train_gim_on_imgs.py:6
    au, im = G.get_au(S, C, style_dim).to(device), G.get_im(S, C, style_dim).to(device)
    with tempfile.TemporaryDirectory() as td:
        tr = G.GIMImgTrainer(td, m, n, k, au, im, au_lr=1e-4, im_lr=1e-4, env_noise_mapping_lr=1e-6, beta1=0.0, beta2=0.99,
                             reg_param=reg_param)
    return G, tr


# ----------------------------------------------------------------------------------------------------------------------
# --gpus N without a launcher: start the ranks ourselves
# ----------------------------------------------------------------------------------------------------------------------
def self_launch(n, argv):
    """N fresh child processes of this script, one per GPU; never os.exec*, and nothing in this parent touches the GPU."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out0)      # rank 0's stdout holds the ONE JSON line (its stderr went straight to ours)
    sys.stdout.flush()
    worst = max(rcs, key=abs)
    if worst:
        print("[bench] rank return codes: %s" % rcs, file=sys.stderr)
    return worst


# ----------------------------------------------------------------------------------------------------------------------
# HBM traffic of this same command (rocprofv3 PMC, two passes in child processes, before this process touches the GPU)
# ----------------------------------------------------------------------------------------------------------------------
def measure_hbm_traffic(args, steps=2, warmup=1, timeout=420):
    """HBM bytes per step: `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` in SEPARATE runs of
    `python3 bench.py --inner` (the same workload, `steps` timed + `warmup` steps), summed over the kernels launched after the
    model is built, corrected as MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE x2: 128-byte requests tallied at 64 B;
    WRITE_SIZE exact; both reported in KiB), divided by the steps run.  None if the profiler is not available."""
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None, "rocprofv3 not found"
    tot = {}
    tmp = tempfile.mkdtemp(prefix="gim_pmc_", dir="/tmp")
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, ctr)
            cmd = [prof, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "r", "--",
                   sys.executable, os.path.abspath(__file__), "--inner", "--workload", args.workload, "--batch", str(args.batch),
                   "--steps", str(steps), "--warmup", str(warmup), "--reg-param", str(args.reg_param), "--matrix-path", args.matrix_path]
            env = dict(os.environ, TMPDIR="/tmp")
            t0 = time.time()
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, "rocprofv3 --pmc %s failed (rc %d): %s" % (ctr, r.returncode, r.stderr[-300:])
            kib, seen_conv = 0.0, False
            rows = list(csv.DictReader(open(files[0])))
            if rows and "Dispatch_Id" in rows[0]:
                rows.sort(key=lambda r_: int(r_["Dispatch_Id"]))
            for row in rows:
                # model construction (fills / copies of the initialisers) comes before the first engine kernel: not the step's
                seen_conv = seen_conv or row["Kernel_Name"].startswith(("snb_", "void conv_", "nchw_to_nhwc"))
                if seen_conv:
                    kib += float(row["Counter_Value"])
            tot[ctr] = kib * 1024.0
            log("traffic pass %s: %.1f s" % (ctr, time.time() - t0))
    except (subprocess.TimeoutExpired, OSError, KeyError, ValueError) as e:
        return None, "traffic measurement failed: %r" % (e,)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    n = steps + warmup
    fetch, write = 2.0 * tot["FETCH_SIZE"] / n, tot["WRITE_SIZE"] / n
    return {"hbm_bytes_per_step": fetch + write, "fetch_bytes_per_step": fetch, "write_bytes_per_step": write,
            "how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py --inner --steps %d "
                   "--warmup %d; FETCH_SIZE x2 (gfx950), kernels from the first engine launch on, / %d steps" % (steps, warmup, n)}, None


# ----------------------------------------------------------------------------------------------------------------------
# executed work and the dominant convolution
# ----------------------------------------------------------------------------------------------------------------------
def summarize_flops(counts, ops):
    """counts = ops.count_flops() of ONE step -> (executed GFLOP, algorithmic GFLOP of the same launches, per-conv table)."""
    exe = algo = 0.0
    per = {}
    for (kind, cfg), (n, f) in counts.items():
        exe += n * f
        if kind == "bgemm":
            algo += n * f
            continue
        algo += n * ops.conv_algorithmic_flops(*cfg)
        ent = per.setdefault(cfg, {"fwd": 0, "dgrad": 0, "wgrad": 0, "gflop": f / 1e9})
        ent[kind] += n
    return exe / 1e9, algo / 1e9, per


def dominant_kernel_roofline(per, device):
    """HIP-event timing, through the C ABI, of the convolution that EXECUTES the most FLOPs per step (launches x executed FLOPs
    per launch, forward + dgrad + wgrad), with the geometry the step launches it in (fold / pool flags, table row in force)."""
    from optimalstrategiesagainstgenerativeattacks_amd import _lib, ops
    lib = _lib.load()
    cfg = max(per, key=lambda c: per[c]["gflop"] * (per[c]["fwd"] + per[c]["dgrad"] + per[c]["wgrad"]))
    N, H, W, Cin, Cout, K, ups, pool, fold = cfg
    sh = _lib.GimConvShape(N, H, W, Cin, Cout, K, ups, 0.2, pool, fold, 0)
    sh.prec = 1 if ops.matrix_path() == "fp16" else 0
    peak = PEAK_FP16_MFMA_TFLOPS if sh.prec else PEAK_FP32_MFMA_TFLOPS
    KF = K + 1 if fold else K
    x = torch.randn(N, H >> ups, W >> ups, Cin, device=device)
    w = torch.randn(Cout, KF, KF, Cin, device=device) * 0.05     # folded layout when fold
    y = torch.randn(N, H >> pool, W >> pool, Cout, device=device)
    lo = 1 if (ups and fold) else 0
    dx = torch.empty(N, H >> lo, W >> lo, Cin, device=device)
    acc = torch.zeros(Cout * KF * KF * Cin, device=device)
    st = torch.cuda.current_stream().cuda_stream
    gf = per[cfg]["gflop"]
    dgrad = lambda: lib.gim_conv2d_dgrad(y.data_ptr(), w.data_ptr(), None, x.data_ptr(), dx.data_ptr(), sh, st)   # noqa: E731
    out = {}
    for name, fn in (("fwd", lambda: lib.gim_conv2d_fwd(x.data_ptr(), w.data_ptr(), None, None, None, y.data_ptr(), sh, st)),
                     ("dgrad", dgrad),
                     ("wgrad", lambda: lib.gim_conv2d_wgrad_acc(y.data_ptr(), x.data_ptr(), acc.data_ptr(), None, sh, st))):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        reps = 20
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        tf = gf / ms
        out[name] = {"ms": round(ms, 4), "executed_tflops": round(tf, 2), "frac": round(tf / peak, 4)}
    return {"kernel": "conv N=%d %dx%d %d->%d %dx%d%s%s (%s)" % (N, H, W, Cin, Cout, K, K, ", avg-pool folded (stride-2, %dx%d taps)" % (KF, KF) if pool else "",
                                                              ", sub-pixel form of the upsampled conv" if (ups and fold) else "",
                                                              "fp16 operands" if sh.prec else "fp32 MFMA"),
            "launches_per_step": {k_: per[cfg][k_] for k_ in ("fwd", "dgrad", "wgrad")},
            "executed_gflop_per_launch": round(gf, 2),
            "algorithmic_gflop_per_launch": round(ops.conv_algorithmic_flops(*cfg) / 1e9, 2), **out}


def cpu_baseline(workload, m, n, k, sample_B):
    """The oracle (CPU port of the reference's path, parity-pinned by tests/golden) on a bounded sample: steps of `sample_B`
    episodes (the GPU's own batch where one such step fits the ~10-30 s budget) after a small warm-up step."""
    from oracle import gim_oracle as go
    import optimalstrategiesagainstgenerativeattacks_amd as G
    u = UNIT[workload]
    cores, how = host_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(1)
    au, im = G.get_au(u["S"], u["C"], 512), G.get_im(u["S"], u["C"], 512)
    au_sd = {kk: v.detach().clone().contiguous() for kk, v in au.state_dict().items()}
    im_sd = {kk: v.detach().clone().contiguous() for kk, v in im.state_dict().items()}
    otr = go.OracleTrainer(au_sd, im_sd, n, 1e-4, 1e-4, 1e-6)
    leaked, real, si = synthetic_batch(sample_B, m, n, k, u["C"], u["S"], "cpu", 1234)
    z = torch.randn(sample_B, n, 512)
    wb = min(2, sample_B)
    t0 = time.time()
    otr.step(leaked[:wb], real[:wb], si[:wb], z[:wb])  # warm-up (thread pool, allocator) on two episodes
    warm = time.time() - t0
    # a step of sample_B episodes costs ~ warm * sample_B / wb: if one such step would blow the budget, time the small batch instead
    if warm * sample_B / wb > 45.0:
        sample_B = wb
        leaked, real, si, z = leaked[:wb], real[:wb], si[:wb], z[:wb]
    t0 = time.time()
    steps = 0
    while steps < 1 or (time.time() - t0 < 10.0 and steps < 40):   # about 10-30 s of CPU work
        otr.step(leaked, real, si, z)
        steps += 1
    dt = time.time() - t0
    return {"value": round(sample_B * steps / dt, 4), "unit": "episodes/s", "cores": cores, "kind": "port",
            "sample": "%d timed step(s) (%.1f s, after a %d-episode warm-up step) of B=%d episodes of the same workload, torch-CPU eager "
                      "fp32, %d threads (%s)" % (steps, dt, wb, sample_B, cores, how)}


def host_cores():
    """Threads the CPU baseline may use: the cgroup CPU quota of this container when there is one (a GPU box gives one GPU a share
    of the host whatever the affinity mask says: more threads only oversubscribe), else the affinity mask."""
    aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            q = max(1, int(int(quota) / int(period)))
            return min(aff, q), "cgroup cpu.max %s/%s, affinity mask %d" % (quota, period, aff)
    except (OSError, ValueError):
        pass
    cap = int(os.environ.get("GIM_CPU_BASELINE_THREADS", "0"))
    if cap > 0:
        return min(aff, cap), "GIM_CPU_BASELINE_THREADS=%d, affinity mask %d" % (cap, aff)
    return min(aff, 64), "affinity mask %d, no cgroup quota" % aff


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="vox64", choices=sorted(UNIT))
    ap.add_argument("--batch", type=int, default=0, help="episodes per GPU (default: 16 for vox64, 32 for om32, 2 for vox128)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-bench", action="store_true")
    ap.add_argument("--no-traffic", action="store_true", help="skip the two rocprofv3 PMC passes (roofline.traffic = null)")
    ap.add_argument("--no-bf16x3", action="store_true", help="accepted and ignored (round-2 scripts): the bf16x3 matrix path was removed in round 3")
    ap.add_argument("--reg-param", type=float, default=0.0, help="R1 weight (BASELINE's metric is quoted at 0; 10 = the paper's VoxCeleb2 setting)")
    ap.add_argument("--matrix-path", default="fp32", choices=["fp32", "fp16"],
                    help="fp16: the opt-in 16-bit operand path (BASELINE config 5 'fp16 MFMA'; v_mfma_f32_32x32x16_f16, fp32 accumulate, fp32 master "
                         "weights; eligible convolutions only).  BASELINE's metric and every parity bound are stated for fp32, the default")
    ap.add_argument("--dry-run", action="store_true", help="rendezvous only: argument parsing, process group, one all-reduce, JSON line")
    ap.add_argument("--inner", action="store_true", help="(used by the traffic passes) run the steps and print nothing else")
    args = ap.parse_args()
    u = UNIT[args.workload]
    m, n, k = u.get("mnk", (1, 5, 10))
    args.batch = args.batch or u.get("batch", 16 if args.workload == "vox64" else 32)
    B = args.batch

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "WORLD_SIZE=%d but --gpus %d" % (world, args.gpus)
    # a core set per rank, before anything touches the GPU (the runtime's helper threads inherit the mask)
    from optimalstrategiesagainstgenerativeattacks_amd.training_utils import pin_rank_to_cores
    pinned = pin_rank_to_cores(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))
    # rehearsal knobs for a 1-GPU box / the CPU tests: GIM_BENCH_BACKEND=gloo; GIM_BENCH_ONE_DEVICE=1 runs all ranks on cuda:0
    backend = os.environ.get("GIM_BENCH_BACKEND", "nccl")

    if args.dry_run:
        if world > 1:
            dist.init_process_group(backend if backend != "nccl" or torch.cuda.is_available() else "gloo", rank=rank, world_size=world)
            t = torch.ones(1) * (rank + 1)
            dist.all_reduce(t)
            assert float(t) == world * (world + 1) / 2
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "rccl_ranks": dist.get_world_size() if world > 1 else 1,
                              "backend": dist.get_backend() if world > 1 else None, "steps": args.steps, "warmup": args.warmup,
                              "config": {"workload": args.workload, "global_batch": B * world}}))
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    # HBM traffic first: the profiled runs are child processes and this process has not touched the GPU yet
    traffic, traffic_note = None, "not measured"
    if world == 1 and not (args.no_traffic or args.inner or os.environ.get("GIM_BENCH_NO_TRAFFIC")):
        traffic, traffic_note = measure_hbm_traffic(args)
        log("HBM traffic: %s" % (traffic["hbm_bytes_per_step"] if traffic else traffic_note))

    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    if os.environ.get("GIM_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1 or os.environ.get("GIM_FORCE_ALLREDUCE"):   # the latter: one-rank RCCL rehearsal (needs MASTER_ADDR/PORT)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    G, tr = build_trainer(u["S"], u["C"], n, m, k, device, reg_param=args.reg_param)
    trainer = G.EpisodeParallel(tr)
    trainer.broadcast_parameters()
    leaked, real, si = synthetic_batch(B, m, n, k, u["C"], u["S"], device, 1234 + rank)
    zgen = torch.Generator(device=device).manual_seed(4321 + rank)

    # do the engine's streams run concurrently here (one hardware queue each)?  warns loudly when they do not
    streams_check = G.stream_concurrency_check(device)
    log("stream self-check: %s" % streams_check)

    def step():
        z = torch.randn((B, n, 512), device=device, generator=zgen)
        tr.do_global_step()
        tr.update_learning_rate()
        return G.gim_step(trainer, leaked, real, si, z=z, defer_join=True)   # joined by the next step / the final synchronize

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    import warnings
    warnings.filterwarnings("ignore")
    from optimalstrategiesagainstgenerativeattacks_amd import ops as _ops
    _ops.set_matrix_path(args.matrix_path)

    def timed(label):
        """W warm-up steps, then EXACTLY K steps between barrier + synchronize;
        (seconds [max over ranks], device ms of the region, per-step device ms on the caller's stream, last out)."""
        # the warm-up steps run the way the timed ones do - enqueued back to back, next iteration under the tail of the previous
        # one - so that the allocator pools, the lane streams and (multi-GPU) the communicator have seen that pattern before the
        # timed region starts; one synchronisation at their end
        for i in range(args.warmup):
            out = step()
        _ops.join_lanes()
        torch.cuda.synchronize()
        log("%s: %d warm-up steps done" % (label, args.warmup))
        gc.collect()   # start the timed region with an empty young generation (a full collection mid-region stalls the enqueue thread)
        tr.impersonator_opt.allreduce_timing, tr.authenticator_opt.allreduce_timing = [], []
        fence()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        cur = torch.cuda.current_stream()
        t0 = time.time()
        c0 = os.times()
        ev0.record()
        marks = []
        for _ in range(args.steps):
            out = step()
            marks.append(cur.record_event(torch.cuda.Event(enable_timing=True)))   # no synchronisation: one event record per step
        _ops.join_lanes()   # the last discriminator step runs on its own stream: ev1 must come after it
        ev1.record()
        t_enq = time.time() - t0     # the host has enqueued every step (it runs ahead of the device unless it is the bound)
        fence()
        dt = time.time() - t0
        c1 = os.times()
        host = {"enqueue_ms_per_step": round(t_enq / args.steps * 1e3, 3),
                "cpu_ms_per_step": round((c1.user - c0.user + c1.system - c0.system) / args.steps * 1e3, 3)}
        dev_ms = ev0.elapsed_time(ev1)
        ts = [ev0.elapsed_time(mk) for mk in marks]
        per_step = [b - a for a, b in zip([0.0] + ts[:-1], ts)]
        log("%s: timed region done: %.3f s for %d steps" % (label, dt, args.steps))
        # the gradient all-reduces as the step saw them (events on the stream that issued them: the generator's on the caller's
        # stream right after its backward, under the discriminator step; the discriminator's on lane 1, under the next generator forward)
        ar = {}
        for name, opt in (("g", tr.impersonator_opt), ("d", tr.authenticator_opt)):
            ts_ = [a.elapsed_time(b) for a, b in (opt.allreduce_timing or [])]
            ar[name] = round(sum(ts_) / len(ts_), 4) if ts_ else None
            ar[name + "_bucket_bytes"] = int(opt.flat_g.numel()) * 4 if getattr(opt, "_built", False) else None
            opt.allreduce_timing = None
        ranks_dt = [dt]
        if world > 1:
            tall = [torch.zeros(1, device=device, dtype=torch.float64) for _ in range(world)]
            dist.all_gather(tall, torch.tensor([dt], device=device, dtype=torch.float64))
            ranks_dt = [float(t_.item()) for t_ in tall]
            dt = max(ranks_dt)
        return dt, dev_ms, per_step, out, ar, ranks_dt, host

    if args.inner:   # a traffic pass: the steps only
        for _ in range(args.warmup + args.steps):
            step()
        _ops.join_lanes()
        fence()
        return

    log("models built; counting the executed work of one step")
    with _ops.count_flops() as counts:   # one extra (un-timed) step with the per-launch accounting switched on
        step()
        _ops.join_lanes()
        torch.cuda.synchronize()
    exe_gf, algo_launch_gf, per_conv = summarize_flops(counts, _ops)

    matrix_path = "fp32 MFMA" if args.matrix_path == "fp32" else "fp16 operands / fp32 accumulate on eligible convolutions (v_mfma_f32_32x32x16_f16), fp32 MFMA elsewhere"
    dt, dev_ms, per_step, out, allreduce_ms, ranks_dt, host_ms = timed(matrix_path)
    g_loss, d_loss = float(out[0][0]), float(out[1][0])

    if rank == 0:
        eps = B * world * args.steps / dt
        med = sorted(per_step)[len(per_step) // 2] if len(per_step) % 2 else 0.5 * sum(sorted(per_step)[len(per_step) // 2 - 1:len(per_step) // 2 + 1])
        algo = algo_gflop_per_episode(args.workload, m, n, k)
        achieved = eps * algo / 1e3 / world            # ALGORITHMIC TFLOP/s per GPU (SURVEY.md 8(d) convention)
        executed = exe_gf / 1e3 / (dt / args.steps)     # EXECUTED TFLOP/s per GPU (this rank's launches)
        PEAK = PEAK_FP32_MFMA_TFLOPS if args.matrix_path == "fp32" else PEAK_FP16_MFMA_TFLOPS
        line = {
            "metric": "GIM episodes/sec (%dx%dx%d, m=%d n=%d k=%d)" % (u["S"], u["S"], u["C"], m, n, k),
            "value": round(eps, 3), "unit": "episodes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if args.matrix_path == "fp32" else "f16 operands, f32 accumulate / master weights", "data": "synthetic",
            "ms_per_step_median": round(med, 3), "value_at_median_step": round(B * world / med * 1e3, 3),
            "per_step_ms": [round(t, 2) for t in per_step],   # device time between consecutive steps on the caller's stream (rank 0)
            "rccl_ranks": dist.get_world_size() if dist.is_initialized() else 1,
            # multi-GPU attribution: per-rank step times of the same timed region (value uses the max), the gradient all-reduces as
            # the step sees them (ms per step; null without a communicator), the hardware-queue setting in force and the result of
            # the start-up spin test of the engine's streams, the cores this rank pinned itself to
            "ms_per_step_ranks": {"min": round(min(ranks_dt) / args.steps * 1e3, 3), "max": round(max(ranks_dt) / args.steps * 1e3, 3)},
            "allreduce_ms": allreduce_ms,
            "hw_queues": dict(G.hw_queues_state(), streams_check=streams_check),
            "cpu_affinity": pinned,
            # host side of the timed region (rank 0): wall time until the last step was enqueued (< ms_per_step: the host ran ahead of the
            # device) and CPU time of the process (all threads; includes the wait of the closing synchronisation)
            "host": host_ms,
            "config": {"workload": "%s: %dx%dx%d synthetic episodes, m=%d n=%d k=%d, %d episodes/GPU, style_dim=512, "
                                   "G step + D step + 2 Adam updates per step, reg_param=%g" % (args.workload, u["S"], u["S"], u["C"], m, n, k, B, args.reg_param),
                       "global_batch": B * world, "parallelism": "dp%d (episodes sharded, 1 RCCL all-reduce per optimizer step)" % world,
                       "backend": dist.get_backend() if dist.is_initialized() else None,
                       "launch": "eager", "matrix_path": matrix_path},
            "roofline": {"bound": "mfma", "achieved": round(executed, 3), "peak": PEAK, "unit": "TFLOP/s",
                         "frac": round(executed / PEAK, 4),
                         "what": "achieved / frac: FLOPs the kernels EXECUTE per second (every conv / linear / batched-GEMM launch of one step "
                                 "counted per launch; the pool / sub-pixel folds run (K+1)^2 taps at a quarter of the pixels) over the fp32 MFMA "
                                 "peak = the matrix-pipe utilisation of the step (<= 1 by construction).  algo_tflops / algo_frac: the "
                                 "ALGORITHMIC figure of SURVEY.md 8(d) (the unfused reference ops at full resolution) - it counts taps the "
                                 "folded kernels never execute and can exceed 1; it is not a utilisation",
                         "algo_gflop_per_episode": round(algo, 1), "algo_gflop_per_step": round(algo * B, 1),
                         "algo_tflops": round(achieved, 3), "algo_frac": round(achieved / PEAK, 4),
                         "executed_gflop_per_step": round(exe_gf, 1), "executed_tflops": round(executed, 3),
                         "executed_frac": round(executed / PEAK, 4),
                         "launched_ops_at_full_resolution_gflop_per_step": round(algo_launch_gf, 1),
                         "traffic": traffic["hbm_bytes_per_step"] if traffic else None,
                         "traffic_unit": "HBM bytes/step", "traffic_detail": traffic if traffic else traffic_note,
                         "device_ms_per_step": round(dev_ms / args.steps, 3)},
            "final_losses": {"g": round(g_loss, 5), "d": round(d_loss, 5)},
        }
        if not args.no_kernel_bench:
            line["dominant_kernel"] = dominant_kernel_roofline(per_conv, device)
            log("kernel microbenchmark done")
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.workload, m, n, k, sample_B=B)
            log("cpu baseline done")
            line["cpu_baseline"]["gpu_over_cpu"] = round(eps / line["cpu_baseline"]["value"], 1)
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
