"""GIM episodes/s on MI355X: one "step" = one training iteration of the hot path on one episode batch
(generator step + discriminator step, both Adam updates; training/gim_img_training.py:225-239) over synthetic
inputs already resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload vox64|om32] [--batch B] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task statement): whole-job episodes/s, the MFMA roofline
fraction of the step (ALGO FLOP/episode x episodes / time / fp32-MFMA peak), the dominant
kernel's own roofline from a HIP-event microbenchmark through the C ABI, and the CPU baseline (the oracle, a
port of the reference's path, timed on the host cores on a bounded sample).
"""
import argparse
import gc
import json
import os
import sys
import tempfile
import time

# Eight HIP hardware queues instead of the default four (read by the HIP runtime when it initialises): the engine's four
# streams, torch's copy streams and RCCL's then sit on queues of their own instead of aliasing by creation order
# (optimalstrategiesagainstgenerativeattacks_amd/gim_img_models.py, role -> stream map; profiles/r01_k_stream_map.txt)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
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


def measured_hbm_traffic(workload, B):
    """HBM bytes per step from the committed rocprofv3 PMC passes of this same command (FETCH_SIZE x2 gfx950
    correction + WRITE_SIZE; profiles/*hbm_traffic*.json).  bench.py cannot run the profiler on itself, so this
    is the last profiled value for the workload, or None."""
    path = os.path.join(ROOT, "profiles", "r01_k_hbm_traffic_%s_B%d.json" % (workload, B))
    try:
        with open(path) as f:
            return float(json.load(f)["hbm_bytes_per_step"])
    except (OSError, KeyError, ValueError):
        return None


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


def dominant_kernel_roofline(device):
    """HIP-event timing of the heaviest convolution of the workload through the C ABI: the encoder's
    64->64 3x3 conv at 64x64 on the D-step image count (320 images): fwd, dgrad, wgrad."""
    from optimalstrategiesagainstgenerativeattacks_amd import _lib
    lib = _lib.load()
    N, H, Cin, Cout, K = 320, 64, 64, 64, 3
    sh = _lib.GimConvShape(N, H, H, Cin, Cout, K, 0, 0.2)
    x = torch.randn(N, H, H, Cin, device=device)
    w = torch.randn(Cout, K, K, Cin, device=device) * 0.05
    y = torch.empty(N, H, H, Cout, device=device)
    dx = torch.empty_like(x)
    ns = lib.gim_conv2d_wgrad_slabs(sh)
    slabs = torch.empty(ns * Cout * K * K * Cin, device=device)
    st = torch.cuda.current_stream().cuda_stream
    flops = 2.0 * N * H * H * Cout * Cin * K * K
    out = {}
    x3 = lib.gim_conv_precision(-1) == 1
    dgrad = lambda: lib.gim_conv2d_dgrad(y.data_ptr(), w.data_ptr(), None, None, dx.data_ptr(), sh, st)   # noqa: E731
    if x3:   # the engine's dgrad then runs the k-contiguous kernel on cached transposed weights (ops._conv_dgrad)
        wt = torch.empty(Cin * K * K * Cout, device=device)
        lib.gim_conv2d_transpose_weights(w.data_ptr(), wt.data_ptr(), Cout, Cin, K, st)
        dgrad = lambda: lib.gim_conv2d_dgrad_t(y.data_ptr(), wt.data_ptr(), None, None, dx.data_ptr(), sh, st)   # noqa: E731
    for name, fn in (("fwd", lambda: lib.gim_conv2d_fwd(x.data_ptr(), w.data_ptr(), None, None, None, y.data_ptr(), sh, st)),
                     ("dgrad", dgrad),
                     ("wgrad", lambda: lib.gim_conv2d_wgrad(y.data_ptr(), x.data_ptr(), slabs.data_ptr(), None, ns, sh, st))):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        reps = 10
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        out[name] = {"ms": round(ms, 4), "tflops": round(flops / ms / 1e9, 2)}
    return {"kernel": "conv_igemm 64->64 3x3 @64x64 x320 img (%s)" % ("fwd / dgrad: bf16x3 split on the bf16 MFMA, wgrad: fp32 MFMA" if x3 else "fp32 MFMA"), "gflop_per_launch": round(flops / 1e9, 2), **out,
            "frac_fwd": round(out["fwd"]["tflops"] / PEAK_FP32_MFMA_TFLOPS, 4)}


def cpu_baseline(workload, m, n, k, sample_B):
    """The oracle (CPU port of the reference's path, parity-pinned by tests/golden) on a bounded sample."""
    from oracle import gim_oracle as go
    import optimalstrategiesagainstgenerativeattacks_amd as G
    u = UNIT[workload]
    # the GPU box gives one GPU a 16-CPU share whatever the affinity mask says: more threads only oversubscribe
    cores = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count())
    torch.set_num_threads(cores)
    torch.manual_seed(1)
    au, im = G.get_au(u["S"], u["C"], 512), G.get_im(u["S"], u["C"], 512)
    au_sd = {kk: v.detach().clone().contiguous() for kk, v in au.state_dict().items()}
    im_sd = {kk: v.detach().clone().contiguous() for kk, v in im.state_dict().items()}
    otr = go.OracleTrainer(au_sd, im_sd, n, 1e-4, 1e-4, 1e-6)
    leaked, real, si = synthetic_batch(sample_B, m, n, k, u["C"], u["S"], "cpu", 1234)
    z = torch.randn(sample_B, n, 512)
    otr.step(leaked, real, si, z)  # warm-up
    t0 = time.time()
    steps = 0
    while steps < 3 or (time.time() - t0 < 10.0 and steps < 40):   # about 10 s of CPU work
        otr.step(leaked, real, si, z)
        steps += 1
    dt = time.time() - t0
    return {"value": round(sample_B * steps / dt, 4), "unit": "episodes/s", "cores": cores, "kind": "port",
            "sample": "%d timed steps (%.1f s, after 1 warm-up) of B=%d episodes of the same workload, torch-CPU eager fp32, %d threads"
                      % (steps, dt, sample_B, cores)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="vox64", choices=sorted(UNIT))
    ap.add_argument("--batch", type=int, default=0, help="episodes per GPU (default: 16 for vox64, 32 for om32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-bench", action="store_true")
    ap.add_argument("--no-bf16x3", action="store_true", help="skip the informational second measurement on the bf16x3 matrix path")
    ap.add_argument("--reg-param", type=float, default=0.0, help="R1 weight (BASELINE's metric is quoted at 0; 10 = the paper's VoxCeleb2 setting)")
    ap.add_argument("--graph", action="store_true", help="replay the step from a captured hipGraph")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus)
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    # rehearsal knobs for a 1-GPU box: GIM_BENCH_BACKEND=gloo GIM_BENCH_ONE_DEVICE=1 runs all ranks on cuda:0
    backend = os.environ.get("GIM_BENCH_BACKEND", "nccl")
    if os.environ.get("GIM_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1 or os.environ.get("GIM_FORCE_ALLREDUCE"):   # the latter: one-rank RCCL rehearsal (needs MASTER_ADDR/PORT)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    u = UNIT[args.workload]
    m, n, k = u.get("mnk", (1, 5, 10))
    B = args.batch or u.get("batch", 16 if args.workload == "vox64" else 32)
    G, tr = build_trainer(u["S"], u["C"], n, m, k, device, reg_param=args.reg_param)
    trainer = G.EpisodeParallel(tr)
    trainer.broadcast_parameters()
    leaked, real, si = synthetic_batch(B, m, n, k, u["C"], u["S"], device, 1234 + rank)
    zgen = torch.Generator(device=device).manual_seed(4321 + rank)

    graphed = None
    if args.graph:
        from optimalstrategiesagainstgenerativeattacks_amd.graph import GraphedGimStep
        graphed = GraphedGimStep(trainer, leaked, real, si, torch.randn((B, n, 512), device=device, generator=zgen))

    def step():
        z = torch.randn((B, n, 512), device=device, generator=zgen)
        tr.do_global_step()
        tr.update_learning_rate()
        if graphed is not None:
            return graphed(leaked, real, si, z)
        return G.gim_step(trainer, leaked, real, si, z=z, defer_join=True)   # joined by the next step / the final synchronize

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    import warnings
    warnings.filterwarnings("ignore")
    from optimalstrategiesagainstgenerativeattacks_amd import _lib as _glib
    from optimalstrategiesagainstgenerativeattacks_amd import ops as _ops

    def timed(label):
        """W warm-up steps, then EXACTLY K steps between barrier + synchronize; (seconds [max over ranks], device ms, last out)."""
        for i in range(args.warmup):
            out = step()
            torch.cuda.synchronize()
            log("%s: warm-up step %d done" % (label, i))
        gc.collect()   # start the timed region with an empty young generation (a full collection mid-region stalls the enqueue thread)
        if os.environ.get("GIM_BENCH_NOGC"):   # diagnostics: is a slow phase the garbage collector?
            gc.disable()
        fence()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.time()
        ev0.record()
        marks = []
        for _ in range(args.steps):
            out = step()
            if os.environ.get("GIM_BENCH_STEP_EVENTS"):   # diagnostics: where the caller's stream stands after every step
                marks.append(torch.cuda.current_stream().record_event(torch.cuda.Event(enable_timing=True)))
        _ops.join_lanes()   # the last discriminator step runs on its own stream: ev1 must come after it
        ev1.record()
        fence()
        if marks:
            ts = [ev0.elapsed_time(m) for m in marks]
            log("%s: per-step ms on the caller's stream: %s" % (label, " ".join("%.1f" % (b - a) for a, b in zip([0.0] + ts[:-1], ts))))
        dt = time.time() - t0
        dev_ms = ev0.elapsed_time(ev1)
        log("%s: timed region done: %.3f s for %d steps" % (label, dt, args.steps))
        if world > 1:
            tmax = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        return dt, dev_ms, out

    log("models built; warm-up")
    matrix_path = "bf16x3" if _glib.load().gim_conv_precision(-1) == 1 else "fp32 MFMA"
    dt, dev_ms, out = timed(matrix_path)
    # Informational second measurement, never `value`: the same K steps with the forward / dgrad contractions on the bf16
    # matrix pipe by exact 3-way operand splitting (gim_conv_precision(1): fp32-level error, tests/test_gpu_bf16x3.py)
    x3 = None
    if matrix_path == "fp32 MFMA" and not args.no_bf16x3 and graphed is None and world == 1:
        _glib.load().gim_conv_precision(1)
        dt3, _, _ = timed("bf16x3")
        _glib.load().gim_conv_precision(0)
        x3 = {"value": round(B * world * args.steps / dt3, 3), "unit": "episodes/s", "ms_per_step": round(dt3 / args.steps * 1e3, 3),
              "what": "same workload and steps with gim_conv_precision(1): conv / linear forward, dgrad and wgrad on the bf16 MFMA with every "
                      "fp32 operand split exactly into three bf16 (6 partial products, fp32 accumulate; error vs fp64 equal to the fp32 "
                      "MFMA's or lower on every layer: profiles/r01_k_bf16x3_accuracy_vs_fp64.txt; same parity tolerances: tests/test_gpu_bf16x3.py)"}
    g_loss, d_loss = float(out[0][0]), float(out[1][0])

    if rank == 0:
        eps = B * world * args.steps / dt
        algo = algo_gflop_per_episode(args.workload, m, n, k)
        achieved = eps * algo / 1e3 / world  # TFLOP/s per GPU
        line = {
            "metric": "GIM episodes/sec (%dx%dx%d, m=%d n=%d k=%d)" % (u["S"], u["S"], u["C"], m, n, k),
            "value": round(eps, 3), "unit": "episodes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %dx%dx%d synthetic episodes, m=%d n=%d k=%d, %d episodes/GPU, style_dim=512, "
                                   "G step + D step + 2 Adam updates per step, reg_param=%g" % (args.workload, u["S"], u["S"], u["C"], m, n, k, B, args.reg_param),
                       "global_batch": B * world, "parallelism": "dp%d (episodes sharded, 1 RCCL all-reduce per optimizer step)" % world,
                       "launch": "hipGraph replay" if args.graph else "eager", "matrix_path": matrix_path},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 3), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4),
                         "traffic": measured_hbm_traffic(args.workload, B) if world == 1 else None, "traffic_unit": "HBM bytes/step (rocprofv3 PMC, profiles/)",
                         "algo_gflop_per_episode": round(algo, 1), "device_ms_per_step": round(dev_ms / args.steps, 3)},
            "final_losses": {"g": round(g_loss, 5), "d": round(d_loss, 5)},
        }
        if x3 is not None:
            line["bf16x3_path"] = x3
        if not args.no_kernel_bench:
            line["dominant_kernel"] = dominant_kernel_roofline(device)
            log("kernel microbenchmark done")
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.workload, m, n, k, sample_B=2)
            log("cpu baseline done")
            line["cpu_baseline"]["gpu_over_cpu"] = round(eps / line["cpu_baseline"]["value"], 1)
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
