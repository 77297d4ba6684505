"""Launch ONE convolution shape repeatedly through the C ABI (for rocprofv3 --pmc / --kernel-trace runs).

    python tools/kernel_probe.py fwd|dgrad|wgrad N H Cin Cout K [ups] [reps] [pool] [0] [tile] [ksplit|wgrad target]

pool = 1 (or ups = 1 with K > 1): the folded form the engine launches for that layer.  (argument 10 is unused: it selected the removed bf16x3 path).  tile / ksplit:
launch overrides (gim_conv_shape.tune_*; 0 = table / heuristic, tile < 0 = heuristic only).
"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from optimalstrategiesagainstgenerativeattacks_amd import _lib, ops  # noqa: E402


def main():
    kind = sys.argv[1]
    N, H, Cin, Cout, K = [int(a) for a in sys.argv[2:7]]
    arg = lambda i, d: int(sys.argv[i]) if len(sys.argv) > i else d   # noqa: E731
    ups, reps, pool, prec, tile, ks = arg(7, 0), arg(8, 5), arg(9, 0), arg(10, 0), arg(11, 0), arg(12, 0)
    fold = 1 if (pool or (ups and K > 1)) else 0
    dev = torch.device("cuda:0")
    lib = _lib.load()
    slope = float(os.environ.get("PROBE_SLOPE", "0.2"))   # pre-activation slope (1.0 = none)
    sh = _lib.GimConvShape(N, H, H, Cin, Cout, K, ups, slope, pool, fold, 0, tile, 0 if kind == "wgrad" else ks, ks if kind == "wgrad" else 0)
    KF = K + 1 if fold else K
    x = torch.randn(N, H >> ups, H >> ups, Cin, device=dev)
    w = torch.randn(Cout, KF, KF, Cin, device=dev) * 0.05
    y = torch.randn(N, H >> pool, H >> pool, Cout, device=dev)
    lo = 1 if (ups and fold) else 0
    dx = torch.empty(N, H >> lo, H >> lo, Cin, device=dev)
    acc = torch.zeros(Cout * KF * KF * Cin, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    dgrad = lambda: lib.gim_conv2d_dgrad(y.data_ptr(), w.data_ptr(), None, None, dx.data_ptr(), sh, st)   # noqa: E731
    plan_kind = {"fwd": 0, "dgrad": 1, "wgrad": 3}[kind]
    if Cin <= 8 and Cout % 16 == 0 and not (ups and not fold):   # image gradient: dgrad on transposed weights (ops._conv_dgrad)
        wt = torch.empty(Cin * KF * KF * Cout, device=dev)
        lib.gim_conv2d_transpose_weights(w.data_ptr(), wt.data_ptr(), Cout, Cin, KF, st)
        dgrad = lambda: lib.gim_conv2d_dgrad_t(y.data_ptr(), wt.data_ptr(), None, None, dx.data_ptr(), sh, st)   # noqa: E731
        plan_kind = 2 if kind == "dgrad" else plan_kind
    fwd = lambda: lib.gim_conv2d_fwd(x.data_ptr(), w.data_ptr(), None, None, None, y.data_ptr(), sh, st)   # noqa: E731
    wgrad = lambda: lib.gim_conv2d_wgrad_acc(y.data_ptr(), x.data_ptr(), acc.data_ptr(), None, sh, st)   # noqa: E731
    plain = not (ups or pool or fold)
    J = ops._xfold_factor(Cin, H) if (K >= 3 and plain and Cout % 16 == 0 and ops._NARROW_XFOLD) else 0
    if J:   # as ops._conv_dgrad: the x-folded gradient w.r.t. images
        wx = torch.empty(J * Cin * K * (K + J - 1) * Cout, device=dev)
        lib.gim_conv2d_xfold_weights(w.data_ptr(), wx.data_ptr(), Cout, Cin, K, J, st)
        dgrad = lambda: lib.gim_conv2d_dgrad_xfold(y.data_ptr(), wx.data_ptr(), None, None, dx.data_ptr(), sh, J, st)   # noqa: E731
    if ops._ROWS_FORM and K >= 3 and Cin <= 8 and K * Cin <= 64 and Cout >= 16 and Cout % 4 == 0 and plain and tile == 0:
        # as ops.ConvFn: row-contiguous K on a zero-padded, activated copy of the image (the copy is part of the forward)
        pad, CaP = (K - 1) // 2, (K * Cin + 15) & ~15
        xp = torch.empty(N, H + 2 * pad, H + 2 * pad, Cin, device=dev)
        wr = torch.empty(Cout * K * CaP, device=dev)
        lib.gim_conv2d_pack_rows_weights(w.data_ptr(), wr.data_ptr(), Cout, Cin, K, st)
        shr = _lib.GimConvShape(N, H, H, Cin, Cout, K, 0, 1.0, 0, 0, 0, 0, 1)
        accr = torch.zeros(Cout * K * CaP, device=dev)

        def fwd():
            lib.gim_pad_image(x.data_ptr(), xp.data_ptr(), N, H, H, Cin, pad, slope, st)
            lib.gim_conv2d_fwd_rows(xp.data_ptr(), wr.data_ptr(), None, None, None, y.data_ptr(), shr, st)
        lib.gim_pad_image(x.data_ptr(), xp.data_ptr(), N, H, H, Cin, pad, slope, st)
        wgrad = lambda: lib.gim_conv2d_wgrad_rows_acc(y.data_ptr(), xp.data_ptr(), accr.data_ptr(), None, shr, st)   # noqa: E731
    fn = {"fwd": fwd, "dgrad": dgrad, "wgrad": wgrad}[kind]
    plan = (ctypes.c_int32 * 8)()
    lib.gim_conv_launch_plan(sh, plan_kind, ctypes.cast(plan, ctypes.c_void_p))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    flops = ops.conv_executed_flops(N, H, H, Cin, Cout, K, ups, pool, fold)
    print("%s N=%d H=%d Cin=%d Cout=%d K=%d ups=%d pool=%d prec=%d plan[table,BM,BN,split,gx,gy,gz,path]=%s: %.3f ms  %.1f executed TFLOP/s"
          % (kind, N, H, Cin, Cout, K, ups, pool, prec, list(plan), ms, flops / ms / 1e9))


if __name__ == "__main__":
    main()
