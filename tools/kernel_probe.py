"""Launch ONE convolution shape repeatedly through the C ABI (for rocprofv3 --pmc / --kernel-trace runs).

    python tools/kernel_probe.py fwd|dgrad|wgrad N H Cin Cout K [ups] [reps]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from optimalstrategiesagainstgenerativeattacks_amd import _lib  # noqa: E402


def main():
    kind = sys.argv[1]
    N, H, Cin, Cout, K = [int(a) for a in sys.argv[2:7]]
    ups = int(sys.argv[7]) if len(sys.argv) > 7 else 0
    reps = int(sys.argv[8]) if len(sys.argv) > 8 else 5
    dev = torch.device("cuda:0")
    lib = _lib.load()
    sh = _lib.GimConvShape(N, H, H, Cin, Cout, K, ups, 0.2)
    x = torch.randn(N, H >> ups, H >> ups, Cin, device=dev)
    w = torch.randn(Cout, K, K, Cin, device=dev) * 0.05
    y = torch.randn(N, H, H, Cout, device=dev)
    dx = torch.empty(N, H, H, Cin, device=dev)
    ns = lib.gim_conv2d_wgrad_slabs(sh)
    slabs = torch.empty(ns * Cout * K * K * Cin, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    dgrad = lambda: lib.gim_conv2d_dgrad(y.data_ptr(), w.data_ptr(), None, None, dx.data_ptr(), sh, st)   # noqa: E731
    if lib.gim_conv_precision(-1) == 1 and Cout % 16 == 0 and Cin >= 32 and not ups:   # bf16x3 path: dgrad on transposed weights (ops._conv_dgrad)
        wt = torch.empty(Cin * K * K * Cout, device=dev)
        lib.gim_conv2d_transpose_weights(w.data_ptr(), wt.data_ptr(), Cout, Cin, K, st)
        dgrad = lambda: lib.gim_conv2d_dgrad_t(y.data_ptr(), wt.data_ptr(), None, None, dx.data_ptr(), sh, st)   # noqa: E731
    fn = {"fwd": lambda: lib.gim_conv2d_fwd(x.data_ptr(), w.data_ptr(), None, None, None, y.data_ptr(), sh, st),
          "dgrad": dgrad,
          "wgrad": lambda: lib.gim_conv2d_wgrad(y.data_ptr(), x.data_ptr(), slabs.data_ptr(), None, ns, sh, st)}[kind]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    flops = 2.0 * N * H * H * Cout * Cin * K * K
    print("%s N=%d H=%d Cin=%d Cout=%d K=%d ups=%d: %.3f ms  %.1f TFLOP/s" % (kind, N, H, Cin, Cout, K, ups, ms, flops / ms / 1e9))


if __name__ == "__main__":
    main()
