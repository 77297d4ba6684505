"""Device time of the two lanes of one iteration, each alone and together (HIP events on the caller's stream, 64x64x3 workload):
    generator step alone (im_train_step: G forward, D forward of the fake sample, backward through D into G, G's Adam),
    discriminator step alone (au_train_step), both back to back (sequential protocol), and the overlapped gim_step.
The generator lane alone is the floor of the overlapped iteration: the discriminator step can only hide under it.
    python tools/lane_times.py [episodes=16] [repeats=10]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev = torch.device("cuda:0")
u = bench.UNIT["vox64"]
m, n, k = 1, 5, 10
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
R = int(sys.argv[2]) if len(sys.argv) > 2 else 10
G, tr = bench.build_trainer(u["S"], u["C"], n, m, k, dev)
trainer = G.DataParallelMock(tr)
leaked, real, si = bench.synthetic_batch(B, m, n, k, u["C"], u["S"], dev, 1)


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    e[0].record()
    for i in range(reps):
        fn()
        e[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(e[i].elapsed_time(e[i + 1]) for i in range(reps))
    return ts[len(ts) // 2]


fake = [None]


def g_step():
    out = G.im_train_step(trainer, leaked, si)
    fake[0] = out[1] if isinstance(out, (tuple, list)) and len(out) > 1 and torch.is_tensor(out[1]) and out[1].dim() == 5 else fake[0]
    return out


def d_step():
    return G.au_train_step(trainer, real, fake[0], si)


out = g_step()
if fake[0] is None:   # the fake sample is whichever returned tensor has the sample's shape
    for t in out:
        if torch.is_tensor(t) and t.dim() == 5:
            fake[0] = t
assert fake[0] is not None
print("GIM_WGRAD_STREAM=%r GIM_STREAM_MAP=%r  episodes %d" % (os.environ.get("GIM_WGRAD_STREAM", ""), os.environ.get("GIM_STREAM_MAP", ""), B))
tg = timed(g_step, R)
td = timed(d_step, R)
ts = timed(lambda: G.gim_step(trainer, leaked, real, si, overlap=False), R)
to = timed(lambda: G.gim_step(trainer, leaked, real, si), R)
tj = timed(lambda: G.gim_step(trainer, leaked, real, si, defer_join=True), R)
G.ops.join_lanes()
torch.cuda.synchronize()
print("generator step alone %.2f ms | discriminator step alone %.2f ms | back to back %.2f ms | overlapped %.2f ms | overlapped, deferred join %.2f ms"
      % (tg, td, ts, to, tj))
