"""EXPERIMENT RECORD (not part of the product since round 4): hipGraph capture of one whole GIM training iteration
(generator step + discriminator step + both Adam updates).

Measured on MI355X, 64x64x3, 16 episodes (profiles/r03_final_bench_vox64_B16_graph.log): the captured
SEQUENTIAL protocol replays as fast as the same protocol launched eagerly (378 vs 380 episodes/s), i.e. 10 % slower than the
default two-lane eager step (420), and the two-lane capture replays its parallel branches worse than eager streams run them
(round 2: 328 vs 375).  The host keeps up with the device at this batch, so the graph has nothing to win; it was removed
from the package and is kept here so that the measurement can be repeated:
    python tools/graph_replay_experiment.py            # eager vs replay, one process


The eager step issues ~2000 kernel launches through Python (~35-60 ms of host time per step): fine while the GPU
needs longer than that (16 episodes per GPU), the bound once batches are small or kernels get faster.  A captured
step replays with one launch.  Everything the step touches is capture-safe by construction: the C ABI allocates
nothing and syncs nothing, Adam's step counter and learning rates live in device memory, the spectral-norm job
tables are static, the two encoder streams fork from / join into the capturing stream.
"""
import torch

import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optimalstrategiesagainstgenerativeattacks_amd.gim_img_training import gim_step  # noqa: E402


class GraphedGimStep:
    """Usage:
        gs = GraphedGimStep(trainer, leaked, real, si, z)     # shapes are frozen; runs warm-up steps, then captures
        im_out, au_out = gs(leaked, real, si, z)               # copies into the static inputs and replays
    The learning rates may change between replays (they are read from device memory; pushed before each replay).
    Outputs are static tensors overwritten by the next replay.  Parameters, Adam state, spectral-norm buffers are
    updated in place exactly as by the eager ``gim_step``; the host-side step counters are advanced here."""

    def __init__(self, trainer, leaked, real, si, z, warmup=3, overlap=False):
        """overlap: capture the two-lane protocol of gim_step (the discriminator step on its own stream next to the generator's
        backward) instead of the sequential one."""
        self.trainer = trainer
        self.mod = trainer.module
        self.static = [t.clone() for t in (leaked, real, si, z)]
        warmup = max(warmup, 3)  # the batched weight-gradient finish must have seen every job table it will capture
        # warm-up and capture must launch the same kernels: the grouped style projections address their outputs through job tables
        # built per allocation, so THIS trainer's image-to-image module runs them one by one (a per-module switch)
        self.mod.impersonator.img2img.group_style_linears = False
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                gim_step(trainer, *self.static[:3], z=self.static[3], overlap=overlap)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        for opt in (self.mod.impersonator_opt, self.mod.authenticator_opt):
            opt._push_lrs()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = gim_step(trainer, *self.static[:3], z=self.static[3], overlap=overlap)
        # the capture pass itself advanced the host-side counters once without running: undo nothing, but note
        # that parameters were NOT changed by the capture (kernels are only recorded)
        for opt in (self.mod.impersonator_opt, self.mod.authenticator_opt):
            opt.note_steps(-1)
        # With the counters restored, the caches of tensors derived from the weights (folded weights per conv, transposed weights
        # of the image-gradient dgrad) carry the keys of tensors that the capture only RECORDED, never computed: an eager forward before
        # the first replay would read them.  Drop them (the graph keeps its own references and recomputes them on every replay).
        from optimalstrategiesagainstgenerativeattacks_amd import model_blocks as mb
        from optimalstrategiesagainstgenerativeattacks_amd import ops
        for m_ in list(self.mod.authenticator.modules()) + list(self.mod.impersonator.modules()):
            if isinstance(m_, mb.SNConv2d):
                m_._fold_cache = (None, None, None, None, False)
        ops._WT_CACHE.clear()
        # The weight-gradient arenas were re-zeroed by the flush INSIDE the capture: their "zeroed" marks hold an event that was
        # recorded on a capturing stream only.  An eager backward on another stream would wait on that event from outside the
        # capture (undefined in HIP).  Re-mark every page eagerly (the arena is zero: nothing ran since the warm-up's own flush).
        torch.cuda.synchronize()
        for q in ops._QUEUES.values():
            for pg in q.pages:
                pg[2:] = q._zeroed_mark()

    def __call__(self, leaked, real, si, z):
        for dst, src in zip(self.static, (leaked, real, si, z)):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        for opt in (self.mod.impersonator_opt, self.mod.authenticator_opt):
            opt._push_lrs()
        self.graph.replay()
        for opt in (self.mod.impersonator_opt, self.mod.authenticator_opt):
            opt.note_steps(1)
        return self.out


if __name__ == "__main__":
    import tempfile
    import time
    import optimalstrategiesagainstgenerativeattacks_amd as G
    dev = torch.device("cuda:0")
    S, C, D, B, m, n, k = 64, 3, 512, 16, 1, 5, 10
    res = {}
    for mode in ("eager", "graph"):
        torch.manual_seed(1)
        au, im = G.get_au(S, C, D).to(dev), G.get_im(S, C, D).to(dev)
        with tempfile.TemporaryDirectory() as td:
            tr = G.GIMImgTrainer(td, m, n, k, au, im, 1e-4, 1e-4, 1e-6, reg_param=0.0)
        trainer = G.DataParallelMock(tr)
        g = torch.Generator(device=dev).manual_seed(1234)
        leaked, real, si = [torch.rand((B, t, C, S, S), device=dev, generator=g) * 2 - 1 for t in (m, n, k)]
        z = torch.randn((B, n, D), device=dev, generator=g)
        step = GraphedGimStep(trainer, leaked, real, si, z) if mode == "graph" else (lambda a, b, c, zz: gim_step(trainer, a, b, c, z=zz))
        for _ in range(5):
            step(leaked, real, si, z)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            step(leaked, real, si, z)
        torch.cuda.synchronize()
        res[mode] = B * 20 / (time.perf_counter() - t0)
        print("%s: %.1f episodes/s" % (mode, res[mode]), flush=True)
