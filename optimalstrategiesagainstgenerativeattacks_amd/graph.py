"""hipGraph capture of one whole GIM training iteration (generator step + discriminator step + both Adam updates).

The eager step issues ~2000 kernel launches through Python (~35-60 ms of host time per step): fine while the GPU
needs longer than that (16 episodes per GPU), the bound once batches are small or kernels get faster.  A captured
step replays with one launch.  Everything the step touches is capture-safe by construction: the C ABI allocates
nothing and syncs nothing, Adam's step counter and learning rates live in device memory, the spectral-norm job
tables are static, the two encoder streams fork from / join into the capturing stream.
"""
import torch

from .gim_img_training import gim_step


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
        from . import gim_img_models
        gim_img_models.GROUP_STYLE_LINEARS[0] = False   # warm-up and capture must launch the same kernels (see AdaInImage2Image.forward)
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
        from . import model_blocks as mb
        from . import ops
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
