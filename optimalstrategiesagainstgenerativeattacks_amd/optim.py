"""Fused multi-tensor Adam on flat buffers + the data-parallel gradient exchange.

Replaces the two ``torch.optim.Adam`` instances of the reference trainer
(training/gim_img_trainer.py:50-58) and the gradient reduce-add of ``nn.DataParallel``
(training/gim_img_training.py:409):

  * all parameters of an optimizer live in ONE flat fp32 buffer (parameters become views into it; conv
    weights keep their channels-last storage), gradients in a second flat buffer, Adam moments in two more;
  * ``step()`` = [one RCCL all-reduce of the flat gradient buffer when torch.distributed is initialised]
    + ONE kernel launch (gim_adam_step) with per-parameter-group learning rates read from device memory;
  * ``zero_grad()`` = one memset of the flat gradient buffer.

``state_dict()`` / ``load_state_dict()`` keep torch.optim.Adam's format (per-parameter ``step`` /
``exp_avg`` / ``exp_avg_sq``; group keys lr, betas, eps, weight_decay, amsgrad), so reference checkpoints load.
"""
import os

import torch
import torch.distributed as dist

from . import _lib
from ._lib import check


def _phys(t):
    """Memory-order view of a parameter-like tensor (conv weights are stored channels-last)."""
    if t.dim() == 4:
        return t.permute(0, 2, 3, 1)
    return t


def _view_like(flat_slice, ref):
    """View of a flat slice with the logical shape of ``ref`` (channels-last strides for 4-D)."""
    if ref.dim() == 4:
        co, ci, kh, kw = ref.shape
        return flat_slice.view(co, kh, kw, ci).permute(0, 3, 1, 2)
    return flat_slice.view(ref.shape)


def weights_epoch(p):
    """Number of FusedAdam updates applied to parameter ``p``.  The Adam kernel writes parameters through raw
    pointers, which torch's version counters do not see; caches derived from a weight (folded conv weights) key on
    (p._version, weights_epoch(p))."""
    return getattr(p, "_gim_epoch", 0)


def _pad(n, align=64):
    return (n + align - 1) // align * align


# rehearsal switch: issue the collective even with one rank (exercises the RCCL call path on a one-GPU box)
_FORCE_ALLREDUCE = os.environ.get("GIM_FORCE_ALLREDUCE") is not None


def all_reduce_grads_(flat_g, timing=None):
    """Data-parallel gradient exchange: ONE all-reduce(sum) of a flat gradient bucket over RCCL/xGMI (gloo in
    the CPU tests).  Each rank's bucket holds the gradient of its local mean loss, so the global-batch
    gradient is the returned scale (1/world_size) times the reduced bucket; the scale is folded into the
    Adam kernel.  No-op (scale 1) when torch.distributed is not initialised.
    timing: a list that receives (event before, event after) on the caller's stream per collective (bench.py: `allreduce_ms`;
    the events bracket the collective as the STEP sees it - RCCL runs it on its own stream and the caller's stream waits for it)."""
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _FORCE_ALLREDUCE):
        if timing is not None and flat_g.is_cuda:
            cur = torch.cuda.current_stream()
            e0 = cur.record_event(torch.cuda.Event(enable_timing=True))
            dist.all_reduce(flat_g, op=dist.ReduceOp.SUM)
            timing.append((e0, cur.record_event(torch.cuda.Event(enable_timing=True))))
        else:
            dist.all_reduce(flat_g, op=dist.ReduceOp.SUM)
        return 1.0 / dist.get_world_size()
    return 1.0


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False)
        super().__init__(params, defaults)
        self._built = False
        self._host_step = 0
        self._lr_cache = None
        self.grad_divisor = None  # set by dp: world size for the post-all-reduce average
        self.allreduce_timing = None   # a list: step() appends the (before, after) events of its gradient all-reduce (bench.py)

    # ------------------------------------------------------------------ flat storage
    def _all_params(self):
        return [p for g in self.param_groups for p in g["params"]]

    def _build(self):
        params = self._all_params()
        dev = params[0].device
        if dev.type != "cuda":
            raise RuntimeError("FusedAdam runs on the GPU only (parameters are on %s); there is no CPU fallback" % dev)
        # every tensor starts on a 256-byte boundary of the flat buffers: the conv kernels need 16-byte
        # aligned weights for their vector loads (the padding elements are zeros and stay zeros)
        total = sum(_pad(p.numel()) for p in params)
        self.flat_p = torch.zeros(total, device=dev, dtype=torch.float32)
        self.flat_g = torch.zeros(total, device=dev, dtype=torch.float32)
        self.flat_m = torch.zeros(total, device=dev, dtype=torch.float32)
        self.flat_v = torch.zeros(total, device=dev, dtype=torch.float32)
        self._offsets = {}
        seg_end = []
        o = 0
        with torch.no_grad():
            for g in self.param_groups:
                for p in g["params"]:
                    n = p.numel()
                    self.flat_p[o:o + n].copy_(_phys(p.data).reshape(-1))
                    old_grad = p.grad
                    p.data = _view_like(self.flat_p[o:o + n], p)
                    gv = _view_like(self.flat_g[o:o + n], p)
                    if old_grad is not None:
                        gv.copy_(old_grad)
                    p.grad = gv
                    st = self.state.get(p)
                    if st:  # state imported by load_state_dict before the buffers existed
                        _view_like(self.flat_m[o:o + n], p).copy_(st["exp_avg"])
                        _view_like(self.flat_v[o:o + n], p).copy_(st["exp_avg_sq"])
                        self._host_step = max(self._host_step, int(st["step"]))
                    self._offsets[p] = (o, n)
                    o += _pad(n)
                seg_end.append(o)
        self._seg_end = torch.tensor(seg_end, dtype=torch.int64, device=dev)
        self._lr_dev = torch.zeros(len(seg_end), dtype=torch.float32, device=dev)
        self._step_dev = torch.full((1,), self._host_step, dtype=torch.int32, device=dev)
        self._built = True
        self._publish_state()

    def _publish_state(self):
        """Expose the flat moments through ``self.state`` in torch.optim.Adam's per-parameter format."""
        for p, (o, n) in self._offsets.items():
            self.state[p] = {
                "step": torch.tensor(float(self._host_step)),
                "exp_avg": _view_like(self.flat_m[o:o + n], p),
                "exp_avg_sq": _view_like(self.flat_v[o:o + n], p),
            }

    def _ensure(self):
        if not self._built:
            self._build()
            return
        for p, (o, n) in self._offsets.items():
            if p.data_ptr() != self.flat_p.data_ptr() + 4 * o:  # e.g. module.to() / load after build
                self._built = False
                self._build()
                return

    def _sync_grads(self):
        """Gradients normally accumulate in place into the flat buffer; re-home any that did not."""
        base = self.flat_g.data_ptr()
        for p, (o, n) in self._offsets.items():
            g = p.grad
            if g is None:
                p.grad = _view_like(self.flat_g[o:o + n], p)
            elif g.data_ptr() != base + 4 * o:
                gv = _view_like(self.flat_g[o:o + n], p)
                gv.copy_(g)
                p.grad = gv

    # ------------------------------------------------------------------ optimizer protocol
    def zero_grad(self, set_to_none=False):
        self._ensure()
        from . import ops
        ops.reset_wgrad_queues()
        self.flat_g.zero_()
        self._sync_grads()

    def flat_grad(self):
        self._ensure()
        self._sync_grads()
        return self.flat_g

    def _push_lrs(self):
        lrs = [float(g["lr"]) for g in self.param_groups]
        if lrs != self._lr_cache:
            self._lr_dev.copy_(torch.tensor(lrs, dtype=torch.float32), non_blocking=False)
            self._lr_cache = lrs

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0):
        """grad_scale multiplies the gradient bucket inside the Adam kernel (together with the 1 / world_size of the data-parallel
        average): 1 / S after a backward pass of S * loss (the loss scale of the fp16 matrix path, ops.loss_scale)."""
        assert closure is None
        lib = _lib.load()
        self._ensure()
        self._sync_grads()
        self._push_lrs()
        scale = all_reduce_grads_(self.flat_g, self.allreduce_timing)
        g0 = self.param_groups[0]
        for g in self.param_groups:
            assert g["betas"] == g0["betas"] and g["eps"] == g0["eps"], "one (betas, eps) per optimizer"
        check(lib.gim_adam_step(self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.flat_m.data_ptr(), self.flat_v.data_ptr(),
                                self.flat_p.numel(), self._seg_end.data_ptr(), self._lr_dev.data_ptr(), len(self.param_groups),
                                g0["betas"][0], g0["betas"][1], g0["eps"], scale * grad_scale, self._step_dev.data_ptr(),
                                torch.cuda.current_stream().cuda_stream), "adam_step")
        self.note_steps(1)

    def note_steps(self, k):
        """Host-side bookkeeping of k applied updates (called by step(); call it yourself after replaying a captured
        hipGraph that contains step(), since the replay does not run this Python code)."""
        self._host_step += k
        for p in self._offsets:
            p._gim_epoch = getattr(p, "_gim_epoch", 0) + k

    # ------------------------------------------------------------------ checkpoint format
    def state_dict(self):
        if self._built:
            self._publish_state()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        steps = [int(s["step"]) for s in self.state.values() if "step" in s]
        self._host_step = max(steps) if steps else 0
        if self._built:
            with torch.no_grad():
                for p, (o, n) in self._offsets.items():
                    st = self.state.get(p)
                    if st and "exp_avg" in st:
                        _view_like(self.flat_m[o:o + n], p).copy_(st["exp_avg"])
                        _view_like(self.flat_v[o:o + n], p).copy_(st["exp_avg_sq"])
                self._step_dev.fill_(self._host_step)
            self._publish_state()
        self._lr_cache = None
