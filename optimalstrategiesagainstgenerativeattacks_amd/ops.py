"""Autograd operators over the C ABI of libgim_hip.so (include/gim_hip.h).

Every operator here launches hand-written HIP kernels on torch's current stream through ctypes;
torch supplies device memory, streams and the autograd tape only.  There is no CPU path: tensors
must be CUDA(HIP) float32.  Activations are NHWC.
"""
import collections
import ctypes
import os
import weakref

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib
from ._lib import GimConvShape, check

LRELU_SLOPE = 0.2

# Higher-order mode.  The only second-order quantity of the training path is the R1 regulariser
# (training/utils.py:115-124): the gradient of the authenticator's output w.r.t. its INPUT IMAGES, differentiated
# once more w.r.t. the parameters.  Inside `input_grad_only()` a backward that runs with create_graph=True returns
# differentiable input gradients (each first-order backward is itself a Function whose adjoint reuses the same
# kernels) and no parameter gradients; create_graph outside that context is refused rather than silently wrong.
_HIGHER = {"input_grad_only": False}


class input_grad_only:
    def __enter__(self):
        self.prev = _HIGHER["input_grad_only"]
        _HIGHER["input_grad_only"] = True

    def __exit__(self, *exc):
        _HIGHER["input_grad_only"] = self.prev


def _second_order():
    """True when this backward is being recorded (create_graph=True)."""
    if not torch.is_grad_enabled():
        return False
    if not _HIGHER["input_grad_only"]:
        raise NotImplementedError("create_graph=True is supported for input gradients only: wrap the autograd.grad call in "
                                  "ops.input_grad_only() (training_utils.compute_grad2 does)")
    return True


# Executed-work accounting (bench.py, tools/): inside `count_flops()` every convolution / linear / batched-GEMM launch adds
# the multiply-adds its kernel EXECUTES (the pool / sub-pixel folds run (K+1)^2 taps at a quarter of the pixels, not the
# K^2 full-resolution taps of the unfused reference op) to a Counter keyed by (kind, shape).  Off (None) otherwise.
_FLOPS = None
_PHASES = None     # phase_timeline(): a list that gim_step's phase marks are appended to


class phase_timeline:
    """with ops.phase_timeline() as marks: gim_step(...)  ->  marks = [(name, event), ...]: one timing event per phase boundary of the
    overlapped training step, recorded on the stream the phase runs on (lane 0 = the caller's stream, lane 1 = the discriminator's).
    bench.py --phases turns them into the step's timeline; nothing is recorded outside the context."""

    def __enter__(self):
        global _PHASES
        self.prev = _PHASES
        _PHASES = []
        return _PHASES

    def __exit__(self, *a):
        global _PHASES
        _PHASES = self.prev


def mark_phase(name):
    if _PHASES is not None:
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream())
        _PHASES.append((name, ev))


class count_flops:
    """with ops.count_flops() as c: ...   ->  c[(kind, cfg)] = [launches, executed FLOPs per launch]; kinds fwd / dgrad / wgrad / bgemm;
    cfg = (N, H, W, Cin, Cout, KH, ups, pool, fold) for convolutions and linears (H = W = KH = 1), (batch, M, N, K) for bgemm."""

    def __enter__(self):
        global _FLOPS
        self.prev = _FLOPS
        _FLOPS = {}
        return _FLOPS

    def __exit__(self, *exc):
        global _FLOPS
        _FLOPS = self.prev


def conv_executed_flops(N, H, W, Cin, Cout, KH, ups, pool, fold):
    """2 * MACs one forward (= one dgrad = one wgrad) launch of this convolution executes.  (H, W) is the resolution the
    unfused conv would run at (after the nearest upsample, before the average pool)."""
    if fold and pool:        # ONE stride-2 conv, (K+1)^2 taps, on the pooled grid
        return 2.0 * N * (H >> 1) * (W >> 1) * Cin * Cout * (KH + 1) ** 2
    if fold and ups:         # 4 parity classes on the low-resolution grid, ((K+1)/2)^2 taps each
        return 2.0 * N * H * W * Cin * Cout * ((KH + 1) // 2) ** 2
    return 2.0 * N * H * W * Cin * Cout * KH * KH


def conv_algorithmic_flops(N, H, W, Cin, Cout, KH, *_):
    """2 * MACs of the unfused reference op (F.conv2d at the full resolution): the SURVEY.md 8(d) convention."""
    return 2.0 * N * H * W * Cin * Cout * KH * KH


def _note_conv(kind, cfg, sh=None, plan_kind=None):
    """Count one conv launch (ops.count_flops).  sh / plan_kind (the launch's gim_conv_shape and its gim_conv_launch_plan kind): the
    library is asked which share of the launch's K steps its kernel skips (position-major rows on small maps skip padding taps,
    include/gim_hip.h gim_conv_launch_plan out[7]) - skipped multiply-adds are not executed FLOPs."""
    N, H, W, Cin, Cout, KH, ups, pre_slope, has_bias, has_res, pool, fold, res_ups = cfg
    key = (kind, (N, H, W, Cin, Cout, KH, int(ups), int(bool(pool)), int(fold)))
    ent = _FLOPS.get(key)
    if ent is None:
        share = 1.0
        if sh is not None:
            out = (ctypes.c_int32 * 8)()
            check(_lib.load().gim_conv_launch_plan(ctypes.byref(sh), plan_kind, ctypes.cast(out, ctypes.c_void_p)), "conv_launch_plan")
            share = 1.0 - (out[7] >> 8) / 1000.0
        _FLOPS[key] = [1, conv_executed_flops(*key[1]) * share]
    else:
        ent[0] += 1


def _stream():
    """Raw handle of the caller's current HIP stream.  (torch.cuda.current_stream() builds a Stream object through several
    layers of Python, ~4 us; with ~1400 kernel calls per step that was 6 ms of the ~35 ms the host needs per step.)"""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def _p(t, off=0):
    if t is None:
        return None
    return t.data_ptr() + 4 * off


def _req(t, name):
    if not (t.is_cuda and t.dtype == torch.float32):
        raise RuntimeError("%s must be a CUDA float32 tensor (got %s on %s): the GIM engine has no CPU path"
                           % (name, t.dtype, t.device))
    return t if t.is_contiguous() else t.contiguous()


def weight_phys(w):
    """The [Cout][KH][KW][Cin] storage view of a conv weight parameter (logical [Cout,Cin,KH,KW] kept
    channels-last), or the [out][in] linear weight itself.  The view is kept on the tensor object (keyed by its data pointer: a
    view follows in-place updates, only a re-allocated storage invalidates it) - building it costs 3 us, ~520 times per step."""
    try:
        c = w._gim_phys
        if c[0] == w.data_ptr():
            return c[1]
    except AttributeError:
        pass
    if w.dim() == 2:
        return w if w.is_contiguous() else w.contiguous()
    wp = w.permute(0, 2, 3, 1)
    if not wp.is_contiguous():
        return wp.contiguous()      # a copy: not cached (it would not follow updates of w)
    w._gim_phys = (w.data_ptr(), wp)
    return wp


# Deterministic weight-gradient combine (slabs + fixed-order reduce) instead of float atomics, for the non-queued path
_WGRAD_SLABS = os.environ.get("GIM_WGRAD_SLABS") is not None

# Deterministic mode (host switch GIM_DETERMINISTIC=1 at import, or set_deterministic()): the reference's path is reproducible
# from run to run (torch on the CPU, training/gim_img_training.py:157-183); the engine's default is not in its last bits, because
# three of its sums are combined with float atomics whose order varies: the K slices of forward / dgrad launches that split K,
# the pixel slices of a weight gradient (and the jobs of one gradient in the batched finish), the grouped style projections'
# input gradient.  Under the switch no launch splits K (tune_ksplit = 1 on every forward / dgrad shape), weight gradients go the
# non-queued way as slabs added in a fixed order (the GIM_WGRAD_SLABS path), and the style projections run one by one (autograd
# adds their gradients in graph order): two runs of one program are bit-identical, at ~25 % of the speed of the default
# (tests/test_gpu_models.py::test_deterministic_mode_is_bit_reproducible).
_DETERMINISTIC = [os.environ.get("GIM_DETERMINISTIC") is not None]


def deterministic():
    return _DETERMINISTIC[0]


def set_deterministic(flag):
    """Switch the deterministic mode (see above) on or off; returns the previous setting."""
    prev = _DETERMINISTIC[0]
    _DETERMINISTIC[0] = bool(flag)
    if prev != _DETERMINISTIC[0]:
        _SPLITS_K.clear()    # the cached "does this launch split K" answers were given for the other mode
    return prev


# Matrix path of the convolutions: "fp32" (default: fp32 operands on the fp32 MFMA - the reference's arithmetic, what every parity
# bound of north_star is stated for) or "fp16" (opt-in, BASELINE config 5 "fp16 MFMA"): operands rounded to fp16 when they are
# staged into LDS, v_mfma_f32_32x32x16_f16 with fp32 accumulation, fp32 master weights and fp32 activations in HBM
# (csrc/conv_f16.inc).  Host switch GIM_MATRIX_PATH=fp16 at import, or set_matrix_path(); launches that are not eligible (image
# layers with 1 / 3 / 6 channels, < 32 output channels, linears) stay on the fp32 MFMA.
_PREC = [1 if os.environ.get("GIM_MATRIX_PATH", "fp32") == "fp16" else 0]


def matrix_path():
    return "fp16" if _PREC[0] else "fp32"


# Loss scale of the fp16 matrix path.  Gradients of a mean-over-episodes loss reach the convolutions at 1e-3 ... 1e-7 per element;
# fp16 keeps 11 significant bits only down to 6.1e-5 (below that: subnormals, then zero).  The step functions (gim_img_training)
# therefore differentiate loss * S and the fused Adam kernel multiplies the gradient bucket by 1 / S (a power of two: exact in
# fp32, every backward operator is linear in the incoming gradient) - the standard mixed-precision recipe; not in the reference,
# which has no 16-bit path.  Too-large values saturate at +-65504 when they are rounded (conv_f16.inc).  1 on the fp32 path.
_LOSS_SCALE = [float(os.environ.get("GIM_FP16_LOSS_SCALE", "4096"))]


def loss_scale():
    return _LOSS_SCALE[0] if _PREC[0] else 1.0


def set_matrix_path(name):
    """"fp32" or "fp16" (see above); returns the previous setting."""
    if name not in ("fp32", "fp16"):
        raise ValueError("matrix path must be 'fp32' or 'fp16'")
    prev = matrix_path()
    _PREC[0] = 1 if name == "fp16" else 0
    if prev != name:
        _SPLITS_K.clear()    # the cached launch plans were made for the other kernels
    return prev


_SHAPES = {}   # argument tuple -> template struct (a 17-field ctypes constructor costs 1.3 us, a copy of a template 0.4; ~640 per step)


def _shape(N, H, W, Cin, Cout, KH, ups, pre_slope, pool=0, wfold=0, res_ups=0):
    key = (N, H, W, Cin, Cout, KH, ups, pre_slope, pool, wfold, res_ups, _DETERMINISTIC[0], _PREC[0])
    t = _SHAPES.get(key)
    if t is None:
        t = _SHAPES[key] = GimConvShape(N, H, W, Cin, Cout, KH, ups, pre_slope, pool, wfold, res_ups, 0, 1 if _DETERMINISTIC[0] else 0, 0, 0, 0.0,
                                        _PREC[0] if H * W > 1 else 0)   # (linears - 1 x 1 maps - stay fp32: tiny, and the head's logits are built there)
    return GimConvShape.from_buffer_copy(t)    # callers set tune_* / post_slope / out_zeroed on their own copy


# Launch overrides for tools/step_autotune.py (tuning the launch table against the time of the WHOLE overlapped step instead of
# each kernel alone): {(kind, (N, H, W, Cin, Cout, KH, ups, pool, fold)): (tune_tile, tune_ksplit | tune_wgrad)} with kind
# "fwd" / "dgrad" / "wgrad".  Empty in the product: the compiled-in table and the heuristics decide.
_TUNE_OVERRIDE = {}


def _tuned(sh, kind, key):
    ov = _TUNE_OVERRIDE.get((kind, key)) if _TUNE_OVERRIDE else None
    if ov is not None:
        sh.tune_tile = ov[0]
        if kind == "wgrad":
            sh.tune_wgrad = ov[1]
        else:
            sh.tune_ksplit = ov[1]
    return sh


# Outputs of split-K launches.  A layer whose output tiles do not fill the chip is sliced along K over the grid and its slices
# are combined with float atomics into a ZEROED output: ~250 such launches per training step, each with its own memset in front
# (a 7 us fill kernel plus a kernel boundary).  Instead those outputs are carved out of pages that ONE torch.zeros call clears
# (per stream, 64 MB at a time): views keep their page alive, so tensor lifetimes are the usual ones; whether a launch splits K
# is asked from the library once per shape (gim_conv_launch_plan) and cached.
_ZERO_POOL = os.environ.get("GIM_NO_ZERO_POOL") is None   # A/B switch (host side)
_SPLITS_K = {}
_ZERO_PAGES = {}
_ZERO_PAGE = 16 << 20   # floats per page (64 MB)


def _splits_k(sh, plan_kind, key):
    """Does this launch combine K slices with atomics (and so need a zeroed output)?"""
    k_ = (plan_kind,) + key
    v = _SPLITS_K.get(k_)
    if v is None:
        out = (ctypes.c_int32 * 8)()
        check(_lib.load().gim_conv_launch_plan(ctypes.byref(sh), plan_kind, ctypes.cast(out, ctypes.c_void_p)), "conv_launch_plan")
        v = _SPLITS_K[k_] = out[3] > 1
    return v


def _zeros_from_pool(shape, device):
    """A zero-filled float32 tensor of `shape`, cleared together with its neighbours by one fill per 64 MB page (pages are per
    (device, stream): the fill and the kernels that use the page are in stream order).  A view keeps its WHOLE page alive: outputs
    that live for one training step (saved activations) cost nothing extra, a tensor a caller keeps for long pins 64 MB - clone it."""
    n = 1
    for d in shape:
        n *= d
    na = (n + 63) & ~63
    if na > _ZERO_PAGE >> 1:     # a big output: its own fill (a view pins its whole page for as long as it lives)
        return torch.zeros(shape, device=device, dtype=torch.float32)
    raw = (torch._C._cuda_getDevice(), _stream())
    pg = _ZERO_PAGES.get(raw)
    if pg is None or pg[1] + na > pg[0].numel():
        pg = _ZERO_PAGES[raw] = [torch.zeros(_ZERO_PAGE, device=device, dtype=torch.float32), 0]
    v = pg[0][pg[1]:pg[1] + n].view(shape)
    pg[1] += na
    return v


def _conv_out(sh, plan_kind, key, shape, device):
    """Output buffer of a forward (plan_kind 0) / dgrad (1, 2) launch: zeros from the pool when the launch splits K (and
    sh.out_zeroed tells the library not to clear it again), plain torch.empty otherwise."""
    if _ZERO_POOL and _splits_k(sh, plan_kind, key) and not torch.cuda.is_current_stream_capturing():
        sh.out_zeroed = 1
        return _zeros_from_pool(shape, device)
    return torch.empty(shape, device=device, dtype=torch.float32)


# --------------------------------------------------------------------------------------------
# spectral norm power iteration (no autograd: u, v are constants for the gradient, see ConvFn.backward)
# --------------------------------------------------------------------------------------------
@torch.no_grad()
def spectral_sigma(w, u, v, training):
    """torch.nn.utils.spectral_norm semantics; returns (sigma[1], u_used[Cout], v_used[K])."""
    lib = _lib.load()
    wp = weight_phys(w)
    Cout, Cin = w.shape[0], w.shape[1]
    KH = w.shape[2] if w.dim() == 4 else 1
    K = Cin * KH * KH
    sigma = torch.empty(1, device=w.device, dtype=torch.float32)
    u_s = torch.empty_like(u)
    v_s = torch.empty_like(v)
    scratch = torch.empty(9 * K + Cout, device=w.device, dtype=torch.float32)
    check(lib.gim_spectral_sigma(_p(wp), _p(u), _p(v), _p(sigma), _p(u_s), _p(v_s), _p(scratch), Cout, Cin, KH,
                                 1 if training else 0, _stream()), "spectral_sigma")
    return sigma, u_s, v_s


def _folded(wp, Cout, Cin, KH):
    """(KH+1)^2-tap folded weights F (sum of the 2x2-shifted copies of W) for the pool / sub-pixel forms."""
    f = torch.empty(Cout * (KH + 1) * (KH + 1) * Cin, device=wp.device, dtype=torch.float32)
    check(_lib.load().gim_conv2d_fold_weights(_p(wp), _p(f), Cout, Cin, KH, _stream()), "fold_weights")
    return f


# --------------------------------------------------------------------------------------------
# deferred, batched weight-gradient finish
# --------------------------------------------------------------------------------------------
class _WgradJob(ctypes.Structure):
    _fields_ = [("src", ctypes.c_void_p), ("bias_src", ctypes.c_void_p), ("w", ctypes.c_void_p), ("sigma", ctypes.c_void_p),
                ("u", ctypes.c_void_p), ("v", ctypes.c_void_p), ("tmp", ctypes.c_void_p), ("partial", ctypes.c_void_p),
                ("grad_w", ctypes.c_void_p), ("grad_b", ctypes.c_void_p),
                ("Cout", ctypes.c_int32), ("Cin", ctypes.c_int32), ("K", ctypes.c_int32), ("fold", ctypes.c_int32),
                ("n_chunks", ctypes.c_int32), ("exclusive", ctypes.c_int32)]


class WgradQueue:
    """Weight gradients whose destination is the optimizer's flat gradient bucket are not finished one conv at a time
    (zero a slab, reduce it, un-fold it, spectral-norm chain rule, add: ~5 tiny launches per conv, ~900 per step).
    ConvFn.backward accumulates the raw gradient into a slot of a pre-zeroed arena and records a job; when the backward
    pass ends (autograd final callback) ALL jobs are finished by two launches (gim_wgrad_finish_batched) and the arena is
    cleared by one memset.  `.grad` is complete when backward() returns, exactly as before."""
    CHUNK = 4096
    PAGE = 32 << 20  # floats

    def __init__(self):
        self.pages = []       # [tensor, used]
        self.jobs = []        # tuples of ints (the job table signature)
        self.keep = []        # tensors that must outlive the flush
        self.streams = set()
        self.stream_of = {}   # raw stream handle -> torch Stream object
        self.cb_queued = False
        self.cache = collections.OrderedDict()   # job-table signature -> device tables (G / D backward, buffer parities)
        self.enabled = os.environ.get("GIM_WGRAD_IMMEDIATE") is None

    @staticmethod
    def _zeroed_mark():
        """[event after the zero-fill just issued on the current stream, streams ordered behind it, recorded inside a capture]."""
        return [torch.cuda.current_stream().record_event(), {_stream()}, torch.cuda.is_current_stream_capturing()]

    def take(self, n, device):
        """n zeroed floats (64-float aligned) that stay valid until the flush."""
        n = (n + 63) & ~63
        raw = _stream()
        if not self.pages or self.pages[-1][1] + n > self.pages[-1][0].numel():
            # the zero-fill runs on the allocating stream; every OTHER stream that later adds into this page first waits for it
            page = torch.zeros(max(n, self.PAGE), device=device, dtype=torch.float32)
            self.pages.append([page, 0] + self._zeroed_mark())
        pg = self.pages[-1]
        if raw not in pg[3]:
            cur = torch.cuda.current_stream()
            # (an event recorded BEFORE a hipGraph capture began is not waited for inside the capture: that work has completed -
            # GraphedGimStep synchronizes before it captures - and a captured wait on an un-captured event is not a graph edge)
            if pg[4] or not torch.cuda.is_current_stream_capturing():
                cur.wait_event(pg[2])
            pg[0].record_stream(cur)
            pg[3].add(raw)
        ptr = pg[0].data_ptr() + 4 * pg[1]
        pg[1] += n
        return ptr

    def add(self, job, keep):
        self.jobs.append(job)
        self.keep.append(keep)
        raw = _stream()
        if raw not in self.stream_of:
            self.stream_of[raw] = torch.cuda.current_stream()
        self.streams.add(self.stream_of[raw])
        if not self.cb_queued:
            torch.autograd.Variable._execution_engine.queue_callback(self.flush)
            self.cb_queued = True

    def flush(self):
        self.cb_queued = False
        if not self.jobs:
            return
        cur = torch.cuda.current_stream()
        for st in self.streams:   # slots were written on the encoders' side streams too
            stream_wait(cur, st)
        device = self.pages[0][0].device
        sig = tuple(self.jobs)
        dev_tabs = self.cache.get(sig)
        if dev_tabs is None:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("WgradQueue: this backward pass has a job table not seen before; run eager warm-up steps "
                                   "(at least 2) before capturing a hipGraph")
            arr = (_WgradJob * len(sig))()
            tab, tab_sn = [], []
            targets = collections.Counter(jb[8] for jb in sig)   # grad_w pointers: a conv called several times in the pass has several jobs
            for j, jb in enumerate(sig):
                a = arr[j]
                (a.src, a.bias_src, a.w, a.sigma, a.u, a.v, a.tmp, a.partial, a.grad_w, a.grad_b,
                 a.Cout, a.Cin, a.K, a.fold, a.n_chunks) = jb
                a.exclusive = 1 if targets[jb[8]] == 1 else 0
                blocks = [(j, c) for c in range(jb[14])]
                tab += blocks
                if jb[3]:
                    tab_sn += blocks
            host_jobs = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).pin_memory()
            host_tab = torch.tensor(tab, dtype=torch.int32).pin_memory()
            host_sn = torch.tensor(tab_sn if tab_sn else [(0, 0)], dtype=torch.int32).pin_memory()
            dev_tabs = (host_jobs.to(device, non_blocking=True), host_tab.to(device, non_blocking=True),
                        host_sn.to(device, non_blocking=True), len(tab), len(tab_sn), (host_jobs, host_tab, host_sn))
            self.cache[sig] = dev_tabs
            while len(self.cache) > 16:
                self.cache.popitem(last=False)
        else:
            self.cache.move_to_end(sig)
        dj, dt, dsn, nb, nbs, _ = dev_tabs
        check(_lib.load().gim_wgrad_finish_batched(dj.data_ptr(), len(sig), dt.data_ptr(), nb, dsn.data_ptr(), nbs, _stream()),
              "wgrad_finish_batched")
        if len(self.pages) > 1:  # first backward of a new shape: merge into one page for the next pass
            total = sum(pg[1] for pg in self.pages)
            self.pages = [[torch.zeros(total + (total >> 3), device=device, dtype=torch.float32), 0] + self._zeroed_mark()]
            self.cache.clear()
        else:
            pg = self.pages[0]
            pg[0][:pg[1]].zero_()
            pg[1] = 0
            # the re-zeroing runs on the flushing stream; the next pass's first user on every other stream waits for it
            pg[2:] = self._zeroed_mark()
        self.jobs, self.keep, self.streams = [], [], set()


# Lanes: the generator step's backward and the discriminator step overlap on the GPU (gim_img_training.gim_step);
# each lane has its own arena / job queue (and its own side streams, gim_img_models._side_streams).
_LANE = [0]
_QUEUES = {0: WgradQueue()}
wgrad_queue = _QUEUES[0]


def reset_wgrad_queues():
    """Drop weight-gradient jobs that a failed backward pass (exception inside autograd) left behind, re-zero their arena
    slots and re-arm the end-of-backward callback.  Called by FusedAdam.zero_grad(): a new iteration never inherits them."""
    for q in _QUEUES.values():
        if q.jobs or q.cb_queued:
            for pg in q.pages:
                if pg[1]:
                    pg[0][:pg[1]].zero_()
                    pg[1] = 0
                    pg[2:] = q._zeroed_mark()
            q.jobs, q.keep, q.streams, q.cb_queued = [], [], set(), False


def current_lane():
    return _LANE[0]


def _queue():
    q = _QUEUES.get(_LANE[0])
    if q is None:
        q = _QUEUES[_LANE[0]] = WgradQueue()
    return q


def stream_wait(waiter, other):
    """waiter.wait_stream(other), skipped when both are the same HIP stream.  The role -> stream map of gim_img_models aliases
    roles onto shared streams (lane 1's first encoder runs on lane 1's own stream): a stream "waiting for itself" is a no-op in
    eager execution, but inside a hipGraph capture it records an event on the capturing stream and then makes that same stream
    wait for it - a self-edge in the captured graph (see graph.GraphedGimStep)."""
    if waiter.cuda_stream != other.cuda_stream:
        waiter.wait_stream(other)


_PENDING_JOIN = []   # streams of lane-1 work the caller's stream has not waited for yet (gim_step(defer_join=True))


def defer_join(stream):
    _PENDING_JOIN.append(stream)


def join_lanes():
    """Make the current stream wait for lane work whose join was deferred.  Called before anything reads what the
    discriminator step wrote: its weights (the next generator step's D forward), its outputs, a checkpoint."""
    if _PENDING_JOIN:
        cur = torch.cuda.current_stream()
        for st in _PENDING_JOIN:
            stream_wait(cur, st)
        del _PENDING_JOIN[:]


class lane:
    """Context manager: work issued inside belongs to lane `idx` (host-side selection, read by the autograd worker)."""

    def __init__(self, idx):
        self.idx = idx

    def __enter__(self):
        self.prev = _LANE[0]
        _LANE[0] = self.idx

    def __exit__(self, *exc):
        _LANE[0] = self.prev


# --------------------------------------------------------------------------------------------
# convolution / linear
# --------------------------------------------------------------------------------------------
class ConvFn(Function):
    """y = [avgpool2]( conv(up2^ups(lrelu(x, pre_slope)), w) ) / sigma + bias + res   (NHWC; linear when x is 2-D).

    pool: the 2x2 average pool behind the conv is folded into ONE stride-2 convolution; ups with a KxK kernel
    (K > 1) runs in its sub-pixel form on the low-resolution input: neither pooling nor upsampling costs conv FLOPs.
    res_ups: the residual is stored at half the output resolution."""

    @staticmethod
    def forward(ctx, x, w, bias, res, sigma, u_s, v_s, ups, pre_slope, pool, res_ups, wf=None, guard=None, post_slope=1.0, x_act=False):
        lib = _lib.load()
        x = _req(x, "x")
        wp = weight_phys(_req_w(w))
        if x.dim() == 2:
            N, Hs, Ws, Cin = x.shape[0], 1, 1, x.shape[1]
        else:
            N, Hs, Ws, Cin = x.shape
        Cout = w.shape[0]
        KH = w.shape[2] if w.dim() == 4 else 1
        if w.shape[1] != Cin:
            raise RuntimeError("conv: weight expects %d input channels, got %d" % (w.shape[1], Cin))
        H, W = Hs << ups, Ws << ups
        fold = 1 if (pool or (ups and KH > 1)) else 0
        # x_act: x is stored ACTIVATED (its producer applied this conv's LeakyReLU in its epilogue, post_slope below): forward and
        # wgrad then run without the per-tap activation; dgrad keeps the slope - its mask only needs the sign, which is the same
        sh = _shape(N, H, W, Cin, Cout, KH, ups, 1.0 if x_act else pre_slope, 1 if pool else 0, fold, 1 if res_ups else 0)
        Ho, Wo = (H >> 1, W >> 1) if pool else (H, W)
        key = (N, H, W, Cin, Cout, KH, ups, 1 if pool else 0, fold)
        _tuned(sh, "fwd", key)
        # post_slope != 1: store lrelu(y) for a consumer that is the ONLY reader of y and runs with x_act.  conv2d_post_act has
        # resolved it (1.0 when the launch splits K: the slices combine by addition); a split-K launch refuses it here
        merged = bool(ups) and _merged_subpixel(x, w, ups, res, sh)
        if post_slope != 1.0 and not merged:
            if x.dim() != 4 or _splits_k(sh, 0, key):
                raise RuntimeError("conv: an activated output (post_slope) cannot be combined with a split-K launch or a linear layer")
            sh.post_slope = post_slope
        if res is not None:
            res = _req(res, "res")
        xp = None
        if _ROWS_FORM and x.dim() == 4 and KH >= 3 and Cin <= 8 and KH * Cin <= 64 and Cout >= 16 and Cout % 4 == 0 and not (ups or pool or res_ups) \
                and sh.tune_tile == 0:
            # image layers: row-contiguous K on a zero-padded, activated copy of the image (include/gim_hip.h gim_conv2d_fwd_rows)
            pad = (KH - 1) // 2
            xp = torch.empty((N, H + 2 * pad, W + 2 * pad, Cin), device=x.device, dtype=torch.float32)
            check(lib.gim_pad_image(_p(x), _p(xp), N, H, W, Cin, pad, 1.0 if x_act else pre_slope, _stream()), "pad_image")
            wrows = _transposed(lib, w, wp, Cout, Cin, KH, rows=True)
            sh.tune_ksplit = 1      # one K slice: these layers have >= 10^5 output pixels; keeps an activated output (post_slope) legal
            y = torch.empty((N, Ho, Wo, Cout), device=x.device, dtype=torch.float32)
            check(lib.gim_conv2d_fwd_rows(_p(xp), _p(wrows), _p(bias), _p(sigma), _p(res), _p(y), sh, _stream()), "conv2d_fwd_rows")
        elif merged:
            # the generator's last layer (9x9 64->3 behind the upsample): the four output-parity classes as ONE plain 5-tap convolution
            # to 4 * Cout channels + a depth-to-space copy (include/gim_hip.h gim_conv2d_pack_subpixel_weights); gradients: the sub-pixel forms
            if wf is None:
                wf = _folded(wp, Cout, Cin, KH)
            wm = _transposed(lib, w, wf, Cout, Cin, KH, subpix=True)
            shm = _shape(N, Hs, Ws, Cin, 4 * Cout, (KH + 1) // 2, 0, 1.0 if x_act else pre_slope, 0, 0, 0)
            keym = (N, Hs, Ws, Cin, 4 * Cout, (KH + 1) // 2, 0, 0, 0)
            _tuned(shm, "fwd", keym)
            y4 = _conv_out(shm, 0, keym, (N, Hs, Ws, 4 * Cout), x.device)
            check(lib.gim_conv2d_fwd(_p(x), _p(wm), None, _p(sigma), None, _p(y4), shm, _stream()), "conv2d_fwd")
            y = torch.empty((N, H, W, Cout), device=x.device, dtype=torch.float32)
            check(lib.gim_depth_to_space2(_p(y4), _p(bias), _p(y), N, Hs, Ws, Cout, post_slope, _stream()), "depth_to_space2")
        else:
            y = _conv_out(sh, 0, key, (N, Cout) if x.dim() == 2 else (N, Ho, Wo, Cout), x.device)
            if fold and wf is None:
                wf = _folded(wp, Cout, Cin, KH)
            wk = wf if fold else wp
            check(lib.gim_conv2d_fwd(_p(x), _p(wk), _p(bias), _p(sigma), _p(res), _p(y), sh, _stream()), "conv2d_fwd")
        ctx.save_for_backward(x, w, sigma, u_s, v_s, wf if fold else None, bias, xp)
        ctx.cfg = (N, H, W, Cin, Cout, KH, ups, pre_slope, bias is not None, res is not None, bool(pool), fold, bool(res_ups))
        ctx.guard = guard
        ctx.x_act = bool(x_act)
        if _FLOPS is not None:
            _note_conv("fwd", ctx.cfg, sh, 0)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, w, sigma, u_s, v_s, wf, bias, xp = ctx.saved_tensors
        N, H, W, Cin, Cout, KH, ups, pre_slope, has_bias, has_res, pool, fold, res_ups = ctx.cfg
        dy = _req(dy, "dy")
        if ctx.guard is not None and ctx.guard[0].stale(ctx.guard[1]):
            raise RuntimeError("the spectral-norm state (sigma, u, v) of this forward pass was overwritten by later forward "
                               "passes of the same model: run backward before the third forward")
        if _second_order():
            if ups or res_ups:
                raise NotImplementedError("second-order backward of an upsampling convolution is not on the R1 path")
            dx = ConvDgradFn.apply(dy, w, x, sigma, u_s, v_s, wf, ctx.cfg) if ctx.needs_input_grad[0] else None
            return dx, None, None, (dy if has_res and ctx.needs_input_grad[3] else None), None, None, None, None, None, None, None, None, None, None, None
        wp = weight_phys(w)
        sh = _shape(N, H, W, Cin, Cout, KH, ups, pre_slope, 1 if pool else 0, fold, 1 if res_ups else 0)
        st = _stream()
        dev = dy.device
        dx = dw = db = dres = None
        if ctx.needs_input_grad[0]:
            sh_d = sh
            if _TUNE_OVERRIDE:
                sh_d = _tuned(_shape(N, H, W, Cin, Cout, KH, ups, pre_slope, 1 if pool else 0, fold, 1 if res_ups else 0), "dgrad",
                              (N, H, W, Cin, Cout, KH, ups, 1 if pool else 0, fold))
            dx = _conv_dgrad(lib, dy, x, wp, wf, sigma, sh_d, ctx.cfg, st, w, getattr(ctx, "dgrad_res", None))
        want_w = ctx.needs_input_grad[1]
        want_b = has_bias and ctx.needs_input_grad[2]
        Mo = N * (H >> 1) * (W >> 1) if pool else N * H * W  # pixels of dy
        if want_w:
            sh_w = sh
            if ctx.x_act or _TUNE_OVERRIDE:   # x_act: the stored x is already lrelu(x): no activation on the wgrad operand
                sh_w = _tuned(_shape(N, H, W, Cin, Cout, KH, ups, 1.0 if ctx.x_act else pre_slope, 1 if pool else 0, fold, 1 if res_ups else 0),
                              "wgrad", (N, H, W, Cin, Cout, KH, ups, 1 if pool else 0, fold))
            dw, db = _conv_wgrad(lib, dy, x, w, wp, bias, sigma, u_s, v_s, sh_w, ctx.cfg, want_b, st, xp)
        elif want_b:
            db = torch.empty(Cout, device=dev, dtype=torch.float32)
            scratch = torch.empty(256 * Cout, device=dev, dtype=torch.float32)
            check(lib.gim_colsum(_p(dy), _p(db), _p(scratch), Mo, Cout, st), "colsum")
        if has_res and ctx.needs_input_grad[3]:
            if res_ups:
                dres = torch.empty((N, H >> 1, W >> 1, Cout), device=dev, dtype=torch.float32)
                check(lib.gim_upsample2x_bwd(_p(dy), None, 1.0, _p(dres), N, H >> 1, W >> 1, Cout, st), "upsample2x_bwd")
            else:
                dres = dy
        return dx, dw, db, dres, None, None, None, None, None, None, None, None, None, None, None


# Backward's Python on the CALLING thread.  torch's autograd engine hands a backward pass to a per-device worker thread; every node
# of this engine is a Python Function, and with the hand-off the host needs 31-37 ms to enqueue one training step where it needs 26-28 ms
# when the calling thread runs the nodes itself (64x64x3, 16 episodes: the device needs 37 ms, so with the worker thread the host IS the
# bound in part of the runs - 418-426 episodes/s instead of 429-431; at 1 / 4 episodes per step: 31 -> 42 / 132 -> 159 episodes/s;
# profiles/r04_m_host_enqueue.txt).  Same graph, same kernels, same streams (the engine sets each node's forward stream either way).
# GIM_MT_AUTOGRAD=1 restores torch's default.
_CALLER_THREAD_BACKWARD = os.environ.get("GIM_MT_AUTOGRAD") is None


def caller_thread_backward():
    """Context for .backward() / autograd.grad() calls of the training step: autograd's nodes run on the calling thread."""
    return torch.autograd.set_multithreading_enabled(not _CALLER_THREAD_BACKWARD)


_ACT_STORAGE = os.environ.get("GIM_NO_ACT_STORAGE") is None   # A/B switch (host side)
_ROWS_FORM = os.environ.get("GIM_NO_ROWS_FORM") is None   # A/B switch (host side): row-contiguous K for the image layers
_MERGED_SUBPIXEL = os.environ.get("GIM_NO_MERGED_SUBPIXEL") is None   # A/B switch (host side): stacked parity classes for the 9x9 64->3 layer


def _merged_subpixel(x, w, ups, res, sh):
    """Does this forward run as the stacked-parity-class convolution (ConvFn.forward; include/gim_hip.h gim_conv2d_pack_subpixel_weights)?"""
    KH = w.shape[2] if w.dim() == 4 else 1
    return bool(_MERGED_SUBPIXEL and ups and KH >= 5 and (KH & 3) == 1 and w.shape[0] <= 4 and x.dim() == 4 and res is None and sh.tune_tile == 0)

_NARROW_DGRAD_T = os.environ.get("GIM_NO_NARROW_DGRAD_T") is None   # A/B switch (host side)
_NARROW_XFOLD = os.environ.get("GIM_NO_NARROW_XFOLD") is None   # A/B switch (host side)
_WT_CACHE = {}   # (weight data_ptr, taps per dim) -> (version key, WT, ready event, stream, weakref to the parameter)


def _transposed(lib, w, wk, Cout, Cin, KF, xfold=0, rows=False, subpix=False):
    """WT[Cin][KF][KF][Cout] of the (plain or folded) weights `wk` of parameter `w` - or, xfold = J, the x-folded
    WX[J * Cin][KF][KF + J - 1][Cout] of gim_conv2d_xfold_weights; or, rows, the row-padded WP[Cout][KF][KF * Cin -> 16] of
    gim_conv2d_pack_rows_weights; or, subpix (wk = the folded taps, KF = the conv's K), the stacked parity classes
    WM[4 Cout][(KF+1)/2][(KF+1)/2][Cin] of gim_conv2d_pack_subpixel_weights - recomputed only when the weights changed (autograd version
    counter for torch-side writes, optim.weights_epoch for the fused Adam kernel)."""
    from . import optim
    key = (w._version, optim.weights_epoch(w))
    slot = (w.data_ptr(), KF, "rows" if rows else ("subpix" if subpix else xfold))
    ent = _WT_CACHE.get(slot)
    raw = _stream()
    if ent is None or ent[0] != key or ent[4]() is not w:
        cur = torch.cuda.current_stream()
        if rows:
            wt = torch.empty(Cout * KF * ((KF * Cin + 15) & ~15), device=wk.device, dtype=torch.float32)
            check(lib.gim_conv2d_pack_rows_weights(_p(wk), _p(wt), Cout, Cin, KF, raw), "pack_rows_weights")
        elif subpix:
            wt = torch.empty(4 * Cout * ((KF + 1) // 2) ** 2 * Cin, device=wk.device, dtype=torch.float32)
            check(lib.gim_conv2d_pack_subpixel_weights(_p(wk), _p(wt), Cout, Cin, KF, raw), "pack_subpixel_weights")
        elif xfold:
            wt = torch.empty(xfold * Cin * KF * (KF + xfold - 1) * Cout, device=wk.device, dtype=torch.float32)
            check(lib.gim_conv2d_xfold_weights(_p(wk), _p(wt), Cout, Cin, KF, xfold, raw), "xfold_weights")
        else:
            wt = torch.empty(Cin * KF * KF * Cout, device=wk.device, dtype=torch.float32)
            check(lib.gim_conv2d_transpose_weights(_p(wk), _p(wt), Cout, Cin, KF, raw), "transpose_weights")
        ent = (key, wt, cur.record_event(), raw, weakref.ref(w, lambda _r, slot=slot: _WT_CACHE.pop(slot, None)),
               torch.cuda.is_current_stream_capturing())
        _WT_CACHE[slot] = ent
    elif ent[3] != raw:
        cur = torch.cuda.current_stream()
        if ent[5] or not torch.cuda.is_current_stream_capturing():   # see WgradQueue.take
            cur.wait_event(ent[2])
        ent[1].record_stream(cur)
    return ent[1]


def _xfold_factor(Cin, W):
    """J of the x-folded dgrad, dividing W.  Cin <= 4: J = 4 -> up to 16 output columns, the 16-column MFMA tile (12 of 16 carry data
    for RGB).  Cin = 5, 6 (the generator's 6-channel image pair): J = 4 -> 24 columns on the 32-column tile of
    v_mfma_f32_32x32x2_f32 instead of 12 on the 16-column tile of the 16 x 16 x 4 form (the same peak rate, but twice the LDS operand
    reads per FLOP and a quarter of the work per instruction): 20 % more taps (K + 3 instead of K + 1 columns) and still
    0.51 -> 0.32 ms on the 9x9 64->6 gradient (round 4).  Cin = 7, 8: J = 2."""
    J = 4 if Cin <= 6 else 2
    return J if (Cin <= 8 and W % J == 0 and W // J >= 1) else 0


def _conv_dgrad(lib, dy, x, wp, wf, sigma, sh, cfg, st, w=None, res_half=None):
    """dx = lrelu'(x) * dgrad(dy, w) / sigma  (through the pool / sub-pixel folds when the forward used them).
    res_half [N, H/2, W/2, Cin]: the gradient w.r.t. avgpool2(x) of a second reader of x (ConvForkPoolFn): 0.25 * up2(res_half) is added -
    in the dgrad kernel's epilogue where the launch form allows it (gim_conv2d_dgrad_res), by gim_add_avgpool2_bwd otherwise.
    With the parameter `w` given, the gradient w.r.t. IMAGES (Cin <= 8, Cout % 16 == 0) runs the k-contiguous kernel on cached
    transposed weights (gim_conv2d_dgrad_t)."""
    N, H, W, Cin, Cout, KH, ups, pre_slope, has_bias, has_res, pool, fold, res_ups = cfg
    if res_half is not None:
        plain = not (ups or pool or fold) and pre_slope != 1.0 and not (w is not None and Cout % 16 == 0 and Cin <= 8 and _NARROW_DGRAD_T) \
            and not (sh.prec == 1 and Cout % 32 == 0 and Cin >= 32)
        if plain:
            if _FLOPS is not None:
                _note_conv("dgrad", cfg, sh, 1)
            key = (N, H, W, Cin, Cout, KH, ups, 0, 0)
            dx = _conv_out(sh, 1, key, tuple(x.shape), x.device)
            check(lib.gim_conv2d_dgrad_res(_p(dy), _p(wp), _p(sigma), _p(x), _p(_req(res_half, "res_half")), 0.25, _p(dx), sh, st), "conv2d_dgrad_res")
            return dx
        g = _conv_dgrad(lib, dy, x, wp, wf, sigma, sh, cfg, st, w)
        out = torch.empty_like(g)
        check(lib.gim_add_avgpool2_bwd(_p(g), _p(_req(res_half, "res_half")), _p(out), N, H, W, Cin, st), "add_avgpool2_bwd")
        return out
    if _FLOPS is not None:
        _note_conv("dgrad", cfg, sh, 1)
    mask = x if pre_slope != 1.0 else None
    key = (N, H, W, Cin, Cout, KH, ups, 1 if pool else 0, fold)
    wk = wf if fold else wp
    # dgrad on cached transposed weights WT[Cin][KF][KF][Cout] (rows k-contiguous: the forward kernel's operand path, vector weight
    # loads) for the gradient w.r.t. IMAGES (<= 8 input
    # channels: 3 / 6 / 1), where the k-major kernel falls back to scalar weight loads (output channels not a multiple of 4)
    # (fp16 matrix path: EVERY eligible dgrad goes this way - the fp16 kernel exists in the k-contiguous operand form only)
    f16_t = sh.prec == 1 and Cout % 32 == 0 and Cin >= 32
    if w is not None and Cout % 16 == 0 and not (ups and not fold) and ((Cin <= 8 and _NARROW_DGRAD_T) or f16_t):
        J = _xfold_factor(Cin, W) if KH >= 3 and not (ups or pool or fold) and _NARROW_XFOLD and not f16_t else 0
        if J:   # J adjacent dx pixels as the output columns of one stride-(1, J) convolution: 12 of 16 MFMA columns carry data
            wx = _transposed(lib, w, wk, Cout, Cin, KH, xfold=J)
            dx = torch.empty(tuple(x.shape), device=x.device, dtype=torch.float32)
            check(lib.gim_conv2d_dgrad_xfold(_p(dy), _p(wx), _p(sigma), _p(mask), _p(dx), sh, J, st), "conv2d_dgrad_xfold")
            return dx
        wt = _transposed(lib, w, wk, Cout, Cin, KH + 1 if fold else KH)
        dx = _conv_out(sh, 2, key, tuple(x.shape), x.device)
        check(lib.gim_conv2d_dgrad_t(_p(dy), _p(wt), _p(sigma), _p(mask), _p(dx), sh, st), "conv2d_dgrad_t")
        return dx
    if ups and not fold:
        dxu = _conv_out(sh, 1, key, (N, H, W, Cin), dy.device)
        dx = torch.empty_like(x)
        check(lib.gim_conv2d_dgrad(_p(dy), _p(wk), _p(sigma), None, _p(dxu), sh, st), "conv2d_dgrad")
        check(lib.gim_upsample2x_bwd(_p(dxu), _p(mask), pre_slope, _p(dx), N, H >> 1, W >> 1, Cin, st), "upsample2x_bwd")
    else:
        dx = _conv_out(sh, 1, key, tuple(x.shape), x.device)
        check(lib.gim_conv2d_dgrad(_p(dy), _p(wk), _p(sigma), _p(mask), _p(dx), sh, st), "conv2d_dgrad")
    return dx


def _conv_wgrad(lib, dy, x, w, wp, bias, sigma, u_s, v_s, sh, cfg, want_b, st, xp=None):
    """(dw, db) of one convolution; either may come back None because it was ADDED into the parameter's .grad.
    xp: the padded, activated copy of x that a forward in the row-contiguous form made (image layers): the queued path then takes
    the weight gradient from it (gim_conv2d_wgrad_rows_acc)."""
    N, H, W, Cin, Cout, KH, ups, pre_slope, has_bias, has_res, pool, fold, res_ups = cfg
    if _FLOPS is not None:
        _note_conv("wgrad", cfg)
    dev = dy.device
    dw = db = None
    Mo = N * (H >> 1) * (W >> 1) if pool else N * H * W  # pixels of dy
    ns = 1   # pixel slices combined with float atomics; GIM_WGRAD_SLABS=1: deterministic slabs (non-queued path only)
    if _WGRAD_SLABS or _DETERMINISTIC[0]:
        ns = lib.gim_conv2d_wgrad_slabs(sh)
        if ns <= 0:
            check(ns, "conv2d_wgrad_slabs")
    K = KH * KH * Cin
    KFF = (KH + 1) * (KH + 1) * Cin if fold else K
    slab_bias = want_b and not (fold and ups)  # the role-swapped sub-pixel wgrad does not stream dy as its A operand
    # Megatron-style direct accumulation: when the parameter's .grad already exists as a dense buffer in the
    # weight's own memory order (FusedAdam's flat gradient bucket) and no higher-order graph is being built,
    # the finish kernels ADD into it and autograd gets None (no AccumulateGrad add kernel per parameter).
    acc_w = _grad_target(w) if (sigma is not None or not fold) and not torch.is_grad_enabled() else None
    acc_b = _grad_target(bias) if (slab_bias and acc_w is not None) else None
    if acc_w is not None and wgrad_queue.enabled and not _DETERMINISTIC[0] and not (want_b and slab_bias and acc_b is None):
        q = _queue()
        # deferred: raw gradient into an arena slot now, finish of all convs in two launches when backward ends
        n = Cout * K
        if Cout * KFF >= 1 << 31:
            raise RuntimeError("weight gradient of more than 2^31 elements (gim_wgrad_finish_batched indexes with 32 bits)")
        n_chunks = (n + q.CHUNK - 1) // q.CHUNK
        rows = xp is not None and dy.is_contiguous()
        src = q.take(Cout * KH * ((KH * Cin + 15) & ~15) if rows else Cout * KFF, dev)
        bsrc = q.take(Cout, dev) if slab_bias else None
        sn = sigma is not None
        tmp = q.take(n, dev) if ((fold or rows) and sn) else None
        part = q.take(n_chunks, dev) if sn else None
        if rows:   # slot [Cout][KH][KH * Cin -> 16]; the batched finish un-pads it (fold code 3)
            check(lib.gim_conv2d_wgrad_rows_acc(_p(dy), _p(xp), src, bsrc, sh, st), "conv2d_wgrad_rows_acc")
        else:
            check(lib.gim_conv2d_wgrad_acc(_p(dy), _p(x), src, bsrc, sh, st), "conv2d_wgrad_acc")
        q.add((src, bsrc or 0, _p(wp) if sn else 0, _p(sigma) or 0, _p(u_s) or 0, _p(v_s) or 0, tmp or 0, part or 0,
               _p(acc_w), _p(acc_b) or 0, Cout, Cin, KH, 3 if rows else ((2 if ups else 1) if fold else 0), n_chunks), (sigma, u_s, v_s))
        if want_b and not slab_bias:
            scr = torch.empty(256 * Cout, device=dev, dtype=torch.float32)
            tgt_b = _grad_target(bias)
            if tgt_b is not None:
                check(lib.gim_colsum_acc(_p(dy), _p(tgt_b), _p(scr), Mo, Cout, st), "colsum_acc")
            else:
                db = torch.empty(Cout, device=dev, dtype=torch.float32)
                check(lib.gim_colsum(_p(dy), _p(db), _p(scr), Mo, Cout, st), "colsum")
        return None, db
    dwp = torch.empty(Cout * K, device=dev, dtype=torch.float32)
    if want_b and acc_b is None:
        db = torch.empty(Cout, device=dev, dtype=torch.float32)
    if ns == 1 and sigma is None and not fold and acc_w is None:
        check(lib.gim_conv2d_wgrad(_p(dy), _p(x), _p(dwp), _p(db) if slab_bias else None, 1, sh, st), "conv2d_wgrad")
    else:
        slabs = torch.empty(ns * Cout * KFF, device=dev, dtype=torch.float32)
        bslabs = torch.empty(ns * Cout, device=dev, dtype=torch.float32) if slab_bias else None
        scratch = torch.empty(512 + (Cout * KFF if fold else 0), device=dev, dtype=torch.float32)
        check(lib.gim_conv2d_wgrad(_p(dy), _p(x), _p(slabs), _p(bslabs), ns, sh, st), "conv2d_wgrad")
        check(lib.gim_wgrad_finish(_p(slabs), _p(bslabs), ns, _p(wp), _p(sigma), _p(u_s), _p(v_s), _p(dwp),
                                   _p(db) if (slab_bias and acc_b is None) else None, _p(scratch), Cout, Cin, KH,
                                   (2 if ups else 1) if fold else 0, _p(acc_w), _p(acc_b), st), "wgrad_finish")
    if want_b and not slab_bias:
        scr = torch.empty(256 * Cout, device=dev, dtype=torch.float32)
        check(lib.gim_colsum(_p(dy), _p(db), _p(scr), Mo, Cout, st), "colsum")
    if acc_w is None:
        dw = dwp.view(Cout, KH, KH, Cin).permute(0, 3, 1, 2) if w.dim() == 4 else dwp.view(Cout, Cin)
    return dw, db


class ConvDgradFn(Function):
    """The first-order input gradient of ConvFn as an operator of (dy, w):  dx = lrelu'(x) * conv^T(dy, w / sigma(w)).
    It is bilinear in (dy, w/sigma), so its own adjoints are the other two kernels of the same convolution:
    d/d(dy) is the FORWARD conv of the masked cotangent and d/dw is the WGRAD with that cotangent in the role of
    the layer input (followed by the same spectral-norm chain rule).  The mask is piecewise constant in x."""

    @staticmethod
    def forward(ctx, dy, w, x, sigma, u_s, v_s, wf, cfg):
        lib = _lib.load()
        N, H, W, Cin, Cout, KH, ups, pre_slope, has_bias, has_res, pool, fold, res_ups = cfg
        sh = _shape(N, H, W, Cin, Cout, KH, 0, pre_slope, 1 if pool else 0, fold, 0)
        dx = _conv_dgrad(lib, dy, x, weight_phys(w), wf, sigma, sh, cfg, _stream(), w)
        ctx.save_for_backward(dy, w, x, sigma, u_s, v_s, wf)
        ctx.cfg = cfg
        return dx

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        dy, w, x, sigma, u_s, v_s, wf = ctx.saved_tensors
        N, H, W, Cin, Cout, KH, ups, pre_slope, has_bias, has_res, pool, fold, res_ups = ctx.cfg
        if torch.is_grad_enabled():
            raise NotImplementedError("third-order gradients are not supported")
        g = _req(g, "g")
        st = _stream()
        wp = weight_phys(w)
        if pre_slope != 1.0:
            gm = torch.empty_like(g)
            check(lib.gim_lrelu_mask_mul(_p(g), _p(x), pre_slope, _p(gm), g.numel(), st), "lrelu_mask_mul")
        else:
            gm = g
        lin = (N, H, W, Cin, Cout, KH, 0, 1.0, False, False, pool, fold, False)   # the conv as a linear map: no prologue, bias, residual
        sh = _shape(N, H, W, Cin, Cout, KH, 0, 1.0, 1 if pool else 0, fold, 0)
        g_dy = g_w = None
        if ctx.needs_input_grad[0]:
            g_dy = torch.empty_like(dy)
            check(lib.gim_conv2d_fwd(_p(gm), _p(wf if fold else wp), None, _p(sigma), None, _p(g_dy), sh, st), "conv2d_fwd")
            if _FLOPS is not None:
                _note_conv("fwd", lin, sh, 0)
        if ctx.needs_input_grad[1]:
            g_w, _ = _conv_wgrad(lib, dy, gm, w, wp, None, sigma, u_s, v_s, sh, lin, False, st)
        return g_dy, g_w, None, None, None, None, None, None


def _grad_target(p):
    """The parameter's existing .grad as a flat buffer in the parameter's own memory order, or None.  (The checked view is kept
    on the gradient tensor, keyed by its data pointer - FusedAdam's gradients are fixed views of its flat bucket.)"""
    if p is None:
        return None
    g = p.grad
    if g is None:
        return None
    try:
        c = g._gim_tgt
        if c[0] == g.data_ptr():
            return c[1]
    except AttributeError:
        pass
    if not g.is_cuda or g.dtype != torch.float32 or g.shape != p.shape:
        return None
    gp = g.permute(0, 2, 3, 1) if g.dim() == 4 else g
    if not gp.is_contiguous():
        return None
    g._gim_tgt = (g.data_ptr(), gp)
    return gp


def _req_w(w):
    if not (w.is_cuda and w.dtype == torch.float32):
        raise RuntimeError("weights must be CUDA float32 (got %s on %s): the GIM engine has no CPU path" % (w.dtype, w.device))
    return w


def conv2d(x, w, bias=None, res=None, sigma=None, u_s=None, v_s=None, ups=0, pre_slope=1.0, pool=False, res_ups=False, wf=None,
           guard=None, x_act=False):
    return ConvFn.apply(x, w, bias, res, sigma, u_s, v_s, ups, pre_slope, pool, res_ups, wf, guard, 1.0, x_act)


def conv2d_post_act(x, w, bias=None, res=None, sigma=None, u_s=None, v_s=None, ups=0, pre_slope=1.0, pool=False, res_ups=False, wf=None,
                    guard=None, post_slope=1.0, x_act=False):
    """conv2d whose output may be stored ACTIVATED, y_stored = lrelu(y, post_slope), for a consumer conv that is the only reader
    of y and is then called with x_act=True (it skips its per-tap LeakyReLU in forward and wgrad; its dgrad masks by the sign,
    which activation does not change, and hands back the gradient w.r.t. the RAW y - so this conv's backward is unchanged).
    Returns (y_stored, activated): launches that split K cannot activate (their slices combine by addition) and return raw y."""
    act = False
    if _ACT_STORAGE and post_slope != 1.0 and x.dim() == 4:
        # decided HERE, for this call's own shape, and handed to ConvFn as the resolved slope: nothing about the stored form of y
        # travels through module state (another conv running in between could not change what this call reports)
        N, Hs, Ws, Cin = x.shape
        Cout, KH = w.shape[0], (w.shape[2] if w.dim() == 4 else 1)
        H, W = Hs << ups, Ws << ups
        fold = 1 if (pool or (ups and KH > 1)) else 0
        key = (N, H, W, Cin, Cout, KH, ups, 1 if pool else 0, fold)
        sh = _tuned(_shape(N, H, W, Cin, Cout, KH, ups, pre_slope, 1 if pool else 0, fold, 1 if res_ups else 0), "fwd", key)
        act = _merged_subpixel(x, w, ups, res, sh) or not _splits_k(sh, 0, key)   # (the stacked form activates in its depth-to-space copy)
    y = ConvFn.apply(x, w, bias, res, sigma, u_s, v_s, ups, pre_slope, pool, res_ups, wf, guard, post_slope if act else 1.0, x_act)
    return y, act


class ConvForkPoolFn(Function):
    """(y, pooled) = (ConvFn(x, w, ...), avgpool2(raw x)) for the TWO readers of a ResBlockDown's input (models/model_blocks.py:497-514:
    conv_r1 behind a LeakyReLU, the 1x1 skip conv on the pooled input).  Forward = the two launches they always were; backward = ONE
    dgrad launch whose epilogue adds 0.25 * up2(d pooled) to the masked conv gradient (gim_conv2d_dgrad_res) - round 3 summed the two
    branches in a kernel of its own (gim_add_avgpool2_bwd: a read-modify-write of the block's whole input gradient, 36 launches and
    0.58 ms per training step).  Inputs = ConvFn's, in ConvFn's order (so that ConvFn.backward's bookkeeping applies as it is), plus
    in_slope (x stored activated: the pool inverts the LeakyReLU on the fly, see AvgPool2Fn)."""

    @staticmethod
    def forward(ctx, x, w, bias, res, sigma, u_s, v_s, ups, pre_slope, pool, res_ups, wf, guard, post_slope, x_act, in_slope):
        lib = _lib.load()
        x = _req(x, "x")
        N, H, W, C = x.shape
        pooled = torch.empty((N, H // 2, W // 2, C), device=x.device, dtype=torch.float32)
        if in_slope != 1.0:
            check(lib.gim_avgpool2_fwd_act(_p(x), _p(pooled), N, H, W, C, in_slope, _stream()), "avgpool2_fwd_act")
        else:
            check(lib.gim_avgpool2_fwd(_p(x), _p(pooled), N, H, W, C, _stream()), "avgpool2_fwd")
        y = ConvFn.forward(ctx, x, w, bias, None, sigma, u_s, v_s, 0, pre_slope, False, False, None, guard, post_slope, x_act)
        return y, pooled

    @staticmethod
    def backward(ctx, dy, dpooled):
        N, H, W, Cin = ctx.cfg[0], ctx.cfg[1], ctx.cfg[2], ctx.cfg[3]
        if dy is None:      # the conv branch is unused: only the pool's backward
            return (AvgPool2BwdFn.apply(dpooled, (N, H, W, Cin)) if dpooled is not None else None,) + (None,) * 15
        if dpooled is not None and (torch.is_grad_enabled() or not ctx.needs_input_grad[0]):
            # second-order pass (R1) - or no input gradient wanted at all: the unfused sum of differentiable pieces
            grads = ConvFn.backward(ctx, dy)
            gp = AvgPool2BwdFn.apply(dpooled, (N, H, W, Cin)) if ctx.needs_input_grad[0] else None
            dx = gp if grads[0] is None else (grads[0] if gp is None else grads[0] + gp)
            return (dx,) + tuple(grads[1:]) + (None,)
        ctx.dgrad_res = dpooled
        try:
            grads = ConvFn.backward(ctx, dy)
        finally:
            ctx.dgrad_res = None
        return tuple(grads) + (None,)


_FUSED_FORKPOOL = os.environ.get("GIM_NO_FUSED_FORKPOOL") is None   # A/B switch (host side)


def conv2d_forkpool(x, w, bias, sigma, u_s, v_s, pre_slope, guard, post_slope, x_act, in_slope):
    """-> (y_stored, activated, pooled): conv2d_post_act of a plain convolution plus avgpool2 of its (raw) input, as ONE autograd node
    whose backward folds the pooled branch's gradient into the dgrad epilogue (ConvForkPoolFn)."""
    act = False
    N, H, W, Cin = x.shape
    Cout, KH = w.shape[0], w.shape[2]
    if _ACT_STORAGE and post_slope != 1.0:
        key = (N, H, W, Cin, Cout, KH, 0, 0, 0)
        sh = _tuned(_shape(N, H, W, Cin, Cout, KH, 0, pre_slope), "fwd", key)
        act = not _splits_k(sh, 0, key)
    ps = post_slope if act else 1.0
    if _FUSED_FORKPOOL and x.requires_grad and torch.is_grad_enabled():
        y, pooled = ConvForkPoolFn.apply(x, w, bias, None, sigma, u_s, v_s, 0, pre_slope, False, False, None, guard, ps, x_act, in_slope)
    else:
        xa, pooled = fork_pool(x, in_slope)
        y = ConvFn.apply(xa, w, bias, None, sigma, u_s, v_s, 0, pre_slope, False, False, None, guard, ps, x_act)
    return y, act, pooled


def linear(x, w, bias=None, pre_slope=1.0):
    """nn.Linear on the last dim (optionally with a fused LeakyReLU on the input)."""
    shp = x.shape
    y = ConvFn.apply(x.reshape(-1, shp[-1]), w, bias, None, None, None, None, 0, pre_slope, False, False, None, None)
    return y.view(*shp[:-1], w.shape[0])


# --------------------------------------------------------------------------------------------
# grouped linears (many nn.Linear on one shared input: the style projections of the AdaIN blocks)
# --------------------------------------------------------------------------------------------
class _GemmJob(ctypes.Structure):
    _fields_ = [("A", ctypes.c_void_p), ("B", ctypes.c_void_p), ("C", ctypes.c_void_p), ("bias", ctypes.c_void_p),
                ("M", ctypes.c_int32), ("N", ctypes.c_int32), ("K", ctypes.c_int32), ("ldc", ctypes.c_int32),
                ("sAi", ctypes.c_int64), ("sAk", ctypes.c_int64), ("sBk", ctypes.c_int64), ("sBj", ctypes.c_int64),
                ("flags", ctypes.c_int32), ("reserved", ctypes.c_int32)]


_GEMM_TABLES = collections.OrderedDict()   # job-list signature -> (device jobs, device tiles, n_tiles, pinned host copies)


def _gemm_tables(jobs, col_tiles_only=False):
    """Device job / tile tables of a grouped launch (include/gim_hip.h gim_gemm_job), cached by their content: the operands are
    parameters, their .grad buffers and allocator blocks whose addresses repeat from step to step."""
    sig = (tuple(jobs), col_tiles_only)
    ent = _GEMM_TABLES.get(sig)
    if ent is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("grouped linears: a job table not seen before; run eager warm-up steps before capturing a hipGraph")
        arr = (_GemmJob * len(jobs))()
        tiles = []
        for j, jb in enumerate(jobs):
            a = arr[j]
            (a.A, a.B, a.C, a.bias, a.M, a.N, a.K, a.ldc, a.sAi, a.sAk, a.sBk, a.sBj, a.flags) = jb
            for it in range(1 if col_tiles_only else (jb[4] + 63) // 64):
                for jt in range((jb[5] + 63) // 64):
                    tiles.append((j, it, jt))
        hj = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).pin_memory()
        ht = torch.tensor(tiles, dtype=torch.int32).pin_memory()
        dev = torch.device("cuda", torch._C._cuda_getDevice())
        ent = _GEMM_TABLES[sig] = (hj.to(dev, non_blocking=True), ht.to(dev, non_blocking=True), len(tiles), (hj, ht))
        while len(_GEMM_TABLES) > 64:
            _GEMM_TABLES.popitem(last=False)
    else:
        _GEMM_TABLES.move_to_end(sig)
    return ent


class GroupedLinearFn(Function):
    """(y_0, ..., y_{n-1}) with y_g = x @ w_g^T + b_g for n nn.Linear layers on ONE input x [M, K]: one launch forward, and one
    grouped product + one grouped column sum backward (dx = sum_g dy_g @ w_g with float atomics into a zeroed buffer, dw_g =
    dy_g^T @ x and db_g added straight into the optimizer's gradient bucket where it exists).  Replaces 3 x n launches of 5-7 us
    (the 36 style projections of the generator: models/model_blocks.py:786-789,829-832) and the n - 1 autograd additions of dx."""

    @staticmethod
    def forward(ctx, x, *wb):
        lib = _lib.load()
        x = _req(x, "x")
        M, K = x.shape
        ws, bs = wb[0::2], wb[1::2]
        for w in ws:
            if not (w.is_cuda and w.dtype == torch.float32 and w.dim() == 2 and w.shape[1] == K and w.is_contiguous()):
                raise RuntimeError("grouped linear: weights must be contiguous CUDA float32 [out, %d]" % K)
        sizes = [w.shape[0] for w in ws]
        buf = torch.empty(M * sum(sizes), device=x.device, dtype=torch.float32)
        ys, off = [], 0
        for n_ in sizes:
            ys.append(buf[off:off + M * n_].view(M, n_))
            off += M * n_
        jobs = [(_p(x), _p(w), _p(y), _p(b) or 0, M, n_, K, n_, K, 1, 1, K, 0) for w, b, y, n_ in zip(ws, bs, ys, sizes)]
        dj, dt, nt, _ = _gemm_tables(jobs)
        check(lib.gim_bgemm_grouped(dj.data_ptr(), dt.data_ptr(), nt, _stream()), "bgemm_grouped")
        if _FLOPS is not None:
            for n_ in sizes:
                ent = _FLOPS.setdefault(("bgemm", (1, M, n_, K)), [0, 2.0 * M * n_ * K])
                ent[0] += 1
        ctx.save_for_backward(x, *wb)
        return tuple(ys)

    @staticmethod
    @once_differentiable
    def backward(ctx, *dys):
        lib = _lib.load()
        x, *wb = ctx.saved_tensors
        ws, bs = wb[0::2], wb[1::2]
        M, K = x.shape
        st = _stream()
        need_x = ctx.needs_input_grad[0]
        dx = torch.zeros((M, K), device=x.device, dtype=torch.float32) if need_x else None
        jobs, cjobs, grads = [], [], []
        for g, (w, b, dy) in enumerate(zip(ws, bs, dys)):
            gw = gb = None
            if dy is not None:
                dy = _req(dy, "dy")
                n_ = w.shape[0]
                if need_x:      # dx += dy_g @ w_g : A = dy_g [M, n], B(k = n, j) = w_g[k][j]
                    jobs.append((_p(dy), _p(w), _p(dx), 0, M, K, n_, K, n_, 1, K, 1, 2))
                if ctx.needs_input_grad[1 + 2 * g]:   # dw_g = dy_g^T @ x : A(i = n, k = m) = dy_g[m][i], B(k = m, j) = x[k][j]
                    tgt = _grad_target(w)
                    if tgt is None:
                        gw = torch.empty_like(w)
                    jobs.append((_p(dy), _p(x), _p(tgt if tgt is not None else gw), 0, n_, K, M, K, 1, n_, K, 1, 1 if tgt is not None else 0))
                if b is not None and ctx.needs_input_grad[2 + 2 * g]:
                    tgt = _grad_target(b)
                    if tgt is None:
                        gb = torch.empty_like(b)
                    cjobs.append((_p(dy), 0, _p(tgt if tgt is not None else gb), 0, M, n_, 0, 0, n_, 1, 0, 0, 1 if tgt is not None else 0))
                if _FLOPS is not None:
                    ent = _FLOPS.setdefault(("bgemm", (1, M, n_, K)), [0, 2.0 * M * n_ * K])
                    ent[0] += 2
            grads += [gw, gb]
        if jobs:
            dj, dt, nt, _ = _gemm_tables(jobs)
            check(lib.gim_bgemm_grouped(dj.data_ptr(), dt.data_ptr(), nt, st), "bgemm_grouped")
        if cjobs:
            dj, dt, nt, _ = _gemm_tables(cjobs, col_tiles_only=True)
            check(lib.gim_colsum_grouped(dj.data_ptr(), dt.data_ptr(), nt, st), "colsum_grouped")
        return (dx, *grads)


def grouped_linear(x, layers):
    """[y_g] for the (weight, bias) pairs in `layers`, all applied to x [M, K]."""
    flat = []
    for w, b in layers:
        flat += [w, b]
    return GroupedLinearFn.apply(x, *flat)


# --------------------------------------------------------------------------------------------
# instance norm / AdaIN
# --------------------------------------------------------------------------------------------
class NormFn(Function):
    """mode 0: InstanceNorm2d(affine) with scale/shift [C]; mode 1: ada_in with scale/shift [N,C].
    Optional residual added to the output.  x NHWC [N,H,W,C]."""

    @staticmethod
    def forward(ctx, x, scale, shift, res, mode, eps, post_slope=1.0):
        """post_slope != 1: the output is stored ACTIVATED for a conv that is its only reader and runs with x_act (ConvFn): that
        conv's dgrad hands back the gradient w.r.t. the RAW output, so this backward is unchanged (it never reads y)."""
        lib = _lib.load()
        x = _req(x, "x")
        scale_in, shift_in = scale, shift
        scale = _req(scale, "scale")
        shift = _req(shift, "shift")
        N, H, W, C = x.shape
        y = torch.empty_like(x)
        stats = torch.empty((N, C, 3), device=x.device, dtype=torch.float32)
        if res is not None:
            res = _req(res, "res")
        if post_slope != 1.0:
            check(lib.gim_norm_fwd_act(_p(x), _p(scale), _p(shift), _p(res), _p(y), _p(stats), N, H * W, C, mode, eps, post_slope, _stream()), "norm_fwd_act")
        else:
            check(lib.gim_norm_fwd(_p(x), _p(scale), _p(shift), _p(res), _p(y), _p(stats), N, H * W, C, mode, eps, _stream()), "norm_fwd")
        ctx.save_for_backward(x, scale, stats)
        ctx.cfg = (N, H * W, C, mode, res is not None)
        ctx.scale_param, ctx.shift_param = (scale_in, shift_in) if mode == 0 else (None, None)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        lib = _lib.load()
        x, scale, stats = ctx.saved_tensors
        N, HW, C, mode, has_res = ctx.cfg
        dy = _req(dy, "dy")
        st = _stream()
        dx = torch.empty_like(x)
        dsc = torch.empty((N, C), device=x.device, dtype=torch.float32)
        dsh = torch.empty((N, C), device=x.device, dtype=torch.float32)
        check(lib.gim_norm_bwd(_p(dy), _p(x), _p(scale), _p(stats), _p(dx), _p(dsc), _p(dsh), N, HW, C, mode, st), "norm_bwd")
        if mode == 0:
            # affine gradients: one launch, added straight into the parameters' .grad when that is the optimizer's flat
            # bucket (no AccumulateGrad add kernels: 6 launches -> 1 per InstanceNorm layer)
            tgt_s, tgt_b = _grad_target(ctx.scale_param), _grad_target(ctx.shift_param)
            if not (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]):
                dscale = dshift = None
            elif (tgt_s is not None and tgt_b is not None and ctx.needs_input_grad[1] and ctx.needs_input_grad[2]
                  and not torch.is_grad_enabled()):
                check(lib.gim_colsum2(_p(dsc), _p(dsh), _p(tgt_s), _p(tgt_b), N, C, 1, st), "colsum2")
                dscale = dshift = None
            else:
                dscale = torch.empty(C, device=x.device, dtype=torch.float32)
                dshift = torch.empty(C, device=x.device, dtype=torch.float32)
                check(lib.gim_colsum2(_p(dsc), _p(dsh), _p(dscale), _p(dshift), N, C, 0, st), "colsum2")
        else:
            dscale, dshift = dsc.view_as(scale), dsh.view_as(scale)
        return dx, dscale, dshift, (dy if has_res else None), None, None, None


def instance_norm(x, weight, bias, eps=1e-5, post_slope=1.0):
    """post_slope != 1 (and activated storage on): returns lrelu(y) for a conv called with x_act=True - see NormFn.forward."""
    return NormFn.apply(x, weight, bias, None, 0, eps, post_slope if _ACT_STORAGE else 1.0)


def ada_in(x, mean_style, std_style, res=None, eps=1e-5, post_slope=1.0):
    return NormFn.apply(x, std_style, mean_style, res, 1, eps, post_slope if _ACT_STORAGE else 1.0)


def act_storage():
    """Is activated storage in use (ops.conv2d_post_act / norm post_slope + x_act consumers)?  Host-side A/B switch GIM_NO_ACT_STORAGE."""
    return _ACT_STORAGE


# --------------------------------------------------------------------------------------------
# pooling / pointwise
# --------------------------------------------------------------------------------------------
class AvgPool2Fn(Function):
    """2x2 average pool.  in_slope != 1: x is stored ACTIVATED (lrelu(x, in_slope) written by its producer for the conv that reads
    it next to this pool); the kernel inverts the activation on the fly.  The gradient handed back is w.r.t. the RAW x - what
    every consumer of an activated tensor returns (ConvFn with x_act) - and does not depend on x: the backward is unchanged."""

    @staticmethod
    def forward(ctx, x, in_slope=1.0):
        lib = _lib.load()
        x = _req(x, "x")
        N, H, W, C = x.shape
        y = torch.empty((N, H // 2, W // 2, C), device=x.device, dtype=torch.float32)
        if in_slope != 1.0:
            check(lib.gim_avgpool2_fwd_act(_p(x), _p(y), N, H, W, C, in_slope, _stream()), "avgpool2_fwd_act")
        else:
            check(lib.gim_avgpool2_fwd(_p(x), _p(y), N, H, W, C, _stream()), "avgpool2_fwd")
        ctx.cfg = (N, H, W, C)
        return y

    @staticmethod
    def backward(ctx, dy):
        return AvgPool2BwdFn.apply(dy, ctx.cfg), None


class AvgPool2BwdFn(Function):
    """dx of the 2x2 average pool (a linear map; its adjoint is the pool itself)."""

    @staticmethod
    def forward(ctx, dy, cfg):
        lib = _lib.load()
        N, H, W, C = cfg
        dy = _req(dy, "dy")
        dx = torch.empty((N, H, W, C), device=dy.device, dtype=torch.float32)
        check(lib.gim_avgpool2_bwd(_p(dy), _p(dx), N, H, W, C, _stream()), "avgpool2_bwd")
        return dx

    @staticmethod
    def backward(ctx, g):
        return AvgPool2Fn.apply(g), None


class ForkFn(Function):
    """n aliases of x for n consumers.  Autograd sums the gradients of a tensor with several consumers pairwise - one launch and one
    full read-modify-write per extra consumer; here each consumer gets an alias of its own and the backward adds all incoming
    gradients in ONE kernel (gim_add_n).  Under create_graph (the R1 term) the sum is built from differentiable additions."""

    @staticmethod
    def forward(ctx, x, n):
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *gs):
        gs = [g for g in gs if g is not None]
        if not gs:
            return None, None
        if len(gs) == 1:
            return gs[0], None
        if _second_order():
            out = gs[0]
            for g in gs[1:]:
                out = out + g
            return out, None
        lib = _lib.load()
        gs = [_req(g, "g") for g in gs]
        out = torch.empty_like(gs[0])
        cur, rest = gs[0], gs[1:]
        while rest:
            take, rest = rest[:3], rest[3:]
            ptrs = [_p(t) for t in take] + [None] * (3 - len(take))
            check(lib.gim_add_n(_p(cur), ptrs[0], ptrs[1], ptrs[2], _p(out), out.numel(), _stream()), "add_n")
            cur = out
        return out, None


def fork(x, n):
    return ForkFn.apply(x, n) if (n > 1 and x.requires_grad and torch.is_grad_enabled()) else (x,) * n


class ForkPoolFn(Function):
    """(alias of x, avgpool2(x)) for the two consumers of a ResBlockDown's input (conv path / pooled skip path); the backward
    adds the conv path's gradient and the un-pooled skip gradient in one kernel (instead of avgpool2_bwd + an autograd addition)."""

    @staticmethod
    def forward(ctx, x, in_slope):
        lib = _lib.load()
        x = _req(x, "x")
        N, H, W, C = x.shape
        y = torch.empty((N, H // 2, W // 2, C), device=x.device, dtype=torch.float32)
        if in_slope != 1.0:
            check(lib.gim_avgpool2_fwd_act(_p(x), _p(y), N, H, W, C, in_slope, _stream()), "avgpool2_fwd_act")
        else:
            check(lib.gim_avgpool2_fwd(_p(x), _p(y), N, H, W, C, _stream()), "avgpool2_fwd")
        ctx.cfg = (N, H, W, C)
        return x.view_as(x), y

    @staticmethod
    def backward(ctx, g, dyp):
        if dyp is None:
            return g, None
        if g is None or _second_order():
            gp = AvgPool2BwdFn.apply(dyp, ctx.cfg)
            return (gp if g is None else g + gp), None
        lib = _lib.load()
        N, H, W, C = ctx.cfg
        g, dyp = _req(g, "g"), _req(dyp, "dy")
        out = torch.empty_like(g)
        check(lib.gim_add_avgpool2_bwd(_p(g), _p(dyp), _p(out), N, H, W, C, _stream()), "add_avgpool2_bwd")
        return out, None


def fork_pool(x, in_slope=1.0):
    """(x for the conv path, avgpool2(x) for the skip path) - see ForkPoolFn; plain pooling when x carries no gradient."""
    if x.requires_grad and torch.is_grad_enabled():
        return ForkPoolFn.apply(x, in_slope)
    return x, AvgPool2Fn.apply(x, in_slope)


class MaxPoolLreluFn(Function):
    """AdaptiveMaxPool2d((1,1)) -> flatten -> LeakyReLU(0.2) on NHWC input; output [N, C]."""

    @staticmethod
    def forward(ctx, x):
        lib = _lib.load()
        x = _req(x, "x")
        N, H, W, C = x.shape
        y = torch.empty((N, C), device=x.device, dtype=torch.float32)
        idx = torch.empty((N, C), device=x.device, dtype=torch.int32)
        check(lib.gim_maxpool_lrelu_fwd(_p(x), _p(y), idx.data_ptr(), N, H * W, C, LRELU_SLOPE, _stream()), "maxpool_lrelu_fwd")
        ctx.save_for_backward(y, idx)
        ctx.cfg = (N, H, W, C)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, idx = ctx.saved_tensors
        return MaxPoolLreluBwdFn.apply(dy, y.detach(), idx, ctx.cfg)


class MaxPoolLreluBwdFn(Function):
    """dx of MaxPoolLreluFn: routes lrelu'(y) * dy to the arg-max pixel (linear in dy; routing is piecewise constant)."""

    @staticmethod
    def forward(ctx, dy, y, idx, cfg):
        lib = _lib.load()
        N, H, W, C = cfg
        dy = _req(dy, "dy")
        dx = torch.empty((N, H, W, C), device=dy.device, dtype=torch.float32)
        check(lib.gim_maxpool_lrelu_bwd(_p(dy), _p(y), idx.data_ptr(), _p(dx), N, H * W, C, LRELU_SLOPE, _stream()), "maxpool_lrelu_bwd")
        ctx.save_for_backward(y, idx)
        ctx.cfg = cfg
        return dx

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        y, idx = ctx.saved_tensors
        N, H, W, C = ctx.cfg
        g = _req(g, "g")
        gdy = torch.empty((N, C), device=g.device, dtype=torch.float32)
        check(lib.gim_maxpool_gather(_p(g), _p(y), idx.data_ptr(), _p(gdy), N, H * W, C, LRELU_SLOPE, _stream()), "maxpool_gather")
        return gdy, None, None, None


class TanhFn(Function):
    @staticmethod
    def forward(ctx, x):
        lib = _lib.load()
        x = _req(x, "x")
        y = torch.empty_like(x)
        check(lib.gim_tanh_fwd(_p(x), _p(y), x.numel(), _stream()), "tanh_fwd")
        ctx.save_for_backward(y)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        lib = _lib.load()
        (y,) = ctx.saved_tensors
        dy = _req(dy, "dy")
        dx = torch.empty_like(y)
        check(lib.gim_tanh_bwd(_p(dy), _p(y), _p(dx), y.numel(), _stream()), "tanh_bwd")
        return dx


class ScaleAddFn(Function):
    """y = gamma * a + x with gamma a 1-element parameter (SelfAttention output).  post_slope != 1: y is stored activated,
    lrelu(y, post_slope), for consumers that take activated storage (ConvFn x_act, AvgPool2Fn in_slope); they hand back the
    gradient w.r.t. the raw y, and this backward never reads y."""

    @staticmethod
    def forward(ctx, a, x, gamma, post_slope=1.0):
        lib = _lib.load()
        a, x = _req(a, "a"), _req(x, "x")
        y = torch.empty_like(x)
        if post_slope != 1.0:
            check(lib.gim_scale_add_fwd_act(_p(a), _p(x), _p(gamma), _p(y), x.numel(), post_slope, _stream()), "scale_add_fwd_act")
        else:
            check(lib.gim_scale_add_fwd(_p(a), _p(x), _p(gamma), _p(y), x.numel(), _stream()), "scale_add_fwd")
        ctx.save_for_backward(a, gamma)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        a, gamma = ctx.saved_tensors
        dy = _req(dy, "dy")
        if _second_order():
            return (MulScalarFn.apply(dy, gamma) if ctx.needs_input_grad[0] else None), dy, None, None
        da = torch.empty_like(a)
        dg = torch.empty(1, device=dy.device, dtype=torch.float32)
        scratch = torch.empty(2048, device=dy.device, dtype=torch.float32)
        check(lib.gim_scale_add_bwd(_p(dy), _p(a), _p(gamma), _p(da), _p(dg), _p(scratch), a.numel(), _stream()), "scale_add_bwd")
        return da, dy, dg.view_as(gamma), None


class MulScalarFn(Function):
    """y = gamma * x with gamma a 1-element tensor (the attention branch of ScaleAddFn's backward, made differentiable)."""

    @staticmethod
    def forward(ctx, x, gamma):
        y, _ = MulScalarFn._run(x, x, gamma)
        ctx.save_for_backward(x, gamma)
        return y

    @staticmethod
    def _run(dy, a, gamma):
        """(gamma * dy, sum(dy * a)) - one launch of the scale_add backward kernel."""
        lib = _lib.load()
        dy, a = _req(dy, "dy"), _req(a, "a")
        da = torch.empty_like(dy)
        dg = torch.empty(1, device=dy.device, dtype=torch.float32)
        scratch = torch.empty(2048, device=dy.device, dtype=torch.float32)
        check(lib.gim_scale_add_bwd(_p(dy), _p(a), _p(gamma), _p(da), _p(dg), _p(scratch), a.numel(), _stream()), "scale_add_bwd")
        return da, dg

    @staticmethod
    def backward(ctx, g):
        x, gamma = ctx.saved_tensors
        gx, gg = MulScalarFn._run(g, x, gamma)
        return gx, gg.view_as(gamma)


class ToNHWCFn(Function):
    """[N, C, H, W] contiguous -> [N, H, W, C]."""

    @staticmethod
    def forward(ctx, x):
        lib = _lib.load()
        x = _req(x, "x")
        N, C, H, W = x.shape
        ctx.cfg = (N, C, H, W)
        if C == 1:
            return x.view(N, H, W, 1)
        y = torch.empty((N, H, W, C), device=x.device, dtype=torch.float32)
        check(lib.gim_nchw_to_nhwc(_p(x), _p(y), N, C, H * W, _stream()), "nchw_to_nhwc")
        return y

    @staticmethod
    def backward(ctx, dy):
        return ToNCHWFn.apply(dy)


class ToNCHWFn(Function):
    """[N, H, W, C] contiguous -> [N, C, H, W]."""

    @staticmethod
    def forward(ctx, x):
        lib = _lib.load()
        x = _req(x, "x")
        N, H, W, C = x.shape
        ctx.cfg = (N, C, H, W)
        if C == 1:
            return x.view(N, 1, H, W)
        y = torch.empty((N, C, H, W), device=x.device, dtype=torch.float32)
        check(lib.gim_nhwc_to_nchw(_p(x), _p(y), N, C, H * W, _stream()), "nhwc_to_nchw")
        return y

    @staticmethod
    def backward(ctx, dy):
        return ToNHWCFn.apply(dy)


# --------------------------------------------------------------------------------------------
# self-attention core
# --------------------------------------------------------------------------------------------
def _bgemm(A, B, C, batch, M, N, K, sA, sB):
    if _FLOPS is not None:
        ent = _FLOPS.setdefault(("bgemm", (batch, M, N, K)), [0, 2.0 * batch * M * N * K])
        ent[0] += 1
    check(_lib.load().gim_bgemm(_p(A), _p(B), _p(C), batch, M, N, K, sA[0], sA[1], sA[2], sB[0], sB[1], sB[2], _stream()), "bgemm")


class BgemmFn(Function):
    """C[b] = op(A[b]) @ op(B[b]) for contiguous [batch, rows, cols] operands, op = transpose when tA / tB.
    Bilinear: its backward is two more batched GEMMs, expressed through BgemmFn so that they are differentiable too."""

    @staticmethod
    def forward(ctx, A, B, tA, tB):
        A, B = _req(A, "A"), _req(B, "B")
        nb, ra, ca = A.shape
        _, rb, cb = B.shape
        M, K = (ca, ra) if tA else (ra, ca)
        K2, N = (cb, rb) if tB else (rb, cb)
        if K != K2 or B.shape[0] != nb:
            raise RuntimeError("bgemm: inner dimensions differ (%s %s, tA=%d tB=%d)" % (tuple(A.shape), tuple(B.shape), tA, tB))
        C = torch.empty((nb, M, N), device=A.device, dtype=torch.float32)
        _bgemm(A, B, C, nb, M, N, K, (ra * ca, 1, ca) if tA else (ra * ca, ca, 1), (rb * cb, 1, cb) if tB else (rb * cb, cb, 1))
        ctx.save_for_backward(A, B)
        ctx.t = (tA, tB)
        return C

    @staticmethod
    def backward(ctx, dC):
        A, B = ctx.saved_tensors
        tA, tB = ctx.t
        dA = dB = None
        if ctx.needs_input_grad[0]:
            # C = A op(B): dA = dC op(B)^T ;  C = A^T op(B): dA = op(B) dC^T
            dA = BgemmFn.apply(B, dC, tB, 1) if tA else BgemmFn.apply(dC, B, 0, 1 - tB)
        if ctx.needs_input_grad[1]:
            # C = op(A) B: dB = op(A)^T dC ;  C = op(A) B^T: dB = dC^T op(A)
            dB = BgemmFn.apply(dC, A, 1, tA) if tB else BgemmFn.apply(A, dC, 1 - tA, 0)
        return dA, dB, None, None


class SoftmaxDim1Fn(Function):
    """softmax over dim 1 of [B, R, C] (the reference's attention normalises over the key index, dim -2)."""

    @staticmethod
    def forward(ctx, S):
        S = _req(S, "S")
        Nb, R, C = S.shape
        P = torch.empty_like(S)
        check(_lib.load().gim_softmax_dim1_fwd(_p(S), _p(P), Nb, R, C, _stream()), "softmax_dim1_fwd")
        ctx.save_for_backward(P)
        return P

    @staticmethod
    def backward(ctx, dP):
        (P,) = ctx.saved_tensors
        return SoftmaxDim1BwdFn.apply(dP, P)


class SoftmaxDim1BwdFn(Function):
    """dS = P * (dP - colsum(P * dP)) as an operator of (dP, P)."""

    @staticmethod
    def forward(ctx, dP, P):
        dP, P = _req(dP, "dP"), _req(P, "P")
        Nb, R, C = P.shape
        dS = torch.empty_like(P)
        check(_lib.load().gim_softmax_dim1_bwd(_p(dP), _p(P), _p(dS), Nb, R, C, _stream()), "softmax_dim1_bwd")
        ctx.save_for_backward(dP, P)
        return dS

    @staticmethod
    def backward(ctx, gS):
        lib = _lib.load()
        dP, P = ctx.saved_tensors
        gS = _req(gS, "gS")
        Nb, R, C = P.shape
        g_dP = g_P = None
        if ctx.needs_input_grad[0]:   # the Jacobian w.r.t. dP is symmetric
            g_dP = torch.empty_like(P)
            check(lib.gim_softmax_dim1_bwd(_p(gS), _p(P), _p(g_dP), Nb, R, C, _stream()), "softmax_dim1_bwd")
        if ctx.needs_input_grad[1]:
            g_P = torch.empty_like(P)
            check(lib.gim_softmax_dim1_bwd_dp(_p(gS), _p(dP), _p(P), _p(g_P), Nb, R, C, _stream()), "softmax_dim1_bwd_dp")
        return g_dP, g_P


class AttnProbFn(Function):
    """A[b] = softmax over dim 1 of f[b] g[b]^T in ONE kernel (gim_attn_prob_fwd: the T x T energy never goes through memory) for
    the shape the benchmark networks use (T = 256 tokens, C/8 = 16 channels).  Its backward is the unfused one, expressed through
    SoftmaxDim1BwdFn and BgemmFn so that it stays differentiable (the R1 term differentiates the discriminator's attention twice)."""

    @staticmethod
    def supported(f, g):
        return f.dim() == 3 and f.shape[1] == 256 and f.shape[2] == 16 and g.shape == f.shape

    @staticmethod
    def forward(ctx, f, g):
        f, g = _req(f, "f"), _req(g, "g")
        nb, T, K = f.shape
        A = torch.empty((nb, T, T), device=f.device, dtype=torch.float32)
        if _FLOPS is not None:
            ent = _FLOPS.setdefault(("bgemm", (nb, T, T, K)), [0, 2.0 * nb * T * T * K])
            ent[0] += 1
        check(_lib.load().gim_attn_prob_fwd(_p(f), _p(g), _p(A), nb, T, K, _stream()), "attn_prob_fwd")
        ctx.save_for_backward(f, g, A)
        return A

    @staticmethod
    def backward(ctx, dA):
        f, g, A = ctx.saved_tensors
        dS = SoftmaxDim1BwdFn.apply(dA, A)
        df = BgemmFn.apply(dS, g, 0, 0) if ctx.needs_input_grad[0] else None     # S = f g^T: dS g
        dg = BgemmFn.apply(dS, f, 1, 0) if ctx.needs_input_grad[1] else None     #            dS^T f
        return df, dg


def attn_core(f, g, h):
    """out[b, j, :] = sum_i softmax_i(f[b,i,:] . g[b,j,:]) * h[b,i,:]   (tokens = pixels, NHWC rows)."""
    if _FUSED_ATTN and AttnProbFn.supported(f, g):
        A = AttnProbFn.apply(f, g)
    else:
        S = BgemmFn.apply(f, g, 0, 1)        # S[i, j] = f_i . g_j
        A = SoftmaxDim1Fn.apply(S)
    return BgemmFn.apply(A, h, 1, 0)     # out = A^T h


_FUSED_ATTN = os.environ.get("GIM_NO_FUSED_ATTN") is None   # A/B switch (host side)


# --------------------------------------------------------------------------------------------
# set pooling (authenticator head) and small glue of the generator
# --------------------------------------------------------------------------------------------
class SumDim1Fn(Function):
    """y[b] = scale * sum_j x[b, j]  for x [B, t, D]."""

    @staticmethod
    def forward(ctx, x, scale):
        lib = _lib.load()
        x = _req(x, "x")
        B, t, D = x.shape
        y = torch.empty((B, D), device=x.device, dtype=torch.float32)
        check(lib.gim_sum_dim1(_p(x), _p(y), B, t, D, scale, _stream()), "sum_dim1")
        ctx.cfg = (B, t, D, scale)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        lib = _lib.load()
        B, t, D, scale = ctx.cfg
        dy = _req(dy, "dy")
        dx = torch.empty((B, t, D), device=dy.device, dtype=torch.float32)
        check(lib.gim_repeat_dim1(_p(dy), _p(dx), B, t, D, scale, _stream()), "repeat_dim1")
        return dx, None


class RepeatDim1Fn(Function):
    """y[b, j] = x[b] for j < t: [B, D] -> [B, t, D]."""

    @staticmethod
    def forward(ctx, x, t):
        lib = _lib.load()
        x = _req(x, "x")
        B, D = x.shape
        y = torch.empty((B, t, D), device=x.device, dtype=torch.float32)
        check(lib.gim_repeat_dim1(_p(x), _p(y), B, t, D, 1.0, _stream()), "repeat_dim1")
        ctx.cfg = (B, t, D)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        lib = _lib.load()
        B, t, D = ctx.cfg
        dy = _req(dy, "dy")
        dx = torch.empty((B, D), device=dy.device, dtype=torch.float32)
        check(lib.gim_sum_dim1(_p(dy), _p(dx), B, t, D, 1.0, _stream()), "sum_dim1")
        return dx, None


def mean_dim1(x):
    return SumDim1Fn.apply(x, 1.0 / x.shape[1])


class NoiseCombineFn(Function):
    """noisy[b, j] = env[b] + w[b, j] - mean_j w[b, j]  (mean term iff remove_mean)."""

    @staticmethod
    def forward(ctx, env, w, remove_mean):
        lib = _lib.load()
        env, w = _req(env, "env"), _req(w, "w")
        B, t, D = w.shape
        y = torch.empty_like(w)
        check(lib.gim_noise_combine(_p(env), _p(w), _p(y), B, t, D, 1 if remove_mean else 0, _stream()), "noise_combine")
        ctx.cfg = (B, t, D, remove_mean)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        lib = _lib.load()
        B, t, D, remove_mean = ctx.cfg
        dy = _req(dy, "dy")
        denv = torch.empty((B, D), device=dy.device, dtype=torch.float32)
        check(lib.gim_sum_dim1(_p(dy), _p(denv), B, t, D, 1.0, _stream()), "sum_dim1")
        dw = torch.empty_like(dy)
        check(lib.gim_noise_combine(None, _p(dy), _p(dw), B, t, D, 1 if remove_mean else 0, _stream()), "noise_combine")
        return denv, dw, None


class Concat2Fn(Function):
    """cat((a, b broadcast), channel) for NHWC a [R_img, H, W, Ca] and b [R_img/rep, H, W, Cb]
    (each b image serves `rep` consecutive a images).  Gradient flows to a only."""

    @staticmethod
    def forward(ctx, a, b, rep):
        lib = _lib.load()
        a, b = _req(a, "a"), _req(b, "b")
        Ni, H, W, Ca = a.shape
        Cb = b.shape[3]
        y = torch.empty((Ni, H, W, Ca + Cb), device=a.device, dtype=torch.float32)
        check(lib.gim_concat2(_p(a), _p(b), _p(y), Ni * H * W, Ca, Cb, H * W, rep, _stream()), "concat2")
        ctx.cfg = (Ni, H, W, Ca, Cb)
        ctx.rep = rep
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        lib = _lib.load()
        Ni, H, W, Ca, Cb = ctx.cfg
        dy = _req(dy, "dy")
        da = db = None
        if ctx.needs_input_grad[0]:
            da = torch.empty((Ni, H, W, Ca), device=dy.device, dtype=torch.float32)
            check(lib.gim_slice_channels(_p(dy), _p(da), Ni * H * W, Ca, Ca + Cb, _stream()), "slice_channels")
        if ctx.needs_input_grad[1]:
            if ctx.rep != 1:
                raise RuntimeError("concat2: gradient w.r.t. a broadcast second operand is not implemented")
            db = torch.empty((Ni, H, W, Cb), device=dy.device, dtype=torch.float32)
            check(lib.gim_slice_channels(_p(dy, Ca), _p(db), Ni * H * W, Cb, Ca + Cb, _stream()), "slice_channels")
        return da, db, None


class HeadCatFn(Function):
    """The [B, 2*(Ds + 2*De + Df)] input of the authenticator's MLP:
    cat(mean(test_src), mean(si_src), [mean, custom_std](test_env), mean(fc_test), [mean, custom_std](si_env), mean(fc_si))
    where fc_* are the per-sample outputs of the FC-stat MLP."""

    @staticmethod
    def forward(ctx, test_src, test_env, si_src, si_env, fc_test, fc_si):
        lib = _lib.load()
        ts, te, ss, se, ft, fs = [_req(t, "head input") for t in (test_src, test_env, si_src, si_env, fc_test, fc_si)]
        B, n, Ds = ts.shape
        k = ss.shape[1]
        De, Df = te.shape[2], ft.shape[2]
        L = 2 * (Ds + 2 * De + Df)
        out = torch.empty((B, L), device=ts.device, dtype=torch.float32)
        st = _stream()
        o_te = 2 * Ds
        o_se = o_te + 2 * De + Df
        check(lib.gim_set_stats_fwd(_p(ts), _p(out, 0), None, B, n, Ds, L, L, st), "set_stats_fwd")
        check(lib.gim_set_stats_fwd(_p(ss), _p(out, Ds), None, B, k, Ds, L, L, st), "set_stats_fwd")
        check(lib.gim_set_stats_fwd(_p(te), _p(out, o_te), _p(out, o_te + De), B, n, De, L, L, st), "set_stats_fwd")
        check(lib.gim_set_stats_fwd(_p(ft), _p(out, o_te + 2 * De), None, B, n, Df, L, L, st), "set_stats_fwd")
        check(lib.gim_set_stats_fwd(_p(se), _p(out, o_se), _p(out, o_se + De), B, k, De, L, L, st), "set_stats_fwd")
        check(lib.gim_set_stats_fwd(_p(fs), _p(out, o_se + 2 * De), None, B, k, Df, L, L, st), "set_stats_fwd")
        ctx.save_for_backward(ts, te, ss, se, ft, fs)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        ts, te, ss, se, ft, fs = ctx.saved_tensors
        dout = _req(dout, "dout")
        B, n, Ds = ts.shape
        k = ss.shape[1]
        De, Df = te.shape[2], ft.shape[2]
        L = dout.shape[1]
        st = _stream()
        o_te = 2 * Ds
        o_se = o_te + 2 * De + Df
        need = ctx.needs_input_grad
        grads = [None] * 6

        second = _second_order()

        def bwd(i, x, t, D, o_mean, o_std):
            if not need[i]:
                return
            if second:
                grads[i] = SetStatsBwdFn.apply(dout, x, o_mean, o_std)
                return
            dx = torch.empty_like(x)
            check(lib.gim_set_stats_bwd(_p(x), _p(dout, o_mean), (_p(dout, o_std) if o_std is not None else None), _p(dx),
                                        B, t, D, L, L, st), "set_stats_bwd")
            grads[i] = dx

        bwd(0, ts, n, Ds, 0, None)
        bwd(1, te, n, De, o_te, o_te + De)
        bwd(2, ss, k, Ds, Ds, None)
        bwd(3, se, k, De, o_se, o_se + De)
        bwd(4, ft, n, Df, o_te + 2 * De, None)
        bwd(5, fs, k, Df, o_se + 2 * De, None)
        return tuple(grads)


class StatCatFn(Function):
    """cat(mean(x, 1), custom_std(x), mean(fc, 1)) for x [B, t, D] and the per-sample FC features fc [B, t, Df] -> [B, 2 D + Df]
    (GIMMeanStdFcStat.forward, models/gim_basic_models.py:152-172, as a standalone operator; the authenticator head uses the
    two-set form HeadCatFn)."""

    @staticmethod
    def forward(ctx, x, fc):
        lib = _lib.load()
        x, fc = _req(x, "x"), _req(fc, "fc")
        B, t, D = x.shape
        Df = fc.shape[2]
        L = 2 * D + Df
        out = torch.empty((B, L), device=x.device, dtype=torch.float32)
        st = _stream()
        check(lib.gim_set_stats_fwd(_p(x), _p(out, 0), _p(out, D), B, t, D, L, L, st), "set_stats_fwd")
        check(lib.gim_set_stats_fwd(_p(fc), _p(out, 2 * D), None, B, t, Df, L, L, st), "set_stats_fwd")
        ctx.save_for_backward(x, fc)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        x, fc = ctx.saved_tensors
        dout = _req(dout, "dout")
        B, t, D = x.shape
        Df = fc.shape[2]
        L = dout.shape[1]
        st = _stream()
        second = _second_order()
        grads = [None, None]
        for i, (src, Dn, o_mean, o_std) in enumerate(((x, D, 0, D), (fc, Df, 2 * D, None))):
            if not ctx.needs_input_grad[i]:
                continue
            if second:
                grads[i] = SetStatsBwdFn.apply(dout, src, o_mean, o_std)
                continue
            dx = torch.empty_like(src)
            check(lib.gim_set_stats_bwd(_p(src), _p(dout, o_mean), (_p(dout, o_std) if o_std is not None else None), _p(dx),
                                        B, t, Dn, L, L, st), "set_stats_bwd")
            grads[i] = dx
        return tuple(grads)


stat_cat = StatCatFn.apply


class SetStatsBwdFn(Function):
    """dx[b, j] = dout[b, o_mean:] / t + dout[b, o_std:] * (x[b, j] - mean) / ((t - 1) * custom_std)  as an operator of (dout, x):
    the backward of one mean / [mean, custom_std] slot of the authenticator head, differentiable for the R1 term."""

    @staticmethod
    def forward(ctx, dout, x, o_mean, o_std):
        dout, x = _req(dout, "dout"), _req(x, "x")
        B, t, D = x.shape
        L = dout.shape[1]
        dx = torch.empty_like(x)
        check(_lib.load().gim_set_stats_bwd(_p(x), _p(dout, o_mean), (_p(dout, o_std) if o_std is not None else None), _p(dx),
                                            B, t, D, L, L, _stream()), "set_stats_bwd")
        ctx.save_for_backward(dout, x)
        ctx.o = (o_mean, o_std)
        return dx

    @staticmethod
    def backward(ctx, g):
        dout, x = ctx.saved_tensors
        o_mean, o_std = ctx.o
        g = _req(g, "g")
        B, t, D = x.shape
        L = dout.shape[1]
        g_dout = torch.zeros_like(dout) if ctx.needs_input_grad[0] else None
        gx = torch.empty_like(x) if (ctx.needs_input_grad[1] and o_std is not None) else None
        check(_lib.load().gim_set_stats_bwd_bwd(_p(x), (_p(dout, o_std) if o_std is not None else None), _p(g),
                                                (_p(g_dout, o_mean) if g_dout is not None else None),
                                                (_p(g_dout, o_std) if (g_dout is not None and o_std is not None) else None),
                                                _p(gx), B, t, D, L, L, _stream()), "set_stats_bwd_bwd")
        return g_dout, gx, None, None


class SqSumRowsFn(Function):
    """out[b] = sum_i x[b, i]^2  (the g.pow(2).view(B, -1).sum(1) of compute_grad2, training/utils.py:122)."""

    @staticmethod
    def forward(ctx, x):
        x = _req(x, "x")
        B, L = x.shape
        out = torch.empty(B, device=x.device, dtype=torch.float32)
        check(_lib.load().gim_sqsum_rows_fwd(_p(x), _p(out), B, L, _stream()), "sqsum_rows_fwd")
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, dout):
        (x,) = ctx.saved_tensors
        if torch.is_grad_enabled():
            raise NotImplementedError("third-order gradients are not supported")
        dout = _req(dout, "dout")
        B, L = x.shape
        dx = torch.empty_like(x)
        check(_lib.load().gim_sqsum_rows_bwd(_p(x), _p(dout), _p(dx), B, L, _stream()), "sqsum_rows_bwd")
        return dx


sqsum_rows = SqSumRowsFn.apply


class BCELogitsFn(Function):
    """binary_cross_entropy_with_logits(x, full_like(x, target), reduction='none')."""

    @staticmethod
    def forward(ctx, x, target):
        lib = _lib.load()
        x = _req(x, "x")
        loss = torch.empty_like(x)
        check(lib.gim_bce_logits_fwd(_p(x), _p(loss), target, x.numel(), _stream()), "bce_logits_fwd")
        ctx.save_for_backward(x)
        ctx.target = target
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, dl):
        lib = _lib.load()
        (x,) = ctx.saved_tensors
        dl = _req(dl, "dloss")
        dx = torch.empty_like(x)
        check(lib.gim_bce_logits_bwd(_p(dl), _p(x), _p(dx), ctx.target, x.numel(), _stream()), "bce_logits_bwd")
        return dx, None


avg_pool2 = AvgPool2Fn.apply
maxpool_lrelu = MaxPoolLreluFn.apply
tanh = TanhFn.apply
scale_add = ScaleAddFn.apply
to_nhwc = ToNHWCFn.apply
to_nchw = ToNCHWFn.apply
noise_combine = NoiseCombineFn.apply
concat2 = Concat2Fn.apply
head_cat = HeadCatFn.apply
bce_logits = BCELogitsFn.apply


def repeat_dim1(x, t):
    return RepeatDim1Fn.apply(x, t)


class MeanStdCatFn(Function):
    """cat([mean(x_i, 1), custom_std(x_i)] for each x_i [B, t_i, D]) -> [B, 2*D*len(xs)]
    (GIMMeanStdStat of both sample sets + torch.cat, models/gim_gaussian_models.py:37-39)."""

    @staticmethod
    def forward(ctx, *xs):
        lib = _lib.load()
        xs = [_req(x, "x") for x in xs]
        B, _, D = xs[0].shape
        L = 2 * D * len(xs)
        out = torch.empty((B, L), device=xs[0].device, dtype=torch.float32)
        st = _stream()
        for i, x in enumerate(xs):
            check(lib.gim_set_stats_fwd(_p(x), _p(out, 2 * D * i), _p(out, 2 * D * i + D), B, x.shape[1], D, L, L, st), "set_stats_fwd")
        ctx.save_for_backward(*xs)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        xs = ctx.saved_tensors
        dout = _req(dout, "dout")
        B, _, D = xs[0].shape
        L = dout.shape[1]
        st = _stream()
        grads = []
        second = _second_order()
        for i, x in enumerate(xs):
            if not ctx.needs_input_grad[i]:
                grads.append(None)
                continue
            if second:
                grads.append(SetStatsBwdFn.apply(dout, x, 2 * D * i, 2 * D * i + D))
                continue
            dx = torch.empty_like(x)
            check(lib.gim_set_stats_bwd(_p(x), _p(dout, 2 * D * i), _p(dout, 2 * D * i + D), _p(dx), B, x.shape[1], D, L, L, st), "set_stats_bwd")
            grads.append(dx)
        return tuple(grads)


def mean_std_cat(*xs):
    return MeanStdCatFn.apply(*xs)


class ImgAttMixFn(Function):
    """out = x1 * a1 + v2 * a2 with (a1, a2) = softmax(sum_c q1*k1, sum_c q2*k2)  (ImgAttention, NHWC)."""

    @staticmethod
    def forward(ctx, q1, k1, q2, k2, x1, v2):
        lib = _lib.load()
        ts = [_req(t, "img_att input") for t in (q1, k1, q2, k2, x1, v2)]
        N, H, W, C = ts[0].shape
        out = torch.empty_like(ts[0])
        att = torch.empty((N, H, W), device=out.device, dtype=torch.float32)
        check(lib.gim_img_att_mix_fwd(*[_p(t) for t in ts], _p(out), _p(att), N * H * W, C, _stream()), "img_att_mix_fwd")
        ctx.save_for_backward(*ts, att)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        lib = _lib.load()
        q1, k1, q2, k2, x1, v2, att = ctx.saved_tensors
        dout = _req(dout, "dout")
        N, H, W, C = q1.shape
        gs = [torch.empty_like(q1) for _ in range(6)]
        check(lib.gim_img_att_mix_bwd(_p(dout), _p(q1), _p(k1), _p(q2), _p(k2), _p(x1), _p(v2), _p(att), *[_p(g) for g in gs],
                                      N * H * W, C, _stream()), "img_att_mix_bwd")
        return tuple(gs)


img_att_mix = ImgAttMixFn.apply
