"""Building blocks of the GIM image agents on the MI355X engine.

Host-side mirror of the used subset of the reference's ``models/model_blocks.py``: same class names,
constructor arguments, sub-module / parameter names and registration order (so ``state_dict()`` keys,
their order and ``parameters()`` order are those of the reference, incl. the spectral-norm
``weight_orig`` / ``weight_u`` / ``weight_v`` triple), but every ``forward`` runs hand-written HIP
kernels (``ops.py`` -> ``libgim_hip.so``) on NHWC activations, with the element-wise neighbours of each
convolution fused into it:

  LeakyReLU in front of a conv                     -> conv prologue (gather)
  1/sigma of spectral norm, bias, residual add     -> conv epilogue
  AvgPool2d(2) behind a conv                       -> folded into ONE stride-2 conv (16/36 of the FLOPs for 3x3)
  nearest-upsample in front of a KxK conv          -> sub-pixel form on the low-resolution input (same ratio)
  1x1 skip convs next to a pool / upsample         -> computed at the low resolution

Unless noted, tensors between blocks are NHWC ``[N, H, W, C]`` float32 on the GPU.
"""
import collections
import math

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from . import ops

LRELU = 0.2


def _kaiming_uniform_like_torch(weight, bias, fan_in):
    """nn.Conv2d / nn.Linear default init (kaiming_uniform_(a=sqrt(5)); bias U(+-1/sqrt(fan_in)))."""
    nn.init.kaiming_uniform_(weight, a=math.sqrt(5))
    if bias is not None:
        bound = 1.0 / math.sqrt(fan_in) if fan_in > 0 else 0.0
        nn.init.uniform_(bias, -bound, bound)


class GimLinear(nn.Module):
    """nn.Linear (reference: models/model_blocks.py:86,89,786-789) with an optional fused LeakyReLU on its input."""

    def __init__(self, in_features, out_features):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features))
        _kaiming_uniform_like_torch(self.weight, self.bias, in_features)

    def forward(self, x, pre_slope=1.0):
        return ops.linear(x, self.weight, self.bias, pre_slope)


class FusedLeakyReLU(nn.Module):
    """Placeholder keeping the reference's nn.Sequential indices (Linear at 0, 2, 4, ...): the activation
    itself is applied in the prologue of the following linear kernel."""

    def forward(self, x):  # pragma: no cover - never called, see MLP.forward
        raise RuntimeError("FusedLeakyReLU is fused into the next GimLinear")


class MLP(nn.Module):
    """models/model_blocks.py:77-94."""

    def __init__(self, layer_dims):
        super().__init__()
        assert len(layer_dims) >= 2
        layers = []
        inp_dim = layer_dims[0]
        for out_dim in layer_dims[1:-1]:
            layers.append(GimLinear(inp_dim, out_dim))
            layers.append(FusedLeakyReLU())
            inp_dim = out_dim
        layers.append(GimLinear(inp_dim, layer_dims[-1]))
        self.model = nn.Sequential(*layers)

    def forward(self, x):
        slope = 1.0
        for layer in self.model:
            if isinstance(layer, FusedLeakyReLU):
                slope = LRELU
            else:
                x = layer(x, pre_slope=slope)
                slope = 1.0
        return x


class SNConv2d(nn.Module):
    """nn.utils.spectral_norm(nn.Conv2d(in, out, k, padding=(k-1)//2)) restated
    (call sites models/model_blocks.py:492-495,522-526,744-750,792-793,836-840).

    Parameters ``bias`` and ``weight_orig`` (logical [Cout, Cin, k, k], stored channels-last = the
    [Cout][kh][kw][Cin] layout the kernels read), buffers ``weight_u`` / ``weight_v``.  Each call in
    training mode does one power iteration in place (kernel), then the conv kernel applies 1/sigma in its
    epilogue; backward differentiates through sigma w.r.t. ``weight_orig`` with u, v constant.
    """

    def __init__(self, in_channels, out_channels, kernel_size, padding=0):
        super().__init__()
        assert padding == (kernel_size - 1) // 2, "only 'same' stride-1 convolutions are on the GIM hot path"
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, kernel_size
        w = torch.empty(out_channels, in_channels, kernel_size, kernel_size)
        b = torch.empty(out_channels)
        _kaiming_uniform_like_torch(w, b, in_channels * kernel_size * kernel_size)
        # torch.nn.utils.spectral_norm.apply: u ~ N(0,1)[Cout], v ~ N(0,1)[Cin*k*k], both normalised
        u = nn.functional.normalize(w.new_empty(out_channels).normal_(0, 1), dim=0, eps=1e-12)
        v = nn.functional.normalize(w.new_empty(in_channels * kernel_size * kernel_size).normal_(0, 1), dim=0, eps=1e-12)
        self.bias = nn.Parameter(b)
        self.weight_orig = nn.Parameter(w.contiguous(memory_format=torch.channels_last))
        self.register_buffer("weight_u", u)
        self.register_buffer("weight_v", v)

        self._sn_queue = collections.deque()  # (sigma, u, v) triples precomputed by an SNPlan round
        self._fold_cache = (None, None, None, None, False)   # ((storage, version, optimizer epoch), folded weights F, ready event, stream, recorded in a capture)
        self._wants_fold = False    # set by the first folded() call: SNPlan.run then folds this conv together with the others

    def folded(self):
        """The (k+1)^2-tap folded weights for the pool / sub-pixel forms, recomputed only when weight_orig changed
        (autograd version counter for torch-side writes, optim.weights_epoch(w) for the fused Adam kernel)."""
        from . import optim
        w = self.weight_orig
        self._wants_fold = True
        key = (w.data_ptr(), w._version, optim.weights_epoch(w))
        raw = ops._stream()   # raw handle: building a torch Stream object per call costs microseconds of host time
        if self._fold_cache[0] != key:
            with torch.no_grad():
                f = ops._folded(ops.weight_phys(w), self.out_channels, self.in_channels, self.kernel_size)
            # other streams (one per encoder pass) reuse F: they wait for the kernel that wrote it
            self._fold_cache = (key, f, torch.cuda.current_stream().record_event(), raw, torch.cuda.is_current_stream_capturing())
        elif self._fold_cache[3] != raw:
            cur = torch.cuda.current_stream()
            if self._fold_cache[4] or not torch.cuda.is_current_stream_capturing():   # see ops.WgradQueue.take
                cur.wait_event(self._fold_cache[2])
            self._fold_cache[1].record_stream(cur)   # F was allocated on another stream: keep its memory until this one is done with it
        return self._fold_cache[1]

    def forward(self, x, res=None, ups=0, pre_slope=1.0, pool=False, res_ups=False, post_slope=1.0, x_act=False, fork_pool_slope=None):
        """post_slope != 1: returns (y, activated) - see ops.conv2d_post_act; x_act: x was stored activated by such a producer.
        fork_pool_slope (plain convs only): returns (y, activated, avgpool2(raw x)) as one autograd node - ops.conv2d_forkpool."""
        guard = None
        if self._sn_queue:
            sigma, u_s, v_s, guard = self._sn_queue.popleft()
        else:
            sigma, u_s, v_s = ops.spectral_sigma(self.weight_orig, self.weight_u, self.weight_v, self.training)
        if fork_pool_slope is not None:
            assert res is None and not ups and not pool and not res_ups
            return ops.conv2d_forkpool(x, self.weight_orig, self.bias, sigma, u_s, v_s, pre_slope, guard, post_slope, x_act, fork_pool_slope)
        wf = self.folded() if (pool or (ups and self.kernel_size > 1)) else None
        if post_slope != 1.0:
            return ops.conv2d_post_act(x, self.weight_orig, self.bias, res, sigma, u_s, v_s, ups, pre_slope, pool, res_ups, wf, guard, post_slope, x_act)
        return ops.conv2d(x, self.weight_orig, self.bias, res, sigma, u_s, v_s, ups, pre_slope, pool, res_ups, wf, guard, x_act)

    def extra_repr(self):
        return "%d, %d, kernel_size=%d (spectral norm)" % (self.in_channels, self.out_channels, self.kernel_size)


class SNPlan:
    """Runs the spectral-norm power iterations of MANY convs ahead of their use: the iterations depend on the
    weights and on u only, never on activations, so a forward pass that will call each conv `rounds` times runs
    `rounds` batched rounds up front (4 launches each, gim_spectral_sigma_batched) and every SNConv2d call then
    just pops its precomputed (sigma, u, v).  Sequential semantics are those of the per-call hook: round r uses
    the u left by round r-1.  The job table is static and rebuilt only if a tensor moves.

    The per-round outputs live in persistent buffers (two alternating sets), so their addresses are the same every
    step - which lets the batched weight-gradient finish (ops.WgradQueue) and hipGraph capture keep static tables.
    A backward pass that still needs a set which later forwards have overwritten (more than two forwards of the same
    model before its backward) is refused by ops.ConvFn rather than computed from stale values."""

    _JOB = np.dtype([("w", "<u8"), ("u", "<u8"), ("v", "<u8"), ("off_sigma", "<i8"), ("off_u", "<i8"), ("off_v", "<i8"),
                     ("off_scratch", "<i8"), ("Cout", "<i4"), ("Cin", "<i4"), ("KH", "<i4"), ("reserved", "<i4")])

    def __init__(self, convs):
        self.convs = [c for c in convs if isinstance(c, SNConv2d)]
        self._key = None
        self._gen = 0      # number of run() calls so far; run g writes buffer set g & 1
        self._bufs = {}

    def stale(self, gen):
        """True when the outputs of run number `gen` have been overwritten."""
        return self._gen - gen > 2

    def _build(self, key):
        dev = self.convs[0].weight_orig.device
        jobs = np.zeros(len(self.convs), dtype=self._JOB)
        cols, rows, self._views = [], [], []
        off = 0

        def take(n):
            nonlocal off
            o = off
            off += (n + 3) // 4 * 4
            return o
        for j, c in enumerate(self.convs):
            Cout, Cin, KH = c.out_channels, c.in_channels, c.kernel_size
            K = Cin * KH * KH
            if not c.weight_orig.permute(0, 2, 3, 1).is_contiguous():
                raise RuntimeError("SNPlan: weight_orig must be stored channels-last")
            o_s, o_u, o_v, o_scr = take(1), take(Cout), take(K), take(10 * K + Cout)
            jobs[j] = (c.weight_orig.data_ptr(), c.weight_u.data_ptr(), c.weight_v.data_ptr(), o_s, o_u, o_v, o_scr, Cout, Cin, KH, 0)
            R = min(8, (Cout + 63) // 64)
            rows_per = (Cout + R - 1) // R
            for xb in range((K + 255) // 256):
                for r in range(R):
                    cols.append((j, xb, r, rows_per))
            for rb in range((Cout + 3) // 4):
                rows.append((j, rb))
            self._views.append((o_s, o_u, Cout, o_v, K))
        self._total = off
        self._jobs = torch.from_numpy(jobs.view(np.uint8).copy()).to(dev)
        self._cols = torch.tensor(cols, dtype=torch.int32).to(dev)
        self._rows = torch.tensor(rows, dtype=torch.int32).to(dev)
        self._n = (len(self.convs), len(cols), len(rows))
        self._key = key

    @torch.no_grad()
    def run(self, rounds, training):
        if not self.convs:
            return
        w0 = self.convs[0].weight_orig
        if not (w0.is_cuda and w0.dtype == torch.float32):
            raise RuntimeError("weights must be CUDA float32 (got %s on %s): the GIM engine has no CPU path" % (w0.dtype, w0.device))
        key = tuple(t.data_ptr() for c in self.convs for t in (c.weight_orig, c.weight_u, c.weight_v))
        if key != self._key:
            self._build(key)
        lib = _lib.load()
        dev = self.convs[0].weight_orig.device
        for c in self.convs:
            c._sn_queue.clear()
        gen = self._gen
        self._gen += 1
        for r in range(rounds):
            out = self._bufs.get((gen & 1, r))
            if out is None or out.numel() != self._total or out.device != dev:
                out = self._bufs[(gen & 1, r)] = torch.empty(self._total, device=dev, dtype=torch.float32)
            _lib.check(lib.gim_spectral_sigma_batched(self._jobs.data_ptr(), self._n[0], self._cols.data_ptr(), self._n[1],
                                                      self._rows.data_ptr(), self._n[2], out.data_ptr(), 1 if training else 0,
                                                      torch.cuda.current_stream().cuda_stream), "spectral_sigma_batched")
            for c, (o_s, o_u, Cout, o_v, K) in zip(self.convs, self._views):
                c._sn_queue.append((out[o_s:o_s + 1], out[o_u:o_u + Cout], out[o_v:o_v + K], (self, gen)))
        self._fold_stale()

    def _fold_stale(self):
        """The folded weights (pool / sub-pixel forms) of every conv of the plan whose weights changed since its last fold, in ONE
        launch (each conv would otherwise fold itself at its first use: ~30 launches of ~10 us per training step, on the
        forward's critical path).  Results go where SNConv2d.folded() keeps them."""
        from . import optim
        todo = []
        for c in self.convs:
            if c._wants_fold:
                w = c.weight_orig
                key = (w.data_ptr(), w._version, optim.weights_epoch(w))
                if c._fold_cache[0] != key:
                    todo.append((c, key))
        if len(todo) < 2:
            return     # a single stale conv folds itself at its call
        dev = todo[0][0].weight_orig.device
        sig = tuple(c.weight_orig.data_ptr() for c, _ in todo)
        ent = getattr(self, "_fold_tab", {}).get(sig)
        if ent is None:
            if torch.cuda.is_current_stream_capturing():
                return
            jobs = np.zeros(len(todo), dtype=np.dtype([("w", "<u8"), ("f", "<u8"), ("Cout", "<i4"), ("Cin", "<i4"), ("KH", "<i4"), ("r", "<i4")]))
            bufs, tab = [], []
            for j, (c, _) in enumerate(todo):
                KF = c.kernel_size + 1
                n = c.out_channels * KF * KF * c.in_channels
                if n >= 1 << 31:
                    raise RuntimeError("folded weights of more than 2^31 elements (gim_conv2d_fold_weights_batched indexes with 32 bits)")
                f = torch.empty(n, device=dev, dtype=torch.float32)
                bufs.append(f)
                jobs[j] = (c.weight_orig.data_ptr(), f.data_ptr(), c.out_channels, c.in_channels, c.kernel_size, 0)
                tab += [(j, ch) for ch in range((n + 65535) // 65536)]
            ent = (torch.from_numpy(jobs.view(np.uint8).copy()).to(dev), torch.tensor(tab, dtype=torch.int32).to(dev), len(tab), bufs)
            if not hasattr(self, "_fold_tab"):
                self._fold_tab = {}
            self._fold_tab[sig] = ent
        dj, dt, nb, bufs = ent
        _lib.check(_lib.load().gim_conv2d_fold_weights_batched(dj.data_ptr(), dt.data_ptr(), nb, ops._stream()), "fold_weights_batched")
        ev = torch.cuda.current_stream().record_event()
        raw, cap = ops._stream(), torch.cuda.is_current_stream_capturing()
        for (c, key), f in zip(todo, bufs):
            c._fold_cache = (key, f, ev, raw, cap)


def sn_convs(*modules):
    """All SNConv2d sub-modules of the given modules, in registration order."""
    return [m for mod in modules for m in mod.modules() if isinstance(m, SNConv2d)]


class GimInstanceNorm2d(nn.Module):
    """nn.InstanceNorm2d(C, affine=True) (models/gim_img_models.py:126, models/model_blocks.py:747-748)."""

    def __init__(self, num_features, eps=1e-5):
        super().__init__()
        self.num_features, self.eps = num_features, eps
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))

    def forward(self, x, post_slope=1.0):
        return ops.instance_norm(x, self.weight, self.bias, self.eps, post_slope)


def custom_std(x):
    """models/model_blocks.py:41-48 for x [B, t, D] -> [B, D] (standalone form for the training loop's encoding
    statistics; the authenticator head uses the fused HeadCatFn)."""
    D = x.shape[-1]
    return ops.mean_std_cat(x)[:, D:]


class ResBlockDown(nn.Module):
    """models/model_blocks.py:486-514, with both average pools folded away (exact in real arithmetic):
    avgpool(conv1x1(x)) = conv1x1(avgpool(x)) for the skip (the 1x1 conv runs on a quarter of the pixels; folding the pool INTO
    it as a 2x2-tap stride-2 conv was measured in round 2 and costs more than the pooling kernel it removes: 4x the FLOPs of
    the skip conv, profiles/r02_avgpool_fold_1x1.txt), and avgpool(conv_r2(.)) + skip is ONE stride-2 convolution with the
    2x2-folded weights and the low-resolution skip as its epilogue residual."""

    def __init__(self, in_channel, out_channel, conv_size=3, padding_size=1):
        super().__init__()
        self.conv_l1 = SNConv2d(in_channel, out_channel, 1)
        self.conv_r1 = SNConv2d(in_channel, out_channel, conv_size, padding=padding_size)
        self.conv_r2 = SNConv2d(out_channel, out_channel, conv_size, padding=padding_size)

    def forward(self, x):
        return self.forward_act(x)[0]

    def forward_act(self, x, x_act=False, post_slope=1.0):
        """-> (y, y_act).  x_act: the block input is stored ACTIVATED, lrelu(x) (its producer - the previous block's last conv, an
        InstanceNorm, the attention's output kernel - wrote it that way): conv_r1, whose K loop would otherwise redo the LeakyReLU
        for every tap and output tile (10-15 % of that kernel and of its wgrad: every vector instruction of a wave waits for a
        gap between the other waves' MFMAs, profiles/r03_a_igemm_loop_lab.txt), reads it as it is, and the skip path - which
        needs the RAW x - inverts the activation inside its pooling kernel (LeakyReLU is a bijection).  post_slope != 1 asks for
        the block OUTPUT in activated form in turn; y_act says whether it is (a launch that splits K cannot activate)."""
        # (Round 4 tried the skip path - pool + 1x1 conv, two launches of 5-17 us - on a side stream of the lane, under conv_r1 instead of
        #  in front of it: 295-297 against 418-420 episodes/s.  One more stream per lane couples the two encoder streams that share
        #  it and adds two cross-queue waits per block and direction; profiles/r04_e_skip_stream_rejected.txt.)
        # the two readers of x - conv_r1 behind its LeakyReLU, the skip conv on the pooled RAW x - are ONE autograd node: their
        # gradients meet in conv_r1's dgrad epilogue (ops.ConvForkPoolFn).  conv_r2 is the only reader of conv_r1's output and applies
        # LeakyReLU to it: conv_r1 stores it activated (once per element in its epilogue); launches that split K hand back the raw tensor
        out, act, pooled = self.conv_r1(x, pre_slope=LRELU, post_slope=LRELU, x_act=x_act, fork_pool_slope=LRELU if x_act else 1.0)
        left = self.conv_l1(pooled)
        if post_slope != 1.0 and ops.act_storage():
            return self.conv_r2(out, res=left, pre_slope=LRELU, pool=True, x_act=act, post_slope=post_slope)
        return self.conv_r2(out, res=left, pre_slope=LRELU, pool=True, x_act=act), False


class SelfAttention(nn.Module):
    """models/model_blocks.py:517-549."""

    def __init__(self, in_channel):
        super().__init__()
        self.conv_f = SNConv2d(in_channel, in_channel // 8, 1)
        self.conv_g = SNConv2d(in_channel, in_channel // 8, 1)
        self.conv_h = SNConv2d(in_channel, in_channel, 1)
        self.gamma = nn.Parameter(torch.zeros(1))

    def forward(self, x, post_slope=1.0):
        """post_slope != 1: the output gamma * attention + x is written activated (for a ResBlockDown.forward_act behind it)."""
        N, H, W, C = x.shape
        xf, xg, xh, xr = ops.fork(x, 4)     # four consumers of x: their gradients are added by one kernel, not three
        f = self.conv_f(xf).view(N, H * W, -1)
        g = self.conv_g(xg).view(N, H * W, -1)
        h = self.conv_h(xh).view(N, H * W, C)
        out = ops.attn_core(f, g, h).view(N, H, W, C)
        return ops.scale_add(out, xr, self.gamma, post_slope)


class ImgAttConvBlock(nn.Module):
    """models/model_blocks.py:551-578 (only reachable with use_img_att=True)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv_l1 = SNConv2d(in_channels, out_channels, 1)
        self.conv_r1 = SNConv2d(in_channels, out_channels, 9, padding=4)
        self.conv_r2 = SNConv2d(out_channels, out_channels, 3, padding=1)

    def forward(self, x):
        left = self.conv_l1(x)
        out = self.conv_r1(x, pre_slope=LRELU)
        return self.conv_r2(out, res=left, pre_slope=LRELU)


class ImgAttention(nn.Module):
    """models/model_blocks.py:581-608: per-pixel 2-way attention between the leaked image x1 and the generated image
    x2 (only reached with use_img_att=True; the parameters always exist).  NHWC in, NHWC out."""

    def __init__(self, img1_channels, img2_channels):
        super().__init__()
        self.q1conv = ImgAttConvBlock(img1_channels + img2_channels, img1_channels)
        self.q2conv = ImgAttConvBlock(img1_channels + img2_channels, img1_channels)
        self.k1conv = ImgAttConvBlock(img1_channels, img1_channels)
        self.k2conv = ImgAttConvBlock(img2_channels, img1_channels)
        self.v2conv = ImgAttConvBlock(img2_channels, img1_channels)

    def forward(self, x1, x2):
        x = ops.concat2(x1, x2, 1)
        q1, q2 = self.q1conv(x), self.q2conv(x)
        k1, k2, v2 = self.k1conv(x1), self.k2conv(x2), self.v2conv(x2)
        return ops.img_att_mix(q1, k1, q2, k2, x1, v2)


class ResBlockUp(nn.Module):
    """models/model_blocks.py:733-773."""

    def __init__(self, in_channel, out_channel, out_size=None, scale=2, conv_size=3, padding_size=1, use_norm=True):
        super().__init__()
        assert out_size is None and scale == 2
        self.in_channel, self.out_channel = in_channel, out_channel
        self.conv_l1 = SNConv2d(in_channel, out_channel, 1)
        self.in1 = GimInstanceNorm2d(in_channel)
        self.in2 = GimInstanceNorm2d(out_channel)
        self.conv_r1 = SNConv2d(in_channel, out_channel, conv_size, padding=padding_size)
        self.conv_r2 = SNConv2d(out_channel, out_channel, conv_size, padding=padding_size)

    def forward(self, x):
        # conv1x1(up(x)) = up(conv1x1(x)): the skip is computed at low resolution and upsampled by the residual
        # read of conv_r2; conv_r1(up(.)) runs in its sub-pixel form (ops.ConvFn)
        left = self.conv_l1(x)
        act = ops.act_storage()   # each norm output has ONE reader, a conv behind a LeakyReLU: the norm stores it activated
        out = self.in1(x, post_slope=LRELU)
        out = self.conv_r1(out, ups=1, pre_slope=LRELU, x_act=act)
        out = self.in2(out, post_slope=LRELU)
        return self.conv_r2(out, res=left, res_ups=True, pre_slope=LRELU, x_act=act)


class AdaResBlock2(nn.Module):
    """models/model_blocks.py:776-814.  ``style`` is [N, style_dim]."""

    def __init__(self, channels, style_dim):
        super().__init__()
        self.style_dim, self.channels = style_dim, channels
        self.lin1_mean = GimLinear(style_dim, channels)
        self.lin1_std = GimLinear(style_dim, channels)
        self.lin2_mean = GimLinear(style_dim, channels)
        self.lin2_std = GimLinear(style_dim, channels)
        self.conv1 = SNConv2d(channels, channels, 3, padding=1)
        self.conv2 = SNConv2d(channels, channels, 3, padding=1)

    def style_vectors(self, style):
        """(mean1, std1, mean2, std2) of the two AdaINs: they depend on `style` only, so the image-to-image module
        computes them for all its blocks up front on a side stream (Img2Img modules in gim_img_models.py)."""
        return (self.lin1_mean(style), self.lin1_std(style), self.lin2_mean(style), self.lin2_std(style))

    def forward(self, x, style, sv=None):
        m1, s1, m2, s2 = sv if sv is not None else self.style_vectors(style)
        out = self.conv1(x)
        out = ops.ada_in(out, m1, s1, post_slope=LRELU)   # stored activated for conv2, its only reader
        out = self.conv2(out, pre_slope=LRELU, x_act=ops.act_storage())
        return ops.ada_in(out, m2, s2, res=x)


class AdaResBlockUp2(nn.Module):
    """models/model_blocks.py:817-865."""

    def __init__(self, in_channels, out_channels, style_dim, out_size=None, scale=2, conv_size=3, padding_size=1):
        super().__init__()
        assert out_size is None and scale == 2
        self.in_channels, self.out_channels, self.style_dim = in_channels, out_channels, style_dim
        self.lin1_mean = GimLinear(style_dim, in_channels)
        self.lin1_std = GimLinear(style_dim, in_channels)
        self.lin2_mean = GimLinear(style_dim, out_channels)
        self.lin2_std = GimLinear(style_dim, out_channels)
        self.conv_l1 = SNConv2d(in_channels, out_channels, 1)
        self.conv_r1 = SNConv2d(in_channels, out_channels, conv_size, padding=padding_size)
        self.conv_r2 = SNConv2d(out_channels, out_channels, conv_size, padding=padding_size)

    def style_vectors(self, style):
        return (self.lin1_mean(style), self.lin1_std(style), self.lin2_mean(style), self.lin2_std(style))

    def forward(self, x, style, sv=None):
        m1, s1, m2, s2 = sv if sv is not None else self.style_vectors(style)
        left = self.conv_l1(x)
        act = ops.act_storage()
        out = ops.ada_in(x, m1, s1, post_slope=LRELU)
        out = self.conv_r1(out, ups=1, pre_slope=LRELU, x_act=act)
        out = ops.ada_in(out, m2, s2, post_slope=LRELU)
        return self.conv_r2(out, res=left, res_ups=True, pre_slope=LRELU, x_act=act)
