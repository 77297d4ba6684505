"""The GIM image agents on the MI355X engine: host-side mirror of the reference's
``models/gim_img_models.py`` (same classes, constructor arguments, attribute names, ``get_au`` / ``get_im``
factories and state-dict keys), every forward running the HIP kernels of ``libgim_hip.so``.

API tensors keep the reference's layout (samples are ``[B, set, C, S, S]``, NCHW); inside, everything is
NHWC.  ``GIMFaceImpersonator.forward`` takes one extra optional argument ``z`` so that the latent noise
(drawn inside forward by the reference, gim_img_models.py:374) can be injected for parity tests.
"""
import math
import os

import torch
import torch.nn as nn

from . import model_blocks as mb
from . import ops
from .gim_basic_models import GIMMeanStdFcStat


_TWO_STREAMS = os.environ.get("GIM_SINGLE_STREAM") is None  # A/B switch
_PER_SAMPLE_STREAMS = os.environ.get("GIM_PER_SAMPLE_STREAMS") is not None   # one stream per (encoder, sample set): measured slower (310 vs 339 episodes/s)
_LANE1_SIDE = os.environ.get("GIM_NO_LANE1_SIDE") is None   # A/B switch: side streams inside lane 1 as well (+5 %)
_STREAMS = {}


def _use_side_streams(t):
    """Side streams for this call?  Lane 1 forks side streams of its own - a fork of a fork.  Under hipGraph capture that shape
    is avoided: on ROCm 7.2 a captured stream that waits for an event recorded on ANOTHER FORKED stream (anything but the
    capture's origin stream) takes the process down with a segmentation fault - plain torch, nothing of this package involved:
    tools/capture_probe.py cases t_fork2 / t_selfwait / t_alias crash, t_fork (one level: fork from the origin, join into the
    origin) captures and replays, t_unjoined is refused with hipErrorStreamCaptureUnjoined as it should be.  So while capturing,
    lane 1 keeps to its own stream (one level below the origin), and lane 0's side streams fork from / join into the origin."""
    if not (_TWO_STREAMS and t.is_cuda):
        return False
    if ops.current_lane() == 0:
        return True
    return _LANE1_SIDE and not torch.cuda.is_current_stream_capturing()


# Role -> stream map.  Roles: 0, 1 = lane 0's two side streams, 2 = lane 1's main stream, 3, 4 = lane 1's side streams
# (lane 0's main stream is the caller's).  Roles with the same id share ONE torch stream: their work is serialized in issue
# order.  Which roles may run concurrently matters more than how many streams exist (MI355X, 64x64x3, 16 episodes,
# profiles/r01_k_stream_map.txt): every role on a hardware queue of its own (GPU_MAX_HW_QUEUES=8, map 0,1,2,3,4) runs
# 240-255 episodes/s - four conv streams at once starve the critical generator lane - while with HIP's default of 4 hardware
# queues that same map gives 346 or 292 depending on which streams happen to alias onto one queue (creating an RCCL
# communicator shifts the assignment by one).  The default below keeps at most three conv streams in flight (lane 1's
# first encoder on lane 1's own stream, its second one behind lane 0's first side stream): 335-340 episodes/s with 4 or 8
# hardware queues, with or without RCCL.
_STREAM_MAP = [int(v) for v in os.environ.get("GIM_STREAM_MAP", "0,1,2,2,0").split(",")]
assert len(_STREAM_MAP) == 5, "GIM_STREAM_MAP: five comma-separated stream ids (roles 0..4)"
_POOL = {}


def _role_stream(device, role):
    key = (device.type, device.index, _STREAM_MAP[role])
    if key not in _POOL:
        _POOL[key] = torch.cuda.Stream(device=device)
    return _POOL[key]


def _side_streams(device, n=2):
    """The side streams of the current lane (see ops.lane)."""
    assert n == 2 or _PER_SAMPLE_STREAMS
    lane = ops.current_lane()
    if n == 2:
        return [_role_stream(device, 3 * lane), _role_stream(device, 3 * lane + 1)]
    key = (device.type, device.index, "many", lane)
    have = _STREAMS.setdefault(key, [])
    while len(have) < n:
        have.append(torch.cuda.Stream(device=device))
    return have[:n]


def lane_stream(device, lane):
    """The main stream of lane 1 (lane 0 runs on the caller's current stream)."""
    assert lane == 1
    return _role_stream(device, 2)


def stream_concurrency_check(device, usec=300, warn=True):
    """Do the engine's streams (the caller's + the role streams above) run CONCURRENTLY on this process's HIP runtime?
    One `usec`-microsecond single-wave spin kernel (gim_spin) is launched on every distinct stream at the same moment and the
    total is event-timed against one spin alone: streams that HIP dealt onto the same hardware queue run their kernels one after
    the other (GPU_MAX_HW_QUEUES, read by the runtime when it starts: 4 by default - two of the step's streams on one queue cost
    6-12 % of the step, profiles/r03_q_hw_queue_sweep.txt; how streams alias depends on how many exist, a real 8-rank RCCL
    communicator creates more than a one-rank rehearsal).  Returns {"streams", "one_ms", "all_ms", "serialization" (all / one:
    1 = fully concurrent, >= 2 = at least two streams share a queue), "concurrent", "GPU_MAX_HW_QUEUES"} and warns loudly when
    the streams serialize.  Costs < 2 ms; bench.py and train_gim_imgs run it once at start-up."""
    import warnings
    from . import _lib
    lib = _lib.load()
    cur = torch.cuda.current_stream(device)
    streams, seen = [cur], {cur.cuda_stream}
    for role in range(5):
        st = _role_stream(device, role)
        if st.cuda_stream not in seen:
            seen.add(st.cuda_stream)
            streams.append(st)

    def run(sts):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(cur)
        for st in sts[1:]:
            st.wait_stream(cur)
        for st in sts:
            _lib.check(lib.gim_spin(usec, st.cuda_stream), "spin")
        for st in sts[1:]:
            cur.wait_stream(st)
        e1.record(cur)
        e1.synchronize()
        return e0.elapsed_time(e1)
    run(streams)                      # first launch of the kernel / first use of the streams
    one = min(run(streams[:1]) for _ in range(3))
    alls = min(run(streams) for _ in range(3))
    ser = alls / max(one, 1e-6)
    res = {"streams": len(streams), "one_ms": round(one, 4), "all_ms": round(alls, 4), "serialization": round(ser, 2),
           "concurrent": ser < 1.5, "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES")}
    if warn and not res["concurrent"]:
        warnings.warn("GIM engine: its %d HIP streams do NOT run concurrently (spin test: %.2f ms together vs %.2f ms alone): at least two "
                      "share a hardware queue and the training step will be 6-12 %% slower.  Set GPU_MAX_HW_QUEUES=8 in the environment "
                      "BEFORE the process makes its first GPU call (importing this package first does it)." % (len(streams), alls, one),
                      RuntimeWarning, stacklevel=2)
    return res


class Encoder(nn.Module):
    """models/gim_img_models.py:19-57.  forward: NHWC [N, S, S, C] -> [N, style_dim]."""

    def __init__(self, img_size, img_channels, style_dim=512, min_n_channels=64, use_out_lrelu=True):
        super().__init__()
        assert use_out_lrelu, "the hot path fuses the output LeakyReLU into the max-pool kernel"
        self.img_size, self.img_channels, self.style_dim = img_size, img_channels, style_dim
        self.use_out_lrelu = use_out_lrelu
        self.n_down_blocks = int(math.log2(img_size)) - 2
        self.min_n_channels = int(max(min_n_channels, style_dim / (2 ** (self.n_down_blocks - 1))))
        self.channel_sizes = [img_channels] + [min(style_dim, int(self.min_n_channels * (2 ** i)))
                                               for i in range(self.n_down_blocks)]
        self.att_loc = int(math.ceil(self.n_down_blocks / 2))
        self.down_blocks = nn.ModuleList()
        for i in range(self.n_down_blocks):
            self.down_blocks.append(mb.ResBlockDown(self.channel_sizes[i], self.channel_sizes[i + 1]))
        self.att = mb.SelfAttention(self.channel_sizes[self.att_loc])

    def forward(self, x):
        # Tensors between the blocks are stored ACTIVATED wherever their readers are ResBlockDowns (mb.ResBlockDown.forward_act):
        # block i's last conv writes lrelu(y) when block i + 1 reads it directly; the attention block reads a raw tensor and
        # writes an activated one; the last block's output (global max pool) and the image itself stay raw.
        slope = mb.LRELU if ops.act_storage() else 1.0
        act = False
        for i in range(self.n_down_blocks):
            if i == self.att_loc:
                x = self.att(x, post_slope=slope)
                act = slope != 1.0
            want = slope if (i + 1 < self.n_down_blocks and i + 1 != self.att_loc) else 1.0
            x, act = self.down_blocks[i].forward_act(x, x_act=act, post_slope=want)
        return ops.maxpool_lrelu(x)


class EnvDecoder(nn.Module):
    """models/gim_img_models.py:63-95.  forward: [N, style_dim] -> NHWC [N, S, S, C]."""

    def __init__(self, img_size, img_channels, style_dim=512, min_n_channels=64):
        super().__init__()
        self.img_size, self.img_channels, self.style_dim = img_size, img_channels, style_dim
        self.min_n_channels = min_n_channels
        self.n_up_blocks = int(math.log2(img_size))
        self.channel_sizes = list(
            reversed([min(style_dim, int(self.min_n_channels * (2 ** i))) for i in range(self.n_up_blocks)])
        ) + [img_channels]
        self.att_loc = int(math.ceil(self.n_up_blocks / 2))
        self.up_blocks = nn.ModuleList()
        for i in range(self.n_up_blocks):
            self.up_blocks.append(mb.ResBlockUp(self.channel_sizes[i], self.channel_sizes[i + 1]))
        self.att = mb.SelfAttention(self.channel_sizes[self.att_loc])

    def forward(self, x):
        x = x.view(x.size(0), 1, 1, x.size(1))
        for i in range(self.n_up_blocks):
            if i == self.att_loc:
                x = self.att(x)
            x = self.up_blocks[i](x)
        return x


class Img2ImgDownModule(nn.Module):
    """models/gim_img_models.py:101-139."""

    def __init__(self, img_size, img_channels, style_dim=512, min_n_channels=64):
        super().__init__()
        self.img_size, self.img_channels, self.style_dim = img_size, img_channels, style_dim
        self.n_down_blocks = int(math.log2(img_size)) - 2
        self.min_n_channels = int(max(min_n_channels, style_dim / (2 ** (self.n_down_blocks - 1))))
        self.channel_sizes = [img_channels] + [min(style_dim, int(self.min_n_channels * (2 ** i)))
                                               for i in range(self.n_down_blocks)]
        self.att_loc = int(math.ceil(self.n_down_blocks / 2))
        self.down_blocks = nn.ModuleList()
        self.in_layers = nn.ModuleList()
        for i in range(self.n_down_blocks):
            if i == 0:
                self.down_blocks.append(mb.ResBlockDown(self.channel_sizes[i], self.channel_sizes[i + 1], conv_size=9, padding_size=4))
            else:
                self.down_blocks.append(mb.ResBlockDown(self.channel_sizes[i], self.channel_sizes[i + 1]))
            self.in_layers.append(mb.GimInstanceNorm2d(self.channel_sizes[i + 1]))
        self.att = mb.SelfAttention(self.channel_sizes[self.att_loc])

    def forward(self, x):
        # as in Encoder.forward: the InstanceNorm behind block i writes lrelu(.) when block i + 1 reads it directly
        slope = mb.LRELU if ops.act_storage() else 1.0
        act = False
        for i in range(self.n_down_blocks):
            if i == self.att_loc:
                x = self.att(x, post_slope=slope)
                act = slope != 1.0
            x, _ = self.down_blocks[i].forward_act(x, x_act=act)
            want = slope if (i + 1 < self.n_down_blocks and i + 1 != self.att_loc) else 1.0
            x = self.in_layers[i](x, post_slope=want)
            act = want != 1.0
        return x


class Img2ImgAdaInResModule(nn.Module):
    """models/gim_img_models.py:142-162."""

    def __init__(self, style_dim=512, n_blocks=5):
        super().__init__()
        self.style_dim, self.n_blocks = style_dim, n_blocks
        self.res_blocks = nn.ModuleList()
        for _ in range(self.n_blocks):
            self.res_blocks.append(mb.AdaResBlock2(channels=style_dim, style_dim=style_dim))

    def forward(self, x, style, svs=None):
        for i in range(self.n_blocks):
            x = self.res_blocks[i](x=x, style=style, sv=None if svs is None else svs[i])
        return x


class Img2ImgAdaInUpModule(nn.Module):
    """models/gim_img_models.py:165-215."""

    def __init__(self, img_size, img_channels, style_dim=512, min_n_channels=64):
        super().__init__()
        self.img_size, self.img_channels, self.style_dim = img_size, img_channels, style_dim
        self.n_up_blocks = int(math.log2(img_size)) - 2
        self.min_n_channels = int(max(min_n_channels, style_dim / (2 ** (self.n_up_blocks - 1))))
        self.channel_sizes = list(
            reversed([min(style_dim, int(self.min_n_channels * (2 ** i))) for i in range(self.n_up_blocks)])
        ) + [img_channels]
        self.att_loc = int(math.ceil(self.n_up_blocks / 2))
        self.up_blocks = nn.ModuleList()
        for i in range(self.n_up_blocks):
            if i == (self.n_up_blocks - 1):
                self.up_blocks.append(mb.AdaResBlockUp2(in_channels=self.channel_sizes[i], out_channels=self.channel_sizes[i + 1],
                                                        style_dim=style_dim, conv_size=9, padding_size=4))
            else:
                self.up_blocks.append(mb.AdaResBlockUp2(in_channels=self.channel_sizes[i], out_channels=self.channel_sizes[i + 1],
                                                        style_dim=style_dim))
        self.att = mb.SelfAttention(self.channel_sizes[self.att_loc])

    def forward(self, x, style, svs=None):
        for i in range(self.n_up_blocks):
            if i == self.att_loc:
                x = self.att(x)
            x = self.up_blocks[i](x=x, style=style, sv=None if svs is None else svs[i])
        return ops.tanh(x)


class AdaInImage2Image(nn.Module):
    """models/gim_img_models.py:218-257."""

    def __init__(self, img_size, in_channels, out_channels, style_dim, n_adain_res_blocks=5, min_n_channels=64):
        super().__init__()
        self.img_size, self.in_channels, self.out_channels = img_size, in_channels, out_channels
        self.style_dim, self.n_adain_res_blocks, self.min_n_channels = style_dim, n_adain_res_blocks, min_n_channels
        self.down_block = Img2ImgDownModule(img_size=img_size, img_channels=in_channels, style_dim=style_dim,
                                            min_n_channels=min_n_channels)
        self.adain_res_block = Img2ImgAdaInResModule(style_dim=style_dim, n_blocks=n_adain_res_blocks)
        self.adain_up_block = Img2ImgAdaInUpModule(img_size=img_size, img_channels=out_channels, style_dim=style_dim,
                                                   min_n_channels=min_n_channels)
        # the 36 style projections as ONE grouped launch (ops.GroupedLinearFn); a per-module switch (plain attribute, not state)
        self.group_style_linears = True

    def forward(self, x, style):
        # the 36 style linears (tiny, latency-bound GEMMs) depend on `style` only: run them all now on a side stream,
        # under the down path, instead of in front of each AdaIN (autograd overlaps their backward the same way)
        svs_res = svs_up = None
        if _use_side_streams(x):
            cur = torch.cuda.current_stream()
            side = _side_streams(x.device)[0]
            ops.stream_wait(side, cur)
            with torch.cuda.stream(side):
                if not self.group_style_linears or ops.deterministic():
                    # (the grouped launch addresses its outputs and gradients through job tables built per allocation and adds the
                    #  36 input gradients with float atomics: a hipGraph capture - tools/graph_replay_experiment.py - and the
                    #  deterministic mode run the layers one by one)
                    svs_res = [b.style_vectors(style) for b in self.adain_res_block.res_blocks]
                    svs_up = [b.style_vectors(style) for b in self.adain_up_block.up_blocks]
                else:
                    # all 36 style projections (4 per AdaIN block) in ONE grouped launch (ops.GroupedLinearFn)
                    blocks = list(self.adain_res_block.res_blocks) + list(self.adain_up_block.up_blocks)
                    lins = [l_ for b in blocks for l_ in (b.lin1_mean, b.lin1_std, b.lin2_mean, b.lin2_std)]
                    shp = style.shape
                    ys = ops.grouped_linear(style.reshape(-1, shp[-1]), [(l_.weight, l_.bias) for l_ in lins])
                    ys = [y.view(*shp[:-1], y.shape[-1]) for y in ys]
                    svs = [tuple(ys[4 * i:4 * i + 4]) for i in range(len(blocks))]
                    svs_res, svs_up = svs[:len(self.adain_res_block.res_blocks)], svs[len(self.adain_res_block.res_blocks):]
            style.record_stream(side)
        x = self.down_block(x)
        if svs_res is not None:
            ops.stream_wait(cur, side)
            for sv in svs_res + svs_up:
                for t in sv:
                    t.record_stream(cur)
        x = self.adain_res_block(x=x, style=style, svs=svs_res)
        return self.adain_up_block(x=x, style=style, svs=svs_up)


def _set_training(module, mode):
    """nn.Module.train(mode) for a whole model in a third of the time: the same walk over self.modules(), with `training` written
    straight into each module's __dict__ instead of through 220 recursive train() calls and nn.Module.__setattr__'s type checks
    (the reference calls .train() on both networks at the start of every training step, training/gim_img_training.py:160,170:
    0.65 + 0.2 ms of host time per step)."""
    if not isinstance(mode, bool):
        raise ValueError("training mode is expected to be boolean")
    for m in module.modules():
        m.__dict__["training"] = mode
    return module



class GIMFaceDis(nn.Module):
    """models/gim_img_models.py:263-299."""

    def __init__(self, src_dim, env_dim, stat):
        super().__init__()
        self.src_dim, self.env_dim = src_dim, env_dim
        self.stat = stat
        self.n_stats = stat.n_stats
        mlp_input_dim = 2 * (self.n_stats * env_dim + src_dim)
        self.mlp = mb.MLP((mlp_input_dim, env_dim + src_dim, 2 * (env_dim + src_dim), 1))
        # weights_init('kaiming') of the reference (models/model_blocks.py:18-38, applied at :277)
        for m in self.mlp.model:
            if isinstance(m, mb.GimLinear):
                nn.init.kaiming_normal_(m.weight.data, a=0.2)
                nn.init.constant_(m.bias.data, 0.0)

    def forward(self, test_src, test_env, si_src, si_env):
        fc_test = self.stat.fc.per_sample(test_env)
        fc_si = self.stat.fc.per_sample(si_env)
        x = ops.head_cat(test_src, test_env, si_src, si_env, fc_test, fc_si)
        return self.mlp(x)


def _encode_sample(encoder, sample):
    """[B, t, C, S, S] (NCHW) -> [B, t, style_dim]."""
    B, t = sample.size(0), sample.size(1)
    x = ops.to_nhwc(sample.reshape(B * t, *sample.size()[2:]))
    return encoder(x).view(B, t, -1)


class GIMFaceAuthenticator(nn.Module):
    """models/gim_img_models.py:304-340."""

    def __init__(self, src_encoder, env_encoder, dis):
        super().__init__()
        self.src_encoder = src_encoder
        self.env_encoder = env_encoder
        self.dis = dis
        self._sn_plan = None

    def train(self, mode=True):
        return _set_training(self, mode)

    def prefetch_spectral(self, rounds):
        """Run the power iterations of the next `rounds` calls of each encoder up front (mb.SNPlan)."""
        if self._sn_plan is None:
            self._sn_plan = mb.SNPlan(mb.sn_convs(self.src_encoder, self.env_encoder))
        self._sn_plan.run(rounds, self.training)

    def forward(self, test_sample, si_sample):
        self.prefetch_spectral(2)
        (test_src, si_src), (test_env, si_env) = self.encode_samples([test_sample, si_sample])
        return self.dis(test_src=test_src, test_env=test_env, si_src=si_src, si_env=si_env)

    def encode_samples(self, samples):
        """src- and env-encode every sample set.  The two encoders are independent networks: they run on two HIP
        streams so that the small layers of one (8x8 and smaller maps, too few workgroups to fill 256 CUs at 16
        episodes per GPU) overlap with the other's.  The HOST call order per encoder is the reference's (samples in list
        order), which is what fixes the order of the spectral-norm power iterations (SNPlan hands out precomputed
        (sigma, u, v) in call order).  autograd replays each backward op on its forward stream, so the backward overlaps
        the same way.  (One stream per (encoder, sample set) pass - GIM_PER_SAMPLE_STREAMS=1 - is slower.)"""
        cur = torch.cuda.current_stream()
        if not _use_side_streams(samples[0]):
            return ([self.src_encode_sample(s) for s in samples], [self.env_encode_sample(s) for s in samples])
        per_sample = _PER_SAMPLE_STREAMS
        streams = _side_streams(samples[0].device, 2 * len(samples) if per_sample else 2)
        outs = []
        for e, enc in enumerate((self.src_encoder, self.env_encoder)):
            feats = []
            for i, smp in enumerate(samples):
                st = streams[e * len(samples) + i] if per_sample else streams[e]
                if per_sample or i == 0:
                    ops.stream_wait(st, cur)
                with torch.cuda.stream(st):
                    f = _encode_sample(enc, smp)
                f.record_stream(cur)  # produced on the side stream, consumed by the head on the current stream
                smp.record_stream(st)
                feats.append(f)
            outs.append(feats)
        for st in streams:
            ops.stream_wait(cur, st)
        return outs[0], outs[1]

    def src_encode_sample(self, sample):
        return _encode_sample(self.src_encoder, sample)

    def env_encode_sample(self, sample):
        return _encode_sample(self.env_encoder, sample)


class GIMFaceImpersonator(nn.Module):
    """models/gim_img_models.py:346-423."""

    def __init__(self, src_encoder, env_encoder, env_decoder, img2img, env_noise_mapper, use_img_att=False):
        super().__init__()
        self.src_encoder = src_encoder
        self.env_encoder = env_encoder
        self.env_decoder = env_decoder
        self.img2img = img2img
        self.env_noise_mapper = env_noise_mapper
        self.style_dim = src_encoder.style_dim
        assert src_encoder.style_dim == env_encoder.style_dim == env_decoder.style_dim == img2img.style_dim
        self.use_img_att = use_img_att
        self.img_att = mb.ImgAttention(img1_channels=self.src_encoder.img_channels, img2_channels=self.img2img.out_channels)
        self._sn_plan = None

    def train(self, mode=True):
        return _set_training(self, mode)

    def forward(self, leaked_sample, n, remove_noise_mean=True, z=None):
        if self._sn_plan is None:
            mods = [self.src_encoder, self.env_encoder, self.env_decoder, self.img2img] + ([self.img_att] if self.use_img_att else [])
            self._sn_plan = mb.SNPlan(mb.sn_convs(*mods))
        self._sn_plan.run(1, self.training)
        B, m, C, S, _ = leaked_sample.size()
        leaked = ops.to_nhwc(leaked_sample.reshape(B * m, C, S, S))
        # the source code is first needed by img2img: its encoder (B*m images, far too few workgroups to fill the
        # chip) runs on a side stream under the env encoder -> noise mapper -> env decoder chain
        side = None
        if _use_side_streams(leaked):
            cur = torch.cuda.current_stream()
            side = _side_streams(leaked.device)[1]
            ops.stream_wait(side, cur)
            with torch.cuda.stream(side):
                src = ops.mean_dim1(self.src_encoder(leaked).view(B, m, -1))
            leaked.record_stream(side)
        else:
            src = ops.mean_dim1(self.src_encoder(leaked).view(B, m, -1))
        env = ops.mean_dim1(self.env_encoder(leaked).view(B, m, -1))
        if z is None:
            z = torch.randn((B, n, self.style_dim), device=leaked_sample.device)
        w = self.env_noise_mapper(z)
        noisy_env = ops.noise_combine(env, w, remove_noise_mean)
        env_img = self.env_decoder(noisy_env.view(B * n, -1))
        if side is not None:
            ops.stream_wait(cur, side)
            src.record_stream(cur)
        first = leaked.view(B, m, S, S, C)[:, 0]
        x = ops.concat2(env_img, first, n)
        style = ops.repeat_dim1(src, n).view(B * n, self.style_dim)
        out = self.img2img(x=x, style=style)
        if self.use_img_att:  # models/gim_img_models.py:391-396
            x1 = ops.repeat_dim1(first.reshape(B, S * S * C), n).view(B * n, S, S, C)
            out = self.img_att(x1=x1, x2=out)
        return ops.to_nchw(out).view(B, n, C, S, S)

    def src_encode_sample(self, sample):
        return _encode_sample(self.src_encoder, sample)

    def env_encode_sample(self, sample):
        return _encode_sample(self.env_encoder, sample)


def get_im(img_size, img_channels, style_dim, use_img_att=False, num_env_noise_layers=4):
    """models/gim_img_models.py:429-449."""
    src_encoder = Encoder(img_size=img_size, img_channels=img_channels, style_dim=style_dim)
    env_encoder = Encoder(img_size=img_size, img_channels=img_channels, style_dim=style_dim)
    decoder = EnvDecoder(img_size=img_size, img_channels=img_channels, style_dim=style_dim)
    img2img = AdaInImage2Image(img_size=img_size, in_channels=2 * img_channels, out_channels=img_channels, style_dim=style_dim)
    env_noise_mapper = mb.MLP([style_dim for _ in range(num_env_noise_layers + 1)])
    return GIMFaceImpersonator(src_encoder=src_encoder, env_encoder=env_encoder, env_decoder=decoder, img2img=img2img,
                               env_noise_mapper=env_noise_mapper, use_img_att=use_img_att)


def get_au(img_size, img_channels, style_dim):
    """models/gim_img_models.py:452-463."""
    stat = GIMMeanStdFcStat(style_dim=style_dim, fc_n_stats=2, fc_hidden_layers=(style_dim * 2, style_dim * 3, style_dim * 2))
    dis = GIMFaceDis(src_dim=style_dim, env_dim=style_dim, stat=stat)
    src_encoder = Encoder(img_size=img_size, img_channels=img_channels, style_dim=style_dim)
    env_encoder = Encoder(img_size=img_size, img_channels=img_channels, style_dim=style_dim)
    return GIMFaceAuthenticator(src_encoder=src_encoder, env_encoder=env_encoder, dis=dis)
