"""CPU suite (no GPU): host logic of the product package, the C-ABI library's exports, and the
data-parallel sharding / gradient exchange over gloo with world_size 2."""
import ctypes
import os
import re
import sys
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from tests.helpers import load_keys  # noqa: E402


def test_state_dict_keys_match_reference():
    """Names, shapes and ORDER of every state-dict entry and parameter equal the reference's (captured by
    oracle/make_golden.py), incl. the spectral-norm weight_orig / weight_u / weight_v triple."""
    import optimalstrategiesagainstgenerativeattacks_amd as G
    for cfg in ("16_1_32", "32_1_512"):
        s, c, d = map(int, cfg.split("_"))
        ref = load_keys(cfg)
        au, im = G.get_au(s, c, d), G.get_im(s, c, d)
        assert [[k, list(v.shape)] for k, v in au.state_dict().items()] == ref["au"]
        assert [[k, list(v.shape)] for k, v in im.state_dict().items()] == ref["im"]
        assert [k for k, _ in au.named_parameters()] == ref["au_params"]
        assert [k for k, _ in im.named_parameters()] == ref["im_params"]
        groups = [len(list(getattr(im, g).parameters())) for g in
                  ("src_encoder", "env_encoder", "env_decoder", "img2img", "img_att", "env_noise_mapper")]
        assert groups == ref["im_groups"]


def test_conv_weights_are_stored_channels_last():
    import optimalstrategiesagainstgenerativeattacks_amd as G
    au = G.get_au(16, 1, 32)
    w = au.src_encoder.down_blocks[0].conv_r2.weight_orig
    assert w.shape == (32, 32, 3, 3) and w.permute(0, 2, 3, 1).is_contiguous()
    sd = {k: v.clone() for k, v in au.state_dict().items()}
    au.load_state_dict(sd)  # plain NCHW-contiguous checkpoints load; storage stays channels-last
    assert au.src_encoder.down_blocks[0].conv_r2.weight_orig.permute(0, 2, 3, 1).is_contiguous()


def test_trainer_host_protocol():
    import optimalstrategiesagainstgenerativeattacks_amd as G
    au, im = G.get_au(16, 1, 32), G.get_im(16, 1, 32)
    with tempfile.TemporaryDirectory() as td:
        tr = G.GIMImgTrainer(td, 1, 3, 4, au, im, au_lr=2e-3, im_lr=1e-3, env_noise_mapping_lr=1e-4,
                             lr_milestones=(2,), lr_gamma=0.5, reg_param=0.0)
        assert os.path.isdir(os.path.join(td, "ckpts"))
        assert tr.global_step == -1
        assert len(tr.impersonator_opt.param_groups) == 6 and len(tr.authenticator_opt.param_groups) == 1
        assert tr.impersonator_opt.param_groups[-1]["lr"] == 1e-4
        assert tr.authenticator_opt.param_groups[0]["betas"] == (0.0, 0.99)
        lrs = []
        for _ in range(3):
            tr.do_global_step()
            tr.update_learning_rate()
            lrs.append((tr.au_lr, tr.im_lr, tr.im_noise_mapping_lr, tr.global_step))
        # MultiStepLR built with last_epoch=-1 and stepped before the optimiser: counter = it + 1
        assert lrs == [(2e-3, 1e-3, 1e-4, 0), (1e-3, 5e-4, 5e-5, 1), (1e-3, 5e-4, 5e-5, 2)]
        with pytest.raises(ValueError):
            tr.forward(mode="nope")
        tr.save(epoch=3)
        path = os.path.join(td, "ckpts", "model_%08d.pt" % 2)
        ck = torch.load(path, weights_only=False)
        assert set(ck) == {"global_step", "last_epoch", "authenticator", "impersonator", "authenticator_opt", "impersonator_opt"}
        assert ck["global_step"] == {"global_step": 2} and ck["last_epoch"] == 3
        au2, im2 = G.get_au(16, 1, 32), G.get_im(16, 1, 32)
        tr2 = G.GIMImgTrainer(td, 1, 3, 4, au2, im2, 2e-3, 1e-3, 1e-4, reg_param=0.0)
        tr2.resume_from_ckpt(path)
        assert tr2.global_step == 2
        for (k, a), (_, b) in zip(au.state_dict().items(), au2.state_dict().items()):
            assert torch.equal(a, b), k


def test_no_cpu_compute_path():
    """The product must fail loudly without the GPU: no silent CPU fallback anywhere."""
    import optimalstrategiesagainstgenerativeattacks_amd as G
    from optimalstrategiesagainstgenerativeattacks_amd import ops
    au = G.get_au(16, 1, 32)
    with pytest.raises(RuntimeError, match="no CPU path"):
        au(torch.zeros(1, 2, 1, 16, 16), torch.zeros(1, 2, 1, 16, 16))
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.linear(torch.zeros(2, 4), torch.zeros(3, 4), torch.zeros(3))
    opt = G.FusedAdam(au.parameters(), lr=1e-3)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        opt.step()


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "optimalstrategiesagainstgenerativeattacks_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f
                assert "gim_oracle" not in src, f


def test_library_exports_every_header_symbol():
    from optimalstrategiesagainstgenerativeattacks_amd import _lib
    header = open(os.path.join(ROOT, "include", "gim_hip.h")).read()
    declared = set(re.findall(r"\b(gim_[a-z0-9_]+)\s*\(", header))
    declared -= {"gim_conv_shape"}
    assert len(declared) >= 30
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), "libgim_hip.so does not export %s" % name
    assert declared == set(_lib.SIGNATURES) | {"gim_last_error"}, declared ^ (set(_lib.SIGNATURES) | {"gim_last_error"})
    assert lib.gim_version() >= 1


# ---------------------------------------------------------------------------------------------------
# data parallelism over episodes: gloo, world_size 2
# ---------------------------------------------------------------------------------------------------
def _dp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from oracle import gim_oracle as go
    from optimalstrategiesagainstgenerativeattacks_amd.optim import all_reduce_grads_
    from optimalstrategiesagainstgenerativeattacks_amd.training_utils import EpisodeParallel
    from tests.helpers import episode, filled_sd
    keys = load_keys("16_1_32")
    au = filled_sd(keys["au"], "dp/au/")
    go.set_requires_grad(au)
    B, n, k = 4, 2, 3
    _, real, si, _ = episode("dp", B, 1, n, k, 1, 16, 32)
    fake = real.flip(0).contiguous()
    ep = EpisodeParallel(module=None)
    assert (ep.rank, ep.world_size) == (rank, world)
    real_l, fake_l, si_l = ep.shard(real, fake, si)
    assert real_l.shape[0] == B // world
    loss = go.authenticator_forward(au, fake_l, real_l, si_l, True, 0.0)[0].mean()  # local mean, as au_train_step does
    loss.backward()
    names = [kk for kk in au if go.is_param(kk)]
    flat = torch.cat([au[kk].grad.reshape(-1) for kk in names])
    scale = all_reduce_grads_(flat)
    flat = flat * scale
    if rank == 0:
        q.put((flat, [au[kk].detach().clone() for kk in ("src_encoder.down_blocks.0.conv_r1.weight_u",)]))
    dist.barrier()
    dist.destroy_process_group()


def test_episode_data_parallel_equals_big_batch_gloo():
    """Sharding the episode batch over 2 ranks + one all-reduce(sum) of the flat gradient bucket (scaled by
    1/world) equals the single-process gradient of the whole batch; spectral-norm buffers need no sync."""
    from oracle import gim_oracle as go
    from tests.helpers import episode, filled_sd, relerr
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    flat_dp, bufs = q.get(timeout=300)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    keys = load_keys("16_1_32")
    au = filled_sd(keys["au"], "dp/au/")
    go.set_requires_grad(au)
    _, real, si, _ = episode("dp", 4, 1, 2, 3, 1, 16, 32)
    fake = real.flip(0).contiguous()
    go.authenticator_forward(au, fake, real, si, True, 0.0)[0].mean().backward()
    flat = torch.cat([au[kk].grad.reshape(-1) for kk in au if go.is_param(kk)])
    assert relerr(flat_dp, flat) < 1e-10
    assert relerr(bufs[0], au["src_encoder.down_blocks.0.conv_r1.weight_u"]) < 1e-12


def test_logger_mirror_and_training_loop_cadence(tmp_path):
    """The caller loop of training/gim_img_training.py:186-354 on a stub trainer (no GPU): n_au_steps gating of the generator
    step, log / save / eval cadences keyed on global_step, one device->host fetch per log point, Logger round trip."""
    import optimalstrategiesagainstgenerativeattacks_amd as G
    from optimalstrategiesagainstgenerativeattacks_amd import gim_img_training as gt
    calls = []

    class StubTrainer:
        def __init__(self):
            self.global_step = -1
            self.au_lr, self.im_lr, self.im_noise_mapping_lr = 1e-4, 1e-4, 1e-6
            self.saved = []

        def do_global_step(self):
            self.global_step += 1

        def update_learning_rate(self):
            pass

        def get_global_step(self):
            return self.global_step

        def save(self, epoch):
            self.saved.append((epoch, self.global_step))

        class authenticator:   # encoding statistics are logged at global_step 0 (0 % tb_log_enc_every == 0)
            src_encode_sample = staticmethod(lambda smp: torch.zeros(2, 3, 4))
            env_encode_sample = staticmethod(lambda smp: torch.zeros(2, 3, 4))

    class Wrap:
        def __init__(self, m):
            self.module = m

    one = torch.ones(())
    au_ret = (one, one, one, one * 0, one, -one, torch.ones(2, 1, dtype=torch.bool), torch.zeros(2, 1, dtype=torch.bool), torch.zeros(2))
    orig = (gt.gim_step, gt.im_eval_step, gt.au_train_step, gt.au_eval_step)
    orig_std = gt.mb.custom_std
    gt.mb.custom_std = lambda x: x.std(1)
    gt.gim_step = lambda tr, l, r, s, z=None, overlap=None, defer_join=False: (calls.append(("G+D", tr.module.global_step)), ((one, torch.zeros(2), one), au_ret))[1]
    gt.im_eval_step = lambda trainer, leaked_sample, si_sample, z=None: (calls.append(("Geval", trainer.module.global_step)), (one, torch.zeros(2), one))[1]
    gt.au_train_step = lambda trainer, real_sample, fake_sample, si_sample: (calls.append(("D", trainer.module.global_step)), au_ret)[1]
    gt.au_eval_step = lambda trainer, real_sample, fake_sample, si_sample: (calls.append(("Deval", trainer.module.global_step)), au_ret)[1]
    try:
        class DS(torch.utils.data.Dataset):
            def __len__(self):
                return 12

            def __getitem__(self, i):
                return {"real_sample": torch.zeros(3, 1, 4, 4), "leaked_sample": torch.zeros(1, 1, 4, 4), "si_sample": torch.zeros(2, 1, 4, 4), "class": i}
        tr = Wrap(StubTrainer())
        logger = G.Logger(log_dir=str(tmp_path / "logs"), img_dir=str(tmp_path / "imgs"))
        gt.train_epoch(device=torch.device("cpu"), logger=logger, epoch=0, trainer=tr, train_ds=DS(), val_ds=DS(), train_batch_size=2,
                       val_batch_size=4, num_workers=0, save_every=4, eval_every=5, save_imgs_every=10 ** 9, train_eval_indices=[],
                       val_eval_indices=[], tb_log_every=2, tb_log_enc_every=10 ** 9, n_au_steps=2)
    finally:
        gt.gim_step, gt.im_eval_step, gt.au_train_step, gt.au_eval_step = orig
        gt.mb.custom_std = orig_std
    assert [c[1] for c in calls if c[0] == "G+D"] == [1, 3, 5]          # (global_step + 1) % n_au_steps == 0 trains G
    assert [c[1] for c in calls if c[0] == "D"] == [0, 2, 4]            # the D step runs every iteration (inside G+D otherwise)
    assert tr.module.saved == [(0, 0), (0, 4)]
    assert sum(1 for c in calls if c[0] == "Deval") == 3 * 2            # eval at steps 0 and 5: 12 // 4 = 3 batches each
    assert [s_ for s_, _ in logger.stats["train_losses"]["dis_loss"]] == [0, 2, 4]
    assert logger.get_last_scalar("eval accuracy", "dis acc") == 1.0 and logger.get_last_scalar("nope", "x", default=7.0) == 7.0
    logger.save_stats("stats.p")
    l2 = G.Logger(log_dir=str(tmp_path / "logs"), img_dir=str(tmp_path / "imgs"))
    l2.load_stats("stats.p")
    assert l2.stats == logger.stats
    logger.add_imgs(torch.rand(7, 3, 4, 4), "cat a", "k", 3)
    assert len(os.listdir(str(tmp_path / "imgs" / "cat_a" / "k"))) == 1


def test_roc_auc_matches_sklearn():
    """authentication_eval.roc_auc (rank statistic with mid-ranks) == sklearn.metrics.roc_auc_score, ties included."""
    import numpy as np
    from sklearn.metrics import roc_auc_score
    from optimalstrategiesagainstgenerativeattacks_amd.authentication_eval import comp_acc, roc_auc
    rng = np.random.default_rng(0)
    for trial in range(5):
        y = rng.integers(0, 2, 200)
        s = np.round(rng.normal(size=200) + 0.7 * y, 1 if trial % 2 else 6)   # coarse rounding -> ties
        assert abs(roc_auc(y, s) - roc_auc_score(y, s)) < 1e-12
    acc, acc_f, acc_r = comp_acc(torch.tensor([1, 1, 0, 1]), torch.tensor([0, 1, 0, 0]))
    assert float(acc_r) == 0.75 and float(acc_f) == 0.75 and float(acc) == 0.75


# ---------------------------------------------------------------------------------------------------
# bench.py --gpus N launches itself; the library keeps no process-wide state
# ---------------------------------------------------------------------------------------------------
def test_bench_self_launch_dry_run_gloo_world2():
    """`python bench.py --gpus 2` with no launcher around it (WORLD_SIZE unset) starts two ranks of itself, which rendezvous
    (gloo here, RCCL on the GPU node), all-reduce once, and rank 0's ONE JSON line comes out of the parent; the parent's return
    code is the worst rank's.  --dry-run stops before anything touches a GPU."""
    import json
    import subprocess
    env = {k_: v for k_, v in os.environ.items() if k_ not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["GIM_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["dry_run"] and line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["backend"] == "gloo"
    assert line["steps"] == 3 and line["warmup"] == 1 and line["config"]["global_batch"] == 32
    # a launcher's environment is used as given: WORLD_SIZE that disagrees with --gpus is refused, and the failure propagates
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"],
                         env=dict(env, WORLD_SIZE="1", RANK="0"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert bad.returncode != 0 and "WORLD_SIZE=1 but --gpus 2" in bad.stderr


def test_conv_shape_struct_matches_the_header():
    """The ctypes mirror of gim_conv_shape has the header's fields in the header's order (matrix path and launch overrides are
    per-call fields: the library has no precision / tuning globals)."""
    from optimalstrategiesagainstgenerativeattacks_amd import _lib
    header = open(os.path.join(ROOT, "include", "gim_hip.h")).read()
    body = re.search(r"typedef struct \{([^}]*)\} gim_conv_shape;", header).group(1)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if decl:
            fields += [f.strip() for f in decl.split(None, 1)[1].split(",")]
    assert fields == [f for f, _ in _lib.GimConvShape._fields_], fields
    assert ctypes.sizeof(_lib.GimConvShape) == 4 * len(fields)
    assert "gim_conv_precision" not in header and "gim_conv_tune_override" not in header


def test_library_sources_read_no_environment_and_keep_no_mutable_globals():
    """include/gim_hip.h promises: no environment variables, no mutable process-wide state.  Checked on the sources: no getenv,
    and every namespace-scope `static` object is const / constexpr or thread_local (the error string, the plan-recording hook)."""
    csrc = os.path.join(ROOT, "optimalstrategiesagainstgenerativeattacks_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if not f.endswith((".hip", ".h")):
            continue
        src = open(os.path.join(csrc, f)).read()
        assert "getenv" not in src, f
        assert "GIM_DBG" not in src, f
        for mt in re.finditer(r"^static\s+(?!inline|const|constexpr|thread_local|__device__|__global__|void|int\s+\w+\(|bool\s+\w+\(|Geo\s+\w+\(|"
                              r"WgPlan\s+\w+\(|size_t\s+\w+\(|float\s+\w+\(|long\s+long\s+\w+\()([^;{(]*)[;=]", src, re.M):
            raise AssertionError("%s: mutable file-scope static: %s" % (f, mt.group(0)))


def test_episode_sampler_ranks_stay_in_step_over_epochs():
    """Data-parallel episode sharding (training/gim_img_training.py:406-411 scatters the batch over devices): every rank's
    process shuffles the SAME epoch order - also in later epochs, although each rank has drawn a different, data-dependent
    number of random values for its own episodes in between (classes of unequal size) - and the rank slices of a global batch
    are disjoint and complete."""
    from optimalstrategiesagainstgenerativeattacks_amd.data import EpisodeSampler
    sizes = [9, 30, 8, 17, 12, 25, 8, 40]
    offs = [0]
    for s_ in sizes:
        offs.append(offs[-1] + s_)
    world, bs = 2, 4
    ranks = [EpisodeSampler(offs, 1, 3, 4, example_cnt_per_class=2, mirror=True, seed=7) for _ in range(world)]   # one per process
    single = EpisodeSampler(offs, 1, 3, 4, example_cnt_per_class=2, mirror=True, seed=7)
    for epoch in range(4):
        per_rank = [list(r.epoch_rows(bs, True, True, rank, world)) for rank, r in enumerate(ranks)]
        whole = list(single.epoch_rows(bs, True, True, 0, 1))
        assert len(per_rank[0]) == len(per_rank[1]) == len(whole) == len(single) // bs
        for i, rows in enumerate(whole):
            assert per_rank[0][i] + per_rank[1][i] == rows, (epoch, i)     # the two slices ARE the global batch, in order
        for rank, r in enumerate(ranks):                                     # each rank now draws its episodes (different counts)
            for rows in per_rank[rank]:
                idx, flip = r.draw(rows, rank)
                assert idx.shape == (len(rows), 8) and all(len(set(row.tolist())) == 8 for row in idx)
        single.draw(whole[0], 0)
    a, _ = ranks[0].draw([1, 1], 0)
    b, _ = ranks[1].draw([1, 1], 1)
    assert not (a == b).all()      # ranks draw different images


def test_deterministic_switch_forces_one_k_slice_and_slab_weight_gradients():
    """ops.set_deterministic (GIM_DETERMINISTIC=1): every convolution shape carries tune_ksplit = 1, and the library then plans
    ONE K slice (no float atomics) for a forward / dgrad launch that splits K by default; the weight-gradient path asks for slabs.
    (Bit-reproducibility itself is a GPU test: tests/test_gpu_models.py::test_deterministic_mode_is_bit_reproducible.)"""
    from optimalstrategiesagainstgenerativeattacks_amd import _lib, ops
    lib = _lib.load()

    def plan(sh, kind):
        out = (ctypes.c_int32 * 8)()
        assert lib.gim_conv_launch_plan(ctypes.byref(sh), kind, ctypes.cast(out, ctypes.c_void_p)) == 0
        return list(out)
    prev = ops.set_deterministic(False)
    try:
        args = (80, 4, 4, 512, 512, 3, 0, 0.2)      # 80 images of 4 x 4 pixels: 1280 output pixels, K = 4608 - splits K by default
        sh = ops._shape(*args)
        assert sh.tune_ksplit == 0 and plan(sh, 0)[3] > 1 and plan(sh, 1)[3] > 1
        assert ops.set_deterministic(True) is False and ops.deterministic()
        sh = ops._shape(*args)
        assert sh.tune_ksplit == 1 and plan(sh, 0)[3] == 1 and plan(sh, 1)[3] == 1
        assert plan(sh, 0)[7] & 255 == 0      # and the tap-major loop (one K order whatever the batch): not the patch-resident kernel
        # ... on position-major rows (a map of <= 16 pixels, >= 32 images): plan[7] >> 8 = the skipped share of its K steps in 1/1000.
        # 80 images on a 4 x 4 map in 64-row tiles: between nothing and the 30.6 % of (pixel, tap) pairs that fall into the padding
        assert 150 <= plan(sh, 0)[7] >> 8 <= 306 and 150 <= plan(sh, 1)[7] >> 8 <= 306, (plan(sh, 0), plan(sh, 1))
        one = ops._shape(64, 2, 2, 512, 512, 3, 0, 0.2)   # 64 images, 2 x 2 map: one tile = one pixel, 4 of its 9 taps are inside the map
        assert plan(one, 0)[7] >> 8 == 556 and plan(one, 1)[7] >> 8 == 556, (plan(one, 0), plan(one, 1))
        few = ops._shape(16, 4, 4, 512, 512, 3, 0, 0.2)   # 16 images: image-major rows, nothing skipped
        big = ops._shape(80, 16, 16, 128, 128, 3, 0, 0.2)
        assert plan(few, 0)[7] >> 8 == 0 and plan(big, 0)[7] >> 8 == 0
        assert lib.gim_conv2d_wgrad_slabs(ctypes.byref(sh)) >= 1
    finally:
        ops.set_deterministic(prev)
    assert ops.deterministic() == prev


def _pin_worker(rank, world, q):
    from optimalstrategiesagainstgenerativeattacks_amd.training_utils import pin_rank_to_cores
    before = sorted(os.sched_getaffinity(0))
    got = pin_rank_to_cores(rank, world)
    q.put((rank, before, got, sorted(os.sched_getaffinity(0))))


@pytest.mark.skipif(not hasattr(os, "sched_setaffinity") or len(os.sched_getaffinity(0)) < 4, reason="needs >= 4 cores to split")
def test_ranks_pin_themselves_to_disjoint_core_sets():
    """pin_rank_to_cores: the cores this job may use are dealt out in contiguous, disjoint blocks by LOCAL_RANK (bench.py and
    train_gim_imgs call it before the first GPU call); one rank alone changes nothing."""
    from optimalstrategiesagainstgenerativeattacks_amd.training_utils import pin_rank_to_cores
    assert pin_rank_to_cores(0, 1) is None
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_pin_worker, args=(r, 2, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, before0, got0, after0), (_, before1, got1, after1) = res
    assert before0 == before1 and got0 == after0 and got1 == after1
    assert not set(got0) & set(got1) and len(got0) == len(got1) == len(before0) // 2
    assert set(got0) | set(got1) <= set(before0)


def test_hw_queue_default_is_reported():
    """Importing the package before the first GPU call sets GPU_MAX_HW_QUEUES=8 unless the user chose a value; hw_queues_state()
    says which value is in force and whether HIP had already been initialised (then the default could not take effect and the
    import warns) - ADVICE r03; bench.py prints it with the result of the start-up spin test of the engine's streams."""
    import subprocess
    code = ("import os, json, warnings\n"
            "os.environ.pop('GPU_MAX_HW_QUEUES', None)\n"
            "import optimalstrategiesagainstgenerativeattacks_amd as G\n"
            "print(json.dumps(G.hw_queues_state()))\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300,
                       env={k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"})
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    st = json.loads(r.stdout.strip().splitlines()[-1])
    assert st == {"GPU_MAX_HW_QUEUES": "8", "preset_by_user": False, "hip_initialised_before_import": False}
    r = subprocess.run([sys.executable, "-c", code.replace("os.environ.pop('GPU_MAX_HW_QUEUES', None)", "os.environ['GPU_MAX_HW_QUEUES'] = '4'")],
                       cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    st = json.loads(r.stdout.strip().splitlines()[-1])
    assert st["GPU_MAX_HW_QUEUES"] == "4" and st["preset_by_user"]


def test_gaussian_loop_scatters_one_global_batch_over_the_ranks(monkeypatch):
    """train() of the Gaussian game under EpisodeParallel: every rank draws the SAME global batch (identical host
    seeds) and trains on ITS slice - nn.DataParallel scatters one batch of batch_size (training/gim_gaussian_training.py:198-201) -
    instead of every rank training on the whole batch.  The step itself is stubbed (it needs the GPU); what is checked is what the
    step is handed, for world_size 2 against world_size 1."""
    from optimalstrategiesagainstgenerativeattacks_amd import gim_gaussian_training as gg
    from optimalstrategiesagainstgenerativeattacks_amd.training_utils import EpisodeParallel
    seen = []

    def fake_step(trainer, leaked, real, si, z=None):
        seen.append((leaked.clone(), real.clone(), si.clone(), None if z is None else z.clone()))
        B = leaked.size(0)
        s0 = torch.zeros(())
        return (s0, real, s0), (s0, s0, s0, s0, s0, s0, torch.ones(B, 1, dtype=torch.bool), torch.zeros(B, 1, dtype=torch.bool), real)
    monkeypatch.setattr(gg, "gim_step", fake_step)

    class Mod:
        m, n, k = 1, 3, 4
        gs = 0      # the first iteration is global step 1: no statistics / checkpoint cadence hit (those run GPU kernels)

        def do_global_step(self):
            self.gs += 1

        def get_global_step(self):
            return self.gs

        def save(self):
            pass

    class Log:
        def add_scalar(self, **kw):
            pass
    runs = {}
    for world, rank in ((1, 0), (2, 0), (2, 1)):
        tr = EpisodeParallel(Mod())
        tr.world_size, tr.rank = world, rank
        torch.manual_seed(11)
        del seen[:]
        gg.train(torch.device("cpu"), tr, Log(), n_iters=2, batch_size=6, src_dim=5, src_sigma=1.0, prior_sigma=2.0,
                 save_stats_every=1000, save_every=1000, host_noise=True)
        runs[(world, rank)] = list(seen)
    for it in range(2):
        whole = runs[(1, 0)][it]
        for j in range(4):
            assert whole[j].shape[0] == 6
            both = torch.cat([runs[(2, 0)][it][j], runs[(2, 1)][it][j]])
            assert torch.equal(both, whole[j]), (it, j)


def test_training_step_runs_autograd_nodes_on_the_calling_thread():
    """ops.caller_thread_backward(): the context the training step's .backward() / autograd.grad() calls run under switches torch's
    autograd engine to the calling thread (every node of the engine is a Python Function: the hand-off to the per-device worker thread
    cost 15-25 % of the host's time per step, profiles/r04_m_host_enqueue.txt) and restores the previous mode on exit."""
    import torch
    from optimalstrategiesagainstgenerativeattacks_amd import ops
    assert ops._CALLER_THREAD_BACKWARD      # the default (GIM_MT_AUTOGRAD unset)
    before = torch.autograd.is_multithreading_enabled()
    with ops.caller_thread_backward():
        assert not torch.autograd.is_multithreading_enabled()
        x = torch.ones(3, requires_grad=True)
        (x * 2).sum().backward()
        assert torch.equal(x.grad, torch.full((3,), 2.0))
    assert torch.autograd.is_multithreading_enabled() == before


def test_model_train_eval_switch_every_submodule():
    """GIMFaceImpersonator / GIMFaceAuthenticator override train() with a direct walk (gim_img_models._set_training: the reference calls
    .train() on both networks every step): same contract as nn.Module.train - every submodule switched, eval() = train(False), a
    non-boolean mode refused, self returned."""
    import optimalstrategiesagainstgenerativeattacks_amd as G
    for net in (G.get_im(16, 1, 32), G.get_au(16, 1, 32)):
        assert net.eval() is net and not any(m.training for m in net.modules())
        assert net.train() is net and all(m.training for m in net.modules())
        net.train(False)
        assert not any(m.training for m in net.modules())
        with pytest.raises(ValueError):
            net.train("yes")
