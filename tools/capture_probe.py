"""Which shape of stream fork / join breaks hipStreamEndCapture?  One case per process (a failure was a segfault).

    python tools/capture_probe.py <case>

Product-independent cases (plain torch ops, nothing of this repo is imported):
    t_fork        C -> A -> C                      one fork, joined
    t_fork2       C -> A -> B -> A -> C            a fork of a fork, every stream joined back into its parent
    t_selfwait    C -> A; A.wait_stream(A); A -> C a stream waits for an event it has just recorded itself (what a role -> stream
                                                   map does when two roles share one stream)
    t_alias       C -> A, C -> B; later B is forked again from A (a stream that belongs to two parents), all joined
    t_unjoined    C -> A, A never joined           must FAIL with hipErrorStreamCaptureUnjoined, not crash
Product cases (the tiny config): op | zero | adam | fwd | fwdg | bwd | full on lane 1's stream forked from the capture stream,
    gimstep       GraphedGimStep(overlap=True): the whole two-lane iteration captured, replayed, compared with eager
"""
import faulthandler
import os
import sys

import torch

faulthandler.enable()


def mark(msg):
    print("  [%s] %s" % (sys.argv[1], msg), flush=True)

case = sys.argv[1]
dev = torch.device("cuda:0")


def torch_only():
    C = torch.cuda.Stream()
    A, B = torch.cuda.Stream(), torch.cuda.Stream()
    x = torch.ones(1 << 16, device=dev)
    out = {}

    def body():
        cur = torch.cuda.current_stream()
        A.wait_stream(cur)
        with torch.cuda.stream(A):
            y = x * 2
            if case == "t_fork2":
                mark("B.wait_stream(A): event recorded on forked A, waited for by B")
                B.wait_stream(A)
                mark("... returned")
                with torch.cuda.stream(B):
                    z = y + 1
                A.wait_stream(B)
                y = z * 3
            elif case == "t_selfwait":
                mark("A.wait_stream(A): event recorded on forked A, waited for by A")
                A.wait_stream(A)
                mark("... returned")
                y = y + 5
                A.wait_stream(A)
            elif case == "t_originwait":        # a forked stream waits twice for events of the ORIGIN: must be fine
                A.wait_stream(cur)
                y = y + 5
            elif case == "t_alias":
                B.wait_stream(cur)          # B forked from the origin ...
                with torch.cuda.stream(B):
                    z0 = x - 1
                B.wait_stream(A)            # ... and again from A
                with torch.cuda.stream(B):
                    z = y + z0
                A.wait_stream(B)
                y = z * 3
        if case != "t_unjoined":
            cur.wait_stream(A)
        out["y"] = y

    with torch.cuda.stream(C):
        for _ in range(2):
            body()
    torch.cuda.synchronize()
    eager = out["y"].clone()
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g, stream=C):
            body()
            mark("body captured; leaving the capture (hipStreamEndCapture)")
    except RuntimeError as e:
        print("case %s: capture raised (no crash): %s" % (case, str(e).splitlines()[0][:150]))
        return
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out["y"], eager)
    print("case %s: ok" % case)


def product():
    import tempfile
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT)
    import optimalstrategiesagainstgenerativeattacks_amd as G
    from optimalstrategiesagainstgenerativeattacks_amd import ops
    from optimalstrategiesagainstgenerativeattacks_amd.gim_img_models import lane_stream
    S, C, D, B, m, n, k = 16, 1, 32, 2, 1, 3, 4

    def make():
        torch.manual_seed(1)
        au, im = G.get_au(S, C, D).to(dev), G.get_im(S, C, D).to(dev)
        with tempfile.TemporaryDirectory() as td:
            tr = G.GIMImgTrainer(td, m, n, k, au, im, 1e-4, 1e-4, 1e-6, reg_param=0.0)
        return tr, G.DataParallelMock(tr)
    tr, trainer = make()
    g0 = torch.Generator(device=dev).manual_seed(3)
    mk = lambda t: torch.rand(B, t, C, S, S, device=dev, generator=g0) * 2 - 1  # noqa: E731
    leaked, real, si, fake = mk(m), mk(n), mk(k), mk(n)
    z = torch.randn(B, n, D, device=dev, generator=g0)
    if case == "gimstep":
        from graph_replay_experiment import GraphedGimStep   # tools/graph_replay_experiment.py (the experiment record)
        tr_e, trainer_e = make()
        gs = GraphedGimStep(trainer, leaked, real, si, z, warmup=3, overlap=True)
        # same number of eager steps on the twin
        for _ in range(3):
            G.gim_step(trainer_e, leaked, real, si, z=z)
        for _ in range(2):
            gi, di = gs(leaked, real, si, z)
            ge, de = G.gim_step(trainer_e, leaked, real, si, z=z)
        torch.cuda.synchronize()
        print("case gimstep: ok  graph vs eager: G loss %.6f / %.6f, D loss %.6f / %.6f" % (float(gi[0]), float(ge[0]), float(di[0]), float(de[0])))
        return
    ds = lane_stream(dev, 1)
    for opt in (tr.impersonator_opt, tr.authenticator_opt):
        opt._ensure()

    def body():
        cur = torch.cuda.current_stream()
        ops.stream_wait(ds, cur)
        with torch.cuda.stream(ds), ops.lane(1):
            if case == "op":
                _ = real * 2
            elif case == "zero":
                tr.authenticator_opt.zero_grad()
            elif case == "adam":
                tr.authenticator_opt.step()
            elif case == "fwd":
                with torch.no_grad():
                    trainer.forward(mode='authenticator_forward', fake_sample=fake, real_sample=real, si_sample=si, grad=False)
            elif case == "fwdg":
                trainer.forward(mode='authenticator_forward', fake_sample=fake, real_sample=real, si_sample=si)
            elif case == "bwd":
                tr.authenticator_opt.zero_grad()
                out = trainer.forward(mode='authenticator_forward', fake_sample=fake, real_sample=real, si_sample=si)
                out[0].mean().backward()
            elif case == "full":
                G.au_train_step(trainer, real, fake, si)
            else:
                raise SystemExit("unknown case " + case)
        ops.stream_wait(cur, ds)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            body()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    for opt in (tr.impersonator_opt, tr.authenticator_opt):
        opt._ensure()
        opt._push_lrs()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    g.replay()
    torch.cuda.synchronize()
    print("case %s: ok" % case)


if case.startswith("t_"):
    torch_only()
else:
    product()
