import faulthandler, os, sys, time
faulthandler.dump_traceback_later(60, repeat=True)
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import filled_sd, load_keys
import optimalstrategiesagainstgenerativeattacks_amd as G
from optimalstrategiesagainstgenerativeattacks_amd import authentication_eval as ae
def P(*a): print(*a, flush=True)
dev = torch.device("cuda:0")
S, C, D, m, n, k = 16, 1, 32, 1, 3, 4
keys = load_keys("16_1_32")
au, im = G.get_au(S, C, D), G.get_im(S, C, D)
au.load_state_dict(filled_sd(keys["au"], "aeval/au/", torch.float32)); im.load_state_dict(filled_sd(keys["im"], "aeval/im/", torch.float32))
au, im = au.to(dev), im.to(dev)
imgs, offs = G.synthetic_bank(8, 10, S, C, dev, seed=2)
ds = G.EpisodeBank(imgs, offs, m, n, k, example_cnt_per_class=1, mirror=False, seed=5)
P("bank ok")
authenticator = ae.get_gim_authenticator(au)
impersonator = ae.get_gim_impersonator(im, {"remove_noise_mean": True})
b = next(iter(ds.gpu_batches(4, True)))
torch.cuda.synchronize(); P("batch ok")
o = authenticator.act(test_sample=b["real_sample"], si_sample=b["si_sample"]); torch.cuda.synchronize(); P("au act ok", o[0].flatten().tolist())
f = impersonator.act(leaked_sample=b["leaked_sample"], n=n); torch.cuda.synchronize(); P("im act ok")
o = authenticator.act(test_sample=f, si_sample=b["si_sample"]); torch.cuda.synchronize(); P("au act on fake ok")
r = ae.eval_authenticator_and_impersonator(dev, ds, 4, 0, authenticator, impersonator); P("eval ok", r)
