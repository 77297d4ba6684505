"""Where inside the overlapped training step does the DEVICE spend its time?  Timing events at the phase boundaries of gim_step
(ops.phase_timeline: generator forward / backward / update on lane 0, discriminator forward / backward / update on lane 1), mean over
the steps, relative to the step's own start.  Eight events per step: the step runs as in bench.py.
    python tools/step_timeline.py [steps] [--batch B] [--reg-param R]"""
import argparse
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("steps", type=int, nargs="?", default=20)
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--reg-param", type=float, default=0.0)
args = ap.parse_args()
dev = torch.device("cuda:0")
u = bench.UNIT["vox64"]
m, n, k = 1, 5, 10
G, tr = bench.build_trainer(u["S"], u["C"], n, m, k, dev, reg_param=args.reg_param)
from optimalstrategiesagainstgenerativeattacks_amd import ops  # noqa: E402
trainer = G.DataParallelMock(tr)
leaked, real, si = bench.synthetic_batch(args.batch, m, n, k, u["C"], u["S"], dev, 1234)


def step():
    tr.do_global_step()
    tr.update_learning_rate()
    return G.gim_step(trainer, leaked, real, si, defer_join=True)


for _ in range(5):
    step()
ops.join_lanes()
torch.cuda.synchronize()
with ops.phase_timeline() as marks:
    for _ in range(args.steps):
        step()
    ops.join_lanes()
torch.cuda.synchronize()
if not marks:
    sys.exit("no phase marks: the sequential protocol (GIM_NO_STEP_OVERLAP=1) has no lanes to time")
names = [nm for nm, _ in marks[:8]]
assert names[0] == "step start" and len(marks) == 8 * args.steps, (names, len(marks))
t0 = marks[0][1]
T = [[t0.elapsed_time(ev) for _, ev in marks[8 * i:8 * i + 8]] for i in range(args.steps)]
period = [T[i + 1][0] - T[i][0] for i in range(args.steps - 1)]
print("%d steps of %d episodes, reg_param %g: step period %.2f ms (start of one generator forward to the next, device time)"
      % (args.steps, args.batch, args.reg_param, sum(period) / len(period)))
print("phase boundaries, ms after the step's own start on lane 0 (mean over steps 2..; lane 1 marks are on the discriminator's stream):")
for j, nm in enumerate(names):
    v = [T[i][j] - T[i][0] for i in range(1, args.steps)]
    print("   %-18s %7.2f   (min %.2f, max %.2f)" % (nm, sum(v) / len(v), min(v), max(v)))
# the previous step's discriminator tail under this step's generator forward
tail = [T[i][7] - T[i + 1][0] for i in range(args.steps - 1)]
print("the discriminator update of step i ends %.2f ms AFTER step i + 1's generator forward began (mean; the deferred join)" % (sum(tail) / len(tail)))
