"""EXPERIMENT: spatial partition of the chip between the two lanes of the training step.  The engine's streams are created with CU masks
(hipExtStreamCreateWithCUMask through ctypes on the HIP runtime torch has loaded): lane 0 (generator step: its main stream and its side
streams) on the first n0 mask bits, lane 1 (discriminator step) on the last n1 bits.  Question: does each lane's kernel, seeing a
"smaller GPU" (more rounds of workgroups per launch, relatively shorter ramp and tail), beat time-sharing the whole chip?
    python tools/cu_partition_experiment.py n0 n1 [steps]        (256 256 = unmasked control on the same kind of streams)"""
import ctypes
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

n0, n1 = int(sys.argv[1]), int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
torch.zeros(1, device=dev)
hip = ctypes.CDLL("libamdhip64.so")
NCU = torch.cuda.get_device_properties(0).multi_processor_count


def masked_stream(lo, hi):
    words = (NCU + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    for b in range(lo, hi):
        mask[b // 32] |= 1 << (b % 32)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), words, mask)
    assert rc == 0, "hipExtStreamCreateWithCUMask -> %d" % rc
    return torch.cuda.ExternalStream(st.value, device=dev)


from optimalstrategiesagainstgenerativeattacks_amd import gim_img_models as M, ops  # noqa: E402
lane_rng = {0: (0, n0), 1: (NCU - n1, NCU)}
pool = {}


def role_stream(device, role):
    # the default topology (roles sharing a stream id share ONE stream: more streams cost 25 %): the stream's partition is that of the
    # first role that asks for it - with the default map 0,1,2,2,0: ids 0, 1 -> lane 0's partition (role 4, lane 1's second side
    # stream, rides on id 0 as it always did), id 2 -> lane 1's
    sid = M._STREAM_MAP[role]
    if sid not in pool:
        pool[sid] = masked_stream(*lane_rng[0 if role < 2 else 1])
    return pool[sid]


for role in range(5):
    role_stream(dev, role)


M._role_stream = role_stream
main0 = masked_stream(*lane_rng[0])
u = bench.UNIT["vox64"]
m, n, k, B = 1, 5, 10, 16
with torch.cuda.stream(main0):
    G, tr = bench.build_trainer(u["S"], u["C"], n, m, k, dev)
    trainer = G.DataParallelMock(tr)
    leaked, real, si = bench.synthetic_batch(B, m, n, k, u["C"], u["S"], dev, 1234)
    torch.cuda.synchronize()

    def step():
        tr.do_global_step()
        tr.update_learning_rate()
        return G.gim_step(trainer, leaked, real, si, defer_join=True)

    for _ in range(5):
        step()
    ops.join_lanes()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(steps):
        step()
    ops.join_lanes()
    torch.cuda.synchronize()
    dt = time.time() - t0
print("lane 0 on mask bits [0, %d), lane 1 on [%d, %d) of %d CUs: %.1f episodes/s, %.2f ms per step" % (n0, NCU - n1, NCU, NCU, B * steps / dt, dt / steps * 1e3))
