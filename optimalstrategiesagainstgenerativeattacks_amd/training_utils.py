"""Small host-side pieces of the reference's ``training/utils.py`` / ``training/checkpoints.py`` that the
trainer protocol needs (GlobalStep :15-33, DataParallelMock :36-42, adjust_batch_size :167-171,
CheckpointIO training/checkpoints.py:9-44), plus ``EpisodeParallel``: the one-process-per-GPU replacement of
``nn.DataParallel`` (training/gim_img_training.py:406-411)."""
import math
import os

import torch
import torch.distributed as dist


def compute_grad2(out, x_in):
    """Per-episode squared norm of d(sum out)/d(x) summed over the tensors of x_in - the R1 term of
    training/utils.py:115-124.  The returned [B] tensor carries a second-order graph: backward through it reaches
    the authenticator's parameters (ops.input_grad_only documents how)."""
    from . import ops
    batch_size = x_in[0].size(0)
    with ops.input_grad_only(), ops.caller_thread_backward():
        grad_out = torch.autograd.grad(outputs=out.sum(), inputs=x_in, create_graph=True, retain_graph=True, only_inputs=True)
    reg = None
    for g in grad_out:
        r = ops.sqsum_rows(g.reshape(batch_size, -1))
        reg = r if reg is None else reg + r
    return reg


class GlobalStep(object):
    def __init__(self, gs=-1):
        self._gs = gs

    def step(self):
        self._gs += 1

    def get(self):
        return self._gs

    def set(self, gs):
        self._gs = gs

    def state_dict(self):
        return {"global_step": self._gs}

    def load_state_dict(self, d):
        self.set(d["global_step"])


class DataParallelMock:
    """Gives a bare trainer the ``.module`` attribute the training loop addresses."""

    def __init__(self, module):
        self.module = module

    def forward(self, *inputs, **kwargs):
        return self.module.forward(*inputs, **kwargs)


class EpisodeParallel(DataParallelMock):
    """Data parallelism over episodes, MI355X style: one process per GPU, both agents and both Adam states
    replicated, the episode batch sharded across ranks by the caller (``shard``), and ONE RCCL all-reduce of
    each optimizer's flat gradient bucket per step (done inside ``FusedAdam.step``).  Spectral-norm u/v need no
    synchronisation (data independent); the latent noise z is drawn per rank.  Same ``.module`` /
    ``.forward(mode=...)`` surface as ``nn.DataParallel`` has in the reference loop."""

    def __init__(self, module):
        super().__init__(module)
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world_size = dist.get_world_size() if dist.is_initialized() else 1

    def shard(self, *tensors):
        """The local slice (dim 0 = episodes) of globally batched tensors; batch % world_size must be 0
        (the reference asserts the same, training/utils.py:167-171)."""
        out = []
        for t in tensors:
            assert t.size(0) % self.world_size == 0, "episode batch must be divisible by the number of GPUs"
            per = t.size(0) // self.world_size
            out.append(t[self.rank * per:(self.rank + 1) * per])
        return out[0] if len(out) == 1 else tuple(out)

    def broadcast_parameters(self):
        """Make every rank start from rank 0's parameters and buffers."""
        if self.world_size > 1:
            for t in list(self.module.parameters()) + list(self.module.buffers()):
                dist.broadcast(t.data, src=0)


def pin_rank_to_cores(local_rank=None, local_world=None):
    """Give this rank a core set of its own: the ranks of one node are one Python process each, and each spends 25-30 ms of host
    time per 38 ms step enqueueing kernels from two threads (the caller's and autograd's worker, profiles/r03_mid_host_profile.txt)
    - eight of them migrating over one another's cores add jitter that shows as max-over-ranks step time.  The cores this
    process may use (its affinity mask: a container's share) are dealt out in contiguous blocks by LOCAL_RANK; call it BEFORE
    the first GPU call so that the runtime's helper threads inherit the mask.  (nn.DataParallel of the reference runs one
    thread per device inside one process, training/gim_img_training.py:406-411: nothing to pin there.)
    Returns the sorted core list now in force, or None when nothing was changed (one rank, no sched_setaffinity, fewer than
    two cores per rank - a rank needs two threads)."""
    if local_rank is None:
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if local_world is None:
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    if local_world <= 1 or not hasattr(os, "sched_setaffinity"):
        return None
    cores = sorted(os.sched_getaffinity(0))
    per = len(cores) // local_world
    if per < 2:
        return None
    mine = cores[local_rank * per:(local_rank + 1) * per]
    os.sched_setaffinity(0, mine)
    return mine


def get_device(device_type, device_ids, verbose=True):
    """training/utils.py:48-60.  One process drives ONE GPU here: under torch.distributed the device is this rank's
    (LOCAL_RANK), otherwise the smallest id of device_ids.  There is no CPU path: 'cpu' is refused."""
    if device_type != 'cuda' or not torch.cuda.is_available():
        raise RuntimeError("the GIM engine runs on an MI355X only (device_type='cuda'); there is no CPU path")
    if dist.is_initialized():
        idx = int(os.environ.get("LOCAL_RANK", "0"))
    else:
        idx = min(device_ids) if device_ids else 0
    if verbose:
        print('Using device cuda:{}'.format(idx))
    torch.cuda.set_device(idx)
    return torch.device("cuda", idx)


def adjust_batch_size(ds_length, curr_batch_size, n_devices):
    batch_size = min(curr_batch_size, ds_length)
    batch_size = int(n_devices * math.floor(batch_size / n_devices))
    assert batch_size % n_devices == 0 and batch_size > 0
    return batch_size


def num_parameters(parameter_list):
    return float(sum(p.numel() for p in parameter_list))


class CheckpointIO:
    """The reference's checkpoint file (training/checkpoints.py:9-44): ONE ``torch.save`` dict
    ``{"global_step", "last_epoch", <registered name>: state_dict(), ...}`` - a registered GlobalStep overwrites the integer
    ``global_step`` entry with its own state dict, as it does there, so files written by either side load on the other.
    Written through a temporary file and ``os.replace``: a job killed while saving leaves the previous checkpoint intact."""

    def __init__(self, checkpoint_dir, **modules):
        self.checkpoint_dir = checkpoint_dir
        self.module_dict = dict(modules)
        os.makedirs(checkpoint_dir, exist_ok=True)

    def register_modules(self, **modules):
        self.module_dict.update(modules)

    def save(self, global_step, last_epoch, filename):
        payload = {"global_step": global_step, "last_epoch": last_epoch}
        payload.update((name, obj.state_dict()) for name, obj in self.module_dict.items())
        path = os.path.join(self.checkpoint_dir, filename)
        tmp = path + ".tmp.%d" % os.getpid()
        torch.save(payload, tmp)
        os.replace(tmp, path)

    def load(self, filepath):
        """Restore every registered object the file has an entry for; (-1, -1) when the file does not exist."""
        if not os.path.exists(filepath):
            return -1, -1
        print("=> Loading checkpoint...")
        payload = torch.load(filepath, map_location="cpu", weights_only=False)
        for name, obj in self.module_dict.items():
            if name in payload:
                obj.load_state_dict(payload[name])
            else:
                print("Warning: Could not find %s in checkpoint!" % name)
        return payload["global_step"], payload["last_epoch"]
