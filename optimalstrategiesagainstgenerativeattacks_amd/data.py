"""GPU-resident episode pipeline: the sample contract of the reference's ``ImgGIMDataSet``
(data_handling/img_datasets.py:23-110: ``{"real_sample": [n,C,S,S], "leaked_sample": [m,C,S,S], "si_sample": [k,C,S,S],
"class", "class_name"}`` in [-1, 1], m+n+k DISTINCT images of one class, independent random horizontal flips) served from
a uint8 image bank that lives in HBM, so that the engine (hundreds of episodes per second) is not starved by JPEG
decoding and host-side collation.  The bank holds the images already resized to S x S (the reference resizes with PIL at
load time, img_datasets.py:298); conversion to float, the (0,1) -> (-1,1) range map and the flip run in one HIP kernel
(``gim_episode_gather``).  288 GB of HBM hold 24 M images of 64x64x3.

Index sampling stays on the host (a few dozen integers per episode); ``EpisodeBank`` is also a ``torch.utils.data.Dataset``
with the reference's per-index semantics (``index // example_cnt_per_class`` = class), so the reference's DataLoader-based
loop works on it unchanged.
"""
import numpy as np
import torch

from . import _lib
from ._lib import check


class EpisodeSampler:
    """Host side of the pipeline (no tensors): which images make up which episode, for which rank.

    Two independent generators.  The epoch SHUFFLE must be the same permutation on every rank of a data-parallel job (each rank
    slices its share out of every global batch), while the per-episode image DRAWS differ per rank and consume a data-dependent
    number of values (class sizes differ): drawn from one generator they would knock the ranks' shuffles out of step after the
    first epoch, and ranks would then train on overlapping / missing episodes."""

    def __init__(self, class_offsets, m, n, k, example_cnt_per_class=1, mirror=True, seed=0):
        self.offsets = np.asarray(class_offsets, dtype=np.int64)
        self.m, self.n, self.k = m, n, k
        self.t = m + n + k
        sizes = np.diff(self.offsets)
        keep = sizes >= self.t   # "Filtering classes with less then n+m+k images" (img_datasets.py:59-61)
        self.class_ids = np.nonzero(keep)[0]
        self.n_classes = int(keep.sum())
        self.example_cnt_per_class = example_cnt_per_class
        self.mirror = mirror
        self.seed = seed
        self.shuffle_rng = np.random.default_rng([seed, 0x5EED])
        self._draw_rngs = {}

    def __len__(self):
        return self.n_classes * self.example_cnt_per_class

    def _rng(self, rank):
        if rank not in self._draw_rngs:
            self._draw_rngs[rank] = np.random.default_rng([self.seed, 0xD4A3, rank])
        return self._draw_rngs[rank]

    def draw(self, cls_rows, rank=0):
        """[B, t] image indices (distinct within a row) and [B, t] flip flags for the given class rows (rank's own generator)."""
        rng = self._rng(rank)
        B = len(cls_rows)
        idx = np.empty((B, self.t), dtype=np.int32)
        for b, c in enumerate(cls_rows):
            lo, hi = self.offsets[self.class_ids[c]], self.offsets[self.class_ids[c] + 1]
            idx[b] = lo + rng.choice(hi - lo, size=self.t, replace=False)   # random.sample (img_datasets.py:79)
        flip = (rng.random((B, self.t)) < 0.5) if self.mirror else np.zeros((B, self.t), dtype=bool)
        return idx, flip.astype(np.uint8)

    def epoch_rows(self, batch_size, shuffle, drop_last=True, rank=0, world=1):
        """Per global batch of one epoch: the class rows of THIS rank's slice (what DataLoader(shuffle, drop_last) + the
        episode-dim scatter of nn.DataParallel give, training/gim_img_training.py:210,409)."""
        if batch_size % world != 0:
            raise ValueError("the global batch (%d) must divide by the number of ranks (%d) (training/utils.py:167-171)" % (batch_size, world))
        if world > 1 and not drop_last:
            raise ValueError("drop_last=False with more than one rank: the short last batch would leave ranks with unequal or empty "
                             "slices (mismatched collectives); the training loop always drops it (training/gim_img_training.py:210)")
        order = np.arange(len(self))
        if shuffle:
            self.shuffle_rng.shuffle(order)     # identical on every rank: same seed, and nothing else draws from this generator
        n_full = len(order) // batch_size if drop_last else -(-len(order) // batch_size)
        per = batch_size // world
        for i in range(n_full):
            rows = order[i * batch_size:(i + 1) * batch_size] // self.example_cnt_per_class
            yield list(rows[rank * per:(rank + 1) * per])


class EpisodeBank(torch.utils.data.Dataset):
    def __init__(self, images_u8, class_offsets, m, n, k, example_cnt_per_class=1, mirror=True, class_names=None, seed=0):
        """images_u8: uint8 [n_img, S, S, C] (NHWC) on the GPU, images of one class contiguous;
        class_offsets: [n_classes + 1] start offsets into images_u8."""
        if not (images_u8.is_cuda and images_u8.dtype == torch.uint8 and images_u8.dim() == 4 and images_u8.is_contiguous()):
            raise RuntimeError("EpisodeBank: images must be a contiguous CUDA uint8 [n_img, S, S, C] tensor (no CPU path)")
        self.bank = images_u8
        self.sampler = EpisodeSampler(class_offsets, m, n, k, example_cnt_per_class, mirror, seed)
        self.offsets, self.class_ids, self.n_classes = self.sampler.offsets, self.sampler.class_ids, self.sampler.n_classes
        self.m, self.n, self.k, self.t = m, n, k, m + n + k
        self.example_cnt_per_class = example_cnt_per_class
        self.mirror = mirror
        self.class_names = class_names
        self.S, self.C = images_u8.shape[1], images_u8.shape[3]

    def __len__(self):
        return len(self.sampler)

    def _draw(self, cls_rows, rank=0):
        return self.sampler.draw(cls_rows, rank)

    def gather(self, idx, flip):
        """float [len(idx), C, S, S] in [-1, 1] from flat image indices / flip flags (numpy or tensors)."""
        dev = self.bank.device
        idx_t = torch.as_tensor(np.ascontiguousarray(idx).reshape(-1), dtype=torch.int32).to(dev, non_blocking=True)
        flip_t = torch.as_tensor(np.ascontiguousarray(flip).reshape(-1), dtype=torch.uint8).to(dev, non_blocking=True)
        n_out = idx_t.numel()
        out = torch.empty((n_out, self.C, self.S, self.S), device=dev, dtype=torch.float32)
        check(_lib.load().gim_episode_gather(self.bank.data_ptr(), idx_t.data_ptr(), flip_t.data_ptr(), out.data_ptr(), n_out,
                                             self.S, self.S, self.C, torch.cuda.current_stream().cuda_stream), "episode_gather")
        return out

    def batch(self, cls_rows, rank=0):
        """One collated batch (dict of device tensors, the DataLoader's output format) for the given class rows."""
        idx, flip = self.sampler.draw(cls_rows, rank)
        B = len(cls_rows)
        x = self.gather(idx, flip).view(B, self.t, self.C, self.S, self.S)
        m, n = self.m, self.n
        out = {"leaked_sample": x[:, :m], "real_sample": x[:, m:m + n], "si_sample": x[:, m + n:],
               "class": torch.as_tensor(np.asarray(cls_rows, dtype=np.int64))}
        if self.class_names is not None:
            out["class_name"] = [self.class_names[self.class_ids[c]] for c in cls_rows]
        return out

    def __getitem__(self, index):
        """The reference's per-example contract (one episode, no batch dimension)."""
        c = index // self.example_cnt_per_class
        b = self.batch([c])
        ex = {"real_sample": b["real_sample"][0], "leaked_sample": b["leaked_sample"][0], "si_sample": b["si_sample"][0], "class": c}
        ex["class_name"] = self.class_names[self.class_ids[c]] if self.class_names is not None else str(int(self.class_ids[c]))
        return ex

    def gpu_batches(self, batch_size, shuffle, drop_last=True, rank=0, world=1):
        """Iterator over collated device batches: what ``DataLoader(ds, batch_size, shuffle, drop_last)`` yields, without
        the host round trip; with world > 1 each rank takes its own slice of every global batch (episodes are the
        data-parallel unit)."""
        for rows in self.sampler.epoch_rows(batch_size, shuffle, drop_last, rank, world):
            yield self.batch(rows, rank)

    def num_batches(self, batch_size, drop_last=True):
        return len(self) // batch_size if drop_last else -(-len(self) // batch_size)


class OmniglotEpisodeBank(EpisodeBank):
    """The sample contract of the reference's ``OmniglotGIMDataSet`` (data_handling/img_datasets.py:118-215) on the resident bank:
    one-channel ('L' mode) images, NO mirroring (its ``augment_transforms`` is None), at most ``NUM_EXAMPLES_PER_CLASS`` = 20
    images drawn per episode - ``m + n + si`` beyond that raises the reference's ValueError - and ``class_name`` =
    "alphabet/character".  Everything else (index -> class, distinct images of one class, [-1, 1] range) is EpisodeBank's."""
    NUM_EXAMPLES_PER_CLASS = 20

    def __init__(self, images_u8, class_offsets, m, n, si, example_cnt_per_class=1, class_names=None, seed=0):
        if m + n + si > self.NUM_EXAMPLES_PER_CLASS:
            raise ValueError("Max allowed value for m+n+si is {}".format(self.NUM_EXAMPLES_PER_CLASS))
        if images_u8.dim() == 4 and images_u8.shape[3] != 1:
            raise RuntimeError("OmniglotEpisodeBank: one-channel images expected (the reference loads Omniglot in mode 'L')")
        super().__init__(images_u8, class_offsets, m, n, si, example_cnt_per_class, mirror=False, class_names=class_names, seed=seed)


def synthetic_bank(n_classes, imgs_per_class, S, C, device, seed=0):
    """Random uint8 image bank (benchmarks / tests)."""
    g = torch.Generator().manual_seed(seed)
    imgs = torch.randint(0, 256, (n_classes * imgs_per_class, S, S, C), generator=g, dtype=torch.uint8).to(device)
    return imgs, np.arange(n_classes + 1) * imgs_per_class
