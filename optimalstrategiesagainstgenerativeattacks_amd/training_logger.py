"""Host-side mirror of the reference's ``training/logger.py`` ``Logger`` (:12-86) for the training loop: the same method
names and arguments; scalars are kept in ``stats[category][k] = [(global_step, v), ...]`` and pickled by ``save_stats``;
image grids are written as PNG when PIL is importable (else ``.npy``).  tensorboardX / torchvision are optional in the
reference's environment and absent here; nothing on the hot path depends on them."""
import os
import pickle

import numpy as np
import torch


class Logger(object):
    def __init__(self, log_dir='./logs', img_dir='./imgs', tensorboard_dir=None):
        self.stats = dict()
        self.log_dir = log_dir
        self.img_dir = img_dir
        os.makedirs(log_dir, exist_ok=True)
        os.makedirs(img_dir, exist_ok=True)
        self.tb = None
        if tensorboard_dir is not None:
            self.setup_monitoring(tensorboard_dir)

    def setup_monitoring(self, tensorboard_dir):
        try:
            import tensorboardX
            self.tb = tensorboardX.SummaryWriter(tensorboard_dir)
        except ImportError:
            self.tb = None

    def add_scalar(self, category, k, v, global_step):
        self.stats.setdefault(category, {}).setdefault(k, []).append((global_step, v))
        if self.tb is not None:
            self.tb.add_scalar('%s/%s' % (category, k), v, global_step)

    def add_imgs(self, imgs, category, k, global_step, nrow=5):
        """imgs: [t, C, H, W] in [0, 1] (CPU); one grid image per call."""
        outdir = os.path.join(self.img_dir, category.replace(' ', '_'), k)
        os.makedirs(outdir, exist_ok=True)
        x = torch.as_tensor(imgs).detach().float().cpu().numpy()
        t, C, H, W = x.shape
        rows = -(-t // nrow)
        grid = np.zeros((C, rows * H, nrow * W), dtype=np.float32)
        for i in range(t):
            r, c = divmod(i, nrow)
            grid[:, r * H:(r + 1) * H, c * W:(c + 1) * W] = x[i]
        path = os.path.join(outdir, '%08d' % global_step)
        try:
            from PIL import Image
            arr = (np.clip(grid, 0, 1) * 255 + 0.5).astype(np.uint8).transpose(1, 2, 0)
            Image.fromarray(arr[:, :, 0] if C == 1 else arr).save(path + '.png')
        except ImportError:
            np.save(path + '.npy', grid)

    def get_last_scalar(self, category, k, default=0.):
        if category not in self.stats or k not in self.stats[category]:
            return default
        return self.stats[category][k][-1][1]

    def save_stats(self, filename):
        with open(os.path.join(self.log_dir, filename), 'wb') as f:
            pickle.dump(self.stats, f)

    def load_stats(self, filename):
        path = os.path.join(self.log_dir, filename)
        if not os.path.exists(path):
            print('Warning: file "%s" does not exist!' % path)
            return
        with open(path, 'rb') as f:
            self.stats = pickle.load(f)
