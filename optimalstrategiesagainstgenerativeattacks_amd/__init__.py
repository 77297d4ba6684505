"""MI355X-native engine for the GIM image adversarial-training hot path
(conv impersonator / authenticator forward-backward, set-pooling head, min-max BCE losses, two Adam updates).

Host-side mirror of the reference's Python API for that path (``get_au`` / ``get_im``,
``GIMFaceAuthenticator`` / ``GIMFaceImpersonator``, ``GIMImgTrainer`` and the train/eval step functions) over
hand-written HIP kernels for gfx950 (``csrc/`` -> ``libgim_hip.so``, C ABI in ``include/gim_hip.h``).
There is no CPU compute path: forward/backward/step need an MI355X and the built library.
"""
import os as _os

# The step runs on the caller's stream plus three streams of its own; RCCL adds its streams.  HIP deals streams onto hardware queues
# round-robin (4 by default): when two of the step's streams land on one queue their kernels serialize and the step loses 6-12 %
# (390 instead of 416 episodes/s; which streams collide depends on how many streams exist when they are created - a communicator
# shifts it: profiles/r03_q_hw_queue_sweep.txt).  With 8 queues every stream has its own.  Read by the HIP runtime when it starts:
# effective when this package is imported before the first GPU call; a value set by the user is kept.
# hw_queues_state(): what happened here, for bench.py's JSON line and for anyone debugging a slow step.
_HWQ = {"preset_by_user": "GPU_MAX_HW_QUEUES" in _os.environ, "hip_initialised_before_import": False}
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch as _torch

if _torch.cuda.is_initialized():
    # the HIP runtime has already read (or not found) the variable: the default above can no longer take effect
    _HWQ["hip_initialised_before_import"] = True
    if not _HWQ["preset_by_user"]:
        import warnings as _warnings
        _warnings.warn("optimalstrategiesagainstgenerativeattacks_amd was imported AFTER the process initialised the GPU and "
                       "GPU_MAX_HW_QUEUES was not set: the engine's streams may share HIP hardware queues (6-12 % slower steps). "
                       "Import the package (or export GPU_MAX_HW_QUEUES=8) before the first torch.cuda call; "
                       "gim_img_models.stream_concurrency_check() measures what is in force.", RuntimeWarning, stacklevel=2)


def hw_queues_state():
    """{"GPU_MAX_HW_QUEUES": value in this process's environment, "preset_by_user", "hip_initialised_before_import"}."""
    return dict(_HWQ, GPU_MAX_HW_QUEUES=_os.environ.get("GPU_MAX_HW_QUEUES"))


from .gim_img_models import (AdaInImage2Image, Encoder, EnvDecoder, GIMFaceAuthenticator, GIMFaceDis,
                             GIMFaceImpersonator, get_au, get_im, stream_concurrency_check)
from .gim_gaussian_trainer import GIMGaussianTrainer
from .gim_gaussian_training import train_gim_gaussian
from .gim_img_trainer import GIMImgTrainer
from .data import EpisodeBank, OmniglotEpisodeBank, synthetic_bank
from .gim_img_training import (au_eval_step, au_train_step, eval_step, gim_step, im_eval_step, im_train_step, train_epoch,
                               train_gim_imgs)
from .training_logger import Logger
from .optim import FusedAdam
from .training_utils import CheckpointIO, DataParallelMock, EpisodeParallel, GlobalStep, adjust_batch_size, pin_rank_to_cores

__all__ = [
    "get_au", "get_im", "Encoder", "EnvDecoder", "AdaInImage2Image", "GIMFaceDis", "GIMFaceAuthenticator",
    "GIMFaceImpersonator", "GIMImgTrainer", "GIMGaussianTrainer", "im_train_step", "au_train_step", "im_eval_step", "au_eval_step",
    "gim_step", "train_epoch", "eval_step", "train_gim_imgs", "train_gim_gaussian", "EpisodeBank", "OmniglotEpisodeBank", "synthetic_bank", "Logger", "FusedAdam", "DataParallelMock", "EpisodeParallel", "GlobalStep", "CheckpointIO", "adjust_batch_size",
    "pin_rank_to_cores", "stream_concurrency_check", "hw_queues_state",
]
