"""MI355X-native engine for the GIM image adversarial-training hot path
(conv impersonator / authenticator forward-backward, set-pooling head, min-max BCE losses, two Adam updates).

Host-side mirror of the reference's Python API for that path (``get_au`` / ``get_im``,
``GIMFaceAuthenticator`` / ``GIMFaceImpersonator``, ``GIMImgTrainer`` and the train/eval step functions) over
hand-written HIP kernels for gfx950 (``csrc/`` -> ``libgim_hip.so``, C ABI in ``include/gim_hip.h``).
There is no CPU compute path: forward/backward/step need an MI355X and the built library.
"""
from .gim_img_models import (AdaInImage2Image, Encoder, EnvDecoder, GIMFaceAuthenticator, GIMFaceDis,
                             GIMFaceImpersonator, get_au, get_im)
from .gim_gaussian_trainer import GIMGaussianTrainer
from .gim_gaussian_training import train_gim_gaussian
from .gim_img_trainer import GIMImgTrainer
from .data import EpisodeBank, synthetic_bank
from .gim_img_training import (au_eval_step, au_train_step, eval_step, gim_step, im_eval_step, im_train_step, train_epoch,
                               train_gim_imgs)
from .training_logger import Logger
from .optim import FusedAdam
from .training_utils import CheckpointIO, DataParallelMock, EpisodeParallel, GlobalStep, adjust_batch_size

__all__ = [
    "get_au", "get_im", "Encoder", "EnvDecoder", "AdaInImage2Image", "GIMFaceDis", "GIMFaceAuthenticator",
    "GIMFaceImpersonator", "GIMImgTrainer", "GIMGaussianTrainer", "im_train_step", "au_train_step", "im_eval_step", "au_eval_step",
    "gim_step", "train_epoch", "eval_step", "train_gim_imgs", "train_gim_gaussian", "EpisodeBank", "synthetic_bank", "Logger", "FusedAdam", "DataParallelMock", "EpisodeParallel", "GlobalStep", "CheckpointIO", "adjust_batch_size",
]
