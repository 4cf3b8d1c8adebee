"""mr_gan_amd -- MI355X-native feature-matching semi-supervised GAN training path of
Healthcare-Robotics/mr-gan (mr_gan.py), behind the reference's own entry points.

    from mr_gan_amd import mr_gan, dataset, MRGAN
"""
from mr_gan_amd.data import (select_labeled, standard_scale, synthetic_blobs, synthetic_mreo,  # noqa: F401
                             tiled_permutation)
from mr_gan_amd.model import MRGAN  # noqa: F401
from mr_gan_amd.mr_gan import dataset, mr_gan  # noqa: F401

__all__ = ["mr_gan", "dataset", "MRGAN", "synthetic_mreo", "synthetic_blobs", "standard_scale", "select_labeled",
           "tiled_permutation"]
