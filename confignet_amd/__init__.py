"""confignet_amd -- MI355X-native implementation of the ConfigNet GAN hot path behind the reference's
own Python surface (reference: confignet/__init__.py:3-14 for the names callers import)."""
from ._lib import LIB_PATH  # noqa: F401  (raises ImportError if the HIP library is not built)
from .confignet_first_stage import ConfigNetFirstStage
from .confignet_second_stage import ConfigNet
from .latent_gan import LatentGAN
from .confignet_utils import load_confignet
from .synthetic_data import SyntheticFaceDataset

__all__ = ["ConfigNetFirstStage", "ConfigNet", "LatentGAN", "load_confignet", "SyntheticFaceDataset"]
