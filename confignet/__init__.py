"""`import confignet` resolves to the MI355X implementation: the names the reference's callers use
(reference confignet/__init__.py:3-14; train_confignet.py:11-13, train_latent_gan.py:7-8, evaluation/confignet_demo.py)
re-exported from confignet_amd, so train_confignet.py / confignet_demo.py-shaped callers run unchanged from this
repository's root.  InceptionFeatureExtractor / compute_FID / compute_KID / InceptionMetrics are provided
(confignet.metrics); ControllabilityMetrics, CelebaAttributeClassifier and FaceImageNormalizer are out of scope (SURVEY.md
section 2.1) and are not."""
from confignet_amd import ConfigNet, ConfigNetFirstStage, LatentGAN, load_confignet   # noqa: F401
from confignet_amd.neural_renderer_dataset import NeuralRendererDataset   # noqa: F401
from confignet_amd.metrics import InceptionFeatureExtractor, InceptionMetrics, compute_FID, compute_KID   # noqa: F401

from . import metrics, azure_ml_utils, confignet_first_stage, confignet_second_stage, confignet_utils, latent_gan, neural_renderer_dataset   # noqa: F401,E402
