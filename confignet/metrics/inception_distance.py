from confignet_amd.metrics.inception_distance import *                                    # noqa: F401,F403
from confignet_amd.metrics.inception_distance import InceptionFeatureExtractor, compute_FID, compute_KID  # noqa: F401
