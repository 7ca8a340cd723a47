"""Alias of confignet_amd.metrics under the reference's package name (confignet/metrics/__init__.py)."""
from confignet_amd.metrics import *                                                       # noqa: F401,F403
from confignet_amd.metrics import InceptionFeatureExtractor, InceptionMetrics, compute_FID, compute_KID   # noqa: F401
