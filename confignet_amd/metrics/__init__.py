"""Training-time metrics of the reference (confignet/metrics/): FID / KID on InceptionV3 features, on the HIP path."""
from .inception_distance import InceptionFeatureExtractor, compute_FID, compute_KID      # noqa: F401
from .metrics import InceptionMetrics                                                    # noqa: F401
