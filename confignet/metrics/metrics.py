from confignet_amd.metrics.metrics import InceptionMetrics                                # noqa: F401
