"""Alias of confignet_amd.neural_renderer_dataset under the reference's module path."""
from confignet_amd.neural_renderer_dataset import *   # noqa: F401,F403
from confignet_amd import neural_renderer_dataset as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
