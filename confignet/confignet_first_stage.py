"""Alias of confignet_amd.confignet_first_stage under the reference's module path."""
from confignet_amd.confignet_first_stage import *   # noqa: F401,F403
from confignet_amd import confignet_first_stage as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
