"""Alias of confignet_amd.latent_gan under the reference's module path."""
from confignet_amd.latent_gan import *   # noqa: F401,F403
from confignet_amd import latent_gan as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
