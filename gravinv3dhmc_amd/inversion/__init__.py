"""Potential (GravMagModule) and sampler (HamitonianMC, HMCSample)."""
from .hmc import HamitonianMC, HMCSample, HMCSampleBatch  # noqa: F401
from .potential import GravMagModule  # noqa: F401
from .reginv import BootStrap, ConjugateGradient  # noqa: F401
