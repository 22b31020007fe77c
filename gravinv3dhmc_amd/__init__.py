"""gravinv3dhmc_amd -- MI355X-native HMC gravity inversion (hot path of GravInv3DHMC).

Python host code keeps the reference's class / mesher API; all numerical work on the hot
path (kernel assembly, sensitivity weighting, G*rho, G^T*r, potential + gradient, leapfrog
trajectories) runs in hand-written HIP kernels behind the C-ABI of include/gravhmc.h
(libgravhmc.so).  There is no CPU implementation in this package.
"""
from . import constants, mesher  # noqa: F401
from .engine import DeviceMatrix, Engine  # noqa: F401
from .gravmag import prism, tesseroid  # noqa: F401
from .inversion import (BootStrap, ConjugateGradient, GravMagModule, HamitonianMC, HMCSample,  # noqa: F401
                        HMCSampleBatch)

__all__ = ["constants", "mesher", "prism", "tesseroid", "Engine", "DeviceMatrix",
           "GravMagModule", "HamitonianMC", "HMCSample", "HMCSampleBatch", "ConjugateGradient", "BootStrap"]
