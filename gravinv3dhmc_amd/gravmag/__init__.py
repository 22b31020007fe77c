"""Forward operators (gz) of prisms and tesseroids, evaluated by the HIP kernels."""
from . import prism, tesseroid  # noqa: F401
