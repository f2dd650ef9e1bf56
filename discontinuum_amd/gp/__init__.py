"""gpytorch-shaped building blocks (kernels, constraints, priors, means, likelihoods) for the HIP engine."""
from . import constraints, kernels, likelihoods, means, priors  # noqa: F401
from .models import ExactGP  # noqa: F401
from .mll import ExactMarginalLogLikelihood, NotPSDError  # noqa: F401
