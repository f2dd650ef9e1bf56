"""LOADEST-GP on the MI355X engine (counterpart of ``src/loadest_gp/__init__.py``)."""
from .models import LoadestGPMarginalHIP  # noqa: F401

LoadestGP = LoadestGPMarginalHIP
