"""rating-gp on the MI355X engine."""
from .models import RatingGPMarginalHIP  # noqa: F401

RatingGP = RatingGPMarginalHIP
