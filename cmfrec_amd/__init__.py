"""cmfrec_amd -- MI355X-native ALS / CG collective-matrix-factorization solver behind the cmfrec
interface (fit_collective_*_als C ABI + CMF / CMF_implicit ``fit``)."""
from .models import CMF, CMF_implicit  # noqa: F401
from .session import AlsSession  # noqa: F401
from . import ops  # noqa: F401

__all__ = ["CMF", "CMF_implicit", "AlsSession", "ops"]
