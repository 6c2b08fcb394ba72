"""MI355X-native PSF reconstruction for MUSE WFM-AO (hot path of musevlt/muse-psfr)."""
from ._lib import Context, MpsfrError, NFIT  # noqa: F401
from .synthetic import synthetic_rows, grid_pixscale  # noqa: F401
