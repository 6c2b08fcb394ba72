"""MI355X-native PSF reconstruction for MUSE WFM-AO (hot path of musevlt/muse-psfr).

Drop-in API (mirrors muse_psfr/psfrec.py of the reference): compute_psf, compute_psf_from_sparta,
create_sparta_table, fit_psf_cube, muse_intrinsic_psf, fit_psf_with_polynom, and the stages
simul_psd_wfm, psf_muse, convolve_final_psf.
Low level: Context (ctypes binding of libmpsfr.so).
"""
from ._lib import Context, ContextPool, MpsfrError, NFIT, FIT_ILL_CONDITIONED  # noqa: F401
from .synthetic import synthetic_rows, grid_pixscale  # noqa: F401
from .psfrec import (MAX_L0, MIN_L0, compute_psf, compute_psf_from_sparta,  # noqa: F401
                     create_sparta_table, direction_perf, fit_psf_cube, fit_psf_with_polynom,
                     host_cutoff_masks, muse_intrinsic_psf, plot_psf, radial_profile,
                     simul_psd_wfm, psf_muse, convolve_final_psf)

__version__ = '0.1.0'
