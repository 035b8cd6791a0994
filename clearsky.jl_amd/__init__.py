"""clearsky.jl_amd -- MI355X-native (gfx950, HIP) line-by-line radiative-transfer core behind ClearSky.jl's
operator surface for ONE hot path: Voigt line sums -> layer optical depth -> Planck/multi-stream fluxes -> band
integrals (reference src/absorption/line_shapes.jl, src/core/discretized.jl, src/core/shared.jl, src/fluxes.jl).

The directory name is not a Python identifier; import it as `clearsky_jl_amd` through the shim at the repo root.
"""
from . import constants
from ._lib import ClearSkyHIPError, build_native, lib, check, dptr, as_f64, SHAPES, LIB_PATH, SIGNATURES, HEADER, HEADER_DEV
from .hitran import MOLPARAM, TMIN, TMAX, ISOINDEX, MolParam, SpectralLines, readpar
from .cia import CIATables, cia, readcia
from .core import (interp_plan, phco2_plan, MultiContext, balanced_ranges, rebalance_ranges, CIA, AcceleratedAbsorber, Sigma, update_, checkpressures, pressurelimits, temperaturelimits, shape_points, AtmosphericDomain, AtmosphericProfile, Column, Gas, reconcentrate, Context, DirectGas, Discretized, HIPDiscretized, FluxPack, GrayGas, SemiGrayGas, UnifiedAbsorber, hipfluxes, hipnetfluxes, opacityerror,
                   PHCO2, PHCO2_, chebygrid, default_context, doppler, doppler_, dtaudP, faddeeva, device_function, fluxes, formprofile,
                   lobattoevaluations, lobattonodes, logrange, lorentz, lorentz_, monochromaticfluxes,
                   monochromaticfluxes_, netfluxes, nodepressures, nodevalues, opticaldepth, ozonelayer, planck,
                   pressuregrid, psatH2O, radiate, radiate_, shape_batch, stefanboltzmann, streamnodes, transmittance,
                   trapz, trapz_weights, unifyabsorbers, voigt, voigt_)

__version__ = "0.1.0"
