"""bazinga.jl_amd — MI355X-native PANOCplus inner solve behind Bazinga.alps' oracle API.

Import as ``import bazinga_jl_amd`` (the loader module at the repo root maps the
importable name onto this directory, whose name is not a valid Python identifier).
"""
from . import _lib, synth
from ._lib import BazingaHipError
from .device import Context, Problem, default_context, runtime_tuning, set_default_context, shard_bounds
from .oracles import (CallbackError, ClosedSet, DenseAffine, DiagQuadratic, FreeSet, IdentityFunction, IndBox, IndFree, IndicatorSet,
                      LeastSquares, NormL0Box, NormLpPowerBox, NormLpPowerNonneg, Quadratic,
                      NormL1, NormL1Box, NormL1Nonneg, Stencil5ptQuadratic, UnsupportedOracle, Zero, ZeroSet,
                      PairwiseSet, VanishingConstraintPairs, ComplementarityPairs, EitherOrPairs, XorPairs)
from .solvers import (LBFGS, NoAcceleration, AndersonAcceleration, Broyden, AugLagFun, AugLagFunSlack, AugLagUpdate, NonsmoothCostFun, NonsmoothCostFunSlack,
                      PANOCplus, alps, als,
                      default_dual_safeguard, default_penalty_parameter, default_subsolver)

__all__ = [n for n in dir() if not n.startswith("_")]
