"""mesm_amd — MI355X-native (gfx950) implementation of the MESM training hot path.

Public surface = the reference's factory functions (runner.py): build_model, build_criterion,
build_optimizer.  Compute runs in libmesm_gfx950.so (include/mesm_gfx950.h); there is no CPU path.
"""
from .runner import build_criterion, build_model, build_optimizer  # noqa: F401
