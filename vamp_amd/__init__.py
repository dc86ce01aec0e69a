"""vamp_amd -- MI355X-native implementation of VAMP's MCMC hot path.

The per-walker log-posterior (Voigt/Gaussian optical depth -> flux -> chi^2 + priors) and the
affine-invariant stretch move run as hand-written HIP kernels (csrc/vamp_hip.hip) behind the C
ABI of include/vamp_hip.h; Python mirrors the reference's ``vpfits.VPfit`` surface on top.
There is no CPU fallback: without libvamp_hip.so and a GPU the compute entry points raise.
"""
from . import _lib  # noqa: F401
from .hip_backend import (F32, F64, MODE_GAUSS3, MODE_NBZ3, MODE_VOIGT4, HipContext,  # noqa: F401
                          default_split_block, device_count)

__all__ = ["HipContext", "device_count", "default_split_block", "MODE_GAUSS3", "MODE_VOIGT4", "MODE_NBZ3", "F64", "F32"]
