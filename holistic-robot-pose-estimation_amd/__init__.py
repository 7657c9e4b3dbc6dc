"""hrpe_amd: MI355X-native implementation of the HoRoPose image->pose hot path.

HIP kernels + C ABI: ``csrc/`` -> ``libhrp_hip.so`` (declared in ``include/hrp.h``).
Host side: ``plan.py`` (static op plans), ``runtime.py`` (nn.Module / autograd boundary) and ``lib/``
which mirrors the reference's ``lib/models`` + ``lib/utils`` module names and call signatures.
"""
from . import _native  # noqa: F401

__all__ = ["_native"]
