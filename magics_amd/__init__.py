"""magics_amd — MI355X-native GBP message-passing engine behind the reference's FactorGraph API.

Product package: the HIP kernels + C ABI live in ``csrc/`` (built into ``lib/libmgx.so``),
``hostlib`` binds the ABI, ``world`` mirrors the reference's host interface, ``scenarios``
builds the synthetic graphs of BASELINE.json.  Nothing here imports ``oracle``.
"""
from . import hostlib  # noqa: F401
from .hostlib import MgxError  # noqa: F401
from .world import FactorGraph, World  # noqa: F401
