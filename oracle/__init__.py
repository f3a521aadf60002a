"""CPU oracle for the GBP hot path — TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package; the product package ``magics_amd`` never does.
``gbp_oracle.c`` is the restatement of the reference (see its header for the parity
status); this module is a thin ctypes binding exposing the same ``World`` interface as
``magics_amd.World`` so one scenario script can drive either.
"""
from .binding import FLAVOURS, OracleWorld, build, build_flavour, lib, schedule, variable_timesteps  # noqa: F401
