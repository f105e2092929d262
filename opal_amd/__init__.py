"""opal_amd: MI355X-native implementation of Opal's feature-extraction + OAA-scoring hot path.

Layout:
  csrc/            hand-written HIP kernels (gfx950) + the C-ABI of include/opal_hip.h -> libopal_hip.so
  _lib.py          ctypes binding (no fallback: raises if the library is missing)
  engine.py        per-GPU driver (torch is used for device buffers and streams only)
"""
from ._lib import OpalError, lib  # noqa: F401
from .engine import Engine, GenomeSet, PackedReads, Rng, parse_patterns  # noqa: F401

__version__ = "0.1.0"
