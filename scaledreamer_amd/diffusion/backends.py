"""Registers the DiffusionBackend factories (guidance.backend = "hip" | "eager")."""
import importlib.util

from . import eager  # noqa: F401

if importlib.util.find_spec(__package__ + ".engine") is not None:
    from . import engine  # noqa: F401  (hand-written HIP path)
