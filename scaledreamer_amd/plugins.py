"""Importing this module registers every plugin class under the reference's registry names
(the counterpart of `from . import data, models, systems` in threestudio/__init__.py:55)."""
from . import background, geometry, hyper, materials, renderer, sampled_geometry, volsdf_renderer  # noqa: F401

for _opt in ("guidance", "data", "system", "multiprompt", "prompt_processors"):
    try:
        __import__(f"{__name__.rsplit('.', 1)[0]}.{_opt}")
    except ModuleNotFoundError as e:  # module not written yet; anything else must surface
        if e.name != f"{__name__.rsplit('.', 1)[0]}.{_opt}":
            raise
