"""MI355X-native (gfx950, HIP) implementation of C2-Ray's evolve3D hot path.

Python host interface; the drop-in boundary itself is the C ABI in include/c2ray_hip.h
(bound from Fortran through fortran/*.F90).  Import name: the directory name contains characters
that are not valid in a Python identifier, so load it with

    import importlib.util, sys
    spec = importlib.util.spec_from_file_location("c2ray_helium_amd", "<repo>/c2-ray3dm1d_helium_amd/__init__.py",
                                                  submodule_search_locations=["<repo>/c2-ray3dm1d_helium_amd"])

(__graft_entry__.load_package() and tests/conftest.py do exactly that).
"""
from . import _build, _lib, hostphys, parallel  # noqa: F401
from ._lib import C2RayHipError  # noqa: F401
from .evolve import (Cosmology, Evolve, GridProps, HipEngine, Material, RadiationTables,  # noqa: F401
                     SourceProps)


def build(force=False):
    return _build.build(force=force)
