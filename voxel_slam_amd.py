"""Import shim: exposes the package directory ``voxel-slam_amd/`` under the importable name ``voxel_slam_amd``."""
import importlib.util
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
_pkg_dir = os.path.join(_here, "voxel-slam_amd")
_spec = importlib.util.spec_from_file_location(
    "voxel_slam_amd", os.path.join(_pkg_dir, "__init__.py"), submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["voxel_slam_amd"] = _mod
_spec.loader.exec_module(_mod)
