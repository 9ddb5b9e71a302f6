"""voxel-slam_amd: MI355X-native local-mapping BA hot path of Voxel-SLAM.

The directory name carries a hyphen (as the project layout prescribes); import it as
``voxel_slam_amd`` through the root shim ``voxel_slam_amd.py``.
"""
from . import synth  # noqa: F401

__all__ = ["synth", "capi"]


def __getattr__(name):
    if name == "capi":
        import importlib
        return importlib.import_module(__name__ + ".capi")
    raise AttributeError(name)
