"""deplex_amd.pybind — the reference's compiled binding surface (`deplex.pybind`, cpp/pybind/deplex_pybind.cpp:19-25):
submodules `plane_extraction` (Config, PlaneExtractor) and `utils` (DepthImage), implemented by the pybind11 module
deplex_amd/cpp/pybind/deplex_pybind.cpp over the HIP library.  `python/deplex/__init__.py:1-2` re-exports
`plane_extraction`; so does this package.
"""
import sys as _sys

from .. import _capi as _capi

_capi._preload_hip_runtime()  # one HIP runtime per process (torch bundles its own copy of libamdhip64)
try:
    from .. import _pybind as _m
except ImportError as e:  # not built yet: build in-tree (g++ + pybind11 headers), then import
    from ..build import build_pybind as _build_pybind
    _build_pybind()
    from .. import _pybind as _m

plane_extraction = _m.plane_extraction
utils = _m.utils
_sys.modules[__name__ + ".plane_extraction"] = plane_extraction
_sys.modules[__name__ + ".utils"] = utils
__all__ = ["plane_extraction", "utils"]
