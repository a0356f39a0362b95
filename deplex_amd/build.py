"""Builds libdeplex_amd.so (HIP kernels + C-ABI) in-tree with hipcc for gfx950."""
from __future__ import annotations

import os
import shutil
import subprocess
from pathlib import Path

PKG = Path(__file__).resolve().parent
SRC = PKG / "csrc"
OUT = PKG / "lib" / "libdeplex_amd.so"
# -ffp-contract=off: the reference is built without FMA and every f32/f64 product must round
# before the following add for results to be bit-identical.
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17",
         "-Wno-unused-value", "-Wno-pass-failed"]
LIBS = ["-lz"]  # the PNG reader of the DepthImage path constructor (dpx_png.h) inflates with zlib


def sources():
    """Everything the library is compiled from: the one translation unit, its headers and the kernel-body fragments."""
    return [SRC / "dpx_api.hip", *sorted(SRC.glob("*.h")), *sorted(SRC.glob("*.inc")), PKG.parent / "include" / "deplex_amd.h"]


def needs_build() -> bool:
    if not OUT.exists():
        return True
    t = OUT.stat().st_mtime
    return any(s.stat().st_mtime > t for s in sources())


def build(force: bool = False, verbose: bool = False) -> Path:
    if not force and not needs_build():
        return OUT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    OUT.parent.mkdir(parents=True, exist_ok=True)
    tmp = OUT.with_suffix(".so.tmp")
    cmd = [hipcc, *FLAGS, str(SRC / "dpx_api.hip"), "-o", str(tmp), *LIBS]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    os.replace(tmp, OUT)
    return OUT


def variant_path(name: str) -> Path:
    return OUT.parent / "variants" / f"{name}.so"


def build_variant(name: str, defines=(), force: bool = False) -> Path:
    """An experimental / diagnostic build of the same library under lib/variants/<name>.so (selected with DPX_LIB_PATH):
    "trace" = -DDPX_TRACE, the wave trace of tools/wave_trace.py; "diag" = -DDPX_DIAG: the phase-stamp instantiations of the
    region kernels (dpx_debug_region_phases / dpx_debug_cores_phases), the timing-only plane-fit instantiations
    (DPX_PF_DIAG), DPX_SKIP (stages left out: results are garbage) and DPX_SYNC_EACH — none of which the product library has."""
    if name == "trace" and not defines:
        defines = ("DPX_TRACE",)
    if name == "diag" and not defines:
        defines = ("DPX_DIAG",)
    out = variant_path(name)
    if not force and out.exists() and all(s.stat().st_mtime <= out.stat().st_mtime for s in sources()):
        return out
    out.parent.mkdir(parents=True, exist_ok=True)
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    subprocess.run([hipcc, *FLAGS, *[f"-D{d}" for d in defines], str(SRC / "dpx_api.hip"), "-o", str(out), *LIBS], check=True)
    return out


PYB_SRC = PKG / "cpp" / "pybind" / "deplex_pybind.cpp"


def pybind_path() -> Path:
    import sysconfig
    return PKG / ("_pybind" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))


def build_pybind(force: bool = False, verbose: bool = False) -> Path:
    """pybind11 module with the reference's binding surface (deplex_amd/cpp/pybind), g++ against libdeplex_amd.so."""
    import sysconfig
    import pybind11
    out = pybind_path()
    deps = [PYB_SRC, PKG.parent / "include" / "deplex_amd.h", *(PKG / "cpp" / "include" / "deplex").rglob("*.h")]
    if not force and out.exists() and all(d.stat().st_mtime <= out.stat().st_mtime for d in deps):
        return out
    build()  # the module links against the HIP library
    tmp = out.with_suffix(".tmp")
    cmd = ["g++", "-O2", "-shared", "-fPIC", "-std=c++17", "-fvisibility=hidden",
           f"-I{sysconfig.get_paths()['include']}", f"-I{pybind11.get_include()}", str(PYB_SRC),
           f"-L{OUT.parent}", "-ldeplex_amd", "-Wl,-rpath,$ORIGIN/lib", "-o", str(tmp)]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    os.replace(tmp, out)
    return out


if __name__ == "__main__":
    print(build(force=True, verbose=True))
    print(build_pybind(force=True, verbose=True))
