"""Build libdmet_hip.so (gfx950 only) in-tree with hipcc.  `python -m deepmetv2_amd.build [--force]`."""
from __future__ import annotations

import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
OBJ_DIR = os.path.join(CSRC, "_obj")
LIB_PATH = os.path.join(PKG_DIR, "libdmet_hip.so")
SOURCES = ["knn.hip", "edgeconv.hip", "edgemlp.hip", "misc.hip", "dense.hip", "encoder.hip", "norm.hip", "edgeconv_bwd.hip",
           "head.hip", "finalize.hip"]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "nls_body.h"), os.path.join(PKG_DIR, "..", "include", "dmet.h")]
ARCH = "gfx950"
CXXFLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-ffp-contract=off", "-Wall",
            "-Wno-unused-function", "-DNDEBUG"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build libdmet_hip.so")
    return exe


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _compile(src: str, force: bool) -> str:
    obj = os.path.join(OBJ_DIR, os.path.splitext(src)[0] + ".o")
    path = os.path.join(CSRC, src)
    if force or _stale(obj, [path] + HEADERS):
        cmd = [_hipcc()] + CXXFLAGS + ["-c", path, "-o", obj]
        subprocess.check_call(cmd)
    return obj


def build_hip(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ_DIR, exist_ok=True)
    with cf.ThreadPoolExecutor(max_workers=min(4, len(SOURCES))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), SOURCES))
    if force or _stale(LIB_PATH, objs):
        cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB_PATH] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    print(build_hip(force="--force" in sys.argv, verbose=True))
