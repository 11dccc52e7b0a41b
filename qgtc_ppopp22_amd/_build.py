"""In-tree build of the native pieces (called by __graft_entry__.build()).

  libqgtc_hip.so                    hipcc --offload-arch=gfx950  csrc/qgtc_hip.hip + qgtc_mfma.hip + qgtc_fp4.hip + qgtc_wide.hip + qgtc_epoch.hip + qgtc_chainx.hip + qgtc_stream.hip (seven translation
                                    units compiled in parallel; they include csrc/*.hip.h, the kernels)
  QGTC.cpython-*.so                 g++                          csrc/qgtc_torch.cpp (pybind11 binding)

Both land next to this file so that they travel with the repo snapshot to the GPU box (they are
git-ignored, not gpurun-ignored). hipcc cross-compiles gfx950 without a GPU. The translation units'
objects and -MD dependency files are kept under qgtc_ppopp22_amd/build/ (git- and gpurun-ignored):
only the units whose sources changed are recompiled.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
import sysconfig

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
INC = os.path.join(ROOT, "include")

HIP_LIB = os.path.join(PKG, "libqgtc_hip.so")
EXT_SUFFIX = sysconfig.get_config_var("EXT_SUFFIX") or ".so"
TORCH_EXT = os.path.join(PKG, "QGTC" + EXT_SUFFIX)


def _newer(target: str, sources: list[str]) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def _run(cmd: list[str]) -> None:
    print("[qgtc build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)


HIP_UNITS = ("qgtc_hip.hip", "qgtc_mfma.hip", "qgtc_fp4.hip", "qgtc_wide.hip", "qgtc_epoch.hip", "qgtc_chainx.hip", "qgtc_stream.hip")   # translation units of libqgtc_hip.so, compiled in parallel


OBJ_DIR = os.path.join(PKG, "build")    # objects + dependency files of the translation units (git- and gpurun-ignored)


def _unit_stale(obj: str, dep: str) -> bool:
    """A translation unit is rebuilt when its object is missing or older than any file its last compile read (-MD)."""
    if not (os.path.exists(obj) and os.path.exists(dep)):
        return True
    t = os.path.getmtime(obj)
    text = open(dep).read().replace("\\\n", " ")
    files = text.split(":", 1)[1].split() if ":" in text else []
    return any((not os.path.exists(f)) or os.path.getmtime(f) > t for f in files if not f.startswith("/opt/rocm"))


def build_hip(force: bool = False) -> str:
    units = [os.path.join(CSRC, u) for u in HIP_UNITS]
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    objs, procs = [], []
    for u in units:
        obj = os.path.join(OBJ_DIR, os.path.basename(u) + ".o")
        dep = obj[:-2] + ".d"
        objs.append(obj)
        if not force and not _unit_stale(obj, dep):
            continue
        # kernels with scalar arguments get them preloaded into SGPRs (no s_load round trip at the head of the kernel)
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-Wno-unused-value", "-Wno-pass-failed",
               "-mllvm", "-amdgpu-kernarg-preload-count=16", f"-I{INC}", "-MD", "-MF", dep, "-o", obj, u]
        print("[qgtc build]", " ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    if procs or not os.path.exists(HIP_LIB) or any(os.path.getmtime(o) > os.path.getmtime(HIP_LIB) for o in objs):
        _run([hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", HIP_LIB] + objs)
    return HIP_LIB


def build_torch_ext(force: bool = False) -> str:
    src = [os.path.join(CSRC, "qgtc_torch.cpp"), os.path.join(INC, "qgtc.h")]
    if not force and _newer(TORCH_EXT, src + [HIP_LIB]):
        return TORCH_EXT
    import torch
    from torch.utils import cpp_extension as ce

    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    incs = [f"-I{p}" for p in ce.include_paths()] + [f"-I{INC}", "-I/opt/rocm/include",
                                                      f"-I{sysconfig.get_paths()['include']}"]
    abi = int(torch.compiled_with_cxx11_abi())
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function",
           "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DTORCH_EXTENSION_NAME=QGTC",
           "-DTORCH_API_INCLUDE_EXTENSION_H", f"-D_GLIBCXX_USE_CXX11_ABI={abi}",
           *incs, src[0], "-o", TORCH_EXT,
           f"-L{PKG}", "-lqgtc_hip", f"-L{tlib}", "-lc10", "-lc10_hip", "-ltorch", "-ltorch_cpu",
           "-ltorch_hip", "-ltorch_python",
           "-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{tlib}", "-Wl,-rpath,/opt/rocm/lib"]
    _run(cmd)
    return TORCH_EXT


def kernel_source_hash() -> str:
    """12 hex digits over the sources that define the kernels and their launchers (csrc/*.hip, csrc/*.hip.h, include/qgtc.h):
    profiles/ summaries carry it, and bench.py prints counter-derived traffic only when the summary's hash is the tree's."""
    import hashlib

    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip") or f.endswith(".hip.h"))
    for f in files + [os.path.join(INC, "qgtc.h")]:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:12]


def build_all(force: bool = False) -> dict:
    return {"hip": build_hip(force), "torch_ext": build_torch_ext(force)}


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv))
