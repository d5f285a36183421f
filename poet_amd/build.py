"""Build poet_amd/csrc/*.hip into the in-tree C-ABI library libpoet_hip.so for gfx950.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so is
git-ignored but travels with the tree to the GPU box.  No JIT, no torch extension machinery:
the boundary is a plain C ABI (include/poet_hip.h) loaded with ctypes.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libpoet_hip.so")
SOURCES = ["core.hip", "gemm.hip", "gemm_ws.hip", "msda.hip", "norm.hip", "attn.hip", "misc.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-fPIC", "-fno-gpu-rdc",
         "-Wno-unused-result"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.sep not in cand or os.path.exists(cand)):
            return cand
    return "hipcc"


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = True) -> str:
    hdrs = [os.path.join(CSRC, "common.cuh"), os.path.join(CSRC, "gemm.cuh"), os.path.join(os.path.dirname(HERE), "include", "poet_hip.h")]
    objs, jobs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            jobs.append([_hipcc(), *FLAGS, "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print("[poet_amd.build]", " ".join(cmd[-4:]), flush=True)
        subprocess.check_call(cmd)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB])
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv))
