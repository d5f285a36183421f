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
SOURCES = ["core.hip", "gemm.hip", "gemm_ws.hip", "gemm_dw.hip", "gemm_dwr.hip", "gemm_small.hip", "gemm_pipe.hip", "msda.hip", "norm.hip", "attn.hip", "misc.hip"]
# Kernels that were built, measured and LOST live under profiles/probes/kernels/ and are NOT in the product
# library.  POET_BUILD_PROBES=1 compiles them in (-DPOET_PROBE_KERNELS: gemm_wr.hip + the msda_*.inc fragments msda.hip includes)
# so that their A/B scripts under profiles/probes/ still run; each stays behind its own opt-in environment switch.
PROBES = os.environ.get("POET_BUILD_PROBES", "0") not in ("", "0")
PROBE_DIR = os.path.join(os.path.dirname(HERE), "profiles", "probes", "kernels")
PROBE_SOURCES = ["gemm_wr.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-fPIC", "-fno-gpu-rdc",
         "-Wno-unused-result", "-Rpass-analysis=kernel-resource-usage"]
RESOURCES = os.path.join(CSRC, "kernel_resources.txt")      # per-kernel VGPRs / scratch of the last build (git-ignored)


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


def _is_remark_context(line: str) -> bool:
    t = line.strip()
    return t.startswith("|") or (t[:1].isdigit() and " | " in t)


def _resource_report(stderr: str, src: str):
    """Parse -Rpass-analysis=kernel-resource-usage remarks: [(source, kernel, vgprs, agprs, scratch_bytes, occupancy)]."""
    out, cur = [], None
    for line in stderr.splitlines():
        if "remark:" not in line:
            continue
        body = line.split("remark:", 1)[1].split("[-Rpass")[0].strip()
        if body.startswith("Function Name:"):
            cur = dict(src=src, name=body.split(":", 1)[1].strip(), vgprs=0, agprs=0, scratch=0, occ=0)
            out.append(cur)
        elif cur is not None and ":" in body:
            k, v = body.rsplit(":", 1)
            try:
                v = int(v)
            except ValueError:
                continue
            if k.startswith("VGPRs Spill"):
                cur["spill"] = v
            elif k.startswith("VGPRs"):
                cur["vgprs"] = v
            elif k.startswith("AGPRs"):
                cur["agprs"] = v
            elif k.startswith("ScratchSize"):
                cur["scratch"] = v
            elif k.startswith("Occupancy"):
                cur["occ"] = v
    return out


def _write_resources(reports, verbose):
    """A kernel that starts using scratch memory (register spills, or a by-value argument array indexed at run time) is
    typically 2-3x slower and nothing else tells you: list them loudly and keep the full table next to the library."""
    rows = [k for rep in reports for k in rep]
    old = {}
    rebuilt = {k["src"] for k in rows}
    if os.path.exists(RESOURCES):
        for line in open(RESOURCES):
            parts = line.rstrip("\n").split("\t")
            if len(parts) == 6 and parts[0] not in rebuilt:  # rows of a recompiled source are replaced wholesale (no stale kernels)
                old[(parts[0], parts[1])] = line
    for k in rows:
        old[(k["src"], k["name"])] = f'{k["src"]}\t{k["name"]}\t{k["vgprs"]}\t{k["agprs"]}\t{k["scratch"]}\t{k["occ"]}\n'
    with open(RESOURCES, "w") as f:
        f.writelines(old[key] for key in sorted(old))
    bad = [k for k in rows if k["scratch"] > 0]
    if bad:
        print(f"[poet_amd.build] ERROR: {len(bad)} kernel(s) use scratch memory:", file=sys.stderr)
        for k in bad:
            print(f'    {k["src"]}: {k["name"]}  scratch={k["scratch"]} B/lane  vgprs={k["vgprs"]}', file=sys.stderr)
        if os.environ.get("POET_ALLOW_SCRATCH", "0") in ("", "0"):
            for k in bad:                                     # make the next build recompile (and re-check) the offender
                obj = os.path.join(CSRC, k["src"].replace(".hip", ".o"))
                if os.path.exists(obj):
                    os.remove(obj)
            raise RuntimeError("poet_amd.build: kernels with scratch memory (set POET_ALLOW_SCRATCH=1 to build anyway)")


def build_library(force: bool = False, verbose: bool = True) -> str:
    hdrs = [os.path.join(CSRC, "common.cuh"), os.path.join(CSRC, "gemm.cuh"), os.path.join(os.path.dirname(HERE), "include", "poet_hip.h")]
    flags = FLAGS + (["-DPOET_PROBE_KERNELS", "-I" + CSRC] if PROBES else [])
    stamp = os.path.join(CSRC, ".probes_on")                    # switching POET_BUILD_PROBES invalidates every object
    if PROBES != os.path.exists(stamp):
        force = True
        open(stamp, "w").close() if PROBES else os.remove(stamp)
    if PROBES:
        hdrs += [os.path.join(PROBE_DIR, f) for f in os.listdir(PROBE_DIR) if f.endswith(".inc")]
    objs, jobs = [], []
    for s in SOURCES + (PROBE_SOURCES if PROBES else []):
        src = os.path.join(CSRC if s in SOURCES else PROBE_DIR, s)
        obj = os.path.join(CSRC, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            jobs.append([_hipcc(), *flags, "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print("[poet_amd.build]", " ".join(cmd[-4:]), flush=True)
        r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
        usage = _resource_report(r.stderr, os.path.basename(cmd[-3]))
        rest = "\n".join(l for l in r.stderr.splitlines() if "kernel-resource-usage" not in l and not _is_remark_context(l))
        if rest.strip():
            print(rest, file=sys.stderr, flush=True)
        if r.returncode:
            raise subprocess.CalledProcessError(r.returncode, cmd)
        return usage

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            reports = list(ex.map(run, jobs))
        _write_resources([r for r in reports if r], verbose)
    if force or jobs or _stale(LIB, objs):
        run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB])
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv))
