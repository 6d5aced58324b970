"""Compile csrc/*.hip into libsparsify_hip.so for gfx950 (hipcc cross-compiles without a GPU).

    python -m sparsify_clip_amd._build [--force]

Objects are rebuilt only when a source or header is newer; the shared library is kept in-tree
(sparsify_clip_amd/libsparsify_hip.so, git-ignored) so that it travels with the repo snapshot.
"""
from __future__ import annotations

import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libsparsify_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -amdgpu-spill-vgpr-to-agpr=0: two GEMM kernels own the AGPRs (accumulators named directly in inline assembly); the register
# allocator must never use them as VGPR spill space (csrc/gemm_bf16.hip, audited by _asm_check.py)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-mllvm", "-amdgpu-spill-vgpr-to-agpr=0"]
# SC_EXTRA_HIPCC_FLAGS="-DSC_DEBUG_KNOBS": a diagnostic build in which the closed A/B knobs (sc_debug_env) read the environment again;
# use with --force, and rebuild without it before committing to a measurement
FLAGS += os.environ.get("SC_EXTRA_HIPCC_FLAGS", "").split()


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_extension(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    sources = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")))
    headers = sorted(glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h")))
    jobs = []
    objs = []
    for src in sources:
        obj = os.path.join(OBJ, os.path.basename(src) + ".o")
        objs.append(obj)
        if force or _newer(obj, [src] + headers):
            cmd = [HIPCC] + FLAGS + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", src, "-o", obj]
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        return r.stderr

    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            for warn in ex.map(run, jobs):
                if warn.strip() and verbose:
                    print(warn)
        gemm = os.path.join(CSRC, "gemm_bf16.hip")
        if any(gemm in cmd for cmd in jobs):   # the kernels with hand-managed AGPRs / in-flight LDS reads are audited whenever they are rebuilt
            from ._asm_check import check
            try:
                check(HIPCC, gemm, [f for f in FLAGS if f != "-fPIC"])
            except Exception:
                os.remove(os.path.join(OBJ, os.path.basename(gemm) + ".o"))   # a failed audit must not leave an object the next build would link
                raise
            if verbose:
                print("asm audit of gemm_bf16.hip: ok", flush=True)
    if force or jobs or _newer(LIB, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build_extension(force="--force" in sys.argv, verbose=True))
