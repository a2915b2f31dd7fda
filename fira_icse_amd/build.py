"""Build libfira_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

The shared library is the product's compute path; there is no CPU fallback.  Sources are compiled in
parallel, one object per file, then linked; objects are cached by source mtime.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libfira_hip.so")
SOURCES = ["gemm_f32.hip", "gemm_bf16.hip", "gemm_wgrad_panel.hip", "gemm_bf16_panel.hip", "gemm_small.hip", "spmm.hip", "spmm_dense.hip", "gcn_fused.hip", "comb_fused.hip", "head_x3.hip", "rowops.hip", "attention.hip", "copyhead.hip", "beam.hip", "engine.hip", "layout.cpp", "hostlists.cpp"]
HEADERS = ["common.h", "engine.h", "epilogue.h", "mfma_frag.h", "x3.h", os.path.join("..", "..", "include", "fira_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-result"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def build(force: bool = False, verbose: bool = False) -> str:
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    hdr_time = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)
    hipcc = _hipcc()
    jobs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(objdir, src + ".o")
        if force or not os.path.exists(op) or os.path.getmtime(op) < max(os.path.getmtime(sp), hdr_time):
            cmd = [hipcc] + FLAGS + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", sp, "-o", op]
            jobs.append((src, cmd))

    def run(job):
        src, cmd = job
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, r.stderr[-4000:]))
        return src

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(run, jobs))
    objs = [os.path.join(objdir, s + ".o") for s in SOURCES]
    if jobs or not os.path.exists(LIB):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s" % r.stderr[-4000:])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
