"""Build libcaptioner_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m embodied_captioning_amd.build [--force] [--experiments]

--experiments adds -DCAP_EXPERIMENTS: the instrumented (cycle-stamp) instantiations of the 256x256 GEMM kernels that
tools/gemm_cycles.py and tools/bench_gemm_split.py --cycles read (tile ids 9 / 13).  The default library has none of them and
reads no environment variable.

One translation unit per .hip file, compiled in parallel, linked into embodied_captioning_amd/lib/.
"""
from __future__ import annotations

import concurrent.futures
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "build")
LIB = os.path.join(LIBDIR, "libcaptioner_hip.so")
SOURCES = ["captioner.hip", "gemm.hip", "gemm_pp.hip", "gemm_skinny.hip", "elementwise.hip", "attention.hip", "decode_small.hip", "beam.hip", "preprocess.hip"]
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def _digest(flags) -> str:
    h = hashlib.sha256()
    for fn in sorted(os.listdir(CSRC)) + ["../../include/captioner_hip.h"]:
        with open(os.path.join(CSRC, fn), "rb") as f:
            h.update(fn.encode()); h.update(f.read())
    h.update(" ".join(flags).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True, experiments: bool = False) -> str:
    flags = FLAGS + (["-DCAP_EXPERIMENTS"] if experiments else [])
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    stamp = os.path.join(LIBDIR, "libcaptioner_hip.sha256")
    digest = _digest(flags)
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == digest:
        return LIB
    hipcc = _hipcc()

    def compile_one(src: str) -> str:
        obj = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        cmd = [hipcc, *flags, "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    with concurrent.futures.ThreadPoolExecutor(max_workers=min(4, len(SOURCES))) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", LIB]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr}")
    with open(stamp, "w") as f:
        f.write(digest)
    if verbose:
        print(f"built {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, experiments="--experiments" in sys.argv)
