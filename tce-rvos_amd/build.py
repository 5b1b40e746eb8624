"""Builds libtce_rvos.so (all HIP kernels + the C ABI) for gfx950, in-tree.

    python -m tce_rvos_amd.build        (or __graft_entry__.build())

hipcc cross-compiles without a GPU.  Objects are cached by source mtime under csrc/_obj/; translation units are
compiled in parallel (TCE_BUILD_JOBS, default = CPU count capped at 8).
"""
import glob
import os
from concurrent.futures import ThreadPoolExecutor
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libtce_rvos.so")
# slowest first (they gate the parallel build): the GEMM translation units carry one epilogue body per (act, res) combination
SOURCES = ["gemm_f16x3_big_256_split.hip", "gemm_f16x3_big_256_single.hip", "gemm_f16x3_big_128_split.hip", "gemm_f16x3_big_128_single.hip",
           "chain_ffn.hip", "chain_rowlin.hip", "gemm_f16x3_big.hip", "gemm_f16x3_small.hip", "gemm.hip", "attn.hip", "misc.hip", "norm.hip", "msda.hip",
           "text.hip", "resnet.hip", "frontend.hip", "fewrow.hip", "swinattn.hip", "thin.hip", "capi.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"] + \
    os.environ.get("HIPCC_EXTRA", "").split()  # audit builds only (e.g. -DFEWROW_RPT2); the shipped library has none


def _newer(src_list, target):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in src_list)


def build(verbose=True, force=False):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    # chain.hip is text shared by two translation units (chain_ffn.hip / chain_rowlin.hip): a dependency like a header
    headers = sorted(glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.inc"))) + [os.path.join(HERE, "..", "include", "tce_rvos.h"),
                                                             os.path.join(CSRC, "chain.hip")]
    objs, todo = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s.replace(".hip", ".o"))
        if force or _newer([src] + headers, obj):
            todo.append([HIPCC] + FLAGS + ["-c", src, "-o", obj])
        objs.append(obj)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    jobs = int(os.environ.get("TCE_BUILD_JOBS", min(8, os.cpu_count() or 1)))
    with ThreadPoolExecutor(max_workers=max(1, jobs)) as pool:
        list(pool.map(run, todo))
    if force or _newer(objs, LIB):
        tmp = LIB + f".tmp{os.getpid()}"  # link beside the target, rename into place: no reader ever sees a partial file
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
