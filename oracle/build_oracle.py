"""Compiles the oracle's plain-C restatement (gcc) into oracle/libmsda_ref.so.  Test infrastructure only.
The reference's own native extension is NOT buildable here (CUDA-only: setup.py:36-47 raises without CUDA,
sources use THC/THCAtomics.cuh; its CPU branch is a stub that throws) so there is no oracle/_ref."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def build():
    src, lib = os.path.join(HERE, "msda_ref.c"), os.path.join(HERE, "libmsda_ref.so")
    if not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", lib, src, "-lm"])
    return lib


if __name__ == "__main__":
    print(build())
