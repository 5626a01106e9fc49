"""Builds flag variants of libetl_gfx950.so into etl_amd/variants/ (git-ignored; they travel with gpurun) so that
one GPU call can time them back to back (ETLG_LIB_PATH selects the library). Usage: python tools/build_variants.py"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "etl_amd", "csrc")
OUT = os.path.join(ROOT, "etl_amd", "variants")
SOURCES = ["kernels.hip", "fused.hip", "cells.hip", "scan.hip", "copy.hip", "host.cpp"]
VARIANTS = {
    "base": ["-O3"],
    "nounroll": ["-O3", "-fno-unroll-loops"],
    "os": ["-Os"],
    "maxilp": ["-O3", "-mllvm", "-amdgpu-sched-strategy=max-ilp"],
    "waves5": ["-O3", "-DETLG_MINWAVES=5"],
    "waves3": ["-O3", "-DETLG_MINWAVES=3"],
    "noinl": ["-O3", "-DETLG_DECODE_NOINLINE"],
    "ablate": ["-Os", "-DETLG_ABLATE"],
}


def build(name, flags):
    tmp = os.path.join("/tmp/var", name)
    os.makedirs(tmp, exist_ok=True)
    objs = []
    for src in SOURCES:
        obj = os.path.join(tmp, os.path.splitext(src)[0] + ".o")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-Wno-unused-function"] + flags
        if src.endswith(".cpp"):
            cmd += ["-x", "hip"]
        cmd += ["-c", os.path.join(CSRC, src), "-o", obj]
        subprocess.check_call(cmd)
        objs.append(obj)
    lib = os.path.join(OUT, f"libetl_gfx950_{name}.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    return name, os.path.getsize(lib)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    names = sys.argv[1:] or list(VARIANTS)
    with ThreadPoolExecutor(4) as ex:
        for n, sz in ex.map(lambda n: build(n, VARIANTS[n]), names):
            print(n, sz)
