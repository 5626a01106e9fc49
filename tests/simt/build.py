"""TEST INFRASTRUCTURE — builds tests/simt/_build/libetlg_simt.so: the kernel sources of etl_amd/csrc and host.cpp
compiled with g++ against the SIMT emulator (tests/simt/simt.h) instead of the HIP runtime. Never part of the product."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "etl_amd", "csrc")
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(OUT, "libetlg_simt.so")
KERNEL_SOURCES = ["kernels.hip", "fused.hip", "cells.hip", "rows.hip", "plan.hip", "scan.hip", "copy.hip", "columns.hip", "host.cpp"]
DEPS = KERNEL_SOURCES + ["dev_types.h", "host_state.h", "host_control.inc", "host_handoff.inc", "host_orchestrate.inc", "codec.hip.h", "lookback.hip.h", "fixed_tile.hip.h", "utf8_swar.h", "float_fast.h", "pow5_table.h"]
CXX = os.environ.get("CXX", "g++")
sys.path.insert(0, ROOT)
from etl_amd.build import DEFS as PRODUCT_DEFS  # noqa: E402
FLAGS = ["-std=c++17", "-O1", "-g", "-fPIC", "-pthread", "-fno-strict-aliasing", "-Wno-unknown-pragmas", "-Wno-attributes",
         "-I", os.path.join(HERE, "include"), "-x", "c++"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, extra_flags=(), lib=LIB):
    deps = [os.path.join(CSRC, d) for d in DEPS] + [os.path.join(ROOT, "include", "etlg.h"), os.path.join(HERE, "simt.h"),
                                                   os.path.join(HERE, "simt.cpp"), os.path.abspath(__file__),
                                                   os.path.join(ROOT, "etl_amd", "build.py")]
    if not force and not _stale(lib, deps):
        return lib
    os.makedirs(OUT, exist_ok=True)
    tag = os.path.splitext(os.path.basename(lib))[0]

    def one(src):
        path = os.path.join(CSRC, src) if src != "simt.cpp" else os.path.join(HERE, src)
        obj = os.path.join(OUT, f"{tag}_{os.path.splitext(src)[0]}.o")
        # the default emulator build mirrors the product's per-source feature flags (etl_amd/build.py: DEFS)
        defs = [] if extra_flags else PRODUCT_DEFS.get(src, [])
        subprocess.check_call([CXX] + FLAGS + defs + list(extra_flags) + ["-c", path, "-o", obj])
        return obj

    with ThreadPoolExecutor(4) as ex:
        objs = list(ex.map(one, KERNEL_SOURCES + ["simt.cpp"]))
    subprocess.check_call([CXX, "-shared", "-pthread", "-o", lib] + objs)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
