"""Builds the native pieces in-tree (no JIT cache): libetl_gfx950.so (HIP kernels
+ C ABI, cross-compiled for gfx950 with hipcc — works without a GPU) and
libetlg_synth.so (synthetic WAL generator, plain g++)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libetl_gfx950.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

SOURCES = ["kernels.hip", "fused.hip", "cells.hip", "plan.hip", "scan.hip", "copy.hip", "host.cpp"]
# per-source optimisation level: k_fused is measurably faster built for size (88 vs 93 us on cfg2, tools/variants.sh);
# k_cells and the rest are not
OPT = {"fused.hip": "-Os"}
# per-source feature flags. k_fused is built with the fixed-width plan, the register / address-space fixes, scalar descriptor
# loads, the short kernel head and 8-deep staging: 78.2 us against 87.8 us for the plain build on cfg2, both timed in one run
# (profiles/r01j_ab_quick_*.json; tools/build_variants.py "all" vs "plain"). The other kernels keep the plain build until the
# same flags have been measured on them (tools/ab_variants.sh).
DEFS = {"fused.hip": ["-DETLG_FIXED_TILE", "-DETLG_HOT_FIXES", "-DETLG_SCALAR_COLS", "-DETLG_EARLY_SPAN", "-DETLG_STAGE_WIDE=8"]}
DEPS = SOURCES + ["../build.py", "dev_types.h", "codec.hip.h", "lookback.hip.h", "fixed_tile.hip.h", "utf8_swar.h", "float_fast.h", "pow5_table.h", os.path.join("..", "..", "include", "etlg.h")]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_native(force=False, verbose=False):
    deps = [os.path.join(CSRC, d) for d in DEPS]
    if not force and not _stale(LIB, deps):
        return LIB
    objs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, os.path.splitext(src)[0] + ".o")
        if force or _stale(obj, deps):
            cmd = [HIPCC, "--offload-arch=gfx950", OPT.get(src, "-O3"), "-std=c++17", "-fPIC", "-Wall",
                   "-Wno-unused-function"] + DEFS.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj]
            if src.endswith(".cpp"):
                cmd[1:1] = ["-x", "hip"]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


def build_all(force=False, verbose=False):
    from . import synth
    build_native(force=force, verbose=verbose)
    synth.build(force=force)
    return LIB


if __name__ == "__main__":
    build_all(force="--force" in os.sys.argv, verbose=True)
