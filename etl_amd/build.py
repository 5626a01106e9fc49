"""Builds the native pieces in-tree (no JIT cache): libetl_gfx950.so (HIP kernels
+ C ABI, cross-compiled for gfx950 with hipcc — works without a GPU) and
libetlg_synth.so (synthetic WAL generator, plain g++)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libetl_gfx950.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

SOURCES = ["kernels.hip", "fused.hip", "cells.hip", "rows.hip", "plan.hip", "scan.hip", "copy.hip", "columns.hip", "host.cpp"]
# per-source optimisation level: k_fused is measurably faster built for size (88 vs 93 us on cfg2, tools/variants.sh);
# k_cells and the rest are not
# (round 6: so is k_copy_cells — cells.hip at -Os 468 / 304 us against 482-486 / 313-317 on the two table-copy workloads, same box;
# -O2, -fno-unroll-loops and the scheduler strategies move nothing on rows.hip / plan.hip: profiles/r06zv_compile_flag_sweep.txt)
OPT = {"fused.hip": "-Os", "cells.hip": "-Os"}
DEFS = {}   # no per-source feature flags: one code path per kernel
# what every object depends on beside its own source (the shared headers); host.cpp also on its parts
COMMON = ["../build.py", "dev_types.h", "codec.hip.h", "lookback.hip.h", "utf8_swar.h", "float_fast.h", "float_slow.h", "pow5_table.h", os.path.join("..", "..", "include", "etlg.h")]
EXTRA = {"fused.hip": ["fixed_tile.hip.h"], "host.cpp": ["host_state.h", "host_control.inc", "host_handoff.inc", "host_orchestrate.inc"]}
DEPS = SOURCES + COMMON + [d for v in EXTRA.values() for d in v]   # (the library as a whole)


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
        if force or _stale(obj, [os.path.join(CSRC, d) for d in [src] + COMMON + EXTRA.get(src, [])]):
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
