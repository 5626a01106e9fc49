"""Builds flag variants of libetl_gfx950.so into etl_amd/variants/ (git-ignored; they travel with gpurun) so that
one GPU call can time them back to back (ETLG_LIB_PATH selects the library). Usage: python tools/build_variants.py"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from etl_amd.build import OPT as PRODUCT_OPT, DEFS as PRODUCT_DEFS  # noqa: E402
CSRC = os.path.join(ROOT, "etl_amd", "csrc")
OUT = os.path.join(ROOT, "etl_amd", "variants")
SOURCES = ["kernels.hip", "fused.hip", "cells.hip", "scan.hip", "copy.hip", "host.cpp"]
# A variant is a list of extra flags. Unless it names an optimisation level itself, every source is built at the
# level the product uses for it (etl_amd/build.py: OPT, -O3 otherwise). "product" IS the shipped library (it alone also gets the
# product's per-source feature flags, DEFS); every other variant is the plain sources plus exactly its flags, on every source.
VARIANTS = {
    "product": [],                       # per-source flags of etl_amd/build.py (DEFS): the shipped library
    "plain": [],                         # no feature flags anywhere (what shipped up to r01i)
    "base": ["-O3"],
    "os": ["-Os"],
    "nounroll": ["-fno-unroll-loops"],
    "maxilp": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"],
    "waves5": ["-DETLG_MINWAVES=5"],
    "waves3": ["-DETLG_MINWAVES=3"],
    "noinl": ["-DETLG_DECODE_NOINLINE"],
    "ablate": ["-DETLG_ABLATE"],          # early-exit ablations for tools/run_pmc_split.sh
    "stage8": ["-DETLG_STAGE_WIDE=8"],    # 8 staging loads in flight per lane instead of 4
    "hotfix": ["-DETLG_HOT_FIXES"],       # parsed integers stay in registers, row stores are global (not flat) stores, side tables are read with ds_read
    "hotfix_stage8": ["-DETLG_HOT_FIXES", "-DETLG_STAGE_WIDE=8"],
    # fixed-width plan (csrc/fixed_tile.hip.h): tiles of Begin / Commit / Insert-into-a-fixed-width-table frames are sized from
    # schema constants (one scan instead of three, no generic size_frame); everything else takes the generic body
    "fixed": ["-DETLG_FIXED_TILE"],
    "fixed_hotfix": ["-DETLG_FIXED_TILE", "-DETLG_HOT_FIXES"],
    "fixed_hotfix_stage8": ["-DETLG_FIXED_TILE", "-DETLG_HOT_FIXES", "-DETLG_STAGE_WIDE=8"],
    # wave-uniform row writer reads its slot / column descriptors with s_load (whole dwords through constant-address-space
    # pointers) instead of one global_load_ubyte/_ushort + vmcnt(0) + v_readfirstlane round trip per column
    "scols": ["-DETLG_SCALAR_COLS"],
    "hotfix_scols": ["-DETLG_HOT_FIXES", "-DETLG_SCALAR_COLS"],
    "fixed_hotfix_scols": ["-DETLG_FIXED_TILE", "-DETLG_HOT_FIXES", "-DETLG_SCALAR_COLS"],
    # kernel head with two dependent global round trips instead of six (side tables read as one concatenation and stored to LDS
    # after the staging loads were issued, span through real scalar loads, no barrier before the span loads)
    "early": ["-DETLG_EARLY_SPAN"],
    "early_stage8": ["-DETLG_EARLY_SPAN", "-DETLG_STAGE_WIDE=8"],
    "early_hotfix": ["-DETLG_EARLY_SPAN", "-DETLG_HOT_FIXES"],   # the candidate for k_cells: 8-deep staging costs it VGPR spills (12-14 vs 6-8)
    "all": ["-DETLG_FIXED_TILE", "-DETLG_HOT_FIXES", "-DETLG_SCALAR_COLS", "-DETLG_EARLY_SPAN", "-DETLG_STAGE_WIDE=8"],
    # "all" plus a k_fused instance with 128 frames per tile (two waves: output look-backs on one, transaction look-back on the
    # other); selected at run time with ETLG_FUSED_BLK=128 (without it the library behaves like "all")
    "all_blk128": ["-DETLG_FIXED_TILE", "-DETLG_HOT_FIXES", "-DETLG_SCALAR_COLS", "-DETLG_EARLY_SPAN", "-DETLG_STAGE_WIDE=8", "-DETLG_BLK128"],
}


def build(name, flags):
    tmp = os.path.join("/tmp/var", name)
    os.makedirs(tmp, exist_ok=True)
    objs = []
    for src in SOURCES:
        obj = os.path.join(tmp, os.path.splitext(src)[0] + ".o")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-Wno-unused-function"] + flags
        if name == "product":
            cmd += PRODUCT_DEFS.get(src, [])
        if not any(f.startswith("-O") for f in flags):
            cmd.append(PRODUCT_OPT.get(src, "-O3"))
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
