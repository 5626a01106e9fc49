"""Builds build/variants/<name>.so: libetl_gfx950.so with ONE source recompiled under extra -D flags (measurement variants
for tools/*_probe.py through ETLG_LIB_PATH; never shipped).  usage: python tools/build_variant.py name source.hip -DX=1 ..."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from etl_amd import build as B

name, src, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
B.build_native()
out_dir = os.path.join(os.path.dirname(B.HERE), "build", "variants")
os.makedirs(out_dir, exist_ok=True)
obj = os.path.join(out_dir, f"{name}_{os.path.splitext(src)[0]}.o")
cmd = [B.HIPCC, "--offload-arch=gfx950", B.OPT.get(src, "-O3"), "-std=c++17", "-fPIC", "-Wno-everything"] + flags + ["-c", os.path.join(B.CSRC, src), "-o", obj]
if src.endswith(".cpp"):
    cmd[1:1] = ["-x", "hip"]
subprocess.check_call(cmd)
objs = [obj if s == src else os.path.join(B.CSRC, os.path.splitext(s)[0] + ".o") for s in B.SOURCES]
lib = os.path.join(out_dir, name + ".so")
subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
print(lib)
