"""TEST INFRASTRUCTURE — the emulated library (tests/simt) built by ROCm's clang++ (hipcc's front and middle end) at -O2 instead of
g++ -O1, and pytest run against it: a second reading of every undefined or implementation-defined corner of the kernel sources and of
host.cpp. (The AMDGPU back end — where a miscompiled switch of round 4 lived — stays out of reach without a GPU.)

    python tools/simt_other_compiler.py [pytest args ...]     # default: every scenario on every kernel path
Exit code 77 when the compiler is not there."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "simt"))
import build as simt_build  # noqa: E402

CLANG = os.environ.get("ETLG_SIMT_CLANG", "/opt/rocm/lib/llvm/bin/clang++")


def main():
    if not os.path.exists(CLANG):
        print("no", CLANG)
        return 77
    simt_build.CXX = CLANG
    lib = simt_build.build(extra_flags=["-O2", "-Wno-everything"], lib=os.path.join(simt_build.OUT, "libetlg_simt_clang.so"))
    env = dict(os.environ, ETLG_LIB_PATH=lib, ETLG_SIMT_RUN="1", ETLG_SIMT_WATCHDOG="3000")
    args = sys.argv[1:] or ["tests/test_gpu_parity.py", "-k", "scenario_parity"]
    return subprocess.call([sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-p", "no:cacheprovider"] + args, cwd=ROOT, env=env)


if __name__ == "__main__":
    sys.exit(main())
