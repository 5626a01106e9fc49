"""TEST INFRASTRUCTURE — the emulated library (tests/simt: kernel sources + host.cpp compiled with g++) under AddressSanitizer and
UndefinedBehaviorSanitizer, with "device" allocations of EXACTLY the size that was asked for (the default emulator build pads them):
an out-of-bounds read or write of a kernel or of the host code — past an input buffer, an arena, a descriptor block — is a report,
and so is undefined behaviour hipcc is as free to exploit as g++ (shifts by the width, signed overflow, out-of-range enum loads).
Alignment checks are off: the GPU reads unaligned words by design.

    python tools/simt_sanitize.py [pytest args ...]        # default: the table-copy, scan, plan, hand-off, async and shard files
    python tools/simt_sanitize.py --grid 8 tests/test_gpu_fuzz.py -k back_to_back     # with workgroups resident and interleaved

Prints the pytest tail and every sanitizer report (none expected). A run of round 5 is in profiles/r05_simt_sanitizers.txt."""
import glob
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "simt"))
import build as simt_build  # noqa: E402

SAN_FLAGS = ["-fsanitize=address,undefined", "-fno-sanitize=alignment,vptr", "--param", "asan-stack=0", "-fno-omit-frame-pointer", "-DSIMT_MALLOC_SLACK=0"]
DEFAULT = ["tests/test_gpu_copy.py", "tests/test_gpu_scan.py", "tests/test_gpu_fixed_plan.py", "tests/test_gpu_columns.py", "tests/test_gpu_rowbinary.py",
           "tests/test_gpu_protobuf.py", "tests/test_gpu_size_hints.py", "tests/test_arrow_kats.py", "tests/test_gpu_json_display.py", "tests/test_gpu_async.py",
           "tests/test_shard_decode.py", "-k", "not 16777216 and not device_resident and not device_input and not 8-"]


def main():
    args = sys.argv[1:]
    grid = None
    if args[:1] == ["--grid"]:
        grid, args = args[1], args[2:]
    lib = simt_build.build(extra_flags=SAN_FLAGS, lib=os.path.join(simt_build.OUT, "libetlg_simt_asan.so"))
    logs = os.path.join(simt_build.OUT, "san_logs")
    shutil.rmtree(logs, ignore_errors=True)
    os.makedirs(logs)
    gcc_lib = lambda n: subprocess.check_output(["gcc", f"-print-file-name={n}"], text=True).strip()   # noqa: E731
    env = dict(os.environ, ETLG_LIB_PATH=lib, ETLG_SIMT_RUN="1", ETLG_SIMT_WATCHDOG="3400",
               ASAN_OPTIONS=f"detect_leaks=0:detect_stack_use_after_return=0:halt_on_error=1:log_path={logs}/asan",
               UBSAN_OPTIONS=f"print_stacktrace=1:halt_on_error=0:log_path={logs}/ubsan",
               LD_PRELOAD=f"{gcc_lib('libasan.so')} {gcc_lib('libubsan.so')}")
    if grid:
        env["ETLG_SIMT_GRID"] = grid
    cmd = [sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-p", "no:cacheprovider", "-n", "3"] + (args or DEFAULT)
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True)
    print("\n".join(out.stdout.strip().splitlines()[-4:]))
    reports = 0
    for f in sorted(glob.glob(os.path.join(logs, "*"))):
        text = open(f).read()
        body = [ln for ln in text.splitlines() if "doesn't fully support makecontext" not in ln and ln.strip()]
        if body:
            reports += 1
            print(f"---- {os.path.basename(f)}")
            print("\n".join(body[:60]))
    print(f"sanitizer reports: {reports}")
    return 1 if (out.returncode or reports) else 0


if __name__ == "__main__":
    sys.exit(main())
