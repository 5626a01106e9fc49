"""The four-bytes-at-a-time UTF-8 predicate used by the kernels (etl_amd/csrc/utf8_swar.h)
against a byte-serial restatement of core::str::from_utf8, exhaustively over every 1-4 byte
combination of boundary values at every alignment (host build of the same header)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_swar_utf8_matches_serial_validator(tmp_path):
    exe = str(tmp_path / "utf8_swar_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "etl_amd", "csrc"),
                           os.path.join(ROOT, "tests", "native", "utf8_swar_check.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:]
    assert "mismatches 0" in out.stdout
