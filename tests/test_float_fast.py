"""The device's float4 / float8 text parser (etl_amd/csrc/float_fast.h: exact fast path + deferral rule)
against glibc strtod / strtof on 6 million texts, through a host build of the same header."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_float_fast_path_matches_strtod(tmp_path):
    exe = str(tmp_path / "float_fast_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "etl_amd", "csrc"),
                           os.path.join(ROOT, "tests", "native", "float_fast_check.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-3000:]
    assert "mismatches 0" in out.stdout


def test_oracle_states_the_deferral_rule_independently(tmp_path):
    """The oracle decides "decoded on the device or DEFERRED" without the device's header (two glibc roundings of the decimals
    that bracket a long mantissa); the device decides with Clinger + Eisel-Lemire. Same verdict on 3.6 million texts x 2 widths."""
    src = open(os.path.join(ROOT, "oracle", "oracle_codec.hpp")).read()
    assert "float_fast.h\"" not in src.replace("etl_amd/csrc/float_fast.h is not", ""), "the oracle must not include the product's float header"
    exe = str(tmp_path / "float_rule_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "etl_amd", "csrc"), "-I", os.path.join(ROOT, "oracle"),
                           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "native", "float_rule_check.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-3000:]
    assert "mismatches 0" in out.stdout


def test_float_exact_fallback_matches_strtod(tmp_path):
    """etl_amd/csrc/float_slow.h — the finish pass's exact decimal -> binary conversion (Rust dec2flt's slow path, core::num::dec2flt::slow) —
    against glibc on 4 million texts: random decimals, the exact half-way points of both widths with their neighbours, mantissas beyond
    768 digits, and 260 k texts the fast path calls inconclusive."""
    exe = str(tmp_path / "float_slow_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "etl_amd", "csrc"),
                           os.path.join(ROOT, "tests", "native", "float_slow_check.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-3000:]
    assert "mismatches 0" in out.stdout
