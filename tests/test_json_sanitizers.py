"""json_valid / json_display as host C++ under AddressSanitizer + UndefinedBehaviorSanitizer (g++) and MemorySanitizer (ROCm's clang++),
over seeded random documents held in heap blocks of their exact size (tools/json_sanitizers.py): no read past a text's end, no
undefined behaviour, no read of an uninitialised value, and the bytes the oracle writes."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"), reason="ROCm's clang++ (MemorySanitizer) is not here")
def test_json_display_under_the_sanitizers():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "json_sanitizers.py"), "2500", "11"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("outputs equal the oracle: True") == 2, r.stdout
