"""The finish pass (etlg_batch_finish_cells) against the oracle's restatement over many seeds: python tools/finish_fuzz.py [seconds=120] [seed0=100]
(tests/test_gpu_finish.py::test_fuzzed_literals_updates_and_key_images with fresh seeds: array literals of every element class, quoting,
escapes, damage, Insert / Update / Delete frames with key and full old images and unchanged-toast aliases; arena against arena)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_finish import test_fuzzed_literals_updates_and_key_images as one
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 100
t_end = time.time() + seconds
n = bad = 0
while time.time() < t_end:
    try:
        one(seed + n)
    except AssertionError as e:
        bad += 1
        print("seed", seed + n, "MISMATCH", str(e)[:400], flush=True)
    n += 1
print(f"finish fuzz: {n} rounds of 400 frames x 14 array / float columns, {bad} mismatches, seeds {seed}..{seed + n - 1}")
