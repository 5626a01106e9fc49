"""Randomised stress of the ASYNC batch chain against the oracle, on the real library (MI355X) or the emulator build:
python tools/async_fuzz.py [seconds=60] [seed=1]        (the round itself: tests/test_gpu_async.py async_fuzz_round)
Every round: one synthetic stream (cfg2 / cfg3 / cfg5), sometimes with a few damaged bytes (errors, plan give-ups), cut into 2-40
batches at arbitrary FRAME boundaries; the batches are enqueued ETLG_F_ASYNC with a random window of batches in flight, with or without
the offsets sidecar, with or without the caller's no-control assertion, and synced in issue order. Each batch must match what the
oracle makes of the same bytes in the same order: error (code, frame) and every array of the arena."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_async import async_fuzz_round

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t_end = time.time() + seconds
rounds = batches = bad = 0
seed = seed0
while time.time() < t_end or rounds == 0:
    if os.environ.get("ETLG_FUZZ_VERBOSE"):
        print("round", seed, flush=True)   # (a crash names its seed)
    n, problems = async_fuzz_round(seed)
    for line in problems:
        print("MISMATCH", line, flush=True)
    bad += len(problems)
    batches += n
    rounds += 1
    seed += 1
print(f"async fuzz: {rounds} chains, {batches} batches checked, {bad} mismatches, seeds {seed0}..{seed - 1}")
sys.exit(1 if bad else 0)
