"""bench.py's all-cores CPU leg (SURVEY.md §8(d) leg (ii)): every frame of every batch is decoded exactly once
across the commit-aligned shards, by independent oracle contexts."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import bench  # noqa: E402
from etl_amd import synth  # noqa: E402


def test_threads_leg_covers_every_frame():
    w = synth.cfg3()
    pool = [w.fill(2 << 20) for _ in range(2)]
    r = bench.cpu_baseline_threads(w, pool, 3, 0.2)
    assert r["cores"] == 3 and r["unit"] == "GB/s" and r["value"] > 0
    frames = sum(len(o) - 1 for _, o in pool)
    nbytes = sum(len(b) for b, _ in pool)
    # events_per_s and value are quoted on the same best pass: their ratio is frames / bytes of the pool
    assert abs(r["events_per_s"] / (r["value"] * 1e9) - frames / nbytes) < 0.01 * frames / nbytes


def test_cpu_threads_default_is_bounded():
    assert 1 <= bench.cpu_threads(0) <= 64
    assert bench.cpu_threads(5) == 5
