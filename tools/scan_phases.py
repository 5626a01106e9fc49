"""Phase clocks of k_bounds (ETLG_SCAN_DBG=1) on one 64 MiB cfg2 / cfg3 batch: stage, guess, walk, stitch, serial walk, look-back, write."""
import os
import sys

os.environ["ETLG_SCAN_DBG"] = "1"
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from etl_amd import synth
from etl_amd.decoder import Decoder

for mk in (synth.cfg2, synth.cfg3):
    w = mk()
    d = Decoder(0)
    buf, offs = w.fill(64 << 20)
    tb = torch.from_numpy(buf.copy()).cuda()
    out = torch.zeros(len(offs) + 64, dtype=torch.int32, device="cuda")
    for _ in range(2):
        n = d.scan_boundaries_device(tb.data_ptr(), tb.numel(), out.data_ptr(), out.numel())
    assert n == len(offs) - 1
    print(w.name, "frames", n, "reruns/seq", d.debug_scan(), file=sys.stderr)
