cd $GRAFT_REPO_ROOT
ETLG_SCAN_DBG=1 timeout 120 python - <<'PY' 2>&1 | tail -4
import numpy as np
from etl_amd import synth
from etl_amd.decoder import Decoder
d = Decoder(0)
for mk in (synth.cfg2, synth.cfg3):
    w = mk(); buf, offs = w.fill(64 << 20)
    o = d.scan_boundaries(buf, max_frames=len(offs))
    o = d.scan_boundaries(buf, max_frames=len(offs))
    assert np.array_equal(o, offs)
PY
