"""The fixed-width plan's sidecar pre-pass (etl_amd/csrc/plan.hip, k_plan_pre) prices a frame by its length: a pgoutput Begin is 51 bytes
on the wire, a Commit 56 (CopyData 'd' + Int32 length, XLogData 'w' + wal_start + wal_end + timestamp, then the message — 'B' final_lsn:8
timestamp:8 xid:4; 'C' flags:1 commit_lsn:8 end_lsn:8 timestamp:8: the layouts of postgres-replication 0.6.7's LogicalReplicationMessage
parser, which the reference decodes through, crates/etl/src/replication/apply.rs:2037-2125). This file pins the two constants to the
frames the test suite and the oracle agree on; what a stream does to break the assumption is the GPU tier's business
(tests/test_gpu_fixed_plan.py::test_prepass_*)."""
import re

import numpy as np

from tests import pgwire as W


def _frame_len(msg):
    s = W.Stream()
    s.add(msg)
    return len(s.bytes())


def test_begin_and_commit_frames_have_the_lengths_the_pre_pass_assumes():
    assert _frame_len(W.begin(0x5000, ts=7, xid=9)) == 51
    assert _frame_len(W.commit(0x5000, 0x5008, ts=7, flags=0)) == 56
    src = open("etl_amd/csrc/plan.hip").read()
    body_off = int(re.search(r"constexpr uint32_t kBodyOff = (\d+);", open("etl_amd/csrc/codec.hip.h").read()).group(1))
    m = re.search(r"kPreBeginLen = kBodyOff \+ (\d+)u, kPreCommitLen = kBodyOff \+ (\d+)u", src)
    assert m and body_off + int(m.group(1)) == 51 and body_off + int(m.group(2)) == 56


def test_the_oracle_reads_those_frames_as_a_transaction():
    from oracle import oracle
    o = oracle.Oracle()
    s = W.Stream()
    s.add(W.begin(0x5000, ts=7, xid=9))
    s.add(W.commit(0x5000, 0x5008, ts=8, flags=0))
    rb = o.decode(np.frombuffer(s.bytes(), dtype=np.uint8), s.offsets)
    assert rb.err_code == 0
    h = rb.host_batch()
    assert [chr(k) for k in h.kind] == ["B", "C"] and int(h.commit_lsn[0]) == 0x5000
