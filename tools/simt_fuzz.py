"""Long mutation fuzz of the EMULATED kernels (tests/simt) against the oracle: the GPU fuzz test is bounded by GPU
minutes, this one only by CPU time. usage: python tools/simt_fuzz.py [seconds=600] [seed0=1]
Prints one line per disagreement (seed, path, workload, iteration) and a summary; exit code 1 if any."""
import os
import random
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "simt"))

WORKER = r'''
import os, sys, random, time
sys.path.insert(0, %(root)r)
from etl_amd import synth
from etl_amd.decoder import Decoder
from oracle import oracle
from tests.test_gpu_fuzz import _mutate
seed, budget = int(sys.argv[1]), float(sys.argv[2])
rng = random.Random(seed)
mk = rng.choice([synth.cfg2, synth.cfg3, synth.cfg5])
w = mk()
for _ in range(rng.randrange(4)):
    w.fill(rng.randrange(8 << 10, 64 << 10))       # start somewhere else in the stream
buf, offs = w.fill(rng.randrange(16 << 10, 80 << 10))
t0 = time.time(); it = 0; bad = 0
while time.time() - t0 < budget:
    mb, mo = buf, offs
    for _ in range(rng.choice([1, 1, 2, 3])):
        if len(mo) < 3:
            break
        try:
            mb, mo = _mutate(rng, mb, mo)
        except Exception:      # a second mutation can hit a frame the first one cut short
            break
    sidecar = rng.random() < 0.8
    o, d = oracle.Oracle(), Decoder(0)
    w.register(o, ready=not w.cfg.emit_relations); w.register(d, ready=not w.cfg.emit_relations)
    rb = o.decode(mb, mo if sidecar else None); gb = d.decode(mb, mo if sidecar else None)
    e = gb.error
    got = (e.code, e.kind, e.description, e.frame_index) if e else (0, 0, "", -1)
    want = (rb.err_code, rb.err_kind, rb.err_desc, rb.err_frame)
    diff = [] if want != got else rb.host_batch().diff(gb.host())
    if want != got or diff:
        bad += 1
        print("MISMATCH seed", seed, "it", it, mk.__name__, os.environ.get("ETLG_FUSED_KERNEL"), os.environ.get("ETLG_FORCE_MULTIPASS"), "sidecar", sidecar, want, got, diff[:3], flush=True)
    d.close(); it += 1
print("DONE seed", seed, mk.__name__, "iterations", it, "mismatches", bad, flush=True)
'''

COPY_WORKER = r'''
import os, sys, random, time
sys.path.insert(0, %(root)r)
import numpy as np
from tests.test_gpu_copy import _gen_rows, GEN_COLS, both
seed, budget = int(sys.argv[1]), float(sys.argv[2])
rng = random.Random(seed)
t0 = time.time(); it = 0; bad = 0
while time.time() - t0 < budget:
    rows = [bytearray(r) for r in _gen_rows(rng.randrange(1, 300), rng.randrange(1 << 30))]
    for _ in range(rng.choice([0, 1, 1, 2, 4])):       # byte-level damage: flips, dropped / doubled separators, cut rows, raw bytes
        r = rows[rng.randrange(len(rows))]
        if not r: continue
        k = rng.choice(["flip", "tab", "untab", "cut", "raw", "nl", "bs"])
        i = rng.randrange(len(r))
        if k == "flip": r[i] ^= 1 << rng.randrange(8)
        elif k == "tab": r[i:i] = b"\t"
        elif k == "untab":
            j = r.find(b"\t")
            if j >= 0: del r[j]
        elif k == "cut": del r[i:]
        elif k == "raw": r[i] = rng.choice(b"\x00\xff\xc3\x80\\N")
        elif k == "nl": r[i:i] = b"\n"
        elif k == "bs": r[i:i] = b"\\"
    o, d, rb, gb = both(GEN_COLS, [bytes(r) for r in rows])
    e = gb.error
    got = (e.code, e.kind, e.description, e.frame_index) if e else (0, 0, "", -1)
    want = (rb.err_code, rb.err_kind, rb.err_desc, rb.err_frame)
    diff = [] if want != got else rb.host_batch().diff(gb.host())
    if want != got or diff:
        bad += 1
        print("MISMATCH copy seed", seed, "it", it, os.environ.get("ETLG_FUSED_KERNEL"), os.environ.get("ETLG_FORCE_MULTIPASS"), want, got, diff[:3], flush=True)
    d.close(); it += 1
print("DONE seed", seed, "copy", "iterations", it, "mismatches", bad, flush=True)
'''

SCAN_WORKER = r'''
import os, sys, random, struct, time
sys.path.insert(0, %(root)r)
import numpy as np
from etl_amd.decoder import Decoder
from tests.test_gpu_scan import ref_scan
seed, budget = int(sys.argv[1]), float(sys.argv[2])
rng = random.Random(seed)
dec = Decoder(0)
t0 = time.time(); it = 0; bad = 0
def payload(n):
    k = rng.random()
    if k < 0.3: return bytes(rng.getrandbits(8) for _ in range(n))
    if k < 0.6: return bytes(rng.choice(b"d\x00\x00\x01\x10w") for _ in range(n))      # header look-alikes everywhere
    return (b"d" + struct.pack(">I", rng.choice([4, 5, 17, 60, 200, 4000])) + b"w") * (n // 6 + 1)
while time.time() - t0 < budget:
    parts = []
    total = rng.choice([0, 1, 4, 5, 300, 5000, 9000, 40000, 150000])
    size = 0
    while size < total:
        n = rng.choice([0, 1, 20, 108, 108, 108, 500, 3000, 9000, 20000, 70000])
        body = payload(n)[:n]
        fr = b"d" + struct.pack(">I", len(body) + 4) + body
        k = rng.random()
        if k < 0.03: fr = fr[:rng.randrange(1, len(fr) + 1)]                       # truncated frame in the middle
        elif k < 0.05: fr = bytes([rng.getrandbits(8)]) + fr[1:]                    # not a 'd'
        elif k < 0.07: fr = fr[:1] + struct.pack(">I", rng.choice([0, 3, 2**31, 2**32 - 1, len(body) + 5])) + fr[5:]
        parts.append(fr); size += len(fr)
    buf = np.frombuffer(b"".join(parts), dtype=np.uint8)
    if rng.random() < 0.3 and len(buf) > 3: buf = buf[:rng.randrange(len(buf))]
    got = dec.scan_boundaries(buf); want = ref_scan(buf)
    if len(got) != len(want) or not np.array_equal(got, want):
        bad += 1
        print("MISMATCH scan seed", seed, "it", it, "len", len(buf), "frames", len(want) - 1, len(got) - 1, flush=True)
    it += 1
dec.close()
print("DONE seed", seed, "scan", "iterations", it, "mismatches", bad, flush=True)
'''

PATHS = [{}, {"ETLG_FUSED_KERNEL": "0"}, {"ETLG_FUSED_KERNEL": "1"}, {"ETLG_FUSED_KERNEL": "2"}, {"ETLG_FORCE_MULTIPASS": "1"},
         {"ETLG_FUSED_KERNEL": "3"}, {"ETLG_FUSED_KERNEL": "3", "ETLG_PLAN_PRE": "0"}]   # ... the plan forced: behind its sidecar pre-pass, with its own look-back


def main():
    import build as simt_build
    lib = os.environ.get("ETLG_SIMT_FUZZ_LIB") or simt_build.build()   # a variant build of the emulator library (tests/simt/build.py: extra_flags)
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 600.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    t_end = time.time() + seconds
    total = bad = 0
    procs = []
    nproc = int(os.environ.get("ETLG_FUZZ_PROCS", 0)) or max(1, (os.cpu_count() or 2) - 1)   # (on a GPU box with the gfx950 library as ETLG_SIMT_FUZZ_LIB: a handful of contexts, not one per core)
    while time.time() < t_end or procs:
        while time.time() < t_end and len(procs) < nproc:
            env = dict(os.environ, ETLG_LIB_PATH=lib, ETLG_SIMT_RUN="1", ETLG_SIMT_WATCHDOG="120", **PATHS[seed % len(PATHS)])
            worker = COPY_WORKER if seed % 4 == 0 else SCAN_WORKER if seed % 4 == 2 else WORKER   # stream / table-copy / boundary-scan
            procs.append(subprocess.Popen([sys.executable, "-c", worker % {"root": ROOT}, str(seed), "20"], env=env, cwd=ROOT,
                                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
            seed += 1
        for p in list(procs):
            if p.poll() is not None:
                out = p.stdout.read()
                procs.remove(p)
                for line in out.splitlines():
                    if line.startswith("MISMATCH") or "watchdog" in line or "Error" in line or "Traceback" in line:
                        print(line, flush=True)
                        bad += 1
                    if line.startswith("DONE"):
                        total += int(line.split("iterations")[1].split()[0])
                if p.returncode != 0:
                    print("worker exit", p.returncode, out[-400:], flush=True)
                    bad += 1
        time.sleep(0.2)
    print(f"simt fuzz: {total} mutated batches, {bad} problems, seeds up to {seed - 1}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
