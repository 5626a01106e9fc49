#!/bin/bash
# Hardware counters of one kernel: separate rocprofv3 --pmc passes over a short command, averages per launch -> gpurun_out/<label>_counters.json
# usage: bash tools/pmc_kernel.sh <label> <kernel substring> <command...>      (measurement tool, not product)
L=$1; K=$2; shift 2
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_$L; rm -rf $O; mkdir -p $O
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_INSTS_BRANCH SQ_INSTS_FLAT SQ_INST_LEVEL_SMEM SQ_INSTS_FLAT_LDS_ONLY" \
           "FETCH_SIZE WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $O/p$i -o p -- "$@" > $O/p$i.log 2>&1
done
python - $O "$K" "$L" <<'PY'
import csv, sys, json, collections, glob, shutil
agg = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] not in r["Kernel_Name"]: continue
        agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
out = {c: round(v / n[c], 2) for c, v in agg.items()}
out["launches_seen"] = max(n.values()) if n else 0
json.dump(out, open(f"gpurun_out/{sys.argv[3]}_counters.json", "w"), indent=1)
print(json.dumps(out))
for d in glob.glob(sys.argv[1] + "/p*/"): shutil.rmtree(d, ignore_errors=True)
PY
