#!/bin/bash
# hardware counters of the fixed-width path (k_plan behind k_plan_pre) on cfg2: separate rocprofv3 --pmc passes over a short bench command,
# averages per launch and kernel -> gpurun_out/r04aw_cfg2_k_plan_counters.json
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B="python bench.py --workload cfg2 --steps 1 --warmup 1 --inner 4 --prime 2 --legs= --no-cpu-baseline"
O=gpurun_out/r04aw; rm -rf $O; mkdir -p $O
i=0
for set in "VALUBusy SALUBusy MemUnitStalled LDSBankConflict" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM" \
           "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"; do
  i=$((i+1))
  ETLG_OVERLAP=0 timeout 200 rocprofv3 --pmc $set --output-format csv -d $O/p$i -o p -- $B > $O/p$i.log 2>&1
done
python - $O <<'PY'
import csv, sys, json, collections, glob
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(sys.argv[1] + "/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "etlg::k_plan" not in k: continue
        k = k.split("etlg::")[1].split("(")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
out = {k: {c: round(v / n[(k, c)], 2) for c, v in d.items()} for k, d in agg.items()}
out["how"] = "rocprofv3 --pmc, four separate passes over `ETLG_OVERLAP=0 python bench.py --workload cfg2 --steps 1 --warmup 1 --inner 4 --prime 2 --legs= --no-cpu-baseline` (tools/r04_counters.sh); averages per launch"
json.dump(out, open("gpurun_out/r04aw_cfg2_k_plan_counters.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf $O/p*/
