# PMC counter passes for the dominant kernel: bash tools/pmc.sh [cfg2|cfg3] ; results -> gpurun_out/pmc_<wl>_*.csv
WL=${1:-cfg2}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B="python bench.py --workload $WL --steps 4 --warmup 1 --no-cpu-baseline --no-check"
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum" \
           "FETCH_SIZE WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_${WL}_$i -o p -- $B > gpurun_out/pmc_${WL}_$i.log 2>&1
  f=$(ls gpurun_out/pmc_${WL}_$i/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][:40]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    n[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    if "k_" in k:
        print(k, {c: round(v / n[(k, c)]) for c, v in d.items()})
PY
done
