# rocprofv3 evidence for the shipped build: --kernel-trace --stats of the bench command (cfg2 headline; cfg3 workload), then
# the HBM traffic passes (separate --pmc runs, short command) and one SQ counter pass on cfg2
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r02p; mkdir -p $O
for wl in cfg2 cfg3; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$wl -o p -- python bench.py --workload $wl --steps 6 --warmup 1 --legs= --no-cpu-baseline > $O/bench_${wl}_under_rocprof.json 2> $O/prof_$wl.err
  f=$(ls $O/prof_$wl/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && { cp $f $O/${wl}_kernel_stats.csv; head -4 $f; }
  t=$(ls $O/prof_$wl/*kernel_trace.csv 2>/dev/null | head -1); [ -n "$t" ] && head -40 $t > $O/${wl}_kernel_trace_head.csv
  rm -rf $O/prof_$wl
done
bash tools/traffic.sh cfg2; bash tools/traffic.sh cfg3
cp gpurun_out/traffic_cfg2.json gpurun_out/traffic_cfg3.json $O/ 2>/dev/null
B="python bench.py --workload cfg2 --steps 1 --warmup 1 --inner 4 --prime 2 --legs= --no-cpu-baseline"
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --output-format csv -d $O/pmc_$i -o p -- $B > $O/pmc_$i.log 2>&1
  f=$(ls $O/pmc_$i/*counter_collection.csv 2>/dev/null | head -1)
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
