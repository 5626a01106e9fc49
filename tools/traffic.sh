# HBM traffic of the dominant kernel for bench.py's roofline.traffic: two rocprofv3 --pmc passes
# (FETCH_SIZE and WRITE_SIZE do not fit one pass) over a SHORT bench command, averaged per launch.
#   bash tools/traffic.sh [cfg2|cfg3]   ->  gpurun_out/traffic_<wl>.json  (copy to profiles/)
#   bash tools/traffic.sh <name> <command ...>   ->  the same for any command (wide70, copy, cfg5 ...: tools/one_kernel.py), gpurun_out/traffic_<name>.json
# The file names the kernel sources it was taken with (sources_sha); bench.py quotes it only while that hash is the build's.
WL=${1:-cfg2}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B="python bench.py --workload $WL --steps 1 --warmup 1 --inner 4 --prime 2 --legs= --no-cpu-baseline"
if [ $# -gt 1 ]; then shift; B="$*"; fi
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d gpurun_out/traffic_${WL}_$c -o p -- $B > gpurun_out/traffic_${WL}_$c.log 2>&1
done
python - $WL <<'PY'
import csv, sys, json, collections, glob, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
wl = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"gpurun_out/traffic_{wl}_{c}/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "etlg::k_" not in k: continue
            k = k.split("etlg::")[1].split("<")[0].split("(")[0]
            if k == "k_cells" and wl.startswith("copy"): k = "k_copy_cells"   # (the table-copy instantiation of k_cells: bench.py names it so)
            if k == "k_plan2": k = "k_plan"   # the library's profiler reports both instantiations of the plan kernel under one name
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
best = os.environ.get("TRAFFIC_KERNEL") or max(tot, key=lambda k: tot[k].get("FETCH_SIZE", 0))   # (TRAFFIC_KERNEL=k_bounds_local: a kernel that is not the largest reader of the command)
fetch_kb = tot[best]["FETCH_SIZE"] / n[(best, "FETCH_SIZE")]
write_kb = tot[best]["WRITE_SIZE"] / n[(best, "WRITE_SIZE")]
import bench
out = {"workload": wl, "kernel": best, "batch_mib": 64, "launches": n[(best, "FETCH_SIZE")], "sources_sha": bench.kernel_sources_sha(),
       "FETCH_SIZE_kb_per_launch": round(fetch_kb, 1), "WRITE_SIZE_kb_per_launch": round(write_kb, 1),
       "correction": "FETCH_SIZE x2 (gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported",
       "hbm_bytes_per_launch": int((2 * fetch_kb + write_kb) * 1024),
       "all_kernels_hbm_bytes_per_launch": {k: int((2 * tot[k].get("FETCH_SIZE", 0) / max(1, n[(k, "FETCH_SIZE")]) + tot[k].get("WRITE_SIZE", 0) / max(1, n[(k, "WRITE_SIZE")])) * 1024) for k in tot}}
json.dump(out, open(f"gpurun_out/traffic_{wl}.json", "w"), indent=1)
print(out)
PY
