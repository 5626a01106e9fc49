"""Folds the rocprofv3 --pmc passes of tools/run_r04p.sh into one JSON per kernel (averages per launch)."""
import collections
import csv
import glob
import json
import shutil

for wl, kn in (("cfg2", "k_plan2"), ("cfg3", "k_cells")):
    out = {}
    for d in sorted(glob.glob(f"gpurun_out/r04p/{wl}_p*/")):
        for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
            tot = collections.defaultdict(float)
            n = collections.Counter()
            for r in csv.DictReader(open(f)):
                if kn not in r["Kernel_Name"]:
                    continue
                tot[r["Counter_Name"]] += float(r["Counter_Value"])
                n[r["Counter_Name"]] += 1
            for k in tot:
                out[k] = round(tot[k] / n[k], 2)
        shutil.rmtree(d, ignore_errors=True)
    json.dump(out, open(f"gpurun_out/r04p/{kn}_counters.json", "w"), indent=1)
    print(kn, json.dumps(out))
