# First GPU call of a round: parity, then timing, of the prepared k_fused variants (tools/build_variants.py), one box.
#   here (no GPU):  python tools/build_variants.py product plain hotfix stage8 early_stage8 early_hotfix fixed_hotfix_scols all
#   GPU box:        gpurun --timeout 1500 -- 'bash tools/ab_variants.sh "product plain hotfix stage8 early_stage8 early_hotfix fixed_hotfix_scols all"'
# For every variant: the parity files that exercise what the flags touch (fixed-width plan cases, cfg2 mutation fuzz, MiB-scale
# cfg2 / cfg3 batches on every kernel path) through ETLG_LIB_PATH, then bench.py on cfg2 and cfg3 (kernel average from the library's HIP
# events). A variant whose parity run fails is not timed. Output: gpurun_out/ab/<variant>.{parity.log,json} and one table.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
for v in ${1:-product plain hotfix stage8 early_stage8 early_hotfix fixed_hotfix_scols all}; do
  lib=$GRAFT_REPO_ROOT/etl_amd/variants/libetl_gfx950_$v.so
  [ -f $lib ] || { echo "$v: not built"; continue; }
  expect=0; case $v in fixed*|all|product) expect=1;; esac
  ETLG_LIB_PATH=$lib ETLG_EXPECT_FIXED_TILE=$expect timeout 400 python -m pytest -q -x -m gpu -p no:cacheprovider \
    tests/test_gpu_fixed_plan.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py \
    -k "fixed_plan or (mutations and cfg2) or large_batch or full_size" --deselect tests/test_gpu_parity.py::test_native_library_is_the_one_in_tree \
    > gpurun_out/ab/$v.parity.log 2>&1
  rc=$?
  echo "$v parity rc=$rc  $(tail -1 gpurun_out/ab/$v.parity.log)"
  [ $rc = 0 ] || continue
  for wl in ${2:-cfg2 cfg3}; do
    ETLG_LIB_PATH=$lib timeout 120 python bench.py --workload $wl --steps 40 --warmup 5 --pool 4 --no-cpu-baseline --no-scan-leg \
      > gpurun_out/ab/${v}_$wl.json 2> gpurun_out/ab/${v}_$wl.err
    python - $v $wl <<'PY'
import json, sys
v, wl = sys.argv[1:3]
try:
    j = json.loads(open(f"gpurun_out/ab/{v}_{wl}.json").read().strip().splitlines()[-1])
    r = j["roofline"]
    print(f"{v:20s} {wl} value {j['value']:8.1f} GB/s  ms/step {j['ms_per_step']:.4f}  {r['kernel']} {r['kernel_avg_us']:.1f} us  frac {r['frac']}")
except Exception as e:
    print(v, wl, "FAILED", e)
PY
  done
done
# tile size: 128 frames per tile (variant all_blk128, emulator parity only so far: every scenario, cfg2 fuzz, 2 MiB cfg2 / cfg5)
lib=$GRAFT_REPO_ROOT/etl_amd/variants/libetl_gfx950_all_blk128.so
if [ -f $lib ]; then
  ETLG_FUSED_BLK=128 ETLG_LIB_PATH=$lib timeout 120 python bench.py --workload cfg2 --steps 40 --warmup 5 --pool 4 --no-cpu-baseline --no-scan-leg \
    > gpurun_out/ab/all_blk128_cfg2.json 2> gpurun_out/ab/all_blk128_cfg2.err
  python - <<'PY'
import json
try:
    j = json.loads(open("gpurun_out/ab/all_blk128_cfg2.json").read().strip().splitlines()[-1])
    r = j["roofline"]
    print(f"all_blk128, 128-frame tiles cfg2 value {j['value']:8.1f} GB/s  {r['kernel']} {r['kernel_avg_us']:.1f} us  frac {r['frac']}")
except Exception as e:
    print("all_blk128 FAILED", e)
PY
fi
# tile size: k_fused with 64 frames per tile on cfg2 (the product instantiates 256 and 64; 512 was slower than 256)
lib=$GRAFT_REPO_ROOT/etl_amd/variants/libetl_gfx950_product.so
if [ -f $lib ]; then
  ETLG_FUSED_KERNEL=1 ETLG_LIB_PATH=$lib timeout 120 python bench.py --workload cfg2 --steps 40 --warmup 5 --pool 4 --no-cpu-baseline --no-scan-leg \
    > gpurun_out/ab/product_blk64_cfg2.json 2> gpurun_out/ab/product_blk64_cfg2.err
  python - <<'PY'
import json
try:
    j = json.loads(open("gpurun_out/ab/product_blk64_cfg2.json").read().strip().splitlines()[-1])
    r = j["roofline"]
    print(f"product, 64-frame tiles cfg2 value {j['value']:8.1f} GB/s  {r['kernel']} {r['kernel_avg_us']:.1f} us  frac {r['frac']}")
except Exception as e:
    print("product blk64 FAILED", e)
PY
fi
