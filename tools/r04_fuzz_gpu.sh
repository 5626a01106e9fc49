#!/bin/bash
# the mutation fuzzers against the REAL library on the MI355X (they normally run on the SIMT emulator): WAL / table-copy / boundary-scan
# batches on every kernel path incl. the plan behind its sidecar pre-pass, then ASYNC chains. usage: tools/r04_fuzz_gpu.sh [seconds each]
T=${1:-70}
mkdir -p gpurun_out
{
ETLG_SIMT_FUZZ_LIB=$PWD/etl_amd/libetl_gfx950.so ETLG_FUZZ_PROCS=6 python tools/simt_fuzz.py $T 61001 2>&1 | grep -v amdgpu.ids | tail -5
python tools/async_fuzz.py $T 71001 2>&1 | grep -v amdgpu.ids | tail -3
} | tee gpurun_out/r04as_gpu_fuzz_prepass.txt
