#!/bin/bash
# Disassembly + resource notes of one object's gfx950 code: tools/isa.sh cells [outdir]   (measurement tool, not product)
set -e
B=/opt/rocm/lib/llvm/bin
n=$1; out=${2:-/tmp/isa}; mkdir -p $out
$B/llvm-objcopy -O binary --only-section=.hip_fatbin etl_amd/csrc/$n.o $out/$n.fatbin
$B/clang-offload-bundler --unbundle --type=o --input=$out/$n.fatbin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$out/$n.co
$B/llvm-objdump -d --no-show-raw-insn $out/$n.co > $out/$n.s
$B/llvm-readelf --notes $out/$n.co | grep -E "\.name:|\.vgpr_count|\.sgpr_count|private_segment_fixed_size|group_segment_fixed_size|vgpr_spill" | paste - - - - - - | sed 's/ \+/ /g'
