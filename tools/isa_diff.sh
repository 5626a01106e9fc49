# Is the gfx950 code of the working tree the same as that of a commit? Compiles every kernel source of both to
# assembly (device only) and diffs them, ignoring comments and the per-compilation unit id symbol. Used to prove
# that a refactor (macros, test hooks) is a no-op for the GPU before spending GPU minutes on it.
# usage: bash tools/isa_diff.sh [commit=HEAD]
REV=${1:-HEAD}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
mkdir -p $T/new $T/old $T/src
git -C $ROOT archive $REV etl_amd/csrc include | tar -x -C $T/src
for f in fused cells copy kernels scan; do
  O=-O3; [ $f = fused ] && O="-Os $(cd $ROOT && python -c 'from etl_amd.build import DEFS; print(" ".join(DEFS.get("fused.hip", [])))')"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 $O -std=c++17 --cuda-device-only -S $ROOT/etl_amd/csrc/$f.hip -o $T/new/$f.s 2>/dev/null &
  /opt/rocm/bin/hipcc --offload-arch=gfx950 $O -std=c++17 --cuda-device-only -S $T/src/etl_amd/csrc/$f.hip -o $T/old/$f.s 2>/dev/null &
  wait
  sed 's/;.*//' $T/new/$f.s | grep -v '^\s*$' | grep -v __hip_cuid > $T/n.txt
  sed 's/;.*//' $T/old/$f.s | grep -v '^\s*$' | grep -v __hip_cuid > $T/o.txt
  echo "$f.hip: $(diff $T/o.txt $T/n.txt | grep -c '^[<>]') differing lines of $(wc -l < $T/n.txt)"
done
rm -rf $T
