#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for T in "$@"; do
  export RT355_LIB=$ROOT/variants/librt355_$T.so
  B=$(timeout -k 10 300 python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])') || exit 1
  R=$(timeout -k 10 300 python $ROOT/tools/render_configs.py C1 C2 C4 --check 4 2>/dev/null | python -c '
import sys,json
print(" ".join("%s %.1f(%d)" % (d["config"], d["Mrays_s"], d["oracle_pixels_different"]) for d in map(json.loads, sys.stdin)))') || exit 1
  R5=$(timeout -k 10 300 python $ROOT/tools/render_configs.py C5 --spp-scale 0.125 --check 0 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["Mrays_s"])') || exit 1
  echo "$T: bench = $B ; $R ; C5/8 $R5"
done
