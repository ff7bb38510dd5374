#!/bin/bash
# A/B of prebuilt library variants (variants/librt355_<tag>.so, built by hand with extra -D flags): bench + traversal micro-benchmark each.
# usage: tools/ab.sh tag1 tag2 ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for T in "$@"; do
  export RT355_LIB=$ROOT/variants/librt355_$T.so
  B=$(timeout -k 10 300 python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])') || exit 1
  M=$(timeout -k 10 300 python $ROOT/tools/trace_bench.py week 24 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["primary"]["Mrays_s"], d["bounce1"]["Mrays_s"], d["bounce1_any_hit"]["Mrays_s"])') || exit 1
  echo "$T: bench Mrays/s,ms = $B ; trace primary,bounce1,bounce1_any = $M"
done
