#!/bin/bash
# A/B of library variants (variants/librt355_<tag>.so) on the large synthetic scene: a quarter of the C5 frame
# and the trace micro-benchmark.  usage: tools/ab_c5.sh tag1 tag2 ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for T in "$@"; do
  export RT355_LIB=$ROOT/variants/librt355_$T.so
  F=$(timeout -k 10 300 python $ROOT/tools/render_configs.py C5 --check 0 --spp-scale 0.25 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_frame"], d["Mrays_s"])') || exit 1
  M=$(timeout -k 10 300 python $ROOT/tools/trace_bench.py synthetic 24 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["primary"]["Mrays_s"], d["bounce1"]["Mrays_s"], d["bounce1_any_hit"]["Mrays_s"])') || exit 1
  echo "$T: quarter frame ms, Mrays/s = $F ; trace primary, bounce1, bounce1 any-hit = $M"
done
