#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for T in "$@"; do
  export RT355_LIB=$ROOT/variants/librt355_$T.so
  M=$(timeout -k 10 300 python $ROOT/tools/trace_bench.py synthetic 24 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["primary"]["Mrays_s"], d["bounce1"]["Mrays_s"], d["bounce1_any_hit"]["Mrays_s"])') || exit 1
  R=$(timeout -k 10 300 python $ROOT/tools/render_configs.py C5 --spp-scale 0.125 --check 0 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["Mrays_s"], d["ms_per_frame"])') || exit 1
  echo "$T: trace primary,bounce1,any = $M ; render Mrays/s,ms = $R"
done
