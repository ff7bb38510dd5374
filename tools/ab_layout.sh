#!/bin/bash
# A/B of record layouts (RT355_NODE_CLUSTER = size[,align]) on the large synthetic scene: the trace
# micro-benchmark and a quarter of the C5 frame.  usage: tools/ab_layout.sh "1" "8" "7,8" ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for T in "$@"; do
  export RT355_NODE_CLUSTER=$T
  M=$(timeout -k 10 300 python $ROOT/tools/trace_bench.py synthetic 24 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["primary"]["Mrays_s"], d["bounce1"]["Mrays_s"], d["bounce1_any_hit"]["Mrays_s"])') || exit 1
  F=$(timeout -k 10 300 python $ROOT/tools/render_configs.py C5 --check 0 --spp-scale 0.25 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_frame"], d["Mrays_s"])') || exit 1
  echo "cluster $T: trace primary, bounce1, bounce1 any-hit = $M ; quarter frame ms, Mrays/s = $F"
done
