#!/bin/bash
# A/B of library variants on rt_intersect_dev / rt_occluded_dev (tools/trace_bench.py week 24): camera rays closest / any hit, first bounce closest / any hit (Mrays/s).  usage: tools/ab_trace_bench.sh tag1 tag2 ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for T in "$@"; do
  export RT355_LIB=$ROOT/variants/librt355_$T.so
  R=$(timeout -k 10 200 python $ROOT/tools/trace_bench.py week 24 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["primary"]["Mrays_s"], d["primary_any_hit"]["Mrays_s"], d["bounce1"]["Mrays_s"], d["bounce1_any_hit"]["Mrays_s"])') || exit 1
  echo "$T: $R"
done
