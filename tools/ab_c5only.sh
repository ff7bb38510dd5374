#!/bin/bash
# A/B of library variants on the large scene at an eighth of its samples.  usage: tools/ab_c5only.sh tag1 tag2 ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for T in "$@"; do
  export RT355_LIB=$ROOT/variants/librt355_$T.so
  R5=$(timeout -k 10 300 python $ROOT/tools/render_configs.py C5 --spp-scale 0.125 --check 0 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["Mrays_s"])') || exit 1
  echo "$T: C5/8 $R5"
done
