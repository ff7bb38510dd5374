#!/bin/bash
# A/B of library variants on the large synthetic scene with four-box records in every launch (RT355_QUADS=2)
# and with the library's own choice: an eighth of the C5 frame each.  usage: tools/ab_c5q.sh tag1 tag2 ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for T in "$@"; do
  export RT355_LIB=$ROOT/variants/librt355_$T.so
  A=$(RT355_QUADS=2 timeout -k 10 300 python $ROOT/tools/render_configs.py C5 --check 0 --spp-scale 0.125 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_frame"], d["Mrays_s"])') || exit 1
  B=$(timeout -k 10 300 python $ROOT/tools/render_configs.py C5 --check 0 --spp-scale 0.125 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_frame"], d["Mrays_s"])') || exit 1
  echo "$T: every launch ms, Mrays/s = $A ; from the third bounce = $B"
done
