#!/bin/bash
# A/B of library variants on the headline bench and on the sharded frame (slowest shard for 1, 2, 4, 8 ranks).
# usage: tools/ab_shards.sh tag1 tag2 ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for T in "$@"; do
  export RT355_LIB=$ROOT/variants/librt355_$T.so
  B=$(timeout -k 10 300 python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])') || exit 1
  S=$(timeout -k 10 300 python $ROOT/tools/shard_times.py 2>/dev/null | awk '{printf "%s ", $5}')
  echo "$T: bench Mrays/s,ms = $B ; slowest shard ms for 1,2,4,8 ranks = $S"
done
