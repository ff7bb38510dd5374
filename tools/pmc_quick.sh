#!/bin/bash
# SQ_INSTS_VALU / wave cycles of the bench workload per library variant.
OUT=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/$OUT
cd /tmp && export TMPDIR=/tmp
for T in "$@"; do
  export RT355_LIB=$ROOT/variants/librt355_$T.so
  timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_INSTS_VMEM_RD --output-format csv -d $ROOT/$OUT/$T/pass1 -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $ROOT/$OUT/$T.log 2>&1 || exit 1
  echo "$T done"
done
