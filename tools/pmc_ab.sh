#!/bin/bash
# One SQ counter pass of the bench workload per library variant (variants/librt355_<tag>.so).
# usage: tools/pmc_ab.sh <outdir> tag1 tag2 ...
OUT=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/$OUT
cd /tmp && export TMPDIR=/tmp
for T in "$@"; do
  export RT355_LIB=$ROOT/variants/librt355_$T.so
  timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --output-format csv -d $ROOT/$OUT/$T/pass1 -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $ROOT/$OUT/$T.log 2>&1 || exit 1
  timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_INSTS_BRANCH --output-format csv -d $ROOT/$OUT/$T/pass2 -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline >> $ROOT/$OUT/$T.log 2>&1 || exit 1
  echo "$T done"
done
