#!/bin/bash
# Evidence for the bench line, run on the GPU box through gpurun:
#   1. rocprofv3 --kernel-trace --stats of the default bench command (no CPU leg)
#   2. separate --pmc passes of the same workload (full 64 spp, one frame)
# usage: tools/profile_round.sh <outdir under gpurun_out> [C3|C5]   (C5: the large synthetic scene, 4K x 128 spp)
set -u
OUT=${1:-gpurun_out/profile}
CFG=${2:-C3}
if [ "$CFG" = C5 ]; then SPP=128; CUR=pmc_current_c5.json; LIMIT=420; RAW=--raw-fetch; KERNELS="k_trace<false;k_shade<"; else SPP=64; CUR=pmc_current.json; LIMIT=240; RAW=; KERNELS="k_bounce<false;k_bounce_loop<false"; fi
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL $LIMIT rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/trace" -- \
  python3 $ROOT/bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline > "$ROOT/$OUT/trace.log" 2>&1
rc=$?; echo "trace rc=$rc"; [ $rc -ne 0 ] && { tail -5 "$ROOT/$OUT/trace.log"; exit $rc; }
PASSES=(
 "FETCH_SIZE GRBM_GUI_ACTIVE"
 "WRITE_SIZE TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU"
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum TCP_PENDING_STALL_CYCLES_sum"
)
i=0
for P in "${PASSES[@]}"; do
  i=$((i+1))
  timeout -s KILL $LIMIT rocprofv3 --pmc $P --output-format csv -d "$ROOT/$OUT/pass$i" -- \
    python3 $ROOT/bench.py --config $CFG --steps 1 --warmup 0 --no-cpu-baseline > "$ROOT/$OUT/pass$i.log" 2>&1
  rc=$?; echo "pass $i rc=$rc"
  if [ $rc -ne 0 ]; then tail -5 "$ROOT/$OUT/pass$i.log"; exit $rc; fi
done
python3 $ROOT/tools/summarize_pmc.py "$ROOT/$OUT" "$KERNELS" --current "$ROOT/$OUT/$CUR" --spp $SPP $RAW > "$ROOT/$OUT/pmc_summary.json"
cp "$ROOT/$OUT"/trace/*/*kernel_stats.csv "$ROOT/$OUT/kernel_stats.csv" 2>/dev/null || true
echo "summaries: $OUT/pmc_summary.json $OUT/$CUR $OUT/kernel_stats.csv"
