#!/bin/bash
# Hardware-counter passes for the bench workload (run on the GPU box through gpurun).
# One rocprofv3 process per pass; counters never combined with tracing options.
# usage: tools/collect_pmc.sh <outdir> [script.py args...]   (default: bench.py at 8 spp)
set -u
OUT=${1:-gpurun_out/pmc}; shift || true
ARGS=${@:-bench.py --spp 8 --steps 1 --warmup 0 --no-cpu-baseline}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
PASSES=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU"
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS"
 "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum TCP_PENDING_STALL_CYCLES_sum"
 "FETCH_SIZE GRBM_GUI_ACTIVE"
 "WRITE_SIZE TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"
)
i=0
for P in "${PASSES[@]}"; do
  i=$((i+1))
  timeout -s KILL 170 rocprofv3 --pmc $P --output-format csv -d "$ROOT/$OUT/pass$i" -- python3 $ROOT/$ARGS > "$ROOT/$OUT/pass$i.log" 2>&1
  rc=$?
  echo "pass $i rc=$rc"
  if [ $rc -ne 0 ]; then tail -5 "$ROOT/$OUT/pass$i.log"; exit $rc; fi
done
