#!/bin/bash
# rocprofv3 evidence for bench.py on the GPU box.  Usage (through gpurun, from the repo root):
#   bash scripts/profile.sh <tag> <scene> <trace_spp|0> <pmc_spp> ["extra bench args"]
#   e.g. bash scripts/profile.sh r02 cornell 0 512      -> gpurun_out/prof_r02_cornell/...
# Pass 1: kernel trace + stats of the bench command (trace_spp = 0: the scene's DEFAULT bench line, so the kernel's
# average duration in the stats equals the one bench.py reports).  Passes 2-4: PMC counters, each in its own run with
# --kernel-trace only (never combined with sys/hip/hsa tracing); FETCH_SIZE and WRITE_SIZE do not fit one pass.
set -u
TAG=${1:-r02}
SCENE=${2:-cornell}
SPP=${3:-0}
PSPP=${4:-4}
EXTRA="${5:-}"
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_${TAG}_${SCENE}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ "$SPP" = "0" ]; then CMD="python $REPO/bench.py --scene $SCENE $EXTRA"; else CMD="python $REPO/bench.py --scene $SCENE --steps 2 --warmup 1 --spp $SPP --no-cpu-baseline $EXTRA"; fi
echo "$CMD" > $OUT/trace_command.txt
timeout -k 5 240 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- $CMD > $OUT/bench_trace.log 2>&1
PCMD="python $REPO/bench.py --scene $SCENE --steps 1 --warmup 0 --spp $PSPP --no-cpu-baseline $EXTRA"
echo "$PCMD" > $OUT/pmc_command.txt
timeout -k 5 240 rocprofv3 --pmc FETCH_SIZE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --kernel-trace -d $OUT/pmc_a -o bench -- $PCMD > $OUT/pmc_a.log 2>&1
timeout -k 5 240 rocprofv3 --pmc WRITE_SIZE SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM --kernel-trace -d $OUT/pmc_b -o bench -- $PCMD > $OUT/pmc_b.log 2>&1
timeout -k 5 240 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --kernel-trace -d $OUT/pmc_c -o bench -- $PCMD > $OUT/pmc_c.log 2>&1
find $OUT -name '*.db' | head -10
grep -h '^{' $OUT/bench_trace.log | tail -1 | cut -c1-400
