#!/bin/bash
# rocprofv3 evidence for bench.py on the GPU box.  Usage (through gpurun, from the repo root):
#   bash scripts/profile.sh r01            -> gpurun_out/prof_r01/...
# Pass 1: kernel trace + stats of the bench command.  Passes 2-4: PMC counters, one pass each
# (FETCH_SIZE and WRITE_SIZE do not fit one pass; never combined with sys/hip/hsa tracing).
set -u
TAG=${1:-r01}
SPP=${2:-32}
EXTRA="${3:-}"
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# pass 1: the bench command itself.  With no extra args this is the DEFAULT bench line
# (`python bench.py`: C2 at 512 spp, 3 steps + 1 warmup), so the kernel's average duration in the
# stats equals the one bench.py reports; SPP > 0 profiles a shorter variant instead.
if [ "$SPP" = "0" ]; then CMD="python $REPO/bench.py $EXTRA"; else CMD="python $REPO/bench.py --steps 2 --warmup 1 --spp $SPP --no-cpu-baseline $EXTRA"; fi
echo "$CMD" > $OUT/trace_command.txt
rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- $CMD > $OUT/bench_trace.log 2>&1
PCMD="python $REPO/bench.py --steps 1 --warmup 0 --spp 4 --no-cpu-baseline $EXTRA"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o bench -- $PCMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o bench -- $PCMD > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $OUT/pmc_sq -o bench -- $PCMD > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $OUT/pmc_tcc -o bench -- $PCMD > $OUT/pmc_tcc.log 2>&1
find $OUT -name '*.csv' | head -40
tail -2 $OUT/bench_trace.log
