#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r03q2; mkdir -p $O
P=$PWD/rpt_amd/lib/librptgpu_prof.so
for f in 1 0; do
echo "## NEST_LOOP=$f"
RPTGPU_NEST_LOOP=$f RPTGPU_LIB=$P RPTGPU_PRINT_PHASES=1 timeout 300 python bench.py --scene fractal_teapots --bounces 8 --steps 1 --warmup 0 --spp 4 --no-cpu-baseline --no-live-pmc 2>&1 >/dev/null | grep "^prof"
done
export TMPDIR=/tmp
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_tp -o tp -- python $GRAFT_REPO_ROOT/bench.py --scene fractal_teapots --bounces 8 --spp 8 --steps 2 --warmup 1 --no-cpu-baseline --no-live-pmc > /dev/null 2>&1
find /tmp/prof_tp -name "*stats*" | head; f=$(find /tmp/prof_tp -name "*kernel_stats.csv" | head -1); head -12 "$f" | cut -c1-200
