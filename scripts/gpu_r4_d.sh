#!/bin/bash
# round 4: deferred exact tests in rpt_tree_trace — parity (full suite), A/B of the variants on C3 and C5, phase tables
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r4d; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > $O/pytest.txt
cat $O/pytest.txt
REP=1 bash scripts/gpu_ab.sh r4d "dragon:32 wine_glass:16" "d0l7 d0l6 base d1 d2w6 d2w24 d3l5"
REP=1 bash scripts/gpu_ab.sh r4d "dragon:32 wine_glass:16" "d0l7 base"
for sc in dragon:16 wine_glass:4; do
  scene=${sc%%:*}; spp=${sc##*:}
  RPTGPU_LIB=$PWD/rpt_amd/lib/librptgpu_prof.so RPTGPU_PRINT_PHASES=1 timeout 300 python bench.py --scene $scene --steps 1 --warmup 0 --spp $spp --no-cpu-baseline --no-live-pmc > /dev/null 2> $O/prof_$scene.txt
  echo "## $scene $spp spp"; grep "prof\[" $O/prof_$scene.txt | grep tree_trace
done
