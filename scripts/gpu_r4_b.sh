#!/bin/bash
# round 4: queue-backed Philox stream in rpt_paths — parity suite, A/B against the register stream, queue sizes, phase table
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r4b; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $O/pytest.txt
cat $O/pytest.txt
RPTGPU_PRINT_LAUNCH=1 REP=2 bash scripts/gpu_ab.sh r4b "cornell:512" "noq base"
grep -h "rpt_paths<" $O/err_cornell_base.txt | head -2
for q in 16,8 8,8 8,4 4,4 4,0; do
  echo "queue $q" | tee -a $O/ab.txt
  RPTGPU_RNG_QUEUE=$q REP=1 bash scripts/gpu_ab.sh r4b "cornell:512" "base"
done
RPTGPU_LIB=$PWD/rpt_amd/lib/librptgpu_prof.so RPTGPU_PRINT_PHASES=1 timeout 300 python bench.py --scene cornell --steps 1 --warmup 0 --spp 64 --no-cpu-baseline --no-live-pmc > $O/prof_cornell.json 2> $O/prof_cornell.txt
grep "prof\[" $O/prof_cornell.txt
