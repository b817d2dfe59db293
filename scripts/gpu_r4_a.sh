#!/bin/bash
# round 4, first GPU call: parity suite with the pipelined fold + converged pair draws, A/B of the variants on C2, phase table
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r4a; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $O/pytest.txt
cat $O/pytest.txt
REP=2 bash scripts/gpu_ab.sh r4a "cornell:512" "nopipe pipe nopair base"
RPTGPU_LIB=$PWD/rpt_amd/lib/librptgpu_prof.so RPTGPU_PRINT_PHASES=1 timeout 300 python bench.py --scene cornell --steps 1 --warmup 0 --spp 64 --no-cpu-baseline --no-live-pmc > $O/prof_cornell.json 2> $O/prof_cornell.txt
grep "prof\[" $O/prof_cornell.txt
