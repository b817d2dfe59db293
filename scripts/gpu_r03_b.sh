#!/bin/bash
# round 3, call b: rpt_tree_walk (wave-scheduled traversal) — parity, A/B against rpt_tree_trace, threshold sweep
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03c; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
run() { # scene spp label env...
  sc=$1; spp=$2; lab=$3; shift 3
  env "$@" timeout 300 python bench.py --scene $sc --steps 2 --warmup 1 --spp $spp --no-cpu-baseline --no-live-pmc 2>$O/err_$sc_$lab.txt | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']
print('$sc','$lab',round(d['value'],1),'tt',round(k.get('rpt_tree_trace',{}).get('total_ms',0),1),'enter',round(k.get('rpt_tree_enter+sort',{}).get('total_ms',0),1),'shade',round(k.get('rpt_shade',{}).get('total_ms',0),1))" | tee -a $O/ab.txt
}
for sc in "dragon 32" "wine_glass 8"; do
  set -- $sc
  run $1 $2 old RPTGPU_TREE_WALK=0
  for th in 16,8,16 20,12,16 12,8,16 12,4,24 8,1,56 16,8,32 24,8,16 16,16,16; do
    run $1 $2 walk_$th RPTGPU_WALK_TH=$th
  done
done
P=$PWD/rpt_amd/lib/librptgpu_prof.so
for sc in "dragon 16" "wine_glass 4"; do
  set -- $sc
  RPTGPU_LIB=$P RPTGPU_PRINT_PHASES=1 timeout 300 python bench.py --scene $1 --steps 1 --warmup 0 --spp $2 --no-cpu-baseline --no-live-pmc > $O/prof_$1.json 2> $O/prof_$1.txt
  grep -h "^prof" $O/prof_$1.txt
done
