#!/bin/bash
# A/B of library variants on the GPU box:  bash scripts/gpu_ab.sh <outdir> "<scene spp> ..." "<variant> ..."   (variant "base" = librptgpu.so)
# prints one line per (scene, variant): Msamples/s and the ms of the main kernels; repeats each measurement REP times (default 2)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/$1; mkdir -p $O
export TMPDIR=/tmp
REP=${REP:-2}
for sc in $2; do
  scene=${sc%%:*}; spp=${sc##*:}
  for r in $(seq $REP); do
  for vv in $3; do
    # a variant may carry environment settings: name@KEY=VALUE[,KEY=VALUE...]
    v=${vv%%@*}; E=""; if [ "$vv" != "$v" ]; then E=$(echo "${vv#*@}" | tr ',' ' '); fi
    if [ "$v" = base ]; then L=$PWD/rpt_amd/lib/librptgpu.so; else L=$PWD/rpt_amd/lib/librptgpu_$v.so; fi
    v=$(echo "$vv" | tr '@=,' '___')
    env $E RPTGPU_LIB=$L timeout ${RUN_TIMEOUT:-300} python bench.py --scene $scene --steps 2 --warmup 1 --spp $spp --no-cpu-baseline --no-live-pmc 2>$O/err_${scene}_$v.txt | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']
    print('$scene','$v',round(d['value'],1),' '.join('%s=%.1f'%(n.replace('rpt_',''),k[n]['total_ms']) for n in k if k[n]['total_ms']>=1.0))
except Exception as e: print('$scene','$v','FAILED',e)" | tee -a $O/ab.txt
  done; done
done
