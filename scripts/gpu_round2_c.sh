#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
O=gpurun_out/r02c
mkdir -p $O
timeout 120 scripts/ubench/gather > $O/gather.txt 2>&1
cat $O/gather.txt
for lib in librptgpu librptgpu_tt2 librptgpu_tt2l12; do
  RPTGPU_MESH_PAIRS=0 RPTGPU_LIB=$PWD/rpt_amd/lib/$lib.so timeout 200 python bench.py --scene dragon --spp 8 --steps 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']
print('$lib %7.1f Msamples/s' % d['value'], {n:round(v['total_ms'],1) for n,v in k.items()})"
done 2>&1 | tee $O/variants.txt
