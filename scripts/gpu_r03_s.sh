#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
run() { # lib scene spp
  local v=$1 sc=$2 spp=$3
  if [ "$v" = base ]; then L=$PWD/rpt_amd/lib/librptgpu.so; else L=$PWD/rpt_amd/lib/librptgpu_$v.so; fi
  RPTGPU_LIB=$L timeout 300 python bench.py --scene $sc --bounces 8 --spp $spp --steps 2 --warmup 1 --no-cpu-baseline --no-live-pmc 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$sc $v', round(d['value'], 1), d['ms_per_step'])"
}
for v in base wf4; do run $v fractal_teapots 64; done
for v in base wf4; do run $v fractal_teapots 64; done
