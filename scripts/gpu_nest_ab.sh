#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
timeout 900 python -m pytest tests -m gpu -q -x -k "nest or teapot or fixture or golden or scheduling or metal" 2>&1 | tail -3
run() { # scene spp env...
  local sc=$1 spp=$2; shift 2
  env "$@" timeout 300 python bench.py --scene $sc --bounces 8 --spp $spp --steps 2 --warmup 1 --no-cpu-baseline --no-live-pmc 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$sc $*', round(d['value'], 1), d['ms_per_step'])"
}
for rep in 1 2; do
run fractal_teapots 64 A=1
run fractal_teapots 64 RPTGPU_NEST_TRACE=0
done
