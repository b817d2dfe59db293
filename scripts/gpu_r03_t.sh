#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
run() { # scene spp env...
  local sc=$1 spp=$2; shift 2
  env "$@" timeout 300 python bench.py --scene $sc --bounces 8 --spp $spp --steps 2 --warmup 1 --no-cpu-baseline --no-live-pmc 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$sc $*', round(d['value'], 1), d['ms_per_step'])"
}
run fractal_teapots 64 A=1
run fractal_teapots 64 RPTGPU_DEEP_DEPTH=1
run fractal_teapots 64 RPTGPU_DEEP_DEPTH=1 RPTGPU_SORT_RAYS=1
run fractal_teapots 64 RPTGPU_DEEP_DEPTH=5
run fractal_spheres 8 A=1
run fractal_spheres 8 RPTGPU_DEEP_DEPTH=5
run fractal_spheres 8 RPTGPU_DEEP_DEPTH=5 RPTGPU_SORT_RAYS=1
