#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r03p; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "deep_nest or nested or teapots or nesting or fixture or golden or metal or teapot" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
run() { # name spp env...
  local sc=$1 spp=$2; shift 2
  env "$@" timeout 300 python bench.py --scene $sc --bounces 8 --spp $spp --steps 2 --warmup 1 --no-cpu-baseline --no-live-pmc 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$sc $*', round(d['value'], 1), d['ms_per_step'], d['config']['workload'])"
}
for rep in 1 2; do
run fractal_teapots 8 RPTGPU_NEST_LOOP=1
run fractal_teapots 8 RPTGPU_NEST_LOOP=0
done
run fractal_teapots 32 RPTGPU_NEST_LOOP=1
run fractal_teapots 32 RPTGPU_NEST_LOOP=0
