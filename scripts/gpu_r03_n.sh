#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r03n; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "deep_nest or nested or teapots or nesting or fixture or golden" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
run() { # name spp env...
  local sc=$1 spp=$2; shift 2
  env "$@" timeout 300 python bench.py --scene $sc --spp $spp --steps 2 --warmup 1 --no-cpu-baseline --no-live-pmc 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$sc $*', round(d['value'], 1), d['ms_per_step'], d['config']['workload'])"
}
run fractal_teapots 8 A=1
run fractal_spheres 8 A=1
