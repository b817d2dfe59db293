#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r03u; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -x --durations=12 -k "nest or teapot or fixture or golden or scheduling" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -18 $O/pytest.log
run() { # scene spp env...
  local sc=$1 spp=$2; shift 2
  env "$@" timeout 300 python bench.py --scene $sc --bounces 8 --spp $spp --steps 2 --warmup 1 --no-cpu-baseline --no-live-pmc 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$sc $*', round(d['value'], 1), d['ms_per_step'])"
}
run fractal_teapots 64 A=1
run fractal_teapots 64 RPTGPU_NEST_PER_TREE=0

