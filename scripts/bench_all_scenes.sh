#!/bin/bash
# every example scene of rpt_amd/scenes.py at its own frame size, a few samples per pixel: one line each (a survey for
# pathologically slow cases, not a benchmark)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for sc in sphere cornell dragon fractal_spheres glass wine_glass fractal_teapots basic monomial_glass spheres compound teapot cylinder rustacean pegasus metal room23; do
  timeout 300 python bench.py --scene $sc --spp ${1:-16} --steps 2 --warmup 1 --no-cpu-baseline --no-live-pmc 2>/dev/null | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.readline()); c = d['config']
    print('%-16s %9.1f Msamples/s  %8.1f ms/step  %s  [%s]' % ('$sc', d['value'], d['ms_per_step'], c['workload'], c['pipeline']))
except Exception as e: print('$sc FAILED', e)"
done
