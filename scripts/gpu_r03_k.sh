#!/bin/bash
# object filter of flat scenes: parity of the flat-scene tests, then room23 / cornell with the filter on and off
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r03k; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "flat or polygon or random or edge or tie" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for rep in 1 2; do
for sc in "room23 64" "cornell 128"; do
  set -- $sc
  for f in 8 0; do
    RPTGPU_OBJECT_FILTER_MIN=$f timeout 300 python bench.py --scene $1 --spp $2 --steps 3 --warmup 1 --no-cpu-baseline --no-live-pmc 2>$O/err_$1_$f.log | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$1 filter_min=$f', round(d['value'], 1), d['ms_per_step'])"
  done
done
done
RPTGPU_PRINT_LAUNCH=1 timeout 120 python bench.py --scene room23 --spp 8 --steps 1 --warmup 0 --no-cpu-baseline --no-live-pmc 2>&1 | grep rpt_paths | head -2
