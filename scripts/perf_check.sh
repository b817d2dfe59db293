#!/bin/bash
# Standard A/B throughput check (run through gpurun): same workloads every time.
#   bash scripts/perf_check.sh [lib.so ...]     (default: the in-tree library)
LIBS=${@:-rpt_amd/lib/librptgpu.so}
for lib in $LIBS; do
  echo "== $lib"
  for cfg in "cornell 64 2" "dragon 16 1" "wine_glass 16 1" "fractal_spheres 16 1"; do
    set -- $cfg
    RPTGPU_LIB=$PWD/$lib python bench.py --scene $1 --steps $3 --warmup 1 --spp $2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('  %-16s %-22s %8.1f Msamples/s' % ('$1', d['config']['pipeline'], d['value']))"
  done
done
