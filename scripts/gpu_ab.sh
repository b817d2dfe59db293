#!/bin/bash
# generic A/B: scripts/gpu_ab.sh <outdir> "<ENV=val ENV2=val2|...>" "scene:spp scene:spp" [pytest]
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
O=gpurun_out/$1; mkdir -p $O
if [ "${4:-}" = "pytest" ]; then
  timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
fi
IFS='|' read -ra ENVS <<< "$2"
for e in "${ENVS[@]}"; do
  for cfg in $3; do
    sc=${cfg%%:*}; spp=${cfg##*:}
    tag=$(echo "$e" | tr ' =/' '___')
    env $e timeout 300 python bench.py --scene $sc --spp $spp --steps 2 --no-cpu-baseline > $O/bench_${sc}_$tag.json 2> $O/bench_${sc}_$tag.err
    python -c "
import json
try:
    d=json.loads(open('$O/bench_${sc}_$tag.json').read().strip().splitlines()[-1]); k=d['roofline']['kernels']
    print('%-28s %-16s %7.1f Msamples/s' % ('$e', '$sc', d['value']), {n:round(v['total_ms'],1) for n,v in k.items()})
except Exception as ex: print('$e $sc FAILED', ex)"
  done
done 2>&1 | tee -a $O/ab.txt
