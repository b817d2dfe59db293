#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
O=gpurun_out/r02d
mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
for lb in 0 1; do
  for cfg in "dragon 16" "wine_glass 4"; do
    set -- $cfg
    RPTGPU_LEAF_BOXES=$lb timeout 300 python bench.py --scene $1 --spp $2 --steps 2 --no-cpu-baseline > $O/bench_$1_lb$lb.json 2> $O/bench_$1_lb$lb.err
    python -c "
import json
d=json.loads(open('$O/bench_$1_lb$lb.json').read().strip().splitlines()[-1]); k=d['roofline']['kernels']
print('leaf_boxes=$lb %-12s %7.1f Msamples/s' % ('$1', d['value']), {n:round(v['total_ms'],1) for n,v in k.items()})"
  done
done 2>&1 | tee $O/ab.txt
