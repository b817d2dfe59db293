#!/bin/bash
# round 4: rpt_tree_generic + unbounded nesting — parity suite (twice: suite time must be stable), nest scenes' throughput
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r4e; mkdir -p $O
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > $O/pytest.txt
cat $O/pytest.txt
( timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 ) > $O/pytest2.txt
cat $O/pytest2.txt
for a in "fractal_teapots 64 --bounces 8" "fractal_teapots 8 --bounces 8"; do
  set -- $a
  timeout 300 python bench.py --scene $1 --steps 2 --warmup 1 --spp $2 $3 $4 --no-cpu-baseline --no-live-pmc 2>$O/err_$1_$2.txt | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']
print('$1 $2spp', round(d['value'],1), ' '.join('%s=%.1f'%(n.replace('rpt_',''),k[n]['total_ms']) for n in k if k[n]['total_ms']>=1.0))" | tee -a $O/bench.txt
done
