#!/bin/bash
# round 4: gather collective + ABI v5 + bench line — parity suite, then the driver's default bench command
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r4c; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > $O/pytest.txt
cat $O/pytest.txt
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | tail -4
tail -c 600 $O/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4c/bench_default.json').read().strip().splitlines()[-1])
print('C2', round(d['value'],1), 'ms/step', round(d['ms_per_step'],1), 'frac', d['roofline'].get('frac'), 'lanes', d['roofline'].get('lanes_active'), 'cpu', d['cpu_baseline'] and round(d['cpu_baseline']['value'],2))
print('per_rank', d['per_rank'])
print('rays', d['config']['rays_per_s'], d['config']['reference_rays_per_s'])
for o in d.get('other_configs',[]):
    print(o.get('workload'), round(o.get('value',0),1), o.get('error'), 'cpu', o.get('cpu_baseline') and round(o['cpu_baseline']['value'],3), 'wall', round(o.get('wall_s_of_this_entry',0),1))
PY
