#!/bin/bash
# first GPU call of round 2: tests, host probe, bench lines of every BASELINE config at its real frame size, dragon profile
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
mkdir -p gpurun_out/r02a
O=gpurun_out/r02a
( nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; grep -m1 "model name" /proc/cpuinfo; python -c "import os; print(len(os.sched_getaffinity(0)), os.cpu_count())"; rocm-smi --showmeminfo vram 2>/dev/null | head -8 ) > $O/host.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
python - > $O/oracle_threads.txt 2>&1 <<'PY'
import sys, time
sys.path.insert(0, '.')
from oracle import oracle_ffi as O
from rpt_amd import scenes, make_params
s, c, cfg = scenes.cornell()
L, how = O.baseline_lib(True)
print(how)
osc = O.OracleScene(s, L)
for th in (8, 32, 64, 128, 256):
    p = make_params(1920, 1080, 8, 2)
    t = time.time(); osc.render(c, p, threads=th); dt = time.time() - t
    print(th, "threads: %.2f Msamples/s" % (1920 * 1080 * 2 / dt / 1e6))
PY
cat $O/oracle_threads.txt
timeout 600 python bench.py > $O/bench_cornell.json 2> $O/bench_cornell.err; tail -c 600 $O/bench_cornell.json
timeout 300 python bench.py --scene dragon --spp 16 --steps 2 > $O/bench_dragon.json 2> $O/bench_dragon.err
timeout 300 python bench.py --scene wine_glass --spp 4 --steps 2 > $O/bench_wine_glass.json 2> $O/bench_wine_glass.err
timeout 300 python bench.py --scene fractal_spheres --spp 8 --steps 2 > $O/bench_fractal_spheres.json 2> $O/bench_fractal_spheres.err
timeout 300 python bench.py --scene glass --spp 32 --steps 2 > $O/bench_glass.json 2> $O/bench_glass.err
timeout 300 python bench.py --scene room23 --spp 64 --steps 2 > $O/bench_room23.json 2> $O/bench_room23.err
for f in $O/bench_*.json; do python -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']; c=d['cpu_baseline']
    print('%-60s %8.1f Ms/s  %s frac=%.3f  cpu=%s' % (d['config']['workload'][:60], d['value'], r['kernel'], r['frac'] or 0, ('%.2f'%c['value']) if c else None))
except Exception as e: print('$f', 'ERR', e)
"; done
timeout 600 bash scripts/profile.sh r02 dragon 8 4 > $O/profile_dragon.log 2>&1
tail -3 $O/profile_dragon.log
