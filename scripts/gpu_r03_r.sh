#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
run() { # name spp env...
  local sc=$1 spp=$2; shift 2
  env "$@" timeout 300 python bench.py --scene $sc --bounces 8 --spp $spp --steps 2 --warmup 1 --no-cpu-baseline --no-live-pmc 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$sc $*', round(d['value'], 1), d['ms_per_step'], d['config']['workload'])"
}
run fractal_teapots 128 RPTGPU_NEST_LOOP=1
run fractal_teapots 128 RPTGPU_NEST_LOOP=0
run fractal_teapots 256 RPTGPU_NEST_LOOP=1
run fractal_teapots 256 RPTGPU_NEST_LOOP=0
export TMPDIR=/tmp
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_tp -o tp -- python $GRAFT_REPO_ROOT/bench.py --scene fractal_teapots --bounces 8 --spp 128 --steps 2 --warmup 1 --no-cpu-baseline --no-live-pmc > /dev/null 2>&1
find /tmp/prof_tp -type f | head -20
python - <<'PY'
import glob, sqlite3
for f in glob.glob('/tmp/prof_tp/**/*.db', recursive=True):
    c = sqlite3.connect(f)
    try:
        for r in c.execute("select name, total_calls, total_duration, average, percentage from top_kernels order by total_duration desc limit 14"):
            print("%-80s calls %6d total_ms %10.2f avg_us %10.1f %6.2f%%" % (r[0][:80], r[1], r[2]/1e6, r[3]/1e3, r[4]))
    except Exception as e:
        print(f, e); print([r[0] for r in c.execute("select name from sqlite_master")][:40])
PY
