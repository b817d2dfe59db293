#!/bin/bash
# fractal_teapots at 8 bounces: where the time goes (kernel stats + phase table of the -DRPT_PROF build)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r03o; mkdir -p $O
export TMPDIR=/tmp
ARGS="--scene fractal_teapots --bounces 8 --spp 8 --steps 2 --warmup 1 --no-cpu-baseline --no-live-pmc"
timeout 300 python bench.py $ARGS 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print(round(d['value'], 1), d['ms_per_step'], d['config']['workload'])"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_tp -o tp -- python $OLDPWD/bench.py $ARGS > /dev/null 2>&1)
python - <<'PY' > gpurun_out/r03o/kernel_stats.txt
import glob, csv
for f in glob.glob('/tmp/prof_tp/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:14]:
        print("%-90s calls %6s total_ms %9.2f avg_us %9.1f %5s%%" % (r['Name'][:90], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3, r['Percentage']))
PY
cat $O/kernel_stats.txt
P=$PWD/rpt_amd/lib/librptgpu_prof.so
RPTGPU_LIB=$P RPTGPU_PRINT_PHASES=1 timeout 300 python bench.py --scene fractal_teapots --bounces 8 --steps 1 --warmup 0 --spp 4 --no-cpu-baseline --no-live-pmc 2>&1 >/dev/null | grep "^prof" > $O/phase_teapots.txt
cat $O/phase_teapots.txt
