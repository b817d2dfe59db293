#!/bin/bash
# final evidence of round 2: default bench line + rocprofv3 trace / PMC of C2 at its own spp, the same for the C3 stand-in,
# bench lines of every BASELINE config at its real frame size with CPU baselines
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
O=gpurun_out/r02final; mkdir -p $O
# the rocpd databases are summarised here, on the box, and deleted: only the summaries travel back (64 MiB limit)
export RPT_PROFILE_DST=$REPO/$O/profiles
mkdir -p $RPT_PROFILE_DST
prof() { bash scripts/profile.sh r02 $1 $2 $3 > $O/profile_$1.log 2>&1; python scripts/summarize_profile.py r02 $1 > $O/summary_$1.txt 2>&1; rm -rf gpurun_out/prof_r02_$1; }
prof cornell 0 512
prof dragon 64 16
prof wine_glass 16 4
prof fractal_spheres 16 4
for cfg in "sphere 100" "dragon 256" "fractal_spheres 64" "glass 64" "wine_glass 64" "room23 128"; do
  set -- $cfg
  timeout 400 python bench.py --scene $1 --spp $2 --steps 2 2>$O/bench_$1.err | tail -1 >> $O/other_configs.jsonl
done
python - <<'PY'
import json
for l in open('gpurun_out/r02final/other_configs.jsonl'):
    try:
        d = json.loads(l); c = d['cpu_baseline']
        print('%-62s %8.1f Msamples/s  cpu %.2f (%d threads)  create %.0f ms' % (d['config']['workload'], d['value'], c['value'], c['cores'], d['config']['scene_create_ms']))
    except Exception as e: print('ERR', e)
PY
