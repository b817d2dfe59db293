#!/bin/bash
# final evidence of round 2: rocprofv3 trace + PMC of the bench command per scene (C2 at its own 512 spp), summarised on
# the box (the rocpd databases do not travel: 64 MiB limit), and bench lines of every BASELINE config at its real frame size
#   bash scripts/gpu_round2_final.sh ["scene:trace_spp:pmc_spp ..."] ["scene spp,scene spp,..."]
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
O=gpurun_out/r02final; mkdir -p $O
export RPT_PROFILE_DST=$REPO/$O/profiles
mkdir -p $RPT_PROFILE_DST
LIST=${1:-"cornell:0:512 dragon:64:16 wine_glass:16:4 fractal_spheres:16:4 sphere:100:100 glass:64:64 room23:128:128"}
for item in $LIST; do
  IFS=: read sc tspp pspp <<< "$item"
  bash scripts/profile.sh r02 $sc $tspp $pspp > $O/profile_$sc.log 2>&1
  python scripts/summarize_profile.py r02 $sc > $O/summary_$sc.txt 2>&1
  cp $RPT_PROFILE_DST/r02_${sc}_pmc.json profiles/ 2>/dev/null   # the bench lines below quote THIS build's counters
  rm -rf gpurun_out/prof_r02_$sc
done
rm -f $O/other_configs.jsonl
IFS=, read -ra CFGS <<< "${2:-sphere 100,dragon 256,fractal_spheres 64,glass 64,wine_glass 64,room23 128}"
for cfg in "${CFGS[@]}"; do
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
