#!/bin/bash
# the evidence of a round (run through gpurun from the repo root):  TAG=r05 bash scripts/gpu_round_final.sh ["scene:trace_spp:pmc_spp ..."]
#   1. the GPU test suite
#   2. the DEFAULT bench command (what the driver runs): headline C2 + the other BASELINE configs + live counters
#   3. rocprofv3 --kernel-trace --stats of the bench command per scene + PMC passes, summarised ON the box (the rocpd
#      databases do not travel: 64 MiB limit) into gpurun_out/r03final/profiles/
#   4. per-phase lane-occupancy tables from the -DRPT_PROF build (rpt_amd/lib/librptgpu_prof.so, if present)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
TAG=${TAG:-r06}
O=gpurun_out/${TAG}final; mkdir -p $O
export TMPDIR=/tmp
export RPT_PROFILE_DST=$REPO/$O/profiles
mkdir -p $RPT_PROFILE_DST
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -2
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
# (trace spp of the two configs BASELINE assigns to 8 GPUs = 256, bench.py's MULTI_GPU_OTHER_CONFIGS: their <scene>_bench_line.json is the 1-GPU line of the same
# workload that an N > 1 bench line carries along as n1_reference)
LIST=${1:-"cornell:0:512 dragon:32:16 wine_glass:256:4 fractal_spheres:256:4 room23:64:64 glass:64:64"}
for item in $LIST; do
  IFS=: read sc tspp pspp <<< "$item"
  bash scripts/profile.sh $TAG $sc $tspp $pspp "--no-live-pmc" > $O/profile_$sc.log 2>&1
  python scripts/summarize_profile.py $TAG $sc > $O/summary_$sc.txt 2>&1
  rm -rf gpurun_out/prof_${TAG}_$sc
done
P=$PWD/rpt_amd/lib/librptgpu_prof.so
rm -f $O/phase_tables.txt
if [ -f $P ]; then
  for sc in "cornell 64" "room23 32" "glass 16" "dragon 16" "wine_glass 4" "fractal_spheres 4" "fractal_teapots 16 --bounces 8"; do
    set -- $sc
    echo "## $1, $2 spp, 1 step (bench.py --scene $1 --steps 1 --warmup 0 --spp $2 $3 $4, librptgpu_prof.so = -DRPT_PROF build)" >> $O/phase_tables.txt
    RPTGPU_LIB=$P RPTGPU_PRINT_PHASES=1 timeout 300 python bench.py --scene $1 --steps 1 --warmup 0 --spp $2 $3 $4 --no-cpu-baseline --no-live-pmc 2>&1 >/dev/null | grep "^prof" >> $O/phase_tables.txt
  done
fi
# 5. the full-size parity sweep on THIS build (two seeds, 1/8 of the tiles each), headed by the commit it ran on
( echo "# scripts/parity_sweep.py on commit ${COMMIT:-unknown} (the tree gpurun sent), $(date -u +%Y-%m-%dT%H:%MZ)"
  timeout 1500 python scripts/parity_sweep.py 16 8 3 0xABCDE
  timeout 1500 python scripts/parity_sweep.py 16 8 6 0x5EED5 ) > $O/parity_sweep.txt 2>&1
tail -2 $O/parity_sweep.txt
timeout 300 python bench.py --scene simple_video > $O/simple_video.json 2>/dev/null
timeout 300 python bench.py --scene fractal_teapots --bounces 8 --spp 64 --steps 2 --warmup 1 --no-live-pmc > $O/fractal_teapots_b8.json 2>/dev/null
TAG=$TAG python - <<'PY'
import json
import os
d=json.load(open("gpurun_out/%sfinal/bench_default.json" % os.environ.get("TAG", "r05")))
r=d["roofline"]
print("C2 %.1f Msamples/s  %.1f ms/step  frac %.3f (valu_busy %.3f x lanes %.1f/64)  hbm_frac %s  acc_frac %.2f  cpu %.2f  src: %s" % (d["value"], d["ms_per_step"], r.get("frac") or 0, r.get("valu_busy") or 0, r.get("lanes_active") or 0, r.get("hbm_frac"), r.get("accounting_frac") or 0, (d.get("cpu_baseline") or {}).get("value",0), (r.get("pmc_source") or "")[:40]))
for o in d.get("other_configs",[]):
    rr=o.get("roofline",{})
    print("%-60s %8.1f  kernel %s frac %s lanes %s cpu %s  %.1f s" % (o.get("workload"), o.get("value",0), rr.get("kernel"), rr.get("frac"), rr.get("lanes_active"), (o.get("cpu_baseline") or {}).get("value"), o.get("wall_s_of_this_entry",0)))
PY
