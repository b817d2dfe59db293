#!/bin/bash
# round 3, first GPU call: parity tests, the default bench line (headline + other configs + live counters),
# per-phase lane-occupancy tables from the -DRPT_PROF build
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03a; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -3 $O/pytest.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; tail -c 600 $O/bench_default.err
P=$PWD/rpt_amd/lib/librptgpu_prof.so
for sc in "cornell 64" "room23 32" "dragon 16" "wine_glass 4" "fractal_spheres 4" "fractal_teapots 4"; do
  set -- $sc
  RPTGPU_LIB=$P RPTGPU_PRINT_PHASES=1 timeout 300 python bench.py --scene $1 --steps 1 --warmup 0 --spp $2 --no-cpu-baseline --no-live-pmc > $O/prof_$1.json 2> $O/prof_$1.txt; echo "prof $1 rc=$?"
done
grep -h "^prof" $O/prof_dragon.txt | head -30
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03a/bench_default.json"))
print("C2", d["value"], d["ms_per_step"], {k:d["roofline"].get(k) for k in ("frac","bound","valu_busy","lanes_active","hbm_frac","pmc_source","accounting_frac")})
for o in d.get("other_configs",[]): print(o.get("workload"), o.get("value"), o.get("error"), {k:o.get("roofline",{}).get(k) for k in ("kernel","frac","valu_busy","lanes_active","pmc_source")}, o.get("wall_s_of_this_entry"))
PY
