#!/bin/bash
# object filter: the whole parity file, then where else the filter might pay (C2 without its plane table, small scenes)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r03l; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
run() { # name spp env...
  local sc=$1 spp=$2; shift 2
  env "$@" timeout 300 python bench.py --scene $sc --spp $spp --steps 3 --warmup 1 --no-cpu-baseline --no-live-pmc 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$sc $*', round(d['value'], 1), d['ms_per_step'])"
}
run cornell 128 A=1
run cornell 128 RPTGPU_NO_PLANE_TABLE=1
run cornell 128 RPTGPU_NO_PLANE_TABLE=1 RPTGPU_OBJECT_FILTER_MIN=1
run sphere 100 A=1
run sphere 100 RPTGPU_OBJECT_FILTER_MIN=1
run glass 16 A=1
run glass 16 RPTGPU_OBJECT_FILTER_MIN=1
run spheres 64 A=1
run spheres 64 RPTGPU_OBJECT_FILTER_MIN=1
run basic 64 A=1
run basic 64 RPTGPU_OBJECT_FILTER_MIN=1
P=$PWD/rpt_amd/lib/librptgpu_prof.so
echo "## room23, 32 spp, object filter" > $O/phase_room23.txt
RPTGPU_LIB=$P RPTGPU_PRINT_PHASES=1 timeout 300 python bench.py --scene room23 --steps 1 --warmup 0 --spp 32 --no-cpu-baseline --no-live-pmc 2>&1 >/dev/null | grep "^prof" >> $O/phase_room23.txt
cat $O/phase_room23.txt
