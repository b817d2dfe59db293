#!/bin/bash
# second GPU call of round 2: parity of the wave-cooperative mesh traversal, A/B throughput, counters incl. TA/TCP
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
O=gpurun_out/r02b
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
for mp in 0 1; do
  for cfg in "dragon 16" "wine_glass 4"; do
    set -- $cfg
    RPTGPU_MESH_PAIRS=$mp timeout 300 python bench.py --scene $1 --spp $2 --steps 2 --no-cpu-baseline > $O/bench_$1_mp$mp.json 2> $O/bench_$1_mp$mp.err
    python -c "
import json
d=json.loads(open('$O/bench_$1_mp$mp.json').read().strip().splitlines()[-1]); k=d['roofline']['kernels']
print('mesh_pairs=$mp %-12s %7.1f Msamples/s' % ('$1', d['value']), {n:round(v['total_ms'],1) for n,v in k.items()})"
  done
done
cd /tmp && export TMPDIR=/tmp
for mp in 0 1; do
  P=$REPO/$O/pmc_ta_mp$mp
  RPTGPU_MESH_PAIRS=$mp rocprofv3 --pmc TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_INSTS_SMEM SQ_WAIT_INST_LDS --kernel-trace -d $P -o bench -- python $REPO/bench.py --scene dragon --steps 1 --warmup 0 --spp 4 --no-cpu-baseline > $P.log 2>&1
  python - <<PY
import re, sqlite3
c = sqlite3.connect("$P/bench_results.db")
q = "select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection group by kernel_name, counter_name"
for k, cn, tot, n in c.execute(q):
    k = re.sub(r"\\b\\w+::", "", k.split("(")[0]).replace("void ", "").strip()
    if "trace" in k: print("mp=$mp %-28s %-36s %.4g (%d launches)" % (k, cn, tot, n))
for name, tot, n in c.execute("select name, total_duration, total_calls from top_kernels limit 6"):
    print("mp=$mp", re.sub(r"\\b\\w+::", "", name.split("(")[0])[:40], tot, n)
PY
done > $REPO/$O/ta_counters.txt 2>&1
cat $REPO/$O/ta_counters.txt
