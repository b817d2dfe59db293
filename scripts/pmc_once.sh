#!/bin/bash
# one PMC pass over a bench command:  bash scripts/pmc_once.sh <tag> "<counters>" "<bench args>"
TAG=$1; CTRS=$2; ARGS=$3
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout -k 5 240 rocprofv3 --pmc $CTRS --kernel-trace -d $OUT -o bench -- python $REPO/bench.py --steps 1 --warmup 0 --spp 4 --no-cpu-baseline $ARGS > $OUT/log.txt 2>&1
python - <<PY
import re, sqlite3
c = sqlite3.connect("$OUT/bench_results.db")
q = "select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection group by kernel_name, counter_name"
for k, cn, tot, n in c.execute(q):
    k = re.sub(r"\\b\\w+::", "", k.split("(")[0]).replace("void ", "").strip()
    if k.startswith("rpt_"): print("%-12s %-28s %.4g (%d launches)" % (k, cn, tot, n))
PY
