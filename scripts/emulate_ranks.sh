#!/bin/bash
# The per-rank cost that bounds N-GPU scaling, measured on ONE GPU:  bash scripts/emulate_ranks.sh <outfile> ["scene:spp ..."]
# For every scene: the full frame (N = 1), then rank 0's tiles among N = 2, 4, 8 ranks (bench.py --emulate-part-of: the
# interleaved 32x8-tile partition of rptgpu_render_batch_reduce, no collective).  ideal = the N = 1 time / N.
cd ${GRAFT_REPO_ROOT:-.}
OUT=${1:-gpurun_out/emulated_ranks.txt}; mkdir -p $(dirname $OUT)
LIST=${2:-"cornell:128 dragon:16 fractal_spheres:4 wine_glass:16"}
echo "# scene spp | N = 1 ms/step | N: rank-0 ms/step (ideal, ratio) ...   [bench.py --emulate-part-of N, 3 steps after 1 warm-up]" > $OUT
for sc in $LIST; do
  scene=${sc%%:*}; spp=${sc##*:}
  t1=$(timeout 600 python bench.py --scene $scene --spp $spp --steps 3 --warmup 1 --no-cpu-baseline --no-live-pmc 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  line="$scene $spp | $t1 |"
  for n in 2 4 8; do
    tn=$(timeout 600 python bench.py --scene $scene --spp $spp --steps 3 --warmup 1 --emulate-part-of $n 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step_of_this_rank'])")
    line="$line $(python -c "print('N=%d: %.2f (ideal %.2f, x%.3f)' % ($n, $tn, $t1/$n, $tn/($t1/$n)))")"
  done
  echo "$line" | tee -a $OUT
done
