REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
O=gpurun_out/r02teapots; mkdir -p $O
export RPT_PROFILE_DST=$REPO/$O/profiles
mkdir -p $RPT_PROFILE_DST
bash scripts/profile.sh r02 fractal_teapots 64 16 "--bounces 8" > $O/profile.log 2>&1
python scripts/summarize_profile.py r02 fractal_teapots > $O/summary.txt 2>&1
rm -rf gpurun_out/prof_r02_fractal_teapots
cat $O/profiles/*summary.md
