REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
O=gpurun_out/r02x; mkdir -p $O
export RPT_PROFILE_DST=$REPO/$O/profiles
mkdir -p $RPT_PROFILE_DST
prof() { bash scripts/profile.sh r02 $1 $2 $3 > $O/profile_$1.log 2>&1; python scripts/summarize_profile.py r02 $1 > $O/summary_$1.txt 2>&1; rm -rf gpurun_out/prof_r02_$1; }
prof sphere 100 100
prof glass 64 64
prof room23 128 128
ls $RPT_PROFILE_DST
