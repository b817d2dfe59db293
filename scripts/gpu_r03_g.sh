cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
REP=2 bash scripts/gpu_ab.sh r03i "cornell:256 room23:64 sphere:100 glass:32" "base"
