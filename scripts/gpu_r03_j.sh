cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03j; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8
python bench.py --scene simple_video > $O/simple_video.json 2>$O/simple_video.err; tail -c 1500 $O/simple_video.json
