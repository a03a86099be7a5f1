cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -n 5
show='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"]), round(d["ms_per_step"],4))'
echo COLLECTIVE_1RANK; timeout 300 python bench.py --no-cpu-baseline --no-roofline --force-collective-path 2>/dev/null | tail -n 1 | tee gpurun_out/collective.json | python -c "$show"
echo DEFAULT; timeout 300 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | tail -n 1 | python -c "$show"
echo LARGE_bf16; timeout 600 python bench.py --no-cpu-baseline --no-roofline --model gatres_large --batch-size 128 --steps 20 --warmup 5 --dtype bf16 2>/dev/null | tail -n 1 | python -c "$show"
