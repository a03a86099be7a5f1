cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
show='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["ms_per_step"],4), "dropped", d["config"]["dropped_steps"], "loss", d["config"]["final_loss"])'
echo LONG; timeout 900 python bench.py --no-cpu-baseline --no-roofline --steps 20000 --warmup 20 2>/dev/null | tail -n 1 | python -c "$show"
echo LONG_bs16; timeout 900 python bench.py --no-cpu-baseline --no-roofline --steps 10000 --warmup 20 --batch-size 16 2>/dev/null | tail -n 1 | python -c "$show"
echo LONG_shuffle; timeout 900 python bench.py --no-cpu-baseline --no-roofline --steps 10000 --warmup 20 --shuffle-nodes 2>/dev/null | tail -n 1 | python -c "$show"
for k in 1 2 3; do timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_configs.py -x -q 2>&1 | grep -E "passed|failed" | tail -n 1; done
