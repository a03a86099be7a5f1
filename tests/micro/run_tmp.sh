cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
show='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["ms_per_step"],4))'
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_bf16.py tests/test_gpu_model.py -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -n 8
echo LARGE_bf16; timeout 600 python bench.py --no-cpu-baseline --no-roofline --model gatres_large --batch-size 128 --steps 20 --warmup 5 --dtype bf16 2>/dev/null | tail -n 1 | python -c "$show"
echo LARGE_fp32; timeout 600 python bench.py --no-cpu-baseline --no-roofline --model gatres_large --batch-size 128 --steps 20 --warmup 5 2>/dev/null | tail -n 1 | python -c "$show"
echo PEROP_small; timeout 600 python bench.py --no-cpu-baseline --no-roofline --per-op 2>/dev/null | tail -n 1 | python -c "$show"
