cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_configs.py tests/test_next_rows.py tests/test_gpu_ops.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -n 5
show='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(round(d["value"]), round(d["ms_per_step"],4), round(r["avg_launch_us"],1), r["second_kernel"] and round(r["second_kernel"]["avg_launch_us"],1), round(r["frac"],4))'
echo FROM_STORE; timeout 300 python bench.py --no-cpu-baseline --from-store 2>/dev/null | tail -n 1 | tee gpurun_out/from_store.json | python -c "$show"
echo FROM_STORE_NO_STAGE; GATRES_NO_STAGE_MASK=1 timeout 300 python bench.py --no-cpu-baseline --from-store 2>/dev/null | tail -n 1 | python -c "$show"
echo DEFAULT; timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -n 1 | python -c "$show"
