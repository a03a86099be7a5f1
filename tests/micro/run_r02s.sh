cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_configs.py tests/test_gpu_model.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -n 5
show='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(round(d["value"]), round(d["ms_per_step"],4), round(r["avg_launch_us"],1), r["second_kernel"] and round(r["second_kernel"]["avg_launch_us"],1), round(r["frac"],4))'
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -n 1 | python -c "$show"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt -o kt -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/kt.log 2>&1
python3 tests/micro/summarize_prof.py stats gpurun_out/kt gpurun_out/kt_stats.csv; head -6 gpurun_out/kt_stats.csv | cut -c1-70,150-230
rm -rf gpurun_out/kt
