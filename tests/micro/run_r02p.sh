cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02p; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -x -k "split or window or golden or bitwise or parity" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 3 $O/pytest.log
timeout 200 python tests/stage_profile.py > $O/stage_default.txt 2>&1; tail -n 19 $O/stage_default.txt | head -14
show='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(round(d["value"]), round(d["ms_per_step"],4), round(r["avg_launch_us"],1), round(r["frac"],4), r["second_kernel"] and round(r["second_kernel"]["avg_launch_us"],1), d["config"]["workload"][-110:])'
echo DEFAULT; timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -n 1 | tee $O/bench.json | python -c "$show"
