cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02q; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_configs.py -m gpu -q -x -k "split or window or shuffled or golden or native_train or bitwise or edge_cases or parity or fault or full_size" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 12 $O/pytest.log
show='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(round(d["value"]), round(d["ms_per_step"],4), round(r["avg_launch_us"],1), round(r["frac"],4), r["second_kernel"] and round(r["second_kernel"]["avg_launch_us"],1), d["config"]["workload"][-110:])'
echo DEFAULT; timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -n 1 | tee $O/bench.json | python -c "$show"
for m in 8; do echo "SPLIT $m"; GATRES_FUSED_SPLIT=$m timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -n 1 | python -c "$show"; done
echo "SPLIT 8 old second launch"; GATRES_PARAM_GRADS_NO_STREAM=1 GATRES_FUSED_SPLIT=8 timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -n 1 | python -c "$show"
for bs in 64 256; do echo "BS $bs"; timeout 300 python bench.py --no-cpu-baseline --batch-size $bs 2>/dev/null | tail -n 1 | python -c "$show"; done
