cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
show='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(round(d["value"]), round(d["ms_per_step"],4), round(r["avg_launch_us"],1), round(r["frac"],4), r["second_kernel"] and round(r["second_kernel"]["avg_launch_us"],1), d["config"]["workload"][-110:])'
for v in ${DIAGS:-0 1 2 4 7}; do echo "DIAG_IS=$v"; GATRES_DIAG_IS=$v timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -n 1 | python -c "$show"; done
timeout 200 python tests/stage_profile.py 2>&1 | grep -v "^B\|^bar\|^agg\|^dX\|^edge\|^soft\|^mean\|^top\|^step\|^  \(fold\|chunk\|last\|partials\|slab\)\|^consumer" | tail -n 40
timeout 600 python -m pytest tests/test_gpu_model.py -x -q 2>&1 | tail -n 3
