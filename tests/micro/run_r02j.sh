cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
show='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(round(d["value"]), round(d["ms_per_step"],4), round(r["avg_launch_us"],1), round(r["frac"],4), r["second_kernel"] and round(r["second_kernel"]["avg_launch_us"],1), d["config"]["workload"][-110:])'
echo DEFAULT; timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$show"
for m in 7 5; do echo "SPLIT $m"; GATRES_FUSED_SPLIT=$m timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$show"; done
