cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for m in 8 2; do echo "SPLIT $m"; GATRES_FUSED_SPLIT=$m timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], r['avg_launch_us'], r['second_kernel'], d['config']['workload'][-120:])"; done
for bs in 16 64; do echo "BS $bs"; timeout 300 python bench.py --no-cpu-baseline --batch-size $bs 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], r['avg_launch_us'], r['second_kernel'], d['config']['workload'][-120:])"; done
