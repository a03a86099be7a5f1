# scratch driver of the round-2 sessions: a few bench lines + the stage profile on the GPU box (see profile_r02.sh for the
# evidence run that fills profiles/)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
show='import sys,json; d=json.loads(sys.stdin.read()); r=d.get("roofline") or {}; sk=r.get("second_kernel"); print(round(d["value"],1), round(d["ms_per_step"],4), r.get("avg_launch_us") and round(r["avg_launch_us"],1), sk and round(sk["avg_launch_us"],1), r.get("frac") and round(r["frac"],4))'
echo DEFAULT; timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -n 1 | python -c "$show"
echo "6 + 2 consumers"; GATRES_FUSED_PREFER_CONSUMERS=1 timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -n 1 | python -c "$show"
echo LARGE_bf16; timeout 600 python bench.py --no-cpu-baseline --model gatres_large --batch-size 128 --steps 20 --warmup 5 --dtype bf16 2>/dev/null | tail -n 1 | python -c "$show"
timeout 200 python tests/stage_profile.py 2>&1 | grep -v "^B\|^bar\|^agg\|^dX\|^edge\|^soft\|^mean\|^top\|^step" | tail -n 24
