cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/r05d && mkdir -p $O
python -m pytest tests/test_gpu_bf16.py -q -x > $O/bf16tests.log 2>&1; tail -4 $O/bf16tests.log
for v in 0 1; do
GATRES_BLOCKED=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt$v -o kt -- python3 bench.py --model gatres_large --batch-size 128 --steps 10 --warmup 3 --dtype bf16 --no-cpu-baseline --no-roofline > $O/kt$v.log 2>&1
python3 tests/micro/summarize_prof.py stats $O/kt$v $O/large_stats_nb$v.csv; echo "== NO_BLOCKED=$v"; head -22 $O/large_stats_nb$v.csv | cut -c1-60,200-400 | awk -F, '{print substr($0,1,60), $(NF-6), $(NF-5), $(NF-4)}' ; rm -rf $O/kt$v
done
