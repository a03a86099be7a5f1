# Round-6 mid-round measurements (MI355X).   gpurun --timeout 1500 -- 'bash tests/micro/r06_mid.sh'
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_mid; mkdir -p $O
B="python3 bench.py"
python3 -m pytest tests/test_gpu_headline.py tests/test_gpu_model.py tests/test_next_rows.py -m gpu -x -q > $O/tests.txt 2>&1; tail -3 $O/tests.txt
timeout 300 python3 tests/micro/drop_in_breakdown.py > $O/drop_in_breakdown.txt 2>&1; tail -12 $O/drop_in_breakdown.txt
for f in "" "--flat-adam" "--fused-adam"; do
timeout 300 $B --drop-in $f --steps 200 --warmup 20 2>/dev/null | tail -1 > "$O/drop_in${f}.json"; cut -c1-160 "$O/drop_in${f}.json"
done
timeout 300 $B --eval --steps 100 2>/dev/null | tail -1 > $O/eval_small_bs32.json; cut -c1-400 $O/eval_small_bs32.json
timeout 600 $B --eval --steps 30 --model gatres_large --batch-size 128 --dtype bf16 2>/dev/null | tail -1 > $O/eval_large_bs128_bf16.json; cut -c1-400 $O/eval_large_bs128_bf16.json
P="--steps 100 --warmup 10 --no-cpu-baseline --no-roofline --no-graph"
GATRES_BENCH_NO_HARD_EXIT=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_coll -o kt -- $B $P --force-collective-path > $O/kt_coll.log 2>&1
python3 tests/micro/summarize_prof.py stats $O/kt_coll $O/collective_path_kernel_stats.csv; head -10 $O/collective_path_kernel_stats.csv | cut -c1-220
rm -rf $O/kt_coll
