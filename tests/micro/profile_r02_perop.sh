# Round-2 evidence, per-op configurations only (re-run after the per-op kernels changed): gpurun_out/r02s_perop/ -> profiles/ by hand.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02s_perop; mkdir -p $O
B="python3 bench.py"
short='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"]), "snapshots/s", round(d["ms_per_step"],4), "ms/step")'
timeout 300 $B --no-cpu-baseline --per-op 2>/dev/null | tail -1 > $O/per_op.json; python3 -c "$short" < $O/per_op.json
timeout 600 $B --no-cpu-baseline --model gatres_large --batch-size 128 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/large_fp32.json; python3 -c "$short" < $O/large_fp32.json
timeout 600 $B --no-cpu-baseline --model gatres_large --batch-size 128 --steps 20 --warmup 5 --dtype bf16 2>/dev/null | tail -1 > $O/large_bf16.json; python3 -c "$short" < $O/large_bf16.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_l16 -o kt -- $B --model gatres_large --batch-size 128 --steps 10 --warmup 3 --dtype bf16 --no-cpu-baseline --no-roofline > $O/kt_l16.log 2>&1
python3 tests/micro/summarize_prof.py stats $O/kt_l16 $O/large_bf16_kernel_stats.csv
timeout 900 $B --no-cpu-baseline --model gatres_large --nodes 50000 --pipes 75000 --batch-size 2 --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/large_50k_bs2.json; python3 -c "$short" < $O/large_50k_bs2.json
timeout 900 $B --no-cpu-baseline --model gatres_large --nodes 50000 --pipes 75000 --batch-size 2 --steps 10 --warmup 3 --dtype bf16 2>/dev/null | tail -1 > $O/large_50k_bs2_bf16.json; python3 -c "$short" < $O/large_50k_bs2_bf16.json
timeout 900 $B --no-cpu-baseline --nodes 50000 --pipes 75000 --batch-size 16 --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/small_50k_bs16.json; python3 -c "$short" < $O/small_50k_bs16.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_50k -o kt -- $B --model gatres_large --nodes 50000 --pipes 75000 --batch-size 2 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $O/kt_50k.log 2>&1
python3 tests/micro/summarize_prof.py stats $O/kt_50k $O/large_50k_kernel_stats.csv
rm -rf $O/kt_l16 $O/kt_50k
