# Round-6 mid-round measurements (MI355X): stage times of the new instantiation, drop-in / eval lines, the data-parallel step's
# kernel trace at world size 1.   gpurun --timeout 1500 -- 'bash tests/micro/r06_mid.sh'
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_mid; mkdir -p $O
B="python3 bench.py"
python3 -m pytest tests/test_gpu_headline.py tests/test_gpu_model.py -m gpu -x -q > $O/tests.txt 2>&1; tail -3 $O/tests.txt
GATRES_DIAG_LIB=1 timeout 200 python3 tests/stage_profile.py > $O/stage_times.txt 2>&1; head -24 $O/stage_times.txt
GATRES_DIAG_LIB=1 GATRES_WINDOW_RUNTIME_PHASES=1 timeout 200 python3 tests/stage_profile.py > $O/stage_times_runtime_phases.txt 2>&1; head -24 $O/stage_times_runtime_phases.txt
timeout 300 python3 tests/micro/drop_in_breakdown.py > $O/drop_in_breakdown.txt 2>&1; tail -12 $O/drop_in_breakdown.txt
timeout 300 $B --drop-in --steps 200 --warmup 20 2>/dev/null | tail -1 > $O/drop_in_torch_adam.json; cut -c1-200 $O/drop_in_torch_adam.json
timeout 300 $B --drop-in --fused-adam --steps 200 --warmup 20 2>/dev/null | tail -1 > $O/drop_in_fused_adam.json; cut -c1-200 $O/drop_in_fused_adam.json
timeout 300 $B --eval --steps 100 2>/dev/null | tail -1 > $O/eval_small_bs32.json; cut -c1-330 $O/eval_small_bs32.json
timeout 600 $B --eval --steps 30 --model gatres_large --batch-size 128 --dtype bf16 2>/dev/null | tail -1 > $O/eval_large_bs128_bf16.json; cut -c1-330 $O/eval_large_bs128_bf16.json
P="--steps 100 --warmup 10 --no-cpu-baseline --no-roofline --no-graph"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_eager -o kt -- $B $P > $O/kt_eager.log 2>&1
python3 tests/micro/summarize_prof.py stats $O/kt_eager $O/eager_kernel_stats.csv; head -8 $O/eager_kernel_stats.csv | cut -c1-220
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_coll -o kt -- $B $P --force-collective-path > $O/kt_coll.log 2>&1
python3 tests/micro/summarize_prof.py stats $O/kt_coll $O/collective_path_kernel_stats.csv; head -10 $O/collective_path_kernel_stats.csv | cut -c1-220
tail -1 $O/kt_eager.log | cut -c1-150; tail -1 $O/kt_coll.log | cut -c1-150
timeout 300 $B --no-cpu-baseline --no-roofline --no-graph 2>/dev/null | tail -1 > $O/eager.json; cut -c1-160 $O/eager.json
timeout 300 $B --no-cpu-baseline --no-roofline --no-graph --force-collective-path 2>/dev/null | tail -1 > $O/collective_path_1rank_eager.json; cut -c1-160 $O/collective_path_1rank_eager.json; python3 -c "import json;print(json.load(open('$O/collective_path_1rank_eager.json'))['config'].get('collective'))"
rm -rf $O/kt_eager $O/kt_coll
