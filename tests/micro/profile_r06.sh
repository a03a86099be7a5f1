# Round-6 evidence run (MI355X): everything lands under gpurun_out/r06_final/; the summaries are copied into profiles/ by
# tests/micro/collect_r06.py (run here, after the call).
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tests/micro/profile_r06.sh'; then HERE: python tests/micro/collect_r06.py publish gpurun_out/r06_final
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_final; mkdir -p $O
B="python3 bench.py"
short='import sys,json; d=json.loads(sys.stdin.read()); r=d.get("roofline") or {}; sk=r.get("second_kernel"); print(round(d["value"]), "snapshots/s", round(d["ms_per_step"],4), "ms/step | dominant:", r.get("kernel","")[:40], round(r.get("avg_launch_us",0),1), "us  frac", round(r.get("frac",0),4), "| 2nd launch us:", sk and round(sk["avg_launch_us"],1), "|", d["config"]["workload"][-95:])'
# ---- headline config: kernel trace, four counter passes (program directly behind `--`), then the bench line (which reads the PMC file)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- $B --steps 100 --warmup 10 --no-cpu-baseline --no-roofline > $O/kt.log 2>&1
python3 tests/micro/summarize_prof.py stats $O/kt $O/fused_kernel_stats.csv; head -6 $O/fused_kernel_stats.csv | cut -c1-200
P="--steps 20 --warmup 5 --no-cpu-baseline --no-roofline"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- $B $P > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- $B $P > $O/pmc_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq -o s -- $B $P > $O/pmc_sq.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $O/pmc_inst -o i -- $B $P > $O/pmc_inst.log 2>&1
python3 tests/micro/summarize_prof.py pmc $O/fused_pmc.json gatres_window_kernel $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_inst
python3 tests/micro/summarize_prof.py pmc $O/param_grads_pmc.json param_grads_reg_kernel $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_inst
python3 tests/micro/collect_r06.py pmc_raw $O      # -> $O/fused_pmc_raw.json AND profiles/r06_fused_pmc_raw.json (bench.py reads that one)
timeout 600 $B 2>$O/bench.err | tail -1 > $O/bench_n1.json; python3 -c "$short" < $O/bench_n1.json
[ -f gnn-pressure-estimation_amd/lib/libgatres_hip_diag.so ] || python3 gnn-pressure-estimation_amd/_build.py --diag > /dev/null 2>&1
GATRES_DIAG_LIB=1 timeout 200 python3 tests/stage_profile.py > $O/stage_times.txt 2>&1; head -22 $O/stage_times.txt
{ for bs in 8 16 32 64 128 256; do echo "bs $bs"; timeout 300 $B --batch-size $bs --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "$short"; done; } > $O/batch_scaling.txt 2>&1; cat $O/batch_scaling.txt
timeout 300 $B --no-cpu-baseline --shuffle-nodes 2>/dev/null | tail -1 > $O/shuffle_nodes.json; python3 -c "$short" < $O/shuffle_nodes.json
timeout 300 $B --no-cpu-baseline --from-store 2>/dev/null | tail -1 > $O/from_store.json; python3 -c "$short" < $O/from_store.json
timeout 300 $B --no-cpu-baseline --host-batches 2>/dev/null | tail -1 > $O/host_batches.json; python3 -c "$short" < $O/host_batches.json
timeout 300 $B --no-cpu-baseline --per-op 2>/dev/null | tail -1 > $O/per_op.json; python3 -c "$short" < $O/per_op.json
timeout 300 $B --no-cpu-baseline --no-roofline --force-collective-path 2>/dev/null | tail -1 > $O/collective_path_1rank.json; python3 -c "$short" < $O/collective_path_1rank.json
timeout 300 $B --no-cpu-baseline --no-roofline --no-graph 2>/dev/null | tail -1 > $O/eager.json; python3 -c "$short" < $O/eager.json
timeout 300 $B --no-cpu-baseline --no-roofline --no-graph --force-collective-path 2>/dev/null | tail -1 > $O/collective_path_1rank_eager.json; python3 -c "$short" < $O/collective_path_1rank_eager.json
timeout 300 $B --no-cpu-baseline --no-roofline --no-graph --force-collective-path --fused-buckets 2 2>/dev/null | tail -1 > $O/collective_path_1rank_eager_2buckets.json; python3 -c "$short" < $O/collective_path_1rank_eager_2buckets.json
# ---- where the data-parallel sequence's extra microseconds go at world size 1: kernel traces of the eager plain step and of the eager collective path
PT="--steps 100 --warmup 10 --no-cpu-baseline --no-roofline --no-graph"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_eager -o kt -- $B $PT > $O/kt_eager.log 2>&1
python3 tests/micro/summarize_prof.py stats $O/kt_eager $O/eager_kernel_stats.csv
GATRES_BENCH_NO_HARD_EXIT=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_coll -o kt -- $B $PT --force-collective-path > $O/kt_coll.log 2>&1
python3 tests/micro/summarize_prof.py stats $O/kt_coll $O/collective_path_kernel_stats.csv; head -8 $O/collective_path_kernel_stats.csv | cut -c1-200
# ---- the drop-in module (reference loop body verbatim) and the inference line
for f in "" "--flat-adam" "--fused-adam"; do timeout 300 $B --drop-in $f --steps 200 --warmup 20 2>/dev/null | tail -1 > "$O/drop_in${f}.json"; cut -c1-150 "$O/drop_in${f}.json"; done
timeout 300 python3 tests/micro/drop_in_breakdown.py > $O/drop_in_breakdown.txt 2>&1; tail -12 $O/drop_in_breakdown.txt
timeout 300 $B --eval --steps 100 2>/dev/null | tail -1 > $O/eval_small_bs32.json; cut -c1-300 $O/eval_small_bs32.json
timeout 600 $B --eval --steps 30 --model gatres_large --batch-size 128 --dtype bf16 2>/dev/null | tail -1 > $O/eval_large_bs128_bf16.json; cut -c1-300 $O/eval_large_bs128_bf16.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_eval -o kt -- $B --eval --steps 200 > $O/kt_eval.log 2>&1
python3 tests/micro/summarize_prof.py stats $O/kt_eval $O/eval_kernel_stats.csv; rm -rf $O/kt_eval
timeout 300 $B --no-cpu-baseline --copy-batches 2>/dev/null | tail -1 > $O/copy_batches.json
bash tests/micro/r06_phase_ab.sh > /dev/null 2>&1; cp gpurun_out/r06_phase_ab.txt $O/phase_ab.txt; tail -7 $O/phase_ab.txt
bash tests/micro/r06_sync_start_ab.sh > /dev/null 2>&1; cp gpurun_out/r06_sync_start_ab.txt $O/sync_start_ab.txt; tail -4 $O/sync_start_ab.txt
timeout 300 $B --no-cpu-baseline --no-roofline --graph-steps 1 2>/dev/null | tail -1 > $O/one_step_per_graph.json; python3 -c "$short" < $O/one_step_per_graph.json
# ---- the launcher path: 2 ranks on this one GPU (test overrides: gloo, shared device)
GATRES_DIST_BACKEND=gloo GATRES_BENCH_SHARE_GPU=1 timeout 600 $B --gpus 2 --batch-size 8 --steps 50 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/launcher_2ranks_one_gpu.json; python3 -c "$short" < $O/launcher_2ranks_one_gpu.json
# ---- config 3 (gatres_large, C-Town, bs 128, bf16): bench line, kernel stats, counters incl. MFMA utilisation
timeout 600 $B --no-cpu-baseline --model gatres_large --batch-size 128 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/large_fp32.json; python3 -c "$short" < $O/large_fp32.json
L="$B --model gatres_large --batch-size 128 --steps 4 --warmup 2 --dtype bf16 --no-cpu-baseline --no-roofline"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_l16 -o kt -- $B --model gatres_large --batch-size 128 --steps 10 --warmup 3 --dtype bf16 --no-cpu-baseline --no-roofline > $O/kt_l16.log 2>&1
python3 tests/micro/summarize_prof.py stats $O/kt_l16 $O/large_bf16_kernel_stats.csv
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/lb -o s -- $L > $O/lb.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/lc -o s -- $L > $O/lc.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --output-format csv -d $O/la -o s -- $L > $O/la.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $O/ld -o s -- $L > $O/ld.log 2>&1
python3 tests/micro/collect_r06.py large_pmc $O
# (the bench line of config 3 AFTER its counter file exists: it attaches the HBM-side rates of the per-op kernels from it)
timeout 600 $B --no-cpu-baseline --model gatres_large --batch-size 128 --steps 50 --warmup 10 --dtype bf16 2>/dev/null | tail -1 > $O/large_bf16.json; python3 -c "$short" < $O/large_bf16.json
timeout 900 $B --no-cpu-baseline --model gatres_large --nodes 50000 --pipes 75000 --batch-size 2 --steps 10 --warmup 3 --dtype bf16 2>/dev/null | tail -1 > $O/large_50k_bs2_bf16.json; python3 -c "$short" < $O/large_50k_bs2_bf16.json
timeout 900 $B --no-cpu-baseline --nodes 50000 --pipes 75000 --batch-size 16 --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/small_50k_bs16.json; python3 -c "$short" < $O/small_50k_bs16.json
rm -rf $O/kt_eager $O/kt_coll $O/kt $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_inst $O/kt_l16 $O/la $O/lb $O/lc $O/ld
ls -la $O
# ---- the bench line at the driver's protocol (BENCH_rNN.json: python3 bench.py --gpus 1 --steps 20 --warmup 5)
timeout 600 $B --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_driver_protocol.json; python3 -c "$short" < $O/bench_driver_protocol.json
