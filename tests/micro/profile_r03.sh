# Round-3 evidence run (MI355X): everything lands under gpurun_out/r03_final/, the summaries are copied into profiles/ by hand.
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tests/micro/profile_r03.sh'
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_final; mkdir -p $O
B="python3 bench.py"
short='import sys,json; d=json.loads(sys.stdin.read()); r=d.get("roofline") or {}; sk=r.get("second_kernel"); print(round(d["value"]), "snapshots/s", round(d["ms_per_step"],4), "ms/step | dominant:", r.get("kernel","")[:40], round(r.get("avg_launch_us",0),1), "us  frac", round(r.get("frac",0),4), "| 2nd launch us:", sk and round(sk["avg_launch_us"],1), "|", d["config"]["workload"][-95:])'
timeout 600 $B 2>$O/bench.err | tail -1 > $O/bench_n1.json; python3 -c "$short" < $O/bench_n1.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- $B --steps 100 --warmup 10 --no-cpu-baseline > $O/kt.log 2>&1
python3 tests/micro/summarize_prof.py stats $O/kt $O/fused_kernel_stats.csv; head -6 $O/fused_kernel_stats.csv | cut -c1-200
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- $B --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- $B --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/pmc_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq -o s -- $B --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/pmc_sq.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $O/pmc_inst -o i -- $B --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/pmc_inst.log 2>&1
python3 tests/micro/summarize_prof.py pmc $O/fused_pmc.json gatres_window_kernel $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_inst; head -50 $O/fused_pmc.json
python3 tests/micro/summarize_prof.py pmc $O/pgs_pmc.json param_grads_stream_kernel $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_inst
[ -f gnn-pressure-estimation_amd/lib/libgatres_hip_diag.so ] || python3 gnn-pressure-estimation_amd/_build.py --diag > /dev/null 2>&1
GATRES_DIAG_LIB=1 timeout 200 python3 tests/stage_profile.py > $O/stage_times.txt 2>&1; head -22 $O/stage_times.txt
{ for m in 4 6 8; do echo "GATRES_FUSED_SPLIT=$m"; GATRES_FUSED_SPLIT=$m timeout 300 $B --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "$short"; done
  echo "GATRES_NO_PART_TABLES=1 (prologue tables derived in every launch)"; GATRES_NO_PART_TABLES=1 timeout 300 $B --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "$short"
  echo "GATRES_FUSED_HEARTBEAT=1 (hand-offs paced by heartbeat granules although the plan is symmetric)"; GATRES_FUSED_HEARTBEAT=1 timeout 300 $B --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "$short"
  echo "GATRES_FUSED_SAFE_SYNC=1 (agent-scope granules: the cross-XCD form)"; GATRES_FUSED_SAFE_SYNC=1 timeout 300 $B --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "$short"; } > $O/variants.txt 2>&1; cat $O/variants.txt
{ for bs in 8 16 32 64 128 256; do echo "bs $bs"; timeout 300 $B --batch-size $bs --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "$short"; done; } > $O/batch_scaling.txt 2>&1; cat $O/batch_scaling.txt
timeout 300 $B --no-cpu-baseline --shuffle-nodes 2>/dev/null | tail -1 > $O/shuffle_nodes.json; python3 -c "$short" < $O/shuffle_nodes.json
timeout 300 $B --no-cpu-baseline --from-store 2>/dev/null | tail -1 > $O/from_store.json; python3 -c "$short" < $O/from_store.json
timeout 300 $B --no-cpu-baseline --host-batches 2>/dev/null | tail -1 > $O/host_batches.json; python3 -c "$short" < $O/host_batches.json
timeout 300 $B --no-cpu-baseline --per-op 2>/dev/null | tail -1 > $O/per_op.json; python3 -c "$short" < $O/per_op.json
timeout 300 $B --no-cpu-baseline --no-roofline --force-collective-path 2>/dev/null | tail -1 > $O/collective_path_1rank.json; python3 -c "$short" < $O/collective_path_1rank.json
timeout 600 $B --no-cpu-baseline --model gatres_large --batch-size 128 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/large_fp32.json; python3 -c "$short" < $O/large_fp32.json
timeout 600 $B --no-cpu-baseline --model gatres_large --batch-size 128 --steps 50 --warmup 10 --dtype bf16 2>/dev/null | tail -1 > $O/large_bf16.json; python3 -c "$short" < $O/large_bf16.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_l16 -o kt -- $B --model gatres_large --batch-size 128 --steps 10 --warmup 3 --dtype bf16 --no-cpu-baseline --no-roofline > $O/kt_l16.log 2>&1
python3 tests/micro/summarize_prof.py stats $O/kt_l16 $O/large_bf16_kernel_stats.csv
timeout 900 $B --no-cpu-baseline --model gatres_large --nodes 50000 --pipes 75000 --batch-size 2 --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/large_50k_bs2.json; python3 -c "$short" < $O/large_50k_bs2.json
timeout 900 $B --no-cpu-baseline --model gatres_large --nodes 50000 --pipes 75000 --batch-size 2 --steps 10 --warmup 3 --dtype bf16 2>/dev/null | tail -1 > $O/large_50k_bs2_bf16.json; python3 -c "$short" < $O/large_50k_bs2_bf16.json
timeout 900 $B --no-cpu-baseline --nodes 50000 --pipes 75000 --batch-size 16 --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/small_50k_bs16.json; python3 -c "$short" < $O/small_50k_bs16.json
[ -x tests/micro/lds_dma_probe ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tests/micro/lds_dma_probe.hip -o tests/micro/lds_dma_probe
./tests/micro/lds_dma_probe > $O/lds_dma_probe.txt 2>&1; cat $O/lds_dma_probe.txt
[ -f gnn-pressure-estimation_amd/lib/probe_st.so ] || bash tests/micro/build_probes.sh
{ echo "== bf16 projections of gatres_large alone (C-ABI, hipGraph of 12 launches over 12 buffer sets): proj_bf16_tile_kernel (default)"; timeout 120 python3 tests/micro/proj_probe.py 2>/dev/null | tail -4
  echo "== GATRES_PROJ_STREAM=1: proj_bf16_stream_kernel"; GATRES_PROJ_STREAM=1 timeout 120 python3 tests/micro/proj_probe.py 2>/dev/null | tail -4
  echo "== GATRES_PROJ_STREAM=1, per-wave stamps inside the kernel (lib/probe_st.so, tests/micro/build_probes.sh); us since the first wave started"; GATRES_PROJ_STREAM=1 timeout 120 python3 tests/micro/proj_probe.py --lib gnn-pressure-estimation_amd/lib/probe_st.so --stamps 2>/dev/null | tail -32; } > $O/proj_probe.txt 2>&1; cat $O/proj_probe.txt
{ for v in 0 1; do for dt in fp32 bf16; do echo "gatres_large C-Town bs 128 $dt GATRES_SIDE_STREAM=$v"; GATRES_SIDE_STREAM=$v timeout 600 $B --no-cpu-baseline --model gatres_large --batch-size 128 --steps 20 --warmup 5 --dtype $dt 2>/dev/null | tail -1 | python3 -c "$short"; done; done; } > $O/side_stream.txt 2>&1; cat $O/side_stream.txt
rm -rf $O/kt $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_inst $O/kt_l16
ls -la $O
