cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r01b
mkdir -p $O
python bench.py --steps 200 --warmup 20 2>$O/bench.err | tail -1 > $O/bench_n1.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/pmc_write.log 2>&1
python tests/stage_profile.py > $O/stage_times.txt 2>&1
for bs in 8 16 32 64 128 256; do echo "bs $bs"; python bench.py --batch-size $bs --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-140; python bench.py --batch-size $bs --no-cpu-baseline 2>/dev/null | tail -1 | grep -o '"roofline".*' | cut -c1-260; done > $O/batch_scaling.txt 2>&1
ls -R $O | head -40
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq -o s -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/pmc_sq.log 2>&1
