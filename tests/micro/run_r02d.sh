cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02d; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_configs.py -m gpu -q -x -k "split or window or shuffled or golden or native_train or transient" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -2 $O/bench.err | head -1
timeout 200 python tests/stage_profile.py > $O/stage_times.txt 2>&1; tail -17 $O/stage_times.txt
GATRES_FUSED_SAFE_SYNC=1 timeout 300 python bench.py --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c1-200
