cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02a; mkdir -p $O
timeout 1700 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -15 $O/pytest.log
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err; cut -c1-600 $O/bench.json
timeout 200 python tests/stage_profile.py > $O/stage_times.txt 2>&1; tail -22 $O/stage_times.txt
