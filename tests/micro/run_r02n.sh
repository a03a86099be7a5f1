cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02n; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 5 $O/pytest.log
timeout 300 python tests/micro/hub_bench.py > $O/hub_new.json 2> $O/hub_new.err; tail -n 1 $O/hub_new.json
timeout 300 python tests/micro/hub_bench.py --hub-degree 100 --hubs 480 > $O/hub_new100.json 2> $O/hub_new.err; tail -n 1 $O/hub_new100.json
