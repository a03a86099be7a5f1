cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02p
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r02p/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02p/pytest.log; tail -4 gpurun_out/r02p/pytest.log
bash tests/micro/profile_r02.sh
