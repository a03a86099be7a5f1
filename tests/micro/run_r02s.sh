cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -n 5
timeout 300 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; cat gpurun_out/bench_final.json | cut -c1-900
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 2
