cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02n; mkdir -p $O
timeout 300 python tests/micro/hub_bench.py > $O/hub_new.json 2> $O/hub_new.err; tail -n 1 $O/hub_new.json
GATRES_LIB=tests/micro/_ab/libgatres_prehub.so timeout 300 python tests/micro/hub_bench.py > $O/hub_old.json 2> $O/hub_old.err; tail -n 1 $O/hub_old.json
