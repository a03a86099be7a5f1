# kernel-trace stats of the headline bench under env settings given as arguments: kt_probe.sh NAME [VAR=VALUE ...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
name=$1; shift
for kv in "$@"; do export "$kv"; done
O=gpurun_out/kt_$name; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline > $O/kt.log 2>&1
python3 tests/micro/summarize_prof.py stats $O/kt $O/stats.csv; head -7 $O/stats.csv | cut -c1-150
rm -rf $O/kt
