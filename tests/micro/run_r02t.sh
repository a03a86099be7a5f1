cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/ic; mkdir -p $O
B="python3 bench.py"
rocprofv3 -L 2>/dev/null | grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_INST[A-Z_]*\|SQC_INST[A-Z_]*" | sort -u | head -20
timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/a -o s -- $B --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/a.log 2>&1
python3 tests/micro/summarize_prof.py pmc $O/is.json gatres_window_kernel $O/a; python3 -c "
import json; d=json.load(open('$O/is.json'))['counters']
for k,v in d.items(): print(k, round(v['mean_per_launch']))"
export GATRES_FUSED_NO_INSTAGE=1
timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/b -o s -- $B --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/b.log 2>&1
python3 tests/micro/summarize_prof.py pmc $O/nois.json gatres_window_kernel $O/b; python3 -c "
import json; d=json.load(open('$O/nois.json'))['counters']
for k,v in d.items(): print(k, round(v['mean_per_launch']))"
tail -3 $O/a.log
rm -rf $O/a $O/b
