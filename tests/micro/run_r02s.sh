cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_bf16.py -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -n 5
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt -o kt -- python3 bench.py --model gatres_large --batch-size 128 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/kt.log 2>&1
python3 tests/micro/summarize_prof.py stats gpurun_out/kt gpurun_out/large_fp32_kernel_stats.csv
python3 - <<'PY'
import csv
rows=list(csv.reader(open('gpurun_out/large_fp32_kernel_stats.csv')))
for r in rows[1:]:
    print(f"{r[0][:100]:100s} calls {int(r[1]):5d} avg {float(r[3])/1000:8.1f} us  {float(r[4]):5.1f}%")
PY
rm -rf gpurun_out/kt
