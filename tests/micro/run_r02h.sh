cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02h; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bf16.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -30 $O/pytest.log
show='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(round(d["value"]), round(d["ms_per_step"],4), r["kernel"], round(r["avg_launch_us"],1), round(r["frac"],4), round(r["frac_of_step_time"],4)); [print("   ",k["kernel"],k["avg_us"],k["gbs"]) for k in d.get("kernels",[])]'
echo "LARGE fp32"; timeout 600 python bench.py --no-cpu-baseline --model gatres_large --batch-size 128 --steps 20 --warmup 5 2>$O/l32.err | tail -1 | tee $O/large_fp32.json | python -c "$show"
echo "LARGE bf16"; timeout 600 python bench.py --no-cpu-baseline --model gatres_large --batch-size 128 --steps 20 --warmup 5 --dtype bf16 2>$O/l16.err | tail -1 | tee $O/large_bf16.json | python -c "$show"
tail -3 $O/l16.err
