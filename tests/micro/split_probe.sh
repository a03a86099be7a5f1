# bench.py under environment settings: split_probe.sh runs a fixed list (edit it), prints ms/step and the two launches' durations
cd "$GRAFT_REPO_ROOT"
run() {
  bs=$1; shift
  echo "bs $bs $*: $(env "$@" timeout 300 python bench.py --batch-size $bs --no-cpu-baseline 2>gpurun_out/split.err | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); r=d["roofline"]; print("%.5f ms/step %.0f snapshots/s" % (d["ms_per_step"], d["value"]), r.get("avg_launch_us"), (r.get("second_kernel") or {}).get("avg_launch_us"))')"
}
{
run 8 GATRES_FUSED_WITH_CONSUMERS=1
run 8 X=1
run 16 GATRES_FUSED_WITH_CONSUMERS=1
run 16 X=1
run 24 GATRES_FUSED_WITH_CONSUMERS=1
run 24 X=1
} 2>&1 | tee gpurun_out/split_probe.txt
