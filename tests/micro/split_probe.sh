cd "$GRAFT_REPO_ROOT"
run() {
  echo "$*: $(env "$@" timeout 300 python bench.py --no-cpu-baseline 2>gpurun_out/split.err | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); r=d["roofline"]; print(d["ms_per_step"], d["config"]["workload"][-120:-60], r.get("avg_launch_us"), (r.get("second_kernel") or {}).get("avg_launch_us"))')"
}
{
run GATRES_FUSED_SPLIT=8
run GATRES_FUSED_SPLIT=7 GATRES_FUSED_NO_CONSUMERS=1
run GATRES_FUSED_SPLIT=6 GATRES_FUSED_NO_CONSUMERS=1
run GATRES_FUSED_SPLIT=7
} 2>&1 | tee gpurun_out/split_probe.txt
