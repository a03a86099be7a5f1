# PMC counters of the top per-op kernels of gatres_large (C-Town, bs 128, bf16): separate passes, no trace domains
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pj; mkdir -p $O
B="python3 bench.py --model gatres_large --batch-size 128 --steps 4 --warmup 2 --dtype bf16 --no-cpu-baseline --no-roofline"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/a -o s -- $B > $O/a.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/b -o s -- $B > $O/b.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/c -o s -- $B > $O/c.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/d -o s -- $B > $O/d.log 2>&1
python3 - <<'PY'
import json, subprocess, sys
out={}
for k in ["proj_bf16_kernel<128, 256, 2, 1>", "proj_bf16_kernel<256, 128, 1, 2>", "dw2d_bf16_kernel<256, 128>", "gat_aggregate_bwd_dst", "gat_aggregate_bwd_src", "gat_aggregate_fwd_kernel<true", "conv_param_grads_bf16"]:
    subprocess.run([sys.executable, "tests/micro/summarize_prof.py", "pmc", "gpurun_out/pj/x.json", k, "gpurun_out/pj/a", "gpurun_out/pj/b", "gpurun_out/pj/c", "gpurun_out/pj/d"], check=True)
    d=json.load(open("gpurun_out/pj/x.json"))["counters"]
    out[k]={c: round(v["mean_per_launch"]) for c,v in d.items()}
    if "FETCH_SIZE" in out[k] and "WRITE_SIZE" in out[k]:
        out[k]["hbm_side_bytes_per_launch"]=int((2*out[k]["FETCH_SIZE"]+out[k]["WRITE_SIZE"])*1024)
json.dump({"workload": "gatres_large (25 x 128), C-Town-sized batch of 128 snapshots, bf16, per-op kernels", "units": "FETCH_SIZE / WRITE_SIZE in KB (bytes = (2 x FETCH + WRITE) x 1024 on gfx950); SQ_* cycle counters in quad-cycles", "mean_per_launch": out}, open("gpurun_out/pj/large_bf16_pmc.json","w"), indent=1)
print(json.dumps(out)[:600])
PY
rm -rf $O/a $O/b $O/c $O/d
