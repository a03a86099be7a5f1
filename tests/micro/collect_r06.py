"""Post-processing of tests/micro/profile_r06.sh on the GPU box (files under gpurun_out/r06_final -> profiles/r06_*).

  collect_r06.py pmc_raw <dir>     profiles/r06_fused_pmc_raw.json: FETCH / WRITE of the two heavy launches of the headline step,
                                   EACH KERNEL'S AVERAGE DURATION in the profiled run (from the kernel-stats table of the same
                                   session: bench.py accepts the file only while its own HIP-event durations agree within 5 %)
                                   and the wait / issue ratios of the dominant kernel
  collect_r06.py large_pmc <dir>   profiles/r06_large_ctown_bs128_bf16_pmc.json: counters of config 3's top kernels incl. the
                                   MFMA utilisation of the projections
"""
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def kernel_avg_us(stats_csv, substring):
    for row in csv.DictReader(open(stats_csv)):
        name = row.get("Name") or row.get("KernelName") or ""
        if substring in name:
            for key in ("AverageNs", "Average(ns)", "AvgNs"):
                if key in row:
                    return float(row[key]) / 1e3
            total, calls = float(row.get("TotalDurationNs", 0)), float(row.get("Calls", 0))
            if calls:
                return total / calls / 1e3
    return None


def pmc_raw(d):
    main = json.load(open(os.path.join(d, "fused_pmc.json")))["counters"]
    pg = json.load(open(os.path.join(d, "param_grads_pmc.json")))["counters"]
    stats = os.path.join(d, "fused_kernel_stats.csv")
    kb = lambda c, k: {"launches": c[k]["launches"], "mean_counter_value_KB": c[k]["mean_per_launch"], "min": c[k]["min"], "max": c[k]["max"]}
    b = lambda c: int((2.0 * c["FETCH_SIZE"]["mean_per_launch"] + c["WRITE_SIZE"]["mean_per_launch"]) * 1024)
    wc, wa = main["SQ_WAVE_CYCLES"]["mean_per_launch"], main["SQ_WAIT_ANY"]["mean_per_launch"]
    out = {
        "kernel": "gatres_window_kernel<32, 1024, 0x1f16> (the training instantiation with every launch fact), bs=32, 8 CUs per snapshot (256 workgroups); second_kernel: "
                  "param_grads_reg_kernel<32> (960 workgroups x 256 threads, the deferred parameter gradients)",
        "command": "rocprofv3 --pmc FETCH_SIZE --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline "
                   "--no-roofline (and the same with --pmc WRITE_SIZE / the SQ sets: separate passes, tests/micro/profile_r06.sh)",
        "units": "rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KB; FETCH_SIZE counts 64 B per 128-B request on gfx950 "
                 "(MI355X_MICROARCH.md, HBM): bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024",
        "kernel_avg_us": kernel_avg_us(stats, "gatres_window_kernel"),
        "kernel_avg_us_source": "rocprofv3 --kernel-trace --stats of the same session (profiles/r06_fused_kernel_stats.csv)",
        "FETCH_SIZE": kb(main, "FETCH_SIZE"), "WRITE_SIZE": kb(main, "WRITE_SIZE"),
        "second_kernel": {"kernel_avg_us": kernel_avg_us(stats, "param_grads_reg_kernel"),
                          "FETCH_SIZE": kb(pg, "FETCH_SIZE"), "WRITE_SIZE": kb(pg, "WRITE_SIZE"),
                          "hbm_side_bytes_per_launch": b(pg)},
        "hbm_side_bytes_per_launch": b(main), "hbm_side_bytes_both_launches": b(main) + b(pg),
        "wait": {"SQ_WAIT_ANY_over_SQ_WAVE_CYCLES": wa / wc,
                 "SQ_ACTIVE_INST_ANY_over_SQ_WAVE_CYCLES": main["SQ_ACTIVE_INST_ANY"]["mean_per_launch"] / wc,
                 "SQ_WAIT_INST_ANY_over_SQ_WAVE_CYCLES": main["SQ_WAIT_INST_ANY"]["mean_per_launch"] / wc,
                 "note": "gatres_window_kernel: share of its wave-cycles spent parked at s_waitcnt / barriers, issuing, and "
                         "waiting to issue (profiles/r06_fused_pmc.json)"},
    }
    for path in (os.path.join(d, "fused_pmc_raw.json"), os.path.join(ROOT, "profiles", "r06_fused_pmc_raw.json")):
        with open(path, "w") as f:
            json.dump(out, f, indent=1)
    print("pmc_raw:", out["kernel_avg_us"], out["second_kernel"]["kernel_avg_us"], out["hbm_side_bytes_both_launches"], out["wait"]["SQ_WAIT_ANY_over_SQ_WAVE_CYCLES"])


def large_pmc(d):
    # (the weight-gradient kernels appear under their mangled names in rocprofv3's tables)
    kernels = ["blocked_kernel<0, 256, 128", "blocked_kernel<1, 128, 256", "blocked_kernel<2, 128, 256", "blocked_kernel<2, 256, 128",
               "blocked_kernelILi0ELi256ELi128", "blocked_kernelILi1ELi128ELi256", "blocked_kernelILi2ELi128ELi256", "blocked_kernelILi2ELi256ELi128",
               "proj_bf16_tile_kernel<128, 256", "proj_bf16_tile_kernel<256, 128", "dw2d_bf16_kernelILi256ELi128", "dw2d_bf16_kernelILi128ELi256",
               "gat_aggregate_bwd_dst", "gat_aggregate_bwd_src", "gat_aggregate_fwd_kernel", "mean_residual_relu", "mean_bwd"]
    stats = os.path.join(d, "large_bf16_kernel_stats.csv")
    out = {}
    for k in kernels:
        tmp = os.path.join(d, "x.json")
        subprocess.run([sys.executable, os.path.join(ROOT, "tests", "micro", "summarize_prof.py"), "pmc", tmp, k,
                        os.path.join(d, "la"), os.path.join(d, "lb"), os.path.join(d, "lc"), os.path.join(d, "ld")], check=True)
        c = json.load(open(tmp))["counters"]
        if not c:
            continue
        e = {n: round(v["mean_per_launch"]) for n, v in c.items()}
        e["launches_counted"] = max(v["launches"] for v in c.values())
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            e["hbm_side_bytes_per_launch"] = int((2 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024)
        us = kernel_avg_us(stats, k)
        if us:
            e["avg_us_kernel_trace"] = round(us, 2)
            e["avg_us"] = round(us, 2)                # (bench.py: hbm_side_rates)
            if "hbm_side_bytes_per_launch" in e:
                e["hbm_side_gbs"] = round(e["hbm_side_bytes_per_launch"] / us * 1e-3, 1)
        if e.get("SQ_BUSY_CYCLES") and "SQ_VALU_MFMA_BUSY_CYCLES" in e:
            # SQ_BUSY_CYCLES: quad-cycles with a wave on the shader engine, summed over the XCDs' SQs; MFMA busy is per SIMD
            # (4 per CU, 256 CUs).  Utilisation = MFMA-busy SIMD-cycles / (kernel cycles x 1024 SIMDs), from the kernel's
            # GRBM_GUI_ACTIVE (cycles the GPU was busy with it).
            if e.get("GRBM_GUI_ACTIVE"):
                # GRBM_GUI_ACTIVE is summed over the 8 XCDs by summarize_prof (one row per XCD)
                cyc = e["GRBM_GUI_ACTIVE"] / 8.0
                e["mfma_pipe_utilisation"] = round(e["SQ_VALU_MFMA_BUSY_CYCLES"] * 4.0 / (cyc * 1024), 4)
        if e.get("SQ_WAVE_CYCLES"):
            e["wait_any_ratio"] = round(e.get("SQ_WAIT_ANY", 0) / e["SQ_WAVE_CYCLES"], 3)
            e["active_inst_ratio"] = round(e.get("SQ_ACTIVE_INST_ANY", 0) / e["SQ_WAVE_CYCLES"], 3)
        out[k] = e
    res = {"workload": "gatres_large (25 x 128), C-Town-sized batch of 128 snapshots, bf16, per-op + blocked kernels (round 6)",
           "units": "FETCH_SIZE / WRITE_SIZE in KB (bytes = (2 x FETCH + WRITE) x 1024 on gfx950); SQ_* cycle counters in "
                    "quad-cycles, summed over the chip; mfma_pipe_utilisation = SQ_VALU_MFMA_BUSY_CYCLES x 4 / (GPU-busy "
                    "cycles x 1024 SIMDs)",
           "command": "tests/micro/profile_r06.sh (separate rocprofv3 --pmc passes, the program directly behind `--`)",
           "mean_per_launch": out}
    for path in (os.path.join(d, "large_bf16_pmc.json"), os.path.join(ROOT, "profiles", "r06_large_ctown_bs128_bf16_pmc.json")):
        with open(path, "w") as f:
            json.dump(res, f, indent=1)
    print(json.dumps({k: {n: v[n] for n in ("avg_us_kernel_trace", "hbm_side_gbs", "mfma_pipe_utilisation", "wait_any_ratio") if n in v} for k, v in out.items()}))


def publish(d):
    """HERE (after the gpurun call merged its files back): gpurun_out/r06_final/* -> profiles/r06_* under the names the documents
    cite."""
    import shutil
    same = ["batch_scaling.txt", "bench_driver_protocol.json", "bench_n1.json", "collective_path_1rank.json",
            "collective_path_1rank_eager.json", "collective_path_1rank_eager_2buckets.json", "collective_path_kernel_stats.csv",
            "eager_kernel_stats.csv", "eager.json", "from_store.json", "fused_kernel_stats.csv", "fused_pmc.json",
            "fused_pmc_raw.json", "host_batches.json", "large_50k_bs2_bf16.json", "launcher_2ranks_one_gpu.json",
            "one_step_per_graph.json", "param_grads_pmc.json", "per_op.json", "shuffle_nodes.json", "small_50k_bs16.json",
            "drop_in_breakdown.txt", "eval_large_bs128_bf16.json", "eval_small_bs32.json", "copy_batches.json",
            "eval_kernel_stats.csv"]
    renamed = {"large_bf16.json": "large_ctown_bs128_bf16.json", "large_bf16_kernel_stats.csv": "large_ctown_bs128_bf16_kernel_stats.csv",
               "large_bf16_pmc.json": "large_ctown_bs128_bf16_pmc.json", "large_fp32.json": "large_ctown_bs128_fp32.json",
               "stage_times.txt": "window_stage_times.txt", "drop_in.json": "drop_in_torch_adam.json",
               "drop_in--flat-adam.json": "drop_in_flat_adam.json", "drop_in--fused-adam.json": "drop_in_fused_adam.json",
               "phase_ab.txt": "window_instantiations_ab.txt", "sync_start_ab.txt": "window_sync_start_ab.txt"}
    for src, dst in [(f, f) for f in same] + list(renamed.items()):
        a = os.path.join(d, src)
        if os.path.exists(a) and os.path.getsize(a) > 0:
            shutil.copyfile(a, os.path.join(ROOT, "profiles", "r06_" + dst))
        else:
            print("missing:", a)
    pr = os.path.join(ROOT, "gpurun_out", "parity_report.json")
    if os.path.exists(pr):
        shutil.copyfile(pr, os.path.join(ROOT, "profiles", "r06_parity_report.json"))


if __name__ == "__main__":
    {"pmc_raw": pmc_raw, "large_pmc": large_pmc, "publish": publish}[sys.argv[1]](sys.argv[2])
