"""`bench.py --gpus 2` (and `--gpus 4`) END TO END on the one GPU a test box has: the launcher path the driver uses for its
scaling runs -- self_launch (a child ``torch.distributed.run``, started before anything touches the GPU), rank / local-rank
wiring, distinct per-rank seeds with a rank-0 broadcast, the barrier + MAX-over-ranks timing, one JSON line from rank 0 --
had never executed anywhere.  Two TEST-ONLY overrides make it runnable here: GATRES_DIST_BACKEND=gloo (RCCL wants one GPU
per rank) and GATRES_BENCH_SHARE_GPU=1 (both ranks on cuda:0); the line says so in config.test_overrides.

bench.py is started from the fork server's clean child: a process that has initialised HIP must not exec another program."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_bench(argv, env_extra, out_path):
    env = dict(os.environ, **env_extra)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, cwd=ROOT, capture_output=True,
                       text=True, timeout=900)
    with open(out_path, "w") as f:
        json.dump({"rc": r.returncode, "stdout": r.stdout, "stderr": r.stderr[-4000:]}, f)


def _launch(fork_ctx, tmp_path, argv, env_extra):
    out = str(tmp_path / "bench_out.json")
    p = fork_ctx.Process(target=_run_bench, args=(argv, env_extra, out))
    p.start()
    try:
        p.join(1000)
        assert p.exitcode == 0, f"launcher process: exit code {p.exitcode}"
    finally:
        if p.is_alive():
            p.terminate()
            p.join(10)
    with open(out) as f:
        res = json.load(f)
    assert res["rc"] == 0, res["stderr"]
    lines = [l for l in res["stdout"].splitlines() if l.strip()]
    assert lines, res["stderr"]
    line = json.loads(lines[-1])                     # the JSON line is the LAST line of stdout
    assert sum(1 for l in lines if l.lstrip().startswith("{") and '"metric"' in l) == 1      # ... and there is exactly one
    return line


@pytest.mark.parametrize("gpus,model_args,workload", [
    (2, ["--batch-size", "8"], "gatres_small"),                                     # fused path: the single-GPU launches around one all-reduce
    (2, ["--model", "gatres_large", "--batch-size", "2", "--dtype", "bf16"], "gatres_large"),  # per-op path: a bucket per block group
    (4, ["--batch-size", "4"], "gatres_small"),                                     # four ranks (round 5): 4 x 40 workgroups resident
], ids=["gatres_small_fused", "gatres_large_per_op", "gatres_small_fused_4_ranks"])
def test_bench_gpus_n_runs_end_to_end_on_one_gpu(fork_ctx, tmp_path, gpus, model_args, workload):
    line = _launch(fork_ctx, tmp_path,
                   ["--gpus", str(gpus), "--steps", "4", "--warmup", "2", "--repeats", "2", "--no-cpu-baseline", "--no-roofline"]
                   + model_args,
                   {"GATRES_DIST_BACKEND": "gloo", "GATRES_BENCH_SHARE_GPU": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    bs = int(model_args[model_args.index("--batch-size") + 1])
    assert line["n_gpus"] == gpus and line["steps"] == 4 and line["warmup"] == 2 and line["scaling"] == "weak"
    assert line["metric"] == "train snapshots/sec" and line["value"] > 0 and line["higher_is_better"] is True
    assert abs(line["value"] - gpus * bs * 4 / (line["ms_per_step"] * 4e-3)) < 1e-6 * line["value"]      # whole-job aggregate
    cfg = line["config"]
    assert workload in cfg["workload"] and "RCCL all-reduce" in cfg["workload"]      # the multi-rank sequence ran
    assert cfg["global_batch"] == gpus * bs and cfg["parallelism"] == f"dp{gpus}" and cfg["dropped_steps"] == 0
    assert cfg["test_overrides"] == {"backend": "gloo", "ranks_share_gpu": True}
    assert cfg["final_loss"] == cfg["final_loss"] and cfg["final_loss"] < 1e3


def test_bench_refuses_a_smaller_job_than_asked_for(fork_ctx, tmp_path):
    """`--gpus 2` with one visible GPU and no override: refuse, never report a 1-GPU run as a 2-GPU one."""
    out = str(tmp_path / "o.json")
    p = fork_ctx.Process(target=_run_bench, args=(["--gpus", "2", "--steps", "2", "--warmup", "1"], {}, out))
    p.start(); p.join(600)
    res = json.load(open(out))
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has two GPUs: the refusal does not apply")
    assert res["rc"] != 0 and "refusing to report a smaller job" in (res["stderr"] + res["stdout"])
