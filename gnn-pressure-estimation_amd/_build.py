"""Build the gfx950 C-ABI library in-tree: csrc/*.hip -> lib/libgatres_hip.so (hipcc cross-compiles without a GPU).

The .so is git-ignored but travels with the repo snapshot to the GPU box.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_DIR = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_DIR = os.path.join(PKG_DIR, "lib")
OBJ_DIR = os.path.join(PKG_DIR, "build")
LIB_PATH = os.path.join(LIB_DIR, "libgatres_hip.so")
ARCH = "gfx950"

SOURCES = ["k_window.hip", "k_fused_whole.hip", "k_fused_host.hip", "k_aggregate.hip", "k_proj.hip", "k_misc.hip", "k_blocked.hip", "graph_plan.hip",
           "model_driver.hip", "train_driver.hip"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; the gfx950 library cannot be built")
    return exe


def _flags():
    extra = os.environ.get("GATRES_HIPCC_FLAGS", "").split()        # experiments only (e.g. -mllvm options)
    return ["-O3", f"--offload-arch={ARCH}", "-fPIC", "-std=c++17", "-ffp-contract=off", "-Wall",
            "-Wno-unused-function", f"-I{os.path.join(REPO_DIR, 'include')}", f"-I{CSRC}"] + extra


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


DIAG_LIB_PATH = os.path.join(LIB_DIR, "libgatres_hip_diag.so")


def source_id() -> str:
    """Hash of everything the library is compiled from (csrc/*.hip, csrc/*.h, include/gatres.h): the build id that
    ``gatres_version()`` reports and ``_native.load()`` checks, so a stale or foreign .so is never loaded silently."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h")))
    files.append(os.path.join(REPO_DIR, "include", "gatres.h"))
    for p in files:
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def build_native(force: bool = False, verbose: bool = False, diag: bool = False) -> str:
    """Compile (only what changed) and link; returns the path of the shared library.

    ``diag=True`` builds the DIAGNOSTIC library instead (lib/libgatres_hip_diag.so, -DGATRES_DIAG_BUILD): stage stamps
    (gatres_fused_set_stamps), the wrong-result switches GATRES_XCH_NOWAIT / GATRES_DIAG_NOMASK and the nc > 32
    instantiations of the per-snapshot kernels.  The product library carries none of them.  ``_native.load()`` takes
    the diagnostic library when GATRES_DIAG_LIB=1 (tests/stage_profile.py)."""
    os.makedirs(LIB_DIR, exist_ok=True)
    obj_dir = OBJ_DIR + ("_diag" if diag else "")
    lib_path = DIAG_LIB_PATH if diag else LIB_PATH
    os.makedirs(obj_dir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(REPO_DIR, "include", "gatres.h"))
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    hipcc = _hipcc()
    jobs = []
    objs = []
    bid = source_id()
    stamp = os.path.join(obj_dir, "BUILD_ID")
    try:
        with open(stamp) as f:
            same_id = f.read().strip() == bid
    except OSError:
        same_id = False
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(obj_dir, s.replace(".hip", ".o"))
        objs.append(obj)
        # model_driver.hip carries the build id (gatres_version): recompiled whenever any source changed
        if force or _stale(obj, [src] + headers) or (s == "model_driver.hip" and not same_id):
            jobs.append([hipcc] + _flags() + (["-DGATRES_DIAG_BUILD"] if diag else []) +
                        ([f'-DGATRES_BUILD_ID="{bid}"'] if s == "model_driver.hip" else []) + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr:
            print(r.stderr, file=sys.stderr)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if jobs or force or _stale(lib_path, objs):
        run([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", lib_path] + objs)
    with open(stamp, "w") as f:
        f.write(bid + "\n")
    return lib_path


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv, verbose=True, diag="--diag" in sys.argv))
