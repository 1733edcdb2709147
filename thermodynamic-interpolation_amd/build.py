"""Builds libti_hip.so (gfx950) in-tree with hipcc.  `python -m` style entry: build(force=False, jobs=3)."""
from __future__ import annotations

import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "libti_hip.so")
SOURCES = ["ti_api.hip", "painn_kernels.hip", "painn_edge_nb1.hip", "painn_edge_nb2.hip", "painn_edge_nb4.hip", "painn_edge_nb8.hip",
           "painn_pair_nb1.hip", "painn_pair_nb2.hip", "painn_pair_nb4.hip", "painn_jvp_kernels.hip", "adw_kernels.hip", "ode_kernels.hip"]
HEADERS = ["mfma_chain.hpp", "ti_internal.hpp", "painn_edge_kernel.hpp", "painn_pair_kernel.hpp", "pair_template.hpp", os.path.join("..", "..", "include", "ti_hip.h")]
# -packed-fp32-ops off: v_pk_{fma,mul,add}_f32 do NOT run next to another wave's matrix instructions on gfx950 (a wave of them and a
# wave of 16x16x32 fp16 MFMAs on one SIMD take the SUM of their times; scalar v_fma_f32, conversions and transcendentals overlap:
# tools/micro/coexec_classes.hip, profiles/r03i_microbench_coexec.txt), and register pairs cost the message kernels registers (pair
# kernel: 19 -> 4 spilled).  Measured, same box: pair message kernel -2.0 %, directed -0.6 %, divergence workload +1.2 %
# (profiles/r03i_nopk_timing.txt); results move in the last bit (fma contraction), parity unchanged.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
# Per-source flags.  painn_edge_nb8.hip (F = 256, one wave per SIMD): hipcc (ROCm 7.2) spills 28 SGPRs of the one-accumulator message
# kernel into lanes of a VGPR, and that build faults on the device in its first launch (HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION,
# profiles/r02h_f256_one_chain_fault.txt); the same source with the lane spills switched off allocates them to registers (no scratch) and
# passes every stage at 1.6e-6 (profiles/r03e_f256_one_chain_no_lane_spill.txt).  DESIGN.md 3.4.
EXTRA_FLAGS = {"painn_edge_nb8.hip": ["-mllvm", "-amdgpu-spill-sgpr-to-vgpr=0"]}


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found (need ROCm to build libti_hip.so)")
    return exe


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, jobs: int = 8, verbose: bool = False) -> str:
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    if not force and not _stale(SO, deps):
        return SO
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    cc = hipcc()

    def compile_one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        if force or _stale(obj, [os.path.join(CSRC, src)] + [os.path.join(CSRC, h) for h in HEADERS]):
            cmd = [cc, *FLAGS, *EXTRA_FLAGS.get(src, []), "-c", os.path.join(CSRC, src), "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=jobs) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    subprocess.check_call([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO, *objs])
    return SO


if __name__ == "__main__":
    print(build(verbose=True))
