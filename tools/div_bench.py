#!/usr/bin/env python3
"""Exact divergence (3A forward-mode directions per molecule) of the headline shape, torch-free: per-kernel HIP-event times and
molecule-divergences/s.   python tools/div_bench.py [molecules=2048] [repeats=3] [precision=f16x2]
TI_LIB_PATH selects an experiment build (tools/variant_bench.py build).  Used under rocprofv3 for profiles/r03*_divergence_*."""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    prec = sys.argv[3] if len(sys.argv) > 3 else "f16x2"
    ti = importlib.import_module("thermodynamic-interpolation_amd")
    syn, W, E = ti.synthetic, ti.weights, ti.engine
    F, L, A = 128, 5, 18
    tpl = syn.fully_connected_template(A)
    flat = W.flatten_state_dict(syn.painn_state_dict(W.AMBIENT, F, L, 25, 0), W.painn_param_spec(W.AMBIENT, F, L, 25))
    eng = E.PainnEngine(W.AMBIENT, F, L, A, *tpl, np.arange(A), flat, temp_length=100.0, precision=prec)
    x, cond = syn.molecule_coords(B, A, seed=0), syn.ambient_cond(B, A)
    b, div = eng.drift_div(x, 0.5, cond)                   # warm-up (workspace allocation)
    eng.profile(True)
    t0 = time.perf_counter()
    for _ in range(reps):
        b, div = eng.drift_div(x, 0.5, cond)
    dt = (time.perf_counter() - t0) / reps
    eng.profile(False)
    kern = {}
    for k in ("painn_jvp_edge", "painn_jvp_update", "painn_jvp_readout", "painn_jvp_filter", "painn_edge", "painn_update", "painn_embed", "painn_readout"):
        n, ms = eng.profile_read(k)
        kern[k] = {"launches_per_eval": n / reps, "ms_per_eval": ms / reps, "avg_launch_ms": ms / max(n, 1)}
    print(json.dumps({"workload": f"exact divergence, {B} molecules x 54 directions, F=128 L=5 A=18", "precision": prec,
                      "molecule_divergences_per_s": B / dt, "ms_per_evaluation": dt * 1e3, "finite": bool(np.isfinite(div).all()),
                      "div_checksum": float(np.abs(div).sum()), "kernels": kern}))


if __name__ == "__main__":
    main()
