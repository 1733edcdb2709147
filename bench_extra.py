#!/usr/bin/env python3
"""Secondary measurements (not the driver's contract line): adw double well (BASELINE config 2), mdqm9 latent (config 3),
other molecule sizes / feature widths.  One JSON line per workload.  Needs a GPU."""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--which", default="adw,latent,a9,a25,f256,div,dopri5,cfg5")
    ap.add_argument("--steps", type=int, default=5)
    args = ap.parse_args()
    import torch
    ti = importlib.import_module("thermodynamic-interpolation_amd")
    syn, W = ti.synthetic, ti.weights
    dev = torch.device("cuda", 0)
    which = args.which.split(",")

    def timed(fn):
        fn(1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(args.steps)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.steps

    if "adw" in which:
        # config 2: 262 144 particles, H=256, 5 layers, Euler-Maruyama; algorithmic 659 456 FLOP per particle-eval
        B = 262144
        flat = W.flatten_state_dict(syn.adw_state_dict(256, 5, 0), W.adw_param_spec(256, 5), dtype=np.float64)
        x0 = torch.from_numpy(syn.adw_x0(B, 0)).to(dev)
        b0 = torch.full((B,), 1.0, device=dev)
        b1 = torch.full((B,), 1.25, device=dev)
        grid = ti.engine.time_grid(0.0, 1.0, 1001)
        out = torch.empty((1, B), device=dev)
        for prec in ("f16x2", "f32"):
            eng = ti.engine.AdwEngine(256, 5, flat, precision=prec)
            dt = timed(lambda k: eng.rollout(x0, b0, b1, grid[: k + 1], scheme="em", eps=0.01, seed=1, save_every=0, out=out))
            dtd = timed(lambda k: eng.rollout(x0, b0, b1, grid[: k + 1], scheme="euler", save_every=0, out=out, return_dlogp=True))
            print(json.dumps({"workload": "adw double well (config 2): 262144 particles, H=256 x 5 layers", "precision": prec,
                              "em_particle_steps_per_s": B / dt, "em_ms_per_step": dt * 1e3, "algorithmic_tflops": 659456 * B / dt / 1e12,
                              "euler_with_dlogp_particle_steps_per_s": B / dtd, "euler_with_dlogp_ms_per_step": dtd * 1e3}))
            eng.close()

    def painn(tag, variant, F, L, A, B, precision, cond_fn, temp_length):
        src, dst, et = syn.fully_connected_template(A)
        flat = W.flatten_state_dict(syn.painn_state_dict(variant, F, L, 25, 0), W.painn_param_spec(variant, F, L, 25))
        eng = ti.engine.PainnEngine(variant, F, L, A, src, dst, et, np.arange(A), flat, temp_length=temp_length, precision=precision)
        x0 = torch.from_numpy(syn.molecule_coords(B, A, 0, 1.0 if variant else 0.3)).to(dev)
        c = cond_fn(B, A)
        cond = None if c is None else torch.from_numpy(c).to(dev)
        grid = ti.engine.time_grid(0.0, 1.0, 501)
        out = torch.empty((1, B, A, 3), device=dev)
        dt = timed(lambda k: eng.rollout(x0, cond, grid[: k + 1], scheme="euler", save_every=0, out=out))
        nE = W.N_EMBED[variant]
        flop = F * F * (L * (30 * A * (A - 1) + 24 * A) + (2 * nE + 4) * A + 4 * A) + 10 * F * A
        print(json.dumps({"workload": tag, "precision": precision, "trajectory_steps_per_s": B / dt, "ms_per_step": dt * 1e3,
                          "algorithmic_tflops": flop * B / dt / 1e12}))
        eng.close()

    if "cfg5" in which:
        # BASELINE.json configs[4]: ambient sampler, 1 048 576 molecules over 8 GPUs = 131 072 per GPU, "fp16 node features with MFMA
        # linears, roofline report": precision="f16" (state tensors fp16 in HBM, one fp16 product per k-step, fp32 accumulation;
        # DESIGN.md 3.4).  A separately labelled precision: its drift error against the reference is reported in the same line.
        F, L, A, B = 128, 5, 18, int(os.environ.get("TI_BENCH_CFG5_B", "131072"))
        tpl = syn.fully_connected_template(A)
        flat = W.flatten_state_dict(syn.painn_state_dict(0, F, L, 25, 0), W.painn_param_spec(0, F, L, 25))
        x0 = torch.from_numpy(syn.molecule_coords(B, A, 0)).to(dev)
        cond = torch.from_numpy(syn.ambient_cond(B, A)).to(dev)
        grid = ti.engine.time_grid(0.0, 1.0, 2001)
        out = torch.empty((1, B, A, 3), device=dev)
        rel = lambda a, b: float(np.linalg.norm(np.asarray(a, np.float64) - b) / np.linalg.norm(b))
        for prec in ("f16", "f16x2"):
            eng = ti.engine.PainnEngine(0, F, L, A, *tpl, np.arange(A), flat, temp_length=100.0, precision=prec)
            eng.reserve(B)
            dt = timed(lambda k: eng.rollout(x0, cond, grid[: k + 1], scheme="em", eps=0.01, seed=1, save_every=0, out=out))
            eng.profile(True)
            eng.rollout(x0, cond, grid[: args.steps + 1], scheme="em", eps=0.01, seed=1, save_every=0, out=out)
            prof = {k: eng.profile_read(k) for k in ("painn_edge", "painn_update", "painn_embed", "painn_readout")}
            eng.profile(False)
            edge_ms = prof["painn_edge"][1] / max(prof["painn_edge"][0], 1)
            achieved = B * A * (A - 1) * 30 * F * F / (edge_ms * 1e-3) / 1e12
            parity = None
            gp = os.path.join(ROOT, "tests", "golden", "ambient_full.npz")          # the reference's own drift for exactly these weights
            if os.path.exists(gp):
                with np.load(gp) as g:
                    parity = [rel(eng.drift(g["x"], float(t), g["cond"]), g[f"drift_{i}"].astype(np.float64)) for i, t in enumerate(g["ts"])]
            flop = F * F * (L * (30 * A * (A - 1) + 24 * A) + (2 * W.N_EMBED[0] + 4) * A + 4 * A) + 10 * F * A
            print(json.dumps({"workload": f"mdqm9 ambient (config 5 per-GPU share): {B} molecules x 18 atoms, F=128 L=5, Euler-Maruyama step",
                              "precision": prec, "trajectory_steps_per_s": B / dt, "ms_per_step": dt * 1e3, "algorithmic_tflops": flop * B / dt / 1e12,
                              "kernel_ms_per_launch": {k: round(v[1] / max(v[0], 1), 3) for k, v in prof.items()},
                              "drift_rel_l2_vs_reference_pytorch_cpu": parity,
                              "roofline": {"kernel": "painn_edge_kernel", "bound": "mfma", "achieved": achieved, "peak": 2500.0, "unit": "TFLOP/s",
                                           "frac": achieved / 2500.0, "traffic": None,
                                           "algorithmic_flop_per_launch": B * A * (A - 1) * 30 * F * F,
                                           "products_per_algorithmic_product": 1 if prec == "f16" else 3}}))
            eng.close()

    if "div" in which:
        # exact divergence: drift + trace of the Jacobian, 3A = 54 forward-mode directions per molecule.
        # Algorithmic work per molecule ~ (1 + 2 * 54 * (L-1)/L ... ) is quoted as the measured ratio to a plain drift instead.
        F, L, A, B = 128, 5, 18, int(os.environ.get("TI_BENCH_DIV_B", "2048"))
        src, dst, et = syn.fully_connected_template(A)
        flat = W.flatten_state_dict(syn.painn_state_dict(0, F, L, 25, 0), W.painn_param_spec(0, F, L, 25))
        x0 = torch.from_numpy(syn.molecule_coords(B, A, 0)).to(dev)
        cond = torch.from_numpy(syn.ambient_cond(B, A)).to(dev)
        for prec in ("f16x2", "f32"):
            eng = ti.engine.PainnEngine(0, F, L, A, src, dst, et, np.arange(A), flat, temp_length=100.0, precision=prec)
            dt_div = timed(lambda k: [eng.drift_div(x0, 0.5, cond) for _ in range(k)])
            dt_b = timed(lambda k: [eng.drift(x0, 0.5, cond) for _ in range(k)])
            eng.profile(True)
            eng.drift_div(x0, 0.5, cond)
            prof = {k: eng.profile_read(k) for k in ("painn_jvp_filter", "painn_jvp_edge", "painn_jvp_update", "painn_jvp_readout", "painn_edge", "painn_update")}
            eng.profile(False)
            rec = {"workload": f"ambient drift + exact divergence: {B} molecules x 18 atoms, F=128 L=5 (54 tangent directions each)",
                   "precision": prec, "molecule_div_evals_per_s": B / dt_div, "ms_per_eval": dt_div * 1e3,
                   "cost_ratio_vs_plain_drift": dt_div / dt_b,
                   "kernel_ms": {k: round(v[1], 3) for k, v in prof.items()}, "kernel_launches": {k: v[0] for k, v in prof.items()}}
            print(json.dumps(rec))
            eng.close()

    if "dopri5" in which:
        # the reference's default solver on the headline batch: adaptive Dormand-Prince, rtol = atol = 1e-4 (shipped configs),
        # one shared step size per batch like torchdiffeq; reports evaluations taken and wall time for t = 0 -> 1
        F, L, A, B = 128, 5, 18, 65536
        src, dst, et = syn.fully_connected_template(A)
        flat = W.flatten_state_dict(syn.painn_state_dict(0, F, L, 25, 0), W.painn_param_spec(0, F, L, 25))
        x0 = torch.from_numpy(syn.molecule_coords(B, A, 0)).to(dev)
        cond = torch.from_numpy(syn.ambient_cond(B, A)).to(dev)
        grid = ti.engine.time_grid(0.0, 1.0, 100)
        out = torch.empty((1, B, A, 3), device=dev)
        for prec in ("f16x2", "f32"):
            eng = ti.engine.PainnEngine(0, F, L, A, src, dst, et, np.arange(A), flat, temp_length=100.0, precision=prec)
            eng.rollout(x0, cond, grid, scheme="dopri5", rtol=1e-4, atol=1e-4, save_every=0, out=out)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _, nfe = eng.rollout(x0, cond, grid, scheme="dopri5", rtol=1e-4, atol=1e-4, save_every=0, out=out)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(json.dumps({"workload": "ambient dopri5 rollout t=0..1 (100-point output grid), 65536 molecules x 18 atoms, F=128 L=5, rtol=atol=1e-4",
                              "precision": prec, "seconds": dt, "drift_evaluations": nfe, "molecule_evals_per_s": B * nfe / dt}))
            eng.close()

    for prec in ("f16x2", "f32"):
        if "latent" in which:
            painn("mdqm9 latent (config 3): 65536 molecules x 18 atoms, F=128 L=5, Euler ODE step", W.LATENT_MULTI, 128, 5, 18, 65536, prec,
                  lambda B, A: syn.latent_cond(B, A, 800.0), 75.0)
        if "a9" in which:
            painn("ambient, molecule 00031 shape: 65536 x 9 atoms, F=128 L=5", W.AMBIENT, 128, 5, 9, 65536, prec, syn.ambient_cond, 100.0)
        if "a25" in which:
            painn("ambient, molecule 10506 shape: 16384 x 25 atoms, F=128 L=5", W.AMBIENT, 128, 5, 25, 16384, prec, syn.ambient_cond, 100.0)
        if "f256" in which:
            painn("ambient, 10506 config: 8192 x 25 atoms, F=256 L=5", W.AMBIENT, 256, 5, 25, 8192, prec, syn.ambient_cond, 100.0)


if __name__ == "__main__":
    main()
