#!/usr/bin/env python3
"""bench.py -- integration-steps/sec of the mdqm9 ambient sampler hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json north_star / configs[3] shape): 65 536 trajectories per GPU of an 18-atom fully connected
molecule, ambient cPaiNN drift F=128 / L=5 (random-init weights of that architecture, synthetic coordinates), T0 = 1000 K,
T1 round-robin over the 6-rung ladder, Euler-Maruyama steps (eps = 0.01) on the reference grid.  A "step" is one
integrator step of every trajectory of the batch = one drift evaluation + the fused state update.  Trajectories are
independent, so ranks shard them with no data-path collective ("weak" scaling: per-GPU batch fixed); the only
collective is the final RCCL all-gather of the end states, which is inside the timed region.

Matrix path: `--precision f16x2` (default) runs the message/update MLP products on the fp16 matrix cores with every fp32
operand split into two fp16 halves (three products, fp32 accumulation); its drift parity against the reference equals the
f32-MFMA path's (tests/test_gpu_parity.py, profiles/*split_fp16_parity.txt).  `--precision f32` uses v_mfma_f32_16x16x4_f32
only; by default its throughput is measured in the same run and reported as `f32_mfma_path` next to the headline.

Inputs are resident in HBM before the timed region starts.  Prints ONE JSON line on rank 0.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F, L, A = 128, 5, 18
E_M = A * (A - 1)
FLOP_PER_MOL_EVAL = F * F * (L * (30 * E_M + 24 * A) + (2 * 4 + 4) * A + 4 * A) + 10 * F * A      # 7.9216e8
FLOP_PER_EDGE_LAYER = 30 * F * F
PEAK_F32_MFMA_TFLOPS = 157.3                                                                      # MI355X_MICROARCH.md
PEAK_F16_MFMA_TFLOPS = 2500.0                                                                     # dense fp16/bf16 MFMA, same guide


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=65536, help="trajectories per GPU")
    ap.add_argument("--eps", type=float, default=0.01)
    ap.add_argument("--precision", default="f16x2", choices=["f32", "f16x2"], help="matrix path of the message/update MLPs (DESIGN.md §3.4)")
    ap.add_argument("--no-f32-leg", action="store_true", help="skip the additional f32-MFMA measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the drift rel-L2 leg (reference fixture + 64-molecule oracle sample)")
    ap.add_argument("--cpu-sample", type=int, default=512, help="molecules in the CPU-oracle sample")
    return ap.parse_args()


def cpu_baseline(ti, flat, template, sample_mols, steps=4):
    """The CPU oracle (oracle/ti_oracle.c, OpenMP) on a bounded sample of the same workload; rank 0, N=1 only."""
    from oracle import oracle
    src, dst, et = template
    orc = oracle.PainnOracle(ti.weights.AMBIENT, F, L, A, src, dst, et, np.arange(A), flat, temp_length=100.0)
    x, cond = ti.synthetic.molecule_coords(sample_mols, A, 1), ti.synthetic.ambient_cond(sample_mols, A)
    grid = ti.engine.time_grid(0.0, 1.0, 1001)[:steps + 1]
    orc.drift(x[:16], 0.0, cond[:16])                     # warm-up (thread pool, page-in)
    t0 = time.perf_counter()
    orc.rollout(x, cond, grid, scheme="em", save_every=0, eps=0.01, seed=0)
    dt = time.perf_counter() - t0
    return {"value": sample_mols * steps / dt, "unit": "integration-steps/s", "cores": oracle.num_threads(), "kind": "port",
            "sample": f"{sample_mols} molecules x {steps} Euler-Maruyama steps, same F/L/A/graph, OpenMP over molecules, {dt:.1f} s"}


def kernel_sources_sha():
    """Hash of the HIP sources: what the PMC traffic figure is keyed by, so a figure measured on other kernels is not reported."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "thermodynamic-interpolation_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp")):
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def drift_rel_l2(ti, eng, flat, template, n_oracle=64):
    """The second half of BASELINE.json's metric: drift rel-L2 of THIS engine (bench weights, bench precision) against the
    reference's own CPU output -- tests/golden/ambient_full.npz holds the reference PyTorch drift of 3 molecules for exactly
    these weights (seed 0, F=128, L=5, A=18) at t = 0, 0.25, 1 and a 10-step Euler trajectory -- and, over 64 molecules at
    t = 0, 0.5, 1, against the CPU oracle (the restatement pinned to that reference, oracle/)."""
    from oracle import oracle
    out = {}
    gp = os.path.join(ROOT, "tests", "golden", "ambient_full.npz")
    rel = lambda a, b: float(np.linalg.norm(np.asarray(a, np.float64) - b) / np.linalg.norm(b))
    if os.path.exists(gp):
        with np.load(gp) as g:
            errs = [rel(eng.drift(g["x"], float(t), g["cond"]), g[f"drift_{i}"].astype(np.float64)) for i, t in enumerate(g["ts"])]
            out["vs_reference_pytorch_cpu"] = {"t": [float(t) for t in g["ts"]], "rel_l2": errs, "molecules": int(g["B"])}
            path, _ = eng.rollout(g["x"], g["cond"], g["traj_grid"], scheme="euler", save_every=1)
            ref = g["traj_euler"].astype(np.float64)
            out["vs_reference_pytorch_cpu"]["euler_10_steps_displacement_rel_l2"] = rel(path - path[0], ref - ref[0])
    src, dst, et = template
    orc = oracle.PainnOracle(ti.weights.AMBIENT, F, L, A, src, dst, et, np.arange(A), flat, temp_length=100.0)
    x, cond = ti.synthetic.molecule_coords(n_oracle, A, 5), ti.synthetic.ambient_cond(n_oracle, A)
    ts = [0.0, 0.5, 1.0]
    out["vs_cpu_oracle"] = {"t": ts, "rel_l2": [rel(eng.drift(x, t, cond), orc.drift(x, t, cond, precision=64)) for t in ts], "molecules": n_oracle,
                            "oracle_arithmetic": "fp64"}
    return out


def main():
    args = parse()
    import torch
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    dist = None
    # TI_BENCH_TEST_SHARED_GPU=1 (rehearsal on a one-GPU box only): every rank uses cuda:0 and the gather goes through gloo on
    # host copies, so the rank / shard / gather logic of the N > 1 path can be exercised without N GPUs.  Never a measurement.
    shared_gpu_test = bool(os.environ.get("TI_BENCH_TEST_SHARED_GPU"))
    if shared_gpu_test:
        local_rank = 0
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if shared_gpu_test:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the sampling path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    ti = importlib.import_module("thermodynamic-interpolation_amd")
    syn, W = ti.synthetic, ti.weights
    template = syn.fully_connected_template(A)
    flat = W.flatten_state_dict(syn.painn_state_dict(W.AMBIENT, F, L, 25, 0), W.painn_param_spec(W.AMBIENT, F, L, 25))
    B = args.batch
    # synthetic inputs, resident in HBM; rank r owns global trajectories [r*B, (r+1)*B)
    x0 = torch.from_numpy(syn.molecule_coords(B, A, seed=rank)).to(dev)
    cond = torch.from_numpy(syn.ambient_cond(B, A)).to(dev)
    out = torch.empty((1, B, A, 3), dtype=torch.float32, device=dev)
    grid = ti.engine.time_grid(0.0, 1.0, 1001)            # config 4: 1000-step grid; we time K of its steps
    gdev = "cpu" if shared_gpu_test else dev
    gathered = [torch.empty((B, A, 3), dtype=torch.float32, device=gdev) for _ in range(world)] if world > 1 else None

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(precision):
        """W warm-up steps, then exactly K timed steps bracketed by barrier + synchronize; max over ranks."""
        eng = ti.engine.PainnEngine(W.AMBIENT, F, L, A, *template, np.arange(A), flat, temp_length=100.0, device=local_rank,
                                    precision=precision)
        eng.reserve(B)

        def run(k_steps, first_step):
            eng.rollout(x0, cond, grid[first_step:first_step + k_steps + 1], scheme="em", eps=args.eps, seed=1234,
                        traj_offset=rank * B, save_every=0, out=out, step_offset=first_step)
            if world > 1:
                dist.all_gather(gathered, out[0].cpu() if shared_gpu_test else out[0])   # the only collective: final gather (RCCL)

        if args.warmup > 0:
            run(args.warmup, 0)
        sync()
        eng.profile(True)
        t0 = time.perf_counter()
        run(args.steps, args.warmup)
        sync()
        dt = time.perf_counter() - t0
        eng.profile(False)
        if world > 1:
            tmax = torch.tensor([dt], dtype=torch.float64, device=gdev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        prof = {k: eng.profile_read(k) for k in ("painn_edge", "painn_update", "painn_embed", "painn_readout")}
        assert bool(torch.isfinite(out).all())
        parity = drift_rel_l2(ti, eng, flat, template) if rank == 0 and world == 1 and not args.no_parity else None
        layout = eng.template_for(B)
        eng.close()
        return dt, prof, parity, layout

    elapsed, prof, parity, layout = measure(args.precision)
    n_edge, ms_edge = prof["painn_edge"]
    n_upd, ms_upd = prof["painn_update"]
    f32_leg = None
    if args.precision != "f32" and not args.no_f32_leg:
        dt32, prof32, parity32, layout32 = measure("f32")
        f32_leg = {"value": world * B * args.steps / dt32, "ms_per_step": 1e3 * dt32 / args.steps,
                   "edge_kernel_avg_ms": prof32["painn_edge"][1] / max(prof32["painn_edge"][0], 1), "edge_row_layout": layout32,
                   "edge_kernel_frac_of_f32_mfma_peak": B * E_M * FLOP_PER_EDGE_LAYER / (prof32["painn_edge"][1] / max(prof32["painn_edge"][0], 1) * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS}
        if parity32:
            f32_leg["drift_rel_l2"] = parity32

    if rank == 0:
        value = world * B * args.steps / elapsed
        edge_ms = ms_edge / max(n_edge, 1)
        achieved = B * E_M * FLOP_PER_EDGE_LAYER / (edge_ms * 1e-3) / 1e12 if n_edge else None
        split = args.precision == "f16x2"
        peak = PEAK_F16_MFMA_TFLOPS if split else PEAK_F32_MFMA_TFLOPS
        # HBM bytes per edge-kernel launch from the separate rocprofv3 --pmc passes of this same workload (tools/gpu_prof.sh ->
        # tools/pmc_summary.py -> profiles/pmc_edge_traffic.json); used only if it was measured at this batch and precision ON THESE
        # KERNEL SOURCES (kernel_sources_sha): a figure of an older kernel is never carried over, the field is null instead
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_edge_traffic.json")) as f:
                pm = json.load(f)
            if pm.get("batch") == B and pm.get("precision") == args.precision and pm.get("kernel_sources_sha") == kernel_sources_sha():
                traffic = pm["read_bytes_per_launch"] + pm["write_bytes_per_launch"]
        except (OSError, ValueError, KeyError):
            pass
        rec = {
            "metric": "integration-steps/sec (whole node) + drift rel-L2 vs CPU ref", "value": value, "unit": "trajectory-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32 state and accumulation; matrix products on fp16 MFMA with 2-way split fp32 operands (hi + lo halves, 3 products)" if split else "f32", "data": "synthetic",
            "config": {"workload": f"mdqm9 ambient sampler: {B} molecules/GPU x 18 atoms (fully connected, 306 edges), cPaiNN F=128 L=5, "
                                   "Euler-Maruyama steps of the 1000-step grid, T1 over a 6-rung ladder",
                       "trajectories_per_gpu": B, "atoms": A, "n_features": F, "score_layers": L, "scheme": "em", "eps": args.eps,
                       "sharding": f"dp{world} over independent trajectories, final RCCL all-gather of end states"},
            "whole_step_tflops": FLOP_PER_MOL_EVAL * B * args.steps * world / elapsed / 1e12,
            "roofline": {"kernel": "painn_pair_kernel (pair-major message kernel: filter branch once per atom pair)" if layout == "pair" else "painn_edge_kernel", "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak if achieved else None, "traffic": traffic,
                         "peak_is": "dense fp16 MFMA (2.5 PF); the split path spends 3 fp16 products per algorithmic product, so its matrix-side ceiling is peak/3"
                                    if split else "f32 MFMA (157.3 TF)",
                         "launches": n_edge, "avg_launch_ms": edge_ms, "update_kernel_avg_ms": ms_upd / max(n_upd, 1),
                         "flops_per_launch": B * E_M * FLOP_PER_EDGE_LAYER},
        }
        if parity:
            rec["drift_rel_l2"] = parity
        if f32_leg:
            rec["f32_mfma_path"] = f32_leg
        if world == 1 and not args.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline(ti, flat, template, args.cpu_sample)
        print(json.dumps(rec), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
