"""Where one small-batch drift evaluation spends its time: wall per evaluation (async launches, one sync) and per-kernel HIP-event
durations, at the batch sizes of the reference's shipped configs."""
import importlib, sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ti = importlib.import_module("thermodynamic-interpolation_amd")
syn, W = ti.synthetic, ti.weights
import torch
shapes = ((9, 12, 128), (9, 512, 128), (18, 256, 128))
for A, B, F in shapes:
    src, dst, et = syn.fully_connected_template(A)
    flat = W.flatten_state_dict(syn.painn_state_dict(0, F, 5, 25, 0), W.painn_param_spec(0, F, 5, 25))
    for prec in ("f32", "f16x2", "f16"):
        eng = ti.engine.PainnEngine(0, F, 5, A, src, dst, et, np.arange(A), flat, temp_length=100.0, precision=prec)
        x = torch.from_numpy(syn.molecule_coords(B, A, 0)).cuda(); c = torch.from_numpy(syn.ambient_cond(B, A)).cuda()
        out = torch.empty_like(x)
        for _ in range(3): eng.drift(x, 0.5, c, out=out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): eng.drift(x, 0.5, c, out=out)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        grid = ti.engine.time_grid(0.0, 1.0, 101)
        o2 = torch.empty((1,) + tuple(x.shape), device="cuda")
        eng.rollout(x, c, grid[:3], scheme="euler", save_every=0, out=o2); torch.cuda.synchronize()
        t0 = time.perf_counter(); eng.rollout(x, c, grid[:41], scheme="euler", save_every=0, out=o2); torch.cuda.synchronize()
        dr = (time.perf_counter() - t0) / 40
        eng.profile(True)
        eng.drift(x, 0.5, c, out=out)
        prof = {k: eng.profile_read(k) for k in ("painn_embed", "painn_edge", "painn_update", "painn_readout")}
        eng.profile(False)
        ks = "  ".join(f"{k[6:]} {v[0]}x {1e3 * v[1] / max(v[0], 1):.0f}us" for k, v in prof.items())
        print(f"A={A} B={B} F={F} {prec:5s}: drift {dt*1e3:.3f} ms   Euler step in a rollout {dr*1e3:.3f} ms   kernels: {ks}  (sum {sum(v[1] for v in prof.values()):.3f} ms)", flush=True)
        eng.close()
