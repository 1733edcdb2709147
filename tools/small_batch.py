"""Latency of one drift evaluation (and one divergence) at the reference configs' batch sizes (mdqm9/config/ambient/*.json)."""
import importlib, sys, time
import numpy as np
sys.path.insert(0, '.')
ti = importlib.import_module("thermodynamic-interpolation_amd")
syn, W = ti.synthetic, ti.weights
import torch
for A, B, F in ((9, 12, 128), (9, 512, 128), (25, 64, 256), (18, 256, 128), (18, 4096, 128)):
    src, dst, et = syn.fully_connected_template(A)
    flat = W.flatten_state_dict(syn.painn_state_dict(0, F, 5, 25, 0), W.painn_param_spec(0, F, 5, 25))
    for prec in ("f32", "f16x2"):
        eng = ti.engine.PainnEngine(0, F, 5, A, src, dst, et, np.arange(A), flat, temp_length=100.0, precision=prec)
        x = torch.from_numpy(syn.molecule_coords(B, A, 0)).cuda(); c = torch.from_numpy(syn.ambient_cond(B, A)).cuda()
        out = torch.empty_like(x)
        eng.drift(x, 0.5, c, out=out); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): eng.drift(x, 0.5, c, out=out)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        eng.drift_div(x, 0.5, c); torch.cuda.synchronize()
        t0 = time.perf_counter(); eng.drift_div(x, 0.5, c); torch.cuda.synchronize(); dd = time.perf_counter() - t0
        print(f"A={A} B={B} F={F} {prec}: drift {dt*1e3:.2f} ms ({B/dt:.0f} mol/s)   drift+div {dd*1e3:.1f} ms", flush=True)
        eng.close()
