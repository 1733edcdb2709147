import importlib, sys, os, numpy as np
sys.path.insert(0, '.')
ti = importlib.import_module("thermodynamic-interpolation_amd")
syn, W = ti.synthetic, ti.weights
F, L, A, B = 256, 2, 5, 3
src, dst, et = syn.fully_connected_template(A)
flat = W.flatten_state_dict(syn.painn_state_dict(0, F, L, 25, seed=F + A), W.painn_param_spec(0, F, L, 25))
x = syn.molecule_coords(B, A, seed=B); cond = syn.ambient_cond(B, A)
ref = ti.engine.PainnEngine(0, F, L, A, src, dst, et, np.arange(A), flat, temp_length=100.0, precision="f32").drift(x, 0.37, cond)
print("f32 ok", flush=True)
eng = ti.engine.PainnEngine(0, F, L, A, src, dst, et, np.arange(A), flat, temp_length=100.0, precision="f16x2")
print("created", flush=True)
for stage in (1, 2, 3, 4):
    eng.debug_tap(stage); eng.drift(x, 0.37, cond); print("stage", stage, "ok", flush=True)
eng.debug_tap(-1)
d = eng.drift(x, 0.37, cond)
print("drift ok rel", np.linalg.norm(d - ref) / np.linalg.norm(ref), flush=True)
