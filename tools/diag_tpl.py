import importlib, os, sys, numpy as np
sys.path.insert(0, '.')
ti = importlib.import_module("thermodynamic-interpolation_amd")
from oracle import oracle
syn, W = ti.synthetic, ti.weights
F, L, A, B, variant = 128, 2, 18, 50, 0
src, dst, et = syn.fully_connected_template(A)
flat = W.flatten_state_dict(syn.painn_state_dict(variant, F, L, 25, seed=F + A), W.painn_param_spec(variant, F, L, 25))
x = syn.molecule_coords(B, A, seed=B); cond = syn.ambient_cond(B, A)
orc = oracle.PainnOracle(variant, F, L, A, src, dst, et, np.arange(A), flat, temp_length=100.0)
grid = np.linspace(0, 1, 3).astype(np.float32)
p64, d64, _ = orc.rollout_dlogp(x, cond, grid, scheme="heun", save_every=0, precision=64)
p32, d32, _ = orc.rollout_dlogp(x, cond, grid, scheme="heun", save_every=0, precision=32)
print("oracle f32 vs f64 dlogp max abs", np.abs(d32 - d64).max(), "argmax", np.abs(d32 - d64).argmax())
# per-direction magnitudes for the worst molecule
for prec in ("f32", "f16x2"):
    eng = ti.engine.PainnEngine(variant, F, L, A, src, dst, et, np.arange(A), flat, temp_length=100.0, precision=prec)
    for mode in ("throughput", "latency"):
        os.environ["TI_TEMPLATE"] = mode
        p, d, _ = eng.rollout_dlogp(x, cond, grid, scheme="heun", save_every=0)
        err = np.abs(d - d64)[0]
        print(prec, mode, "dlogp err vs f64: max", err.max(), "at", err.argmax(), "median", np.median(err), " path rel", np.linalg.norm(p - p64) / np.linalg.norm(p64 - x))
        b, div = eng.drift_div(x, 0.0, cond)
        ob, odiv = orc.drift_div(x, 0.0, cond, precision=64)
        e2 = np.abs(div - odiv); print("    div(t=0) err max", e2.max(), "at", e2.argmax(), "div", div[e2.argmax()], odiv[e2.argmax()])
m = 6
xd = np.zeros((1, A, 3), np.float32); diag = []
for k in range(3 * A):
    xd[:] = 0; xd.reshape(-1)[k] = 1
    _, tan = orc.jvp(x[m:m+1], xd, 0.0, cond[m:m+1], precision=64)
    diag.append(tan.reshape(-1)[k])
diag = np.array(diag); print("molecule", m, "sum |J_kk|", np.abs(diag).sum(), "trace", diag.sum(), "max", np.abs(diag).max())
