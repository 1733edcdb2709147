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
os.environ["TI_TEMPLATE"] = "throughput"
eng = ti.engine.PainnEngine(variant, F, L, A, src, dst, et, np.arange(A), flat, temp_length=100.0, precision="f16x2")
# the four Heun stage states
dt = 0.5
b1, _ = orc.drift_div(x, 0.0, cond, precision=64); xa = (x + dt * b1).astype(np.float32)
b2, _ = orc.drift_div(xa, 0.5, cond, precision=64); x1 = (x + 0.5 * dt * (b1 + b2)).astype(np.float32)
b3, _ = orc.drift_div(x1, 0.5, cond, precision=64); xb = (x1 + dt * b3).astype(np.float32)
for name, xs, t in (("x0,0", x, 0.0), ("xa,.5", xa, 0.5), ("x1,.5", x1, 0.5), ("xb,1", xb, 1.0)):
    _, div = eng.drift_div(xs, t, cond)
    _, odiv = orc.drift_div(xs, t, cond, precision=64)
    e = np.abs(div - odiv); m = int(e.argmax())
    print(name, "max err", e.max(), "mol", m, "div", div[m], odiv[m], "2nd", np.sort(e)[-2])
    if e.max() > 1e-4:
        xd = np.zeros((B, A, 3), np.float32); worst = []
        for k in range(3 * A):
            xd[:] = 0; xd.reshape(B, -1)[:, k] = 1
            _, tan = eng.jvp(xs, xd, t, cond)
            _, otan = orc.jvp(xs, xd, t, cond, precision=64)
            worst.append((abs(tan[m].reshape(-1)[k] - otan[m].reshape(-1)[k]), k, tan[m].reshape(-1)[k], otan[m].reshape(-1)[k],
                          np.abs(tan[m] - otan[m]).max() / np.abs(otan[m]).max()))
        worst.sort(reverse=True); print("   worst directions (D=1 jvp path):", worst[:3])
        # same through the D=54 path but restricted to molecule m alone
        _, d1 = eng.drift_div(xs[m:m+1], t, cond[m:m+1]); print("   molecule alone:", d1, odiv[m])
        for lo in (0, m - m % 2, m):
            _, d2 = eng.drift_div(xs[lo:lo + 2], t, cond[lo:lo + 2]); print("   pair from", lo, d2, odiv[lo:lo + 2])
