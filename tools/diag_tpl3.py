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
b1, _ = orc.drift_div(x, 0.0, cond, precision=64); xa = (x + 0.5 * b1).astype(np.float32)
_, odiv = orc.drift_div(xa, 0.5, cond, precision=64)
for prec in ("f16x2", "f32"):
    eng = ti.engine.PainnEngine(variant, F, L, A, src, dst, et, np.arange(A), flat, temp_length=100.0, precision=prec)
    for lo, hi in ((0, 50), (0, 50), (0, 10), (0, 6), (4, 50), (2, 8), (4, 6), (0, 49), (0, 26)):
        _, div = eng.drift_div(xa[lo:hi], 0.5, cond[lo:hi])
        e = np.abs(div - odiv[lo:hi])
        print(prec, (lo, hi), "max err", f"{e.max():.2e}", "at mol", lo + int(e.argmax()), "n>1e-4:", int((e > 1e-4).sum()), flush=True)
    os.environ["TI_JVP_WS_GB"] = "0.2"
    _, div = eng.drift_div(xa, 0.5, cond); e = np.abs(div - odiv); print(prec, "chunked 0.2GB", f"{e.max():.2e}", int(e.argmax()))
    del os.environ["TI_JVP_WS_GB"]
    eng.close()
