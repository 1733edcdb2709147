import importlib, sys, numpy as np
sys.path.insert(0, '.')
ti = importlib.import_module("thermodynamic-interpolation_amd")
from oracle import oracle
syn, W = ti.synthetic, ti.weights
F, L, A, B = 128, 5, 18, int(sys.argv[1]) if len(sys.argv) > 1 else 4096
src, dst, et = syn.fully_connected_template(A)
flat = W.flatten_state_dict(syn.painn_state_dict(0, F, L, 25, 0), W.painn_param_spec(0, F, L, 25))
x, cond = syn.molecule_coords(B, A, 0), syn.ambient_cond(B, A)
orc = oracle.PainnOracle(0, F, L, A, src, dst, et, np.arange(A), flat, temp_length=100.0)
ref = orc.drift(x, 0.5, cond)
for prec in ("f32", "f16x2"):
    eng = ti.engine.PainnEngine(0, F, L, A, src, dst, et, np.arange(A), flat, temp_length=100.0, precision=prec)
    b = eng.drift(x, 0.5, cond)
    err = np.linalg.norm((b - ref).reshape(B, -1), axis=1) / np.linalg.norm(ref.reshape(B, -1), axis=1)
    bad = np.nonzero(err > 1e-5)[0]
    print(prec, "total rel", np.linalg.norm(b - ref) / np.linalg.norm(ref), "n_bad", bad.size, "first bad", bad[:24], "max", err.max())
    if bad.size:
        print("  bad mol % 3:", np.bincount(bad % 3, minlength=3), " group idx % 4 (wave):", np.bincount((bad // 3) % 4, minlength=4), " WG idx range", (bad // 12).min(), (bad // 12).max())
        # stage-wise: where does it first deviate?
        for stage in range(0, 2 * L + 1):
            eng.debug_tap(stage); eng.drift(x, 0.5, cond)
            s = eng.debug_read("s", B)
            _, taps = orc.drift(x[bad[:4]], 0.5, cond[bad[:4]], tap_stage=stage)
            es = np.linalg.norm(s[bad[:4]] - taps["s"]) / np.linalg.norm(taps["s"])
            print("   stage", stage, "s err on first bad mols", es)
        eng.debug_tap(-1)
