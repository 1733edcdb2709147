import importlib, os, sys, numpy as np
sys.path.insert(0, '.')
ti = importlib.import_module("thermodynamic-interpolation_amd")
syn, W = ti.synthetic, ti.weights
F, L, A, B = 128, 5, 18, int(sys.argv[1])
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
src, dst, et = syn.fully_connected_template(A)
flat = W.flatten_state_dict(syn.painn_state_dict(0, F, L, 25, 0), W.painn_param_spec(0, F, L, 25))
x, cond = syn.molecule_coords(B, A, 0), syn.ambient_cond(B, A)
cache = f"/tmp/ti_stress_ref_{B}.npy"
if os.path.exists(cache):
    ref = np.load(cache)
else:
    from oracle import oracle
    ref = oracle.PainnOracle(0, F, L, A, src, dst, et, np.arange(A), flat, temp_length=100.0).drift(x, 0.5, cond)
    np.save(cache, ref)
for prec in ("f32", "f16x2"):
    eng = ti.engine.PainnEngine(0, F, L, A, src, dst, et, np.arange(A), flat, temp_length=100.0, precision=prec)
    tot = []
    for r in range(reps):
        b = eng.drift(x, 0.5, cond)
        err = np.linalg.norm((b - ref).reshape(B, -1), axis=1) / np.linalg.norm(ref.reshape(B, -1), axis=1)
        bad = np.nonzero(~(err < 2e-5))[0]
        tot.append(bad.size)
    print(sys.argv[3] if len(sys.argv) > 3 else "", prec, "bad molecules per eval:", tot, "last bad ids", bad[:8], flush=True)
