#!/usr/bin/env python3
"""One-shot probe of the split-fp16 tangent-readout defect: for each library variant, the divergence of 256
molecules (54 directions each: every CU holds two workgroups of each tangent kernel) is evaluated `reps` times with the split
readout build enabled and compared per molecule with the f32 path of the same library.
    python tools/race_probe.py TAG ...        (TAG as in tools/variant_bench.py; run on the GPU box)"""
import importlib, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(reps):
    sys.path.insert(0, ROOT)
    import numpy as np
    ti = importlib.import_module("thermodynamic-interpolation_amd")
    syn, W = ti.synthetic, ti.weights
    F, L, A, B = 128, 2, 18, 256
    os.environ["TI_TEMPLATE"] = "throughput"
    src, dst, et = syn.fully_connected_template(A)
    flat = W.flatten_state_dict(syn.painn_state_dict(0, F, L, 25, seed=F + A), W.painn_param_spec(0, F, L, 25))
    x, cond = syn.molecule_coords(B, A, seed=B), syn.ambient_cond(B, A)
    out = {}
    ref = None
    for prec in ("f32", "f16x2"):
        eng = ti.engine.PainnEngine(0, F, L, A, src, dst, et, np.arange(A), flat, temp_length=100.0, precision=prec)
        bad, worst = [], 0.0
        for _ in range(reps):
            d = eng.drift_div(x, 0.5, cond)[1].astype(np.float64)
            if ref is None:
                ref = d
            e = np.abs(d - ref)
            bad.append(int((e > 5e-5 * (np.abs(ref) + 1.0)).sum())); worst = max(worst, float(e.max()))
        out[prec] = {"bad_per_eval": bad, "worst": worst}
        eng.close()
    print("@@" + json.dumps(out), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "child":
        child(int(sys.argv[2]))
    else:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from variant_bench import lib_of
        for tag in sys.argv[1:]:
            env = dict(os.environ, TI_LIB_PATH=lib_of(tag), TI_JVP_RO_SPLIT="1")
            try:
                p = subprocess.run([sys.executable, os.path.abspath(__file__), "child", "6"], env=env, capture_output=True, text=True, timeout=400)
            except subprocess.TimeoutExpired:
                print(tag, "TIMEOUT", flush=True); break
            line = [l for l in p.stdout.splitlines() if l.startswith("@@")]
            print(tag, line[0][2:] if line else f"FAILED rc={p.returncode} {p.stderr[-500:]}", flush=True)
