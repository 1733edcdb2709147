"""Localise sporadic wrong tangents of the split-precision path: D = 1 runs over replicated molecules with unit seeds (the same
kernels as the divergence), every row checked against the fp64 oracle stage by stage."""
import importlib, os, sys, numpy as np
sys.path.insert(0, '.')
ti = importlib.import_module("thermodynamic-interpolation_amd")
from oracle import oracle
syn, W = ti.synthetic, ti.weights
F, L, A, B0, variant = 128, 2, 18, int(sys.argv[1]) if len(sys.argv) > 1 else 96, 0
prec = sys.argv[2] if len(sys.argv) > 2 else "f16x2"
os.environ["TI_TEMPLATE"] = "throughput"
src, dst, et = syn.fully_connected_template(A)
flat = W.flatten_state_dict(syn.painn_state_dict(variant, F, L, 25, seed=F + A), W.painn_param_spec(variant, F, L, 25))
x0 = syn.molecule_coords(B0, A, seed=B0); c0 = syn.ambient_cond(B0, A)
D = 3 * A
x = np.repeat(x0, D, axis=0); cond = np.repeat(c0, D, axis=0); B = B0 * D
xdot = np.zeros((B, A * 3), np.float32); xdot[np.arange(B), np.arange(B) % D] = 1.0; xdot = xdot.reshape(B, A, 3)
orc = oracle.PainnOracle(variant, F, L, A, src, dst, et, np.arange(A), flat, temp_length=100.0)
eng = ti.engine.PainnEngine(variant, F, L, A, src, dst, et, np.arange(A), flat, temp_length=100.0, precision=prec)
_, otan = orc.jvp(x, xdot, 0.5, cond, precision=64)
scale = np.abs(otan).reshape(B, -1).max(axis=1)
for rep in range(0 if os.environ.get('FAST') else 3):
    _, tan = eng.jvp(x, xdot, 0.5, cond)
    err = np.abs(tan - otan).reshape(B, -1).max(axis=1) / (scale + 1e-3)
    bad = np.nonzero(err > 1e-4)[0]
    print(f"rep {rep}: {bad.size} bad rows of {B}; first: {bad[:8]} (molecule, direction) = {[(int(b // D), int(b % D)) for b in bad[:8]]} err {err[bad[:8]]}", flush=True)
# stage taps for one run: find the first stage at which rows go wrong
stages = [(1, "msg0"), (2, "upd0"), (3, "msg1"), (4, "upd1")]
for stage, tag in ([] if os.environ.get('FAST') else stages):
    eng.debug_tap(stage)
    eng.jvp(x, xdot, 0.5, cond)
    ts, tv, te = eng.debug_read("ts", B), eng.debug_read("tv", B).transpose(0, 1, 3, 2), eng.debug_read("te", B)
    _, _, taps = orc.jvp(x, xdot, 0.5, cond, precision=64, tap_stage=stage)
    for name, g, o in (("ts", ts, taps["s"]), ("tv", tv, taps["v"]), ("te", te, taps["e"])):
        if name == "te" and stage >= 3:
            continue
        e = np.abs(g - o).reshape(B, -1).max(axis=1) / (np.abs(o).reshape(B, -1).max(axis=1) + 1e-3)
        bad = np.nonzero(e > 1e-4)[0]
        msg = ""
        if bad.size:
            r = bad[0]
            d = np.abs(g[r] - o[r]); idx = np.unravel_index(d.argmax(), d.shape)
            msg = f" e.g. row {r} (mol {r // D}, dir {r % D}) worst element {idx} got {g[r][idx]} want {o[r][idx]}; bad elements in row: {(d > 1e-4 * (np.abs(o[r]).max() + 1e-3)).sum()} of {d.size}"
        print(f"{tag} {name}: {bad.size} bad rows{msg}", flush=True)
eng.debug_tap(-1)

# ---- same-run check: full run, then read the final tangent state the readout kernel consumed
print("---- same-run localisation", flush=True)
_, _, taps = orc.jvp(x, xdot, 0.5, cond, precision=64, tap_stage=2 * L)
for rep in range(0 if os.environ.get('FAST') else 4):
    _, tan = eng.jvp(x, xdot, 0.5, cond)
    ts, tv = eng.debug_read("ts", B), eng.debug_read("tv", B).transpose(0, 1, 3, 2)
    s, v = eng.debug_read("s", B), eng.debug_read("v", B).transpose(0, 1, 3, 2)
    err = np.abs(tan - otan).reshape(B, -1).max(axis=1) / (scale + 1e-3)
    bad = np.nonzero(err > 1e-4)[0]
    e_ts = np.abs(ts - taps["s"]).reshape(B, -1).max(axis=1) / (np.abs(taps["s"]).reshape(B, -1).max(axis=1) + 1e-3)
    e_tv = np.abs(tv - taps["v"]).reshape(B, -1).max(axis=1) / (np.abs(taps["v"]).reshape(B, -1).max(axis=1) + 1e-3)
    print(f"rep {rep}: bad output rows {bad[:10]}; bad ts rows {np.nonzero(e_ts > 1e-4)[0][:10]}; bad tv rows {np.nonzero(e_tv > 1e-4)[0][:10]}", flush=True)
    for r in bad[:4]:
        d = np.abs(tan[r] - otan[r]); atoms = np.nonzero(d.max(axis=1) > 1e-4 * (scale[r] + 1e-3))[0]
        nodes = r * A + atoms
        print(f"   row {r}: bad atoms {atoms} -> virtual nodes {nodes} (tile {nodes // 16}, lane-row {nodes % 16}); |ts| max {np.abs(ts[r]).max():.3g} |tv| max {np.abs(tv[r]).max():.3g} |s| max {np.abs(s[r]).max():.3g}", flush=True)

# ---- which factor of  tout = (Vr.tv) gate + (Vr.v) tgate  is wrong in a bad tile?
print("---- factor analysis", flush=True)
Vr = np.asarray(flat, np.float64)[-F:]            # canonical layout: Vr is the last F weights
for rep in range(6):
    _, tan = eng.jvp(x, xdot, 0.5, cond)
    tv = eng.debug_read("tv", B).transpose(0, 1, 3, 2); v = eng.debug_read("v", B).transpose(0, 1, 3, 2)
    err = np.abs(tan - otan).reshape(B, -1).max(axis=1) / (scale + 1e-3)
    bad = np.nonzero(err > 1e-4)[0]
    for r in bad[:3]:
        acc = np.einsum("afc,f->ac", v[r].astype(np.float64), Vr); tacc = np.einsum("afc,f->ac", tv[r].astype(np.float64), Vr)
        for a in range(A):
            e = (tan[r, a] - otan[r, a]).astype(np.float64)
            if np.abs(e).max() < 1e-4 * (scale[r] + 1e-3):
                continue
            M = np.stack([tacc[a], acc[a]], axis=1)
            sol, res, *_ = np.linalg.lstsq(M, e, rcond=None)
            print(f"   rep {rep} row {r} atom {a}: err {e}  -> dgate {sol[0]:.3e} dtgate {sol[1]:.3e} residual {np.abs(M @ sol - e).max():.1e}", flush=True)
            break


# ---- stale-input hypotheses for the bad tiles: was (Vr . v) formed from the v of BEFORE the last update kernel, or (Vr . tv)
# from the tv of before the last tangent update?
print("---- stale-input hypotheses", flush=True)
eng.debug_tap(2 * L - 2 if L > 1 else 0)          # state after the previous layer's update
eng.jvp(x, xdot, 0.5, cond)
v_prev = eng.debug_read("v", B).transpose(0, 1, 3, 2).astype(np.float64); tv_prev = eng.debug_read("tv", B).transpose(0, 1, 3, 2).astype(np.float64)
eng.debug_tap(-1)
ob, _ = orc.jvp(x, xdot, 0.5, cond, precision=64)
for rep in range(6):
    b, tan = eng.jvp(x, xdot, 0.5, cond)
    tv = eng.debug_read("tv", B).transpose(0, 1, 3, 2).astype(np.float64); v = eng.debug_read("v", B).transpose(0, 1, 3, 2).astype(np.float64)
    err = np.abs(tan - otan).reshape(B, -1).max(axis=1) / (scale + 1e-3)
    bad = np.nonzero(err > 1e-4)[0]
    print(f"rep {rep}: bad rows {bad[:6]}; primal drift of the same run: max rel err {np.abs(b - ob).max() / np.abs(ob).max():.1e}", flush=True)
    for r in bad[:3]:
        acc = np.einsum("afc,f->ac", v[r], Vr); tacc = np.einsum("afc,f->ac", tv[r], Vr)
        acc_p = np.einsum("afc,f->ac", v_prev[r], Vr); tacc_p = np.einsum("afc,f->ac", tv_prev[r], Vr)
        gate = np.where(np.abs(acc) > 1e-12, ob[r].astype(np.float64) / acc, 0.0).mean(axis=1)          # b = gate * acc
        for a in range(A):
            e = (tan[r, a] - otan[r, a]).astype(np.float64)
            c = int(np.abs(e).argmax())
            if np.abs(e[c]) < 1e-4 * (scale[r] + 1e-3):
                continue
            tgate = (otan[r, a, c] - tacc[a, c] * gate[a]) / acc[a, c]
            h1 = tacc[a, c] * gate[a] + acc_p[a, c] * tgate          # stale v
            h2 = tacc_p[a, c] * gate[a] + acc[a, c] * tgate          # stale tv
            print(f"   row {r} atom {a} comp {c}: got {tan[r, a, c]:.6f} want {otan[r, a, c]:.6f} | if v stale {h1:.6f} | if tv stale {h2:.6f}", flush=True)
            break
