#!/usr/bin/env python3
"""Times experiment builds of libti_hip.so against each other on one GPU box without touching the product library.

    python tools/variant_bench.py build  TAG[:FLAGS] ...     # here (cross-compile): thermodynamic-interpolation_amd/build/variants/libti_hip_TAG.so
    python tools/variant_bench.py run [--batch B] [--steps K] [--rounds R] [--precision f16x2] TAG ...     # on the GPU box

FLAGS are extra hipcc flags, '+'-separated (e.g. pk:-Xclang+-target-feature+-Xclang++packed-fp32-ops, x:-DTI_SOMETHING=1), appended to the product's own (build.py);
ONLY=file.hip restricts recompilation to that source (the product objects of the others are linked).
TAG `base` means the product library; `TAG@LAYOUT` (run only) pins the edge-row layout of that arm (throughput | latency | pair;
default: TI_VB_TEMPLATE or throughput), e.g. `base@throughput base@pair`.  `run` starts one child process per (round, tag), interleaved, so that clock and box
drift hit every variant alike; each child reports the HIP-event averages of the edge and update kernels, the wall time per
step, and the drift of 64 molecules, which is compared with the first tag's (max abs difference) and with the f32 path's.
No torch import anywhere (a fresh box pays 1-2 minutes for it)."""
import importlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "thermodynamic-interpolation_amd")
VDIR = os.path.join(PKG, "build", "variants")
SOURCES = ["ti_api.hip", "painn_kernels.hip", "painn_edge_nb1.hip", "painn_edge_nb2.hip", "painn_edge_nb4.hip", "painn_edge_nb8.hip",
           "painn_pair_nb1.hip", "painn_pair_nb2.hip", "painn_pair_nb4.hip", "painn_jvp_kernels.hip", "adw_kernels.hip", "ode_kernels.hip"]


def _load_build():
    import importlib.util
    spec = importlib.util.spec_from_file_location("ti_build", os.path.join(PKG, "build.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


BUILD = _load_build()


def lib_of(tag):
    tag = tag.partition("@")[0]
    return os.path.join(PKG, "libti_hip.so") if tag == "base" else os.path.join(VDIR, f"libti_hip_{tag}.so")


def build(specs):
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(VDIR, exist_ok=True)
    for spec in specs:
        tag, _, flags = spec.partition(":")
        flags = [f for f in flags.split("+") if f]
        only = [f[len("ONLY="):] for f in flags if f.startswith("ONLY=")]
        flags = [f for f in flags if not f.startswith("ONLY=")]
        objdir = os.path.join(VDIR, tag)
        os.makedirs(objdir, exist_ok=True)

        def one(src):
            if only and src not in only:
                return os.path.join(PKG, "build", src.replace(".hip", ".o"))
            obj = os.path.join(objdir, src.replace(".hip", ".o"))
            csrc = os.environ.get("TI_VARIANT_SRC") or os.path.join(PKG, "csrc")       # e.g. an extracted `git archive` of another commit
            subprocess.check_call(["hipcc", *BUILD.FLAGS, *BUILD.EXTRA_FLAGS.get(src, []), *flags, "-c", os.path.join(csrc, src), "-o", obj])      # the product's flags + the variant's
            return obj
        with ThreadPoolExecutor(max_workers=8) as ex:
            objs = list(ex.map(one, SOURCES))
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_of(tag), *objs])
        print("built", lib_of(tag), flush=True)


def child(batch, steps, precision, small):
    sys.path.insert(0, ROOT)
    import numpy as np
    ti = importlib.import_module("thermodynamic-interpolation_amd")
    syn, W, E = ti.synthetic, ti.weights, ti.engine
    F, L, A = 128, 5, 18
    tpl = syn.fully_connected_template(A)
    flat = W.flatten_state_dict(syn.painn_state_dict(W.AMBIENT, F, L, 25, 0), W.painn_param_spec(W.AMBIENT, F, L, 25))
    os.environ["TI_TEMPLATE"] = os.environ.get("TI_VB_TEMPLATE", "throughput")
    eng = E.PainnEngine(W.AMBIENT, F, L, A, *tpl, np.arange(A), flat, temp_length=100.0, precision=precision)
    xs, cs = syn.molecule_coords(64, A, seed=7), syn.ambient_cond(64, A)
    d = eng.drift(xs, 0.3, cs)
    rec = {"drift": d.tolist()}
    if not small:
        x0, cond = syn.molecule_coords(batch, A, seed=0), syn.ambient_cond(batch, A)
        grid = E._time_grid_numpy(0.0, 1.0, 1001)
        eng.reserve(batch)
        def roll(g):                         # experiment variants may produce non-finite states (TI_E_NAN): the timing still counts
            try:
                return eng.rollout(x0, cond, g, scheme="em", eps=0.01, seed=1, save_every=0)[0]
            except ti._lib.TiError as e:
                if e.code != ti._lib.TI_E_NAN:
                    raise
                return np.full((1, 1), np.nan, np.float32)
        roll(grid[:2])
        eng.profile(True)
        t0 = time.perf_counter()
        out = roll(grid[1:steps + 2])
        dt = time.perf_counter() - t0
        eng.profile(False)
        prof = {k: eng.profile_read(k) for k in ("painn_edge", "painn_update", "painn_embed", "painn_readout")}
        rec.update(edge_ms=prof["painn_edge"][1] / max(prof["painn_edge"][0], 1), upd_ms=prof["painn_update"][1] / max(prof["painn_update"][0], 1),
                   embed_ms=prof["painn_embed"][1] / max(prof["painn_embed"][0], 1), readout_ms=prof["painn_readout"][1] / max(prof["painn_readout"][0], 1),
                   kernels_ms_per_step=sum(v[1] for v in prof.values()) / steps, wall_ms_per_step=1e3 * dt / steps, finite=bool(np.isfinite(out).all()))
    print("@@" + json.dumps(rec), flush=True)


def run(argv):
    import argparse
    import numpy as np
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32768)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--precision", default="f16x2")
    ap.add_argument("tags", nargs="+")
    a = ap.parse_args(argv)

    def spawn(tag, precision, small):
        env = dict(os.environ, TI_LIB_PATH=lib_of(tag))
        if "@" in tag:
            env["TI_VB_TEMPLATE"] = tag.partition("@")[2]
        cmd = [sys.executable, os.path.abspath(__file__), "child", str(a.batch), str(a.steps), precision, "1" if small else "0"]
        try:
            p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
        except subprocess.TimeoutExpired:
            print(f"{tag}: TIMEOUT", flush=True)
            return None
        line = [l for l in p.stdout.splitlines() if l.startswith("@@")]
        if p.returncode != 0 or not line:
            print(f"{tag}: FAILED rc={p.returncode}\n{p.stdout[-400:]}\n{p.stderr[-800:]}", flush=True)
            return None
        return json.loads(line[0][2:])
    ref32 = spawn(a.tags[0], "f32", True)
    ref32 = np.asarray(ref32["drift"]) if ref32 else None
    first = None
    res = {t: [] for t in a.tags}
    dead = set()
    for rnd in range(a.rounds):
        for tag in a.tags:
            if tag in dead:
                continue
            r = spawn(tag, a.precision, False)
            if r is None:
                dead.add(tag)             # a variant that failed or hung is not started again
                continue
            d = np.asarray(r.pop("drift"))
            if first is None:
                first = d
            r["max_abs_vs_first"] = float(np.abs(d - first).max())
            if ref32 is not None:
                r["rel_l2_vs_f32"] = float(np.linalg.norm(d - ref32) / np.linalg.norm(ref32))
            res[tag].append(r)
            print(f"round {rnd} {tag:>12}: edge {r['edge_ms']:7.3f} ms  upd {r['upd_ms']:6.3f} ms  kernels/step {r['kernels_ms_per_step']:8.2f} ms  "
                  f"wall/step {r['wall_ms_per_step']:8.2f} ms  |d-first| {r['max_abs_vs_first']:.2e}  rel-L2 vs f32 {r.get('rel_l2_vs_f32', float('nan')):.2e}  finite {r['finite']}", flush=True)
    print("---- medians (batch %d, %s)" % (a.batch, a.precision))
    for tag, rs in res.items():
        if rs:
            print(f"{tag:>12}: edge {np.median([r['edge_ms'] for r in rs]):7.3f}  upd {np.median([r['upd_ms'] for r in rs]):6.3f}  kernels/step {np.median([r['kernels_ms_per_step'] for r in rs]):8.2f}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) < 2:
        sys.exit(__doc__)
    if sys.argv[1] == "build":
        build(sys.argv[2:])
    elif sys.argv[1] == "child":
        child(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5] == "1")
    elif sys.argv[1] == "run":
        run(sys.argv[2:])
    else:
        sys.exit(__doc__)
