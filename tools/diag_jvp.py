import sys, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
from conftest import load_golden, rel_l2
from test_gpu_divergence import make_pair
name = sys.argv[1] if len(sys.argv) > 1 else "div_ambient_small"
prec = sys.argv[2] if len(sys.argv) > 2 else "f32"
g = load_golden(name)
eng, orc = make_pair(g, prec)
B, L, t = int(g["B"]), int(g["L"]), float(g["t"])
xdot = np.random.RandomState(5).standard_normal(g["x"].shape).astype(np.float32)
b, tan = eng.jvp(g["x"], xdot, t, g["cond"])
rb, rtan = orc.jvp(g["x"], xdot, t, g["cond"])
print("drift rel", rel_l2(b, rb), "tan rel", rel_l2(tan, rtan), "|tan|", np.linalg.norm(tan), "|rtan|", np.linalg.norm(rtan))
for l in range(L):
    for stage, tag in ((1 + 2 * l, f"msg{l}"), (2 + 2 * l, f"upd{l}")):
        eng.debug_tap(stage)
        eng.jvp(g["x"], xdot, t, g["cond"])
        _, _, taps = orc.jvp(g["x"], xdot, t, g["cond"], tap_stage=stage)
        ts, tv, te = eng.debug_read("ts", B), eng.debug_read("tv", B).transpose(0, 1, 3, 2), eng.debug_read("te", B)
        print(tag, "ts", rel_l2(ts, taps["s"]), np.linalg.norm(ts), np.linalg.norm(taps["s"]),
              "| tv", rel_l2(tv, taps["v"]), np.linalg.norm(tv), np.linalg.norm(taps["v"]),
              "| te", rel_l2(te, taps["e"]), np.linalg.norm(te), np.linalg.norm(taps["e"]))
eng.debug_tap(-1)
