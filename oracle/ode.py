"""numpy restatement of the torchdiffeq 0.2.5 solvers the reference calls -- TEST INFRASTRUCTURE ONLY (see oracle/ti_oracle.c).

torchdiffeq is third-party and absent from the reference checkout (ti_env.yml:14 pins 0.2.5), so this file restates its
published algorithm: parity UNPINNED against the library itself; it pins the GPU host logic (csrc/ti_api.hip rollout_rk).
  dopri5   : rk_common.py RKAdaptiveStepsizeODESolver (_runge_kutta_step, _compute_error_ratio, _optimal_step_size,
             _interp_fit/_interp_evaluate), dopri5.py tableau, misc.py _select_initial_step / _rms_norm / _mixed_norm,
             _PerturbFunc (stages with alpha == 1 are evaluated one fp32 ulp before t1), _ReverseFunc for decreasing grids.
  midpoint, rk4 (3/8 rule): fixed_grid.py on the output grid (FixedGridODESolver with step_size=None).
State is a list of float32 arrays (one entry, or (x, dlogp)); times are float64 and enter state arithmetic as float32.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32
ALPHA = [1 / 5, 3 / 10, 4 / 5, 8 / 9, 1.0, 1.0]
BETA = [[1 / 5], [3 / 40, 9 / 40], [44 / 45, -56 / 15, 32 / 9], [19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729],
        [9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656], [35 / 384, 0.0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84]]
C_ERROR = [35 / 384 - 1951 / 21600, 0.0, 500 / 1113 - 22642 / 50085, 125 / 192 - 451 / 720, -2187 / 6784 - -12231 / 42400,
           11 / 84 - 649 / 6300, -1.0 / 60.0]
C_MID = [6025192743 / 30085553152 / 2, 0.0, 51252292925 / 65400821598 / 2, -2691868925 / 45128329728 / 2,
         187940372067 / 1594534317056 / 2, -1776094331 / 19743644256 / 2, 11237099 / 235043384 / 2]


def _norm(parts):
    """max over state entries of the root-mean-square (a single tensor: plain rms)."""
    return max(float(np.sqrt(np.mean(np.square(p.astype(np.float64))))) for p in parts)


def _comb(y, ks, coefs, scale):
    out = []
    for s, y_s in enumerate(y):
        acc = np.zeros_like(y_s)
        for k, c in zip(ks, coefs):
            acc = acc + F32(F32(c) * F32(scale)) * k[s]
        out.append((y_s + acc).astype(F32))
    return out


def odeint(func, y0, t, method="dopri5", rtol=1e-4, atol=1e-4):
    """func(t: float, y: list[np.ndarray]) -> list[np.ndarray].  Returns (solution: list over entries of [len(t), ...], n_fevals)."""
    y = [np.asarray(a, F32).copy() for a in y0]
    t = np.asarray(t, np.float64)
    sign = -1.0 if len(t) > 1 and t[1] < t[0] else 1.0
    nfe = [0]

    def f(ti, yy):                      # _ReverseFunc: integrate in s = sign * t
        nfe[0] += 1
        return [(F32(sign) * np.asarray(k, F32)).astype(F32) for k in func(float(F32(sign * float(F32(ti)))), yy)]

    s_grid = sign * t
    sol = [[a.copy()] for a in y]
    if method in ("midpoint", "rk4"):
        for i in range(len(t) - 1):
            t0, t1 = F32(s_grid[i]), F32(s_grid[i + 1])
            dt = F32(t1 - t0)
            k1 = f(t0, y)
            if method == "midpoint":
                k2 = f(F32(t0 + F32(0.5) * dt), _comb(y, [k1], [0.5], dt))
                y = _comb(y, [k1, k2], [0.0, 1.0], dt)
            else:
                k2 = f(F32(t0 + dt * F32(1 / 3)), _comb(y, [k1], [1 / 3], dt))
                k3 = f(F32(t0 + dt * F32(2 / 3)), _comb(y, [k1, k2], [-1 / 3, 1.0], dt))
                k4 = f(t1, _comb(y, [k1, k2, k3], [1.0, -1.0, 1.0], dt))
                y = _comb(y, [k1, k2, k3, k4], [0.125, 0.375, 0.375, 0.125], dt)
            for s, a in enumerate(y):
                sol[s].append(a.copy())
        return [np.stack(v) for v in sol], nfe[0]
    if method != "dopri5":
        raise ValueError(method)

    rtol, atol = F32(rtol), F32(atol)
    f0 = f(s_grid[0], y)
    scale = [atol + np.abs(a) * rtol for a in y]
    d0 = _norm([a / sc for a, sc in zip(y, scale)])
    d1 = _norm([k / sc for k, sc in zip(f0, scale)])
    h0 = 1e-6 if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * d0 / d1
    f1 = f(s_grid[0] + h0, _comb(y, [f0], [1.0], h0))
    d2 = _norm([(b - a) / sc for a, b, sc in zip(f0, f1, scale)]) / h0
    h1 = max(1e-6, h0 * 1e-3) if (d1 <= 1e-15 and d2 <= 1e-15) else (0.01 / max(d1, d2)) ** (1.0 / 5.0)
    dt = min(100.0 * h0, h1)
    t0 = t1 = float(s_grid[0])
    coef = None
    k = [f0] + [None] * 6
    for i in range(1, len(t)):
        nxt = float(s_grid[i])
        while nxt > t1:
            ts, te = t1, t1 + dt
            assert te > ts, "step size underflow"
            tsf, dtf, tef = F32(ts), F32(dt), F32(te)
            for sidx in range(6):
                ti = np.nextafter(tef, F32(tef - F32(1.0))) if ALPHA[sidx] == 1.0 else F32(tsf + F32(ALPHA[sidx]) * dtf)
                yi = _comb(y, k[:sidx + 1], BETA[sidx], dtf)
                k[sidx + 1] = f(ti, yi)
            y1 = yi
            err = [(_comb([np.zeros_like(a) for a in y], k, C_ERROR, dtf))[s] for s in range(len(y))]
            ratio = _norm([e / (atol + rtol * np.maximum(np.abs(a), np.abs(b))) for e, a, b in zip(err, y, y1)])
            if ratio <= 1.0:
                ymid = _comb(y, k, C_MID, dtf)
                coef = []
                for s in range(len(y)):
                    a0, a1, g0, g1, ym = y[s], y1[s], k[0][s], k[6][s], ymid[s]
                    coef.append([a0, dtf * g0, dtf * (g1 - F32(4) * g0) - F32(11) * a0 - F32(5) * a1 + F32(16) * ym,
                                 dtf * (F32(5) * g0 - F32(3) * g1) + F32(18) * a0 + F32(14) * a1 - F32(32) * ym,
                                 F32(2) * dtf * (g1 - g0) - F32(8) * (a1 + a0) + F32(16) * ym])
                y, k = y1, [k[6]] + [None] * 6
                t0, t1 = ts, te
            dt = dt * 10.0 if ratio == 0.0 else dt * min(10.0, max(0.9 / ratio ** 0.2, 1.0 if ratio < 1.0 else 0.2))
        x = F32((nxt - t0) / (t1 - t0))
        for s in range(len(y)):
            c = coef[s]
            total = c[0] + x * c[1]
            xp = x
            for cc in c[2:]:
                xp = F32(xp * x)
                total = total + xp * cc
            sol[s].append(total.astype(F32))
    return [np.stack(v) for v in sol], nfe[0]
