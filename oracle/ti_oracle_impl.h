/*
 * ti_oracle_impl.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Included twice by ti_oracle.c with
 * REAL = float (suffix _f32: restates the reference's fp32 arithmetic) and REAL = double (suffix _f64: the
 * "exact" value used to size rounding noise, and the adw reference precision, adw/train.py:29).
 *
 * Every function cites the reference lines it restates (paths relative to /root/reference).
 */

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUFFIX)

/* y[r][o] = b[o] + sum_k x[r][k] * Wt[k][o]   (torch.nn.Linear; Wt is the transposed [in][out] copy made at create) */
TIO_CLONES static void FN(linear)(const REAL* restrict x, int rows, int ldx, int f_in, const REAL* restrict Wt,
                                  const REAL* restrict b, int f_out, REAL* restrict y, int ldy)
{
    for (int r = 0; r < rows; ++r) {
        REAL* restrict yr = y + (size_t)r * ldy;
        if (b) for (int o = 0; o < f_out; ++o) yr[o] = b[o];
        else   for (int o = 0; o < f_out; ++o) yr[o] = 0;
        const REAL* xr = x + (size_t)r * ldx;
        for (int k = 0; k < f_in; ++k) {
            const REAL a = xr[k];
            const REAL* restrict w = Wt + (size_t)k * f_out;
            for (int o = 0; o < f_out; ++o) yr[o] += a * w[o];
        }
    }
}

/* torch.nn.LayerNorm(F, eps=1e-5, affine) followed by torch.nn.SiLU, in place (embedding.py:29-30,32-33) */
static void FN(ln_silu)(REAL* x, int rows, int F, const REAL* g, const REAL* be)
{
    for (int r = 0; r < rows; ++r) {
        REAL* xr = x + (size_t)r * F;
        REAL mean = 0, var = 0;
        for (int i = 0; i < F; ++i) mean += xr[i];
        mean /= (REAL)F;
        for (int i = 0; i < F; ++i) { REAL d = xr[i] - mean; var += d * d; }
        var /= (REAL)F;
        const REAL rstd = (REAL)1 / RSQRT(var + (REAL)1e-5);
        for (int i = 0; i < F; ++i) {
            REAL v = (xr[i] - mean) * rstd * g[i] + be[i];
            xr[i] = v / ((REAL)1 + REXP(-v));
        }
    }
}

typedef struct { const REAL *W0t, *b0, *g0, *be0, *W1t, *b1, *g1, *be1, *W2t, *b2; int f_in, f_h, f_out; } FN(mlp_t);

/* embedding.MLP.forward, skip=False (embedding.py:37-49); tmp: rows*f_h*2 scratch */
static void FN(mlp)(const FN(mlp_t)* m, const REAL* x, int rows, REAL* y, REAL* tmp)
{
    REAL* h1 = tmp; REAL* h2 = tmp + (size_t)rows * m->f_h;
    FN(linear)(x, rows, m->f_in, m->f_in, m->W0t, m->b0, m->f_h, h1, m->f_h);
    FN(ln_silu)(h1, rows, m->f_h, m->g0, m->be0);
    FN(linear)(h1, rows, m->f_h, m->f_h, m->W1t, m->b1, m->f_h, h2, m->f_h);
    FN(ln_silu)(h2, rows, m->f_h, m->g1, m->be1);
    FN(linear)(h2, rows, m->f_h, m->f_h, m->W2t, m->b2, m->f_out, y, m->f_out);
}

/* PositionalEncoder.forward (embedding.py:127-160): for rank k=1..F/2: [cos(x/max_length*k*pi), sin(...)] interleaved.
 * The reference evaluates ((x / max_length) * k) * pi left to right in the tensor dtype. */
static void FN(posenc)(REAL x, REAL max_length, int F, REAL* out)
{
    const REAL xs = x / max_length;
    for (int k = 1; k <= F / 2; ++k) {
        const REAL a = (xs * (REAL)k) * (REAL)M_PI;
        out[2 * (k - 1)] = RCOS(a);
        out[2 * (k - 1) + 1] = RSIN(a);
    }
}

typedef struct {
    ti_painn_desc d;
    int nE, ncond;
    REAL *edge_emb, *atom_emb;
    FN(mlp_t) embed, *phi, *w, *upd, readout;
    REAL **Ut, **Vt;          /* [L] transposed F x F */
    REAL *Vr;                 /* [F] */
    REAL *store; size_t n_store;
} FN(painn_t);

static size_t FN(take_mlp)(FN(mlp_t)* m, const float* src, REAL** dstp, int f_in, int f_h, int f_out)
{
    /* copies one canonical MLP block, transposing the Linear weights; returns floats consumed */
    const float* p = src; REAL* d = *dstp;
    m->f_in = f_in; m->f_h = f_h; m->f_out = f_out;
#define TAKE_T(field, rows, cols) do { m->field = d; for (int r = 0; r < (rows); ++r) for (int c = 0; c < (cols); ++c) d[(size_t)c * (rows) + r] = (REAL)p[(size_t)r * (cols) + c]; d += (size_t)(rows) * (cols); p += (size_t)(rows) * (cols); } while (0)
#define TAKE_V(field, n) do { m->field = d; for (int i = 0; i < (n); ++i) d[i] = (REAL)p[i]; d += (n); p += (n); } while (0)
    TAKE_T(W0t, f_h, f_in); TAKE_V(b0, f_h); TAKE_V(g0, f_h); TAKE_V(be0, f_h);
    TAKE_T(W1t, f_h, f_h);  TAKE_V(b1, f_h); TAKE_V(g1, f_h); TAKE_V(be1, f_h);
    TAKE_T(W2t, f_out, f_h); TAKE_V(b2, f_out);
    *dstp = d;
    return (size_t)(p - src);
}

static FN(painn_t)* FN(painn_parse)(const ti_painn_desc* d, const float* wts, size_t n)
{
    const int F = d->n_features, L = d->n_layers;
    FN(painn_t)* m = (FN(painn_t)*)calloc(1, sizeof(*m));
    m->d = *d;
    m->nE = d->variant == TI_VARIANT_AMBIENT ? 4 : d->variant == TI_VARIANT_LATENT_MULTI ? 3 : 2;
    m->ncond = d->variant == TI_VARIANT_AMBIENT ? 2 : d->variant == TI_VARIANT_LATENT_MULTI ? 1 : 0;
    m->store = (REAL*)malloc(sizeof(REAL) * n); m->n_store = n;
    m->phi = calloc(L, sizeof(FN(mlp_t))); m->w = calloc(L, sizeof(FN(mlp_t))); m->upd = calloc(L, sizeof(FN(mlp_t)));
    m->Ut = calloc(L, sizeof(REAL*)); m->Vt = calloc(L, sizeof(REAL*));
    const float* p = wts; REAL* q = m->store;
    m->edge_emb = q; for (int i = 0; i < 4 * F; ++i) q[i] = (REAL)p[i]; q += 4 * F; p += 4 * F;
    m->atom_emb = q; for (int i = 0; i < d->n_types * F; ++i) q[i] = (REAL)p[i]; q += d->n_types * F; p += d->n_types * F;
    p += FN(take_mlp)(&m->embed, p, &q, m->nE * F, F, F);
    for (int l = 0; l < L; ++l) {
        p += FN(take_mlp)(&m->phi[l], p, &q, 2 * F, F, 5 * F);
        p += FN(take_mlp)(&m->w[l], p, &q, F, F, 5 * F);
        for (int which = 0; which < 2; ++which) {
            REAL* t = q; for (int r = 0; r < F; ++r) for (int c = 0; c < F; ++c) t[(size_t)c * F + r] = (REAL)p[(size_t)r * F + c];
            if (which == 0) m->Ut[l] = t; else m->Vt[l] = t;
            q += (size_t)F * F; p += (size_t)F * F;
        }
        p += FN(take_mlp)(&m->upd[l], p, &q, 2 * F, F, 3 * F);
    }
    p += FN(take_mlp)(&m->readout, p, &q, F, F, 2);
    m->Vr = q; for (int i = 0; i < F; ++i) q[i] = (REAL)p[i]; q += F; p += F;
    if ((size_t)(p - wts) != n) { free(m->store); free(m); return NULL; }
    return m;
}

static void FN(painn_free)(FN(painn_t)* m)
{
    if (!m) return;
    free(m->store); free(m->phi); free(m->w); free(m->upd); free(m->Ut); free(m->Vt); free(m);
}

/* One molecule.  x [A][3], cond [A][ncond]; out [A][3].
 * taps (may be NULL): after `tap_stage` (0 embed, 1+2l message l, 2+2l update l) copy s [A][F], v [A][F][3], e [E][F]. */
static void FN(painn_molecule)(const FN(painn_t)* m, const int* src, const int* dst, const int* etype, const int* atom_ids,
                               const float* x, float t, const float* cond, float* out,
                               int tap_stage, float* tap_s, float* tap_v, float* tap_e, REAL* ws)
{
    const int F = m->d.n_features, L = m->d.n_layers, A = m->d.n_atoms, E = m->d.n_edges, nE = m->nE;
    REAL* s = ws;                 ws += (size_t)A * F;
    REAL* v = ws;                 ws += (size_t)A * F * 3;
    REAL* e = ws;                 ws += (size_t)E * F;
    REAL* dist = ws;              ws += E;
    REAL* dir = ws;               ws += (size_t)E * 3;
    REAL* enc = ws;               ws += (size_t)E * F;
    REAL* big_in = ws;            ws += (size_t)(E > A ? E : A) * (nE > 2 ? nE : 2) * F;
    REAL* phi_o = ws;             ws += (size_t)E * 5 * F;
    REAL* w_o = ws;               ws += (size_t)E * 5 * F;
    REAL* tmp = ws;               ws += (size_t)(E > A ? E : A) * 2 * F;
    REAL* ds = ws;                ws += (size_t)A * F;
    REAL* dv = ws;                ws += (size_t)A * F * 3;
    REAL* vv = ws;                ws += (size_t)A * F * 3;
    REAL* uv = ws;                ws += (size_t)A * F * 3;
    REAL* upd_o = ws;             ws += (size_t)A * 3 * F;

    /* K1 AddSpatialFeatures (graph.py:25-33): r = x[src]-x[dst], d = |r|, edge_dir = r/(1+d) */
    for (int k = 0; k < E; ++k) {
        REAL r[3]; REAL n2 = 0;
        for (int c = 0; c < 3; ++c) { r[c] = (REAL)x[src[k] * 3 + c] - (REAL)x[dst[k] * 3 + c]; n2 += r[c] * r[c]; }
        dist[k] = RSQRT(n2);
        for (int c = 0; c < 3; ++c) dir[k * 3 + c] = r[c] / ((REAL)1 + dist[k]);
    }
    /* K2 AddEquivariantFeatures (graph.py:36-48) */
    memset(v, 0, sizeof(REAL) * (size_t)A * F * 3);
    /* K3 embeddings (cpainn.py:70-84): e = edge_emb[type]; s_in = [atom | T0 | T1 | t] (latent: [atom | T | t] / [atom | t]) */
    for (int k = 0; k < E; ++k) memcpy(e + (size_t)k * F, m->edge_emb + (size_t)etype[k] * F, sizeof(REAL) * F);
    for (int a = 0; a < A; ++a) {
        REAL* row = big_in + (size_t)a * nE * F;
        memcpy(row, m->atom_emb + (size_t)atom_ids[a] * F, sizeof(REAL) * F);
        for (int c = 0; c < m->ncond; ++c) {
            /* TemperatureEncoder.forward (embedding.py:200-212) */
            REAL u = (REAL)cond[a * m->ncond + c] - (REAL)m->d.temp_mean * (REAL)1;
            u = u / (REAL)m->d.temp_range;
            FN(posenc)(u, (REAL)m->d.temp_length, F, row + (size_t)(1 + c) * F);
        }
        /* batch.t = t * ones_like(atoms) (ode_wrapper.py:112); PositionalEmbedding("t") */
        FN(posenc)((REAL)t, (REAL)m->d.time_length, F, row + (size_t)(nE - 1) * F);
    }
    /* K4 CombineInvariantFeatures (embedding.py:249-261) */
    FN(mlp)(&m->embed, big_in, A, s, tmp);
#define TAP(stage) do { if (tap_stage == (stage)) { \
        if (tap_s) for (size_t i = 0; i < (size_t)A * F; ++i) tap_s[i] = (float)s[i]; \
        if (tap_v) for (size_t i = 0; i < (size_t)A * F * 3; ++i) tap_v[i] = (float)v[i]; \
        if (tap_e) for (size_t i = 0; i < (size_t)E * F; ++i) tap_e[i] = (float)e[i]; } } while (0)
    TAP(0);
    /* positional encoding of the edge distances is layer independent (cpainn.py:284) */
    for (int k = 0; k < E; ++k) FN(posenc)(dist[k], (REAL)m->d.length_scale, F, enc + (size_t)k * F);

    for (int l = 0; l < L; ++l) {
        /* K5 SE3Message.forward (cpainn.py:263-310) */
        for (int k = 0; k < E; ++k) {
            memcpy(big_in + (size_t)k * 2 * F, s + (size_t)src[k] * F, sizeof(REAL) * F);
            memcpy(big_in + (size_t)k * 2 * F + F, e + (size_t)k * F, sizeof(REAL) * F);
        }
        FN(mlp)(&m->phi[l], big_in, E, phi_o, tmp);
        FN(mlp)(&m->w[l], enc, E, w_o, tmp);
        memset(ds, 0, sizeof(REAL) * (size_t)A * F);
        memset(dv, 0, sizeof(REAL) * (size_t)A * F * 3);
        for (int k = 0; k < E; ++k) {                       /* scatter-sum in edge-list order (torch_scatter CPU) */
            const REAL* h = phi_o + (size_t)k * 5 * F; const REAL* g = w_o + (size_t)k * 5 * F;
            const REAL* d3 = dir + k * 3;
            const REAL* vs = v + (size_t)src[k] * F * 3;    /* gated_features use v[src]   (cpainn.py:291) */
            const REAL* vd = v + (size_t)dst[k] * F * 3;    /* cross product uses v[dst]   (cpainn.py:296-298) */
            for (int f = 0; f < F; ++f) {
                const REAL gate = h[f] * g[f], sed = h[F + f] * g[F + f], dsf = h[2 * F + f] * g[2 * F + f],
                           def = h[3 * F + f] * g[3 * F + f], cg = h[4 * F + f] * g[4 * F + f];
                const REAL b0 = vd[f * 3], b1 = vd[f * 3 + 1], b2 = vd[f * 3 + 2];
                const REAL cr[3] = { d3[1] * b2 - d3[2] * b1, d3[2] * b0 - d3[0] * b2, d3[0] * b1 - d3[1] * b0 };
                for (int c = 0; c < 3; ++c)
                    dv[((size_t)dst[k] * F + f) * 3 + c] += (sed * d3[c] + gate * vs[f * 3 + c]) + cg * cr[c];
                ds[(size_t)dst[k] * F + f] += dsf;
                phi_o[(size_t)k * 5 * F + 3 * F + f] = def;   /* keep de for the edge update below */
            }
        }
        for (size_t i = 0; i < (size_t)A * F * 3; ++i) v[i] += dv[i];
        for (size_t i = 0; i < (size_t)A * F; ++i) s[i] += ds[i];
        for (int k = 0; k < E; ++k) for (int f = 0; f < F; ++f) e[(size_t)k * F + f] += phi_o[(size_t)k * 5 * F + 3 * F + f];
        TAP(1 + 2 * l);

        /* K7 Update.forward (cpainn.py:345-376); EquivariantLinear acts on the feature axis (cpainn.py:403) */
        for (int a = 0; a < A; ++a) {
            for (int c = 0; c < 3; ++c) {
                for (int o = 0; o < F; ++o) { vv[((size_t)a * F + o) * 3 + c] = 0; uv[((size_t)a * F + o) * 3 + c] = 0; }
                for (int k = 0; k < F; ++k) {
                    const REAL a_in = v[((size_t)a * F + k) * 3 + c];
                    const REAL* wv = m->Vt[l] + (size_t)k * F; const REAL* wu = m->Ut[l] + (size_t)k * F;
                    for (int o = 0; o < F; ++o) { vv[((size_t)a * F + o) * 3 + c] += a_in * wv[o]; uv[((size_t)a * F + o) * 3 + c] += a_in * wu[o]; }
                }
            }
            REAL* row = big_in + (size_t)a * 2 * F;
            for (int f = 0; f < F; ++f) {
                const REAL* q = vv + ((size_t)a * F + f) * 3;
                row[f] = RSQRT(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);      /* vv.norm(dim=-1) */
                row[F + f] = s[(size_t)a * F + f];
            }
        }
        FN(mlp)(&m->upd[l], big_in, A, upd_o, tmp);
        for (int a = 0; a < A; ++a) for (int f = 0; f < F; ++f) {
            const REAL gate = upd_o[(size_t)a * 3 * F + f], ssn = upd_o[(size_t)a * 3 * F + F + f], add = upd_o[(size_t)a * 3 * F + 2 * F + f];
            const REAL n = big_in[(size_t)a * 2 * F + f];
            s[(size_t)a * F + f] += (n * n) * ssn + add;                       /* vv_norm**2 * scale + add */
            for (int c = 0; c < 3; ++c) v[((size_t)a * F + f) * 3 + c] += uv[((size_t)a * F + f) * 3 + c] * gate;
        }
        TAP(2 + 2 * l);
    }
    /* K8 LayerReadout.forward (cpainn.py:425-437), n_features_out = 1; output = gate * (Vr . v) */
    FN(mlp)(&m->readout, s, A, upd_o, tmp);
    for (int a = 0; a < A; ++a) {
        const REAL gate = upd_o[a * 2 + 1];
        for (int c = 0; c < 3; ++c) {
            REAL acc = 0;
            for (int f = 0; f < F; ++f) acc += v[((size_t)a * F + f) * 3 + c] * m->Vr[f];
            out[a * 3 + c] = (float)(acc * gate);
        }
    }
#undef TAP
}

static size_t FN(painn_ws_size)(const FN(painn_t)* m)
{
    const size_t F = m->d.n_features, A = m->d.n_atoms, E = m->d.n_edges, nE = m->nE, M = E > A ? E : A;
    return A * F + A * F * 3 + E * F + E + E * 3 + E * F + M * (nE > 2 ? nE : 2) * F + 2 * E * 5 * F + M * 2 * F
         + A * F + 3 * (A * F * 3) + A * 3 * F + 64;
}

static int FN(painn_drift)(const FN(painn_t)* m, const int* src, const int* dst, const int* etype, const int* atom_ids,
                           const float* x, float t, const float* cond, long B, float* out,
                           int tap_stage, float* tap_s, float* tap_v, float* tap_e)
{
    const int A = m->d.n_atoms, F = m->d.n_features, E = m->d.n_edges;
    const size_t wsn = FN(painn_ws_size)(m);
    int fail = 0;
#pragma omp parallel
    {
        REAL* ws = (REAL*)malloc(sizeof(REAL) * wsn);
        if (!ws) {
#pragma omp atomic write
            fail = 1;
        } else {
#pragma omp for schedule(dynamic, 1)
            for (long b = 0; b < B; ++b)
                FN(painn_molecule)(m, src, dst, etype, atom_ids, x + (size_t)b * A * 3, t,
                                   cond ? cond + (size_t)b * A * m->ncond : NULL, out + (size_t)b * A * 3, tap_stage,
                                   tap_s ? tap_s + (size_t)b * A * F : NULL, tap_v ? tap_v + (size_t)b * A * F * 3 : NULL,
                                   tap_e ? tap_e + (size_t)b * E * F : NULL, ws);
            free(ws);
        }
    }
    return fail ? -4 : 0;
}

/* ------------------------------------------------------------------------------------- forward-mode twin (JVP)
 * The reference obtains the divergence with 3A reverse-mode passes over the same graph
 * (ODEWrapper.compute_divergence, mdqm9/thermo/ambient/models/ode_wrapper.py:59-91: sum_ij d b_ij / d x_ij).  The
 * restatement differentiates the forward pass above line by line in forward mode (one pass per seed direction); both
 * give the exact derivative of the same arithmetic, the summation orders differ. */
static void FN(ln_silu_jvp)(REAL* x, REAL* dx, int rows, int F, const REAL* g, const REAL* be)
{
    for (int r = 0; r < rows; ++r) {
        REAL* xr = x + (size_t)r * F; REAL* dr = dx + (size_t)r * F;
        REAL mean = 0, var = 0, dmean = 0, proj = 0;
        for (int i = 0; i < F; ++i) { mean += xr[i]; dmean += dr[i]; }
        mean /= (REAL)F; dmean /= (REAL)F;
        for (int i = 0; i < F; ++i) { REAL d = xr[i] - mean; var += d * d; }
        var /= (REAL)F;
        const REAL rstd = (REAL)1 / RSQRT(var + (REAL)1e-5);
        for (int i = 0; i < F; ++i) proj += (xr[i] - mean) * rstd * (dr[i] - dmean);
        proj /= (REAL)F;
        for (int i = 0; i < F; ++i) {
            const REAL n = (xr[i] - mean) * rstd;
            const REAL dn = rstd * ((dr[i] - dmean) - n * proj);
            const REAL v = n * g[i] + be[i], dv = dn * g[i];
            const REAL sg = (REAL)1 / ((REAL)1 + REXP(-v));
            xr[i] = v * sg;
            dr[i] = dv * sg * ((REAL)1 + v * ((REAL)1 - sg));
        }
    }
}

/* tmp: rows*f_h*4 scratch */
static void FN(mlp_jvp)(const FN(mlp_t)* m, const REAL* x, const REAL* dx, int rows, REAL* y, REAL* dy, REAL* tmp)
{
    REAL* h1 = tmp; REAL* h2 = h1 + (size_t)rows * m->f_h; REAL* d1 = h2 + (size_t)rows * m->f_h; REAL* d2 = d1 + (size_t)rows * m->f_h;
    FN(linear)(x, rows, m->f_in, m->f_in, m->W0t, m->b0, m->f_h, h1, m->f_h);
    FN(linear)(dx, rows, m->f_in, m->f_in, m->W0t, NULL, m->f_h, d1, m->f_h);
    FN(ln_silu_jvp)(h1, d1, rows, m->f_h, m->g0, m->be0);
    FN(linear)(h1, rows, m->f_h, m->f_h, m->W1t, m->b1, m->f_h, h2, m->f_h);
    FN(linear)(d1, rows, m->f_h, m->f_h, m->W1t, NULL, m->f_h, d2, m->f_h);
    FN(ln_silu_jvp)(h2, d2, rows, m->f_h, m->g1, m->be1);
    FN(linear)(h2, rows, m->f_h, m->f_h, m->W2t, m->b2, m->f_out, y, m->f_out);
    FN(linear)(d2, rows, m->f_h, m->f_h, m->W2t, NULL, m->f_out, dy, m->f_out);
}

static void FN(posenc_jvp)(REAL x, REAL dx, REAL max_length, int F, REAL* out, REAL* dout)
{
    const REAL xs = x / max_length, dxs = dx / max_length;
    for (int k = 1; k <= F / 2; ++k) {
        const REAL a = (xs * (REAL)k) * (REAL)M_PI, da = (dxs * (REAL)k) * (REAL)M_PI;
        const REAL c = RCOS(a), sn = RSIN(a);
        out[2 * (k - 1)] = c;       dout[2 * (k - 1)] = -sn * da;
        out[2 * (k - 1) + 1] = sn;  dout[2 * (k - 1) + 1] = c * da;
    }
}

/* One molecule, one seed direction xdot [A][3]: out = b(x) [A][3], out_tan = (d b / d x) xdot [A][3].
 * Tangent taps (may be NULL) after `tap_stage` as in painn_molecule: ds [A][F], dv [A][F][3], de [E][F]. */
static void FN(painn_molecule_jvp)(const FN(painn_t)* m, const int* src, const int* dst, const int* etype, const int* atom_ids,
                                   const float* x, const float* xdot, float t, const float* cond, float* out, float* out_tan,
                                   REAL* out_tan_real, int tap_stage, float* tap_s, float* tap_v, float* tap_e, REAL* ws)
{
    const int F = m->d.n_features, L = m->d.n_layers, A = m->d.n_atoms, E = m->d.n_edges, nE = m->nE;
    const size_t M = E > A ? E : A;
#define TAKE(name, n) REAL* name = ws; ws += (size_t)(n); REAL* CAT(d_, name) = ws; ws += (size_t)(n)
    TAKE(s, (size_t)A * F); TAKE(v, (size_t)A * F * 3); TAKE(e, (size_t)E * F); TAKE(dist, E); TAKE(dir, (size_t)E * 3);
    TAKE(enc, (size_t)E * F); TAKE(big_in, M * (nE > 2 ? nE : 2) * F); TAKE(phi_o, (size_t)E * 5 * F); TAKE(w_o, (size_t)E * 5 * F);
    TAKE(acc_s, (size_t)A * F); TAKE(acc_v, (size_t)A * F * 3); TAKE(vv, (size_t)A * F * 3); TAKE(uv, (size_t)A * F * 3);
    TAKE(upd_o, (size_t)A * 3 * F);
#undef TAKE
    REAL* tmp = ws;               /* M * 4 * F */

    for (int k = 0; k < E; ++k) {                                                   /* K1 */
        REAL r[3], dr[3]; REAL n2 = 0, rd = 0;
        for (int c = 0; c < 3; ++c) {
            r[c] = (REAL)x[src[k] * 3 + c] - (REAL)x[dst[k] * 3 + c];
            dr[c] = (REAL)xdot[src[k] * 3 + c] - (REAL)xdot[dst[k] * 3 + c];
            n2 += r[c] * r[c]; rd += r[c] * dr[c];
        }
        dist[k] = RSQRT(n2);
        d_dist[k] = dist[k] > 0 ? rd / dist[k] : 0;
        const REAL q = (REAL)1 + dist[k];
        for (int c = 0; c < 3; ++c) { dir[k * 3 + c] = r[c] / q; d_dir[k * 3 + c] = dr[c] / q - r[c] * d_dist[k] / (q * q); }
    }
    memset(v, 0, sizeof(REAL) * (size_t)A * F * 3); memset(d_v, 0, sizeof(REAL) * (size_t)A * F * 3);      /* K2 */
    for (int k = 0; k < E; ++k) memcpy(e + (size_t)k * F, m->edge_emb + (size_t)etype[k] * F, sizeof(REAL) * F);   /* K3 */
    memset(d_e, 0, sizeof(REAL) * (size_t)E * F);
    for (int a = 0; a < A; ++a) {
        REAL* row = big_in + (size_t)a * nE * F;
        memcpy(row, m->atom_emb + (size_t)atom_ids[a] * F, sizeof(REAL) * F);
        for (int c = 0; c < m->ncond; ++c) {
            REAL u = (REAL)cond[a * m->ncond + c] - (REAL)m->d.temp_mean * (REAL)1;
            u = u / (REAL)m->d.temp_range;
            FN(posenc)(u, (REAL)m->d.temp_length, F, row + (size_t)(1 + c) * F);
        }
        FN(posenc)((REAL)t, (REAL)m->d.time_length, F, row + (size_t)(nE - 1) * F);
    }
    FN(mlp)(&m->embed, big_in, A, s, tmp);                                          /* K4: no dependence on x */
    memset(d_s, 0, sizeof(REAL) * (size_t)A * F);
#define TAPJ(stage) do { if (tap_stage == (stage)) { \
        if (tap_s) for (size_t i = 0; i < (size_t)A * F; ++i) tap_s[i] = (float)d_s[i]; \
        if (tap_v) for (size_t i = 0; i < (size_t)A * F * 3; ++i) tap_v[i] = (float)d_v[i]; \
        if (tap_e) for (size_t i = 0; i < (size_t)E * F; ++i) tap_e[i] = (float)d_e[i]; } } while (0)
    TAPJ(0);
    for (int k = 0; k < E; ++k) FN(posenc_jvp)(dist[k], d_dist[k], (REAL)m->d.length_scale, F, enc + (size_t)k * F, d_enc + (size_t)k * F);

    for (int l = 0; l < L; ++l) {
        for (int k = 0; k < E; ++k) {                                               /* K5 */
            memcpy(big_in + (size_t)k * 2 * F, s + (size_t)src[k] * F, sizeof(REAL) * F);
            memcpy(big_in + (size_t)k * 2 * F + F, e + (size_t)k * F, sizeof(REAL) * F);
            memcpy(d_big_in + (size_t)k * 2 * F, d_s + (size_t)src[k] * F, sizeof(REAL) * F);
            memcpy(d_big_in + (size_t)k * 2 * F + F, d_e + (size_t)k * F, sizeof(REAL) * F);
        }
        FN(mlp_jvp)(&m->phi[l], big_in, d_big_in, E, phi_o, d_phi_o, tmp);
        FN(mlp_jvp)(&m->w[l], enc, d_enc, E, w_o, d_w_o, tmp);
        memset(acc_s, 0, sizeof(REAL) * (size_t)A * F); memset(d_acc_s, 0, sizeof(REAL) * (size_t)A * F);
        memset(acc_v, 0, sizeof(REAL) * (size_t)A * F * 3); memset(d_acc_v, 0, sizeof(REAL) * (size_t)A * F * 3);
        for (int k = 0; k < E; ++k) {
            const REAL* h = phi_o + (size_t)k * 5 * F; const REAL* g = w_o + (size_t)k * 5 * F;
            const REAL* dh = d_phi_o + (size_t)k * 5 * F; const REAL* dg = d_w_o + (size_t)k * 5 * F;
            const REAL* d3 = dir + k * 3; const REAL* dd3 = d_dir + k * 3;
            const REAL* vs = v + (size_t)src[k] * F * 3; const REAL* dvs = d_v + (size_t)src[k] * F * 3;
            const REAL* vd = v + (size_t)dst[k] * F * 3; const REAL* dvd = d_v + (size_t)dst[k] * F * 3;
            for (int f = 0; f < F; ++f) {
#define PROD(i) h[(i) * F + f] * g[(i) * F + f]
#define DPROD(i) (dh[(i) * F + f] * g[(i) * F + f] + h[(i) * F + f] * dg[(i) * F + f])
                const REAL gate = PROD(0), sed = PROD(1), dsf = PROD(2), def = PROD(3), cg = PROD(4);
                const REAL dgate = DPROD(0), dsed = DPROD(1), ddsf = DPROD(2), ddef = DPROD(3), dcg = DPROD(4);
#undef PROD
#undef DPROD
                const REAL b[3] = { vd[f * 3], vd[f * 3 + 1], vd[f * 3 + 2] }, db[3] = { dvd[f * 3], dvd[f * 3 + 1], dvd[f * 3 + 2] };
                const REAL cr[3] = { d3[1] * b[2] - d3[2] * b[1], d3[2] * b[0] - d3[0] * b[2], d3[0] * b[1] - d3[1] * b[0] };
                const REAL dcr[3] = { dd3[1] * b[2] + d3[1] * db[2] - dd3[2] * b[1] - d3[2] * db[1],
                                      dd3[2] * b[0] + d3[2] * db[0] - dd3[0] * b[2] - d3[0] * db[2],
                                      dd3[0] * b[1] + d3[0] * db[1] - dd3[1] * b[0] - d3[1] * db[0] };
                for (int c = 0; c < 3; ++c) {
                    acc_v[((size_t)dst[k] * F + f) * 3 + c] += (sed * d3[c] + gate * vs[f * 3 + c]) + cg * cr[c];
                    d_acc_v[((size_t)dst[k] * F + f) * 3 + c] += (dsed * d3[c] + sed * dd3[c] + dgate * vs[f * 3 + c] + gate * dvs[f * 3 + c])
                                                               + (dcg * cr[c] + cg * dcr[c]);
                }
                acc_s[(size_t)dst[k] * F + f] += dsf; d_acc_s[(size_t)dst[k] * F + f] += ddsf;
                phi_o[(size_t)k * 5 * F + 3 * F + f] = def; d_phi_o[(size_t)k * 5 * F + 3 * F + f] = ddef;
            }
        }
        for (size_t i = 0; i < (size_t)A * F * 3; ++i) { v[i] += acc_v[i]; d_v[i] += d_acc_v[i]; }
        for (size_t i = 0; i < (size_t)A * F; ++i) { s[i] += acc_s[i]; d_s[i] += d_acc_s[i]; }
        for (int k = 0; k < E; ++k) for (int f = 0; f < F; ++f) {
            e[(size_t)k * F + f] += phi_o[(size_t)k * 5 * F + 3 * F + f]; d_e[(size_t)k * F + f] += d_phi_o[(size_t)k * 5 * F + 3 * F + f];
        }
        TAPJ(1 + 2 * l);

        for (int a = 0; a < A; ++a) {                                               /* K7 */
            for (int c = 0; c < 3; ++c) {
                for (int o = 0; o < F; ++o) {
                    const size_t i = ((size_t)a * F + o) * 3 + c; vv[i] = 0; uv[i] = 0; d_vv[i] = 0; d_uv[i] = 0;
                }
                for (int k = 0; k < F; ++k) {
                    const REAL a_in = v[((size_t)a * F + k) * 3 + c], da_in = d_v[((size_t)a * F + k) * 3 + c];
                    const REAL* wv = m->Vt[l] + (size_t)k * F; const REAL* wu = m->Ut[l] + (size_t)k * F;
                    for (int o = 0; o < F; ++o) {
                        const size_t i = ((size_t)a * F + o) * 3 + c;
                        vv[i] += a_in * wv[o]; uv[i] += a_in * wu[o]; d_vv[i] += da_in * wv[o]; d_uv[i] += da_in * wu[o];
                    }
                }
            }
            REAL* row = big_in + (size_t)a * 2 * F; REAL* drow = d_big_in + (size_t)a * 2 * F;
            for (int f = 0; f < F; ++f) {
                const REAL* q = vv + ((size_t)a * F + f) * 3; const REAL* dq = d_vv + ((size_t)a * F + f) * 3;
                row[f] = RSQRT(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
                drow[f] = row[f] > 0 ? (q[0] * dq[0] + q[1] * dq[1] + q[2] * dq[2]) / row[f] : 0;   /* torch: d|q| = 0 at q = 0 */
                row[F + f] = s[(size_t)a * F + f]; drow[F + f] = d_s[(size_t)a * F + f];
            }
        }
        FN(mlp_jvp)(&m->upd[l], big_in, d_big_in, A, upd_o, d_upd_o, tmp);
        for (int a = 0; a < A; ++a) for (int f = 0; f < F; ++f) {
            const size_t o = (size_t)a * 3 * F;
            const REAL gate = upd_o[o + f], ssn = upd_o[o + F + f], add = upd_o[o + 2 * F + f];
            const REAL dgate = d_upd_o[o + f], dssn = d_upd_o[o + F + f], dadd = d_upd_o[o + 2 * F + f];
            const REAL n = big_in[(size_t)a * 2 * F + f], dn = d_big_in[(size_t)a * 2 * F + f];
            s[(size_t)a * F + f] += (n * n) * ssn + add;
            d_s[(size_t)a * F + f] += ((REAL)2 * n * dn) * ssn + (n * n) * dssn + dadd;
            for (int c = 0; c < 3; ++c) {
                const size_t i = ((size_t)a * F + f) * 3 + c;
                v[i] += uv[i] * gate; d_v[i] += d_uv[i] * gate + uv[i] * dgate;
            }
        }
        TAPJ(2 + 2 * l);
    }
    FN(mlp_jvp)(&m->readout, s, d_s, A, upd_o, d_upd_o, tmp);                       /* K8 */
    for (int a = 0; a < A; ++a) {
        const REAL gate = upd_o[a * 2 + 1], dgate = d_upd_o[a * 2 + 1];
        for (int c = 0; c < 3; ++c) {
            REAL acc = 0, dacc = 0;
            for (int f = 0; f < F; ++f) { acc += v[((size_t)a * F + f) * 3 + c] * m->Vr[f]; dacc += d_v[((size_t)a * F + f) * 3 + c] * m->Vr[f]; }
            out[a * 3 + c] = (float)(acc * gate);
            out_tan[a * 3 + c] = (float)(dacc * gate + acc * dgate);
            if (out_tan_real) out_tan_real[a * 3 + c] = dacc * gate + acc * dgate;
        }
    }
#undef TAPJ
}

static size_t FN(painn_jvp_ws_size)(const FN(painn_t)* m)
{
    const size_t F = m->d.n_features, A = m->d.n_atoms, E = m->d.n_edges, nE = m->nE, M = E > A ? E : A;
    return 2 * (A * F + A * F * 3 + E * F + E + E * 3 + E * F + M * (nE > 2 ? nE : 2) * F + 2 * E * 5 * F + A * F + 3 * (A * F * 3) + A * 3 * F)
         + M * 4 * F + 64;
}

/* xdot [B][A][3] -> out, out_tan [B][A][3] */
static int FN(painn_jvp)(const FN(painn_t)* m, const int* src, const int* dst, const int* etype, const int* atom_ids,
                         const float* x, const float* xdot, float t, const float* cond, long B, float* out, float* out_tan,
                         int tap_stage, float* tap_s, float* tap_v, float* tap_e)
{
    const int A = m->d.n_atoms, F = m->d.n_features, E = m->d.n_edges;
    const size_t wsn = FN(painn_jvp_ws_size)(m);
    int fail = 0;
#pragma omp parallel
    {
        REAL* ws = (REAL*)malloc(sizeof(REAL) * wsn);
        if (!ws) {
#pragma omp atomic write
            fail = 1;
        } else {
#pragma omp for schedule(dynamic, 1)
            for (long b = 0; b < B; ++b)
                FN(painn_molecule_jvp)(m, src, dst, etype, atom_ids, x + (size_t)b * A * 3, xdot + (size_t)b * A * 3, t,
                                       cond ? cond + (size_t)b * A * m->ncond : NULL, out + (size_t)b * A * 3, out_tan + (size_t)b * A * 3,
                                       NULL, tap_stage, tap_s ? tap_s + (size_t)b * A * F : NULL, tap_v ? tap_v + (size_t)b * A * F * 3 : NULL,
                                       tap_e ? tap_e + (size_t)b * E * F : NULL, ws);
            free(ws);
        }
    }
    return fail ? -4 : 0;
}

/* out [B][A][3] = b(x); div [B] = sum_{a,c} d b[a][c] / d x[a][c]  (3A unit seeds per molecule; the diagonal entries are
 * summed in (a, c) order like the reference's double loop, ode_wrapper.py:80-84; no 1e-2 factor here).  div is returned
 * as double so that the REAL = double build keeps its digits; the REAL = float build accumulates in float like torch. */
static int FN(painn_div)(const FN(painn_t)* m, const int* src, const int* dst, const int* etype, const int* atom_ids,
                         const float* x, float t, const float* cond, long B, float* out, double* div)
{
    const int A = m->d.n_atoms, D = 3 * A;
    const size_t wsn = FN(painn_jvp_ws_size)(m);
    REAL* diag = (REAL*)malloc(sizeof(REAL) * (size_t)B * D);
    int fail = diag ? 0 : 1;
    if (!fail) {
#pragma omp parallel
        {
            REAL* ws = (REAL*)malloc(sizeof(REAL) * wsn);
            float* seed = (float*)calloc((size_t)D, sizeof(float)); float* tan = (float*)malloc(sizeof(float) * 2 * D);
            REAL* tan_real = (REAL*)malloc(sizeof(REAL) * D);
            if (!ws || !seed || !tan || !tan_real) {
#pragma omp atomic write
                fail = 1;
            } else {
#pragma omp for schedule(dynamic, 1)
                for (long job = 0; job < B * D; ++job) {
                    const long b = job / D; const int k = (int)(job % D);
                    seed[k] = 1.0f;
                    FN(painn_molecule_jvp)(m, src, dst, etype, atom_ids, x + (size_t)b * D, seed, t, cond ? cond + (size_t)b * A * m->ncond : NULL,
                                           tan + D, tan, tan_real, -1, NULL, NULL, NULL, ws);
                    seed[k] = 0.0f;
                    diag[job] = tan_real[k];
                    if (k == 0) memcpy(out + (size_t)b * D, tan + D, sizeof(float) * D);
                }
            }
            free(ws); free(seed); free(tan); free(tan_real);
        }
        for (long b = 0; b < B && !fail; ++b) {
            REAL acc = 0;
            for (int k = 0; k < D; ++k) acc += diag[(size_t)b * D + k];
            div[b] = (double)acc;
        }
    }
    free(diag);
    return fail ? -4 : 0;
}

/* ------------------------------------------------------------------------------------------------------------ adw */
typedef struct { int H, nl; REAL *be_W0t, *be_b0, *be_W1t, *be_b1, *be_W2, *be_b2; REAL **Wt, **b; REAL* store; } FN(adw_t);

static FN(adw_t)* FN(adw_parse)(const ti_adw_desc* d, const double* w, size_t n)
{
    const int H = d->hidden_size, nl = d->num_layers;
    const size_t need = (size_t)H * 3 + H + (size_t)H * H + H + H + 1 + (size_t)H * 3 + H + (size_t)(nl - 1) * ((size_t)H * H + H) + H + 1;
    if (n != need) return NULL;
    FN(adw_t)* m = calloc(1, sizeof(*m)); m->H = H; m->nl = nl;
    m->store = malloc(sizeof(REAL) * n); m->Wt = calloc(nl + 1, sizeof(REAL*)); m->b = calloc(nl + 1, sizeof(REAL*));
    const double* p = w; REAL* q = m->store;
#define T2(dst, rows, cols) do { dst = q; for (int r = 0; r < (rows); ++r) for (int c = 0; c < (cols); ++c) q[(size_t)c * (rows) + r] = (REAL)p[(size_t)r * (cols) + c]; q += (size_t)(rows) * (cols); p += (size_t)(rows) * (cols); } while (0)
#define V1(dst, nn) do { dst = q; for (int i = 0; i < (nn); ++i) q[i] = (REAL)p[i]; q += (nn); p += (nn); } while (0)
    T2(m->be_W0t, H, 3); V1(m->be_b0, H); T2(m->be_W1t, H, H); V1(m->be_b1, H); V1(m->be_W2, H); V1(m->be_b2, 1);
    T2(m->Wt[0], H, 3); V1(m->b[0], H);
    for (int i = 1; i < nl; ++i) { T2(m->Wt[i], H, H); V1(m->b[i], H); }
    V1(m->Wt[nl], H); V1(m->b[nl], 1);
    return m;
}

static void FN(silu_rows)(REAL* x, size_t n) { for (size_t i = 0; i < n; ++i) x[i] = x[i] / ((REAL)1 + REXP(-x[i])); }

/* FCNetMultiBeta.forward (simple.py:38-41) through ODEWrapper.forward (adw ode_wrapper.py:47-52): ts = ones_like(x)*t.
 * If `div` is not NULL it receives d b / d x (ODEWrapper.compute_divergence, ode_wrapper.py:55-67, WITHOUT its 1e-2 factor),
 * by forward-mode differentiation of `net` (beta_embed does not depend on x): tangent of [x, t, emb] is [1, 0, 0]. */
static void FN(adw_drift)(const FN(adw_t)* m, const REAL* x, REAL t, const REAL* beta0, const REAL* beta1, long B, REAL* out, REAL* div)
{
    const int H = m->H;
#pragma omp parallel
    {
        REAL* h1 = malloc(sizeof(REAL) * 4 * H); REAL* h2 = h1 + H; REAL* d1 = h2 + H; REAL* d2 = d1 + H;
#pragma omp for
        for (long i = 0; i < B; ++i) {
            REAL in[3] = { beta0[i], beta1[i], t }, emb;
            FN(linear)(in, 1, 3, 3, m->be_W0t, m->be_b0, H, h1, H); FN(silu_rows)(h1, H);
            FN(linear)(h1, 1, H, H, m->be_W1t, m->be_b1, H, h2, H); FN(silu_rows)(h2, H);
            emb = m->be_b2[0]; for (int k = 0; k < H; ++k) emb += h2[k] * m->be_W2[k];
            REAL in2[3] = { x[i], t, emb };
            FN(linear)(in2, 1, 3, 3, m->Wt[0], m->b[0], H, h1, H);
            /* silu'(z) = sig(z) * (1 + z * (1 - sig(z))) */
            for (int k = 0; k < H; ++k) { const REAL z = h1[k], sg = (REAL)1 / ((REAL)1 + REXP(-z)); d1[k] = sg * ((REAL)1 + z * ((REAL)1 - sg)) * m->Wt[0][k]; h1[k] = z * sg; }
            REAL *a = h1, *b = h2, *da = d1, *db = d2;
            for (int l = 1; l < m->nl; ++l) {
                FN(linear)(a, 1, H, H, m->Wt[l], m->b[l], H, b, H);
                FN(linear)(da, 1, H, H, m->Wt[l], NULL, H, db, H);
                for (int k = 0; k < H; ++k) { const REAL z = b[k], sg = (REAL)1 / ((REAL)1 + REXP(-z)); db[k] = sg * ((REAL)1 + z * ((REAL)1 - sg)) * db[k]; b[k] = z * sg; }
                REAL* tsw = a; a = b; b = tsw; tsw = da; da = db; db = tsw;
            }
            REAL o = m->b[m->nl][0], dd = 0;
            for (int k = 0; k < H; ++k) { o += a[k] * m->Wt[m->nl][k]; dd += da[k] * m->Wt[m->nl][k]; }
            out[i] = o;
            if (div) div[i] = dd;
        }
        free(h1);
    }
}

#undef FN
#undef CAT
#undef CAT_
