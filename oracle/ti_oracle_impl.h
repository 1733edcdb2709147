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
