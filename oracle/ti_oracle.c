/*
 * ti_oracle.c -- CPU restatement of the thermodynamic-interpolation sampling hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (thermodynamic-interpolation_amd/, libti_hip.so) may
 * import, link or call this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do,
 * and there only as the checker / the timed CPU baseline.
 *
 * Parity status: PINNED for the drift networks -- checked against golden vectors produced by the reference's own
 * PyTorch modules (tests/golden/make_golden.py, tests/test_oracle_golden.py).  UNPINNED for the time stepping:
 * the reference integrates with torchdiffeq (absent here); Euler/Heun below are build-defined and
 * are checked against hand-rolled loops over the reference ODEWrapper in the same golden files.
 *
 * Build: see oracle/Makefile  (gcc -O3 -fopenmp -ffp-contract=off -shared -fPIC).
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/ti_hip.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

#if defined(__x86_64__) && defined(__GNUC__) && !defined(__clang__)
#define TIO_CLONES __attribute__((target_clones("avx512f", "avx2", "default")))
#else
#define TIO_CLONES
#endif

#define REAL float
#define SUFFIX _f32
#define RSQRT sqrtf
#define REXP expf
#define RSIN sinf
#define RCOS cosf
#include "ti_oracle_impl.h"
#undef REAL
#undef SUFFIX
#undef RSQRT
#undef REXP
#undef RSIN
#undef RCOS

#define REAL double
#define SUFFIX _f64
#define RSQRT sqrt
#define REXP exp
#define RSIN sin
#define RCOS cos
#include "ti_oracle_impl.h"
#undef REAL
#undef SUFFIX
#undef RSQRT
#undef REXP
#undef RSIN
#undef RCOS

/* ---------------------------------------------------------------------------------------------- Philox4x32-10
 * Build-defined noise source of the EM scheme (include/ti_hip.h, TI_SCHEME_EM); identical on host and device. */
static inline void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1)
{
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

/* standard normal for (seed, trajectory, step, component): component c uses Philox block c/4, Box-Muller pair (c%4)/2 */
static float ti_normal(uint64_t seed, int64_t traj, int32_t step, int32_t comp)
{
    uint32_t c[4] = { (uint32_t)traj, (uint32_t)((uint64_t)traj >> 32), (uint32_t)step, (uint32_t)(comp >> 2) };
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    const int pair = (comp & 3) >> 1;
    const float u1 = ((float)(c[2 * pair] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float u2 = ((float)(c[2 * pair + 1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float r = sqrtf(-2.0f * logf(u1)), a = 6.283185307179586f * u2;
    return (comp & 1) ? r * sinf(a) : r * cosf(a);
}

float tio_normal(uint64_t seed, int64_t traj, int32_t step, int32_t comp) { return ti_normal(seed, traj, step, comp); }

int64_t tio_rollout_rows(int32_t n_step, int32_t save_every)
{
    if (save_every <= 0) return 1;
    const int64_t steps = n_step - 1;
    return steps / save_every + 1 + (steps % save_every != 0);
}

/* ---------------------------------------------------------------------------------------------------- painn API */
typedef struct {
    ti_painn_desc d;
    painn_t_f32* m32; painn_t_f64* m64;
    int *src, *dst, *etype, *atom_ids;
} tio_painn;

void* tio_painn_create(const ti_painn_desc* d, const float* weights, size_t n, const int32_t* src, const int32_t* dst,
                       const int32_t* etype, const int32_t* atom_ids)
{
    if (!d || !weights || d->n_features <= 0 || d->n_features % 2) return NULL;
    tio_painn* h = calloc(1, sizeof(*h));
    h->d = *d;
    h->m32 = painn_parse_f32(d, weights, n);
    h->m64 = painn_parse_f64(d, weights, n);
    if (!h->m32 || !h->m64) { painn_free_f32(h->m32); painn_free_f64(h->m64); free(h); return NULL; }
    const int E = d->n_edges, A = d->n_atoms;
    h->src = malloc(sizeof(int) * E); h->dst = malloc(sizeof(int) * E); h->etype = malloc(sizeof(int) * E); h->atom_ids = malloc(sizeof(int) * A);
    memcpy(h->src, src, sizeof(int) * E); memcpy(h->dst, dst, sizeof(int) * E); memcpy(h->etype, etype, sizeof(int) * E);
    memcpy(h->atom_ids, atom_ids, sizeof(int) * A);
    for (int k = 0; k < E; ++k) if (src[k] < 0 || src[k] >= A || dst[k] < 0 || dst[k] >= A || etype[k] < 0 || etype[k] > 3) { free(h); return NULL; }
    return h;
}

void tio_painn_destroy(void* hv)
{
    tio_painn* h = hv; if (!h) return;
    painn_free_f32(h->m32); painn_free_f64(h->m64); free(h->src); free(h->dst); free(h->etype); free(h->atom_ids); free(h);
}

/* precision: 32 = fp32 arithmetic (the reference's), 64 = fp64 arithmetic.  taps may be NULL. */
int tio_painn_drift(void* hv, int precision, const float* x, float t, const float* cond, int64_t B, float* out,
                    int tap_stage, float* tap_s, float* tap_v, float* tap_e)
{
    tio_painn* h = hv;
    if (precision == 64)
        return painn_drift_f64(h->m64, h->src, h->dst, h->etype, h->atom_ids, x, t, cond, B, out, tap_stage, tap_s, tap_v, tap_e);
    return painn_drift_f32(h->m32, h->src, h->dst, h->etype, h->atom_ids, x, t, cond, B, out, tap_stage, tap_s, tap_v, tap_e);
}

/* forward-mode derivative along xdot [B,A,3]; tangent taps like tio_painn_drift */
int tio_painn_jvp(void* hv, int precision, const float* x, const float* xdot, float t, const float* cond, int64_t B, float* out,
                  float* out_tan, int tap_stage, float* tap_s, float* tap_v, float* tap_e)
{
    tio_painn* h = hv;
    if (precision == 64)
        return painn_jvp_f64(h->m64, h->src, h->dst, h->etype, h->atom_ids, x, xdot, t, cond, B, out, out_tan, tap_stage, tap_s, tap_v, tap_e);
    return painn_jvp_f32(h->m32, h->src, h->dst, h->etype, h->atom_ids, x, xdot, t, cond, B, out, out_tan, tap_stage, tap_s, tap_v, tap_e);
}

/* drift and exact divergence sum_ij d b_ij / d x_ij (ODEWrapper.compute_divergence without its 1e-2 factor) */
int tio_painn_drift_div(void* hv, int precision, const float* x, float t, const float* cond, int64_t B, float* out, double* div)
{
    tio_painn* h = hv;
    if (precision == 64) return painn_div_f64(h->m64, h->src, h->dst, h->etype, h->atom_ids, x, t, cond, B, out, div);
    return painn_div_f32(h->m32, h->src, h->dst, h->etype, h->atom_ids, x, t, cond, B, out, div);
}

/* MoleculeIntegrator.rollout(return_dlogp=True) on a fixed grid (ambient/integrators.py:36-68, latent/integrators.py:57-89):
 * states (x, dlogp), func = (b, -div_scale*div), or (-b, +div_scale*div) with reverse_ode (ode_wrapper.py:49; the grid is
 * then linspace(end, start)); out_dlogp [rows,B] holds the raw second state -- the ambient caller multiplies by 1e2. */
int tio_painn_rollout_dlogp(void* hv, int precision, const ti_rollout_desc* rd, const float* x0, const float* cond, int64_t B,
                            float div_scale, int reverse_ode, float* out_path, float* out_dlogp, int64_t* n_fevals)
{
    tio_painn* h = hv;
    const int A = h->d.n_atoms; const size_t n = (size_t)B * A * 3;
    if (rd->scheme == TI_SCHEME_EM) return -1;
    float *x = malloc(sizeof(float) * n), *b1 = malloc(sizeof(float) * n), *b2 = malloc(sizeof(float) * n), *xt = malloc(sizeof(float) * n);
    float* dl = calloc((size_t)B, sizeof(float)); double *d1 = malloc(8 * (size_t)B), *d2 = malloc(8 * (size_t)B);
    const float sb = reverse_ode ? -1.0f : 1.0f, sd = reverse_ode ? 1.0f : -1.0f;
    memcpy(x, x0, sizeof(float) * n);
    int64_t row = 0, fe = 0;
#define SAVE_ROW() do { memcpy(out_dlogp + row * B, dl, sizeof(float) * (size_t)B); memcpy(out_path + (row++) * n, x, sizeof(float) * n); } while (0)
    if (rd->save_every > 0) SAVE_ROW();
    for (int k = 0; k < rd->n_step - 1; ++k) {
        const float dt = rd->t_grid[k + 1] - rd->t_grid[k];
        tio_painn_drift_div(hv, precision, x, rd->t_grid[k], cond, B, b1, d1); ++fe;
        if (rd->scheme == TI_SCHEME_HEUN) {
            for (size_t i = 0; i < n; ++i) xt[i] = x[i] + dt * (sb * b1[i]);
            tio_painn_drift_div(hv, precision, xt, rd->t_grid[k + 1], cond, B, b2, d2); ++fe;
            const float hdt = 0.5f * dt;
            for (size_t i = 0; i < n; ++i) x[i] = x[i] + hdt * (sb * b1[i] + sb * b2[i]);
            for (int64_t i = 0; i < B; ++i) dl[i] = dl[i] + hdt * (sd * ((float)d1[i] * div_scale) + sd * ((float)d2[i] * div_scale));
        } else {
            for (size_t i = 0; i < n; ++i) x[i] = x[i] + dt * (sb * b1[i]);
            for (int64_t i = 0; i < B; ++i) dl[i] = dl[i] + dt * (sd * ((float)d1[i] * div_scale));
        }
        const int step = k + 1;
        if (rd->save_every > 0 && (step % rd->save_every == 0 || step == rd->n_step - 1)) SAVE_ROW();
    }
    if (rd->save_every <= 0) SAVE_ROW();
#undef SAVE_ROW
    if (n_fevals) *n_fevals = fe;
    free(x); free(b1); free(b2); free(xt); free(dl); free(d1); free(d2);
    return 0;
}

/* Fixed-step rollout on grid t[0..n_step-1]; out_path [rows,B,A,3] as ti_painn_rollout. fp32 state like the reference. */
int tio_painn_rollout(void* hv, int precision, const ti_rollout_desc* rd, const float* x0, const float* cond, int64_t B,
                      float* out_path, int64_t* n_fevals)
{
    tio_painn* h = hv;
    const int A = h->d.n_atoms; const size_t n = (size_t)B * A * 3;
    float *x = malloc(sizeof(float) * n), *b1 = malloc(sizeof(float) * n), *b2 = malloc(sizeof(float) * n), *xt = malloc(sizeof(float) * n);
    memcpy(x, x0, sizeof(float) * n);
    int64_t row = 0, fe = 0;
    if (rd->save_every > 0) memcpy(out_path + (row++) * n, x, sizeof(float) * n);
    for (int k = 0; k < rd->n_step - 1; ++k) {
        const float dt = rd->t_grid[k + 1] - rd->t_grid[k];
        tio_painn_drift(hv, precision, x, rd->t_grid[k], cond, B, b1, -1, NULL, NULL, NULL); ++fe;
        if (rd->scheme == TI_SCHEME_HEUN) {
            for (size_t i = 0; i < n; ++i) xt[i] = x[i] + dt * b1[i];
            tio_painn_drift(hv, precision, xt, rd->t_grid[k + 1], cond, B, b2, -1, NULL, NULL, NULL); ++fe;
            const float hdt = 0.5f * dt;
            for (size_t i = 0; i < n; ++i) x[i] = x[i] + hdt * (b1[i] + b2[i]);
        } else {
            for (size_t i = 0; i < n; ++i) x[i] = x[i] + dt * b1[i];
            if (rd->scheme == TI_SCHEME_EM && rd->eps > 0.0f) {
                const float sig = sqrtf(2.0f * rd->eps * fabsf(dt));
                for (int64_t m = 0; m < B; ++m) {
                    float com[3] = { 0, 0, 0 };
                    for (int c = 0; c < A * 3; ++c) { xt[c] = ti_normal(rd->seed, rd->traj_offset + m, k, c); com[c % 3] += xt[c]; }
                    for (int c = 0; c < A * 3; ++c) {
                        float z = xt[c]; if (rd->com_free_noise) z -= com[c % 3] / (float)A;
                        x[(size_t)m * A * 3 + c] += sig * z;
                    }
                }
            }
        }
        const int step = k + 1;
        if (rd->save_every > 0 && (step % rd->save_every == 0 || step == rd->n_step - 1)) memcpy(out_path + (row++) * n, x, sizeof(float) * n);
    }
    if (rd->save_every <= 0) memcpy(out_path, x, sizeof(float) * n);
    if (n_fevals) *n_fevals = fe;
    free(x); free(b1); free(b2); free(xt);
    return 0;
}

/* ------------------------------------------------------------------------------------------------------ adw API */
typedef struct { adw_t_f32* m32; adw_t_f64* m64; } tio_adw;

void* tio_adw_create(const ti_adw_desc* d, const double* w, size_t n)
{
    tio_adw* h = calloc(1, sizeof(*h));
    h->m32 = adw_parse_f32(d, w, n); h->m64 = adw_parse_f64(d, w, n);
    if (!h->m32 || !h->m64) { free(h); return NULL; }
    return h;
}
void tio_adw_destroy(void* hv) { free(hv); }

int tio_adw_drift_f64(void* hv, const double* x, double t, const double* b0, const double* b1, int64_t B, double* out)
{ adw_drift_f64(((tio_adw*)hv)->m64, x, t, b0, b1, B, out, NULL); return 0; }

/* drift and d b / d x (the reference's divergence before its 1e-2 scaling) */
int tio_adw_drift_div_f64(void* hv, const double* x, double t, const double* b0, const double* b1, int64_t B, double* out, double* div)
{ adw_drift_f64(((tio_adw*)hv)->m64, x, t, b0, b1, B, out, div); return 0; }

int tio_adw_drift_f32(void* hv, const float* x, float t, const float* b0, const float* b1, int64_t B, float* out)
{ adw_drift_f32(((tio_adw*)hv)->m32, x, t, b0, b1, B, out, NULL); return 0; }

/* fp64 rollout (the reference's adw precision).  out_path [rows,B] */
/* out_dlogp (may be NULL): the reference's second state, d(dlogp)/dt = -div * 1e-2, returned * 1e2 (integrators.py:68) */
int tio_adw_rollout_f64(void* hv, const ti_rollout_desc* rd, const double* x0, const double* b0, const double* b1v, int64_t B,
                        double* out_path, double* out_dlogp, int64_t* n_fevals)
{
    tio_adw* h = hv; const size_t n = (size_t)B;
    double *x = malloc(8 * n), *k1 = malloc(8 * n), *k2 = malloc(8 * n), *xt = malloc(8 * n);
    double *dl = calloc(n, 8), *d1 = malloc(8 * n), *d2 = malloc(8 * n);
    memcpy(x, x0, 8 * n);
    int64_t row = 0, fe = 0;
#define SAVE_ROW() do { if (out_dlogp) for (size_t i = 0; i < n; ++i) out_dlogp[row * n + i] = dl[i] * 1e2; memcpy(out_path + (row++) * n, x, 8 * n); } while (0)
    if (rd->save_every > 0) SAVE_ROW();
    for (int k = 0; k < rd->n_step - 1; ++k) {
        const double dt = (double)rd->t_grid[k + 1] - (double)rd->t_grid[k];
        adw_drift_f64(h->m64, x, (double)rd->t_grid[k], b0, b1v, B, k1, out_dlogp ? d1 : NULL); ++fe;
        if (rd->scheme == TI_SCHEME_HEUN) {
            for (size_t i = 0; i < n; ++i) xt[i] = x[i] + dt * k1[i];
            adw_drift_f64(h->m64, xt, (double)rd->t_grid[k + 1], b0, b1v, B, k2, out_dlogp ? d2 : NULL); ++fe;
            for (size_t i = 0; i < n; ++i) x[i] = x[i] + 0.5 * dt * (k1[i] + k2[i]);
            if (out_dlogp) for (size_t i = 0; i < n; ++i) dl[i] = dl[i] + 0.5 * dt * (-(d1[i] + d2[i]) * 1e-2);
        } else {
            for (size_t i = 0; i < n; ++i) x[i] = x[i] + dt * k1[i];
            if (out_dlogp) for (size_t i = 0; i < n; ++i) dl[i] = dl[i] + dt * (-d1[i] * 1e-2);
            if (rd->scheme == TI_SCHEME_EM && rd->eps > 0.0f) {
                const float sig = sqrtf(2.0f * rd->eps * fabsf((float)dt));
                for (size_t i = 0; i < n; ++i) x[i] += (double)(sig * ti_normal(rd->seed, rd->traj_offset + (int64_t)i, k, 0));
            }
        }
        const int step = k + 1;
        if (rd->save_every > 0 && (step % rd->save_every == 0 || step == rd->n_step - 1)) SAVE_ROW();
    }
    if (rd->save_every <= 0) SAVE_ROW();
#undef SAVE_ROW
    if (n_fevals) *n_fevals = fe;
    free(x); free(k1); free(k2); free(xt); free(dl); free(d1); free(d2);
    return 0;
}

int tio_num_threads(void)
{
#ifdef _OPENMP
    extern int omp_get_max_threads(void);
    return omp_get_max_threads();
#else
    return 1;
#endif
}
