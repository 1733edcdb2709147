/*
 * ti_hip.h -- C ABI of libti_hip.so, the MI355X (gfx950) sampler hot path for thermodynamic-interpolation.
 *
 * The reference (olsson-group/thermodynamic-interpolation) is pure Python and has no FFI seam; the seam this
 * library sits under is the reference's Python API for the sampling path.  Each entry point names the
 * reference interface it replaces (paths relative to the reference checkout):
 *
 *   ti_adw_create / ti_adw_drift      FCNetMultiBeta.__init__/forward      adw/thermo/models/simple.py:11-41
 *                                     + ODEWrapper.forward                 adw/thermo/models/ode_wrapper.py:30-52
 *   ti_adw_rollout                    StandardIntegrator.rollout           adw/thermo/integrators.py:33-68
 *   ti_painn_create / ti_painn_drift  cPaiNN.__init__/forward              mdqm9/thermo/ambient/models/cpainn.py:23-115
 *                                                                          mdqm9/thermo/latent/models/cpainn.py:23-108
 *                                     + ODEWrapper.forward/reset_batch     mdqm9/thermo/{ambient,latent}/models/ode_wrapper.py
 *   ti_painn_rollout                  MoleculeIntegrator.rollout           mdqm9/thermo/ambient/integrators.py:28-68
 *                                                                          mdqm9/thermo/latent/integrators.py:41-89
 *   ti_painn_drift_div / _jvp         ODEWrapper.compute_divergence        mdqm9/thermo/{ambient,latent}/models/ode_wrapper.py:59-91
 *   ti_painn_rollout_dlogp            MoleculeIntegrator.rollout(return_dlogp=True), ODEWrapper.forward (b, -div)
 *
 * Conventions
 *   - Plain pointers and sizes only; no exceptions cross the ABI.  Every int-returning call returns TI_OK (0) or a
 *     negative TI_E* code; ti_last_error() returns a thread-local message for the last failure on this thread.
 *   - The caller owns every buffer it passes.  Weights/graph templates are copied at create(); the handle owns all
 *     device memory and is freed only by ti_destroy().
 *   - Buffers marked [host|device] are interpreted according to ti_rollout_desc.mem / the `mem` argument:
 *     TI_MEM_HOST = ordinary host memory (the library stages through HBM), TI_MEM_DEVICE = pointers into HBM of
 *     the handle's device (e.g. torch.Tensor.data_ptr()); device work is enqueued on the handle's stream and the
 *     call returns after that stream has been synchronised.
 *   - There is no CPU fallback: if no gfx950 device is usable, create() fails with TI_E_HIP.
 *   - Environment (read per call): TI_TEMPLATE=throughput|latency|pair pins the edge-row layout that is otherwise chosen from
 *     the batch size (results agree to fp32 round-off; bit-identical within one layout); TI_JVP_WS_GB = HBM budget in GB for the
 *     tangent state of the divergence (default 48).
 *
 * Weight layout ("canonical flat layout", fp32 unless noted; every tensor row-major in torch's [out, in] order)
 *   MLP(f_in, f_h, f_out) := W0[f_h,f_in] b0[f_h] g0[f_h] be0[f_h]  W1[f_h,f_h] b1[f_h] g1[f_h] be1[f_h]  W2[f_out,f_h] b2[f_out]
 *                            (Linear, LayerNorm(gamma g, beta be, eps 1e-5), SiLU, Linear, LayerNorm, SiLU, Linear;
 *                             mdqm9/thermo/ambient/models/embedding.py:27-35)
 *   painn :  edge_emb[4,F]  atom_emb[n_types,F]  MLP(nE*F, F, F)
 *            L x { phi = MLP(2F,F,5F)  w = MLP(F,F,5F)  U[F,F]  V[F,F]  upd = MLP(2F,F,3F) }
 *            readout MLP(F,F,2)  Vr[1,F]
 *            nE = 4 (ambient: atom|T0|T1|t), 3 (latent multi-T: atom|T|t), 2 (latent single-T: atom|t)
 *   adw   :  beta_embed: W[H,3] b[H] W[H,H] b[H] W[1,H] b[1] ;  net: W[H,3] b[H] (W[H,H] b[H]) x (num_layers-1) W[1,H] b[1]
 *            passed as fp64 (the reference trains/saves in float64, adw/train.py:29); the device computes in fp32.
 */
#ifndef TI_HIP_H
#define TI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TI_ABI_VERSION 5

enum { TI_OK = 0, TI_E_ARG = -1, TI_E_HIP = -2, TI_E_NAN = -3, TI_E_ALLOC = -4, TI_E_UNSUPPORTED = -5 };
enum { TI_MEM_HOST = 0, TI_MEM_DEVICE = 1 };
enum { TI_VARIANT_AMBIENT = 0, TI_VARIANT_LATENT_MULTI = 1, TI_VARIANT_LATENT_SINGLE = 2 };
enum { TI_PREC_F32 = 0, TI_PREC_F16X2 = 1, TI_PREC_F16 = 2 };
/* Fixed-step schemes on a caller-supplied grid t[0..n_step-1] (the reference passes torch.linspace(start,end,n_step),
 * integrators.py:43; reversed grid for reverse_ode).  Build-defined (SURVEY.md F3 / §8a row I-new):
 *   EULER: x_{k+1} = x_k + dt_k b(x_k,t_k)                  (== torchdiffeq method='euler' on that grid)
 *   HEUN : xp = x_k + dt_k b(x_k,t_k); x_{k+1} = x_k + dt_k/2 (b(x_k,t_k) + b(xp,t_{k+1}))
 *   EM   : x_{k+1} = x_k + dt_k b(x_k,t_k) + sqrt(2 eps |dt_k|) xi,  xi ~ N(0,1) from Philox4x32-10 keyed by
 *          (seed, global trajectory id, step, component); eps = 0 reproduces EULER bit-for-bit. */
enum { TI_SCHEME_EULER = 0, TI_SCHEME_HEUN = 1, TI_SCHEME_EM = 2,
       /* torchdiffeq 0.2.5 solvers (the reference's integrator library, ti_env.yml:14; third-party, restated from its published
        * algorithm -- parity unpinned):
        *   DOPRI5   : adaptive Dormand-Prince 5(4) with FSAL, step control err = rms((y1_err)/(atol + rtol max(|y0|,|y1|))) <= 1
        *              (two-state runs: max of the per-state rms), factor = min(10, max(0.9 err^(-1/5), 0.2 | 1)), initial step
        *              by Hairer's rule, quartic dense output evaluated at the grid times (the grid only selects output times);
        *   MIDPOINT : fixed grid, y += dt f(t + dt/2, y + dt/2 f(t, y));
        *   RK4      : fixed grid, the 3/8-rule (torchdiffeq's `rk4`). */
       TI_SCHEME_DOPRI5 = 3, TI_SCHEME_MIDPOINT = 4, TI_SCHEME_RK4 = 5 };

typedef struct ti_handle ti_handle;

typedef struct ti_painn_desc {
    int32_t variant;        /* TI_VARIANT_* */
    int32_t n_features;     /* F: multiple of 32, <= 256 */
    int32_t n_layers;       /* L = score_layers */
    int32_t n_types;        /* rows of the atom embedding (reference: 25) */
    int32_t n_atoms;        /* A atoms per molecule (every molecule of a batch shares one species) */
    int32_t n_edges;        /* E_m directed edges per molecule */
    float   temp_length;    /* PositionalEncoder max_length for temperatures */
    float   time_length;    /* ... for t (reference: 10) */
    float   length_scale;   /* ... for edge distances (reference: 10) */
    float   temp_mean;      /* mean(temperatures)            (embedding.py:209) */
    float   temp_range;     /* max(temperatures) - min(...)  (embedding.py:210) */
    int32_t precision;      /* TI_PREC_F32: f32 MFMA (default); TI_PREC_F16X2: the matrix products on the fp16 matrix rate with
                               every fp32 operand split into two fp16 halves (products hi*hi, hi*lo, lo*hi; fp32 accumulation;
                               ~24 significand bits; un-normalised operand rows are scaled by a power of two first, so any fp32
                               magnitude works; weights must be < 65504).  Two operand formats are in use:
                                 (a) "two accumulators": x = hi + 2^-11 lo with lo = fp16(2^11 (x - hi)); the cross terms have an
                                     accumulator of their own that is folded back with 2^-11.  Used by the update / embed / readout /
                                     tangent kernels at every width and by the message kernels at F = 256;
                                 (b) "one accumulator" (message kernels, F <= 128): each weight matrix is scaled on the host by a
                                     power of two S to the top of the fp16 range, lo = fp16(S w - hi) and lo = fp16(x - hi) are NOT
                                     scaled, and all three products add into one register set (S cancels in the LayerNorm behind
                                     a hidden layer and is divided out of the output products).  It relies on
                                     v_mfma_f32_16x16x32_f16 taking fp16-subnormal inputs at face value, which ti_selftest checks;
                               TI_PREC_F16: fp16 STORAGE mode (BASELINE.json configs[4]): the state tensors s, v, P, e live in HBM
                               as fp16 and every matrix product is one fp16 MFMA with fp32 accumulation; LayerNorm, SiLU,
                               sin/cos, per-atom sums and the integrator state x stay fp32.  A separately labelled precision:
                               drift rel-L2 ~1e-3 against the reference (not the 1e-5 of the other two); values must stay
                               inside the fp16 range; no divergence / dlogp / debug taps in this mode (TI_E_UNSUPPORTED) */
} ti_painn_desc;

typedef struct ti_adw_desc {
    int32_t hidden_size;    /* H: multiple of 32, <= 256 */
    int32_t num_layers;     /* number of hidden layers of `net` (reference: 5) */
    int32_t precision;      /* TI_PREC_F32 | TI_PREC_F16X2 (as ti_painn_desc.precision) */
} ti_adw_desc;

typedef struct ti_rollout_desc {
    int32_t scheme;         /* TI_SCHEME_* */
    int32_t n_step;         /* number of grid points; n_step-1 steps are taken */
    int32_t save_every;     /* k>=1: rows 0,k,2k,... of the path plus the final state are written; 0: final state only */
    int32_t mem;            /* TI_MEM_* for x0 / cond / out_path */
    float   eps;            /* EM noise scale (>= 0) */
    int32_t com_free_noise; /* EM, molecules: remove the per-molecule centre of mass of xi */
    uint64_t seed;          /* EM Philox key */
    int64_t traj_offset;    /* global index of trajectory 0 of this call (multi-GPU shards keep RNG independent of the split) */
    const float* t_grid;    /* [n_step] host memory */
    float   rtol, atol;     /* DOPRI5 tolerances (> 0); ignored by the fixed-grid schemes */
    int64_t step_offset;    /* EM: index of this call's first step in the noise counter (step k of the call draws with counter
                               step_offset + k), so that a trajectory continued by a second call does not reuse the first call's
                               noise; 0 for a rollout that starts at the beginning */
} ti_rollout_desc;

/* number of path rows ti_*_rollout writes for (n_step, save_every) */
int64_t ti_rollout_rows(int32_t n_step, int32_t save_every);

int ti_version(void);
int ti_device_count(void);
const char* ti_last_error(void);

/* ---- adw: 1-D asymmetric double well ------------------------------------------------------------------------ */
ti_handle* ti_adw_create(const ti_adw_desc* desc, const double* weights, size_t n_weights, int device);
/* b[i] = net([x_i, t, beta_embed([beta0_i, beta1_i, t])]);  x,beta0,beta1,out: [B] fp32 [host|device] */
int ti_adw_drift(ti_handle* h, const float* x, float t, const float* beta0, const float* beta1, int64_t B, float* out, int mem);
/* also out_div[i] = d b_i / d x_i, the exact divergence of the 1-D drift by forward-mode differentiation of `net`
 * (ODEWrapper.compute_divergence, adw/thermo/models/ode_wrapper.py:55-67, without its 1e-2 factor) */
int ti_adw_drift_div(ti_handle* h, const float* x, float t, const float* beta0, const float* beta1, int64_t B, float* out,
                     float* out_div, int mem);
/* out_path: [rows, B] fp32 with rows = ti_rollout_rows(...) */
int ti_adw_rollout(ti_handle* h, const ti_rollout_desc* desc, const float* x0, const float* beta0, const float* beta1,
                   int64_t B, float* out_path, int64_t* n_fevals);
/* StandardIntegrator(return_dlogp=True) (adw/thermo/integrators.py:38-68): integrates the second state
 * d(dlogp)/dt = -div * 1e-2 with the same scheme (any but EM with eps > 0) and writes out_dlogp [rows, B] = dlogp * 1e2 */
int ti_adw_rollout_dlogp(ti_handle* h, const ti_rollout_desc* desc, const float* x0, const float* beta0, const float* beta1,
                         int64_t B, float* out_path, float* out_dlogp, int64_t* n_fevals);

/* ---- mdqm9: cPaiNN drift over homogeneous molecule batches ------------------------------------------------------ */
/* edge_src/edge_dst: [E_m] local atom indices of ONE molecule in the reference's (src,dst)-sorted order
 * (edge_index[0]=src, edge_index[1]=dst: messages flow src -> dst, cpainn.py:273-304); edge_type: [E_m] in 0..3;
 * atom_ids: [A] rows of the atom embedding (reference: arange(A), mdqm9_ambient.py:219-220). */
ti_handle* ti_painn_create(const ti_painn_desc* desc, const float* weights, size_t n_weights,
                           const int32_t* edge_src, const int32_t* edge_dst, const int32_t* edge_type,
                           const int32_t* atom_ids, int device);
/* x: [B,A,3]; cond: [B,A,n_cond] per-node conditioning (ambient: T0,T1; latent multi-T: T; single-T: NULL); out: [B,A,3] */
int ti_painn_drift(ti_handle* h, const float* x, float t, const float* cond, int64_t B, float* out, int mem);
/* out_path: [rows, B, A, 3]; *n_fevals = drift evaluations taken (DOPRI5: 2 + 6 per attempted step) */
int ti_painn_rollout(ti_handle* h, const ti_rollout_desc* desc, const float* x0, const float* cond, int64_t B,
                     float* out_path, int64_t* n_fevals);

/* Forward-mode derivative of the drift along xdot [B,A,3]: out = b(x), out_tan = (d b / d x) xdot.  The building block
 * of the divergence below; also what the parity tests tap stage by stage. */
int ti_painn_drift_jvp(ti_handle* h, const float* x, const float* xdot, float t, const float* cond, int64_t B, float* out,
                       float* out_tan, int mem);
/* Exact divergence out_div[b] = sum_{a,c} d b[b,a,c] / d x[b,a,c] by 3A unit-seed forward-mode passes per molecule --
 * what ODEWrapper.compute_divergence obtains with 3A reverse-mode passes (mdqm9/thermo/ambient/models/ode_wrapper.py:59-91,
 * latent/models/ode_wrapper.py:57-86), WITHOUT the ambient wrapper's 1e-2 factor.  Tangent state is processed in chunks of
 * molecules sized to TI_JVP_WS_GB gigabytes of HBM (environment, default 48). */
int ti_painn_drift_div(ti_handle* h, const float* x, float t, const float* cond, int64_t B, float* out, float* out_div, int mem);
/* MoleculeIntegrator.rollout(return_dlogp=True) (ambient/integrators.py:36-68, latent/integrators.py:57-89) on the fixed
 * grid of `desc` (EULER, HEUN, MIDPOINT, RK4 or the adaptive DOPRI5; EM is refused): second state d(dlogp)/dt = -div_scale * div, or with reverse_ode the pair
 * (-b, +div_scale * div) (ode_wrapper.py:49; the caller passes the descending grid linspace(end, start)).
 * out_dlogp [rows, B] = state * out_scale.  Reference values: ambient div_scale 1e-2, out_scale 1e2; latent 1, 1. */
int ti_painn_rollout_dlogp(ti_handle* h, const ti_rollout_desc* desc, const float* x0, const float* cond, int64_t B,
                           float div_scale, float out_scale, int reverse_ode, float* out_path, float* out_dlogp, int64_t* n_fevals);

/* ---- shared ------------------------------------------------------------------------------------------------------ */
void ti_destroy(ti_handle* h);
/* Streams.  A handle enqueues all device work on ONE stream: its own (created non-blocking at create()) or, with
 * ti_set_stream(h, s, TI_STREAM_EXTERNAL), the caller's hipStream_t `s` -- where s == NULL then means the legacy null stream
 * (torch's default stream reports cuda_stream == 0).  TI_STREAM_OWN restores the handle's own stream (`s` is ignored).
 * TI_MEM_DEVICE inputs written by work on ANOTHER stream (a torch kernel that produced x0 / cond) are ordered with
 * ti_wait_stream(h, producer): the handle's stream then waits for everything enqueued on `producer` so far (NULL = the null
 * stream).  Every call returns after synchronising the handle's stream, so outputs need no further ordering. */
enum { TI_STREAM_OWN = 0, TI_STREAM_EXTERNAL = 1 };
int ti_set_stream(ti_handle* h, void* hip_stream, int mode);
int ti_wait_stream(ti_handle* h, void* producer_stream);
/* Pin the edge-row layout of a painn handle: TI_TEMPLATE_AUTO (from the batch size of each call, the default),
 * TI_TEMPLATE_THROUGHPUT, TI_TEMPLATE_LATENCY (directed edge rows sorted by destination) or TI_TEMPLATE_PAIR (pair-major rows:
 * the filter branch w(enc(|r_ij|)) of SE3Message, cpainn.py:283-289, is evaluated once per atom pair and shared by the edges i->j
 * and j->i; needs a symmetric graph, F <= 128 and TI_PREC_F32 / TI_PREC_F16X2, otherwise the request falls back to
 * TI_TEMPLATE_THROUGHPUT; the divergence / tangent entry points always use a directed layout).  Results are bit-identical
 * within one layout and agree to fp32 round-off across them, so a run sharded over ranks pins the layout it would use for the
 * GLOBAL batch and becomes independent of the rank count.  (The TI_TEMPLATE environment variable, read per call, overrides this.) */
enum { TI_TEMPLATE_AUTO = -1, TI_TEMPLATE_THROUGHPUT = 0, TI_TEMPLATE_LATENCY = 1, TI_TEMPLATE_PAIR = 2 };
int ti_painn_set_template(ti_handle* h, int which);
/* the layout ti_painn_drift / ti_painn_rollout would choose for a batch of B molecules (TI_TEMPLATE_THROUGHPUT / _LATENCY / _PAIR) */
int ti_painn_template_for(ti_handle* h, int64_t B);
/* Pre-size the HBM workspace for batches up to B trajectories (otherwise grown on demand). */
int ti_reserve(ti_handle* h, int64_t B);
/* Live kernel timing with HIP events on the handle's stream (bench.py roofline leg). */
enum { TI_KERNEL_PAINN_EDGE = 0, TI_KERNEL_PAINN_UPDATE = 1, TI_KERNEL_PAINN_EMBED = 2, TI_KERNEL_PAINN_READOUT = 3,
       TI_KERNEL_ADW = 4, TI_KERNEL_INTEGRATE = 5, TI_KERNEL_PAINN_JVP_EDGE = 6, TI_KERNEL_PAINN_JVP_UPDATE = 7,
       TI_KERNEL_PAINN_JVP_READOUT = 8, TI_KERNEL_PAINN_JVP_FILTER = 9, TI_KERNEL_COUNT = 10 };
int ti_profile_enable(ti_handle* h, int on);
int ti_profile_read(ti_handle* h, int kernel, int64_t* n_launches, double* total_ms);   /* also resets that slot */
/* Debug taps for parity tests: copy an intermediate of the LAST ti_painn_drift / ti_painn_drift_jvp call to host.
 * what: 0 = s [B,A,F], 1 = v [B,A,3,F] (component-major planes), 2 = e (row-major [B,E_m,F], edges in (dst,src) order);
 * 3, 4, 5 = the tangents of s, v, e of the last ti_painn_drift_jvp call, same shapes. */
int ti_painn_debug_tap(ti_handle* h, int stop_after_stage);   /* stage = 0 embed, 1+2l message l, 2+2l update l; -1 = off */
int ti_painn_debug_read(ti_handle* h, int what, float* out, size_t n_floats);
/* Test hook for the first-touch accumulators (csrc/painn_edge_kernel.hpp: acc_out): fill the per-atom accumulators dsacc / dvacc / cacc of
 * the workspace for B trajectories with `value` (NaN, 1e30 ...) on the handle's stream.  A following ti_painn_drift must return exactly what
 * a handle created with TI_ZERO_ACC=1 in the environment (zeroing path: memsets + adds only) returns.  Reference counterpart: none
 * (torch_scatter allocates its output, cpainn.py:303-304). */
int ti_painn_debug_poison(ti_handle* h, int64_t B, float value);
/* Device self-test of the MFMA operand/accumulator lane maps the kernels rely on, of both fp32 -> (hi, lo) fp16 operand splits
 * (the 8-instruction forms of formats (a) and (b) against the plain arithmetic, bit for bit, fp16-subnormal residuals included),
 * and of the fp16 matrix instruction keeping subnormal inputs. */
int ti_selftest(int device);

#ifdef __cplusplus
}
#endif
#endif /* TI_HIP_H */
