// ti_internal.hpp -- host/device shared declarations of libti_hip.so (not part of the public ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/ti_hip.h"

namespace ti {

// ---- molecule-group edge template -------------------------------------------------------------------------------
// Two templates are built per handle (ti_api.hip: build_templates): "throughput" (G molecules per group, P = 1: fewest padded
// rows, one wave walks G*E_m rows) and "latency" (G = 1, the destination atoms of a molecule cut into P parts, each part padded
// to whole row blocks and walked by its own wave: P times the waves for small batches).  A kernel's group index counts
// (molecule group, part):  gi = mg * parts + part;  rows / slotnode hold [parts][nblk*16] entries.
// slotnode word: -1 = no such slot, else  atom | molecule-in-group << 8 | SLOT_FIRST_TOUCH: this block is the first one (in the owning
// wave's program order) that holds rows of the atom -- its sums REPLACE the accumulator contents instead of adding to them, so nobody
// has to zero the accumulators between layers.  Set only when every atom of the graph has incoming edges (ti_api.hip).
constexpr int32_t SLOT_FIRST_TOUCH = 1 << 30;
__host__ __device__ inline int slot_mol(int32_t sn) { return (sn >> 8) & 0x3fffff; }
// The E_m edges of one molecule are sorted by (dst, src); G molecules form a "group" whose G*E_m edge rows are padded
// to NBLK blocks of EDGE_ROWS_PER_BLOCK rows.  One wave owns one group, so every per-atom sum over incoming edges stays inside a wave
// in that wave's program order (deterministic).  Within a block, the distinct (molecule, dst atom) pairs are numbered as "slots";
// a block holds at most EDGE_MAX_SLOTS of them (a fifth destination atom starts the next block, the rest of the block is
// padding), so the per-slot sums of a block fit the four lane rows of a wave (mfma_chain.hpp: r16::QuarterSum).
//   row word : bit0 valid | mol_local<<1 (5b) | src<<6 (5b) | dst<<11 (5b) | etype<<16 (2b) | slot<<18 (6b, 63 = none)
//   slot word: mol_local<<8 | atom     (-1 = unused)
constexpr int ROW_VALID = 1;
constexpr int EDGE_ROWS_PER_BLOCK = 16;     // painn_edge_kernel walks a group in 16-row blocks (16x16x4 MFMA)
constexpr int EDGE_MAX_SLOTS = 4;
__host__ __device__ inline int row_mol(uint32_t w) { return (w >> 1) & 31; }
__host__ __device__ inline int row_src(uint32_t w) { return (w >> 6) & 31; }
__host__ __device__ inline int row_dst(uint32_t w) { return (w >> 11) & 31; }
__host__ __device__ inline int row_type(uint32_t w) { return (w >> 16) & 3; }
__host__ __device__ inline int row_slot(uint32_t w) { return (w >> 18) & 63; }

// ---- pair-major template (painn_pair_kernel.hpp; ti_api.hip: build_pair_template).  The filter branch w(enc(|r_ij|)) of SE3Message
// (cpainn.py:283-289) depends on the edge length only, so the edges i->j and j->i share it bit for bit.  A row block holds 16 atom PAIRS
// laid out as a 4 x 4 tile: row 4a + b = pair (I[a], J[b]) of up to four "I" atoms and four "J" atoms (slots carry molecule-in-group and
// atom; rows of pairs that do not exist are invalid).  Direction A is the edge I[a] -> J[b] (src I, dst J), direction B the edge
// J[b] -> I[a].  The per-atom sums need no masks: direction B's destination is the same for the four rows a lane holds (in-lane adds),
// direction A's is the same across the four lane rows (two lane-swap levels).
//   row word : bit0 valid | molI<<1 (3b) | atomI<<4 (5b) | molJ<<9 (3b) | atomJ<<12 (5b) | etype<<17 (2b)     (empty slots: a safe atom)
//   slot word: slotnode[blk*16 + k], k = 0..3 the I slots, k = 4..7 the J slots: -1, or atom | mol<<8
// No atomics: slot k of block b writes the sum over its rows of (ds | dv | c) into partial row b*8 + k of its group; a reduction
// (pair_reduce_kernel) adds, per atom, the partial rows the template lists for it (plist), in walk order.
// e rows: [group][block][direction][16][F]; the parked encoding / edge_dir (direction A's) once per pair: [group][block][16].
constexpr int PAIR_MAX_G = 8;
__host__ __device__ inline int prow_molI(uint32_t w) { return (w >> 1) & 7; }
__host__ __device__ inline int prow_atomI(uint32_t w) { return (w >> 4) & 31; }
__host__ __device__ inline int prow_molJ(uint32_t w) { return (w >> 9) & 7; }
__host__ __device__ inline int prow_atomJ(uint32_t w) { return (w >> 12) & 31; }
__host__ __device__ inline int prow_type(uint32_t w) { return (w >> 17) & 3; }

struct MlpVec {           // natural-order per-feature vectors of one reference MLP block (device pointers)
    const float *b0, *g0, *be0, *b1, *g1, *be1, *b2;
};

struct EdgeParams {
    const float4* stream; int nch;         // packed weight chunks of this layer's message block
    const float* vecs;                      // [21][F] bias/gamma/beta of the w and phi MLPs (painn_kernels.hip: struct EV)
    const float* edge_emb;                  // [4][F]  (first layer: e = edge_emb[type])
    const uint32_t* rows; const int32_t* slotnode; const int32_t* nslots;
    int nblk, G, parts, A;                  // parts per group (see above); n_groups counts parts
    int max_slots;                          // most destination atoms any row block of the template holds (<= 4)
    long long B, n_groups;
    float length_scale;
    const float* x;                         // [B*A][3]
    const float* P;                         // [B*A][F]   s @ W0[:, :F]^T + b0
    const float* v;                         // [B*A][3][F]
    float* dsacc;                           // [B*A][F]   += sum ds   (added to s by the update kernel)
    float* dvacc;                           // [B*A][3][F] += sum (sed*dir + gates*v[src])
    float* cacc;                            // [B*A][3][F] += sum cg*dir   (crossed with v[dst] in the update kernel)
    float* e;                               // [n_groups*nblk*16][F]
    float* enc;                             // [n_groups*nblk][operand registers][64] parked encoding operand of every row block (layer 0 writes, the others read)
    float wscale[6];                        // TI_PREC_F16X2: powers of two the host scaled w.W0, w.W1, phi.W0(e), phi.W1, phi.W2, w.W2 by (else 1)
    float* geo;                             // [n_groups*nblk*16][4] parked edge_dir
    float* part;                            // pair-major kernel: [n_groups*nblk][8 slots][7][F] partial sums (ds | dv x3 | c x3) of every slot
    unsigned long long* stamps;             // diagnostic builds (-DTI_STAMPS) only: s_memtime / s_memrealtime stamps, a buffer of their own; else NULL
};

// pair-major kernel: sum of each atom's partial rows (walk order) -> dsacc / dvacc / cacc, which the update kernel then reads
struct PairReduceParams {
    const float* part; const int32_t* plist;        // plist [G*A][kmax]: partial rows (block * 8 + slot) of every atom of a group, -1 = end
    int kmax, G, A, F, nblk, has_c;                 // has_c = 0: the first layer writes no cross-gate sums
    long long B;
    float *dsacc, *dvacc, *cacc;
};
hipError_t launch_pair_reduce(const PairReduceParams& p, hipStream_t st);

struct EmbedParams {
    const float4* stream; int nch;
    MlpVec mlp; const float* pb0;           // pb0 = phi[0].b0
    const float* atom_emb; const int32_t* atom_ids; const float* cond;
    int ncond, A; long long N;
    float t, temp_length, time_length, temp_mean, temp_range;
    float* s; float* P;
};

struct UpdateParams {
    const float4* stream; int nch;
    const float* vecs;                      // [10][F] (painn_kernels.hip: struct UV)
    long long N;
    float *s, *v, *dsacc, *dvacc, *cacc, *P;
    int first_layer;                        // v and cacc are zero by definition (nothing has written them yet): not read
    int zero_acc;                           // reset the accumulators after use (0 when the edge kernels overwrite on first touch)
};

struct ReadoutParams {
    const float4* stream; int nch;
    MlpVec mlp; const float* w2_gate; float b2_gate; const float* Vr;
    long long N;
    const float *s, *v; float* out;
};

// launchers (painn_kernels.hip).  F = 32*NB; return hipError_t of the launch.
// prec = TI_PREC_* of include/ti_hip.h; with TI_PREC_F16 the state tensors s, P, v, e behind the float* fields are fp16
hipError_t launch_embed(int NB, int nseg, int prec, const EmbedParams& p, hipStream_t st);
hipError_t launch_edge(int NB, bool first, bool last, int prec, const EdgeParams& p, hipStream_t st);
bool edge_uses_one_chain(int NB, int prec);      // message kernel on the one-accumulator split format (painn_edge_kernel.hpp: edge_one_chain)
// pair-major message kernel (painn_pair_kernel.hpp): same EdgeParams, rows / slotnode of the pair template, same weight stream
hipError_t launch_pair(int NB, bool first, bool last, int prec, const EdgeParams& p, hipStream_t st);
bool pair_kernel_exists(int NB, int prec);
bool pair_uses_partials();          // the pair kernel writes per-(block, slot) partial sums that launch_pair_reduce adds up (else: atomics, first touch)
hipError_t launch_update(int NB, bool has_next, int prec, const UpdateParams& p, hipStream_t st);
hipError_t launch_readout(int NB, int prec, const ReadoutParams& p, hipStream_t st);
hipError_t configure_painn_kernels(int NB);     // dynamic-LDS attributes

// ---- forward-mode derivative of the drift (painn_jvp_kernels.hip; virtual-molecule layout described there).
// D = 3A and xdot == NULL: unit seeds (direction d -> atom d/3, component d%3); D = 1 with xdot [B*A][3]: that direction.
// Tangent arrays are laid out like their primal twins over ceil(B/G)*D*G virtual molecules; primal arrays are read only.
struct JvpFilterParams {                    // primal pass of one layer's message block (painn_jvp_filter_kernel)
    const float4* stream; int nch; const float* vecs; const float* edge_emb;      // the primal edge stream / vector block
    const uint32_t* rows;
    int nblk, G, parts, A, first, last;
    long long B, n_groups;                  // molecules, primal groups (incl. parts)
    float length_scale;
    const float *x, *P, *e;
    float4* wq;                             // [n_groups*nblk][5][NB][6][64] float4: phi_o, w_o, d w_o / d|r|
    float4* st;                             // [n_groups*nblk][4][NBK][64]   float4: LayerNorm statistics of phi
};
struct JvpEdgeParams {
    const float4* stream; int nch, pad; const float* vecs; const float* edge_emb;
    const uint32_t* rows; const int32_t* slotnode;
    int nblk, G, parts, A, D, first, last;
    long long B, n_groups;                  // molecules, VIRTUAL groups (= molecule groups * D * P)
    const float *x, *xdot;
    const float *P, *v, *e;                 // primal state entering this layer's message block
    const float4 *wq, *st;                  // primal pass output of this layer
    const float *tP, *tv;                   // tangents of P and v
    float *te, *tdsacc, *tdvacc, *tcacc;    // tangent of e (updated in place), tangent accumulators (+=)
};
struct JvpNodeParams {                      // primal node pass of one layer's update block (painn_jvp_node_kernel)
    const float4* stream; int nch; const float* vecs;          // the tangent update stream (V and U once per component)
    long long N;                            // primal nodes
    const float *s, *v, *dsacc, *dvacc, *cacc;
    float4* ns;                             // [ceil(N/16)][13][NBK][64] float4
};
struct JvpUpdateParams {
    const float4* stream; int nch; const float* vecs;
    long long N, B; int A, D, G, has_next;  // N virtual nodes
    const float *v, *cacc;                  // primal, as the primal edge kernel left them
    const float4* ns;                       // primal node pass output of this layer
    float *ts, *tv, *tdsacc, *tdvacc, *tcacc, *tP;
    int zero_acc;                           // reset the tangent accumulators after use (0 with first-touch slot tables)
};
struct JvpReadoutParams {
    const float4* stream; int nch; const float* vecs; float b2_gate;      // vecs: b0 g0 be0 b1 g1 be1 w2_gate Vr (x F)
    long long N, B; int A, D, G;
    const float *s, *v, *ts, *tv;
    float* tout;                            // [virtual nodes][3]
};
hipError_t launch_jvp_filter(int NB, bool split, const JvpFilterParams& p, hipStream_t st);
hipError_t launch_jvp_edge(int NB, bool split, const JvpEdgeParams& p, hipStream_t st);
hipError_t launch_jvp_node(int NB, bool split, const JvpNodeParams& p, hipStream_t st);
hipError_t launch_jvp_update(int NB, bool split, const JvpUpdateParams& p, hipStream_t st);
hipError_t launch_jvp_readout(int NB, bool split, const JvpReadoutParams& p, hipStream_t st);
hipError_t launch_div_reduce(const float* tout, long long B, int D, int G, float* div, hipStream_t st);
hipError_t configure_painn_jvp_kernels(int NB);

// ---- adw (adw_kernels.hip).  One kernel evaluates  Linear(3->H), SiLU, [Linear(H->H), SiLU] x n_hidden, Linear(H->1)
// on rows (a0, a1, a2):  a0 = x[r];  a1 = in1 ? in1[r] : t;  a2 = idx ? emb[idx[r]] : emb ? emb[r] : t.
struct AdwParams {
    const float4* stream; int nch;          // hidden layers, 32-output chunks (16-row format), layer-major
    const float* vecs;                      // w_in [H][3] | b_in [H] | b_hidden [n_hidden][H] | w_out [H]
    float b_out;
    int n_hidden; long long B;
    const float* x; const float* in1; const float* emb; const int32_t* idx;
    float t;
    float* out;
    float* out_div;                         // non-NULL: also d out / d a0 (forward-mode tangent)
};
hipError_t launch_adw(int NB, bool split, const AdwParams& p, hipStream_t st);
hipError_t configure_adw_kernels(int NB, int max_hidden);

// ---- integrator kernels (integrate_kernels.hip)
hipError_t launch_axpy(float* y, const float* x, float a, const float* b, long long n, hipStream_t st);          // y = x + a*b
hipError_t launch_heun(float* x, float hdt, const float* b1, const float* b2, long long n, hipStream_t st);      // x += hdt*(b1+b2)
hipError_t launch_noise(float* x, float sigma, uint64_t seed, long long traj0, int step, long long B, int comps_per_traj,
                        int atoms_for_com /*0 = no COM removal*/, hipStream_t st);
hipError_t launch_scale(float* y, const float* x, float a, long long n, hipStream_t st);                            // y = a*x
hipError_t launch_selftest(float* out /*[64*16]*/, hipStream_t st);
hipError_t launch_split_selftest(unsigned* out /*[2], zeroed: differing halves, subnormal-product mismatches*/, hipStream_t st);
hipError_t launch_nan_check(const float* x, long long n, int* flag, hipStream_t st);

// ---- Runge-Kutta pieces (ode_kernels.hip)
struct RkComb { const float* k[7]; float c[7]; int nk; };     // sum_j c[j] * k[j][i], j < nk
constexpr int RED_PARTIALS = 1024;                             // size of the `partial` scratch (doubles) of the reductions below
hipError_t launch_rk_combo(float* y, const float* y0, const RkComb& c, long long n, hipStream_t st);               // y = y0 + comb
// *out = sum_i (comb_i / (atol + rtol max(|y0_i|, |y1_i|)))^2, fixed summation order
hipError_t launch_rk_ratio_sumsq(double* out, double* partial, const float* y0, const float* y1, const RkComb& c, float rtol, float atol,
                                 long long n, hipStream_t st);
// *out = sum_i ((a_i - b_i) / (atol + rtol |y0_i|))^2   (b may be NULL)
hipError_t launch_scaled_sumsq(double* out, double* partial, const float* a, const float* b, const float* y0, float rtol, float atol,
                               long long n, hipStream_t st);
hipError_t launch_interp_fit(float* coef /*[5][n]*/, const float* y0, const float* y1, const float* f0, const float* f1, const RkComb& mid,
                             float dt, long long n, hipStream_t st);
hipError_t launch_interp_eval(float* out, const float* coef, float x, long long n, hipStream_t st);

}  // namespace ti
