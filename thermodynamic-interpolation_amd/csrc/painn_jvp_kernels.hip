// painn_jvp_kernels.hip -- forward-mode derivative of the cPaiNN drift for gfx950 (MI355X): the exact divergence the
// reference obtains with 3A reverse-mode passes (ODEWrapper.compute_divergence,
// /root/reference/mdqm9/thermo/ambient/models/ode_wrapper.py:59-91; latent twin ode_wrapper.py:57-86).
//
// Virtual molecules.  Molecule b differentiated along seed direction d is a "virtual molecule".  D = 3A with unit seeds
// (d perturbs atom d / 3, component d % 3) gives the Jacobian diagonal, D = 1 with an explicit xdot an arbitrary JVP.
// They are grouped like the primal molecules: primal group pg (G molecules, one wave) and direction d form virtual group
// vg = pg * D + d, whose G members are the G molecules of pg -- so virtual row (blk, j) IS primal row (blk, j) of group pg
// (with P > 1 parts per group, ti_internal.hpp, every part of every direction has its own wave: gi = (pg * D + d) * P + part)
// (same edge, same slot), and the D directions of one primal group are neighbours in the launch (their primal reads hit L2).
//   virtual molecule vm = vg * G + m  <->  molecule pm = pg * G + m;   virtual node = vm * A + atom.
// The kernels carry ONLY tangents in HBM (ts, tv, te, tP and three accumulators, laid out like their primal twins over
// virtual molecules).  Primal state is read from the ordinary drift pipeline, which the host runs in lock step
// (ti_api.hip: [filter pass ->] tangent edge -> primal edge -> tangent update -> primal update per layer); the primal
// activations a tangent needs (LayerNorm statistics, SiLU slopes, gate values) are recomputed in registers next to it, every
// matrix product runs on a (value, tangent) operand pair against one weight chunk in LDS.
//
// The filter branch w(enc(|r|)) depends on x only through the edge length, so its tangent is rank one:
//   d w_o[edge][dir] = d|r|[edge][dir] * Q[edge],  Q = (d w_o / d |r|).
// painn_jvp_filter_kernel (the "primal pass") evaluates w_o and Q ONCE per primal edge and layer (a dual pass seeded with
// d|r| = 1), together with the phi branch's forward values and LayerNorm statistics, and parks them in HBM in the register
// layouts the per-direction edge kernel consumes; that kernel is left with the TANGENT of the phi branch only -- a quarter
// of the matrix work of differentiating both branches per direction, and no transcendental per (edge, direction).
//
// Tangent rules restated from the forward pass (painn_kernels.hip; reference lines there):
//   geometry  r = x_s - x_d, d = |r|, dir = r / (1 + d):   dd = r.dr / d,  ddir = dr / (1 + d) - r dd / (1 + d)^2
//   products  (h g)' = h' g + h g' ;  cross(k, v)' = cross(k', v) + cross(k, v')
//   norm      n = |V v|:  n' = (V v).(V v') / n   (0 at n = 0, like torch.norm's backward)
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include "mfma_chain.hpp"
#include "ti_internal.hpp"


namespace ti {

namespace {

struct EV {          // same vector block as painn_kernels.hip (struct EV)
    static constexpr int W_B0 = 0, W_G0 = 1, W_BE0 = 2, W_B1 = 3, W_G1 = 4, W_BE1 = 5, P_G0 = 6, P_BE0 = 7, P_B1 = 8, P_G1 = 9,
                         P_BE1 = 10, P_B2 = 11, W_B2 = 16, COUNT = 21;
};
struct UV {          // same vector block as painn_kernels.hip (struct UV)
    static constexpr int B0 = 0, G0 = 1, BE0 = 2, B1 = 3, G1 = 4, BE1 = 5, B2 = 6, PB0 = 9, COUNT = 10;
};
struct RV {          // readout vector block: b0 g0 be0 b1 g1 be1 w2_gate Vr
    static constexpr int B0 = 0, G0 = 1, BE0 = 2, B1 = 3, G1 = 4, BE1 = 5, W2G = 6, VR = 7, COUNT = 8;
};

__device__ __forceinline__ void add_noret(float* p, float v) { unsafeAtomicAdd(p, v); }
// accumulator update with the first-touch rule of the primal edge kernel (ti_internal.hpp SLOT_FIRST_TOUCH)
__device__ __forceinline__ void acc_out(float* p, float v, bool first)
{
    if (first) (void)__hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else unsafeAtomicAdd(p, v);
}

#define Z4 (f32x4{0.f, 0.f, 0.f, 0.f})

}  // namespace

// ================================================================================================== primal pass
// Everything the per-direction edge kernel needs from the forward pass of this layer's message block, ONCE per primal edge:
//   wq[((((pg * nblk + blk) * 5 + c) * NB + nbo) * 6 + k) * 64 + lane]  (float4 = the block's rows 4 (lane >> 4) .. + 3)
//        k = 0/1: phi_o + bias (features 32 nbo + {0,16} + (lane & 15)),  2/3: w_o + bias,  4/5: Q = d w_o / d|r|
//   st[(((pg * nblk + blk) * 4 + which) * NBK + nb) * 64 + lane]        which = n, kk of phi's two LayerNorms (ln_silu_stats),
//        float4 = features 16 nb + 4 (lane >> 4) .. + 3 of row (lane & 15)
// The filter branch runs as a (value, tangent) pair seeded with d|r| = 1 (-> Q), the phi branch as the plain forward pass.
// Reads the primal edge stream of painn_edge_kernel (same chunk order); one wave per primal group.
template <int NBK, bool SPLIT>
__global__ __launch_bounds__(256, 1) void painn_jvp_filter_kernel(const JvpFilterParams p)
{
    constexpr int F = 16 * NBK, NB = (F + 31) / 32, WAVES = 4, T = 64 * WAVES, CH4 = 256 * NB;
    using A16 = r16::Act<NBK>;
    using OP = r16::Opnd<NBK, SPLIT>;
    extern __shared__ f32x4 lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), j = lane & 15, q = lane >> 4;
    float* vec = reinterpret_cast<float*>(lds + 4 * CH4);                           // [EV::COUNT][F]
    for (int i = threadIdx.x; i < EV::COUNT * F / 4; i += T)
        reinterpret_cast<f32x4*>(vec)[i] = reinterpret_cast<const f32x4*>(p.vecs)[i];
    PipeDMA<NB, T, 2> pipe;
    pipe.init(reinterpret_cast<const f32x4*>(p.stream), p.nch, lds, wave, lane);

    const long long gi_raw = (long long)blockIdx.x * WAVES + wave;
    const bool group_ok = gi_raw < p.n_groups;
    const long long gi = group_ok ? gi_raw : p.n_groups - 1;
    const bool first = p.first != 0, last = p.last != 0;
    const long long mg = gi / p.parts;                                                  // molecule group; gi also counts its parts
    const uint32_t* rows = p.rows + (size_t)(gi - mg * p.parts) * p.nblk * 16;

    for (int blk = 0; blk < p.nblk; ++blk) {
        const uint32_t meta = rows[blk * 16 + j];
        long long pm = mg * p.G + row_mol(meta);
        pm = pm < p.B ? pm : p.B - 1;
        const long long nsrc = pm * p.A + row_src(meta), ndst = pm * p.A + row_dst(meta);
        const size_t prow0 = ((size_t)gi * p.nblk + blk) * 16;
        const float rx = p.x[nsrc * 3 + 0] - p.x[ndst * 3 + 0];
        const float ry = p.x[nsrc * 3 + 1] - p.x[ndst * 3 + 1];
        const float rz = p.x[nsrc * 3 + 2] - p.x[ndst * 3 + 2];
        const float dist = sqrtf(rx * rx + ry * ry + rz * rz);
        // ---- filter branch, (value, d/d|r|) pair
        OP g2, tg2;
        {
            A16 t1, u1;
            {
                OP enc, tenc;
                {
                    A16 t, u;
                    r16::posenc_dual(t, u, dist / p.length_scale, 1.0f / p.length_scale, q);      // seed d|r| = 1
                    enc.set(t); tenc.set(u);
                }
#pragma unroll
                for (int c = 0; c < NB; ++c) {
                    const f32x4* wl = pipe.acquire();
                    f32x4 a0 = r16::load_block(vec + EV::W_B0 * F, 2 * c, q), a1 = r16::load_block(vec + EV::W_B0 * F, 2 * c + 1, q);
                    f32x4 b0 = Z4, b1 = Z4;
                    r16::gemm_bt2(a0, a1, b0, b1, enc, tenc, wl, lane);
                    t1.b[2 * c] = a0; t1.b[2 * c + 1] = a1; u1.b[2 * c] = b0; u1.b[2 * c + 1] = b1;
                    pipe.release();
                }
            }
            r16::ln_silu_dual(t1, u1, vec + EV::W_G0 * F, vec + EV::W_BE0 * F, q);
            {
                OP g1, tg1;
                g1.set(t1); tg1.set(u1);
#pragma unroll
                for (int c = 0; c < NB; ++c) {
                    const f32x4* wl = pipe.acquire();
                    f32x4 a0 = r16::load_block(vec + EV::W_B1 * F, 2 * c, q), a1 = r16::load_block(vec + EV::W_B1 * F, 2 * c + 1, q);
                    f32x4 b0 = Z4, b1 = Z4;
                    r16::gemm_bt2(a0, a1, b0, b1, g1, tg1, wl, lane);
                    t1.b[2 * c] = a0; t1.b[2 * c + 1] = a1; u1.b[2 * c] = b0; u1.b[2 * c + 1] = b1;
                    pipe.release();
                }
            }
            r16::ln_silu_dual(t1, u1, vec + EV::W_G1 * F, vec + EV::W_BE1 * F, q);
            g2.set(t1); tg2.set(u1);
        }
        // ---- phi branch forward, LayerNorm statistics parked for the tangent passes
        OP h2;
        {
            f32x4* stp = reinterpret_cast<f32x4*>(p.st) + ((size_t)(gi * p.nblk + blk) * 4 * NBK) * 64 + lane;
            auto park = [&](int which, const A16& v) {
                if (group_ok) {
#pragma unroll
                    for (int nb = 0; nb < NBK; ++nb) stp[(size_t)(which * NBK + nb) * 64] = v.b[nb];
                }
            };
            A16 t1;
            {
                OP ein;
                if (first) r16::load_set(t1, p.edge_emb + row_type(meta) * F, q);
                else       r16::load_set(t1, p.e + (prow0 + j) * F, q);
                ein.set(t1);
                const float* prow = p.P + (size_t)nsrc * F;
#pragma unroll
                for (int c = 0; c < NB; ++c) {
                    const f32x4* wl = pipe.acquire();
                    f32x4 a0 = r16::load_block(prow, 2 * c, q), a1 = r16::load_block(prow, 2 * c + 1, q);
                    r16::gemm_bt(a0, a1, ein, wl, lane);
                    t1.b[2 * c] = a0; t1.b[2 * c + 1] = a1;
                    pipe.release();
                }
            }
            {
                A16 nn, kk;
                r16::ln_silu_stats(t1, nn, kk, vec + EV::P_G0 * F, vec + EV::P_BE0 * F, q);
                park(0, nn); park(1, kk);
            }
            {
                OP h1;
                h1.set(t1);
#pragma unroll
                for (int c = 0; c < NB; ++c) {
                    const f32x4* wl = pipe.acquire();
                    f32x4 a0 = r16::load_block(vec + EV::P_B1 * F, 2 * c, q), a1 = r16::load_block(vec + EV::P_B1 * F, 2 * c + 1, q);
                    r16::gemm_bt(a0, a1, h1, wl, lane);
                    t1.b[2 * c] = a0; t1.b[2 * c + 1] = a1;
                    pipe.release();
                }
            }
            {
                A16 nn, kk;
                r16::ln_silu_stats(t1, nn, kk, vec + EV::P_G1 * F, vec + EV::P_BE1 * F, q);
                park(2, nn); park(3, kk);
            }
            h2.set(t1);
        }
        f32x4* wq = reinterpret_cast<f32x4*>(p.wq) + ((size_t)(gi * p.nblk + blk) * 5 * NB) * 6 * 64 + lane;
        auto put = [&](int c, int nbo) {
            f32x4 a0 = Z4, a1 = Z4, b0 = Z4, b1 = Z4, tb0 = Z4, tb1 = Z4;
            const f32x4* wl0 = pipe.acquire();
            r16::gemm_fl(a0, a1, h2, wl0, lane);
            pipe.release();
            const f32x4* wl1 = pipe.acquire();
            r16::gemm_fl2(b0, b1, tb0, tb1, g2, tg2, wl1, lane);
            pipe.release();
            const float* bp = vec + (EV::P_B2 + c) * F + 32 * nbo + j;
            const float* bw = vec + (EV::W_B2 + c) * F + 32 * nbo + j;
            if (group_ok) {
                f32x4* o = wq + (size_t)(c * NB + nbo) * 6 * 64;
                o[0] = a0 + bp[0]; o[64] = a1 + bp[16]; o[128] = b0 + bw[0]; o[192] = b1 + bw[16]; o[256] = tb0; o[320] = tb1;
            }
        };
#pragma unroll 1
        for (int nbo = 0; nbo < NB; ++nbo) {          // consumption order of painn_edge_kernel: ds, de, sed, gates, cross gates
            put(2, nbo);
            if (!last) put(3, nbo);
            put(1, nbo);
            if (!first) { put(0, nbo); put(4, nbo); }
        }
    }
    pipe.drain();
}

// ================================================================================================== tangent edge kernel
// One wave per virtual group.  Per (edge, direction) row only TANGENT products remain: the phi branch's hidden layers and
// output chunks applied to the tangent of [s[src] | e]; every primal quantity comes from the primal pass above.
template <int NBK, bool SPLIT>
__global__ __launch_bounds__(256, (NBK <= 8 ? 2 : 1)) void painn_jvp_edge_kernel(const JvpEdgeParams p)
{
    constexpr int F = 16 * NBK, NB = (F + 31) / 32, WAVES = 4, T = 64 * WAVES, CH4 = 256 * NB;
    using A16 = r16::Act<NBK>;
    using OP = r16::Opnd<NBK, SPLIT>;
    extern __shared__ f32x4 lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), j = lane & 15, q = lane >> 4;
    float* scratch = reinterpret_cast<float*>(lds + 4 * CH4) + wave * 128;         // [16 rows][8]: edge_dir, d|r|', edge_dir', 0
    float* vec = reinterpret_cast<float*>(lds + 4 * CH4) + WAVES * 128;            // [EV::COUNT][F]
    for (int i = threadIdx.x; i < EV::COUNT * F / 4; i += T)
        reinterpret_cast<f32x4*>(vec)[i] = reinterpret_cast<const f32x4*>(p.vecs)[i];
    PipeDMA<NB, T, 2> pipe;
    pipe.init(reinterpret_cast<const f32x4*>(p.stream), p.nch, lds, wave, lane);

#ifndef TI_JVP_XCD
#define TI_JVP_XCD 1
#endif
    // virtual group.  The D directions of a primal group (consecutive virtual groups) read the same primal-pass rows (wq, st: 154 KB per
    // row block at F = 128), the same P / v / e rows; dealt round-robin over the XCDs, every L2 fetched them again from HBM (140 GB
    // read per launch, profiles/r03f_divergence_pmc_summary.txt): neighbours in the logical order share an XCD instead.
    const long long gi_raw = (TI_JVP_XCD ? xcd_swizzle(blockIdx.x, gridDim.x) : (long long)blockIdx.x) * WAVES + wave;
    const bool group_ok = gi_raw < p.n_groups;
    const long long gi = group_ok ? gi_raw : p.n_groups - 1;
    // gi = (molecule group * D + direction) * P + part
    const long long vmg = gi / p.parts;                                                 // virtual molecule group
    const int part = (int)(gi - vmg * p.parts);
    const long long mg = vmg / p.D;                                                 // molecule group
    const int dsel = (int)(vmg - mg * p.D);                                         // seed direction
    const long long pg = mg * p.parts + part;                                         // primal group (incl. part): rows of e, wq, st
    const uint32_t* rows = p.rows + (size_t)part * p.nblk * 16;
    const int32_t* slotnode = p.slotnode + (size_t)part * p.nblk * 16;
    const bool first = p.first != 0, last = p.last != 0;

    for (int blk = 0; blk < p.nblk; ++blk) {
        // ---- geometry of this lane's row and its tangent
        const uint32_t meta = rows[blk * 16 + j];
        long long pm = mg * p.G + row_mol(meta);
        pm = pm < p.B ? pm : p.B - 1;
        const long long nsrc = pm * p.A + row_src(meta), ndst = pm * p.A + row_dst(meta);
        const size_t trow0 = ((size_t)gi * p.nblk + blk) * 16;                       // tangent rows of this block
        {
            const float rx = p.x[nsrc * 3 + 0] - p.x[ndst * 3 + 0];
            const float ry = p.x[nsrc * 3 + 1] - p.x[ndst * 3 + 1];
            const float rz = p.x[nsrc * 3 + 2] - p.x[ndst * 3 + 2];
            float tx, ty, tz;
            if (p.xdot) {
                tx = p.xdot[nsrc * 3 + 0] - p.xdot[ndst * 3 + 0];
                ty = p.xdot[nsrc * 3 + 1] - p.xdot[ndst * 3 + 1];
                tz = p.xdot[nsrc * 3 + 2] - p.xdot[ndst * 3 + 2];
            } else {                                                // unit seed on (atom, component) = (dsel / 3, dsel % 3)
                const int sa = dsel / 3, sc = dsel - 3 * sa;
                const float sg = (float)((row_src(meta) == sa) - (row_dst(meta) == sa));
                tx = sc == 0 ? sg : 0.f; ty = sc == 1 ? sg : 0.f; tz = sc == 2 ? sg : 0.f;
            }
            const float dist = sqrtf(rx * rx + ry * ry + rz * rz);
            const float ddist = dist > 0.f ? (rx * tx + ry * ty + rz * tz) / dist : 0.f;
            const float den = 1.0f + dist, k = ddist / (den * den);
            if (q == 0) {
                *reinterpret_cast<f32x4*>(scratch + j * 8) = f32x4{rx / den, ry / den, rz / den, ddist};
                *reinterpret_cast<f32x4*>(scratch + j * 8 + 4) = f32x4{tx / den - rx * k, ty / den - ry * k, tz / den - rz * k, 0.f};
            }
        }
        // ---- tangent of phi's hidden layers (first layer: s and e do not depend on x yet, the whole tangent is zero)
        OP th2;
        if (!first) {
            const f32x4* stp = reinterpret_cast<const f32x4*>(p.st) + ((size_t)(pg * p.nblk + blk) * 4 * NBK) * 64 + lane;
            auto stat = [&](int which, A16& v) {
#pragma unroll
                for (int nb = 0; nb < NBK; ++nb) v.b[nb] = stp[(size_t)(which * NBK + nb) * 64];
            };
            A16 u1;
            {
                OP tein;
                r16::load_set(u1, p.te + (trow0 + j) * F, q);
                tein.set(u1);
                const float* tprow = p.tP + (size_t)((vmg * p.G + row_mol(meta)) * p.A + row_src(meta)) * F;
#pragma unroll
                for (int c = 0; c < NB; ++c) {
                    const f32x4* wl = pipe.acquire();
                    f32x4 b0 = r16::load_block(tprow, 2 * c, q), b1 = r16::load_block(tprow, 2 * c + 1, q);
                    r16::gemm_bt(b0, b1, tein, wl, lane);
                    u1.b[2 * c] = b0; u1.b[2 * c + 1] = b1;
                    pipe.release();
                }
            }
            {
                A16 nn, kk;
                stat(0, nn); stat(1, kk);
                r16::ln_tangent(u1, nn, kk);
            }
            {
                OP th1;
                th1.set(u1);
#pragma unroll
                for (int c = 0; c < NB; ++c) {
                    const f32x4* wl = pipe.acquire();
                    f32x4 b0 = Z4, b1 = Z4;
                    r16::gemm_bt(b0, b1, th1, wl, lane);
                    u1.b[2 * c] = b0; u1.b[2 * c + 1] = b1;
                    pipe.release();
                }
            }
            {
                A16 nn, kk;
                stat(2, nn); stat(3, kk);
                r16::ln_tangent(u1, nn, kk);
            }
            th2.set(u1);
        } else {
#pragma unroll
            for (int c = 0; c < 2 * NB; ++c) { (void)pipe.acquire(); pipe.release(); }      // keep the stream in phase
        }
        // ---- output layer, flipped (features on lanes, the block's rows 4q + r in registers); see painn_edge_kernel
        uint32_t mi[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) mi[r] = rows[blk * 16 + 4 * q + r];
        // per-atom sums over the block's rows on VALU lane swaps (r16::QuarterSum; a block has at most 4 destination atoms): quarter q
        // of the wave ends up with the 16-row sum of slot q, one atomic instruction per 16-feature half adds every slot of the block
        r16::QuarterSum<4> qs;
#pragma unroll
        for (int r = 0; r < 4; ++r) qs.set_row(r, row_slot(mi[r]));
        int qnode;
        bool qfirst;                                 // first block of that atom: its sums replace the tangent accumulators' contents
        {
            const int sn = slotnode[blk * 16 + q];
            const long long m2 = slot_mol(sn);
            qnode = (sn >= 0 && group_ok && mg * p.G + m2 < p.B) ? (int)((vmg * p.G + m2) * p.A + (sn & 255)) : -1;     // TANGENT node
            qfirst = (sn & SLOT_FIRST_TOUCH) != 0;
        }
        f32x4 dir[4], tdir[4], dd;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            dir[r] = *reinterpret_cast<const f32x4*>(scratch + (4 * q + r) * 8);
            tdir[r] = *reinterpret_cast<const f32x4*>(scratch + (4 * q + r) * 8 + 4);
            dd[r] = dir[r][3];
        }
        const f32x4* wq = reinterpret_cast<const f32x4*>(p.wq) + ((size_t)(pg * p.nblk + blk) * 5 * NB) * 6 * 64 + lane;
        // value and tangent of (phi_c + b)(w_c + b) for output chunk c, 32 features as two 16-feature blocks
        auto out_pair = [&](int c, int nbo, f32x4& r0, f32x4& r1, f32x4& d0, f32x4& d1) {
            // raised issue priority from the filter-product loads to the tangent products (mfma_chain.hpp: gemm_on_pipe): +1.5 % on the
            // divergence workload (profiles/r03i_setprio_timing.txt); around the hidden layers' products it changes nothing, and bracketing
            // only the matrix instructions here costs hipcc 680 spilled registers (s_setprio is a scheduling boundary)
            __builtin_amdgcn_s_setprio(1);
            const f32x4* g = wq + (size_t)(c * NB + nbo) * 6 * 64;
            const f32x4 A0 = g[0], A1 = g[64], B0 = g[128], B1 = g[192], Q0 = g[256], Q1 = g[320];
            f32x4 ta0 = Z4, ta1 = Z4;
            const f32x4* wl = pipe.acquire();
            if (!first) r16::gemm_fl(ta0, ta1, th2, wl, lane);
            pipe.release();
            r0 = A0 * B0; r1 = A1 * B1;
            d0 = ta0 * B0 + A0 * (dd * Q0); d1 = ta1 * B1 + A1 * (dd * Q1);
            __builtin_amdgcn_s_setprio(0);
        };
        auto emit = [&](const f32x4& v0, const f32x4& v1, float* dst, size_t stride) {
            const float z0 = qs.sum(v0), z1 = qs.sum(v1);
            if (qnode >= 0) { float* d = dst + (size_t)qnode * stride; acc_out(d, z0, qfirst); acc_out(d + 16, z1, qfirst); }
        };

#pragma unroll 1
        for (int nbo = 0; nbo < NB; ++nbo) {
            const int fo = 32 * nbo + j;
            {   // ds
                f32x4 v0, v1, d0, d1;
                out_pair(2, nbo, v0, v1, d0, d1);
                emit(d0, d1, p.tdsacc + fo, F);
            }
            if (!last) {   // de: te += d(de).  Every tangent edge row has ONE owner (this wave), so the update is a plain load / add / store
                // instead of the 32 fire-and-forget atomic instructions per row block it used to be (more than half of this kernel's
                // dword atomics).  Measured: the launch takes the same 46.9 ms either way (profiles/r03f_divergence_*): like the primal
                // message kernel this one is bound by its serial per-wave timeline, not by L2's atomic rate.  The old rows are
                // requested before the products and consumed behind them.
                f32x4 o0 = Z4, o1 = Z4;
                if (!first) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float* ep = p.te + (trow0 + 4 * q + r) * F + fo;
                        o0[r] = ep[0]; o1[r] = ep[16];
                    }
                }
                f32x4 v0, v1, d0, d1;
                out_pair(3, nbo, v0, v1, d0, d1);
                if (group_ok) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float* ep = p.te + (trow0 + 4 * q + r) * F + fo;
                        ep[0] = o0[r] + d0[r]; ep[16] = o1[r] + d1[r];
                    }
                }
            }
            {   // equivariant message
                f32x4 sed0, sed1, tsed0, tsed1, gt0 = Z4, gt1 = Z4, tgt0 = Z4, tgt1 = Z4;
                out_pair(1, nbo, sed0, sed1, tsed0, tsed1);
                f32x4 vs[3][2], tvs[3][2];
                if (!first) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        long long pm2 = mg * p.G + row_mol(mi[r]);
                        pm2 = pm2 < p.B ? pm2 : p.B - 1;
                        const float* vp = p.v + (size_t)(pm2 * p.A + row_src(mi[r])) * 3 * F + fo;
                        const float* tp = p.tv + (size_t)((vmg * p.G + row_mol(mi[r])) * p.A + row_src(mi[r])) * 3 * F + fo;
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            vs[c][0][r] = vp[c * F]; vs[c][1][r] = vp[c * F + 16];
                            tvs[c][0][r] = tp[c * F]; tvs[c][1][r] = tp[c * F + 16];
                        }
                    }
                    out_pair(0, nbo, gt0, gt1, tgt0, tgt1);
                }
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    f32x4 v0, v1;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v0[r] = tsed0[r] * dir[r][c] + sed0[r] * tdir[r][c];
                        v1[r] = tsed1[r] * dir[r][c] + sed1[r] * tdir[r][c];
                        if (!first) {
                            v0[r] += tgt0[r] * vs[c][0][r] + gt0[r] * tvs[c][0][r];
                            v1[r] += tgt1[r] * vs[c][1][r] + gt1[r] * tvs[c][1][r];
                        }
                    }
                    emit(v0, v1, p.tdvacc + c * F + fo, 3 * F);
                }
                if (!first) {
                    f32x4 cg0, cg1, tcg0, tcg1;
                    out_pair(4, nbo, cg0, cg1, tcg0, tcg1);
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        f32x4 v0, v1;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            v0[r] = tcg0[r] * dir[r][c] + cg0[r] * tdir[r][c];
                            v1[r] = tcg1[r] * dir[r][c] + cg1[r] * tdir[r][c];
                        }
                        emit(v0, v1, p.tcacc + c * F + fo, 3 * F);
                    }
                }
            }
        }
        if (p.pad) { (void)pipe.acquire(); pipe.release(); }      // odd chunk count: swallow the pad chunk, stay in phase with the superchunk ring
    }
    pipe.drain();
}

// ================================================================================================== primal node pass
// Everything the per-direction update kernel needs from the forward pass of this layer's update block, ONCE per primal node
// (16 nodes per wave, chain layout: float4 = features 16 nb + 4 (lane >> 4) .. + 3 of node row (lane & 15)):
//   ns[((tile * NS_COUNT + which) * NBK + nb) * 64 + lane],  tile = node / 16,  which:
//     0..2 vv_c / |vv|   (c = x, y, z; vv = V v_eff; 0 where |vv| = 0 like torch.norm's backward)
//     3    |vv|          4..7 n, kk of the update MLP's two LayerNorms      8 scale_squared_norm output   9 gates
//     10..12 U v_eff
// Runs after the primal edge kernel and before the primal update kernel of the layer (reads what the latter overwrites).
constexpr int NS_COUNT = 13;
template <int NBK, bool SPLIT>
__global__ __launch_bounds__(256, 1) void painn_jvp_node_kernel(const JvpNodeParams p)
{
    constexpr int F = 16 * NBK, NB = (F + 31) / 32, WAVES = 4, T = 64 * WAVES, CH4 = 256 * NB;
    using A16 = r16::Act<NBK>;
    using OP = r16::Opnd<NBK, SPLIT>;
    extern __shared__ f32x4 lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), j = lane & 15, q = lane >> 4;
    float* vec = reinterpret_cast<float*>(lds + 4 * CH4);
    for (int i = threadIdx.x; i < UV::COUNT * F / 4; i += T)
        reinterpret_cast<f32x4*>(vec)[i] = reinterpret_cast<const f32x4*>(p.vecs)[i];
    PipeDMA<NB, T, 2> pipe;
    pipe.init(reinterpret_cast<const f32x4*>(p.stream), p.nch, lds, wave, lane);

    const long long tile = (long long)blockIdx.x * WAVES + wave;
    const long long node = tile * 16 + j;
    const bool tile_ok = tile * 16 < p.N;
    const size_t pn = (size_t)(node < p.N ? node : p.N - 1);
    const float *vb = p.v + pn * 3 * F, *db = p.dvacc + pn * 3 * F, *cb = p.cacc + pn * 3 * F, *sb = p.s + pn * F, *ab = p.dsacc + pn * F;
    f32x4* nsp = reinterpret_cast<f32x4*>(p.ns) + ((size_t)(tile_ok ? tile : 0) * NS_COUNT * NBK) * 64 + lane;
    auto park = [&](int which, const A16& v) {
        if (tile_ok) {
#pragma unroll
            for (int nb = 0; nb < NBK; ++nb) nsp[(size_t)(which * NBK + nb) * 64] = v.b[nb];
        }
    };
    auto veff = [&](int c, A16& t) {
        const int c1 = (c + 1) % 3, c2 = (c + 2) % 3;
#pragma unroll
        for (int nb = 0; nb < NBK; ++nb) {
            const f32x4 vc = r16::load_block(vb + c * F, nb, q), dd = r16::load_block(db + c * F, nb, q);
            const f32x4 v1 = r16::load_block(vb + c1 * F, nb, q), v2 = r16::load_block(vb + c2 * F, nb, q);
            const f32x4 k1 = r16::load_block(cb + c1 * F, nb, q), k2 = r16::load_block(cb + c2 * F, nb, q);
            t.b[nb] = (vc + dd) + (k1 * v2 - k2 * v1);
        }
    };
    // ---- phase A: vv_c = V v_eff_c (parked raw, normalised at the end of the phase), n2 = |vv|^2
    A16 n2;
#pragma unroll
    for (int nb = 0; nb < NBK; ++nb) n2.b[nb] = Z4;
#pragma unroll 1
    for (int c = 0; c < 3; ++c) {
        OP ve;
        {
            A16 t;
            veff(c, t);
            ve.set(t);
        }
        A16 vv;
#pragma unroll
        for (int ch = 0; ch < NB; ++ch) {
            const f32x4* wl = pipe.acquire();
            f32x4 a0 = Z4, a1 = Z4;
            r16::gemm_bt(a0, a1, ve, wl, lane);
            vv.b[2 * ch] = a0; vv.b[2 * ch + 1] = a1;
            n2.b[2 * ch] += a0 * a0; n2.b[2 * ch + 1] += a1 * a1;
            pipe.release();
        }
        park(c, vv);
    }
#pragma unroll
    for (int nb = 0; nb < NBK; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) n2.b[nb][r] = sqrtf(n2.b[nb][r]);
    park(3, n2);
    if (tile_ok) {                                 // vv_c <- vv_c / |vv| in place (same lane wrote it)
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int nb = 0; nb < NBK; ++nb) {
                f32x4 v = nsp[(size_t)(c * NBK + nb) * 64];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = n2.b[nb][r] > 0.f ? v[r] / n2.b[nb][r] : 0.f;
                nsp[(size_t)(c * NBK + nb) * 64] = v;
            }
    }
    // ---- phase B: MLP([ |vv| , s + ds ]) with parked LayerNorm statistics
    OP h2;
    {
        A16 t;
#pragma unroll
        for (int nb = 0; nb < NBK; ++nb) t.b[nb] = r16::load_block(vec + UV::B0 * F, nb, q);
        {
            OP nn;
            nn.set(n2);
#pragma unroll
            for (int ch = 0; ch < NB; ++ch) {
                const f32x4* wl = pipe.acquire();
                r16::gemm_bt(t.b[2 * ch], t.b[2 * ch + 1], nn, wl, lane);
                pipe.release();
            }
        }
        {
            OP ss;
            {
                A16 x;
#pragma unroll
                for (int nb = 0; nb < NBK; ++nb) x.b[nb] = r16::load_block(sb, nb, q) + r16::load_block(ab, nb, q);
                ss.set(x);
            }
#pragma unroll
            for (int ch = 0; ch < NB; ++ch) {
                const f32x4* wl = pipe.acquire();
                r16::gemm_bt(t.b[2 * ch], t.b[2 * ch + 1], ss, wl, lane);
                pipe.release();
            }
        }
        {
            A16 nn, kk;
            r16::ln_silu_stats(t, nn, kk, vec + UV::G0 * F, vec + UV::BE0 * F, q);
            park(4, nn); park(5, kk);
        }
        {
            OP h1;
            h1.set(t);
#pragma unroll
            for (int ch = 0; ch < NB; ++ch) {
                const f32x4* wl = pipe.acquire();
                f32x4 a0 = r16::load_block(vec + UV::B1 * F, 2 * ch, q), a1 = r16::load_block(vec + UV::B1 * F, 2 * ch + 1, q);
                r16::gemm_bt(a0, a1, h1, wl, lane);
                t.b[2 * ch] = a0; t.b[2 * ch + 1] = a1;
                pipe.release();
            }
        }
        {
            A16 nn, kk;
            r16::ln_silu_stats(t, nn, kk, vec + UV::G1 * F, vec + UV::BE1 * F, q);
            park(6, nn); park(7, kk);
        }
        h2.set(t);
    }
    {   // scale_squared_norm (the add_invariant chunk is skipped: it enters no tangent), then gates
        A16 qq, gg;
#pragma unroll
        for (int ch = 0; ch < NB; ++ch) {
            const f32x4* wl = pipe.acquire();
            f32x4 q0 = r16::load_block(vec + (UV::B2 + 1) * F, 2 * ch, q), q1 = r16::load_block(vec + (UV::B2 + 1) * F, 2 * ch + 1, q);
            r16::gemm_bt(q0, q1, h2, wl, lane);
            qq.b[2 * ch] = q0; qq.b[2 * ch + 1] = q1;
            pipe.release();
            (void)pipe.acquire();
            pipe.release();
        }
        park(8, qq);
#pragma unroll
        for (int ch = 0; ch < NB; ++ch) {
            const f32x4* wl = pipe.acquire();
            f32x4 a0 = r16::load_block(vec + UV::B2 * F, 2 * ch, q), a1 = r16::load_block(vec + UV::B2 * F, 2 * ch + 1, q);
            r16::gemm_bt(a0, a1, h2, wl, lane);
            gg.b[2 * ch] = a0; gg.b[2 * ch + 1] = a1;
            pipe.release();
        }
        park(9, gg);
    }
    // ---- phase C: U v_eff
#pragma unroll 1
    for (int c = 0; c < 3; ++c) {
        OP ve;
        {
            A16 t;
            veff(c, t);
            ve.set(t);
        }
        A16 uv;
#pragma unroll
        for (int ch = 0; ch < NB; ++ch) {
            const f32x4* wl = pipe.acquire();
            f32x4 a0 = Z4, a1 = Z4;
            r16::gemm_bt(a0, a1, ve, wl, lane);
            uv.b[2 * ch] = a0; uv.b[2 * ch + 1] = a1;
            pipe.release();
        }
        park(10 + c, uv);
    }
    pipe.drain();
}

// ================================================================================================== tangent update kernel
// Runs BEFORE the primal update kernel of the same layer: reads the primal v and cacc as the edge kernel left them and the
// primal node pass output, advances ts, tv, tP; the tangent accumulators are consumed and zeroed.  Tangent products only.
template <int NBK, bool SPLIT>
__global__ __launch_bounds__(256, (NBK <= 8 ? 2 : 1)) void painn_jvp_update_kernel(const JvpUpdateParams p)
{
    constexpr int F = 16 * NBK, NB = (F + 31) / 32, WAVES = 4, T = 64 * WAVES, CH4 = 256 * NB;
    using A16 = r16::Act<NBK>;
    using OP = r16::Opnd<NBK, SPLIT>;
    extern __shared__ f32x4 lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), j = lane & 15, q = lane >> 4;
    PipeDMA<NB, T, 2> pipe;
    pipe.init(reinterpret_cast<const f32x4*>(p.stream), p.nch, lds, wave, lane);

    const long long node = ((long long)blockIdx.x * WAVES + wave) * 16 + j;      // virtual node
    const long long nd = node < p.N ? node : p.N - 1;
    const long long vm = nd / p.A, vg = vm / p.G;
    const long long pm_raw = (vg / p.D) * p.G + (vm - vg * p.G);                      // molecule of this virtual molecule
    const bool ok = node < p.N && pm_raw < p.B;
    const size_t pn = (size_t)((pm_raw < p.B ? pm_raw : p.B - 1) * p.A + (nd - vm * p.A));   // primal node
    const float *vb = p.v + pn * 3 * F, *cb = p.cacc + pn * 3 * F;
    float *tvb = p.tv + (size_t)nd * 3 * F, *tdb = p.tdvacc + (size_t)nd * 3 * F, *tcb = p.tcacc + (size_t)nd * 3 * F;
    float *tsb = p.ts + (size_t)nd * F, *tab = p.tdsacc + (size_t)nd * F;
    // node-pass data of THIS lane's primal node: tile pn / 16, row pn % 16, this lane's feature quarter q
    const f32x4* nsp = reinterpret_cast<const f32x4*>(p.ns) + ((pn >> 4) * NS_COUNT * NBK) * 64 + (q * 16 + (pn & 15));
    auto stat = [&](int which, int nb) { return nsp[(size_t)(which * NBK + nb) * 64]; };

    // ---- phase A: tv_eff (parked in tdvacc), n' = sum_c (vv_c / |vv|) . (V tv_eff_c)
    // Feature block outermost, like the primal update kernel: tv, tcacc and tdvacc are read once each (the cross product needs all three
    // components of a block; per-component loops read tv three times and tcacc twice, and those repeats came from HBM).
    A16 tn;
#pragma unroll
    for (int nb = 0; nb < NBK; ++nb) tn.b[nb] = Z4;
    {
        A16 u[3];
#pragma unroll
        for (int nb = 0; nb < NBK; ++nb) {
            f32x4 vv[3], kk[3], tvv[3], tkk[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                vv[c] = r16::load_block(vb + c * F, nb, q); kk[c] = r16::load_block(cb + c * F, nb, q);
                tvv[c] = r16::load_block(tvb + c * F, nb, q); tkk[c] = r16::load_block(tcb + c * F, nb, q);
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int c1 = (c + 1) % 3, c2 = (c + 2) % 3;
                const f32x4 tdd = r16::load_block(tdb + c * F, nb, q);
                u[c].b[nb] = (tvv[c] + tdd) + ((tkk[c1] * vv[c2] + kk[c1] * tvv[c2]) - (tkk[c2] * vv[c1] + kk[c2] * tvv[c1]));
                if (ok) r16::store_block(tdb + c * F, nb, q, u[c].b[nb]);
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            OP tve;
            tve.set(u[c]);
#pragma unroll
            for (int ch = 0; ch < NB; ++ch) {
                const f32x4* wl = pipe.acquire();
                f32x4 b0 = Z4, b1 = Z4;
                r16::gemm_bt(b0, b1, tve, wl, lane);
                tn.b[2 * ch] += stat(c, 2 * ch) * b0; tn.b[2 * ch + 1] += stat(c, 2 * ch + 1) * b1;
                pipe.release();
            }
        }
    }
    // ---- phase B: tangent of MLP([ |vv| , s + ds ])
    A16 tsn;                                        // ts + tds, then ts after the update (phase D's operand): read once, kept in registers
    OP th2;
    {
        A16 u;
#pragma unroll
        for (int nb = 0; nb < NBK; ++nb) u.b[nb] = Z4;
        {
            OP tnn;
            tnn.set(tn);
#pragma unroll
            for (int ch = 0; ch < NB; ++ch) {
                const f32x4* wl = pipe.acquire();
                r16::gemm_bt(u.b[2 * ch], u.b[2 * ch + 1], tnn, wl, lane);
                pipe.release();
            }
        }
        {
            OP tss;
#pragma unroll
            for (int nb = 0; nb < NBK; ++nb) tsn.b[nb] = r16::load_block(tsb, nb, q) + r16::load_block(tab, nb, q);
            tss.set(tsn);
#pragma unroll
            for (int ch = 0; ch < NB; ++ch) {
                const f32x4* wl = pipe.acquire();
                r16::gemm_bt(u.b[2 * ch], u.b[2 * ch + 1], tss, wl, lane);
                pipe.release();
            }
        }
        {
            A16 nn, kk;
#pragma unroll
            for (int nb = 0; nb < NBK; ++nb) { nn.b[nb] = stat(4, nb); kk.b[nb] = stat(5, nb); }
            r16::ln_tangent(u, nn, kk);
        }
        {
            OP th1;
            th1.set(u);
#pragma unroll
            for (int ch = 0; ch < NB; ++ch) {
                const f32x4* wl = pipe.acquire();
                f32x4 b0 = Z4, b1 = Z4;
                r16::gemm_bt(b0, b1, th1, wl, lane);
                u.b[2 * ch] = b0; u.b[2 * ch + 1] = b1;
                pipe.release();
            }
        }
        {
            A16 nn, kk;
#pragma unroll
            for (int nb = 0; nb < NBK; ++nb) { nn.b[nb] = stat(6, nb); kk.b[nb] = stat(7, nb); }
            r16::ln_tangent(u, nn, kk);
        }
        th2.set(u);
    }
    // ---- output chunks: ts += 2 n n' q + n^2 q' + add'
#pragma unroll
    for (int ch = 0; ch < NB; ++ch) {
        const f32x4* wl = pipe.acquire();
        f32x4 tq0 = Z4, tq1 = Z4;
        r16::gemm_bt(tq0, tq1, th2, wl, lane);
        pipe.release();
        wl = pipe.acquire();
        f32x4 ta0 = Z4, ta1 = Z4;
        r16::gemm_bt(ta0, ta1, th2, wl, lane);
        pipe.release();
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int nb = 2 * ch + k;
            f32x4 so = tsn.b[nb];
            const f32x4 tqq = k ? tq1 : tq0, taa = k ? ta1 : ta0, nrm = stat(3, nb), qq = stat(8, nb);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float n = nrm[r], dn = tn.b[nb][r];
                so[r] = so[r] + (((2.0f * n) * dn) * qq[r] + (n * n) * tqq[r] + taa[r]);
            }
            tsn.b[nb] = so;
            if (ok) {
                r16::store_block(tsb, nb, q, so);
                if (p.zero_acc) r16::store_block(tab, nb, q, Z4);
            }
        }
    }
    A16 tgg;
#pragma unroll
    for (int ch = 0; ch < NB; ++ch) {
        const f32x4* wl = pipe.acquire();
        f32x4 b0 = Z4, b1 = Z4;
        r16::gemm_bt(b0, b1, th2, wl, lane);
        tgg.b[2 * ch] = b0; tgg.b[2 * ch + 1] = b1;
        pipe.release();
    }
    // ---- phase C: tv = tv_eff + (U tv_eff) gates + (U v_eff) tgates
#pragma unroll 1
    for (int c = 0; c < 3; ++c) {
        OP tve;
        A16 u;                                      // the parked tv_eff row: operand and addend, read once
        r16::load_set(u, tdb + c * F, q);
        tve.set(u);
#pragma unroll
        for (int ch = 0; ch < NB; ++ch) {
            const f32x4* wl = pipe.acquire();
            f32x4 b0 = Z4, b1 = Z4;
            r16::gemm_bt(b0, b1, tve, wl, lane);
            pipe.release();
            if (ok) {
                r16::store_block(tvb + c * F, 2 * ch, q, u.b[2 * ch] + (b0 * stat(9, 2 * ch) + stat(10 + c, 2 * ch) * tgg.b[2 * ch]));
                r16::store_block(tvb + c * F, 2 * ch + 1, q, u.b[2 * ch + 1] + (b1 * stat(9, 2 * ch + 1) + stat(10 + c, 2 * ch + 1) * tgg.b[2 * ch + 1]));
                if (p.zero_acc) {
                    r16::store_block(tdb + c * F, 2 * ch, q, Z4);
                    r16::store_block(tdb + c * F, 2 * ch + 1, q, Z4);
                }
            }
        }
    }
    if (ok && p.zero_acc) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int nb = 0; nb < NBK; ++nb) r16::store_block(tcb + c * F, nb, q, Z4);
    }
    // ---- phase D: tangent of P for the next message block (no bias)
    if (p.has_next) {
        OP sn;
        sn.set(tsn);                                // the rows written above, still in registers
#pragma unroll
        for (int ch = 0; ch < NB; ++ch) {
            const f32x4* wl = pipe.acquire();
            f32x4 a0 = Z4, a1 = Z4;
            r16::gemm_bt(a0, a1, sn, wl, lane);
            pipe.release();
            if (ok) { r16::store_block(p.tP + (size_t)nd * F, 2 * ch, q, a0); r16::store_block(p.tP + (size_t)nd * F, 2 * ch + 1, q, a1); }
        }
    }
    pipe.drain();
}

// ================================================================================================== tangent readout kernel
// tout[vnode][c] = (Vr . tv_c) gate + (Vr . v_c) tgate   (LayerReadout.forward, cpainn.py:425-437)
// The cross-lane sums below are VALU lane swaps (mfma_chain.hpp: xquarters), and that matters here: with ds_bpermute the split
// build of this kernel lost (Vr . v) sums of whole tiles whenever EXEC was narrowed for the store while the permute was still
// outstanding.
template <int NBK, bool SPLIT>
__global__ __launch_bounds__(256, (NBK <= 8 ? 2 : 1)) void painn_jvp_readout_kernel(const JvpReadoutParams p)
{
    constexpr int F = 16 * NBK, NB = (F + 31) / 32, WAVES = 4, T = 64 * WAVES, CH4 = 256 * NB;
    using A16 = r16::Act<NBK>;
    using OP = r16::Opnd<NBK, SPLIT>;
    extern __shared__ f32x4 lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), j = lane & 15, q = lane >> 4;
    float* vec = reinterpret_cast<float*>(lds + 4 * CH4);
    for (int i = threadIdx.x; i < RV::COUNT * F / 4; i += T)
        reinterpret_cast<f32x4*>(vec)[i] = reinterpret_cast<const f32x4*>(p.vecs)[i];
    PipeDMA<NB, T, 2> pipe;
    pipe.init(reinterpret_cast<const f32x4*>(p.stream), p.nch, lds, wave, lane);

    const long long node = ((long long)blockIdx.x * WAVES + wave) * 16 + j;      // virtual node
    const long long nd = node < p.N ? node : p.N - 1;
    const long long vm = nd / p.A, vg = vm / p.G;
    const long long pm_raw = (vg / p.D) * p.G + (vm - vg * p.G);                      // molecule of this virtual molecule
    const bool ok = node < p.N && pm_raw < p.B;
    const size_t pn = (size_t)((pm_raw < p.B ? pm_raw : p.B - 1) * p.A + (nd - vm * p.A));   // primal node

    A16 t, u;
    {
        OP ss, tss;
        {
            A16 x, y;
            r16::load_set(x, p.s + pn * F, q);
            r16::load_set(y, p.ts + (size_t)nd * F, q);
            ss.set(x); tss.set(y);
        }
#pragma unroll
        for (int ch = 0; ch < NB; ++ch) {
            const f32x4* wl = pipe.acquire();
            f32x4 a0 = r16::load_block(vec + RV::B0 * F, 2 * ch, q), a1 = r16::load_block(vec + RV::B0 * F, 2 * ch + 1, q);
            f32x4 b0 = Z4, b1 = Z4;
            r16::gemm_bt2(a0, a1, b0, b1, ss, tss, wl, lane);
            t.b[2 * ch] = a0; t.b[2 * ch + 1] = a1; u.b[2 * ch] = b0; u.b[2 * ch + 1] = b1;
            pipe.release();
        }
    }
    r16::ln_silu_dual(t, u, vec + RV::G0 * F, vec + RV::BE0 * F, q);
    {
        OP h1, th1;
        h1.set(t); th1.set(u);
#pragma unroll
        for (int ch = 0; ch < NB; ++ch) {
            const f32x4* wl = pipe.acquire();
            f32x4 a0 = r16::load_block(vec + RV::B1 * F, 2 * ch, q), a1 = r16::load_block(vec + RV::B1 * F, 2 * ch + 1, q);
            f32x4 b0 = Z4, b1 = Z4;
            r16::gemm_bt2(a0, a1, b0, b1, h1, th1, wl, lane);
            t.b[2 * ch] = a0; t.b[2 * ch + 1] = a1; u.b[2 * ch] = b0; u.b[2 * ch + 1] = b1;
            pipe.release();
        }
    }
    r16::ln_silu_dual(t, u, vec + RV::G1 * F, vec + RV::BE1 * F, q);
    float gate = 0.f, tgate = 0.f;
#pragma unroll
    for (int nb = 0; nb < NBK; ++nb) {
        const f32x4 w = r16::load_block(vec + RV::W2G * F, nb, q);
#pragma unroll
        for (int r = 0; r < 4; ++r) { gate = fmaf(t.b[nb][r], w[r], gate); tgate = fmaf(u.b[nb][r], w[r], tgate); }
    }
    gate = r16::xquarters(gate) + p.b2_gate;
    tgate = r16::xquarters(tgate);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float acc = 0.f, tacc = 0.f;
#pragma unroll
        for (int nb = 0; nb < NBK; ++nb) {
            const f32x4 w = r16::load_block(vec + RV::VR * F, nb, q);
            const f32x4 vv = r16::load_block(p.v + (pn * 3 + c) * F, nb, q), tv = r16::load_block(p.tv + ((size_t)nd * 3 + c) * F, nb, q);
#pragma unroll
            for (int r = 0; r < 4; ++r) { acc = fmaf(vv[r], w[r], acc); tacc = fmaf(tv[r], w[r], tacc); }
        }
        acc = r16::xquarters(acc); tacc = r16::xquarters(tacc);
        if (ok && q == 0) p.tout[nd * 3 + c] = tacc * gate + acc * tgate;
    }
    pipe.drain();
}

// div[b] = sum_d tangent[(virtual molecule of (b, d))][d]  (unit seeds: direction d = 3 atom + component is also the flat
// index inside [A][3]); fixed summation order
__global__ void painn_div_reduce_kernel(const float* __restrict__ tout, long long B, int D, int G, float* __restrict__ div)
{
    const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const long long pg = b / G, m = b - pg * G;
    float acc = 0.f;
    for (int d = 0; d < D; ++d) acc += tout[(size_t)(((pg * D + d) * G + m)) * D + d];
    div[b] = acc;
}

// ================================================================================================== launchers
static size_t jvp_edge_lds(int NB) { return 4 * (size_t)256 * NB * 16 + 4 * 128 * 4 + EV::COUNT * (size_t)32 * NB * 4; }
static size_t jvp_node_lds(int NB, int count)
{
    return 4 * (size_t)256 * NB * 16 + (size_t)count * 32 * NB * 4;
}

template <typename K>
static hipError_t set_lds(K kernel, size_t bytes)
{
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

#ifdef TI_DEV_NB4_ONLY
#define TI_SMALL_NB(...)
#else
#define TI_SMALL_NB(...) __VA_ARGS__
#endif
#define TI_JVP_DISPATCH(NBv, ...) \
    switch (NBv) {                                                           \
        TI_SMALL_NB(case 1: { constexpr int NBK = 2; __VA_ARGS__; } break;)  \
        TI_SMALL_NB(case 2: { constexpr int NBK = 4; __VA_ARGS__; } break;)  \
        case 4: { constexpr int NBK = 8; __VA_ARGS__; } break;               \
        TI_SMALL_NB(case 8: { constexpr int NBK = 16; __VA_ARGS__; } break;) \
        default: return hipErrorInvalidValue;                                \
    }

template <int NBK>
static hipError_t configure_jvp_nbk(int NB)
{
    hipError_t e;
    if ((e = set_lds(painn_jvp_filter_kernel<NBK, false>, jvp_node_lds(NB, EV::COUNT))) != hipSuccess) return e;
    if ((e = set_lds(painn_jvp_filter_kernel<NBK, true>, jvp_node_lds(NB, EV::COUNT))) != hipSuccess) return e;
    if ((e = set_lds(painn_jvp_edge_kernel<NBK, false>, jvp_edge_lds(NB))) != hipSuccess) return e;
    if ((e = set_lds(painn_jvp_edge_kernel<NBK, true>, jvp_edge_lds(NB))) != hipSuccess) return e;
    if ((e = set_lds(painn_jvp_node_kernel<NBK, false>, jvp_node_lds(NB, UV::COUNT))) != hipSuccess) return e;
    if ((e = set_lds(painn_jvp_node_kernel<NBK, true>, jvp_node_lds(NB, UV::COUNT))) != hipSuccess) return e;
    if ((e = set_lds(painn_jvp_update_kernel<NBK, false>, jvp_node_lds(NB, UV::COUNT))) != hipSuccess) return e;
    if ((e = set_lds(painn_jvp_update_kernel<NBK, true>, jvp_node_lds(NB, UV::COUNT))) != hipSuccess) return e;
    if ((e = set_lds(painn_jvp_readout_kernel<NBK, false>, jvp_node_lds(NB, RV::COUNT))) != hipSuccess) return e;
    if ((e = set_lds(painn_jvp_readout_kernel<NBK, true>, jvp_node_lds(NB, RV::COUNT))) != hipSuccess) return e;
    return hipSuccess;
}

hipError_t configure_painn_jvp_kernels(int NBv)
{
    TI_JVP_DISPATCH(NBv, return configure_jvp_nbk<NBK>(NBv));
    return hipSuccess;
}

hipError_t launch_jvp_filter(int NBv, bool split, const JvpFilterParams& p, hipStream_t st)
{
    const dim3 g((unsigned)((p.n_groups + 3) / 4));
    const size_t l = jvp_node_lds(NBv, EV::COUNT);
    TI_JVP_DISPATCH(NBv, {
        if (split) hipLaunchKernelGGL((painn_jvp_filter_kernel<NBK, true>), g, dim3(256), l, st, p);
        else hipLaunchKernelGGL((painn_jvp_filter_kernel<NBK, false>), g, dim3(256), l, st, p);
    });
    return hipGetLastError();
}

hipError_t launch_jvp_edge(int NBv, bool split, const JvpEdgeParams& p, hipStream_t st)
{
    const dim3 g((unsigned)((p.n_groups + 3) / 4));
    const size_t l = jvp_edge_lds(NBv);
    TI_JVP_DISPATCH(NBv, {
        if (split) hipLaunchKernelGGL((painn_jvp_edge_kernel<NBK, true>), g, dim3(256), l, st, p);
        else hipLaunchKernelGGL((painn_jvp_edge_kernel<NBK, false>), g, dim3(256), l, st, p);
    });
    return hipGetLastError();
}

hipError_t launch_jvp_node(int NBv, bool split, const JvpNodeParams& p, hipStream_t st)
{
    const dim3 g((unsigned)((p.N + 63) / 64));
    const size_t l = jvp_node_lds(NBv, UV::COUNT);
    TI_JVP_DISPATCH(NBv, {
        if (split) hipLaunchKernelGGL((painn_jvp_node_kernel<NBK, true>), g, dim3(256), l, st, p);
        else hipLaunchKernelGGL((painn_jvp_node_kernel<NBK, false>), g, dim3(256), l, st, p);
    });
    return hipGetLastError();
}

hipError_t launch_jvp_update(int NBv, bool split, const JvpUpdateParams& p, hipStream_t st)
{
    const dim3 g((unsigned)((p.N + 63) / 64));
    const size_t l = jvp_node_lds(NBv, UV::COUNT);
    TI_JVP_DISPATCH(NBv, {
        if (split) hipLaunchKernelGGL((painn_jvp_update_kernel<NBK, true>), g, dim3(256), l, st, p);
        else hipLaunchKernelGGL((painn_jvp_update_kernel<NBK, false>), g, dim3(256), l, st, p);
    });
    return hipGetLastError();
}

hipError_t launch_jvp_readout(int NBv, bool split, const JvpReadoutParams& p, hipStream_t st)
{
    const dim3 g((unsigned)((p.N + 63) / 64));
    const size_t l = jvp_node_lds(NBv, RV::COUNT);
    TI_JVP_DISPATCH(NBv, {
        if (split) hipLaunchKernelGGL((painn_jvp_readout_kernel<NBK, true>), g, dim3(256), l, st, p);
        else hipLaunchKernelGGL((painn_jvp_readout_kernel<NBK, false>), g, dim3(256), l, st, p);
    });
    return hipGetLastError();
}

hipError_t launch_div_reduce(const float* tout, long long B, int D, int G, float* div, hipStream_t st)
{
    hipLaunchKernelGGL(painn_div_reduce_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, st, tout, B, D, G, div);
    return hipGetLastError();
}

}  // namespace ti
