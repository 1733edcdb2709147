// painn_kernels.hip -- cPaiNN drift kernels for gfx950 (MI355X).
//
// Restates (never copies) the arithmetic of the reference modules; citations are relative to /root/reference:
//   embed   : InvariantFeatures/NominalEmbedding/TemperatureEncoder/PositionalEmbedding + CombineInvariantFeatures
//             mdqm9/thermo/ambient/models/embedding.py:68-86,127-160,200-212,249-261 (latent twins)
//   edge    : AddSpatialFeatures graph.py:25-33 + SE3Message.forward cpainn.py:263-310
//   update  : Update.forward cpainn.py:345-376, EquivariantLinear cpainn.py:403
//   readout : LayerReadout.forward cpainn.py:425-437, cPaiNN.forward cpainn.py:112-115
//
// Dense layers run on the matrix cores through mfma_chain.hpp (f32 32x32x2 / 16x16x4, or fp16 16x16x32 with split operands);
// activations never leave registers inside an MLP chain; per-atom sums over incoming edges are formed per 16-row block by a
// selection product and added to HBM accumulators with no-return atomics by the one wave that owns the molecule, in program
// order -- deterministic without an ordered reduction tree.
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include "mfma_chain.hpp"
#include "ti_internal.hpp"

namespace ti {

// the edge kernels live in painn_edge_nb{1,2,4,8}.hip (painn_edge_kernel.hpp)
hipError_t configure_edge_nb1(); hipError_t configure_edge_nb2(); hipError_t configure_edge_nb4(); hipError_t configure_edge_nb8();
hipError_t launch_edge_nb1(bool, bool, int, const EdgeParams&, hipStream_t); hipError_t launch_edge_nb2(bool, bool, int, const EdgeParams&, hipStream_t);
hipError_t launch_edge_nb4(bool, bool, int, const EdgeParams&, hipStream_t); hipError_t launch_edge_nb8(bool, bool, int, const EdgeParams&, hipStream_t);
hipError_t configure_pair_nb1(); hipError_t configure_pair_nb2(); hipError_t configure_pair_nb4();
hipError_t launch_pair_nb1(bool, bool, int, const EdgeParams&, hipStream_t); hipError_t launch_pair_nb2(bool, bool, int, const EdgeParams&, hipStream_t);
hipError_t launch_pair_nb4(bool, bool, int, const EdgeParams&, hipStream_t);

// ================================================================================================== update kernel
// v <- v + dv  with  dv = dvacc + cacc x v   (the cross product with v[dst] factors out of the edge sum),
// s <- s + dsacc, then Update.forward (cpainn.py:345-376); finally P for the next layer's message block.
// 16 atoms per wave on the r16 primitives (f32 or split-fp16 matrix path), 4 waves per workgroup, 2 workgroups per CU.
// The three spatial components share every visit of the U / V weight chunks.
struct UV {                                         // per-layer vector block in LDS, x F floats
    static constexpr int B0 = 0, G0 = 1, BE0 = 2, B1 = 3, G1 = 4, BE1 = 5, B2 = 6 /* 3F: gates | scale | add */, PB0 = 9, COUNT = 10;
};

// weight chunks per barrier.  4 (64 KB in flight per workgroup) was tried for the latency regime: no change (A = 9, B = 12: 59 vs 60 us),
// the stream of a lone workgroup runs at the ~12 B/clk a CU gets from beyond L2, not at the number of bytes in flight
// (the fp16 storage mode's hi-only chunks are half the size: four per barrier at F = 128, same LDS bytes)
__host__ __device__ constexpr int update_superchunk(int NB, bool H16) { return H16 && NB == 4 ? 4 : 2; }
__host__ __device__ constexpr int update_chunk4(int NB, bool H16) { return (H16 ? 128 : 256) * NB; }
template <int NBK, bool HAS_NEXT, int PREC>
__global__ __launch_bounds__(256, (NBK <= 8 ? 2 : 1)) void painn_update_kernel(const UpdateParams p)
{
    constexpr bool H16 = PREC == 2;                 // s, v, P are fp16 in HBM (the accumulators stay fp32), hi-only weight chunks
    constexpr int F = 16 * NBK, NB = (F + 31) / 32, WAVES = 4, T = 64 * WAVES, CH4 = update_chunk4(NB, H16);
    using A16 = r16::Act<NBK>;
    using OP = typename r16::OpSel<NBK, PREC>::type;
    extern __shared__ f32x4 lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), j = lane & 15, q = lane >> 4;
    constexpr int SC = update_superchunk(NB, H16);
    float* vec = reinterpret_cast<float*>(lds + 2 * SC * CH4);                     // [UV::COUNT][F]
    for (int i = threadIdx.x; i < UV::COUNT * F / 4; i += T)
        reinterpret_cast<f32x4*>(vec)[i] = reinterpret_cast<const f32x4*>(p.vecs)[i];
    PipeDMA<NB, T, SC, CH4> pipe;
    pipe.init(reinterpret_cast<const f32x4*>(p.stream), p.nch, lds, wave, lane);

    const long long node = ((long long)blockIdx.x * WAVES + wave) * 16 + j;
    const bool ok = node < p.N;
    const size_t nd = (size_t)(ok ? node : p.N - 1);
    const size_t vo = nd * 3 * F, so_ = nd * F;   // element offsets of this node's v / s rows (fp32 or fp16 storage)
    float* db = p.dvacc + nd * 3 * F;
    float* cb = p.cacc + nd * 3 * F;
    float* ab = p.dsacc + nd * F;                 // sum of the invariant messages of this layer (edge kernel)

    // ---- phase A: v_eff = v + dvacc + cacc x v (parked in dvacc), n2 = |V v_eff|^2 over the 3 components
    A16 n2;
    {
        OP ve[3];
        float vsc[3];                               // v is an un-normalised stream: per-row scales of the split operands
        {
            A16 t[3];                               // feature block outermost: v, dvacc and cacc are each read ONCE (all three components of a
#pragma unroll                                      // block are needed for the cross product; re-reading them per component missed L2)
            for (int nb = 0; nb < NBK; ++nb) {
                f32x4 vv[3], kk[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    vv[c] = p.first_layer ? f32x4{0, 0, 0, 0} : r16::load_state<H16>(p.v, vo + c * F, nb, q);
                    kk[c] = p.first_layer ? f32x4{0, 0, 0, 0} : r16::load_block(cb + c * F, nb, q);
                }
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int c1 = (c + 1) % 3, c2 = (c + 2) % 3;
                    const f32x4 dd = r16::load_block(db + c * F, nb, q);
                    t[c].b[nb] = (vv[c] + dd) + (kk[c1] * vv[c2] - kk[c2] * vv[c1]);      // torch.cross(edge_dir, v[dst]) summed over edges
                    if (ok) r16::store_block(db + c * F, nb, q, t[c].b[nb]);
                }
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) vsc[c] = ve[c].set_scaled(t[c]);
        }
#pragma unroll
        for (int nb = 0; nb < NBK; ++nb) n2.b[nb] = f32x4{0, 0, 0, 0};
#pragma unroll
        for (int ch = 0; ch < NB; ++ch) {
            const f32x4* wl = pipe.acquire();
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
                r16::gemm_bt(a0, a1, ve[c], wl, lane);                   // vv = V v  (of the scaled rows)
                a0 *= vsc[c]; a1 *= vsc[c];
                n2.b[2 * ch] += a0 * a0; n2.b[2 * ch + 1] += a1 * a1;
            }
            pipe.release();
        }
    }
    // ---- phase B: MLP([ |vv| , s ])
    A16 snew;                                       // s + ds, then s after the update (phase D's operand): read once, kept in registers
    OP h2;
    {
        A16 t;
#pragma unroll
        for (int nb = 0; nb < NBK; ++nb) t.b[nb] = r16::load_block(vec + UV::B0 * F, nb, q);
        {
            OP nn;
            float nsc;
            {
                A16 u;
#pragma unroll
                for (int nb = 0; nb < NBK; ++nb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) u.b[nb][r] = sqrtf(n2.b[nb][r]);
                nsc = nn.set_scaled(u);
            }
#pragma unroll
            for (int ch = 0; ch < NB; ++ch) {
                const f32x4* wl = pipe.acquire();
                f32x4 g0 = {0, 0, 0, 0}, g1 = {0, 0, 0, 0};
                r16::gemm_bt(g0, g1, nn, wl, lane);
                t.b[2 * ch] += g0 * nsc; t.b[2 * ch + 1] += g1 * nsc;
                pipe.release();
            }
        }
        {
            OP ss;
            float ssc;
#pragma unroll
            for (int nb = 0; nb < NBK; ++nb) snew.b[nb] = r16::load_state<H16>(p.s, so_, nb, q) + r16::load_block(ab, nb, q);      // s += ds
            ssc = ss.set_scaled(snew);
#pragma unroll
            for (int ch = 0; ch < NB; ++ch) {
                const f32x4* wl = pipe.acquire();
                f32x4 g0 = {0, 0, 0, 0}, g1 = {0, 0, 0, 0};
                r16::gemm_bt(g0, g1, ss, wl, lane);
                t.b[2 * ch] += g0 * ssc; t.b[2 * ch + 1] += g1 * ssc;
                pipe.release();
            }
        }
        r16::ln_silu(t, vec + UV::G0 * F, vec + UV::BE0 * F, q);
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
        r16::ln_silu(t, vec + UV::G1 * F, vec + UV::BE1 * F, q);
        h2.set(t);
    }
    // output chunks: [scale_squared_norm, add_invariant] per 32-feature block, then gates
#pragma unroll
    for (int ch = 0; ch < NB; ++ch) {
        const f32x4* wl = pipe.acquire();
        f32x4 q0 = r16::load_block(vec + (UV::B2 + 1) * F, 2 * ch, q), q1 = r16::load_block(vec + (UV::B2 + 1) * F, 2 * ch + 1, q);
        r16::gemm_bt(q0, q1, h2, wl, lane);
        pipe.release();
        wl = pipe.acquire();
        f32x4 a0 = r16::load_block(vec + (UV::B2 + 2) * F, 2 * ch, q), a1 = r16::load_block(vec + (UV::B2 + 2) * F, 2 * ch + 1, q);
        r16::gemm_bt(a0, a1, h2, wl, lane);
        pipe.release();
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int nb = 2 * ch + k;
            f32x4 so = snew.b[nb];
            const f32x4 qq = k ? q1 : q0, aa = k ? a1 : a0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float n = sqrtf(n2.b[nb][r]);
                so[r] = so[r] + ((n * n) * qq[r] + aa[r]);              // s += vv_norm**2 * scale + add
            }
            snew.b[nb] = so;
            if (ok) {
                r16::store_state<H16>(p.s, so_, nb, q, so);
                if (p.zero_acc) r16::store_block(ab, nb, q, f32x4{0, 0, 0, 0});
            }
        }
    }
    A16 gg;
#pragma unroll
    for (int ch = 0; ch < NB; ++ch) {
        const f32x4* wl = pipe.acquire();
        f32x4 a0 = r16::load_block(vec + UV::B2 * F, 2 * ch, q), a1 = r16::load_block(vec + UV::B2 * F, 2 * ch + 1, q);
        r16::gemm_bt(a0, a1, h2, wl, lane);
        gg.b[2 * ch] = a0; gg.b[2 * ch + 1] = a1;
        pipe.release();
    }
    // ---- phase C: v = v_eff + (U v_eff) * gates ; reset the accumulators for the next layer.  One spatial component at a time (the
    // stream holds U three times): the parked v_eff row is read once and serves as operand AND as the value the update is added to
    // -- with the three components sharing each chunk visit it had to be read a second time, and that read missed L2.
#pragma unroll 1
    for (int c = 0; c < 3; ++c) {
        A16 t;
        r16::load_set(t, db + c * F, q);
        OP vc;
        const float usc = vc.set_scaled(t);
#pragma unroll
        for (int ch = 0; ch < NB; ++ch) {
            const f32x4* wl = pipe.acquire();
            f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
            r16::gemm_bt(a0, a1, vc, wl, lane);
            a0 *= usc; a1 *= usc;
            if (ok) {
                r16::store_state<H16>(p.v, vo + c * F, 2 * ch, q, t.b[2 * ch] + a0 * gg.b[2 * ch]);
                r16::store_state<H16>(p.v, vo + c * F, 2 * ch + 1, q, t.b[2 * ch + 1] + a1 * gg.b[2 * ch + 1]);
                if (p.zero_acc) {
                    r16::store_block(db + c * F, 2 * ch, q, f32x4{0, 0, 0, 0});
                    r16::store_block(db + c * F, 2 * ch + 1, q, f32x4{0, 0, 0, 0});
                    r16::store_block(cb + c * F, 2 * ch, q, f32x4{0, 0, 0, 0});
                    r16::store_block(cb + c * F, 2 * ch + 1, q, f32x4{0, 0, 0, 0});
                }
            }
            pipe.release();
        }
    }
    // ---- phase D: P for the next message block
    if (HAS_NEXT) {
        OP sn;
        float psc;
        {
            psc = sn.set_scaled(snew);               // the rows written above, still in registers
        }
#pragma unroll
        for (int ch = 0; ch < NB; ++ch) {
            const f32x4* wl = pipe.acquire();
            f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
            r16::gemm_bt(a0, a1, sn, wl, lane);
            a0 = r16::load_block(vec + UV::PB0 * F, 2 * ch, q) + a0 * psc; a1 = r16::load_block(vec + UV::PB0 * F, 2 * ch + 1, q) + a1 * psc;
            pipe.release();
            if (ok) { r16::store_state<H16>(p.P, nd * F, 2 * ch, q, a0); r16::store_state<H16>(p.P, nd * F, 2 * ch + 1, q, a1); }
        }
    }
    pipe.drain();
}

// ================================================================================================== embed / readout kernels
// embed:   s = MLP([atom_emb | enc(T0) | enc(T1) | enc(t)]),  P = s @ phi0.W0[:, :F]^T + phi0.b0
// readout: gate(MLP(s)) * (Vr . v)
// 16 atoms per wave on the r16 primitives and the precision's matrix path (f32 16x16x4, split fp16, fp16).  (Until round 2 these two ran
// 32 rows per wave on the f32 32x32x2 MFMA whatever the precision: a lone wave spent 1.8 us per weight chunk there against 0.3 us here,
// which is what a small batch pays -- 80 + 35 us of a 620 us evaluation of 12 molecules.)
template <int NBK, int NSEG, int PREC>
__global__ __launch_bounds__(256, (NBK <= 8 ? 2 : 1)) void painn_embed16_kernel(const EmbedParams p)
{
    constexpr bool H16 = PREC == 2;
    constexpr int F = 16 * NBK, NB = (F + 31) / 32, WAVES = 4, T = 64 * WAVES, CH4 = update_chunk4(NB, H16);
    using A16 = r16::Act<NBK>;
    using OP = typename r16::OpSel<NBK, PREC>::type;
    extern __shared__ f32x4 lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), j = lane & 15, q = lane >> 4;
    PipeDMA<NB, T, 2, CH4> pipe;
    pipe.init(reinterpret_cast<const f32x4*>(p.stream), p.nch, lds, wave, lane);
    const long long node = ((long long)blockIdx.x * WAVES + wave) * 16 + j;
    const bool ok = node < p.N;
    const long long nd = ok ? node : p.N - 1;

    A16 acc;
#pragma unroll
    for (int nb = 0; nb < NBK; ++nb) acc.b[nb] = r16::load_block(p.mlp.b0, nb, q);
#pragma unroll
    for (int seg = 0; seg < NSEG; ++seg) {
        OP op;
        float sc = 1.0f;
        {
            A16 in;
            if (seg == 0) {
                r16::load_set(in, p.atom_emb + (size_t)p.atom_ids[nd % p.A] * F, q);
                sc = op.set_scaled(in);                          // an embedding table: any magnitude
            } else if (seg < NSEG - 1) {
                // TemperatureEncoder.forward: (T - mean(temps)) / (max - min), then PositionalEncoder(max_length=temp_length)
                float u = p.cond[nd * p.ncond + (seg - 1)] - p.temp_mean;
                u = u / p.temp_range;
                r16::posenc_set(in, u / p.temp_length, q);
                op.set(in);
            } else {
                r16::posenc_set(in, p.t / p.time_length, q);     // batch.t = t * ones_like(atoms)
                op.set(in);
            }
        }
        const float inv = r16::pow2_inverse(sc);
#pragma unroll
        for (int ch = 0; ch < NB; ++ch) {
            const f32x4* wl = pipe.acquire();
            f32x4 a0 = acc.b[2 * ch] * inv, a1 = acc.b[2 * ch + 1] * inv;     // exact: sc is a power of two
            r16::gemm_bt(a0, a1, op, wl, lane);
            acc.b[2 * ch] = a0 * sc; acc.b[2 * ch + 1] = a1 * sc;
            pipe.release();
        }
    }
    r16::ln_silu(acc, p.mlp.g0, p.mlp.be0, q);
    A16 t;
    {
        OP h1;
        h1.set(acc);
#pragma unroll
        for (int ch = 0; ch < NB; ++ch) {
            const f32x4* wl = pipe.acquire();
            f32x4 a0 = r16::load_block(p.mlp.b1, 2 * ch, q), a1 = r16::load_block(p.mlp.b1, 2 * ch + 1, q);
            r16::gemm_bt(a0, a1, h1, wl, lane);
            t.b[2 * ch] = a0; t.b[2 * ch + 1] = a1;
            pipe.release();
        }
    }
    r16::ln_silu(t, p.mlp.g1, p.mlp.be1, q);
    A16 sset;
    {
        OP h2;
        h2.set(t);
#pragma unroll
        for (int ch = 0; ch < NB; ++ch) {
            const f32x4* wl = pipe.acquire();
            f32x4 a0 = r16::load_block(p.mlp.b2, 2 * ch, q), a1 = r16::load_block(p.mlp.b2, 2 * ch + 1, q);
            r16::gemm_bt(a0, a1, h2, wl, lane);
            sset.b[2 * ch] = a0; sset.b[2 * ch + 1] = a1;
            pipe.release();
        }
    }
    if (ok) {
#pragma unroll
        for (int nb = 0; nb < NBK; ++nb) r16::store_state<H16>(p.s, (size_t)node * F, nb, q, sset.b[nb]);
    }
    {
        OP sn;
        const float psc = sn.set_scaled(sset), pinv = r16::pow2_inverse(psc);
#pragma unroll
        for (int ch = 0; ch < NB; ++ch) {
            const f32x4* wl = pipe.acquire();
            f32x4 a0 = r16::load_block(p.pb0, 2 * ch, q) * pinv, a1 = r16::load_block(p.pb0, 2 * ch + 1, q) * pinv;
            r16::gemm_bt(a0, a1, sn, wl, lane);
            pipe.release();
            if (ok) {
                r16::store_state<H16>(p.P, (size_t)node * F, 2 * ch, q, a0 * psc);
                r16::store_state<H16>(p.P, (size_t)node * F, 2 * ch + 1, q, a1 * psc);
            }
        }
    }
    pipe.drain();
}

template <int NBK, int PREC>
__global__ __launch_bounds__(256, (NBK <= 8 ? 2 : 1)) void painn_readout16_kernel(const ReadoutParams p)
{
    constexpr bool H16 = PREC == 2;
    constexpr int F = 16 * NBK, NB = (F + 31) / 32, WAVES = 4, T = 64 * WAVES, CH4 = update_chunk4(NB, H16);
    using A16 = r16::Act<NBK>;
    using OP = typename r16::OpSel<NBK, PREC>::type;
    extern __shared__ f32x4 lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), j = lane & 15, q = lane >> 4;
    PipeDMA<NB, T, 2, CH4> pipe;
    pipe.init(reinterpret_cast<const f32x4*>(p.stream), p.nch, lds, wave, lane);
    const long long node = ((long long)blockIdx.x * WAVES + wave) * 16 + j;
    const bool ok = node < p.N;
    const size_t nd = (size_t)(ok ? node : p.N - 1);

    A16 h1;
    {
        OP sop;
        float ssc;
        {
            A16 ss;
#pragma unroll
            for (int nb = 0; nb < NBK; ++nb) ss.b[nb] = r16::load_state<H16>(p.s, nd * F, nb, q);
            ssc = sop.set_scaled(ss);
        }
        const float sinv = r16::pow2_inverse(ssc);
#pragma unroll
        for (int ch = 0; ch < NB; ++ch) {
            const f32x4* wl = pipe.acquire();
            f32x4 a0 = r16::load_block(p.mlp.b0, 2 * ch, q) * sinv, a1 = r16::load_block(p.mlp.b0, 2 * ch + 1, q) * sinv;
            r16::gemm_bt(a0, a1, sop, wl, lane);
            h1.b[2 * ch] = a0 * ssc; h1.b[2 * ch + 1] = a1 * ssc;
            pipe.release();
        }
    }
    r16::ln_silu(h1, p.mlp.g0, p.mlp.be0, q);
    A16 h2;
    {
        OP o1;
        o1.set(h1);
#pragma unroll
        for (int ch = 0; ch < NB; ++ch) {
            const f32x4* wl = pipe.acquire();
            f32x4 a0 = r16::load_block(p.mlp.b1, 2 * ch, q), a1 = r16::load_block(p.mlp.b1, 2 * ch + 1, q);
            r16::gemm_bt(a0, a1, o1, wl, lane);
            h2.b[2 * ch] = a0; h2.b[2 * ch + 1] = a1;
            pipe.release();
        }
    }
    pipe.drain();
    r16::ln_silu(h2, p.mlp.g1, p.mlp.be1, q);
    // split(mlp(s), 1): [invariant_out (unused by cPaiNN.forward), gates]; the 1-row output layers are dot products over the features
    float gate = 0.f;
#pragma unroll
    for (int nb = 0; nb < NBK; ++nb) {
        const f32x4 w = r16::load_block(p.w2_gate, nb, q);
#pragma unroll
        for (int r = 0; r < 4; ++r) gate += h2.b[nb][r] * w[r];
    }
    gate = r16::xquarters(gate) + p.b2_gate;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float vv = 0.f;
#pragma unroll
        for (int nb = 0; nb < NBK; ++nb) {
            const f32x4 x = r16::load_state<H16>(p.v, (nd * 3 + c) * F, nb, q), w = r16::load_block(p.Vr, nb, q);
#pragma unroll
            for (int r = 0; r < 4; ++r) vv += x[r] * w[r];
        }
        vv = r16::xquarters(vv);
        if (ok && q == 0) p.out[node * 3 + c] = vv * gate;
    }
}

// ================================================================================================== launchers
template <typename K>
static hipError_t set_lds(K kernel, size_t bytes)
{
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

static size_t update_lds_bytes(int NB, bool h16) { return 2 * update_superchunk(NB, h16) * (size_t)update_chunk4(NB, h16) * 16 + 10 * (size_t)32 * NB * 4; }

template <int NB, int WAVES>
static hipError_t configure_nb()
{
    hipError_t e;
    {
        const size_t n0 = 2 * 2 * (size_t)update_chunk4(NB, false) * 16, n2 = 2 * 2 * (size_t)update_chunk4(NB, true) * 16;
        if ((e = set_lds(painn_embed16_kernel<2 * NB, 2, 0>, n0)) != hipSuccess) return e;
        if ((e = set_lds(painn_embed16_kernel<2 * NB, 3, 0>, n0)) != hipSuccess) return e;
        if ((e = set_lds(painn_embed16_kernel<2 * NB, 4, 0>, n0)) != hipSuccess) return e;
        if ((e = set_lds(painn_embed16_kernel<2 * NB, 2, 1>, n0)) != hipSuccess) return e;
        if ((e = set_lds(painn_embed16_kernel<2 * NB, 3, 1>, n0)) != hipSuccess) return e;
        if ((e = set_lds(painn_embed16_kernel<2 * NB, 4, 1>, n0)) != hipSuccess) return e;
        if ((e = set_lds(painn_embed16_kernel<2 * NB, 2, 2>, n2)) != hipSuccess) return e;
        if ((e = set_lds(painn_embed16_kernel<2 * NB, 3, 2>, n2)) != hipSuccess) return e;
        if ((e = set_lds(painn_embed16_kernel<2 * NB, 4, 2>, n2)) != hipSuccess) return e;
        if ((e = set_lds(painn_readout16_kernel<2 * NB, 0>, n0)) != hipSuccess) return e;
        if ((e = set_lds(painn_readout16_kernel<2 * NB, 1>, n0)) != hipSuccess) return e;
        if ((e = set_lds(painn_readout16_kernel<2 * NB, 2>, n2)) != hipSuccess) return e;
    }
    if ((e = (NB == 1 ? configure_edge_nb1() : NB == 2 ? configure_edge_nb2() : NB == 4 ? configure_edge_nb4() : configure_edge_nb8())) != hipSuccess) return e;
    if ((e = (NB == 1 ? configure_pair_nb1() : NB == 2 ? configure_pair_nb2() : NB == 4 ? configure_pair_nb4() : hipSuccess)) != hipSuccess) return e;
    const size_t bu = update_lds_bytes(NB, false), bh = update_lds_bytes(NB, true);
    if ((e = set_lds(painn_update_kernel<2 * NB, true, 0>, bu)) != hipSuccess) return e;
    if ((e = set_lds(painn_update_kernel<2 * NB, false, 0>, bu)) != hipSuccess) return e;
    if ((e = set_lds(painn_update_kernel<2 * NB, true, 1>, bu)) != hipSuccess) return e;
    if ((e = set_lds(painn_update_kernel<2 * NB, false, 1>, bu)) != hipSuccess) return e;
    if ((e = set_lds(painn_update_kernel<2 * NB, true, 2>, bh)) != hipSuccess) return e;
    if ((e = set_lds(painn_update_kernel<2 * NB, false, 2>, bh)) != hipSuccess) return e;

    return hipSuccess;
}

// NB <= 2: 8 waves / workgroup, 2 per SIMD (<= 256 VGPRs); NB >= 4: 4 waves, 1 per SIMD (512-register budget:
// three resident activation sets of 16*NB registers plus the compiler's address/staging overhead do not fit in 256)
#ifdef TI_DEV_NB4_ONLY
#define TI_SMALL_NB(...)
#else
#define TI_SMALL_NB(...) __VA_ARGS__
#endif
#define TI_DISPATCH_NB(NBv, ...) \
    switch (NBv) {                                                            \
        TI_SMALL_NB(case 1: { constexpr int NB = 1, WAVES = 8; __VA_ARGS__; } break;)             \
        TI_SMALL_NB(case 2: { constexpr int NB = 2, WAVES = 8; __VA_ARGS__; } break;)             \
        case 4: { constexpr int NB = 4, WAVES = 4; __VA_ARGS__; } break;             \
        TI_SMALL_NB(case 8: { constexpr int NB = 8, WAVES = 4; __VA_ARGS__; } break;)             \
        default: return hipErrorInvalidValue;                                 \
    }

hipError_t configure_painn_kernels(int NBv)
{
    TI_DISPATCH_NB(NBv, return (configure_nb<NB, WAVES>()));
    return hipSuccess;
}


// prec: include/ti_hip.h TI_PREC_* (0 f32, 1 f16x2, 2 f16 storage mode: the state tensors are fp16)
static size_t node16_lds_bytes(int NB, bool h16) { return 2 * 2 * (size_t)update_chunk4(NB, h16) * 16; }      // PipeDMA<.., SC = 2>: two superchunks

template <int NB, int PREC>
static void launch_embed16(int nseg, const EmbedParams& p, hipStream_t st)
{
    const dim3 g((unsigned)((p.N + 63) / 64));                 // 4 waves x 16 atoms per workgroup
    const size_t l = node16_lds_bytes(NB, PREC == 2);
    if (nseg == 4) hipLaunchKernelGGL((painn_embed16_kernel<2 * NB, 4, PREC>), g, dim3(256), l, st, p);
    else if (nseg == 3) hipLaunchKernelGGL((painn_embed16_kernel<2 * NB, 3, PREC>), g, dim3(256), l, st, p);
    else hipLaunchKernelGGL((painn_embed16_kernel<2 * NB, 2, PREC>), g, dim3(256), l, st, p);
}

hipError_t launch_embed(int NBv, int nseg, int prec, const EmbedParams& p, hipStream_t st)
{
    TI_DISPATCH_NB(NBv, {
        (void)WAVES;
        if (prec == 2) launch_embed16<NB, 2>(nseg, p, st);
        else if (prec == 1) launch_embed16<NB, 1>(nseg, p, st);
        else launch_embed16<NB, 0>(nseg, p, st);
    });
    return hipGetLastError();
}

hipError_t launch_edge(int NBv, bool first, bool last, int prec, const EdgeParams& p, hipStream_t st)
{
    switch (NBv) {
        case 1: return launch_edge_nb1(first, last, prec, p, st);
        case 2: return launch_edge_nb2(first, last, prec, p, st);
        case 4: return launch_edge_nb4(first, last, prec, p, st);
        case 8: return launch_edge_nb8(first, last, prec, p, st);
        default: return hipErrorInvalidValue;
    }
}

// ---- pair-major message kernel: per-atom sums of the partial rows (painn_pair_kernel.hpp writes one row of 7 F floats per block and
// slot).  One thread per float4 of a node's row; the partial rows of an atom are added in walk order: fixed order, no atomics.
__global__ __launch_bounds__(256) void pair_reduce_kernel(const PairReduceParams p)
{
    const int per = 7 * p.F / 4, F4 = p.F / 4;                  // float4 per partial row: ds [0, F4), dv [F4, 4 F4), c [4 F4, 7 F4)
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long node = t / per;
    const int f4 = (int)(t - node * per);
    if (node >= p.B * p.A || (!p.has_c && f4 >= 4 * F4)) return;
    const long long mol = node / p.A;
    const int atom = (int)(node - mol * p.A);
    const long long g = mol / p.G;
    const int32_t* pl = p.plist + ((size_t)(mol - g * p.G) * p.A + atom) * p.kmax;
    const f32x4* base = reinterpret_cast<const f32x4*>(p.part) + (size_t)g * p.nblk * 8 * per + f4;
    f32x4 s = {0, 0, 0, 0};
    for (int k = 0; k < p.kmax; ++k) {
        const int id = pl[k];
        if (id < 0) break;
        s += base[(size_t)id * per];
    }
    if (f4 < F4) reinterpret_cast<f32x4*>(p.dsacc)[(size_t)node * F4 + f4] = s;
    else if (f4 < 4 * F4) reinterpret_cast<f32x4*>(p.dvacc)[(size_t)node * 3 * F4 + (f4 - F4)] = s;
    else reinterpret_cast<f32x4*>(p.cacc)[(size_t)node * 3 * F4 + (f4 - 4 * F4)] = s;
}
hipError_t launch_pair_reduce(const PairReduceParams& p, hipStream_t st)
{
    const long long threads = p.B * p.A * (7 * p.F / 4);
    hipLaunchKernelGGL(pair_reduce_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, p);
    return hipGetLastError();
}

hipError_t launch_pair(int NBv, bool first, bool last, int prec, const EdgeParams& p, hipStream_t st)
{
    switch (NBv) {
        case 1: return launch_pair_nb1(first, last, prec, p, st);
        case 2: return launch_pair_nb2(first, last, prec, p, st);
        case 4: return launch_pair_nb4(first, last, prec, p, st);
        default: return hipErrorInvalidValue;
    }
}
bool pair_kernel_exists(int NB, int prec) { return NB <= 4 && (prec == TI_PREC_F32 || prec == TI_PREC_F16X2); }

hipError_t launch_update(int NBv, bool has_next, int prec, const UpdateParams& p, hipStream_t st)
{
    TI_DISPATCH_NB(NBv, {
        (void)WAVES;
        const dim3 g((unsigned)((p.N + 63) / 64));                 // 4 waves x 16 atoms per workgroup
        const size_t l = update_lds_bytes(NB, prec == 2);
        if (prec == 2) {
            if (has_next) hipLaunchKernelGGL((painn_update_kernel<2 * NB, true, 2>), g, dim3(256), l, st, p);
            else hipLaunchKernelGGL((painn_update_kernel<2 * NB, false, 2>), g, dim3(256), l, st, p);
        } else if (prec == 1) {
            if (has_next) hipLaunchKernelGGL((painn_update_kernel<2 * NB, true, 1>), g, dim3(256), l, st, p);
            else hipLaunchKernelGGL((painn_update_kernel<2 * NB, false, 1>), g, dim3(256), l, st, p);
        } else {
            if (has_next) hipLaunchKernelGGL((painn_update_kernel<2 * NB, true, 0>), g, dim3(256), l, st, p);
            else hipLaunchKernelGGL((painn_update_kernel<2 * NB, false, 0>), g, dim3(256), l, st, p);
        }
    });
    return hipGetLastError();
}

hipError_t launch_readout(int NBv, int prec, const ReadoutParams& p, hipStream_t st)
{
    TI_DISPATCH_NB(NBv, {
        (void)WAVES;
        const dim3 g((unsigned)((p.N + 63) / 64));
        const size_t l = node16_lds_bytes(NB, prec == 2);
        if (prec == 2) hipLaunchKernelGGL((painn_readout16_kernel<2 * NB, 2>), g, dim3(256), l, st, p);
        else if (prec == 1) hipLaunchKernelGGL((painn_readout16_kernel<2 * NB, 1>), g, dim3(256), l, st, p);
        else hipLaunchKernelGGL((painn_readout16_kernel<2 * NB, 0>), g, dim3(256), l, st, p);
    });
    return hipGetLastError();
}

}  // namespace ti
