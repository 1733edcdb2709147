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

template <int NB, int WAVES>
struct Cfg {
    static constexpr int F = 32 * NB;
    static constexpr int T = 64 * WAVES;
    static constexpr int CH4 = 256 * NB;
    static constexpr size_t lds_bytes = 2 * (size_t)CH4 * 16;     // node kernels: the two weight chunk buffers
};

// ================================================================================================== embed kernel
// s = MLP([atom_emb | enc(T0) | enc(T1) | enc(t)]),  P = s @ phi0.W0[:, :F]^T + phi0.b0
template <int NB, int WAVES, int NSEG>
__global__ __launch_bounds__(64 * WAVES, WAVES / 4) void painn_embed_kernel(const EmbedParams p)
{
    using C = Cfg<NB, WAVES>;
    constexpr int F = C::F;
    extern __shared__ f32x4 lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), j = lane & 31, h = lane >> 5;
    Pipe<NB, C::T> pipe;
    pipe.init(reinterpret_cast<const f32x4*>(p.stream), p.nch, lds);

    const long long node = ((long long)blockIdx.x * WAVES + wave) * 32 + j;
    const bool ok = node < p.N;
    const long long nd = ok ? node : p.N - 1;

    Act<NB> acc;
#pragma unroll
    for (int nbo = 0; nbo < NB; ++nbo) acc.b[nbo] = load_block(p.mlp.b0, nbo, h);
#pragma unroll
    for (int seg = 0; seg < NSEG; ++seg) {
        Act<NB> in;
        if (seg == 0) {
            load_set(in, p.atom_emb + (size_t)p.atom_ids[nd % p.A] * F, h);
        } else if (seg < NSEG - 1) {
            // TemperatureEncoder.forward: (T - mean(temps)) / (max - min), then PositionalEncoder(max_length=temp_length)
            float u = p.cond[nd * p.ncond + (seg - 1)] - p.temp_mean;
            u = u / p.temp_range;
            posenc_set(in, u / p.temp_length, h);
        } else {
            posenc_set(in, p.t / p.time_length, h);      // batch.t = t * ones_like(atoms)
        }
#pragma unroll
        for (int nbo = 0; nbo < NB; ++nbo) {
            const f32x4* wl = pipe.begin();
            gemm_bt(acc.b[nbo], in, wl, lane);
            pipe.end();
        }
    }
    ln_silu(acc, p.mlp.g0, p.mlp.be0, h);
    Act<NB> h2;
#pragma unroll
    for (int nbo = 0; nbo < NB; ++nbo) {
        const f32x4* wl = pipe.begin();
        f32x16 a = load_block(p.mlp.b1, nbo, h);
        gemm_bt(a, acc, wl, lane);
        h2.b[nbo] = a;
        pipe.end();
    }
    ln_silu(h2, p.mlp.g1, p.mlp.be1, h);
    Act<NB> sset;
#pragma unroll
    for (int nbo = 0; nbo < NB; ++nbo) {
        const f32x4* wl = pipe.begin();
        f32x16 a = load_block(p.mlp.b2, nbo, h);
        gemm_bt(a, h2, wl, lane);
        sset.b[nbo] = a;
        pipe.end();
    }
    if (ok) store_set(p.s + (size_t)node * F, h, sset);
#pragma unroll
    for (int nbo = 0; nbo < NB; ++nbo) {
        const f32x4* wl = pipe.begin();
        f32x16 a = load_block(p.pb0, nbo, h);
        gemm_bt(a, sset, wl, lane);
        pipe.end();
        if (ok) store_block(p.P + (size_t)node * F, nbo, h, a);
    }
}

// ================================================================================================== edge kernel
// Per-layer vectors of the two message MLPs, copied once per workgroup into LDS (offsets in floats, F = n_features).
// Global loads of bias/gamma/beta in front of every weight chunk exposed a full memory latency 56 times per row block.
struct EV {
    static constexpr int W_B0 = 0, W_G0 = 1, W_BE0 = 2, W_B1 = 3, W_G1 = 4, W_BE1 = 5, P_G0 = 6, P_BE0 = 7, P_B1 = 8, P_G1 = 9,
                         P_BE1 = 10, P_B2 = 11, W_B2 = 16, COUNT = 21;      // x F
};

// Fire-and-forget fp32 add (global_atomic_add_f32, no return): nothing waits for the memory round trip.  Every
// accumulator element starts at zero and is only ever added to by the one wave that owns the molecule, in program
// order (an atom's <= 31 incoming edges span at most three 16-row blocks of that wave).
#ifdef TI_ABL_NOATOMIC          // ablation build (timing only, wrong results): keep the value alive, drop the memory op
__device__ __forceinline__ void add_noret(float* p, float v) { asm volatile("" ::"v"(v), "v"(p)); }
#else
__device__ __forceinline__ void add_noret(float* p, float v) { unsafeAtomicAdd(p, v); }
#endif

// One wave = one molecule group, walked in blocks of 16 edge rows on the 16x16x4 MFMA (mfma_chain.hpp, namespace r16).
// The waves of a workgroup share the weight-chunk stream (4 or 8 of them, see below); two waves per SIMD (F <= 128) hide each other's
// LayerNorm / reduction / wait phases behind matrix work.
// SPLIT selects the split-fp16 matrix path (mfma_chain.hpp: Opnd<NBK, true>) instead of the f32 MFMA.
#ifndef TI_EDGE_OCC
#define TI_EDGE_OCC 2
#endif
// Workgroup width.  4 waves (two workgroups per CU) is the default; for large split-fp16 launches (>= 2048 groups, F <= 128)
// the launcher picks the 8-wave build: one 512-thread workgroup per CU shares one weight stream (half the LDS-DMA writes) and, at F = 128,
// the freed LDS holds 4-chunk superchunks (half the barriers; needs chunk counts % 4 == 0: 56/48/40/32) -- 1.8 % on the edge
// kernel at B >= 32k.  Small launches keep 4 waves: fatter workgroups cost the latency regime 20-60 %.  F = 256 needs the
// 512-register budget of one wave per SIMD and is always 4 waves.
__host__ __device__ constexpr int edge_superchunk(int NB, int WAVES) { return (WAVES == 8 && NB == 4) ? 4 : 2; }
template <int NBK, bool FIRST, bool LAST, bool SPLIT, int WAVES, int NS>
__global__ __launch_bounds__(64 * WAVES, (NBK <= 8 ? TI_EDGE_OCC * 4 / WAVES : 1)) void painn_edge_kernel(const EdgeParams p)
{
    constexpr int F = 16 * NBK, NB = (F + 31) / 32, T = 64 * WAVES, CH4 = 256 * NB;
    using A16 = r16::Act<NBK>;
    using OP = r16::Opnd<NBK, SPLIT>;
    extern __shared__ f32x4 lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), j = lane & 15, q = lane >> 4;
    constexpr int SC = edge_superchunk(NB, WAVES);                                  // weight chunks per barrier
    float* scratch = reinterpret_cast<float*>(lds + 2 * SC * CH4) + wave * 64;     // [16 rows][4] edge_dir of the block
    float* vec = reinterpret_cast<float*>(lds + 2 * SC * CH4) + WAVES * 64;        // [EV::COUNT][F]
    for (int i = threadIdx.x; i < EV::COUNT * F / 4; i += T)
        reinterpret_cast<f32x4*>(vec)[i] = reinterpret_cast<const f32x4*>(p.vecs)[i];
    PipeDMA<NB, T, SC> pipe;                                                     // weights staged SC chunks per barrier
    pipe.init(reinterpret_cast<const f32x4*>(p.stream), p.nch, lds, wave, lane);   // barrier inside: vec is visible after it

    const long long gi_raw = (long long)blockIdx.x * WAVES + wave;
    const bool group_ok = gi_raw < p.n_groups;
    const long long gi = group_ok ? gi_raw : p.n_groups - 1;
    // a group is P "parts" (ranges of destination atoms, each with its own row blocks and its own wave): ti_internal.hpp
    const long long mg = gi / p.parts;
    const uint32_t* rows = p.rows + (size_t)(gi - mg * p.parts) * p.nblk * 16;
    const int32_t* slotnode = p.slotnode + (size_t)(gi - mg * p.parts) * p.nblk * 16;

    for (int blk = 0; blk < p.nblk; ++blk) {
        // ---- K1 geometry of this lane's row (the 4 quarters compute the same row)
        const uint32_t meta = rows[blk * 16 + j];
        long long mol = mg * p.G + row_mol(meta);
        mol = mol < p.B ? mol : p.B - 1;
        const long long nsrc = mol * p.A + row_src(meta), ndst = mol * p.A + row_dst(meta);
        const size_t erow0 = ((size_t)gi * p.nblk + blk) * 16;
        float dist;
        {
            const float rx = p.x[nsrc * 3 + 0] - p.x[ndst * 3 + 0];
            const float ry = p.x[nsrc * 3 + 1] - p.x[ndst * 3 + 1];
            const float rz = p.x[nsrc * 3 + 2] - p.x[ndst * 3 + 2];
            dist = sqrtf(rx * rx + ry * ry + rz * rz);
            const float den = 1.0f + dist;                       // edge_dir = r / (1 + d)   (not a unit vector)
            if (q == 0) {
                f32x4 dd = {rx / den, ry / den, rz / den, 0.f};
                *reinterpret_cast<f32x4*>(scratch + j * 4) = dd;
            }
        }
        // ---- w(enc(d)) hidden layers
        OP g2;
        {
            OP g1;
            A16 t1;
            {
                OP enc;
                {
                    A16 t;
                    r16::posenc_set(t, dist / p.length_scale, q);
                    enc.set(t);
                }
#pragma unroll
                for (int c = 0; c < NB; ++c) {
                    const f32x4* wl = pipe.acquire();
                    f32x4 a0 = r16::load_block(vec + EV::W_B0 * F, 2 * c, q), a1 = r16::load_block(vec + EV::W_B0 * F, 2 * c + 1, q);
                    r16::gemm_bt(a0, a1, enc, wl, lane);
                    t1.b[2 * c] = a0; t1.b[2 * c + 1] = a1;
                    pipe.release();
                }
            }
            r16::ln_silu(t1, vec + EV::W_G0 * F, vec + EV::W_BE0 * F, q);
            g1.set(t1);
#pragma unroll
            for (int c = 0; c < NB; ++c) {
                const f32x4* wl = pipe.acquire();
                f32x4 a0 = r16::load_block(vec + EV::W_B1 * F, 2 * c, q), a1 = r16::load_block(vec + EV::W_B1 * F, 2 * c + 1, q);
                r16::gemm_bt(a0, a1, g1, wl, lane);
                t1.b[2 * c] = a0; t1.b[2 * c + 1] = a1;
                pipe.release();
            }
            r16::ln_silu(t1, vec + EV::W_G1 * F, vec + EV::W_BE1 * F, q);
            g2.set(t1);
        }
        // ---- phi([s[src] | e]) hidden layers; the s[src] half of the first Linear is P[src] (node kernels)
        OP h2;
        {
            OP h1, ein;
            A16 t1;
            if (FIRST) r16::load_set(t1, p.edge_emb + row_type(meta) * F, q);
            else       r16::load_set(t1, p.e + (erow0 + j) * F, q);
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
            r16::ln_silu(t1, vec + EV::P_G0 * F, vec + EV::P_BE0 * F, q);
            h1.set(t1);
#pragma unroll
            for (int c = 0; c < NB; ++c) {
                const f32x4* wl = pipe.acquire();
                f32x4 a0 = r16::load_block(vec + EV::P_B1 * F, 2 * c, q), a1 = r16::load_block(vec + EV::P_B1 * F, 2 * c + 1, q);
                r16::gemm_bt(a0, a1, h1, wl, lane);
                t1.b[2 * c] = a0; t1.b[2 * c + 1] = a1;
                pipe.release();
            }
            r16::ln_silu(t1, vec + EV::P_G1 * F, vec + EV::P_BE1 * F, q);
            h2.set(t1);
        }
        // ---- output layer, flipped: features on lanes (l & 15), the block's rows 4q + r in registers
        uint32_t mi[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) mi[r] = rows[blk * 16 + 4 * q + r];
        // Per-atom sums over the block's rows, in registers (r16::QuarterSum, mfma_chain.hpp): the template builder gives a block
        // at most 4 destination atoms ("slots", NS = 2 when no block of the template has more than 2); msk[k][r] selects the rows
        // 4q + r of slot k, and after the lane-row exchanges quarter q of the wave holds the sum of ONE slot, so one atomic
        // instruction carries the sums of every slot of the block.  qnode is the atom this quarter adds to (or -1).
        r16::QuarterSum<NS> qs;
#pragma unroll
        for (int r = 0; r < 4; ++r) qs.set_row(r, row_slot(mi[r]));
        int qnode;
        {
            const int sn = slotnode[blk * 16 + r16::QuarterSum<NS>::slot_of_quarter(q)];
            const long long m2 = mg * p.G + (sn >> 8);
            qnode = (sn >= 0 && group_ok && m2 < p.B) ? (int)(m2 * p.A + (sn & 255)) : -1;
        }

        // (phi_c + b) * (w_c + b) for output chunk c (0 gates, 1 scale_edge_dir, 2 ds, 3 de, 4 cross gates), 32 features
        // fo .. fo+31 as two 16-feature blocks
        auto out_pair = [&](int c, int nbo, f32x4& r0, f32x4& r1) {
            f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0}, b0 = {0, 0, 0, 0}, b1 = {0, 0, 0, 0};
            const f32x4* wl0 = pipe.acquire();
            r16::gemm_fl(a0, a1, h2, wl0, lane);
            pipe.release();
            const f32x4* wl1 = pipe.acquire();
            r16::gemm_fl(b0, b1, g2, wl1, lane);
            pipe.release();
            const float* bp = vec + (EV::P_B2 + c) * F + 32 * nbo + j;
            const float* bw = vec + (EV::W_B2 + c) * F + 32 * nbo + j;
            r0 = (a0 + bp[0]) * (b0 + bw[0]);
            r1 = (a1 + bp[16]) * (b1 + bw[16]);
        };
        // add the per-slot sums of (v0 | v1) into dst[node*stride + {0,16}] (dst already offset to component / feature)
        auto emit = [&](const f32x4& v0, const f32x4& v1, float* dst, size_t stride) {
            if (NS == 2) {
                const float z = qs.sum_pair(v0, v1);                 // quarter q: slot q & 1 of (q >> 1 ? v1 : v0)
                if (qnode >= 0) add_noret(dst + (size_t)qnode * stride + 16 * (q >> 1), z);
            } else {
                const float z0 = qs.sum(v0), z1 = qs.sum(v1);        // quarter q: slot q
                if (qnode >= 0) { float* d = dst + (size_t)qnode * stride; add_noret(d, z0); add_noret(d + 16, z1); }
            }
        };

#pragma unroll 1
        for (int nbo = 0; nbo < NB; ++nbo) {
            const int fo = 32 * nbo + j;
            {   // ds: invariant message, summed over incoming edges
                f32x4 v0, v1;
                out_pair(2, nbo, v0, v1);
                emit(v0, v1, p.dsacc + fo, F);
            }
            if (!LAST) {   // de: edge state update  e += de
                f32x4 v0, v1;
                out_pair(3, nbo, v0, v1);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float* ep = p.e + (erow0 + 4 * q + r) * F + fo;
                    if (group_ok) {
                        if (FIRST) {
                            const float* em = p.edge_emb + row_type(mi[r]) * F + fo;
                            ep[0] = em[0] + v0[r]; ep[16] = em[16] + v1[r];
                        } else { add_noret(ep, v0[r]); add_noret(ep + 16, v1[r]); }
                    }
                }
            }
            {   // equivariant message: sum_e (sed * dir_e + gates * v[src_e]) -> dvacc ; sum_e cg * dir_e -> cacc
                f32x4 sed0, sed1, gt0 = {0, 0, 0, 0}, gt1 = {0, 0, 0, 0};
                out_pair(1, nbo, sed0, sed1);
                // v[src] of the block's rows for the gated term: issued here, consumed after the two gate chunks
                f32x4 vs[3][2];
                if (!FIRST) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        long long m2 = mg * p.G + row_mol(mi[r]);
                        m2 = m2 < p.B ? m2 : p.B - 1;
                        const float* vp = p.v + (size_t)(m2 * p.A + row_src(mi[r])) * 3 * F + fo;
#pragma unroll
#ifdef TI_ABL_NOGATHER          // ablation build (timing only, wrong results): no v[src] gather
                        for (int c = 0; c < 3; ++c) { vs[c][0][r] = 0.5f; vs[c][1][r] = 0.25f; (void)vp; }
#else
                        for (int c = 0; c < 3; ++c) { vs[c][0][r] = vp[c * F]; vs[c][1][r] = vp[c * F + 16]; }
#endif
                    }
                    out_pair(0, nbo, gt0, gt1);
                }
                f32x4 dir[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) dir[r] = *reinterpret_cast<const f32x4*>(scratch + (4 * q + r) * 4);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    f32x4 v0, v1;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v0[r] = sed0[r] * dir[r][c]; v1[r] = sed1[r] * dir[r][c];
                        if (!FIRST) { v0[r] = fmaf(gt0[r], vs[c][0][r], v0[r]); v1[r] = fmaf(gt1[r], vs[c][1][r], v1[r]); }
                    }
                    emit(v0, v1, p.dvacc + c * F + fo, 3 * F);
                }
                if (!FIRST) {
                    f32x4 cg0, cg1;
                    out_pair(4, nbo, cg0, cg1);
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        f32x4 v0, v1;
#pragma unroll
                        for (int r = 0; r < 4; ++r) { v0[r] = cg0[r] * dir[r][c]; v1[r] = cg1[r] * dir[r][c]; }
                        emit(v0, v1, p.cacc + c * F + fo, 3 * F);
                    }
                }
            }
        }
    }
    pipe.drain();
}

// ================================================================================================== update kernel
// v <- v + dv  with  dv = dvacc + cacc x v   (the cross product with v[dst] factors out of the edge sum),
// s <- s + dsacc, then Update.forward (cpainn.py:345-376); finally P for the next layer's message block.
// 16 atoms per wave on the r16 primitives (f32 or split-fp16 matrix path), 4 waves per workgroup, 2 workgroups per CU.
// The three spatial components share every visit of the U / V weight chunks.
struct UV {                                         // per-layer vector block in LDS, x F floats
    static constexpr int B0 = 0, G0 = 1, BE0 = 2, B1 = 3, G1 = 4, BE1 = 5, B2 = 6 /* 3F: gates | scale | add */, PB0 = 9, COUNT = 10;
};

template <int NBK, bool HAS_NEXT, bool SPLIT>
__global__ __launch_bounds__(256, (NBK <= 8 ? 2 : 1)) void painn_update_kernel(const UpdateParams p)
{
    constexpr int F = 16 * NBK, NB = (F + 31) / 32, WAVES = 4, T = 64 * WAVES, CH4 = 256 * NB;
    using A16 = r16::Act<NBK>;
    using OP = r16::Opnd<NBK, SPLIT>;
    extern __shared__ f32x4 lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), j = lane & 15, q = lane >> 4;
    float* vec = reinterpret_cast<float*>(lds + 4 * CH4);                          // [UV::COUNT][F]
    for (int i = threadIdx.x; i < UV::COUNT * F / 4; i += T)
        reinterpret_cast<f32x4*>(vec)[i] = reinterpret_cast<const f32x4*>(p.vecs)[i];
    PipeDMA<NB, T, 2> pipe;
    pipe.init(reinterpret_cast<const f32x4*>(p.stream), p.nch, lds, wave, lane);

    const long long node = ((long long)blockIdx.x * WAVES + wave) * 16 + j;
    const bool ok = node < p.N;
    const size_t nd = (size_t)(ok ? node : p.N - 1);
    float* vb = p.v + nd * 3 * F;
    float* db = p.dvacc + nd * 3 * F;
    float* cb = p.cacc + nd * 3 * F;
    float* sb = p.s + nd * F;
    float* ab = p.dsacc + nd * F;                 // sum of the invariant messages of this layer (edge kernel)

    // ---- phase A: v_eff = v + dvacc + cacc x v (parked in dvacc), n2 = |V v_eff|^2 over the 3 components
    A16 n2;
    {
        OP ve[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int c1 = (c + 1) % 3, c2 = (c + 2) % 3;
            A16 t;
#pragma unroll
            for (int nb = 0; nb < NBK; ++nb) {
                const f32x4 vc = r16::load_block(vb + c * F, nb, q), dd = r16::load_block(db + c * F, nb, q);
                const f32x4 v1 = r16::load_block(vb + c1 * F, nb, q), v2 = r16::load_block(vb + c2 * F, nb, q);
                const f32x4 k1 = r16::load_block(cb + c1 * F, nb, q), k2 = r16::load_block(cb + c2 * F, nb, q);
                t.b[nb] = (vc + dd) + (k1 * v2 - k2 * v1);              // torch.cross(edge_dir, v[dst]) summed over edges
                if (ok) r16::store_block(db + c * F, nb, q, t.b[nb]);
            }
            ve[c].set(t);
        }
#pragma unroll
        for (int nb = 0; nb < NBK; ++nb) n2.b[nb] = f32x4{0, 0, 0, 0};
#pragma unroll
        for (int ch = 0; ch < NB; ++ch) {
            const f32x4* wl = pipe.acquire();
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
                r16::gemm_bt(a0, a1, ve[c], wl, lane);                   // vv = V v
                n2.b[2 * ch] += a0 * a0; n2.b[2 * ch + 1] += a1 * a1;
            }
            pipe.release();
        }
    }
    // ---- phase B: MLP([ |vv| , s ])
    OP h2;
    {
        A16 t;
#pragma unroll
        for (int nb = 0; nb < NBK; ++nb) t.b[nb] = r16::load_block(vec + UV::B0 * F, nb, q);
        {
            OP nn;
            {
                A16 u;
#pragma unroll
                for (int nb = 0; nb < NBK; ++nb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) u.b[nb][r] = sqrtf(n2.b[nb][r]);
                nn.set(u);
            }
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
                A16 u;
#pragma unroll
                for (int nb = 0; nb < NBK; ++nb) u.b[nb] = r16::load_block(sb, nb, q) + r16::load_block(ab, nb, q);      // s += ds
                ss.set(u);
            }
#pragma unroll
            for (int ch = 0; ch < NB; ++ch) {
                const f32x4* wl = pipe.acquire();
                r16::gemm_bt(t.b[2 * ch], t.b[2 * ch + 1], ss, wl, lane);
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
            f32x4 so = r16::load_block(sb, nb, q) + r16::load_block(ab, nb, q);
            const f32x4 qq = k ? q1 : q0, aa = k ? a1 : a0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float n = sqrtf(n2.b[nb][r]);
                so[r] = so[r] + ((n * n) * qq[r] + aa[r]);              // s += vv_norm**2 * scale + add
            }
            if (ok) {
                r16::store_block(sb, nb, q, so);
                r16::store_block(ab, nb, q, f32x4{0, 0, 0, 0});
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
    // ---- phase C: v = v_eff + (U v_eff) * gates ; reset the accumulators for the next layer
    {
        OP ve[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            A16 t;
            r16::load_set(t, db + c * F, q);
            ve[c].set(t);
        }
#pragma unroll
        for (int ch = 0; ch < NB; ++ch) {
            const f32x4* wl = pipe.acquire();
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
                r16::gemm_bt(a0, a1, ve[c], wl, lane);
                const f32x4 e0 = r16::load_block(db + c * F, 2 * ch, q), e1 = r16::load_block(db + c * F, 2 * ch + 1, q);
                if (ok) {
                    r16::store_block(vb + c * F, 2 * ch, q, e0 + a0 * gg.b[2 * ch]);
                    r16::store_block(vb + c * F, 2 * ch + 1, q, e1 + a1 * gg.b[2 * ch + 1]);
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
        {
            A16 t;
            r16::load_set(t, sb, q);
            sn.set(t);
        }
#pragma unroll
        for (int ch = 0; ch < NB; ++ch) {
            const f32x4* wl = pipe.acquire();
            f32x4 a0 = r16::load_block(vec + UV::PB0 * F, 2 * ch, q), a1 = r16::load_block(vec + UV::PB0 * F, 2 * ch + 1, q);
            r16::gemm_bt(a0, a1, sn, wl, lane);
            pipe.release();
            if (ok) { r16::store_block(p.P + nd * F, 2 * ch, q, a0); r16::store_block(p.P + nd * F, 2 * ch + 1, q, a1); }
        }
    }
    pipe.drain();
}

// ================================================================================================== readout kernel
template <int NB, int WAVES>
__global__ __launch_bounds__(64 * WAVES, WAVES / 4) void painn_readout_kernel(const ReadoutParams p)
{
    using C = Cfg<NB, WAVES>;
    constexpr int F = C::F;
    extern __shared__ f32x4 lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), j = lane & 31, h = lane >> 5;
    Pipe<NB, C::T> pipe;
    pipe.init(reinterpret_cast<const f32x4*>(p.stream), p.nch, lds);
    const long long node = ((long long)blockIdx.x * WAVES + wave) * 32 + j;
    const bool ok = node < p.N;
    const size_t nd = (size_t)(ok ? node : p.N - 1);

    Act<NB> h1, h2;
    {
        Act<NB> ss;
        load_set(ss, p.s + nd * F, h);
#pragma unroll
        for (int nbo = 0; nbo < NB; ++nbo) {
            const f32x4* wl = pipe.begin();
            f32x16 a = load_block(p.mlp.b0, nbo, h);
            gemm_bt(a, ss, wl, lane);
            h1.b[nbo] = a;
            pipe.end();
        }
    }
    ln_silu(h1, p.mlp.g0, p.mlp.be0, h);
#pragma unroll
    for (int nbo = 0; nbo < NB; ++nbo) {
        const f32x4* wl = pipe.begin();
        f32x16 a = load_block(p.mlp.b1, nbo, h);
        gemm_bt(a, h1, wl, lane);
        h2.b[nbo] = a;
        pipe.end();
    }
    ln_silu(h2, p.mlp.g1, p.mlp.be1, h);
    // split(mlp(s), 1): [invariant_out (unused by cPaiNN.forward), gates]
    const float gate = dot_set(h2, p.w2_gate, h) + p.b2_gate;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        Act<NB> vc;
        load_set(vc, p.v + (nd * 3 + c) * F, h);
        const float vv = dot_set(vc, p.Vr, h);
        if (ok && h == 0) p.out[node * 3 + c] = vv * gate;
    }
}

// ================================================================================================== launchers
template <typename K>
static hipError_t set_lds(K kernel, size_t bytes)
{
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

// edge kernel LDS: two superchunks of two weight chunks, per-wave edge_dir scratch (4 waves x 16 rows x 16 B), layer vectors
static size_t edge_lds_bytes(int NB, int WAVES) { return 2 * edge_superchunk(NB, WAVES) * (size_t)256 * NB * 16 + WAVES * 256 + 21 * (size_t)32 * NB * 4; }

static size_t update_lds_bytes(int NB) { return 4 * (size_t)256 * NB * 16 + 10 * (size_t)32 * NB * 4; }

template <int NB, int EW, int NS>
static hipError_t configure_edge()
{
    const size_t be = edge_lds_bytes(NB, EW);
    hipError_t e;
    if ((e = set_lds(painn_edge_kernel<2 * NB, true, false, false, EW, NS>, be)) != hipSuccess) return e;
    if ((e = set_lds(painn_edge_kernel<2 * NB, false, false, false, EW, NS>, be)) != hipSuccess) return e;
    if ((e = set_lds(painn_edge_kernel<2 * NB, false, true, false, EW, NS>, be)) != hipSuccess) return e;
    if ((e = set_lds(painn_edge_kernel<2 * NB, true, true, false, EW, NS>, be)) != hipSuccess) return e;
    if ((e = set_lds(painn_edge_kernel<2 * NB, true, false, true, EW, NS>, be)) != hipSuccess) return e;
    if ((e = set_lds(painn_edge_kernel<2 * NB, false, false, true, EW, NS>, be)) != hipSuccess) return e;
    if ((e = set_lds(painn_edge_kernel<2 * NB, false, true, true, EW, NS>, be)) != hipSuccess) return e;
    if ((e = set_lds(painn_edge_kernel<2 * NB, true, true, true, EW, NS>, be)) != hipSuccess) return e;
    return hipSuccess;
}

template <int NB, int WAVES>
static hipError_t configure_nb()
{
    const size_t b = Cfg<NB, WAVES>::lds_bytes;
    hipError_t e;
#define TI_SET(k) if ((e = set_lds(k, b)) != hipSuccess) return e
    TI_SET((painn_embed_kernel<NB, WAVES, 2>)); TI_SET((painn_embed_kernel<NB, WAVES, 3>)); TI_SET((painn_embed_kernel<NB, WAVES, 4>));
    if ((e = configure_edge<NB, 4, 2>()) != hipSuccess) return e;
    if ((e = configure_edge<NB, 4, 4>()) != hipSuccess) return e;
    if constexpr (NB <= 4) { if ((e = configure_edge<NB, 8, 2>()) != hipSuccess) return e; }
    const size_t bu = update_lds_bytes(NB);
    if ((e = set_lds(painn_update_kernel<2 * NB, true, false>, bu)) != hipSuccess) return e;
    if ((e = set_lds(painn_update_kernel<2 * NB, false, false>, bu)) != hipSuccess) return e;
    if ((e = set_lds(painn_update_kernel<2 * NB, true, true>, bu)) != hipSuccess) return e;
    if ((e = set_lds(painn_update_kernel<2 * NB, false, true>, bu)) != hipSuccess) return e;
    TI_SET((painn_readout_kernel<NB, WAVES>));
#undef TI_SET
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

template <int NB, int WAVES>
static dim3 node_grid(long long N) { return dim3((unsigned)((N + 32LL * WAVES - 1) / (32LL * WAVES))); }

hipError_t launch_embed(int NBv, int nseg, const EmbedParams& p, hipStream_t st)
{
    TI_DISPATCH_NB(NBv, {
        const dim3 g = node_grid<NB, WAVES>(p.N);
        const size_t l = Cfg<NB, WAVES>::lds_bytes;
        if (nseg == 4) hipLaunchKernelGGL((painn_embed_kernel<NB, WAVES, 4>), g, dim3(64 * WAVES), l, st, p);
        else if (nseg == 3) hipLaunchKernelGGL((painn_embed_kernel<NB, WAVES, 3>), g, dim3(64 * WAVES), l, st, p);
        else hipLaunchKernelGGL((painn_embed_kernel<NB, WAVES, 2>), g, dim3(64 * WAVES), l, st, p);
    });
    return hipGetLastError();
}

template <int NB, int EW, int NS>
static void launch_edge_w(bool first, bool last, bool split, const EdgeParams& p, hipStream_t st)
{
    const dim3 g((unsigned)((p.n_groups + EW - 1) / EW)), t(64 * EW);          // one wave (= one group or part) each
    const size_t l = edge_lds_bytes(NB, EW);
    if (split) {
        if (first && last) hipLaunchKernelGGL((painn_edge_kernel<2 * NB, true, true, true, EW, NS>), g, t, l, st, p);
        else if (first) hipLaunchKernelGGL((painn_edge_kernel<2 * NB, true, false, true, EW, NS>), g, t, l, st, p);
        else if (last) hipLaunchKernelGGL((painn_edge_kernel<2 * NB, false, true, true, EW, NS>), g, t, l, st, p);
        else hipLaunchKernelGGL((painn_edge_kernel<2 * NB, false, false, true, EW, NS>), g, t, l, st, p);
    } else {
        if (first && last) hipLaunchKernelGGL((painn_edge_kernel<2 * NB, true, true, false, EW, NS>), g, t, l, st, p);
        else if (first) hipLaunchKernelGGL((painn_edge_kernel<2 * NB, true, false, false, EW, NS>), g, t, l, st, p);
        else if (last) hipLaunchKernelGGL((painn_edge_kernel<2 * NB, false, true, false, EW, NS>), g, t, l, st, p);
        else hipLaunchKernelGGL((painn_edge_kernel<2 * NB, false, false, false, EW, NS>), g, t, l, st, p);
    }
}

hipError_t launch_edge(int NBv, bool first, bool last, bool split, const EdgeParams& p, hipStream_t st)
{
    if (p.max_slots > 4) return hipErrorInvalidValue;          // build_templates never produces such a block
    // split-fp16 path only: the f32 path is matrix-bound and loses 4 % to the wider barriers (74.2 -> 77.2 ms per launch)
    const bool wide = split && p.n_groups >= 2048 && p.max_slots <= 2;          // enough groups to fill every CU with 8-wave workgroups
    TI_DISPATCH_NB(NBv, {
        (void)WAVES;
        if (NB <= 4 && wide) launch_edge_w<NB, (NB <= 4 ? 8 : 4), 2>(first, last, split, p, st);
        else if (p.max_slots <= 2) launch_edge_w<NB, 4, 2>(first, last, split, p, st);
        else launch_edge_w<NB, 4, 4>(first, last, split, p, st);
    });
    return hipGetLastError();
}

hipError_t launch_update(int NBv, bool has_next, bool split, const UpdateParams& p, hipStream_t st)
{
    TI_DISPATCH_NB(NBv, {
        (void)WAVES;
        const dim3 g((unsigned)((p.N + 63) / 64));                 // 4 waves x 16 atoms per workgroup
        const size_t l = update_lds_bytes(NB);
        if (split) {
            if (has_next) hipLaunchKernelGGL((painn_update_kernel<2 * NB, true, true>), g, dim3(256), l, st, p);
            else hipLaunchKernelGGL((painn_update_kernel<2 * NB, false, true>), g, dim3(256), l, st, p);
        } else {
            if (has_next) hipLaunchKernelGGL((painn_update_kernel<2 * NB, true, false>), g, dim3(256), l, st, p);
            else hipLaunchKernelGGL((painn_update_kernel<2 * NB, false, false>), g, dim3(256), l, st, p);
        }
    });
    return hipGetLastError();
}

hipError_t launch_readout(int NBv, const ReadoutParams& p, hipStream_t st)
{
    TI_DISPATCH_NB(NBv, {
        const dim3 g = node_grid<NB, WAVES>(p.N);
        const size_t l = Cfg<NB, WAVES>::lds_bytes;
        hipLaunchKernelGGL((painn_readout_kernel<NB, WAVES>), g, dim3(64 * WAVES), l, st, p);
    });
    return hipGetLastError();
}

}  // namespace ti
