// painn_kernels.hip -- cPaiNN drift kernels for gfx950 (MI355X).
//
// Restates (never copies) the arithmetic of the reference modules; citations are relative to /root/reference:
//   embed   : InvariantFeatures/NominalEmbedding/TemperatureEncoder/PositionalEmbedding + CombineInvariantFeatures
//             mdqm9/thermo/ambient/models/embedding.py:68-86,127-160,200-212,249-261 (latent twins)
//   edge    : AddSpatialFeatures graph.py:25-33 + SE3Message.forward cpainn.py:263-310
//   update  : Update.forward cpainn.py:345-376, EquivariantLinear cpainn.py:403
//   readout : LayerReadout.forward cpainn.py:425-437, cPaiNN.forward cpainn.py:112-115
//
// All dense layers run on v_mfma_f32_32x32x2_f32 through mfma_chain.hpp; activations never leave registers inside an
// MLP chain; per-atom sums over incoming edges are done inside one wave in a fixed order (deterministic, no atomics).
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include "mfma_chain.hpp"
#include "ti_internal.hpp"

namespace ti {

template <int NB, int WAVES>
struct Cfg {
    static constexpr int F = 32 * NB;
    static constexpr int T = 64 * WAVES;
    static constexpr int CH4 = 256 * NB;
    static constexpr size_t lds_bytes = 2 * (size_t)CH4 * 16 + (size_t)WAVES * 512 + 21 * (size_t)F * 4;   // chunks, edge_dir scratch, layer vectors
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

// sum over the 16 rows of this lane-half that belong to slot t, both halves combined (fixed order -> deterministic)
__device__ __forceinline__ float slot_sum(const f32x16& q, const uint32_t (&mi)[16], int t)
{
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) a += (row_slot(mi[i]) == t) ? q[i] : 0.f;
    return a + xhalf(a);
}

// Fire-and-forget fp32 add (global_atomic_add_f32, no return): nothing waits for the memory round trip.  Every
// accumulator element starts at zero and receives at most two adds per launch, both from the one wave that owns the
// molecule (an atom's <= 31 incoming edges span at most two 32-row blocks), so the result does not depend on their order.
__device__ __forceinline__ void add_noret(float* p, float v) { unsafeAtomicAdd(p, v); }

template <int NB, int WAVES, bool FIRST, bool LAST>
__global__ __launch_bounds__(64 * WAVES, WAVES / 4) void painn_edge_kernel(const EdgeParams p)
{
    using C = Cfg<NB, WAVES>;
    constexpr int F = C::F;
    extern __shared__ f32x4 lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), j = lane & 31, h = lane >> 5;
    float* scratch = reinterpret_cast<float*>(lds + 2 * C::CH4) + wave * 128;      // [32 rows][4] edge_dir of the block
    float* vec = reinterpret_cast<float*>(lds + 2 * C::CH4) + WAVES * 128;         // [EV::COUNT][F]
    for (int i = threadIdx.x; i < EV::COUNT * F / 4; i += C::T)
        reinterpret_cast<f32x4*>(vec)[i] = reinterpret_cast<const f32x4*>(p.vecs)[i];
    Pipe<NB, C::T> pipe;
    pipe.init(reinterpret_cast<const f32x4*>(p.stream), p.nch, lds);               // barrier inside: vec is visible after it

    const long long gi_raw = (long long)blockIdx.x * WAVES + wave;
    const bool group_ok = gi_raw < p.n_groups;
    const long long gi = group_ok ? gi_raw : p.n_groups - 1;
    const int fcol = lane & 31;

    for (int blk = 0; blk < p.nblk; ++blk) {
        // ---- K1 geometry of this lane's row (both halves compute the same row)
        const uint32_t meta = p.rows[blk * 32 + j];
        long long mol = gi * p.G + row_mol(meta);
        mol = mol < p.B ? mol : p.B - 1;
        const long long nsrc = mol * p.A + row_src(meta), ndst = mol * p.A + row_dst(meta);
        const size_t erow0 = ((size_t)gi * p.nblk + blk) * 32;
        // edge-state rows and P[src] are needed by the phi chain only: issue the loads now, use them after the w chain
        Act<NB> ein;
        if (FIRST) load_set(ein, p.edge_emb + row_type(meta) * F, h);
        else       load_set(ein, p.e + (erow0 + j) * F, h);
        Act<NB> pin;
        load_set(pin, p.P + (size_t)nsrc * F, h);
        float dist;
        {
            const float rx = p.x[nsrc * 3 + 0] - p.x[ndst * 3 + 0];
            const float ry = p.x[nsrc * 3 + 1] - p.x[ndst * 3 + 1];
            const float rz = p.x[nsrc * 3 + 2] - p.x[ndst * 3 + 2];
            dist = sqrtf(rx * rx + ry * ry + rz * rz);
            const float den = 1.0f + dist;                       // edge_dir = r / (1 + d)   (not a unit vector)
            if (h == 0) {
                f32x4 dd = {rx / den, ry / den, rz / den, 0.f};
                *reinterpret_cast<f32x4*>(scratch + j * 4) = dd;
            }
        }
        // ---- w(enc(d)) hidden layers
        Act<NB> g2;
        {
            Act<NB> g1;
            {
                Act<NB> enc;
                posenc_set(enc, dist / p.length_scale, h);
#pragma unroll
                for (int nbo = 0; nbo < NB; ++nbo) {
                    const f32x4* wl = pipe.begin();
                    f32x16 a = load_block(vec + EV::W_B0 * F, nbo, h);
                    gemm_bt(a, enc, wl, lane);
                    g1.b[nbo] = a;
                    pipe.end();
                }
            }
            ln_silu(g1, vec + EV::W_G0 * F, vec + EV::W_BE0 * F, h);
#pragma unroll
            for (int nbo = 0; nbo < NB; ++nbo) {
                const f32x4* wl = pipe.begin();
                f32x16 a = load_block(vec + EV::W_B1 * F, nbo, h);
                gemm_bt(a, g1, wl, lane);
                g2.b[nbo] = a;
                pipe.end();
            }
            ln_silu(g2, vec + EV::W_G1 * F, vec + EV::W_BE1 * F, h);
        }
        // ---- phi([s[src] | e]) hidden layers; the s[src] half of the first Linear is P[src] (node kernels)
        Act<NB> h2;
        {
            Act<NB> h1;
#pragma unroll
            for (int nbo = 0; nbo < NB; ++nbo) {
                const f32x4* wl = pipe.begin();
                f32x16 a = pin.b[nbo];
                gemm_bt(a, ein, wl, lane);
                h1.b[nbo] = a;
                pipe.end();
            }
            ln_silu(h1, vec + EV::P_G0 * F, vec + EV::P_BE0 * F, h);
#pragma unroll
            for (int nbo = 0; nbo < NB; ++nbo) {
                const f32x4* wl = pipe.begin();
                f32x16 a = load_block(vec + EV::P_B1 * F, nbo, h);
                gemm_bt(a, h1, wl, lane);
                h2.b[nbo] = a;
                pipe.end();
            }
            ln_silu(h2, vec + EV::P_G1 * F, vec + EV::P_BE1 * F, h);
        }
        // ---- output layer, flipped: features on lanes, the block's 32 rows in registers (row = acc_row(i, h))
        uint32_t mi[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) mi[i] = p.rows[blk * 32 + acc_row(i, h)];
        const int nslots = p.nslots[blk];
        const int32_t* slotnode = p.slotnode + blk * 32;

        // (phi_c + b) * (w_c + b) for output chunk c (0 gates, 1 scale_edge_dir, 2 ds, 3 de, 4 cross gates), block nbo
        auto out_pair = [&](int c, int nbo) -> f32x16 {
            f32x16 a0 = {0}, a1 = {0};
            const f32x4* wl0 = pipe.begin();
            gemm_fl(a0, h2, wl0, lane);
            pipe.end();
            const f32x4* wl1 = pipe.begin();
            gemm_fl(a1, g2, wl1, lane);
            pipe.end();
            const float bp = vec[(EV::P_B2 + c) * F + 32 * nbo + fcol], bw = vec[(EV::W_B2 + c) * F + 32 * nbo + fcol];
            f32x16 r;
#pragma unroll
            for (int i = 0; i < 16; ++i) r[i] = (a0[i] + bp) * (a1[i] + bw);
            return r;
        };
        // add the per-slot sums of q into dst[node*stride] (dst already offset to component/feature)
        auto emit = [&](const f32x16& q, float* dst, size_t stride) {
            for (int t = 0; t < nslots; ++t) {
                const float a = slot_sum(q, mi, t);
                const int sn = slotnode[t];
                const long long m2 = gi * p.G + (sn >> 8);
                if (group_ok && m2 < p.B && h == 0) add_noret(dst + (size_t)(m2 * p.A + (sn & 255)) * stride, a);
            }
        };

#pragma unroll 1
        for (int nbo = 0; nbo < NB; ++nbo) {
            const int fo = 32 * nbo + fcol;
            // v[src] of the block's rows for the gated term: issued here, consumed after two more weight chunks
            f32x16 vs[3];
            if (!FIRST) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    long long m2 = gi * p.G + row_mol(mi[i]);
                    m2 = m2 < p.B ? m2 : p.B - 1;
                    const float* vp = p.v + (size_t)(m2 * p.A + row_src(mi[i])) * 3 * F + fo;
                    vs[0][i] = vp[0]; vs[1][i] = vp[F]; vs[2][i] = vp[2 * F];
                }
            }
            {   // ds: invariant message, summed over incoming edges
                const f32x16 q = out_pair(2, nbo);
                emit(q, p.dsacc + fo, F);
            }
            if (!LAST) {   // de: edge state update  e += de
                const f32x16 q = out_pair(3, nbo);
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    float* ep = p.e + (erow0 + acc_row(i, h)) * F + fo;
                    if (group_ok) {
                        if (FIRST) *ep = p.edge_emb[row_type(mi[i]) * F + fo] + q[i];
                        else add_noret(ep, q[i]);
                    }
                }
            }
            {   // equivariant message: sum_e (sed * dir_e + gates * v[src_e]) -> dvacc ; sum_e cg * dir_e -> cacc
                const f32x16 sed = out_pair(1, nbo);
                f32x16 gates = {0};
                if (!FIRST) gates = out_pair(0, nbo);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    f32x16 q;
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        q[i] = sed[i] * scratch[acc_row(i, h) * 4 + c];
                        if (!FIRST) q[i] = fmaf(gates[i], vs[c][i], q[i]);
                    }
                    emit(q, p.dvacc + c * F + fo, 3 * F);
                }
                if (!FIRST) {
                    const f32x16 cg = out_pair(4, nbo);
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        f32x16 q;
#pragma unroll
                        for (int i = 0; i < 16; ++i) q[i] = cg[i] * scratch[acc_row(i, h) * 4 + c];
                        emit(q, p.cacc + c * F + fo, 3 * F);
                    }
                }
            }
        }
    }
}

// ================================================================================================== update kernel
// v <- v + dv  with  dv = dvacc + cacc x v   (the cross product with v[dst] factors out of the edge sum),
// then Update.forward; finally P for the next layer's message block.
template <int NB, int WAVES, bool HAS_NEXT>
__global__ __launch_bounds__(64 * WAVES, WAVES / 4) void painn_update_kernel(const UpdateParams p)
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
    float* vb = p.v + nd * 3 * F;
    float* db = p.dvacc + nd * 3 * F;
    float* cb = p.cacc + nd * 3 * F;
    float* sb = p.s + nd * F;
    float* ab = p.dsacc + nd * F;                 // sum of the invariant messages of this layer (edge kernel)

    // ---- phase A: v_eff = v + dvacc + cacc x v (parked in dvacc), n2 = |V v_eff|^2 over the 3 components
    Act<NB> n2;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) n2.b[nb] = f32x16{0};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int c1 = (c + 1) % 3, c2 = (c + 2) % 3;
        Act<NB> ve;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const f32x16 vc = load_block(vb + c * F, nb, h), dd = load_block(db + c * F, nb, h);
            const f32x16 v1 = load_block(vb + c1 * F, nb, h), v2 = load_block(vb + c2 * F, nb, h);
            const f32x16 k1 = load_block(cb + c1 * F, nb, h), k2 = load_block(cb + c2 * F, nb, h);
            ve.b[nb] = (vc + dd) + (k1 * v2 - k2 * v1);              // torch.cross(edge_dir, v[dst]) summed over edges
            if (ok) store_block(db + c * F, nb, h, ve.b[nb]);
        }
#pragma unroll
        for (int nbo = 0; nbo < NB; ++nbo) {
            const f32x4* wl = pipe.begin();
            f32x16 a = {0};
            gemm_bt(a, ve, wl, lane);                                 // vv = V v
            pipe.end();
            n2.b[nbo] += a * a;
        }
    }
    // ---- phase B: MLP([ |vv| , s ])
    Act<NB> h2;
    {
        Act<NB> acc;
#pragma unroll
        for (int nbo = 0; nbo < NB; ++nbo) acc.b[nbo] = load_block(p.mlp.b0, nbo, h);
        {
            Act<NB> nn;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int i = 0; i < 16; ++i) nn.b[nb][i] = sqrtf(n2.b[nb][i]);
#pragma unroll
            for (int nbo = 0; nbo < NB; ++nbo) {
                const f32x4* wl = pipe.begin();
                gemm_bt(acc.b[nbo], nn, wl, lane);
                pipe.end();
            }
        }
        {
            Act<NB> ss;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) ss.b[nb] = load_block(sb, nb, h) + load_block(ab, nb, h);      // s += ds
#pragma unroll
            for (int nbo = 0; nbo < NB; ++nbo) {
                const f32x4* wl = pipe.begin();
                gemm_bt(acc.b[nbo], ss, wl, lane);
                pipe.end();
            }
        }
        ln_silu(acc, p.mlp.g0, p.mlp.be0, h);
#pragma unroll
        for (int nbo = 0; nbo < NB; ++nbo) {
            const f32x4* wl = pipe.begin();
            f32x16 a = load_block(p.mlp.b1, nbo, h);
            gemm_bt(a, acc, wl, lane);
            h2.b[nbo] = a;
            pipe.end();
        }
        ln_silu(h2, p.mlp.g1, p.mlp.be1, h);
    }
    // output chunks: [scale_squared_norm, add_invariant] per block, then gates
#pragma unroll
    for (int nbo = 0; nbo < NB; ++nbo) {
        const f32x4* wl = pipe.begin();
        f32x16 q = load_block(p.mlp.b2 + F, nbo, h);
        gemm_bt(q, h2, wl, lane);
        pipe.end();
        wl = pipe.begin();
        f32x16 a = load_block(p.mlp.b2 + 2 * F, nbo, h);
        gemm_bt(a, h2, wl, lane);
        pipe.end();
        f32x16 so = load_block(sb, nbo, h) + load_block(ab, nbo, h);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float n = sqrtf(n2.b[nbo][i]);
            so[i] = so[i] + ((n * n) * q[i] + a[i]);                  // s += vv_norm**2 * scale + add
        }
        if (ok) {
            store_block(sb, nbo, h, so);
            store_block(ab, nbo, h, f32x16{0});
        }
    }
    Act<NB> gg;
#pragma unroll
    for (int nbo = 0; nbo < NB; ++nbo) {
        const f32x4* wl = pipe.begin();
        f32x16 a = load_block(p.mlp.b2, nbo, h);
        gemm_bt(a, h2, wl, lane);
        gg.b[nbo] = a;
        pipe.end();
    }
    // ---- phase C: v = v_eff + (U v_eff) * gates ; reset the accumulators for the next layer
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        Act<NB> ve;
        load_set(ve, db + c * F, h);
#pragma unroll
        for (int nbo = 0; nbo < NB; ++nbo) {
            const f32x4* wl = pipe.begin();
            f32x16 a = {0};
            gemm_bt(a, ve, wl, lane);
            pipe.end();
            const f32x16 vn = ve.b[nbo] + a * gg.b[nbo];
            if (ok) {
                store_block(vb + c * F, nbo, h, vn);
                store_block(db + c * F, nbo, h, f32x16{0});
                store_block(cb + c * F, nbo, h, f32x16{0});
            }
        }
    }
    // ---- phase D: P for the next message block
    if (HAS_NEXT) {
        Act<NB> sn;
        load_set(sn, sb, h);
#pragma unroll
        for (int nbo = 0; nbo < NB; ++nbo) {
            const f32x4* wl = pipe.begin();
            f32x16 a = load_block(p.pb0_next, nbo, h);
            gemm_bt(a, sn, wl, lane);
            pipe.end();
            if (ok) store_block(p.P + nd * F, nbo, h, a);
        }
    }
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

template <int NB, int WAVES>
static hipError_t configure_nb()
{
    const size_t b = Cfg<NB, WAVES>::lds_bytes;
    hipError_t e;
#define TI_SET(k) if ((e = set_lds(k, b)) != hipSuccess) return e
    TI_SET((painn_embed_kernel<NB, WAVES, 2>)); TI_SET((painn_embed_kernel<NB, WAVES, 3>)); TI_SET((painn_embed_kernel<NB, WAVES, 4>));
    TI_SET((painn_edge_kernel<NB, WAVES, true, false>)); TI_SET((painn_edge_kernel<NB, WAVES, false, false>));
    TI_SET((painn_edge_kernel<NB, WAVES, false, true>)); TI_SET((painn_edge_kernel<NB, WAVES, true, true>));
    TI_SET((painn_update_kernel<NB, WAVES, true>)); TI_SET((painn_update_kernel<NB, WAVES, false>));
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

hipError_t launch_edge(int NBv, bool first, bool last, const EdgeParams& p, hipStream_t st)
{
    TI_DISPATCH_NB(NBv, {
        const dim3 g((unsigned)((p.n_groups + WAVES - 1) / WAVES));
        const size_t l = Cfg<NB, WAVES>::lds_bytes;
        if (first && last) hipLaunchKernelGGL((painn_edge_kernel<NB, WAVES, true, true>), g, dim3(64 * WAVES), l, st, p);
        else if (first) hipLaunchKernelGGL((painn_edge_kernel<NB, WAVES, true, false>), g, dim3(64 * WAVES), l, st, p);
        else if (last) hipLaunchKernelGGL((painn_edge_kernel<NB, WAVES, false, true>), g, dim3(64 * WAVES), l, st, p);
        else hipLaunchKernelGGL((painn_edge_kernel<NB, WAVES, false, false>), g, dim3(64 * WAVES), l, st, p);
    });
    return hipGetLastError();
}

hipError_t launch_update(int NBv, bool has_next, const UpdateParams& p, hipStream_t st)
{
    TI_DISPATCH_NB(NBv, {
        const dim3 g = node_grid<NB, WAVES>(p.N);
        const size_t l = Cfg<NB, WAVES>::lds_bytes;
        if (has_next) hipLaunchKernelGGL((painn_update_kernel<NB, WAVES, true>), g, dim3(64 * WAVES), l, st, p);
        else hipLaunchKernelGGL((painn_update_kernel<NB, WAVES, false>), g, dim3(64 * WAVES), l, st, p);
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
