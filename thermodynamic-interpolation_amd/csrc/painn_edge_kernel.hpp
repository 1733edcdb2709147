// painn_edge_kernel.hpp -- the message ("edge") kernel of the cPaiNN drift and its launchers, one translation unit per
// feature width (painn_edge_nb{1,2,4,8}.hip) so that the 88 instantiations compile in parallel.
//   edge    : AddSpatialFeatures graph.py:25-33 + SE3Message.forward cpainn.py:263-310   (reference, /root/reference/mdqm9/thermo/ambient/models)
#pragma once
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include "mfma_chain.hpp"
#include "ti_internal.hpp"

namespace ti {

template <typename K>
static hipError_t set_lds_edge(K kernel, size_t bytes)
{
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
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
__device__ __forceinline__ void add_noret(float* p, float v) { unsafeAtomicAdd(p, v); }
// accumulator update: a fire-and-forget atomic either way -- exchange on the first touch, add afterwards -- so that the later adds of the
// same wave are ordered behind the replacement at L2 (a plain store takes another path)
__device__ __forceinline__ void acc_out(float* p, float v, bool first)
{
    if (first) (void)__hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else unsafeAtomicAdd(p, v);
}

// One wave = one molecule group, walked in blocks of 16 edge rows on the 16x16x4 MFMA (mfma_chain.hpp, namespace r16).
// The waves of a workgroup share the weight-chunk stream (4 or 8 of them, see below); two waves per SIMD (F <= 128) hide each other's
// LayerNorm / reduction / wait phases behind matrix work.
// PREC selects the matrix path (mfma_chain.hpp: OpSel): 0 f32 MFMA, 1 split fp16 (Opnd<NBK, true>), 2 fp16 storage mode (OpndH; the
// state tensors P, v, e are then fp16 in HBM).
// Workgroup width.  4 waves (two workgroups per CU) is the default; for large split-fp16 launches (>= 2048 groups, F <= 128)
// the launcher picks the 8-wave build: one 512-thread workgroup per CU shares one weight stream (half the LDS-DMA writes) and, at F = 128,
// the freed LDS holds 4-chunk superchunks (half the barriers; needs chunk counts % 4 == 0: 56/48/40/32) -- 1.8 % on the edge
// kernel at B >= 32k.  Small launches keep 4 waves: fatter workgroups cost the latency regime 20-60 %.  F = 256 needs the
// 512-register budget of one wave per SIMD and is always 4 waves.
// The fp16 storage mode streams hi-only chunks of half the size: at F = 128 it stages twice as many per barrier (same LDS bytes, half
// the barriers; the chunk counts 56/48/40/32 divide by 8).
__host__ __device__ constexpr int edge_superchunk(int NB, int WAVES, bool H16 = false) { return ((WAVES == 8 && NB == 4) ? 4 : 2) * (H16 && NB == 4 ? 2 : 1); }
__host__ __device__ constexpr int edge_chunk4(int NB, bool H16) { return (H16 ? 128 : 256) * NB; }            // float4 per weight chunk
// F = 32 in the storage mode: a 2-chunk superchunk (4 KB) is smaller than one 16-byte lane per thread of the 8-wave build
// Does this (feature width, precision) run the one-accumulator split format (mfma_chain.hpp: Opnd1)?  TI_PREC_F16X2 at every width;
// ti_api.hip packs the message streams accordingly.  The F = 256 build (one wave per SIMD, operands partly in AGPRs) faulted on the
// device (memory aperture violation in its first launch) when hipcc spilled SGPRs into VGPR lanes: painn_edge_nb8.hip is compiled
// with -mllvm -amdgpu-spill-sgpr-to-vgpr=0 (build.py), which removes the fault.
#ifndef TI_ONE_CHAIN_MAX_NB
#define TI_ONE_CHAIN_MAX_NB 8
#endif
__host__ __device__ constexpr bool edge_one_chain(int NB, int PREC) { return PREC == 1 && NB <= TI_ONE_CHAIN_MAX_NB; }
__host__ __device__ constexpr bool edge_build_exists(int NB, int WAVES, int PREC) { return !(PREC == 2 && NB == 1 && WAVES == 8); }
template <int NBK, bool FIRST, bool LAST, int PREC, int WAVES, int NS>
__global__ __launch_bounds__(64 * WAVES, (NBK <= 8 ? 2 * 4 / WAVES : 1)) void painn_edge_kernel(const EdgeParams p)
{
    constexpr bool H16 = PREC == 2;                 // fp16 state tensors, hi-only weight chunks
    constexpr int F = 16 * NBK, NB = (F + 31) / 32, T = 64 * WAVES, CH4 = edge_chunk4(NB, H16);
    using A16 = r16::Act<NBK>;
    // split-fp16 path: the one-accumulator operand / weight format (mfma_chain.hpp: Opnd1; weights scaled per matrix, p.wscale)
    constexpr bool ONE = edge_one_chain(NB, PREC);
    using OP = std::conditional_t<ONE, r16::Opnd1<NBK>, typename r16::OpSel<NBK, PREC>::type>;
    extern __shared__ f32x4 lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), j = lane & 15, q = lane >> 4;
    constexpr int SC = edge_superchunk(NB, WAVES, H16);                             // weight chunks per barrier
    float* scratch = reinterpret_cast<float*>(lds + 2 * SC * CH4) + wave * 64;     // [16 rows][4] edge_dir of the block
    float* vec = reinterpret_cast<float*>(lds + 2 * SC * CH4) + WAVES * 64;        // [EV::COUNT][F]
    for (int i = threadIdx.x; i < EV::COUNT * F / 4; i += T)
        reinterpret_cast<f32x4*>(vec)[i] = reinterpret_cast<const f32x4*>(p.vecs)[i];
    PipeDMA<NB, T, SC, CH4> pipe;                                                // weights staged SC chunks per barrier
    pipe.init(reinterpret_cast<const f32x4*>(p.stream), p.nch, lds, wave, lane);   // barrier inside: vec is visible after it

    // ONE: accumulators hold S times the product, S = the power of two the host scaled that matrix by (ti_api.hip).  S cancels in
    // the LayerNorm behind a hidden layer (epsilon * S^2; the bias rows of `vec` are already scaled); the output products carry
    // S_phi2 * S_w2, divided out in the row masks of the per-atom sums and in the e update.
    const float eps_w0 = ONE ? 1e-5f * p.wscale[0] * p.wscale[0] : 1e-5f, eps_w1 = ONE ? 1e-5f * p.wscale[1] * p.wscale[1] : 1e-5f;
    const float eps_p0 = ONE ? 1e-5f * p.wscale[2] * p.wscale[2] : 1e-5f, eps_p1 = ONE ? 1e-5f * p.wscale[3] * p.wscale[3] : 1e-5f;
    const float s_p0 = ONE ? p.wscale[2] : 1.0f, inv_out = ONE ? 1.0f / (p.wscale[4] * p.wscale[5]) : 1.0f;
    const long long gi_raw = (long long)blockIdx.x * WAVES + wave;
    const bool group_ok = gi_raw < p.n_groups;
    const long long gi = group_ok ? gi_raw : p.n_groups - 1;
    // a group is P "parts" (ranges of destination atoms, each with its own row blocks and its own wave): ti_internal.hpp
    const long long mg = gi / p.parts;
    const uint32_t* rows = p.rows + (size_t)(gi - mg * p.parts) * p.nblk * 16;
    const int32_t* slotnode = p.slotnode + (size_t)(gi - mg * p.parts) * p.nblk * 16;

#ifdef TI_STAMPS      // diagnostic build only: (s_memtime, s_memrealtime) around the block loop of every wave -> in-kernel clock (MI355X guide, DVFS item 6)
    unsigned long long clk0 = 0, rt0 = 0;
    if (p.stamps) { clk0 = __builtin_amdgcn_s_memtime(); rt0 = __builtin_amdgcn_s_memrealtime(); }
#endif
    for (int blk = 0; blk < p.nblk; ++blk) {
        // ---- K1 geometry of this lane's row (the 4 quarters compute the same row)
        const uint32_t meta = rows[blk * 16 + j];
        long long mol = mg * p.G + row_mol(meta);
        mol = mol < p.B ? mol : p.B - 1;
        const long long nsrc = mol * p.A + row_src(meta), ndst = mol * p.A + row_dst(meta);
        const size_t erow0 = ((size_t)gi * p.nblk + blk) * 16;
        // The geometry and its encoding do not change between the layers of one drift evaluation: layer 0 computes them and parks, per
        // row, edge_dir (16 B) and the encoding AS THE MATRIX OPERAND it is used as (the register image of `enc`, F * 4 bytes, F * 2 in the
        // fp16 mode); the later layers load both -- no x gather behind the row word, no sqrt / divisions, no 16 sincos per lane, no
        // hi/lo conversion of the encoding.
        OP enc;
        f32x4* const enc_park = reinterpret_cast<f32x4*>(p.enc) + (erow0 / 16) * (sizeof(OP) / 16) * 64 + lane;
        f32x4* const geo_park = reinterpret_cast<f32x4*>(p.geo) + erow0 + j;
        if constexpr (FIRST) {
            const float rx = p.x[nsrc * 3 + 0] - p.x[ndst * 3 + 0];
            const float ry = p.x[nsrc * 3 + 1] - p.x[ndst * 3 + 1];
            const float rz = p.x[nsrc * 3 + 2] - p.x[ndst * 3 + 2];
            const float dist = sqrtf(rx * rx + ry * ry + rz * rz);
            const float den = 1.0f + dist;                       // edge_dir = r / (1 + d)   (not a unit vector)
            if (q == 0) {
                f32x4 dd = {rx / den, ry / den, rz / den, 0.f};
                *reinterpret_cast<f32x4*>(scratch + j * 4) = dd;
                if (group_ok) *geo_park = dd;
            }
            A16 t;
            r16::posenc_set(t, dist / p.length_scale, q);
            enc.set(t);
            if (group_ok) r16::opnd_store(enc, enc_park);
        } else {
            if (q == 0) *reinterpret_cast<f32x4*>(scratch + j * 4) = *geo_park;
            r16::opnd_load(enc, enc_park);
        }
        // ---- w(enc(d)) hidden layers
        OP g2;
        {
            OP g1;
            A16 t1;
#pragma unroll
            for (int c = 0; c < NB; ++c) {
                f32x4 a0 = r16::load_block(vec + EV::W_B0 * F, 2 * c, q), a1 = r16::load_block(vec + EV::W_B0 * F, 2 * c + 1, q);
                r16::gemm_on_pipe<false>(a0, a1, enc, pipe, lane);
                t1.b[2 * c] = a0; t1.b[2 * c + 1] = a1;
                pipe.release();
            }
            r16::ln_silu(t1, vec + EV::W_G0 * F, vec + EV::W_BE0 * F, q, eps_w0);
            g1.set(t1);
#pragma unroll
            for (int c = 0; c < NB; ++c) {
                f32x4 a0 = r16::load_block(vec + EV::W_B1 * F, 2 * c, q), a1 = r16::load_block(vec + EV::W_B1 * F, 2 * c + 1, q);
                r16::gemm_on_pipe<false>(a0, a1, g1, pipe, lane);
                t1.b[2 * c] = a0; t1.b[2 * c + 1] = a1;
                pipe.release();
            }
            r16::ln_silu(t1, vec + EV::W_G1 * F, vec + EV::W_BE1 * F, q, eps_w1);
            g2.set(t1);
        }
        // ---- phi([s[src] | e]) hidden layers; the s[src] half of the first Linear is P[src] (node kernels)
        OP h2, ein;                                   // ein stays live in the fp16 mode: e += de reuses it (below)
        {
            OP h1;
            A16 t1;
            float e_scale = 1.0f;
            if constexpr (H16) {
                if (FIRST) { r16::load_set(t1, p.edge_emb + row_type(meta) * F, q); ein.set(t1); }
                else ein.load_row(reinterpret_cast<const _Float16*>(p.e) + (erow0 + j) * F, q);        // the fp16 row IS the operand
            } else {
                if (FIRST) r16::load_set(t1, p.edge_emb + row_type(meta) * F, q);
                else       r16::load_set(t1, p.e + (erow0 + j) * F, q);
                e_scale = ein.set_scaled(t1);                                  // e is an un-normalised stream: per-row 2^k
            }
            const float e_inv = r16::pow2_inverse(e_scale) * s_p0;           // (the matrix scale of phi layer 0 rides on the accumulator init)
#pragma unroll
            for (int c = 0; c < NB; ++c) {
                // P[src] (the s[src] half of the Linear) / 2^k + W e' of the scaled rows, then * 2^k: all exact scalings
                f32x4 a0 = r16::load_state<H16>(p.P, (size_t)nsrc * F, 2 * c, q) * e_inv, a1 = r16::load_state<H16>(p.P, (size_t)nsrc * F, 2 * c + 1, q) * e_inv;
                r16::gemm_on_pipe<false>(a0, a1, ein, pipe, lane);
                a0 *= e_scale; a1 *= e_scale;
                t1.b[2 * c] = a0; t1.b[2 * c + 1] = a1;
                pipe.release();
            }
            r16::ln_silu(t1, vec + EV::P_G0 * F, vec + EV::P_BE0 * F, q, eps_p0);
            h1.set(t1);
#pragma unroll
            for (int c = 0; c < NB; ++c) {
                f32x4 a0 = r16::load_block(vec + EV::P_B1 * F, 2 * c, q), a1 = r16::load_block(vec + EV::P_B1 * F, 2 * c + 1, q);
                r16::gemm_on_pipe<false>(a0, a1, h1, pipe, lane);
                t1.b[2 * c] = a0; t1.b[2 * c + 1] = a1;
                pipe.release();
            }
            r16::ln_silu(t1, vec + EV::P_G1 * F, vec + EV::P_BE1 * F, q, eps_p1);
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
        for (int r = 0; r < 4; ++r) qs.set_row(r, row_slot(mi[r]), inv_out);
        int qnode;
        bool qfirst;                                 // first block of that atom: replace the accumulator instead of adding (ti_internal.hpp)
        {
            const int sn = slotnode[blk * 16 + r16::QuarterSum<NS>::slot_of_quarter(q)];
            const long long m2 = mg * p.G + slot_mol(sn);
            qnode = (sn >= 0 && group_ok && m2 < p.B) ? (int)(m2 * p.A + (sn & 255)) : -1;
            qfirst = (sn & SLOT_FIRST_TOUCH) != 0;
        }

        // (phi_c + b) * (w_c + b) for output chunk c (0 gates, 1 scale_edge_dir, 2 ds, 3 de, 4 cross gates), 32 features
        // fo .. fo+31 as two 16-feature blocks
        auto out_pair = [&](int c, int nbo, f32x4& r0, f32x4& r1) {
            f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0}, b0 = {0, 0, 0, 0}, b1 = {0, 0, 0, 0};
            r16::gemm_on_pipe<true>(a0, a1, h2, pipe, lane);
            pipe.release();
            r16::gemm_on_pipe<true>(b0, b1, g2, pipe, lane);
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
                if (qnode >= 0) acc_out(dst + (size_t)qnode * stride + 16 * (q >> 1), z, qfirst);
            } else {
                const float z0 = qs.sum(v0), z1 = qs.sum(v1);        // quarter q: slot q
                if (qnode >= 0) { float* d = dst + (size_t)qnode * stride; acc_out(d, z0, qfirst); acc_out(d + 16, z1, qfirst); }
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
            if constexpr (!LAST && H16) {
                // de in the row layout (the same two chunks with the operands the other way round): no sum over rows follows, and
                // in this layout the old row is exactly the operand `ein` loaded above (k-step nbo = feature blocks 2 nbo, 2 nbo + 1),
                // so e += de is two 8-byte stores per lane -- no atomics, no second read of e
                f32x4 a0 = r16::load_block(vec + (EV::P_B2 + 3) * F, 2 * nbo, q), a1 = r16::load_block(vec + (EV::P_B2 + 3) * F, 2 * nbo + 1, q);
                f32x4 b0 = r16::load_block(vec + (EV::W_B2 + 3) * F, 2 * nbo, q), b1 = r16::load_block(vec + (EV::W_B2 + 3) * F, 2 * nbo + 1, q);
                r16::gemm_on_pipe<false>(a0, a1, h2, pipe, lane);
                pipe.release();
                r16::gemm_on_pipe<false>(b0, b1, g2, pipe, lane);
                pipe.release();
                r16::h4 n0, n1;
                // runtime nbo: select the k-step of ein without dynamic register indexing
                r16::h8 eo = ein.hi[0];
#pragma unroll
                for (int m = 1; m < NBK / 2; ++m) eo = nbo == m ? ein.hi[m] : eo;
#pragma unroll
                for (int r = 0; r < 4; ++r) { n0[r] = (_Float16)((float)eo[r] + a0[r] * b0[r]); n1[r] = (_Float16)((float)eo[4 + r] + a1[r] * b1[r]); }
                if (group_ok) {
                    _Float16* ep = reinterpret_cast<_Float16*>(p.e) + (erow0 + j) * F + 32 * nbo + 4 * q;
                    *reinterpret_cast<r16::h4*>(ep) = n0; *reinterpret_cast<r16::h4*>(ep + 16) = n1;
                }
            }
            if constexpr (!LAST && !H16) {   // de: edge state update  e += de
                f32x4 v0, v1;
                out_pair(3, nbo, v0, v1);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float* ep = p.e + (erow0 + 4 * q + r) * F + fo;
                    if (group_ok) {
                        if (FIRST) {
                            const float* em = p.edge_emb + row_type(mi[r]) * F + fo;
                            ep[0] = em[0] + v0[r] * inv_out; ep[16] = em[16] + v1[r] * inv_out;
                        } else { add_noret(ep, v0[r] * inv_out); add_noret(ep + 16, v1[r] * inv_out); }
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
                        const size_t vo = (size_t)(m2 * p.A + row_src(mi[r])) * 3 * F + fo;
                        if constexpr (H16) {
                            const _Float16* vp = reinterpret_cast<const _Float16*>(p.v) + vo;
#pragma unroll
                            for (int c = 0; c < 3; ++c) { vs[c][0][r] = (float)vp[c * F]; vs[c][1][r] = (float)vp[c * F + 16]; }
                        } else {
                            const float* vp = p.v + vo;
#pragma unroll
                            for (int c = 0; c < 3; ++c) { vs[c][0][r] = vp[c * F]; vs[c][1][r] = vp[c * F + 16]; }
                        }
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
#ifdef TI_STAMPS
    if (p.stamps && lane == 0) {
        const unsigned long long clk1 = __builtin_amdgcn_s_memtime(), rt1 = __builtin_amdgcn_s_memrealtime();
        unsigned long long* c = p.stamps + 2048 + 2 * (size_t)gi_raw;
        c[0] = clk1 - clk0; c[1] = rt1 - rt0;
    }
#endif
}

// edge kernel LDS: two superchunks of two weight chunks, per-wave edge_dir scratch (4 waves x 16 rows x 16 B), layer vectors
static size_t edge_lds_bytes(int NB, int WAVES, bool h16) { return 2 * edge_superchunk(NB, WAVES, h16) * (size_t)edge_chunk4(NB, h16) * 16 + WAVES * 256 + 21 * (size_t)32 * NB * 4; }


template <int NB, int EW, int NS, int PREC>
static hipError_t configure_edge_prec()
{
    if constexpr (!edge_build_exists(NB, EW, PREC)) return hipSuccess;
    else {
    const size_t be = edge_lds_bytes(NB, EW, PREC == 2);
    hipError_t e;
    if ((e = set_lds_edge(painn_edge_kernel<2 * NB, true, false, PREC, EW, NS>, be)) != hipSuccess) return e;
    if ((e = set_lds_edge(painn_edge_kernel<2 * NB, false, false, PREC, EW, NS>, be)) != hipSuccess) return e;
    if ((e = set_lds_edge(painn_edge_kernel<2 * NB, false, true, PREC, EW, NS>, be)) != hipSuccess) return e;
    if ((e = set_lds_edge(painn_edge_kernel<2 * NB, true, true, PREC, EW, NS>, be)) != hipSuccess) return e;
    return hipSuccess;
    }
}
template <int NB, int EW, int NS>
static hipError_t configure_edge()
{
    hipError_t e;
    if ((e = configure_edge_prec<NB, EW, NS, 0>()) != hipSuccess) return e;
    if ((e = configure_edge_prec<NB, EW, NS, 1>()) != hipSuccess) return e;
    return configure_edge_prec<NB, EW, NS, 2>();
}

template <int NB, int EW, int NS, int PREC>
static void launch_edge_p(bool first, bool last, const EdgeParams& p, hipStream_t st)
{
    if constexpr (edge_build_exists(NB, EW, PREC)) {
    const dim3 g((unsigned)((p.n_groups + EW - 1) / EW)), t(64 * EW);          // one wave (= one group or part) each
    const size_t l = edge_lds_bytes(NB, EW, PREC == 2);
    if (first && last) hipLaunchKernelGGL((painn_edge_kernel<2 * NB, true, true, PREC, EW, NS>), g, t, l, st, p);
    else if (first) hipLaunchKernelGGL((painn_edge_kernel<2 * NB, true, false, PREC, EW, NS>), g, t, l, st, p);
    else if (last) hipLaunchKernelGGL((painn_edge_kernel<2 * NB, false, true, PREC, EW, NS>), g, t, l, st, p);
    else hipLaunchKernelGGL((painn_edge_kernel<2 * NB, false, false, PREC, EW, NS>), g, t, l, st, p);
    }
}
template <int NB, int EW, int NS>
static void launch_edge_w(bool first, bool last, int prec, const EdgeParams& p, hipStream_t st)
{
    if (prec == 2) launch_edge_p<NB, EW, NS, 2>(first, last, p, st);
    else if (prec == 1) launch_edge_p<NB, EW, NS, 1>(first, last, p, st);
    else launch_edge_p<NB, EW, NS, 0>(first, last, p, st);
}

// one feature width: configure every instantiation / launch the one the call needs
template <int NB>
static hipError_t configure_edge_nb()
{
    hipError_t e;
    if ((e = configure_edge<NB, 4, 2>()) != hipSuccess) return e;
    if ((e = configure_edge<NB, 4, 4>()) != hipSuccess) return e;
    if constexpr (NB <= 4) { if ((e = configure_edge<NB, 8, 2>()) != hipSuccess) return e; }
    return hipSuccess;
}
template <int NB>
static hipError_t launch_edge_nb(bool first, bool last, int prec, const EdgeParams& p, hipStream_t st)
{
    if (p.max_slots > EDGE_MAX_SLOTS) return hipErrorInvalidValue;          // build_templates never produces such a block
    // 8-wave workgroups: split-fp16 path only (the f32 path is matrix-bound and loses 4 % to the wider barriers), enough groups to
    // fill every CU, and at most two destination atoms per row block (the only form the wide build is instantiated for)
    const bool wide = NB <= 4 && prec != 0 && p.n_groups >= 2048 && p.max_slots <= 2 && edge_build_exists(NB, 8, prec);
    if constexpr (NB <= 4) { if (wide) { launch_edge_w<NB, 8, 2>(first, last, prec, p, st); return hipGetLastError(); } }
    if (p.max_slots <= 2) launch_edge_w<NB, 4, 2>(first, last, prec, p, st);
    else launch_edge_w<NB, 4, 4>(first, last, prec, p, st);
    return hipGetLastError();
}

}  // namespace ti
