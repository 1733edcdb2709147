// painn_pair_kernel.hpp -- the PAIR-MAJOR message kernel of the cPaiNN drift: SE3Message.forward (cpainn.py:263-310, reference
// /root/reference/mdqm9/thermo/ambient/models) with the filter branch evaluated once per atom pair.
//
//   h = phi([s[src] | e]) * w(enc(|x_src - x_dst|))        (cpainn.py:283-289)
//
// `w(enc(d))` depends on the edge only through its length, so the two directed edges i -> j and j -> i of a pair share it bit for bit
// (|-r| = |r|); it is 14 of the 28 F x F products the directed kernel (painn_edge_kernel.hpp) executes per edge.  Here a row block is 16
// PAIRS (ti_internal.hpp: a 4 x 4 tile of "I" atoms x "J" atoms, row 4a + b = pair (I[a], J[b])):
//   * the w branch (2 hidden layers, 5 output slices) runs once per pair;
//   * the phi branch runs for direction A (I -> J) and direction B (J -> I) in LOCK STEP against the same weight chunk: every LDS
//     fragment feeds both operand sets (r16::gemm_x2), so a chunk visit -- its LDS-DMA, its barrier share, its fragment reads -- serves
//     32 edges instead of 16.  The packed weight stream is the directed kernel's, unchanged;
//   * the per-atom sums need no masks and no slot table walk: in the flipped output layout lane (n, q) holds rows 4q + r, i.e. I slot q
//     and J slots r = 0..3.  Direction B (dst = I[q]) is an in-lane sum of the four registers; direction A (dst = J[r]) is the sum over
//     the four lane rows, two v_permlane swap levels that leave slot q' in lane row q'.  Rows of pairs that do not exist are zeroed
//     through the shared w factor (which also carries 1 / (S_phi S_w) of the one-accumulator format).
//   * Per-atom sums (ds / dv / c): one wave owns a molecule group and walks its blocks in order, so the sums of a (block, slot) go to the
//     atom's accumulator row with fire-and-forget float atomics, the first touch of a launch replacing the stale contents (acc_out,
//     painn_edge_kernel.hpp) -- no reduction pass, nothing to zero (TI_PAIR_ACC_ATOMIC = 1, the default: 29.4 ms against 31.5 ms, same
//     box, for the alternative that is kept behind TI_PAIR_ACC_ATOMIC = 0: per-(block, slot) partial rows with plain stores +
//     `pair_reduce_kernel`, profiles/r03e_*).  The edge state is updated in the ROW layout (the de slice with the operands the other
//     way round, like the fp16 storage mode of the directed kernel): each e row has one owner, so e += de is a 16-byte load and store
//     per lane, no atomics.
// Per 32 directed edges: 84 chunk products and 56 chunk visits instead of 112 and 112, six LayerNorm / SiLU / operand-split phases
// instead of eight.  What it costs: 4 x 4 tiles cover a complete graph of A atoms with (A - 1) / (4 * ceil((A - 1) / 4)) of their rows
// at best (ti_api.hip: build_pair_template; 85 % for 18 atoms), and every atom's accumulators are touched from ~ (A - 1) / 4 blocks.
// Results are deterministic (one wave owns a group of molecules, fixed order); they differ from the directed kernel's by the order
// of the per-atom sums only.
#pragma once
#include "painn_edge_kernel.hpp"

namespace ti {

__host__ __device__ constexpr bool pair_build_exists(int NB, int WAVES, int PREC) { return PREC != 2 && NB <= 4 && (WAVES == 4 || WAVES == 8); }
// weight ring (mfma_chain.hpp PipeDMA): 2-chunk superchunks, two of them (64 KB at F = 128) in both builds.  A deeper ring (the stream
// requested three superchunks ahead: TI_PAIR_NBUF = 4, the round's first choice) buys nothing -- every superchunk barrier drains the
// wave's memory queue anyway as soon as a store or an atomic is in flight -- and its index arithmetic cost the
// last-layer kernel 31 spilled registers: 26.16 ms with two buffers against 26.48 with four, same box (profiles/r03l_ring_depth.txt).
#ifndef TI_PAIR_NBUF
#define TI_PAIR_NBUF 2
#endif
__host__ __device__ constexpr int pair_superchunk() { return 2; }
__host__ __device__ constexpr int pair_ring(int WAVES) { return WAVES == 8 ? TI_PAIR_NBUF : 2; }
static size_t pair_lds_bytes(int NB, int WAVES) { return (size_t)pair_ring(WAVES) * pair_superchunk() * edge_chunk4(NB, false) * 16 + WAVES * 256 + 21 * (size_t)32 * NB * 4; }

template <int NBK, bool FIRST, bool LAST, int PREC, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 2 * 4 / WAVES) void painn_pair_kernel(const EdgeParams p)
{
    static_assert(PREC == 0 || PREC == 1, "the fp16 storage mode keeps the directed message kernel");
    constexpr int F = 16 * NBK, NB = (F + 31) / 32, T = 64 * WAVES, CH4 = edge_chunk4(NB, false);
    using A16 = r16::Act<NBK>;
    constexpr bool ONE = edge_one_chain(NB, PREC);
    using OP = std::conditional_t<ONE, r16::Opnd1<NBK>, typename r16::OpSel<NBK, PREC>::type>;
    extern __shared__ f32x4 lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), j = lane & 15, q = lane >> 4;
    constexpr int SC = pair_superchunk(), NBUF = pair_ring(WAVES);
    float* scratch = reinterpret_cast<float*>(lds + NBUF * SC * CH4) + wave * 64;  // [16 pair rows][4] edge_dir of direction A
    float* vec = reinterpret_cast<float*>(lds + NBUF * SC * CH4) + WAVES * 64;     // [EV::COUNT][F]
    for (int i = threadIdx.x; i < EV::COUNT * F / 4; i += T)
        reinterpret_cast<f32x4*>(vec)[i] = reinterpret_cast<const f32x4*>(p.vecs)[i];
#ifndef TI_PAIR_STAGGER
#define TI_PAIR_STAGGER 0
#endif
    PipeDMA<NB, T, SC, CH4, WAVES == 8 && TI_PAIR_STAGGER, NBUF> pipe;        // 8 waves: SIMD partners half a phase apart (mfma_chain.hpp)
    pipe.init(reinterpret_cast<const f32x4*>(p.stream), p.nch, lds, wave, lane);

    const float eps_w0 = ONE ? 1e-5f * p.wscale[0] * p.wscale[0] : 1e-5f, eps_w1 = ONE ? 1e-5f * p.wscale[1] * p.wscale[1] : 1e-5f;
    const float eps_p0 = ONE ? 1e-5f * p.wscale[2] * p.wscale[2] : 1e-5f, eps_p1 = ONE ? 1e-5f * p.wscale[3] * p.wscale[3] : 1e-5f;
    const float s_p0 = ONE ? p.wscale[2] : 1.0f, inv_out = ONE ? 1.0f / (p.wscale[4] * p.wscale[5]) : 1.0f;
    const long long gi_raw = (long long)blockIdx.x * WAVES + wave;
    const bool group_ok = gi_raw < p.n_groups;
    const long long gi = group_ok ? gi_raw : p.n_groups - 1;
    auto node_of = [&](int mol_local, int atom) {
        long long m = gi * p.G + mol_local;
        m = m < p.B ? m : p.B - 1;
        return m * p.A + atom;
    };

    // Diagnostic build only (-DTI_STAMPS; never the product): shader-clock stamps of ONE row block of a few workgroups, and one
    // (s_memtime, s_memrealtime) pair around the whole block loop of every wave for the in-kernel clock (MI355X guide, DVFS item 6).
    // Stamps go to a buffer of their own that nothing else reads.
#ifdef TI_STAMPS
    constexpr int STAMP_SLOTS = 64;
    const bool st_wg = blockIdx.x >= 300 && blockIdx.x < 304;
    unsigned long long* const st_buf = p.stamps ? p.stamps + ((size_t)(blockIdx.x - 300) * WAVES + wave) * STAMP_SLOTS : nullptr;
    int st_i = 0;
    unsigned long long clk0 = 0, rt0 = 0;
    if (p.stamps) { clk0 = __builtin_amdgcn_s_memtime(); rt0 = __builtin_amdgcn_s_memrealtime(); }
#define TI_STAMP() do { if (st_wg && st_buf && blk == 3 && st_i < STAMP_SLOTS) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (lane == 0) st_buf[st_i] = t_; ++st_i; } } while (0)
#else
#define TI_STAMP() do { } while (0)
#endif

    for (int blk = 0; blk < p.nblk; ++blk) {
        TI_STAMP();
        // ---- K1 geometry of this lane's pair row (the 4 quarters compute the same row); direction A: r = x[I] - x[J]
        const uint32_t meta = p.rows[blk * 16 + j];
        const long long nI = node_of(prow_molI(meta), prow_atomI(meta)), nJ = node_of(prow_molJ(meta), prow_atomJ(meta));
        const size_t brow0 = ((size_t)gi * p.nblk + blk) * 16;       // pair rows: parked encoding / edge_dir
        const size_t erowA = brow0 * 2, erowB = erowA + 16;             // e rows of the two directions
        OP enc;
        f32x4* const enc_park = reinterpret_cast<f32x4*>(p.enc) + (brow0 / 16) * (sizeof(OP) / 16) * 64 + lane;
        f32x4* const geo_park = reinterpret_cast<f32x4*>(p.geo) + brow0 + j;
        if constexpr (FIRST) {
            const float rx = p.x[nI * 3 + 0] - p.x[nJ * 3 + 0];
            const float ry = p.x[nI * 3 + 1] - p.x[nJ * 3 + 1];
            const float rz = p.x[nI * 3 + 2] - p.x[nJ * 3 + 2];
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
        TI_STAMP();
        // ---- w(enc(d)) hidden layers, once per pair
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
            TI_STAMP();
            r16::ln_silu(t1, vec + EV::W_G0 * F, vec + EV::W_BE0 * F, q, eps_w0);
            g1.set(t1);
            TI_STAMP();
#pragma unroll
            for (int c = 0; c < NB; ++c) {
                f32x4 a0 = r16::load_block(vec + EV::W_B1 * F, 2 * c, q), a1 = r16::load_block(vec + EV::W_B1 * F, 2 * c + 1, q);
                r16::gemm_on_pipe<false>(a0, a1, g1, pipe, lane);
                t1.b[2 * c] = a0; t1.b[2 * c + 1] = a1;
                pipe.release();
            }
            TI_STAMP();
            r16::ln_silu(t1, vec + EV::W_G1 * F, vec + EV::W_BE1 * F, q, eps_w1);
            g2.set(t1);
            TI_STAMP();
        }
        // ---- phi([s[src] | e]) hidden layers of both directions in lock step; the s[src] half of the first Linear is P[src]
        OP h2A, h2B;
        {
            OP inA, inB;
            A16 tA, tB;
            float scA, scB;
            if (FIRST) {
                r16::load_set(tA, p.edge_emb + prow_type(meta) * F, q);      // e = edge_emb[type], the same row for both directions
                scA = inA.set_scaled(tA);
                inB = inA; scB = scA;
            } else {
                r16::load_set(tA, p.e + (erowA + j) * F, q);
                r16::load_set(tB, p.e + (erowB + j) * F, q);
                scA = inA.set_scaled(tA);                                    // e is an un-normalised stream: per-row 2^k
                scB = inB.set_scaled(tB);
            }
            const float ivA = r16::pow2_inverse(scA) * s_p0, ivB = r16::pow2_inverse(scB) * s_p0;
            TI_STAMP();
#pragma unroll
            for (int c = 0; c < NB; ++c) {
                f32x4 a0 = r16::load_state<false>(p.P, (size_t)nI * F, 2 * c, q) * ivA, a1 = r16::load_state<false>(p.P, (size_t)nI * F, 2 * c + 1, q) * ivA;
                f32x4 b0 = r16::load_state<false>(p.P, (size_t)nJ * F, 2 * c, q) * ivB, b1 = r16::load_state<false>(p.P, (size_t)nJ * F, 2 * c + 1, q) * ivB;
                r16::gemm_x2_on_pipe<false>(a0, a1, b0, b1, inA, inB, pipe, lane);
                tA.b[2 * c] = a0 * scA; tA.b[2 * c + 1] = a1 * scA;
                tB.b[2 * c] = b0 * scB; tB.b[2 * c + 1] = b1 * scB;
                pipe.release();
            }
            TI_STAMP();
            r16::ln_silu(tA, vec + EV::P_G0 * F, vec + EV::P_BE0 * F, q, eps_p0);
            r16::ln_silu(tB, vec + EV::P_G0 * F, vec + EV::P_BE0 * F, q, eps_p0);
            inA.set(tA); inB.set(tB);
            TI_STAMP();
#pragma unroll
            for (int c = 0; c < NB; ++c) {
                f32x4 a0 = r16::load_block(vec + EV::P_B1 * F, 2 * c, q), a1 = r16::load_block(vec + EV::P_B1 * F, 2 * c + 1, q);
                f32x4 b0 = a0, b1 = a1;
                r16::gemm_x2_on_pipe<false>(a0, a1, b0, b1, inA, inB, pipe, lane);
                tA.b[2 * c] = a0; tA.b[2 * c + 1] = a1;
                tB.b[2 * c] = b0; tB.b[2 * c + 1] = b1;
                pipe.release();
            }
            TI_STAMP();
            r16::ln_silu(tA, vec + EV::P_G1 * F, vec + EV::P_BE1 * F, q, eps_p1);
            r16::ln_silu(tB, vec + EV::P_G1 * F, vec + EV::P_BE1 * F, q, eps_p1);
            h2A.set(tA); h2B.set(tB);
            TI_STAMP();
        }
        // ---- output layer, flipped: features on lanes (l & 15), pair rows 4q + r in registers: I slot q, J slots r
        uint32_t mi[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) mi[r] = p.rows[blk * 16 + 4 * q + r];
        f32x4 wfac;                                      // row mask x 1 / (S_phi S_w): rides on the shared w factor
#pragma unroll
        for (int r = 0; r < 4; ++r) wfac[r] = (mi[r] & 1u) ? inv_out : 0.0f;
        // partial-sum rows of this lane row: direction A's sums belong to J slot q (row 4 + q of the block's eight), direction B's to I slot q
#ifndef TI_PAIR_ACC_ATOMIC
#define TI_PAIR_ACC_ATOMIC 1      // 1 (default, measured 29.4 vs 31.5 ms same box): per-atom sums straight into dsacc / dvacc / cacc with fire-and-forget
                                  // atomics (first touch replaces), no reduction pass; 0: per-(block, slot) partial rows + pair_reduce_kernel
#endif
        float* const part_blk = p.part + ((size_t)gi * p.nblk + blk) * 8 * (7 * F);
        const int snJ = p.slotnode[blk * 16 + 4 + q], snI = p.slotnode[blk * 16 + q];
        const bool haveA = group_ok && snJ >= 0 && (!TI_PAIR_ACC_ATOMIC || gi * p.G + slot_mol(snJ) < p.B);
        const bool haveB = group_ok && snI >= 0 && (!TI_PAIR_ACC_ATOMIC || gi * p.G + slot_mol(snI) < p.B);
        float* const partA = part_blk + (size_t)(4 + q) * (7 * F);
        float* const partB = part_blk + (size_t)q * (7 * F);
        // atomic variant: the accumulator rows of the two destination atoms, laid out as three arrays (ds [F], dv [3F], c [3F] per node)
        const int qnA = (int)((gi * p.G + slot_mol(snJ)) * p.A) + (snJ & 255), qnB = (int)((gi * p.G + slot_mol(snI)) * p.A) + (snI & 255);
        const bool qfA = (snJ & SLOT_FIRST_TOUCH) != 0, qfB = (snI & SLOT_FIRST_TOUCH) != 0;
        auto acc_ptr = [&](int node, int off) {                // off as for the partial rows: ds 0.., dv F.., c 4F..   (node < 2^31 / (3 F))
            return off < F ? p.dsacc + (size_t)node * F + off : off < 4 * F ? p.dvacc + (size_t)node * 3 * F + (off - F) : p.cacc + (size_t)node * 3 * F + (off - 4 * F);
        };
        const long long nIq = node_of(prow_molI(mi[0]), prow_atomI(mi[0]));      // source of direction A for all four rows of this lane

        // (phi_c + b) of both directions times the shared (w_c + b) for output slice c (0 gates, 1 scale_edge_dir, 2 ds, 3 de,
        // 4 cross gates), features fo .. fo+31 as two 16-feature blocks
        auto out3 = [&](int c, int nbo, f32x4& rA0, f32x4& rA1, f32x4& rB0, f32x4& rB1) {
            f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0}, b0 = {0, 0, 0, 0}, b1 = {0, 0, 0, 0}, w0 = {0, 0, 0, 0}, w1 = {0, 0, 0, 0};
            r16::gemm_x2_on_pipe<true>(a0, a1, b0, b1, h2A, h2B, pipe, lane);
            pipe.release();
            r16::gemm_on_pipe<true>(w0, w1, g2, pipe, lane);
            pipe.release();
            const float* bp = vec + (EV::P_B2 + c) * F + 32 * nbo + j;
            const float* bw = vec + (EV::W_B2 + c) * F + 32 * nbo + j;
            w0 = (w0 + bw[0]) * wfac; w1 = (w1 + bw[16]) * wfac;
            rA0 = (a0 + bp[0]) * w0; rA1 = (a1 + bp[16]) * w1;
            rB0 = (b0 + bp[0]) * w0; rB1 = (b1 + bp[16]) * w1;
        };
        // direction A: sum over the four lane rows (the I slots) for each register (J slot); lane row q' ends with J slot q'
        auto sumA = [&](const f32x4& v) {
            using QS = r16::QuarterSum<4>;
            return QS::swap32_add(QS::swap16_add(v[0], v[1]), QS::swap16_add(v[2], v[3]));
        };
        // off: offset of the quantity inside a partial row (ds 0, dv (1 + c) F, c (4 + c) F) plus the lane's feature
        auto putA = [&](float z0, float z1, int off) {
            if (TI_PAIR_ACC_ATOMIC) { if (haveA) { float* d = acc_ptr(qnA, off); acc_out(d, z0, qfA); acc_out(d + 16, z1, qfA); } }
            else if (haveA) { partA[off] = z0; partA[off + 16] = z1; }
        };
        auto putB = [&](float z0, float z1, int off) {
            if (TI_PAIR_ACC_ATOMIC) { if (haveB) { float* d = acc_ptr(qnB, off); acc_out(d, z0, qfB); acc_out(d + 16, z1, qfB); } }
            else if (haveB) { partB[off] = z0; partB[off + 16] = z1; }
        };
        // direction B: the four registers of a lane are the J slots of ONE destination I[q]
        auto sumB = [&](const f32x4& v) { return (v[0] + v[1]) + (v[2] + v[3]); };
        auto emitA = [&](const f32x4& v0, const f32x4& v1, int off) { putA(sumA(v0), sumA(v1), off); };
        auto emitB = [&](const f32x4& v0, const f32x4& v1, int off) { putB(sumB(v0), sumB(v1), off); };
        // the value of lane row r' in every lane row, r' = 0 .. 3 (VALU lane swaps, mfma_chain.hpp)
        auto rows4 = [&](float x, float (&o)[4]) {
            float a = x, b = x;
            lane_swap16(a, b);                       // a = [X0 X0 X2 X2], b = [X1 X1 X3 X3]
            o[0] = a; o[2] = a; lane_swap32(o[0], o[2]);
            o[1] = b; o[3] = b; lane_swap32(o[1], o[3]);
        };
        // v[src] rows of the equivariant slice.  Split path: every VMEM load of the slice is issued and consumed BEFORE its first accumulator
        // atomic, and direction B's four source atoms J[r] are fetched once, by lane row r, and handed round with lane swaps.  Loads and
        // writes share vmcnt but complete out of order with respect to each other, so waiting for a load while atomics are in flight
        // costs `s_waitcnt vmcnt(0)`, the drain of those atomics -- with the gathers between the slice's atomics (three rounds per 32
        // features) the launch took 29.60 ms, this way 28.23 (profiles/r03i_dv_reorder_timing.txt; the r03c stamps had shown the slice
        // at 11 - 13 k cycles against 3 - 5 k for the others).  The f32 path is bound by its matrix instructions and short of registers:
        // it keeps the gathers next to their use.
        constexpr bool GATHER_EARLY = PREC != 0;
        const long long nJq = snJ >= 0 ? node_of(slot_mol(snJ), snJ & 255) : nIq;   // J slot q's atom: lane row q fetches it for all four
        const float wrow = (meta & 1u) ? inv_out : 0.0f;                 // the same row factor in the row layout (lane (j, q): row j)

#pragma unroll 1
        for (int nbo = 0; nbo < NB; ++nbo) {
            const int fo = 32 * nbo + j;
            TI_STAMP();
            {   // ds: invariant message, summed over incoming edges
                f32x4 a0, a1, b0, b1;
                out3(2, nbo, a0, a1, b0, b1);
                emitA(a0, a1, fo);
                emitB(b0, b1, fo);
            }
            TI_STAMP();
            if constexpr (!LAST) {
                // de: edge state update e += de of both directions in the ROW layout (lane (j, q): row j, features 16 (2 nbo) + 4q .. and
                // 16 (2 nbo + 1) + 4q ..): the same two chunks with the operands the other way round; the old row is loaded before the
                // products and stored after them -- one owner per row, no atomics
                f32x4 a0 = r16::load_block(vec + (EV::P_B2 + 3) * F, 2 * nbo, q), a1 = r16::load_block(vec + (EV::P_B2 + 3) * F, 2 * nbo + 1, q);
                f32x4 b0 = a0, b1 = a1;
                f32x4 w0 = r16::load_block(vec + (EV::W_B2 + 3) * F, 2 * nbo, q), w1 = r16::load_block(vec + (EV::W_B2 + 3) * F, 2 * nbo + 1, q);
                float* const ea = p.e + (erowA + j) * F;
                float* const eb = p.e + (erowB + j) * F;
                f32x4 oA0, oA1, oB0, oB1;
                if (FIRST) {
                    const float* em = p.edge_emb + prow_type(meta) * F;
                    oA0 = r16::load_block(em, 2 * nbo, q); oA1 = r16::load_block(em, 2 * nbo + 1, q);
                    oB0 = oA0; oB1 = oA1;
                } else {
                    oA0 = r16::load_block(ea, 2 * nbo, q); oA1 = r16::load_block(ea, 2 * nbo + 1, q);
                    oB0 = r16::load_block(eb, 2 * nbo, q); oB1 = r16::load_block(eb, 2 * nbo + 1, q);
                }
                r16::gemm_x2_on_pipe<false>(a0, a1, b0, b1, h2A, h2B, pipe, lane);
                pipe.release();
                r16::gemm_on_pipe<false>(w0, w1, g2, pipe, lane);
                pipe.release();
                w0 *= wrow; w1 *= wrow;
                if (group_ok) {
                    r16::store_block(ea, 2 * nbo, q, oA0 + a0 * w0); r16::store_block(ea, 2 * nbo + 1, q, oA1 + a1 * w1);
                    r16::store_block(eb, 2 * nbo, q, oB0 + b0 * w0); r16::store_block(eb, 2 * nbo + 1, q, oB1 + b1 * w1);
                }
            }
            TI_STAMP();
            {   // equivariant message: sum_e (sed * dir_e + gates * v[src_e]) -> dvacc ; sum_e cg * dir_e -> cacc
                f32x4 sA0, sA1, sB0, sB1, gA0 = {0, 0, 0, 0}, gA1 = {0, 0, 0, 0}, gB0 = {0, 0, 0, 0}, gB1 = {0, 0, 0, 0};
                out3(1, nbo, sA0, sA1, sB0, sB1);
                float vI[3][2], vJ[3][2];                // v of I[q] (source of direction A for the lane's four rows) and of J[q]
                if (!FIRST) {
                    const float* vp = p.v + (size_t)nIq * 3 * F + fo;
                    const float* vq = p.v + (size_t)nJq * 3 * F + fo;
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        vI[c][0] = vp[c * F]; vI[c][1] = vp[c * F + 16];
                        if (GATHER_EARLY) { vJ[c][0] = vq[c * F]; vJ[c][1] = vq[c * F + 16]; }
                    }
                    out3(0, nbo, gA0, gA1, gB0, gB1);
                }
                f32x4 dir[4];                            // edge_dir of direction A; direction B's is its negative
#pragma unroll
                for (int r = 0; r < 4; ++r) dir[r] = *reinterpret_cast<const f32x4*>(scratch + (4 * q + r) * 4);
                float zA[3][2], zB[3][2];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    f32x4 v0, v1;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v0[r] = sA0[r] * dir[r][c]; v1[r] = sA1[r] * dir[r][c];
                        if (!FIRST) { v0[r] = fmaf(gA0[r], vI[c][0], v0[r]); v1[r] = fmaf(gA1[r], vI[c][1], v1[r]); }
                    }
                    zA[c][0] = sumA(v0); zA[c][1] = sumA(v1);
                    if (!GATHER_EARLY) putA(zA[c][0], zA[c][1], (1 + c) * F + fo);
                }
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    f32x4 v0, v1;
                    float j0[4], j1[4];                  // v[src] of direction B: the J atom of each row
                    if (!FIRST) {
                        if (GATHER_EARLY) { rows4(vJ[c][0], j0); rows4(vJ[c][1], j1); }
                        else
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float* vr = p.v + (size_t)node_of(prow_molJ(mi[r]), prow_atomJ(mi[r])) * 3 * F + c * F + fo;
                                j0[r] = vr[0]; j1[r] = vr[16];
                            }
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v0[r] = -(sB0[r] * dir[r][c]); v1[r] = -(sB1[r] * dir[r][c]);
                        if (!FIRST) { v0[r] = fmaf(gB0[r], j0[r], v0[r]); v1[r] = fmaf(gB1[r], j1[r], v1[r]); }
                    }
                    zB[c][0] = sumB(v0); zB[c][1] = sumB(v1);
                    if (!GATHER_EARLY) putB(zB[c][0], zB[c][1], (1 + c) * F + fo);
                }
                if (GATHER_EARLY)
#pragma unroll
                    for (int c = 0; c < 3; ++c) { putA(zA[c][0], zA[c][1], (1 + c) * F + fo); putB(zB[c][0], zB[c][1], (1 + c) * F + fo); }
                TI_STAMP();
                if (!FIRST) {
                    f32x4 cA0, cA1, cB0, cB1;
                    out3(4, nbo, cA0, cA1, cB0, cB1);
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        f32x4 v0, v1;
#pragma unroll
                        for (int r = 0; r < 4; ++r) { v0[r] = cA0[r] * dir[r][c]; v1[r] = cA1[r] * dir[r][c]; }
                        emitA(v0, v1, (4 + c) * F + fo);
                    }
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        f32x4 v0, v1;
#pragma unroll
                        for (int r = 0; r < 4; ++r) { v0[r] = -(cB0[r] * dir[r][c]); v1[r] = -(cB1[r] * dir[r][c]); }
                        emitB(v0, v1, (4 + c) * F + fo);
                    }
                }
            }
        }
    }
    pipe.drain();
#ifdef TI_STAMPS
    if (p.stamps && lane == 0) {          // whole-loop clock pair of every wave: [4 * WAVES * STAMP_SLOTS + 2 * wave id ...]
        const unsigned long long clk1 = __builtin_amdgcn_s_memtime(), rt1 = __builtin_amdgcn_s_memrealtime();
        unsigned long long* c = p.stamps + 2048 + 2 * (size_t)gi_raw;
        c[0] = clk1 - clk0; c[1] = rt1 - rt0;
    }
#endif
}

template <int NB, int EW, int PREC>
static hipError_t configure_pair_prec()
{
    if constexpr (!pair_build_exists(NB, EW, PREC)) return hipSuccess;
    else {
    const size_t be = pair_lds_bytes(NB, EW);
    hipError_t e;
    if ((e = set_lds_edge(painn_pair_kernel<2 * NB, true, false, PREC, EW>, be)) != hipSuccess) return e;
    if ((e = set_lds_edge(painn_pair_kernel<2 * NB, false, false, PREC, EW>, be)) != hipSuccess) return e;
    if ((e = set_lds_edge(painn_pair_kernel<2 * NB, false, true, PREC, EW>, be)) != hipSuccess) return e;
    if ((e = set_lds_edge(painn_pair_kernel<2 * NB, true, true, PREC, EW>, be)) != hipSuccess) return e;
    return hipSuccess;
    }
}
template <int NB>
static hipError_t configure_pair_nb()
{
    hipError_t e;
    if ((e = configure_pair_prec<NB, 4, 0>()) != hipSuccess) return e;
    if ((e = configure_pair_prec<NB, 4, 1>()) != hipSuccess) return e;
    return configure_pair_prec<NB, 8, 1>();
}

template <int NB, int EW, int PREC>
static void launch_pair_p(bool first, bool last, const EdgeParams& p, hipStream_t st)
{
    if constexpr (pair_build_exists(NB, EW, PREC)) {
    const dim3 g((unsigned)((p.n_groups + EW - 1) / EW)), t(64 * EW);          // one wave = one group of G molecules
    const size_t l = pair_lds_bytes(NB, EW);
    if (first && last) hipLaunchKernelGGL((painn_pair_kernel<2 * NB, true, true, PREC, EW>), g, t, l, st, p);
    else if (first) hipLaunchKernelGGL((painn_pair_kernel<2 * NB, true, false, PREC, EW>), g, t, l, st, p);
    else if (last) hipLaunchKernelGGL((painn_pair_kernel<2 * NB, false, true, PREC, EW>), g, t, l, st, p);
    else hipLaunchKernelGGL((painn_pair_kernel<2 * NB, false, false, PREC, EW>), g, t, l, st, p);
    }
}
static bool pair_writes_partials() { return !TI_PAIR_ACC_ATOMIC; }
template <int NB>
static hipError_t launch_pair_nb(bool first, bool last, int prec, const EdgeParams& p, hipStream_t st)
{
    if (prec != 0 && prec != 1) return hipErrorInvalidValue;
    // 8-wave workgroups (one weight stream per CU, 4-chunk superchunks at F = 128) for the split path once every CU gets a workgroup
    const bool wide = prec == 1 && p.n_groups >= 2048;
    if (wide) launch_pair_p<NB, 8, 1>(first, last, p, st);
    else if (prec == 1) launch_pair_p<NB, 4, 1>(first, last, p, st);
    else launch_pair_p<NB, 4, 0>(first, last, p, st);
    return hipGetLastError();
}

}  // namespace ti
