// ti_api.hip -- host side of libti_hip.so: C ABI (include/ti_hip.h), weight packing, edge templates, HBM workspace,
// drift / rollout orchestration on one HIP stream, live kernel timing with HIP events.
//
// There is deliberately no CPU fallback in this file: every entry point either runs the HIP kernels or fails.
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <utility>

#include "pair_template.hpp"
#include "ti_internal.hpp"

namespace ti {

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }

struct HipError : std::runtime_error { using std::runtime_error::runtime_error; };
#define HIP_CHECK(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) \
    throw HipError(std::string(#expr) + ": " + hipGetErrorString(e__)); } while (0)

// ---------------------------------------------------------------------------------------------------- packing
// 16-row kernels (mfma_chain.hpp, namespace r16): chunk[(blk*NBK + nbi)*64 + l][r] = W[row0 + 16*blk + (l&15)][col0 + 16*nbi + 4*(l>>4) + r]
void pack_chunk16(std::vector<float>& dst, const float* W, int ld, int n_rows, int row0, int col0, int NBK)
{
    const size_t base = dst.size();
    dst.resize(base + (size_t)2 * NBK * 64 * 4);
    for (int blk = 0; blk < 2; ++blk)
        for (int nbi = 0; nbi < NBK; ++nbi)
            for (int l = 0; l < 64; ++l)
                for (int r = 0; r < 4; ++r) {
                    const int row = row0 + 16 * blk + (l & 15), col = col0 + 16 * nbi + 4 * (l >> 4) + r;
                    dst[base + ((size_t)(blk * NBK + nbi) * 64 + l) * 4 + r] = row < n_rows ? W[(size_t)row * ld + col] : 0.f;
                }
}

// split-fp16 chunk of the edge kernel (mfma_chain.hpp, r16::Opnd<NBK,true>): 16-byte h8 fragments
//   frag[((blk*(NBK/2) + m)*2 + {0 hi, 1 lo})*64 + l][i] = half of W[row0 + 16*blk + (l&15)][col0 + 16*(2m + (i>>2)) + 4*(l>>4) + (i&3)]
// with hi = fp16(w), lo = fp16((w - hi) * 2^11).  Same byte size as the fp32 chunk.
void pack_chunk16_split(std::vector<float>& dst, const float* W, int ld, int n_rows, int row0, int col0, int NBK)
{
    const size_t base = dst.size();
    dst.resize(base + (size_t)2 * NBK * 64 * 4);
    _Float16* out = reinterpret_cast<_Float16*>(dst.data() + base);
    const int KS = NBK / 2;
    for (int blk = 0; blk < 2; ++blk)
        for (int m = 0; m < KS; ++m)
            for (int l = 0; l < 64; ++l)
                for (int i = 0; i < 8; ++i) {
                    const int row = row0 + 16 * blk + (l & 15), col = col0 + 16 * (2 * m + (i >> 2)) + 4 * (l >> 4) + (i & 3);
                    const float w = row < n_rows ? W[(size_t)row * ld + col] : 0.f;
                    const _Float16 h = (_Float16)w;
                    const _Float16 lo = (_Float16)((w - (float)h) * 2048.0f);
                    out[((size_t)((blk * KS + m) * 2 + 0) * 64 + l) * 8 + i] = h;
                    out[((size_t)((blk * KS + m) * 2 + 1) * 64 + l) * 8 + i] = lo;
                }
}

// one-accumulator format of the message kernel (mfma_chain.hpp: Opnd1 / gemm_split_chunk1): the matrix is scaled by the power of two S
// first and the residual is NOT scaled:  hi = fp16(S w), lo = fp16(S w - hi).  Same layout and size as pack_chunk16_split.
void pack_chunk16_split1(std::vector<float>& dst, const float* W, int ld, int n_rows, int row0, int col0, int NBK, float S)
{
    const size_t base = dst.size();
    dst.resize(base + (size_t)2 * NBK * 64 * 4);
    _Float16* out = reinterpret_cast<_Float16*>(dst.data() + base);
    const int KS = NBK / 2;
    for (int blk = 0; blk < 2; ++blk)
        for (int m = 0; m < KS; ++m)
            for (int l = 0; l < 64; ++l)
                for (int i = 0; i < 8; ++i) {
                    const int row = row0 + 16 * blk + (l & 15), col = col0 + 16 * (2 * m + (i >> 2)) + 4 * (l >> 4) + (i & 3);
                    const float w = (row < n_rows ? W[(size_t)row * ld + col] : 0.f) * S;
                    const _Float16 h = (_Float16)w;
                    out[((size_t)((blk * KS + m) * 2 + 0) * 64 + l) * 8 + i] = h;
                    out[((size_t)((blk * KS + m) * 2 + 1) * 64 + l) * 8 + i] = (_Float16)(w - (float)h);
                }
}
// power of two that brings the largest |entry| of W[0..rows)[col0..col0+cols) into [2^13, 2^14)   (1 for an all-zero matrix).
// `bias` (n_bias entries, may be NULL) is the vector that is multiplied by the same factor in the kernel (the layer's bias row, or for
// phi's first Linear the per-atom P = s W^T + b it is added to): the factor is capped so that S |b| stays <= 2^14 -- a near-zero matrix
// beside O(1) biases (a pruned or freshly initialised layer) would otherwise scale the biases to 1e18 and overflow the LayerNorm's sum
// of squares; what the cap costs is precision of a product that is negligible beside that bias anyway.
float matrix_pow2_scale(const float* W, int ld, int rows, int col0, int cols, const float* bias = nullptr, int n_bias = 0)
{
    float mx = 0.f, mb = 0.f;
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) mx = std::max(mx, std::fabs(W[(size_t)r * ld + col0 + c]));
    for (int i = 0; i < n_bias; ++i) mb = std::max(mb, std::fabs(bias[i]));
    if (!(mx > 0.f) || !std::isfinite(mx)) return 1.0f;
    int e; std::frexp(mx, &e);                      // mx = f * 2^e, f in [0.5, 1)
    int k = std::min(60, std::max(-60, 14 - e));
    if (mb > 0.f && std::isfinite(mb)) { int eb; std::frexp(mb, &eb); k = std::min(k, std::max(-60, 14 - eb)); }
    return std::ldexp(1.0f, k);
}

// fp16 storage mode (r16::OpndH): the hi fragments alone, half the bytes:  frag[(blk*(NBK/2) + m)*64 + l][i]
void pack_chunk16_half(std::vector<float>& dst, const float* W, int ld, int n_rows, int row0, int col0, int NBK)
{
    const size_t base = dst.size();
    dst.resize(base + (size_t)NBK * 64 * 4);
    _Float16* out = reinterpret_cast<_Float16*>(dst.data() + base);
    const int KS = NBK / 2;
    for (int blk = 0; blk < 2; ++blk)
        for (int m = 0; m < KS; ++m)
            for (int l = 0; l < 64; ++l)
                for (int i = 0; i < 8; ++i) {
                    const int row = row0 + 16 * blk + (l & 15), col = col0 + 16 * (2 * m + (i >> 2)) + 4 * (l >> 4) + (i & 3);
                    out[((size_t)(blk * KS + m) * 64 + l) * 8 + i] = (_Float16)(row < n_rows ? W[(size_t)row * ld + col] : 0.f);
                }
}

struct MlpOff { size_t W0, b0, g0, be0, W1, b1, g1, be1, W2, b2; int f_in, f_h, f_out; };
static size_t take_mlp(MlpOff& m, size_t o, int f_in, int f_h, int f_out)
{
    m.f_in = f_in; m.f_h = f_h; m.f_out = f_out;
    m.W0 = o; o += (size_t)f_h * f_in; m.b0 = o; o += f_h; m.g0 = o; o += f_h; m.be0 = o; o += f_h;
    m.W1 = o; o += (size_t)f_h * f_h;  m.b1 = o; o += f_h; m.g1 = o; o += f_h; m.be1 = o; o += f_h;
    m.W2 = o; o += (size_t)f_out * f_h; m.b2 = o; o += f_out;
    return o;
}

template <typename T>
struct DevBuf {
    T* p = nullptr; size_t n = 0;
    void alloc(size_t count) { release(); if (count) { HIP_CHECK(hipMalloc((void**)&p, count * sizeof(T))); n = count; } }
    void upload(const std::vector<T>& h) { alloc(h.size()); if (n) HIP_CHECK(hipMemcpy(p, h.data(), n * sizeof(T), hipMemcpyHostToDevice)); }
    void release() { if (p) { (void)hipFree(p); p = nullptr; n = 0; } }
    ~DevBuf() { release(); }
};

struct Stream { size_t off4; int nch; };      // offset into the packed buffer in float4 units

}  // namespace ti

using namespace ti;

// ====================================================================================================== handle
struct ti_handle {
    int kind = 0;                 // 0 painn, 1 adw
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    hipEvent_t wait_ev = nullptr;          // ti_wait_stream
    // profiling
    bool prof = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev[TI_KERNEL_COUNT];
    // common device buffers
    DevBuf<float> flat, packed;
    DevBuf<int> nanflag;
    long long cap = 0;            // trajectories the workspace is sized for

    // ---- painn
    ti_painn_desc d{};
    int NB = 0, nE = 0, ncond = 0, G = 1, nblk = 0;
    MlpOff embed{}, readout{}; std::vector<MlpOff> phi, w, upd; std::vector<size_t> U, V;
    size_t edge_emb = 0, atom_emb = 0, Vr = 0; float b2_gate = 0.f;
    Stream st_embed16{}; std::vector<Stream> st_edge, st_edge1, st_update;      // st_edge1: the message kernel's one-accumulator format (TI_PREC_F16X2)
    // edge templates (ti_internal.hpp): [0] throughput (G molecules per group), [1] latency (G = 1, P parts per molecule);
    // G / P / nblk / rows / slotnode below are those of the ACTIVE one (select_template, once per API call)
    struct Tpl {
        int G = 1, P = 1, nblk = 0;
        DevBuf<uint32_t> rows; DevBuf<int32_t> slotnode;
        std::vector<int> part_of, part_start, part_len;     // per sorted edge: its part; per part: first sorted edge, edges
        std::vector<int> pos;                     // row of (molecule-in-group m, sorted edge k) inside its part: pos[m * part_len + (k - part_start)]
        int max_slots = 0;                        // most destination atoms in any row block (<= EDGE_MAX_SLOTS)
    } tpl[3];
    // tpl[2]: the pair-major template (ti_internal.hpp; painn_pair_kernel.hpp), built when the graph is symmetric and the pair kernel
    // exists for this width / precision.  pair_pos[(m * A + src) * A + dst] = e row of that directed edge of molecule-in-group m inside
    // its group: (block * 2 + direction) * 16 + pair row
    bool has_pair = false; std::vector<int> pair_pos; double pair_fill = 0.0;
    DevBuf<int32_t> pair_plist; int pair_kmax = 0;     // per atom of a group: its partial-sum rows (pair_template.hpp)
    DevBuf<float> part;                                // [groups * nblk][8][7 F] partial sums of the pair-major kernel
    int n_tpl = 1, active = 0, parts = 1, max_slots = 0, pinned_tpl = TI_TEMPLATE_AUTO;
    // every atom has incoming edges: the edge kernels' first touch of an accumulator replaces its contents (ti_internal.hpp
    // SLOT_FIRST_TOUCH) and nothing zeroes the accumulators between layers or calls; otherwise the update kernel zeroes them as before
    bool first_touch = false;
    struct { const uint32_t* p = nullptr; } rows; struct { const int32_t* p = nullptr; } slotnode;
    DevBuf<int32_t> atom_ids;
    std::vector<int> perm;        // sorted row -> original edge index
    std::vector<int32_t> esrc, edst;       // the molecule's directed edges as passed to create
    DevBuf<float> x, cond, s, P, v, dsacc, dvacc, cacc, e, enc, geo, b1, b2, xt, edge_vecs, edge_vecs1, upd_vecs;
    std::vector<float> edge_scale;               // [L][6] per-matrix powers of two of the one-accumulator message streams (TI_PREC_F16X2)
    int tap = -1; long long last_B = 0;
    // forward-mode derivative (painn_jvp_kernels.hip): tangent twins over virtual molecules, sized on first use
    std::vector<Stream> st_jvp_update, st_jvp_phi; Stream st_jvp_readout{}; std::vector<int> jvp_phi_pad;
    DevBuf<float> jvp_ro_vecs, ts, tP, tv, tdsacc, tdvacc, tcacc, te, tout, wq, phist, nodest, divb, dl, dlscaled, div2;
    long long jvp_cap = 0, last_VB = 0; int last_D = 1;
    // Runge-Kutta drivers (rollout_rk): stage derivatives, dense-output coefficients, reduction scratch
    DevBuf<float> rk_ws; DevBuf<double> rk_red;

    // ---- adw
    ti_adw_desc ad{};
    size_t a_be_vecs = 0, a_net_vecs = 0;          // offsets of the per-MLP vector blocks in `flat`
    float a_be_b_out = 0.f, a_b_out = 0.f;
    Stream st_be{}, st_net{};
    DevBuf<float> ax, ab1, ab2, axt, aemb_u, abeta0_u, abeta1_u, adl, ad1, ad2; DevBuf<int32_t> aidx;

    ~ti_handle()
    {
        for (auto& v2 : ev) for (auto& pr : v2) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
        if (wait_ev) (void)hipEventDestroy(wait_ev);
        if (own_stream) (void)hipStreamDestroy(own_stream);
    }
    const float* F(size_t off) const { return flat.p + off; }
    const float4* S(const Stream& s2) const { return reinterpret_cast<const float4*>(packed.p) + s2.off4; }
    MlpVec vec(const MlpOff& m) const { return MlpVec{F(m.b0), F(m.g0), F(m.be0), F(m.b1), F(m.g1), F(m.be1), F(m.b2)}; }
};

namespace {

struct Timed {           // RAII: brackets a launch with HIP events when profiling is on
    ti_handle* h; int slot; hipEvent_t a = nullptr, b = nullptr;
    Timed(ti_handle* h_, int slot_) : h(h_), slot(slot_)
    {
        if (!h->prof) return;
        HIP_CHECK(hipEventCreate(&a)); HIP_CHECK(hipEventCreate(&b));
        HIP_CHECK(hipEventRecord(a, h->stream));
    }
    ~Timed()
    {
        if (!a) return;
        (void)hipEventRecord(b, h->stream);
        h->ev[slot].emplace_back(a, b);
    }
};

void set_device(const ti_handle* h) { HIP_CHECK(hipSetDevice(h->device)); }

// ------------------------------------------------------------------------------------------------ painn create
// Packs the rows of `count` molecules' edges with sorted position in [k0, k1) into 16-row blocks: rows keep their (molecule, dst,
// src) order; a block takes at most EDGE_MAX_SLOTS destination atoms (a further one starts the next block, the rest is padding).
// pos[m * (k1 - k0) + (k - k0)] = row of that edge inside the part; returns the number of blocks used.
static int pack_part(const ti_handle* h, const int32_t* dst, int count, int k0, int k1, std::vector<int>& pos)
{
    constexpr int RB = ti::EDGE_ROWS_PER_BLOCK;
    const int per = k1 - k0;
    pos.assign((size_t)count * per, 0);
    int row = 0, nslot = 0, last_key = -1;
    for (int m = 0; m < count; ++m)
        for (int k = k0; k < k1; ++k) {
            const int key = m * 256 + dst[h->perm[k]];
            if (row % RB == 0) { nslot = 0; last_key = -1; }
            if (key != last_key) {
                if (nslot == ti::EDGE_MAX_SLOTS) { row = (row + RB - 1) / RB * RB; nslot = 0; }
                ++nslot; last_key = key;
            }
            pos[(size_t)m * per + (k - k0)] = row++;
        }
    return (row + RB - 1) / RB;
}

// row / slot words of a packed part (ti_internal.hpp); returns the most slots any block holds
static int fill_part(const ti_handle* h, const int32_t* src, const int32_t* dst, const int32_t* etype, int count, int k0, int k1, int nblk,
                     const std::vector<int>& pos, uint32_t* rw, int32_t* sn)
{
    constexpr int RB = ti::EDGE_ROWS_PER_BLOCK;
    const int per = k1 - k0;
    for (int i = 0; i < nblk * RB; ++i) { rw[i] = (uint32_t)63 << 18; sn[i] = -1; }
    std::vector<int> nslot(std::max(nblk, 1), 0), last_key(std::max(nblk, 1), -1);
    std::vector<char> touched((size_t)count * 256, 0);          // rows are visited in increasing order: the first block seen is the first executed
    int most = 0;
    for (int m = 0; m < count; ++m)
        for (int kk = 0; kk < per; ++kk) {
            const int r = pos[(size_t)m * per + kk], blk = r / RB, k = h->perm[k0 + kk];
            const int key = m * 256 + dst[k];
            if (key != last_key[blk]) {
                const int32_t first = (h->first_touch && !touched[key]) ? ti::SLOT_FIRST_TOUCH : 0;
                touched[key] = 1;
                sn[(size_t)blk * RB + nslot[blk]] = first | (m << 8) | dst[k]; ++nslot[blk]; last_key[blk] = key;
            }
            most = std::max(most, nslot[blk]);
            rw[r] = 1u | ((uint32_t)m << 1) | ((uint32_t)src[k] << 6) | ((uint32_t)dst[k] << 11) | ((uint32_t)etype[k] << 16) |
                    ((uint32_t)(nslot[blk] - 1) << 18);
        }
    return most;
}

void build_templates(ti_handle* h, const int32_t* src, const int32_t* dst, const int32_t* etype)
{
    const int A = h->d.n_atoms, E = h->d.n_edges;
    constexpr int RB = ti::EDGE_ROWS_PER_BLOCK;
    {
        std::vector<char> has_in(A, 0);
        for (int k = 0; k < E; ++k) has_in[dst[k]] = 1;
        h->first_touch = E > 0 && std::all_of(has_in.begin(), has_in.end(), [](char c) { return c != 0; });
        // triage switch: TI_ZERO_ACC=1 at creation forces the zeroing path (memsets before an evaluation, the update kernel clears
        // what it consumed, every accumulator update is an add) -- tests/test_gpu_pair.py compares the two paths on poisoned accumulators
        if (const char* z = std::getenv("TI_ZERO_ACC")) if (z[0] == '1') h->first_touch = false;
    }
    h->perm.resize(E);
    for (int k = 0; k < E; ++k) h->perm[k] = k;
    std::stable_sort(h->perm.begin(), h->perm.end(), [&](int a, int b) {
        return dst[a] != dst[b] ? dst[a] < dst[b] : src[a] < src[b];
    });
    // ---- throughput template: smallest G in 1..8 whose padding waste is <= 2 %, else the least wasteful
    {
        int bestG = 1; double bestW = 2.0;
        std::vector<int> pos;
        for (int G = 1; G <= 8 && E > 0; ++G) {
            const int rows = G * E, padded = pack_part(h, dst, G, 0, E, pos) * RB;
            const double waste = double(padded - rows) / padded;
            if (waste < bestW - 1e-12) { bestW = waste; bestG = G; }
            if (waste <= 0.02) { bestG = G; break; }
        }
        ti_handle::Tpl& T = h->tpl[0];
        T.G = bestG; T.P = 1;
        T.nblk = E > 0 ? pack_part(h, dst, bestG, 0, E, T.pos) : 0;
        T.part_of.assign(E, 0); T.part_start.assign(1, 0); T.part_len.assign(1, E);
        std::vector<uint32_t> rw((size_t)std::max(T.nblk, 1) * RB); std::vector<int32_t> sn(rw.size());
        T.max_slots = fill_part(h, src, dst, etype, bestG, 0, E, T.nblk, T.pos, rw.data(), sn.data());
        if (T.nblk == 0) { rw[0] = (uint32_t)63 << 18; sn[0] = -1; }
        T.rows.upload(rw); T.slotnode.upload(sn);
    }
    // ---- latency template: one molecule per group, its destination atoms cut into P ranges of near-equal row count; the
    // largest P <= 8 whose padding waste stays <= 15 % (parts need whole row blocks).  Built only if it offers more waves.
    h->n_tpl = 1;
    if (E >= 2 * RB) {
        std::vector<int> first_of(A + 1, E);           // first sorted edge with destination >= a
        for (int k = E - 1; k >= 0; --k) first_of[dst[h->perm[k]]] = k;
        for (int a = A - 1; a >= 0; --a) first_of[a] = std::min(first_of[a], first_of[a + 1]);
        std::vector<int> pos;
        int bestP = 1, best_nblk = pack_part(h, dst, 1, 0, E, pos); std::vector<int> best_cut{0, E};
        for (int P = 2; P <= 8; ++P) {
            std::vector<int> cut{0};
            for (int q = 1; q < P; ++q) {             // atom boundary closest to q/P of the rows
                const int want = (int)((long long)E * q / P);
                int bk = cut.back();
                for (int a = 0; a <= A; ++a) if (first_of[a] > cut.back() && std::abs(first_of[a] - want) < std::abs(bk - want)) bk = first_of[a];
                if (bk <= cut.back()) { cut.clear(); break; }
                cut.push_back(bk);
            }
            if (cut.empty() || cut.back() >= E) continue;
            cut.push_back(E);
            int nblk = 0;
            for (int q = 0; q < P; ++q) nblk = std::max(nblk, pack_part(h, dst, 1, cut[q], cut[q + 1], pos));
            if (double(P * nblk * RB - E) / (P * nblk * RB) <= 0.15) { bestP = P; best_nblk = nblk; best_cut = cut; }
        }
        if (bestP * h->tpl[0].G > 1) {
            ti_handle::Tpl& T = h->tpl[1];
            T.G = 1; T.P = bestP; T.nblk = best_nblk; T.max_slots = 0;
            T.part_start.assign(best_cut.begin(), best_cut.end() - 1);
            T.part_of.resize(E); T.part_len.resize(bestP); T.pos.assign(E, 0);
            for (int q = 0; q < bestP; ++q) for (int k = best_cut[q]; k < best_cut[q + 1]; ++k) T.part_of[k] = q;
            std::vector<uint32_t> rw((size_t)bestP * best_nblk * RB); std::vector<int32_t> sn(rw.size());
            for (int q = 0; q < bestP; ++q) {
                T.part_len[q] = best_cut[q + 1] - best_cut[q];
                pack_part(h, dst, 1, best_cut[q], best_cut[q + 1], pos);
                std::copy(pos.begin(), pos.end(), T.pos.begin() + best_cut[q]);
                T.max_slots = std::max(T.max_slots, fill_part(h, src, dst, etype, 1, best_cut[q], best_cut[q + 1], best_nblk, pos,
                                                              rw.data() + (size_t)q * best_nblk * RB, sn.data() + (size_t)q * best_nblk * RB));
            }
            T.rows.upload(rw); T.slotnode.upload(sn);
            h->n_tpl = 2;
        }
    }
    h->esrc.assign(src, src + E); h->edst.assign(dst, dst + E);
}


// ---- pair-major template: pair_template.hpp builds it (pure host code, unit-tested on the CPU); this uploads it
static bool build_pair_template(ti_handle* h, const int32_t* src, const int32_t* dst, const int32_t* etype)
{
    ti::PairTemplate pt;
    if (!ti::build_pair_template(h->d.n_atoms, h->d.n_edges, src, dst, etype, pt, h->first_touch && !pair_uses_partials())) return false;
    ti_handle::Tpl& T = h->tpl[2];
    T.G = pt.G; T.P = 1; T.nblk = pt.nblk; T.max_slots = 4;
    T.rows.upload(pt.rows); T.slotnode.upload(pt.slotnode);
    h->pair_pos = pt.pair_pos; h->pair_fill = pt.fill;
    h->pair_plist.upload(pt.plist); h->pair_kmax = pt.kmax;
    return true;
}

// Template for a call over B molecules: the latency template while the throughput one would leave SIMDs without a wave
// (fewer groups than the 1024 SIMDs of the chip); TI_TEMPLATE=throughput|latency pins it (tests, reproducibility across shards).
int template_for(const ti_handle* h, long long B, bool allow_pair = true)
{
    int dir_pick = 0;                        // among the directed layouts: latency while the throughput one would leave SIMDs idle
    if (h->n_tpl > 1) dir_pick = (B + h->tpl[0].G - 1) / h->tpl[0].G < 1024 ? 1 : 0;
    const bool pair_ok = h->has_pair && allow_pair;
    // pair-major rows once they fill the chip (one wave per group of G molecules) and cost less than the directed rows: a pair block
    // runs 84 chunk products for 16 pairs where a directed block runs 56 for 16 edges, at half the weight-chunk visits per edge
    int pick = dir_pick;
    if (pair_ok) {
        const ti_handle::Tpl &T = h->tpl[2], &D = h->tpl[0];
        if ((B + T.G - 1) / T.G >= 1024 && 1.5 * T.nblk / T.G <= (double)D.nblk / D.G) pick = 2;
    }
    int want = h->pinned_tpl;
    if (const char* e = std::getenv("TI_TEMPLATE"))
        want = std::strcmp(e, "latency") == 0 ? 1 : std::strcmp(e, "throughput") == 0 ? 0 : std::strcmp(e, "pair") == 0 ? 2 : want;
    if (want == TI_TEMPLATE_THROUGHPUT) pick = 0;
    else if (want == TI_TEMPLATE_LATENCY) pick = h->n_tpl > 1 ? 1 : 0;
    else if (want == TI_TEMPLATE_PAIR) pick = pair_ok ? 2 : dir_pick;
    return pick;
}

// allow_pair = false: the divergence / tangent entry points (their kernels walk directed rows)
void select_template(ti_handle* h, long long B, bool allow_pair = true)
{
    const int pick = template_for(h, B, allow_pair);
    const ti_handle::Tpl& T = h->tpl[pick];
    h->active = pick; h->G = T.G; h->parts = T.P; h->nblk = T.nblk; h->rows.p = T.rows.p; h->slotnode.p = T.slotnode.p;
    h->max_slots = T.max_slots;
}

// row of (molecule m, sorted edge k) in the e / te layout of the active template
size_t edge_row_of(const ti_handle* h, size_t m, size_t k)
{
    const ti_handle::Tpl& T = h->tpl[h->active];
    if (h->active == 2) {                   // pair-major rows: [group][block][direction][16]
        const size_t A = h->d.n_atoms;
        return (m / T.G) * (size_t)T.nblk * 2 * ti::EDGE_ROWS_PER_BLOCK + (size_t)h->pair_pos[((m % T.G) * A + h->esrc[h->perm[k]]) * A + h->edst[h->perm[k]]];
    }
    // throughput template: one part, pos over (molecule in group, sorted edge); latency template: G = 1, pos over the sorted edge
    const size_t part = T.part_of[k], r = T.pos[(m % T.G) * (size_t)h->d.n_edges + k];
    return ((m / T.G) * T.P + part) * T.nblk * ti::EDGE_ROWS_PER_BLOCK + r;
}

// edge rows (e, te) per molecule-group slot, the larger of the two templates: rows a batch of B molecules needs
size_t edge_rows_for(const ti_handle* h, long long B, long long copies = 1)
{
    size_t best = 1;
    for (int t = 0; t < h->n_tpl; ++t) {
        const ti_handle::Tpl& T = h->tpl[t];
        best = std::max<size_t>(best, (size_t)((B + T.G - 1) / T.G) * (size_t)copies * T.P * T.nblk * ti::EDGE_ROWS_PER_BLOCK);
    }
    if (h->has_pair && copies == 1)         // two directions per pair row
        best = std::max<size_t>(best, (size_t)((B + h->tpl[2].G - 1) / h->tpl[2].G) * h->tpl[2].nblk * 2 * ti::EDGE_ROWS_PER_BLOCK);
    return best;
}

void pack_painn(ti_handle* h, const float* wts)
{
    const int F = h->d.n_features, L = h->d.n_layers, NB = h->NB, nE = h->nE;
    std::vector<float> pk;
    auto begin_stream = [&]() { return pk.size() / 4; };
    size_t o = 0;
    const int NBK = F / 16;
    const int prec = h->d.precision;          // 16-row chunks: f32 image, (hi, lo) fp16 image of the same size, or hi-only fp16 image of half the size
    const size_t ch4 = (prec == TI_PREC_F16 ? 128 : 256) * (size_t)NB;      // float4 per 16-row chunk
    auto chunk16 = [&](size_t W, int ld, int n_rows, int row0, int col0) {
        if (prec == TI_PREC_F16) pack_chunk16_half(pk, wts + W, ld, n_rows, row0, col0, NBK);
        else if (prec == TI_PREC_F16X2) pack_chunk16_split(pk, wts + W, ld, n_rows, row0, col0, NBK);
        else pack_chunk16(pk, wts + W, ld, n_rows, row0, col0, NBK);
    };
    auto layer16 = [&](size_t W, int ld, int n_rows, int col0) { for (int nbo = 0; nbo < NB; ++nbo) chunk16(W, ld, n_rows, 32 * nbo, col0); };
    auto end_stream16 = [&](size_t off4) { return Stream{off4, (int)((pk.size() / 4 - off4) / ch4)}; };
    auto pad_even = [&](size_t off4) { if (((pk.size() / 4 - off4) / ch4) % 2) pk.resize(pk.size() + 4 * ch4, 0.f); };
    o = begin_stream();                              // embed kernel, 16-row chunk format: L1 by input segment, L2, L3, then P for the first message block
    for (int seg = 0; seg < nE; ++seg) layer16(h->embed.W0, nE * F, F, seg * F);
    layer16(h->embed.W1, F, F, 0); layer16(h->embed.W2, F, F, 0);
    if (L > 0) layer16(h->phi[0].W0, 2 * F, F, 0);
    else pk.resize(pk.size() + 4 * ch4 * NB, 0.f);
    pad_even(o);
    h->st_embed16 = end_stream16(o);
    for (int l = 0; l < L; ++l) {
        const bool first = l == 0, last = l == L - 1;
        o = begin_stream();                          // edge kernel: 16-row chunk format
        layer16(h->w[l].W0, F, F, 0); layer16(h->w[l].W1, F, F, 0);
        layer16(h->phi[l].W0, 2 * F, F, F);        // the e half of [s[src] | e]
        layer16(h->phi[l].W1, F, F, 0);
        for (int nbo = 0; nbo < NB; ++nbo)
            for (int c : {2, 3, 1, 0, 4}) {       // consumption order of painn_edge_kernel: ds, de, sed, gates, cross gates
                if (c == 3 && last) continue;
                if ((c == 0 || c == 4) && first) continue;
                chunk16(h->phi[l].W2, F, 5 * F, c * F + 32 * nbo, 0);
                chunk16(h->w[l].W2, F, 5 * F, c * F + 32 * nbo, 0);
            }
        h->st_edge.push_back(end_stream16(o));
        if (edge_uses_one_chain(NB, prec)) {         // the same chunks in the one-accumulator format, each matrix scaled by its own power of two
            const float S[6] = {matrix_pow2_scale(wts + h->w[l].W0, F, F, 0, F, wts + h->w[l].b0, F), matrix_pow2_scale(wts + h->w[l].W1, F, F, 0, F, wts + h->w[l].b1, F),
                                matrix_pow2_scale(wts + h->phi[l].W0, 2 * F, F, F, F, wts + h->phi[l].b0, F), matrix_pow2_scale(wts + h->phi[l].W1, F, F, 0, F, wts + h->phi[l].b1, F),
                                matrix_pow2_scale(wts + h->phi[l].W2, F, 5 * F, 0, F, wts + h->phi[l].b2, 5 * F), matrix_pow2_scale(wts + h->w[l].W2, F, 5 * F, 0, F, wts + h->w[l].b2, 5 * F)};
            auto layer1 = [&](size_t W, int ld, int n_rows, int col0, float sc) { for (int nbo = 0; nbo < NB; ++nbo) pack_chunk16_split1(pk, wts + W, ld, n_rows, 32 * nbo, col0, NBK, sc); };
            o = begin_stream();
            layer1(h->w[l].W0, F, F, 0, S[0]); layer1(h->w[l].W1, F, F, 0, S[1]);
            layer1(h->phi[l].W0, 2 * F, F, F, S[2]);
            layer1(h->phi[l].W1, F, F, 0, S[3]);
            for (int nbo = 0; nbo < NB; ++nbo)
                for (int c : {2, 3, 1, 0, 4}) {
                    if (c == 3 && last) continue;
                    if ((c == 0 || c == 4) && first) continue;
                    pack_chunk16_split1(pk, wts + h->phi[l].W2, F, 5 * F, c * F + 32 * nbo, 0, NBK, S[4]);
                    pack_chunk16_split1(pk, wts + h->w[l].W2, F, 5 * F, c * F + 32 * nbo, 0, NBK, S[5]);
                }
            h->st_edge1.push_back(end_stream16(o));
            h->edge_scale.insert(h->edge_scale.end(), S, S + 6);
        }
        o = begin_stream();                          // update kernel: 16-row chunk format, order of painn_update_kernel
        layer16(h->V[l], F, F, 0);                                                    // phase A (3 components per visit)
        layer16(h->upd[l].W0, 2 * F, F, 0); layer16(h->upd[l].W0, 2 * F, F, F);      // MLP L1: |vv| part, s part
        layer16(h->upd[l].W1, F, F, 0);
        for (int nbo = 0; nbo < NB; ++nbo) {
            chunk16(h->upd[l].W2, F, 3 * F, F + 32 * nbo, 0);                         // scale_squared_norm
            chunk16(h->upd[l].W2, F, 3 * F, 2 * F + 32 * nbo, 0);                     // add_invariant_features
        }
        layer16(h->upd[l].W2, F, 3 * F, 0);                                           // gates
        for (int c = 0; c < 3; ++c) layer16(h->U[l], F, F, 0);                        // phase C (one spatial component per walk)
        if (!last) layer16(h->phi[l + 1].W0, 2 * F, F, 0);                            // phase D
        pad_even(o);                                                                  // whole superchunks
        h->st_update.push_back(end_stream16(o));
        // tangent edge kernel (painn_jvp_kernels.hip): the phi branch's chunks in the consumption order of the edge kernels
        for (int which = 1; which < 2; ++which) {      // the phi branch alone (the primal pass reads the primal edge stream)
            const MlpOff& m = h->phi[l];
            o = begin_stream();
            layer16(m.W0, 2 * F, F, F);
            layer16(m.W1, F, F, 0);
            for (int nbo = 0; nbo < NB; ++nbo)
                for (int c : {2, 3, 1, 0, 4}) {
                    if (c == 3 && last) continue;
                    if ((c == 0 || c == 4) && first) continue;
                    chunk16(m.W2, F, 5 * F, c * F + 32 * nbo, 0);
                }
            const int real = end_stream16(o).nch;   // an odd count gets one pad chunk, which the kernels swallow once per row block
            pad_even(o);
            h->st_jvp_phi.push_back(end_stream16(o));
            h->jvp_phi_pad.push_back(real & 1);
        }
        o = begin_stream();                          // tangent update kernel: same order, V and U once per spatial component
        for (int c = 0; c < 3; ++c) layer16(h->V[l], F, F, 0);
        layer16(h->upd[l].W0, 2 * F, F, 0); layer16(h->upd[l].W0, 2 * F, F, F);
        layer16(h->upd[l].W1, F, F, 0);
        for (int nbo = 0; nbo < NB; ++nbo) {
            chunk16(h->upd[l].W2, F, 3 * F, F + 32 * nbo, 0);
            chunk16(h->upd[l].W2, F, 3 * F, 2 * F + 32 * nbo, 0);
        }
        layer16(h->upd[l].W2, F, 3 * F, 0);
        for (int c = 0; c < 3; ++c) layer16(h->U[l], F, F, 0);
        if (!last) layer16(h->phi[l + 1].W0, 2 * F, F, 0);
        pad_even(o);
        h->st_jvp_update.push_back(end_stream16(o));
    }
    // readout kernels (primal and tangent): 16-row chunk format
    o = begin_stream();
    for (size_t Wm : {h->readout.W0, h->readout.W1})
        for (int nbo = 0; nbo < NB; ++nbo) {
            chunk16(Wm, F, F, 32 * nbo, 0);
        }
    pad_even(o);
    h->st_jvp_readout = end_stream16(o);
    h->packed.upload(pk);
    {
        std::vector<float> rv;                       // order = struct RV in painn_jvp_kernels.hip
        const MlpOff& r = h->readout;
        // Vr follows the 2-float readout bias in the canonical layout (h->Vr itself points at an aligned copy inside `flat`)
        for (size_t off : {r.b0, r.g0, r.be0, r.b1, r.g1, r.be1, r.W2 + (size_t)F, r.b2 + 2}) rv.insert(rv.end(), wts + off, wts + off + F);
        h->jvp_ro_vecs.upload(rv);
    }
    // per-layer vector block of the edge kernel (order = struct EV in painn_kernels.hip)
    std::vector<float> ev;
    for (int l = 0; l < L; ++l) {
        const MlpOff &w = h->w[l], &ph = h->phi[l];
        for (size_t off : {w.b0, w.g0, w.be0, w.b1, w.g1, w.be1, ph.g0, ph.be0, ph.b1, ph.g1, ph.be1}) ev.insert(ev.end(), wts + off, wts + off + F);
        ev.insert(ev.end(), wts + ph.b2, wts + ph.b2 + 5 * F);
        ev.insert(ev.end(), wts + w.b2, wts + w.b2 + 5 * F);
    }
    h->edge_vecs.upload(ev);
    if (edge_uses_one_chain(NB, prec)) {             // the message kernel's copy: bias rows times the scale of their matrix (EV order: W_B0 = 0, W_B1 = 3, P_B1 = 8, P_B2 = 11..15, W_B2 = 16..20)
        std::vector<float> ev1 = ev;
        for (int l = 0; l < L; ++l) {
            const float* S = h->edge_scale.data() + (size_t)l * 6;
            float* b = ev1.data() + (size_t)l * 21 * F;
            auto mul = [&](int row, int n, float sc) { for (int i = 0; i < n * F; ++i) b[(size_t)row * F + i] *= sc; };
            mul(0, 1, S[0]); mul(3, 1, S[1]); mul(8, 1, S[3]); mul(11, 5, S[4]); mul(16, 5, S[5]);
        }
        h->edge_vecs1.upload(ev1);
    }
    std::vector<float> uv;                           // order = struct UV in painn_kernels.hip
    for (int l = 0; l < L; ++l) {
        const MlpOff& u = h->upd[l];
        for (size_t off : {u.b0, u.g0, u.be0, u.b1, u.g1, u.be1}) uv.insert(uv.end(), wts + off, wts + off + F);
        uv.insert(uv.end(), wts + u.b2, wts + u.b2 + 3 * F);
        if (l + 1 < L) uv.insert(uv.end(), wts + h->phi[l + 1].b0, wts + h->phi[l + 1].b0 + F);
        else uv.insert(uv.end(), F, 0.f);
    }
    h->upd_vecs.upload(uv);
}

void ensure_painn_ws(ti_handle* h, long long B)
{
    if (B <= h->cap) return;
    const size_t A = h->d.n_atoms, F = h->d.n_features, N = (size_t)B * A;
    h->x.alloc(N * 3); h->b1.alloc(N * 3); h->b2.alloc(N * 3); h->xt.alloc(N * 3);
    h->cond.alloc(std::max<size_t>(N * h->ncond, 1));
    const size_t se = h->d.precision == TI_PREC_F16 ? 2 : 1;      // state tensors s, P, v, e: fp16 in the storage mode (2 per float slot)
    h->s.alloc((N * F + se - 1) / se); h->P.alloc((N * F + se - 1) / se);
    h->v.alloc((N * 3 * F + se - 1) / se); h->dvacc.alloc(N * 3 * F); h->cacc.alloc(N * 3 * F); h->dsacc.alloc(N * F);
    h->e.alloc((edge_rows_for(h, B) * F + se - 1) / se);
    // parked geometry of a drift evaluation (painn_edge_kernel.hpp): the encoding operand of every edge row (as many bytes as e) and edge_dir
    h->enc.alloc((edge_rows_for(h, B) * F + se - 1) / se); h->geo.alloc(edge_rows_for(h, B) * 4);
    h->divb.alloc(B); h->div2.alloc(B); h->dl.alloc(B); h->dlscaled.alloc(B);
    if (h->has_pair && pair_uses_partials()) h->part.alloc((size_t)((B + h->tpl[2].G - 1) / h->tpl[2].G) * h->tpl[2].nblk * 8 * 7 * F);
    h->cap = B;
}

// ---- forward-mode derivative: tangent workspace over ceil(B/G)*D*G virtual molecules (painn_jvp_kernels.hip header)
long long jvp_virtual_molecules(const ti_handle* h, long long B, int D) { return (B + h->G - 1) / h->G * D * h->G; }

size_t jvp_bytes_per_vm(const ti_handle* h)
{
    const size_t A = h->d.n_atoms, F = h->d.n_features;
    const size_t erows = ((size_t)h->parts * h->nblk * ti::EDGE_ROWS_PER_BLOCK + h->G - 1) / h->G;
    return (A * F * 12 + erows * F + A * 3) * sizeof(float);
}

void ensure_jvp_ws(ti_handle* h, long long B, int D)
{
    const long long VB = jvp_virtual_molecules(h, B, D);
    const size_t A = h->d.n_atoms, F = h->d.n_features, N = (size_t)VB * A;
    const size_t pgroups = ((size_t)B + h->G - 1) / h->G * h->parts;
    const size_t wq_floats = std::max<size_t>(pgroups * h->nblk * 5 * h->NB * 6 * 64 * 4, 4);
    const size_t st_floats = std::max<size_t>(pgroups * h->nblk * 4 * (2 * h->NB) * 64 * 4, 4);
    if (h->wq.n < wq_floats) h->wq.alloc(wq_floats);
    if (h->phist.n < st_floats) h->phist.alloc(st_floats);
    const size_t ns_floats = (((size_t)B * A + 15) / 16) * 13 * (2 * h->NB) * 64 * 4;
    if (h->nodest.n < ns_floats) h->nodest.alloc(ns_floats);
    const size_t te_floats = (size_t)VB / h->G * h->parts * h->nblk * ti::EDGE_ROWS_PER_BLOCK * F;
    if (h->te.n < te_floats) h->te.alloc(std::max<size_t>(te_floats, 1));
    if (VB <= h->jvp_cap) return;
    if (N >= ((size_t)1 << 31)) throw std::invalid_argument("too many tangent nodes in one pass (lower TI_JVP_WS_GB)");
    h->ts.alloc(N * F); h->tP.alloc(N * F); h->tdsacc.alloc(N * F);
    h->tv.alloc(N * 3 * F); h->tdvacc.alloc(N * 3 * F); h->tcacc.alloc(N * 3 * F);
    h->tout.alloc(N * 3);
    h->jvp_cap = VB;
}

// molecules per tangent pass with D directions each, from the HBM budget TI_JVP_WS_GB (default 48 GB of tangent state)
long long jvp_chunk_molecules(const ti_handle* h, int D)
{
    double gb = 48.0;
    if (const char* e = std::getenv("TI_JVP_WS_GB")) gb = std::max(0.001, std::atof(e));
    const double per_mol = (double)jvp_bytes_per_vm(h) * D;
    const long long by_mem = (long long)(gb * 1e9 / per_mol);
    const long long by_index = (long long)(((size_t)1 << 31) - 1) / ((long long)D * h->d.n_atoms) - h->G;
    const long long c = std::max<long long>(1, std::min(by_mem, by_index));
    return c >= h->G ? c / h->G * h->G : c;                  // whole primal groups per pass
}

struct JvpRun {            // one tangent pass riding on a drift evaluation
    int D;                 // seed directions per molecule
    const float* xdot;     // D == 1: explicit direction [B*A*3] (device); NULL: unit seeds, D = 3A
    float* tout;           // [B*D*A*3] tangent of the drift (device)
};

// one drift evaluation, everything on h->stream; x_dev / out_dev are device pointers [B*A*3].  With `jr` the tangent
// kernels run in lock step: each reads the primal state its layer's primal kernel is about to overwrite.
void painn_drift_dev(ti_handle* h, const float* x_dev, float t, const float* cond_dev, long long B, float* out_dev,
                     const JvpRun* jr = nullptr)
{
    const int A = h->d.n_atoms, F = h->d.n_features, L = h->d.n_layers, NB = h->NB;
    const long long N = B * A, groups = (B + h->G - 1) / h->G * h->parts;         // edge-kernel waves: (molecule group, part)
    hipStream_t st = h->stream;
    const bool split = h->d.precision == TI_PREC_F16X2;
    const int prec = h->d.precision;
    if (jr && prec == TI_PREC_F16) throw std::invalid_argument("the fp16 storage mode has no divergence / tangent path (use f32 or f16x2)");
    if (jr && h->active == 2) throw std::logic_error("tangent passes walk directed edge rows (select_template(.., allow_pair = false))");
    const long long VB = jr ? jvp_virtual_molecules(h, B, jr->D) : 0, VN = VB * A, vgroups = VB / h->G * h->parts;
    if (jr) {
        ensure_jvp_ws(h, B, jr->D);
        h->last_VB = B * jr->D; h->last_D = jr->D;
        const size_t nb = (size_t)VN * F * sizeof(float);
        HIP_CHECK(hipMemsetAsync(h->ts.p, 0, nb, st)); HIP_CHECK(hipMemsetAsync(h->tP.p, 0, nb, st));
        HIP_CHECK(hipMemsetAsync(h->tdsacc.p, 0, nb, st));
        HIP_CHECK(hipMemsetAsync(h->tv.p, 0, 3 * nb, st)); HIP_CHECK(hipMemsetAsync(h->tdvacc.p, 0, 3 * nb, st));
        HIP_CHECK(hipMemsetAsync(h->tcacc.p, 0, 3 * nb, st));
    }
    const size_t vbytes = (size_t)N * 3 * F * sizeof(float);
    // With first-touch accumulators (every atom has incoming edges) nothing needs zeroing: layer 0's edge kernel replaces dsacc and
    // dvacc, its update kernel does not read v and cacc (zero by definition), every later layer replaces all three.  The forward-mode
    // passes read the primal v and cacc of layer 0 themselves, and with no layers the readout reads v: then they are cleared.
    const bool ft = h->first_touch && h->tap < 0;      // (the debug taps read "state + pending accumulators": they want them zeroed after use)
    if (!ft || jr || L == 0) {
        HIP_CHECK(hipMemsetAsync(h->v.p, 0, prec == TI_PREC_F16 ? vbytes / 2 : vbytes, st));
        HIP_CHECK(hipMemsetAsync(h->cacc.p, 0, vbytes, st));
    }
    if (!ft) {
        HIP_CHECK(hipMemsetAsync(h->dvacc.p, 0, vbytes, st));
        HIP_CHECK(hipMemsetAsync(h->dsacc.p, 0, (size_t)N * F * sizeof(float), st));
    }
    {
        EmbedParams p{};
        p.stream = h->S(h->st_embed16); p.nch = h->st_embed16.nch; p.mlp = h->vec(h->embed);
        p.pb0 = L > 0 ? h->F(h->phi[0].b0) : h->F(h->embed.b2);
        p.atom_emb = h->F(h->atom_emb); p.atom_ids = h->atom_ids.p; p.cond = cond_dev; p.ncond = h->ncond; p.A = A; p.N = N;
        p.t = t; p.temp_length = h->d.temp_length; p.time_length = h->d.time_length; p.temp_mean = h->d.temp_mean; p.temp_range = h->d.temp_range;
        p.s = h->s.p; p.P = h->P.p;
        Timed tm(h, TI_KERNEL_PAINN_EMBED);
        HIP_CHECK(launch_embed(NB, h->nE, prec, p, st));
    }
    h->last_B = B;
    if (h->tap == 0) return;
    for (int l = 0; l < L; ++l) {
        if (jr && h->nblk > 0) {
            {
                JvpFilterParams p{};
                p.stream = h->S(h->st_edge[l]); p.nch = h->st_edge[l].nch; p.vecs = h->edge_vecs.p + (size_t)l * 21 * F;
                p.edge_emb = h->F(h->edge_emb); p.rows = h->rows.p; p.nblk = h->nblk; p.G = h->G; p.parts = h->parts; p.A = A; p.first = l == 0; p.last = l == L - 1;
                p.B = B; p.n_groups = groups; p.length_scale = h->d.length_scale; p.x = x_dev; p.P = h->P.p; p.e = h->e.p;
                p.wq = reinterpret_cast<float4*>(h->wq.p); p.st = reinterpret_cast<float4*>(h->phist.p);
                Timed tm(h, TI_KERNEL_PAINN_JVP_FILTER);
                HIP_CHECK(launch_jvp_filter(NB, split, p, st));
            }
            JvpEdgeParams p{};
            p.stream = h->S(h->st_jvp_phi[l]); p.nch = h->st_jvp_phi[l].nch; p.pad = h->jvp_phi_pad[l]; p.vecs = h->edge_vecs.p + (size_t)l * 21 * F;
            p.edge_emb = h->F(h->edge_emb); p.rows = h->rows.p; p.slotnode = h->slotnode.p;
            p.nblk = h->nblk; p.G = h->G; p.parts = h->parts; p.A = A; p.D = jr->D; p.first = l == 0; p.last = l == L - 1;
            p.B = B; p.n_groups = vgroups;
            p.x = x_dev; p.xdot = jr->xdot; p.P = h->P.p; p.v = h->v.p; p.e = h->e.p; p.wq = reinterpret_cast<const float4*>(h->wq.p);
            p.st = reinterpret_cast<const float4*>(h->phist.p); p.tP = h->tP.p; p.tv = h->tv.p;
            p.te = h->te.p; p.tdsacc = h->tdsacc.p; p.tdvacc = h->tdvacc.p; p.tcacc = h->tcacc.p;
            Timed tm(h, TI_KERNEL_PAINN_JVP_EDGE);
            HIP_CHECK(launch_jvp_edge(NB, split, p, st));
        }
        if (h->nblk > 0) {
            EdgeParams p{};
            p.stream = h->S(h->st_edge[l]); p.nch = h->st_edge[l].nch; p.vecs = h->edge_vecs.p + (size_t)l * 21 * F;
            p.edge_emb = h->F(h->edge_emb); p.rows = h->rows.p; p.slotnode = h->slotnode.p; p.nslots = nullptr;
            p.nblk = h->nblk; p.G = h->G; p.parts = h->parts; p.A = A; p.max_slots = h->max_slots; p.B = B; p.n_groups = groups; p.length_scale = h->d.length_scale;
            p.x = x_dev; p.P = h->P.p; p.v = h->v.p; p.dsacc = h->dsacc.p; p.dvacc = h->dvacc.p; p.cacc = h->cacc.p; p.e = h->e.p; p.enc = h->enc.p; p.geo = h->geo.p;
            for (int i = 0; i < 6; ++i) p.wscale[i] = 1.0f;
            if (edge_uses_one_chain(NB, prec)) {     // the message kernel's own stream format (the primal pass of the divergence keeps the other one)
                p.stream = h->S(h->st_edge1[l]); p.nch = h->st_edge1[l].nch; p.vecs = h->edge_vecs1.p + (size_t)l * 21 * F;
                for (int i = 0; i < 6; ++i) p.wscale[i] = h->edge_scale[(size_t)l * 6 + i];
            }
            Timed tm(h, TI_KERNEL_PAINN_EDGE);
#ifdef TI_STAMPS      // diagnostic build only: stamps of layer 2's launch, printed to stderr
            static DevBuf<unsigned long long> stamp_buf;
            const size_t n_st = 2048 + 2 * (size_t)groups + 16;
            if (l == 2 && std::getenv("TI_STAMPS_DUMP")) { if (stamp_buf.n < n_st) stamp_buf.alloc(n_st); HIP_CHECK(hipMemsetAsync(stamp_buf.p, 0, n_st * 8, st)); p.stamps = stamp_buf.p; }
            auto dump_stamps = [&]() {
                if (!p.stamps) return;
                std::vector<unsigned long long> hs(n_st);
                HIP_CHECK(hipStreamSynchronize(st));
                HIP_CHECK(hipMemcpy(hs.data(), stamp_buf.p, n_st * 8, hipMemcpyDeviceToHost));
                for (int w = 0; w < 32; ++w) {
                    if (!hs[(size_t)w * 64 + 1]) continue;
                    std::fprintf(stderr, "STAMP %d:", w);
                    for (int k = 1; k < 64 && hs[(size_t)w * 64 + k]; ++k) std::fprintf(stderr, " %llu", hs[(size_t)w * 64 + k] - hs[(size_t)w * 64 + k - 1]);
                    std::fprintf(stderr, "\n");
                }
                std::vector<double> clk;
                for (long long g2 = 0; g2 < groups; ++g2) { const auto c = hs[2048 + 2 * g2], r = hs[2048 + 2 * g2 + 1]; if (r) clk.push_back(100e6 * (double)c / (double)r); }
                std::sort(clk.begin(), clk.end());
                if (!clk.empty()) std::fprintf(stderr, "INKERNEL_CLOCK_GHZ median %.4f  p10 %.4f  p90 %.4f  waves %zu  layout %d  precision %d\n", clk[clk.size() / 2] / 1e9,
                                               clk[clk.size() / 10] / 1e9, clk[clk.size() * 9 / 10] / 1e9, clk.size(), h->active, prec);
            };
#endif
            if (h->active == 2) {
                p.part = h->part.p;
                HIP_CHECK(launch_pair(NB, l == 0, l == L - 1, prec, p, st));
#ifdef TI_STAMPS
                dump_stamps();
#endif
                if (pair_uses_partials()) {
                PairReduceParams r{};
                r.part = h->part.p; r.plist = h->pair_plist.p; r.kmax = h->pair_kmax; r.G = h->G; r.A = A; r.F = F; r.nblk = h->nblk;
                r.has_c = l > 0; r.B = B; r.dsacc = h->dsacc.p; r.dvacc = h->dvacc.p; r.cacc = h->cacc.p;
                HIP_CHECK(launch_pair_reduce(r, st));
                }
            } else {
                HIP_CHECK(launch_edge(NB, l == 0, l == L - 1, prec, p, st));
#ifdef TI_STAMPS
                dump_stamps();
#endif
            }
        }
        if (h->tap == 1 + 2 * l) return;
        if (jr) {
            {
                JvpNodeParams p{};
                p.stream = h->S(h->st_jvp_update[l]); p.nch = h->st_jvp_update[l].nch; p.vecs = h->upd_vecs.p + (size_t)l * 10 * F;
                p.N = N; p.s = h->s.p; p.v = h->v.p; p.dsacc = h->dsacc.p; p.dvacc = h->dvacc.p; p.cacc = h->cacc.p;
                p.ns = reinterpret_cast<float4*>(h->nodest.p);
                Timed tm(h, TI_KERNEL_PAINN_JVP_FILTER);
                HIP_CHECK(launch_jvp_node(NB, split, p, st));
            }
            JvpUpdateParams p{};
            p.stream = h->S(h->st_jvp_update[l]); p.nch = h->st_jvp_update[l].nch; p.vecs = h->upd_vecs.p + (size_t)l * 10 * F;
            p.N = VN; p.B = B; p.A = A; p.D = jr->D; p.G = h->G; p.has_next = l + 1 < L;
            p.v = h->v.p; p.cacc = h->cacc.p; p.ns = reinterpret_cast<const float4*>(h->nodest.p);
            p.ts = h->ts.p; p.tv = h->tv.p; p.tdsacc = h->tdsacc.p; p.tdvacc = h->tdvacc.p; p.tcacc = h->tcacc.p; p.tP = h->tP.p;
            p.zero_acc = !ft;
            Timed tm(h, TI_KERNEL_PAINN_JVP_UPDATE);
            HIP_CHECK(launch_jvp_update(NB, split, p, st));
        }
        {
            UpdateParams p{};
            p.stream = h->S(h->st_update[l]); p.nch = h->st_update[l].nch; p.vecs = h->upd_vecs.p + (size_t)l * 10 * F;
            p.N = N; p.s = h->s.p; p.v = h->v.p; p.dsacc = h->dsacc.p; p.dvacc = h->dvacc.p; p.cacc = h->cacc.p; p.P = h->P.p;
            p.first_layer = l == 0; p.zero_acc = !ft;
            Timed tm(h, TI_KERNEL_PAINN_UPDATE);
            HIP_CHECK(launch_update(NB, l + 1 < L, prec, p, st));
        }
        if (h->tap == 2 + 2 * l) return;
    }
    if (jr) {
        JvpReadoutParams p{};
        p.stream = h->S(h->st_jvp_readout); p.nch = h->st_jvp_readout.nch; p.vecs = h->jvp_ro_vecs.p; p.b2_gate = h->b2_gate;
        p.N = VN; p.B = B; p.A = A; p.D = jr->D; p.G = h->G; p.s = h->s.p; p.v = h->v.p; p.ts = h->ts.p; p.tv = h->tv.p; p.tout = jr->tout;
        Timed tm(h, TI_KERNEL_PAINN_JVP_READOUT);
        HIP_CHECK(launch_jvp_readout(NB, split, p, st));
    }
    {
        ReadoutParams p{};
        p.stream = h->S(h->st_jvp_readout); p.nch = h->st_jvp_readout.nch; p.mlp = h->vec(h->readout);      // the 16-row image of W0, W1
        p.w2_gate = h->F(h->readout.W2 + F); p.b2_gate = h->b2_gate; p.Vr = h->F(h->Vr);
        p.N = N; p.s = h->s.p; p.v = h->v.p; p.out = out_dev;
        Timed tm(h, TI_KERNEL_PAINN_READOUT);
        HIP_CHECK(launch_readout(NB, prec, p, st));
    }
}

// drift and exact divergence: 3A unit-seed tangent passes per molecule, in chunks that fit the tangent HBM budget
void painn_drift_div_dev(ti_handle* h, const float* x_dev, float t, const float* cond_dev, long long B, float* out_dev, float* div_dev)
{
    const int A = h->d.n_atoms, D = 3 * A;
    const long long chunk = jvp_chunk_molecules(h, D);
    for (long long b0 = 0; b0 < B; b0 += chunk) {
        const long long bc = std::min(chunk, B - b0);
        ensure_jvp_ws(h, bc, D);
        JvpRun jr{D, nullptr, h->tout.p};
        painn_drift_dev(h, x_dev + (size_t)b0 * A * 3, t, cond_dev ? cond_dev + (size_t)b0 * A * h->ncond : nullptr, bc,
                        out_dev + (size_t)b0 * A * 3, &jr);
        HIP_CHECK(launch_div_reduce(h->tout.p, bc, D, h->G, div_dev + b0, h->stream));
    }
    h->last_B = std::min(chunk, B);
}

// ------------------------------------------------------------------------------------------------ adw helpers
void adw_mlp_launch(ti_handle* h, bool embed, const float* a0, const float* in1, const float* emb, const int32_t* idx, float t,
                    long long rows, float* out, float* out_div)
{
    AdwParams p{};
    const Stream& s2 = embed ? h->st_be : h->st_net;
    p.stream = h->S(s2); p.nch = s2.nch;
    p.vecs = h->F(embed ? h->a_be_vecs : h->a_net_vecs);
    p.b_out = embed ? h->a_be_b_out : h->a_b_out; p.n_hidden = embed ? 1 : h->ad.num_layers - 1; p.B = rows;
    p.x = a0; p.in1 = in1; p.emb = emb; p.idx = idx; p.t = t; p.out = out; p.out_div = out_div;
    Timed tm(h, TI_KERNEL_ADW);
    HIP_CHECK(launch_adw(h->NB, h->ad.precision == TI_PREC_F16X2, p, h->stream));
}

// upload conditioning: dedupe (beta0, beta1) pairs on the host (the driver uses one pair, adw/sample.py:24)
long long adw_set_cond(ti_handle* h, const float* beta0, const float* beta1, long long B, int mem)
{
    std::vector<float> b0(B), b1(B);
    if (mem == TI_MEM_DEVICE) {
        HIP_CHECK(hipMemcpy(b0.data(), beta0, B * sizeof(float), hipMemcpyDeviceToHost));
        HIP_CHECK(hipMemcpy(b1.data(), beta1, B * sizeof(float), hipMemcpyDeviceToHost));
    } else { std::memcpy(b0.data(), beta0, B * sizeof(float)); std::memcpy(b1.data(), beta1, B * sizeof(float)); }
    std::map<std::pair<float, float>, int> uniq;
    std::vector<int32_t> idx(B);
    std::vector<float> u0, u1;
    for (long long i = 0; i < B; ++i) {
        auto key = std::make_pair(b0[i], b1[i]);
        auto it = uniq.find(key);
        if (it == uniq.end()) { it = uniq.emplace(key, (int)u0.size()).first; u0.push_back(b0[i]); u1.push_back(b1[i]); }
        idx[i] = it->second;
    }
    h->aidx.upload(idx); h->abeta0_u.upload(u0); h->abeta1_u.upload(u1);
    h->aemb_u.alloc(u0.size());
    return (long long)u0.size();
}

// out_div (may be NULL): d b / d x, the divergence of the 1-D drift (beta_embed does not depend on x)
void adw_drift_dev(ti_handle* h, const float* x_dev, float t, long long U, long long B, float* out_dev, float* out_div)
{
    adw_mlp_launch(h, true, h->abeta0_u.p, h->abeta1_u.p, nullptr, nullptr, t, U, h->aemb_u.p, nullptr);   // beta_embed([b0, b1, t])
    adw_mlp_launch(h, false, x_dev, nullptr, h->aemb_u.p, h->aidx.p, t, B, out_dev, out_div);             // net([x, t, embed])
}

void ensure_adw_ws(ti_handle* h, long long B)
{
    if (B <= h->cap) return;
    h->ax.alloc(B); h->ab1.alloc(B); h->ab2.alloc(B); h->axt.alloc(B); h->adl.alloc(B); h->ad1.alloc(B); h->ad2.alloc(B);
    h->cap = B;
}

// ------------------------------------------------------------------------------------------------ shared rollout
// drift(x_dev, t, out_dev) evaluates the drift; state arrays have n floats; comps = floats per trajectory
// Optional second state of the reference ODE: d(dlogp)/dt = -div * 1e-2, returned * 1e2 (adw/thermo/integrators.py:38-68).
struct DlogpAux {
    float *dl = nullptr, *d1 = nullptr, *d2 = nullptr, *scaled = nullptr, *out = nullptr;
    size_t n_dl = 0;                              // entries of the second state (0: same as the first state's n)
    float div_scale = 1e-2f, out_scale = 100.0f;  // d(dlogp)/dt = -div_scale * div, written * out_scale
};

// drift(x_dev, t, out_b, out_div) evaluates the drift (and the divergence if out_div != NULL); state arrays have n floats
template <typename Drift>
int rollout_common(ti_handle* h, const ti_rollout_desc* rd, float* x, float* b1, float* b2, float* xt, size_t n, long long B, int comps,
                   int atoms_for_com, float* out_path, int64_t* n_fevals, Drift&& drift, DlogpAux aux = DlogpAux())
{
    hipStream_t st = h->stream;
    const hipMemcpyKind out_kind = rd->mem == TI_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    int64_t row = 0, fe = 0;
    const size_t ndl = aux.n_dl ? aux.n_dl : n;
    if (aux.dl) HIP_CHECK(hipMemsetAsync(aux.dl, 0, ndl * sizeof(float), st));
    auto save = [&]() {
        if (aux.dl) {
            HIP_CHECK(launch_scale(aux.scaled, aux.dl, aux.out_scale, (long long)ndl, st));
            HIP_CHECK(hipMemcpyAsync(aux.out + (size_t)row * ndl, aux.scaled, ndl * sizeof(float), out_kind, st));
        }
        HIP_CHECK(hipMemcpyAsync(out_path + (size_t)(row++) * n, x, n * sizeof(float), out_kind, st));
    };
    if (rd->save_every > 0) save();
    for (int k = 0; k < rd->n_step - 1; ++k) {
        const float dt = rd->t_grid[k + 1] - rd->t_grid[k];
        drift(x, rd->t_grid[k], b1, aux.d1); ++fe;
        if (rd->scheme == TI_SCHEME_HEUN) {
            { Timed tm(h, TI_KERNEL_INTEGRATE); HIP_CHECK(launch_axpy(xt, x, dt, b1, (long long)n, st)); }
            drift(xt, rd->t_grid[k + 1], b2, aux.d2); ++fe;
            { Timed tm(h, TI_KERNEL_INTEGRATE); HIP_CHECK(launch_heun(x, 0.5f * dt, b1, b2, (long long)n, st)); }
            if (aux.dl) HIP_CHECK(launch_heun(aux.dl, -0.5f * dt * aux.div_scale, aux.d1, aux.d2, (long long)ndl, st));
        } else {
            Timed tm(h, TI_KERNEL_INTEGRATE);
            HIP_CHECK(launch_axpy(x, x, dt, b1, (long long)n, st));
            if (aux.dl) HIP_CHECK(launch_axpy(aux.dl, aux.dl, -dt * aux.div_scale, aux.d1, (long long)ndl, st));
            if (rd->scheme == TI_SCHEME_EM && rd->eps > 0.0f)
                HIP_CHECK(launch_noise(x, std::sqrt(2.0f * rd->eps * std::fabs(dt)), rd->seed, rd->traj_offset, (int)(rd->step_offset + k), B, comps,
                                       rd->com_free_noise ? atoms_for_com : 0, st));
        }
        const int step = k + 1;
        if (rd->save_every > 0 && (step % rd->save_every == 0 || step == rd->n_step - 1)) save();
    }
    if (rd->save_every <= 0) save();
    HIP_CHECK(hipMemsetAsync(h->nanflag.p, 0, sizeof(int), st));
    HIP_CHECK(launch_nan_check(x, (long long)n, h->nanflag.p, st));
    int flag = 0;
    HIP_CHECK(hipMemcpyAsync(&flag, h->nanflag.p, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    if (n_fevals) *n_fevals = fe;
    return flag ? fail(TI_E_NAN, "non-finite value in the final state") : TI_OK;
}

// ---- Runge-Kutta drivers on top of the same drift callback: torchdiffeq 0.2.5's `dopri5` (adaptive), `midpoint`, `rk4`
// (include/ti_hip.h TI_SCHEME_*).  State = segment 0 (x, n floats) and optionally segment 1 (dlogp, aux.n_dl floats) with
// right-hand side (b, -div_scale * div); a decreasing grid is integrated in s = -t with f'(s, y) = -f(-s, y) like
// torchdiffeq's _ReverseFunc.  Times and step sizes are fp64 on the host and enter state arithmetic as fp32, as there.
namespace dp5 {
constexpr double alpha[6] = {1. / 5, 3. / 10, 4. / 5, 8. / 9, 1., 1.};
constexpr double beta[6][6] = {{1. / 5},
                               {3. / 40, 9. / 40},
                               {44. / 45, -56. / 15, 32. / 9},
                               {19372. / 6561, -25360. / 2187, 64448. / 6561, -212. / 729},
                               {9017. / 3168, -355. / 33, 46732. / 5247, 49. / 176, -5103. / 18656},
                               {35. / 384, 0., 500. / 1113, 125. / 192, -2187. / 6784, 11. / 84}};
constexpr double c_error[7] = {35. / 384 - 1951. / 21600, 0., 500. / 1113 - 22642. / 50085, 125. / 192 - 451. / 720,
                               -2187. / 6784 - -12231. / 42400, 11. / 84 - 649. / 6300, -1. / 60};
constexpr double c_mid[7] = {6025192743. / 30085553152. / 2, 0., 51252292925. / 65400821598. / 2, -2691868925. / 45128329728. / 2,
                             187940372067. / 1594534317056. / 2, -1776094331. / 19743644256. / 2, 11237099. / 235043384. / 2};
}  // namespace dp5

template <typename Drift>
int rollout_rk(ti_handle* h, const ti_rollout_desc* rd, float* x, size_t n, float* out_path, int64_t* n_fevals, Drift&& drift,
               DlogpAux aux = DlogpAux())
{
    hipStream_t st = h->stream;
    const hipMemcpyKind out_kind = rd->mem == TI_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    const int nseg = aux.dl ? 2 : 1;
    const size_t ndl = aux.dl ? (aux.n_dl ? aux.n_dl : n) : 0;
    const size_t sn[2] = {n, ndl};
    // workspace: per segment k[7], ytmp, ynew, coef[5]
    const size_t per = 14;
    if (h->rk_ws.n < per * (n + ndl)) h->rk_ws.alloc(per * (n + ndl));
    if (h->rk_red.n < (size_t)RED_PARTIALS + 8) h->rk_red.alloc(RED_PARTIALS + 8);
    float* y[2] = {x, aux.dl};
    float *k[2][7] = {}, *ytmp[2] = {}, *ynew[2] = {}, *coef[2] = {};
    struct Pair { float* p[2]; operator float* const*() const { return p; } };
    auto KP = [&](int j) { return Pair{{k[0][j], k[1][j]}}; };
    {
        float* w = h->rk_ws.p;
        for (int s2 = 0; s2 < nseg; ++s2) {
            for (int j = 0; j < 7; ++j) { k[s2][j] = w; w += sn[s2]; }
            ytmp[s2] = w; w += sn[s2]; ynew[s2] = w; w += sn[s2]; coef[s2] = w; w += 5 * sn[s2];
        }
    }
    if (aux.dl) HIP_CHECK(hipMemsetAsync(aux.dl, 0, ndl * sizeof(float), st));
    const int N = rd->n_step;
    const double sign = (N > 1 && rd->t_grid[1] < rd->t_grid[0]) ? -1.0 : 1.0;
    int64_t fe = 0, row = 0;
    // f(s, y) for every segment; `ti` is the fp32 stage time in the (possibly negated) integration variable
    auto F = [&](float ti, float* const* yin, float* const* kout) {
        drift(yin[0], (float)(sign * (double)ti), kout[0], aux.dl ? aux.d1 : nullptr); ++fe;
        if (aux.dl) HIP_CHECK(launch_scale(kout[1], aux.d1, (float)(-sign) * aux.div_scale, (long long)ndl, st));
        if (sign < 0) HIP_CHECK(launch_scale(kout[0], kout[0], -1.0f, (long long)n, st));
    };
    auto save_from = [&](float* const* src) {
        if (aux.dl) {
            HIP_CHECK(launch_scale(aux.scaled, src[1], aux.out_scale, (long long)ndl, st));
            HIP_CHECK(hipMemcpyAsync(aux.out + (size_t)row * ndl, aux.scaled, ndl * sizeof(float), out_kind, st));
        }
        HIP_CHECK(hipMemcpyAsync(out_path + (size_t)(row++) * n, src[0], n * sizeof(float), out_kind, st));
    };
    auto wants_row = [&](int i) { return rd->save_every > 0 ? (i % rd->save_every == 0 || i == N - 1) : i == N - 1; };
    auto comb = [&](int s2, int nk, const double* c, double scale) {
        RkComb r{};
        r.nk = nk;
        for (int j = 0; j < nk; ++j) { r.k[j] = k[s2][j]; r.c[j] = (float)c[j] * (float)scale; }       // beta_ij * dt in fp32
        return r;
    };
    double* red = h->rk_red.p;
    auto fetch = [&]() { double v = 0; HIP_CHECK(hipMemcpyAsync(&v, red + RED_PARTIALS, sizeof(double), hipMemcpyDeviceToHost, st)); HIP_CHECK(hipStreamSynchronize(st)); return v; };
    const float rtol = rd->rtol, atol = rd->atol;
    if (wants_row(0)) save_from(y);

    if (rd->scheme == TI_SCHEME_MIDPOINT || rd->scheme == TI_SCHEME_RK4) {
        // FixedGridODESolver with step_size = None: one step per grid interval (solvers.py; fixed_grid.py Midpoint / RK4)
        for (int i = 0; i + 1 < N; ++i) {
            const float t0 = (float)(sign * rd->t_grid[i]), t1 = (float)(sign * rd->t_grid[i + 1]), dt = t1 - t0;
            F(t0, y, KP(0));
            if (rd->scheme == TI_SCHEME_MIDPOINT) {
                const double half[1] = {0.5};
                for (int s2 = 0; s2 < nseg; ++s2) HIP_CHECK(launch_rk_combo(ytmp[s2], y[s2], comb(s2, 1, half, dt), (long long)sn[s2], st));
                F(t0 + 0.5f * dt, ytmp, KP(1));
                const double one[2] = {0., 1.};
                for (int s2 = 0; s2 < nseg; ++s2) HIP_CHECK(launch_rk_combo(y[s2], y[s2], comb(s2, 2, one, dt), (long long)sn[s2], st));
            } else {                        // rk4_alt_step_func: the 3/8 rule
                const double c2[1] = {1. / 3}, c3[2] = {-1. / 3, 1.}, c4[3] = {1., -1., 1.}, cs[4] = {0.125, 0.375, 0.375, 0.125};
                for (int s2 = 0; s2 < nseg; ++s2) HIP_CHECK(launch_rk_combo(ytmp[s2], y[s2], comb(s2, 1, c2, dt), (long long)sn[s2], st));
                F(t0 + dt * (1.0f / 3.0f), ytmp, KP(1));
                for (int s2 = 0; s2 < nseg; ++s2) HIP_CHECK(launch_rk_combo(ytmp[s2], y[s2], comb(s2, 2, c3, dt), (long long)sn[s2], st));
                F(t0 + dt * (2.0f / 3.0f), ytmp, KP(2));
                for (int s2 = 0; s2 < nseg; ++s2) HIP_CHECK(launch_rk_combo(ytmp[s2], y[s2], comb(s2, 3, c4, dt), (long long)sn[s2], st));
                F(t1, ytmp, KP(3));
                for (int s2 = 0; s2 < nseg; ++s2) HIP_CHECK(launch_rk_combo(y[s2], y[s2], comb(s2, 4, cs, dt), (long long)sn[s2], st));
            }
            if (wants_row(i + 1)) save_from(y);
        }
    } else {
        // ---- dopri5: RKAdaptiveStepsizeODESolver (rk_common.py) ----
        auto norm_of = [&](auto&& launch_one) {        // mixed norm: max over segments of the rms (misc.py _mixed_norm / _rms_norm)
            double best = 0.0;
            for (int s2 = 0; s2 < nseg; ++s2) {
                launch_one(s2);
                best = std::max(best, std::sqrt(fetch() / (double)sn[s2]));
            }
            return best;
        };
        const double t_first = sign * (double)rd->t_grid[0];
        F((float)t_first, y, KP(0));
        // _select_initial_step(func, t0, y0, order - 1 = 4, rtol, atol, norm, f0)
        const double d0 = norm_of([&](int s2) { HIP_CHECK(launch_scaled_sumsq(red + RED_PARTIALS, red, y[s2], nullptr, y[s2], rtol, atol, (long long)sn[s2], st)); });
        const double d1 = norm_of([&](int s2) { HIP_CHECK(launch_scaled_sumsq(red + RED_PARTIALS, red, k[s2][0], nullptr, y[s2], rtol, atol, (long long)sn[s2], st)); });
        const double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
        {
            const double one[1] = {1.};
            for (int s2 = 0; s2 < nseg; ++s2) HIP_CHECK(launch_rk_combo(ytmp[s2], y[s2], comb(s2, 1, one, h0), (long long)sn[s2], st));
        }
        F((float)(t_first + h0), ytmp, KP(1));
        const double d2 = norm_of([&](int s2) { HIP_CHECK(launch_scaled_sumsq(red + RED_PARTIALS, red, k[s2][1], k[s2][0], y[s2], rtol, atol, (long long)sn[s2], st)); }) / h0;
        const double h1 = (d1 <= 1e-15 && d2 <= 1e-15) ? std::max(1e-6, h0 * 1e-3) : std::pow(0.01 / std::max(d1, d2), 1.0 / 5.0);
        double dt = std::min(100.0 * h0, h1);
        double t0 = t_first, t1 = t_first;            // interpolation interval of the last accepted step
        long long attempts = 0;
        for (int i = 1; i < N; ++i) {
            const double next_t = sign * (double)rd->t_grid[i];
            while (next_t > t1) {
                if (++attempts > 10000000LL) return fail(TI_E_NAN, "dopri5: more than 1e7 step attempts");
                const double ts = t1, te = ts + dt;
                if (!(te > ts)) return fail(TI_E_NAN, "dopri5: step size underflow (dt = " + std::to_string(dt) + ")");
                const float tsf = (float)ts, dtf = (float)dt, tef = (float)te;
                for (int sidx = 0; sidx < 6; ++sidx) {                       // _runge_kutta_step
                    const float ti = dp5::alpha[sidx] == 1.0 ? std::nextafterf(tef, tef - 1.0f) : tsf + (float)dp5::alpha[sidx] * dtf;
                    float* const* dst = sidx == 5 ? ynew : ytmp;             // c_sol == beta[5]: the last stage input IS y1
                    for (int s2 = 0; s2 < nseg; ++s2)
                        HIP_CHECK(launch_rk_combo(dst[s2], y[s2], comb(s2, sidx + 1, dp5::beta[sidx], dtf), (long long)sn[s2], st));
                    F(ti, dst, KP(sidx + 1));
                }
                const double ratio = norm_of([&](int s2) {                  // _compute_error_ratio
                    HIP_CHECK(launch_rk_ratio_sumsq(red + RED_PARTIALS, red, y[s2], ynew[s2], comb(s2, 7, dp5::c_error, dtf), rtol, atol, (long long)sn[s2], st));
                });
                if (!(ratio == ratio)) return fail(TI_E_NAN, "dopri5: non-finite error estimate");
                if (ratio <= 1.0) {                                          // accept: dense output, FSAL
                    for (int s2 = 0; s2 < nseg; ++s2) {
                        HIP_CHECK(launch_interp_fit(coef[s2], y[s2], ynew[s2], k[s2][0], k[s2][6], comb(s2, 7, dp5::c_mid, dtf), dtf, (long long)sn[s2], st));
                        HIP_CHECK(hipMemcpyAsync(y[s2], ynew[s2], sn[s2] * sizeof(float), hipMemcpyDeviceToDevice, st));
                        std::swap(k[s2][0], k[s2][6]);
                    }
                    t0 = ts; t1 = te;
                }
                // _optimal_step_size(dt, ratio, safety 0.9, ifactor 10, dfactor 0.2, order 5)
                if (ratio == 0.0) dt *= 10.0;
                else dt *= std::min(10.0, std::max(0.9 / std::pow(ratio, 0.2), ratio < 1.0 ? 1.0 : 0.2));
            }
            if (wants_row(i)) {                                              // _interp_evaluate at the requested time
                const float xrel = (float)((next_t - t0) / (t1 - t0));
                for (int s2 = 0; s2 < nseg; ++s2) HIP_CHECK(launch_interp_eval(ytmp[s2], coef[s2], xrel, (long long)sn[s2], st));
                save_from(ytmp);
            }
        }
    }
    HIP_CHECK(hipMemsetAsync(h->nanflag.p, 0, sizeof(int), st));
    HIP_CHECK(launch_nan_check(x, (long long)n, h->nanflag.p, st));
    int flag = 0;
    HIP_CHECK(hipMemcpyAsync(&flag, h->nanflag.p, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    if (n_fevals) *n_fevals = fe;
    return flag ? fail(TI_E_NAN, "non-finite value in the final state") : TI_OK;
}

int check_rollout_desc(const ti_rollout_desc* rd)
{
    if (!rd || !rd->t_grid) return fail(TI_E_ARG, "rollout desc / t_grid is NULL");
    if (rd->n_step < 1) return fail(TI_E_ARG, "n_step must be >= 1");
    if (rd->scheme < TI_SCHEME_EULER || rd->scheme > TI_SCHEME_RK4) return fail(TI_E_ARG, "unknown scheme");
    if (rd->scheme == TI_SCHEME_DOPRI5 && !(rd->rtol > 0.f && rd->atol > 0.f)) return fail(TI_E_ARG, "dopri5 needs rtol > 0 and atol > 0");
    if (rd->scheme >= TI_SCHEME_DOPRI5)
        for (int k = 0; k + 2 < rd->n_step; ++k)
            if ((rd->t_grid[k + 1] > rd->t_grid[k]) != (rd->t_grid[k + 2] > rd->t_grid[k + 1]) || rd->t_grid[k + 1] == rd->t_grid[k])
                return fail(TI_E_ARG, "t_grid must be strictly monotonic");
    if (rd->mem != TI_MEM_HOST && rd->mem != TI_MEM_DEVICE) return fail(TI_E_ARG, "unknown mem kind");
    if (rd->eps < 0.f) return fail(TI_E_ARG, "eps must be >= 0");
    if (rd->step_offset < 0 || rd->step_offset + rd->n_step > 0x7fffffffLL) return fail(TI_E_ARG, "step_offset out of range");
    return TI_OK;
}

template <typename Fn>
int guarded(Fn&& fn)
{
    try { return fn(); }
    catch (const HipError& e) { return fail(TI_E_HIP, e.what()); }
    catch (const std::bad_alloc&) { return fail(TI_E_ALLOC, "host allocation failed"); }
    catch (const std::exception& e) { return fail(TI_E_ARG, e.what()); }
}

ti_handle* new_handle(int kind, int device)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) throw HipError("no HIP device available (libti_hip has no CPU fallback)");
    if (device < 0 || device >= ndev) throw std::invalid_argument("device index out of range");
    HIP_CHECK(hipSetDevice(device));
    std::unique_ptr<ti_handle> h(new ti_handle());
    h->kind = kind; h->device = device;
    HIP_CHECK(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
    h->stream = h->own_stream;
    h->nanflag.alloc(1);
    return h.release();
}

}  // namespace

// ====================================================================================================== C ABI
extern "C" {

int ti_version(void) { return TI_ABI_VERSION; }

int ti_device_count(void)
{
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

const char* ti_last_error(void) { return g_err.c_str(); }

int64_t ti_rollout_rows(int32_t n_step, int32_t save_every)
{
    if (save_every <= 0) return 1;
    const int64_t steps = n_step - 1;
    return steps / save_every + 1 + (steps % save_every != 0);
}

ti_handle* ti_painn_create(const ti_painn_desc* d, const float* weights, size_t n_weights, const int32_t* edge_src,
                           const int32_t* edge_dst, const int32_t* edge_type, const int32_t* atom_ids, int device)
{
    ti_handle* out = nullptr;
    const int rc = guarded([&]() -> int {
        if (!d || !weights || !atom_ids) return fail(TI_E_ARG, "NULL argument");
        const int F = d->n_features, L = d->n_layers, A = d->n_atoms, E = d->n_edges;
        if (F != 32 && F != 64 && F != 128 && F != 256) return fail(TI_E_UNSUPPORTED, "n_features must be 32, 64, 128 or 256");
        if (L < 1) return fail(TI_E_ARG, "n_layers must be >= 1");
        if (A < 1 || A > 32) return fail(TI_E_UNSUPPORTED, "n_atoms must be in 1..32 (the reference caps it at n_types = 25)");
        if (E < 0 || (E > 0 && (!edge_src || !edge_dst || !edge_type))) return fail(TI_E_ARG, "edge arrays missing");
        if (d->variant < 0 || d->variant > 2) return fail(TI_E_ARG, "unknown variant");
        if (d->n_types < 1) return fail(TI_E_ARG, "n_types must be >= 1");
        if (d->precision != TI_PREC_F32 && d->precision != TI_PREC_F16X2 && d->precision != TI_PREC_F16) return fail(TI_E_ARG, "unknown precision");
        for (int k = 0; k < E; ++k)
            if (edge_src[k] < 0 || edge_src[k] >= A || edge_dst[k] < 0 || edge_dst[k] >= A || edge_type[k] < 0 || edge_type[k] > 3)
                return fail(TI_E_ARG, "edge index / type out of range");
        for (int a = 0; a < A; ++a) if (atom_ids[a] < 0 || atom_ids[a] >= d->n_types) return fail(TI_E_ARG, "atom id out of range");
        std::unique_ptr<ti_handle> h(new_handle(0, device));
        h->d = *d; h->NB = F / 32;
        h->nE = d->variant == TI_VARIANT_AMBIENT ? 4 : d->variant == TI_VARIANT_LATENT_MULTI ? 3 : 2;
        h->ncond = d->variant == TI_VARIANT_AMBIENT ? 2 : d->variant == TI_VARIANT_LATENT_MULTI ? 1 : 0;
        // canonical layout offsets (include/ti_hip.h)
        size_t o = 0;
        h->edge_emb = o; o += 4 * (size_t)F; h->atom_emb = o; o += (size_t)d->n_types * F;
        o = take_mlp(h->embed, o, h->nE * F, F, F);
        h->phi.resize(L); h->w.resize(L); h->upd.resize(L); h->U.resize(L); h->V.resize(L);
        for (int l = 0; l < L; ++l) {
            o = take_mlp(h->phi[l], o, 2 * F, F, 5 * F); o = take_mlp(h->w[l], o, F, F, 5 * F);
            h->U[l] = o; o += (size_t)F * F; h->V[l] = o; o += (size_t)F * F;
            o = take_mlp(h->upd[l], o, 2 * F, F, 3 * F);
        }
        o = take_mlp(h->readout, o, F, F, 2);
        h->Vr = o; o += F;
        if (o != n_weights) return fail(TI_E_ARG, "weight count mismatch: expected " + std::to_string(o) + ", got " + std::to_string(n_weights));
        h->b2_gate = weights[h->readout.b2 + 1];
        // natural-order copy; Vr follows the 2-float readout bias in the canonical layout, so a 16-byte aligned copy of it
        // is appended for the kernels' float4 loads
        std::vector<float> flat(weights, weights + n_weights);
        if (d->precision != TI_PREC_F32)            // the weights' hi halves are plain fp16: refuse what would round to inf
            for (size_t i = 0; i < n_weights; ++i)
                if (!(std::fabs(weights[i]) < 65504.0f))
                    return fail(TI_E_UNSUPPORTED, "precisions f16x2 / f16 need every weight to be finite and below 65504 in magnitude (weight " + std::to_string(i) + ")");
        while (flat.size() % 4) flat.push_back(0.f);
        const size_t vr_aligned = flat.size();
        flat.insert(flat.end(), weights + h->Vr, weights + h->Vr + F);
        h->Vr = vr_aligned;
        h->flat.upload(flat);
        h->atom_ids.upload(std::vector<int32_t>(atom_ids, atom_ids + A));
        build_templates(h.get(), edge_src, edge_dst, edge_type);
        h->has_pair = pair_kernel_exists(h->NB, d->precision) && build_pair_template(h.get(), edge_src, edge_dst, edge_type);
        select_template(h.get(), 1 << 20);
        pack_painn(h.get(), weights);
        HIP_CHECK(configure_painn_kernels(h->NB));
        HIP_CHECK(configure_painn_jvp_kernels(h->NB));
        out = h.release();
        return TI_OK;
    });
    return rc == TI_OK ? out : nullptr;
}

int ti_reserve(ti_handle* h, int64_t B)
{
    if (!h || B < 0) return fail(TI_E_ARG, "bad handle / B");
    return guarded([&]() -> int { set_device(h); if (h->kind == 0) ensure_painn_ws(h, B); else ensure_adw_ws(h, B); return TI_OK; });
}

int ti_painn_drift(ti_handle* h, const float* x, float t, const float* cond, int64_t B, float* out, int mem)
{
    if (!h || h->kind != 0) return fail(TI_E_ARG, "not a painn handle");
    if (B < 0 || (B > 0 && (!x || !out || (h->ncond > 0 && !cond)))) return fail(TI_E_ARG, "NULL buffer");
    if (B == 0) return TI_OK;
    return guarded([&]() -> int {
        set_device(h);
        select_template(h, B);
        ensure_painn_ws(h, B);
        const size_t n = (size_t)B * h->d.n_atoms * 3, nc = (size_t)B * h->d.n_atoms * h->ncond;
        const float *xd = x, *cd = cond; float* od = out;
        if (mem == TI_MEM_HOST) {
            HIP_CHECK(hipMemcpyAsync(h->x.p, x, n * sizeof(float), hipMemcpyHostToDevice, h->stream));
            if (nc) HIP_CHECK(hipMemcpyAsync(h->cond.p, cond, nc * sizeof(float), hipMemcpyHostToDevice, h->stream));
            xd = h->x.p; cd = h->cond.p; od = h->b1.p;
        }
        painn_drift_dev(h, xd, t, cd, B, od);
        if (mem == TI_MEM_HOST && h->tap < 0) HIP_CHECK(hipMemcpyAsync(out, od, n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
        HIP_CHECK(hipStreamSynchronize(h->stream));
        return TI_OK;
    });
}

int ti_painn_rollout(ti_handle* h, const ti_rollout_desc* rd, const float* x0, const float* cond, int64_t B, float* out_path,
                     int64_t* n_fevals)
{
    if (!h || h->kind != 0) return fail(TI_E_ARG, "not a painn handle");
    if (int rc = check_rollout_desc(rd)) return rc;
    if (B < 0 || (B > 0 && (!x0 || !out_path || (h->ncond > 0 && !cond)))) return fail(TI_E_ARG, "NULL buffer");
    if (B == 0) { if (n_fevals) *n_fevals = 0; return TI_OK; }
    return guarded([&]() -> int {
        set_device(h);
        select_template(h, B);
        ensure_painn_ws(h, B);
        const int A = h->d.n_atoms;
        const size_t n = (size_t)B * A * 3, nc = (size_t)B * A * h->ncond;
        const hipMemcpyKind in_kind = rd->mem == TI_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
        HIP_CHECK(hipMemcpyAsync(h->x.p, x0, n * sizeof(float), in_kind, h->stream));
        const float* cd = cond;
        if (rd->mem == TI_MEM_HOST && nc) { HIP_CHECK(hipMemcpyAsync(h->cond.p, cond, nc * sizeof(float), hipMemcpyHostToDevice, h->stream)); cd = h->cond.p; }
        const int saved_tap = h->tap; h->tap = -1;
        auto drift = [&](const float* xs, float t, float* o, float*) { painn_drift_dev(h, xs, t, cd, B, o); };
        const int rc = rd->scheme >= TI_SCHEME_DOPRI5 ? rollout_rk(h, rd, h->x.p, n, out_path, n_fevals, drift)
                                                      : rollout_common(h, rd, h->x.p, h->b1.p, h->b2.p, h->xt.p, n, B, A * 3, A, out_path, n_fevals, drift);
        h->tap = saved_tap;
        return rc;
    });
}

int ti_painn_drift_jvp(ti_handle* h, const float* x, const float* xdot, float t, const float* cond, int64_t B, float* out,
                       float* out_tan, int mem)
{
    if (!h || h->kind != 0) return fail(TI_E_ARG, "not a painn handle");
    if (h->d.precision == TI_PREC_F16) return fail(TI_E_UNSUPPORTED, "the fp16 storage mode has no divergence / tangent path (use f32 or f16x2)");
    if (B < 0 || (B > 0 && (!x || !xdot || !out || !out_tan || (h->ncond > 0 && !cond)))) return fail(TI_E_ARG, "NULL buffer");
    if (B == 0) return TI_OK;
    return guarded([&]() -> int {
        set_device(h);
        select_template(h, B, false);
        ensure_painn_ws(h, B);
        ensure_jvp_ws(h, B, 1);
        const size_t n = (size_t)B * h->d.n_atoms * 3, nc = (size_t)B * h->d.n_atoms * h->ncond;
        const float *xd = x, *td = xdot, *cd = cond; float *od = out, *otd = out_tan;
        if (mem == TI_MEM_HOST) {
            HIP_CHECK(hipMemcpyAsync(h->x.p, x, n * sizeof(float), hipMemcpyHostToDevice, h->stream));
            HIP_CHECK(hipMemcpyAsync(h->xt.p, xdot, n * sizeof(float), hipMemcpyHostToDevice, h->stream));
            if (nc) HIP_CHECK(hipMemcpyAsync(h->cond.p, cond, nc * sizeof(float), hipMemcpyHostToDevice, h->stream));
            xd = h->x.p; td = h->xt.p; cd = h->cond.p; od = h->b1.p; otd = h->tout.p;
        }
        JvpRun jr{1, td, otd};
        painn_drift_dev(h, xd, t, cd, B, od, &jr);
        if (mem == TI_MEM_HOST && h->tap < 0) {
            HIP_CHECK(hipMemcpyAsync(out, od, n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
            HIP_CHECK(hipMemcpyAsync(out_tan, otd, n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
        }
        HIP_CHECK(hipStreamSynchronize(h->stream));
        return TI_OK;
    });
}

int ti_painn_drift_div(ti_handle* h, const float* x, float t, const float* cond, int64_t B, float* out, float* out_div, int mem)
{
    if (!h || h->kind != 0) return fail(TI_E_ARG, "not a painn handle");
    if (h->d.precision == TI_PREC_F16) return fail(TI_E_UNSUPPORTED, "the fp16 storage mode has no divergence / tangent path (use f32 or f16x2)");
    if (B < 0 || (B > 0 && (!x || !out || !out_div || (h->ncond > 0 && !cond)))) return fail(TI_E_ARG, "NULL buffer");
    if (B == 0) return TI_OK;
    if (h->tap >= 0) return fail(TI_E_ARG, "debug taps apply to ti_painn_drift / ti_painn_drift_jvp only");
    return guarded([&]() -> int {
        set_device(h);
        select_template(h, B, false);
        ensure_painn_ws(h, B);
        const size_t n = (size_t)B * h->d.n_atoms * 3, nc = (size_t)B * h->d.n_atoms * h->ncond;
        const float *xd = x, *cd = cond; float *od = out, *dd = out_div;
        if (mem == TI_MEM_HOST) {
            HIP_CHECK(hipMemcpyAsync(h->x.p, x, n * sizeof(float), hipMemcpyHostToDevice, h->stream));
            if (nc) HIP_CHECK(hipMemcpyAsync(h->cond.p, cond, nc * sizeof(float), hipMemcpyHostToDevice, h->stream));
            xd = h->x.p; cd = h->cond.p; od = h->b1.p; dd = h->divb.p;
        }
        painn_drift_div_dev(h, xd, t, cd, B, od, dd);
        if (mem == TI_MEM_HOST) {
            HIP_CHECK(hipMemcpyAsync(out, od, n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
            HIP_CHECK(hipMemcpyAsync(out_div, dd, (size_t)B * sizeof(float), hipMemcpyDeviceToHost, h->stream));
        }
        HIP_CHECK(hipStreamSynchronize(h->stream));
        return TI_OK;
    });
}

int ti_painn_rollout_dlogp(ti_handle* h, const ti_rollout_desc* rd, const float* x0, const float* cond, int64_t B, float div_scale,
                           float out_scale, int reverse_ode, float* out_path, float* out_dlogp, int64_t* n_fevals)
{
    if (!h || h->kind != 0) return fail(TI_E_ARG, "not a painn handle");
    if (h->d.precision == TI_PREC_F16) return fail(TI_E_UNSUPPORTED, "the fp16 storage mode has no divergence / tangent path (use f32 or f16x2)");
    if (int rc = check_rollout_desc(rd)) return rc;
    if (rd->scheme == TI_SCHEME_EM) return fail(TI_E_UNSUPPORTED, "dlogp is defined for the deterministic schemes only (EULER, HEUN)");
    if (B < 0 || (B > 0 && (!x0 || !out_path || !out_dlogp || (h->ncond > 0 && !cond)))) return fail(TI_E_ARG, "NULL buffer");
    if (B == 0) { if (n_fevals) *n_fevals = 0; return TI_OK; }
    return guarded([&]() -> int {
        set_device(h);
        select_template(h, B, false);
        ensure_painn_ws(h, B);
        const int A = h->d.n_atoms;
        const size_t n = (size_t)B * A * 3, nc = (size_t)B * A * h->ncond;
        const hipMemcpyKind in_kind = rd->mem == TI_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
        HIP_CHECK(hipMemcpyAsync(h->x.p, x0, n * sizeof(float), in_kind, h->stream));
        const float* cd = cond;
        if (rd->mem == TI_MEM_HOST && nc) { HIP_CHECK(hipMemcpyAsync(h->cond.p, cond, nc * sizeof(float), hipMemcpyHostToDevice, h->stream)); cd = h->cond.p; }
        const int saved_tap = h->tap; h->tap = -1;
        DlogpAux aux;
        aux.dl = h->dl.p; aux.d1 = h->divb.p; aux.d2 = h->div2.p; aux.scaled = h->dlscaled.p; aux.out = out_dlogp;
        aux.n_dl = (size_t)B; aux.div_scale = div_scale; aux.out_scale = out_scale;
        auto drift = [&](const float* xs, float t, float* o, float* dv) {
            painn_drift_div_dev(h, xs, t, cd, B, o, dv);
            if (reverse_ode) {      // (-b, +div): ode_wrapper.py:49
                HIP_CHECK(launch_scale(o, o, -1.0f, (long long)n, h->stream));
                HIP_CHECK(launch_scale(dv, dv, -1.0f, (long long)B, h->stream));
            }
        };
        const int rc = rd->scheme >= TI_SCHEME_DOPRI5 ? rollout_rk(h, rd, h->x.p, n, out_path, n_fevals, drift, aux)
                                                      : rollout_common(h, rd, h->x.p, h->b1.p, h->b2.p, h->xt.p, n, B, A * 3, A, out_path, n_fevals, drift, aux);
        h->tap = saved_tap;
        return rc;
    });
}

int ti_painn_debug_tap(ti_handle* h, int stage)
{
    if (!h || h->kind != 0) return fail(TI_E_ARG, "not a painn handle");
    if (h->d.precision == TI_PREC_F16 && stage >= 0) return fail(TI_E_UNSUPPORTED, "debug taps read fp32 state; not available in the fp16 storage mode");
    h->tap = stage;
    return TI_OK;
}

int ti_painn_debug_poison(ti_handle* h, int64_t B, float value)
{
    if (!h || h->kind != 0 || B <= 0) return fail(TI_E_ARG, "not a painn handle / B");
    return guarded([&]() -> int {
        set_device(h);
        ensure_painn_ws(h, B);
        const size_t N = (size_t)B * h->d.n_atoms, F = h->d.n_features;
        const unsigned bits = __builtin_bit_cast(unsigned, value);
        HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)h->dsacc.p, (int)bits, N * F, h->stream));
        HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)h->dvacc.p, (int)bits, N * 3 * F, h->stream));
        HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)h->cacc.p, (int)bits, N * 3 * F, h->stream));
        return TI_OK;
    });
}

int ti_painn_debug_read(ti_handle* h, int what, float* out, size_t n_floats)
{
    if (!h || h->kind != 0 || !out) return fail(TI_E_ARG, "bad argument");
    return guarded([&]() -> int {
        set_device(h);
        const size_t A = h->d.n_atoms, F = h->d.n_features, E = h->d.n_edges, B = (size_t)h->last_B, N = B * A;
        HIP_CHECK(hipStreamSynchronize(h->stream));
        if (what == 0) {
            if (n_floats != N * F) return fail(TI_E_ARG, "size mismatch (s)");
            std::vector<float> ds(n_floats);                  // pending invariant messages (zero after an update stage)
            HIP_CHECK(hipMemcpy(out, h->s.p, n_floats * sizeof(float), hipMemcpyDeviceToHost));
            HIP_CHECK(hipMemcpy(ds.data(), h->dsacc.p, n_floats * sizeof(float), hipMemcpyDeviceToHost));
            for (size_t i = 0; i < n_floats; ++i) out[i] += ds[i];
        } else if (what == 1) {
            // v as the reference sees it at the tap: v + dvacc + cacc x v (the accumulators are zero after an update stage)
            if (n_floats != N * 3 * F) return fail(TI_E_ARG, "size mismatch (v)");
            std::vector<float> v(n_floats), dv(n_floats), cc(n_floats);
            HIP_CHECK(hipMemcpy(v.data(), h->v.p, n_floats * sizeof(float), hipMemcpyDeviceToHost));
            HIP_CHECK(hipMemcpy(dv.data(), h->dvacc.p, n_floats * sizeof(float), hipMemcpyDeviceToHost));
            HIP_CHECK(hipMemcpy(cc.data(), h->cacc.p, n_floats * sizeof(float), hipMemcpyDeviceToHost));
            for (size_t nd = 0; nd < N; ++nd)
                for (size_t f = 0; f < F; ++f)
                    for (int c = 0; c < 3; ++c) {
                        const int c1 = (c + 1) % 3, c2 = (c + 2) % 3;
                        auto at = [&](const std::vector<float>& a, int cc2) { return a[(nd * 3 + cc2) * F + f]; };
                        out[(nd * 3 + c) * F + f] = (at(v, c) + at(dv, c)) + (at(cc, c1) * at(v, c2) - at(cc, c2) * at(v, c1));
                    }
        } else if (what == 2) {
            if (n_floats != B * E * F) return fail(TI_E_ARG, "size mismatch (e)");
            const size_t RB = ti::EDGE_ROWS_PER_BLOCK, rows = (B + h->G - 1) / h->G * h->parts * h->nblk * RB * (h->active == 2 ? 2 : 1);
            std::vector<float> e(rows * F);
            HIP_CHECK(hipMemcpy(e.data(), h->e.p, e.size() * sizeof(float), hipMemcpyDeviceToHost));
            for (size_t m = 0; m < B; ++m)
                for (size_t k = 0; k < E; ++k)                                   // k = sorted position
                    std::memcpy(out + (m * E + h->perm[k]) * F, e.data() + edge_row_of(h, m, k) * F, F * sizeof(float));
        } else if (what >= 3 && what <= 5) {
            // tangents of the last ti_painn_drift_jvp call (one direction per molecule), composed like their primal twins
            if (h->last_D != 1 || (size_t)h->last_VB != B) return fail(TI_E_ARG, "tangent taps need a preceding ti_painn_drift_jvp call");
            auto fetch = [&](const DevBuf<float>& b, size_t n) { std::vector<float> v(n); HIP_CHECK(hipMemcpy(v.data(), b.p, n * sizeof(float), hipMemcpyDeviceToHost)); return v; };
            if (what == 3) {
                if (n_floats != N * F) return fail(TI_E_ARG, "size mismatch (ts)");
                const auto ts = fetch(h->ts, n_floats), td = fetch(h->tdsacc, n_floats);
                for (size_t i = 0; i < n_floats; ++i) out[i] = ts[i] + td[i];
            } else if (what == 4) {
                if (n_floats != N * 3 * F) return fail(TI_E_ARG, "size mismatch (tv)");
                const auto v = fetch(h->v, n_floats), cc = fetch(h->cacc, n_floats);
                const auto tv = fetch(h->tv, n_floats), tdv = fetch(h->tdvacc, n_floats), tcc = fetch(h->tcacc, n_floats);
                for (size_t nd = 0; nd < N; ++nd)
                    for (size_t f = 0; f < F; ++f)
                        for (int c = 0; c < 3; ++c) {
                            const int c1 = (c + 1) % 3, c2 = (c + 2) % 3;
                            auto at = [&](const std::vector<float>& a, int cc2) { return a[(nd * 3 + cc2) * F + f]; };
                            out[(nd * 3 + c) * F + f] = (at(tv, c) + at(tdv, c)) + ((at(tcc, c1) * at(v, c2) + at(cc, c1) * at(tv, c2)) -
                                                                                     (at(tcc, c2) * at(v, c1) + at(cc, c2) * at(tv, c1)));
                        }
            } else {
                if (n_floats != B * E * F) return fail(TI_E_ARG, "size mismatch (te)");
                const size_t RB = ti::EDGE_ROWS_PER_BLOCK, rows = (B + h->G - 1) / h->G * h->parts * h->nblk * RB;
                const auto e = fetch(h->te, rows * F);
                for (size_t m = 0; m < B; ++m)
                    for (size_t k = 0; k < E; ++k)
                        std::memcpy(out + (m * E + h->perm[k]) * F, e.data() + edge_row_of(h, m, k) * F, F * sizeof(float));
            }
        } else return fail(TI_E_ARG, "unknown tap");
        return TI_OK;
    });
}

// ------------------------------------------------------------------------------------------------------------ adw
ti_handle* ti_adw_create(const ti_adw_desc* d, const double* weights, size_t n_weights, int device)
{
    ti_handle* out = nullptr;
    const int rc = guarded([&]() -> int {
        if (!d || !weights) return fail(TI_E_ARG, "NULL argument");
        const int H = d->hidden_size, nl = d->num_layers;
        if (H != 32 && H != 64 && H != 128 && H != 256) return fail(TI_E_UNSUPPORTED, "hidden_size must be 32, 64, 128 or 256");
        if (nl < 1) return fail(TI_E_ARG, "num_layers must be >= 1");
        if (d->precision != TI_PREC_F32 && d->precision != TI_PREC_F16X2) return fail(TI_E_ARG, "unknown precision");
        const size_t need = (size_t)H * 3 + H + (size_t)H * H + H + H + 1 + (size_t)H * 3 + H + (size_t)(nl - 1) * ((size_t)H * H + H) + H + 1;
        if (n_weights != need) return fail(TI_E_ARG, "weight count mismatch: expected " + std::to_string(need) + ", got " + std::to_string(n_weights));
        std::unique_ptr<ti_handle> h(new_handle(1, device));
        h->ad = *d; h->NB = H / 32;
        std::vector<float> w(n_weights);
        for (size_t i = 0; i < n_weights; ++i) w[i] = (float)weights[i];        // the device computes in fp32
        const int NB = h->NB, NBK = H / 16;
        const bool split = d->precision == TI_PREC_F16X2;
        std::vector<float> nat, pk;
        auto chunk16 = [&](const float* W, int row0) {
            if (split) pack_chunk16_split(pk, W, H, H, row0, 0, NBK); else pack_chunk16(pk, W, H, H, row0, 0, NBK);
        };
        // one MLP block: canonical order  W_in[H,3] b_in[H] (W_h[H,H] b_h[H]) x n_hidden  W_out[1,H] b_out[1]
        auto take_mlp = [&](size_t& o, int n_hidden, size_t& vec_off, Stream& st, float& b_out) {
            vec_off = nat.size();
            nat.insert(nat.end(), &w[o], &w[o] + (size_t)H * 3); o += (size_t)H * 3;      // w_in
            nat.insert(nat.end(), &w[o], &w[o] + H); o += H;                              // b_in
            st.off4 = pk.size() / 4;
            std::vector<float> bh;
            for (int l = 0; l < n_hidden; ++l) {
                for (int nbo = 0; nbo < NB; ++nbo) chunk16(&w[o], 32 * nbo);
                o += (size_t)H * H;
                bh.insert(bh.end(), &w[o], &w[o] + H); o += H;
            }
            st.nch = n_hidden * NB;
            nat.insert(nat.end(), bh.begin(), bh.end());
            nat.insert(nat.end(), &w[o], &w[o] + H); o += H;                              // w_out
            b_out = w[o]; o += 1;
        };
        size_t o = 0;
        take_mlp(o, 1, h->a_be_vecs, h->st_be, h->a_be_b_out);
        take_mlp(o, nl - 1, h->a_net_vecs, h->st_net, h->a_b_out);
        if (pk.empty()) pk.assign(4, 0.f);
        h->flat.upload(nat); h->packed.upload(pk);
        HIP_CHECK(configure_adw_kernels(NB, std::max(1, nl - 1)));
        out = h.release();
        return TI_OK;
    });
    return rc == TI_OK ? out : nullptr;
}

static int adw_drift_impl(ti_handle* h, const float* x, float t, const float* beta0, const float* beta1, int64_t B, float* out,
                          float* out_div, int mem)
{
    if (!h || h->kind != 1) return fail(TI_E_ARG, "not an adw handle");
    if (B < 0 || (B > 0 && (!x || !beta0 || !beta1 || !out))) return fail(TI_E_ARG, "NULL buffer");
    if (B == 0) return TI_OK;
    return guarded([&]() -> int {
        set_device(h);
        ensure_adw_ws(h, B);
        const long long U = adw_set_cond(h, beta0, beta1, B, mem);
        const float* xd = x; float* od = out; float* dd = out_div;
        if (mem == TI_MEM_HOST) {
            HIP_CHECK(hipMemcpyAsync(h->ax.p, x, B * sizeof(float), hipMemcpyHostToDevice, h->stream));
            xd = h->ax.p; od = h->ab1.p; dd = out_div ? h->ad1.p : nullptr;
        }
        adw_drift_dev(h, xd, t, U, B, od, dd);
        if (mem == TI_MEM_HOST) {
            HIP_CHECK(hipMemcpyAsync(out, od, B * sizeof(float), hipMemcpyDeviceToHost, h->stream));
            if (out_div) HIP_CHECK(hipMemcpyAsync(out_div, dd, B * sizeof(float), hipMemcpyDeviceToHost, h->stream));
        }
        HIP_CHECK(hipStreamSynchronize(h->stream));
        return TI_OK;
    });
}

int ti_adw_drift(ti_handle* h, const float* x, float t, const float* beta0, const float* beta1, int64_t B, float* out, int mem)
{
    return adw_drift_impl(h, x, t, beta0, beta1, B, out, nullptr, mem);
}

int ti_adw_drift_div(ti_handle* h, const float* x, float t, const float* beta0, const float* beta1, int64_t B, float* out, float* out_div, int mem)
{
    if (!out_div) return fail(TI_E_ARG, "out_div is NULL");
    return adw_drift_impl(h, x, t, beta0, beta1, B, out, out_div, mem);
}

static int adw_rollout_impl(ti_handle* h, const ti_rollout_desc* rd, const float* x0, const float* beta0, const float* beta1, int64_t B,
                            float* out_path, float* out_dlogp, int64_t* n_fevals)
{
    if (!h || h->kind != 1) return fail(TI_E_ARG, "not an adw handle");
    if (int rc = check_rollout_desc(rd)) return rc;
    if (B < 0 || (B > 0 && (!x0 || !beta0 || !beta1 || !out_path))) return fail(TI_E_ARG, "NULL buffer");
    if (out_dlogp && rd->scheme == TI_SCHEME_EM && rd->eps > 0.f)
        return fail(TI_E_UNSUPPORTED, "dlogp is defined for a deterministic flow: EM needs eps = 0");
    if (B == 0) { if (n_fevals) *n_fevals = 0; return TI_OK; }
    return guarded([&]() -> int {
        set_device(h);
        ensure_adw_ws(h, B);
        const long long U = adw_set_cond(h, beta0, beta1, B, rd->mem);
        const hipMemcpyKind in_kind = rd->mem == TI_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
        HIP_CHECK(hipMemcpyAsync(h->ax.p, x0, B * sizeof(float), in_kind, h->stream));
        DlogpAux aux;
        DevBuf<float> scaled_tmp;                    // dlogp * 1e2 staging for the saved rows
        if (out_dlogp) { scaled_tmp.alloc(B); aux.dl = h->adl.p; aux.d1 = h->ad1.p; aux.d2 = h->ad2.p; aux.scaled = scaled_tmp.p; aux.out = out_dlogp; }
        auto drift = [&](const float* xs, float t, float* o, float* dv) { adw_drift_dev(h, xs, t, U, B, o, dv); };
        if (rd->scheme >= TI_SCHEME_DOPRI5) return rollout_rk(h, rd, h->ax.p, (size_t)B, out_path, n_fevals, drift, aux);
        return rollout_common(h, rd, h->ax.p, h->ab1.p, h->ab2.p, h->axt.p, (size_t)B, B, 1, 0, out_path, n_fevals, drift, aux);
    });
}

int ti_adw_rollout(ti_handle* h, const ti_rollout_desc* rd, const float* x0, const float* beta0, const float* beta1, int64_t B,
                   float* out_path, int64_t* n_fevals)
{
    return adw_rollout_impl(h, rd, x0, beta0, beta1, B, out_path, nullptr, n_fevals);
}

int ti_adw_rollout_dlogp(ti_handle* h, const ti_rollout_desc* rd, const float* x0, const float* beta0, const float* beta1, int64_t B,
                         float* out_path, float* out_dlogp, int64_t* n_fevals)
{
    if (!out_dlogp) return fail(TI_E_ARG, "out_dlogp is NULL");
    return adw_rollout_impl(h, rd, x0, beta0, beta1, B, out_path, out_dlogp, n_fevals);
}

// --------------------------------------------------------------------------------------------------------- shared
void ti_destroy(ti_handle* h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    delete h;
}

int ti_set_stream(ti_handle* h, void* hip_stream, int mode)
{
    if (!h) return fail(TI_E_ARG, "NULL handle");
    if (mode != TI_STREAM_OWN && mode != TI_STREAM_EXTERNAL) return fail(TI_E_ARG, "unknown stream mode");
    h->stream = mode == TI_STREAM_EXTERNAL ? reinterpret_cast<hipStream_t>(hip_stream) : h->own_stream;
    return TI_OK;
}

int ti_wait_stream(ti_handle* h, void* producer_stream)
{
    if (!h) return fail(TI_E_ARG, "NULL handle");
    return guarded([&]() -> int {
        set_device(h);
        hipStream_t prod = reinterpret_cast<hipStream_t>(producer_stream);
        if (prod == h->stream) return TI_OK;                     // same stream: already ordered
        if (!h->wait_ev) HIP_CHECK(hipEventCreateWithFlags(&h->wait_ev, hipEventDisableTiming));
        HIP_CHECK(hipEventRecord(h->wait_ev, prod));
        HIP_CHECK(hipStreamWaitEvent(h->stream, h->wait_ev, 0));
        return TI_OK;
    });
}

int ti_painn_set_template(ti_handle* h, int which)
{
    if (!h || h->kind != 0) return fail(TI_E_ARG, "not a painn handle");
    if (which != TI_TEMPLATE_AUTO && which != TI_TEMPLATE_THROUGHPUT && which != TI_TEMPLATE_LATENCY && which != TI_TEMPLATE_PAIR) return fail(TI_E_ARG, "unknown template");
    if (which == TI_TEMPLATE_LATENCY && h->n_tpl < 2) which = TI_TEMPLATE_THROUGHPUT;      // this species has only one layout
    if (which == TI_TEMPLATE_PAIR && !h->has_pair) which = TI_TEMPLATE_THROUGHPUT;         // no pair-major layout for this graph / width / precision
    h->pinned_tpl = which;
    return TI_OK;
}

int ti_painn_template_for(ti_handle* h, int64_t B)
{
    if (!h || h->kind != 0) return fail(TI_E_ARG, "not a painn handle");
    return template_for(h, B);
}

int ti_profile_enable(ti_handle* h, int on)
{
    if (!h) return fail(TI_E_ARG, "NULL handle");
    h->prof = on != 0;
    return TI_OK;
}

int ti_profile_read(ti_handle* h, int kernel, int64_t* n_launches, double* total_ms)
{
    if (!h || kernel < 0 || kernel >= TI_KERNEL_COUNT) return fail(TI_E_ARG, "bad handle / kernel id");
    return guarded([&]() -> int {
        set_device(h);
        HIP_CHECK(hipStreamSynchronize(h->stream));
        double tot = 0; int64_t cnt = 0;
        for (auto& pr : h->ev[kernel]) {
            float ms = 0.f;
            HIP_CHECK(hipEventElapsedTime(&ms, pr.first, pr.second));
            tot += ms; ++cnt;
            (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second);
        }
        h->ev[kernel].clear();
        if (n_launches) *n_launches = cnt;
        if (total_ms) *total_ms = tot;
        return TI_OK;
    });
}

int ti_selftest(int device)
{
    return guarded([&]() -> int {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(TI_E_HIP, "no HIP device");
        HIP_CHECK(hipSetDevice(device));
        DevBuf<float> d; d.alloc(64 * 16);
        HIP_CHECK(launch_selftest(d.p, nullptr));
        std::vector<float> o(64 * 16);
        HIP_CHECK(hipMemcpy(o.data(), d.p, o.size() * sizeof(float), hipMemcpyDeviceToHost));
        // expected D[i][j] = sum_k A[i][k] B[k][j], A[i][k] = 1 + i + 100k, B[k][j] = 1000 + j - 7k;
        // accumulator register r of lane l holds row (r&3) + 8*(r>>2) + 4*(l>>5), column l&31
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 16; ++r) {
                const int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), j = l & 31;
                double ref = 0;
                for (int k = 0; k < 2; ++k) ref += (1.0 + i + 100.0 * k) * (1000.0 + j - 7.0 * k);
                if (std::fabs(o[l * 16 + r] - ref) > 1e-3 * std::fabs(ref))
                    return fail(TI_E_HIP, "MFMA 32x32x2 lane map differs from the layout the kernels assume (lane " + std::to_string(l) +
                                              ", reg " + std::to_string(r) + ")");
            }
        // the 8-instruction operand split (v_fma_mix lo/hi, half-register writes) against the plain arithmetic, 16.8 M values
        DevBuf<unsigned> cnt; cnt.alloc(2);
        HIP_CHECK(hipMemset(cnt.p, 0, 2 * sizeof(unsigned)));
        HIP_CHECK(launch_split_selftest(cnt.p, nullptr));
        unsigned bad[2] = {0, 0};
        HIP_CHECK(hipMemcpy(bad, cnt.p, 2 * sizeof(unsigned), hipMemcpyDeviceToHost));
        if (bad[0]) return fail(TI_E_HIP, "operand split: " + std::to_string(bad[0]) + " fp16 halves differ from the reference arithmetic (either format)");
        if (bad[1]) return fail(TI_E_HIP, "v_mfma_f32_16x16x32_f16 flushed fp16-subnormal inputs: the one-accumulator operand format needs them kept");
        return TI_OK;
    });
}

}  // extern "C"
