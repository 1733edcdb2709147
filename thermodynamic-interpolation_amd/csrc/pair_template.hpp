// pair_template.hpp -- builds the pair-major edge template of the message kernel (painn_pair_kernel.hpp; word formats in
// ti_internal.hpp).  Pure host C++ with no HIP dependency: tests/test_host_logic.py compiles it with g++ and checks its invariants
// on complete and sparse graphs.
//
// The filter branch w(enc(|r_ij|)) of SE3Message (/root/reference/mdqm9/thermo/ambient/models/cpainn.py:283-289) is shared by the
// edges i -> j and j -> i, so the kernel walks PAIRS: a row block is a 4 x 4 tile, row 4a + b = pair (I[a], J[b]) of up to four I
// slots and four J slots (a slot = molecule-in-group, atom).  Every pair of the G molecules of a group must lie in exactly one valid
// row.  Construction: atoms are cut into tiles of four by index; every tile pair with edges between them is a block; spare slots of
// those blocks then absorb the pairs INSIDE a tile (an atom may sit on both sides of a block), greedily by the number of uncovered
// pairs an added atom brings; what is left ("loose" pairs, of all G molecules) is packed into further blocks, preferring pairs that
// reuse slots already opened.  A complete graph on A atoms needs at least ceil((A - 1) / 4) slot places per atom, i.e.
// A * ceil((A - 1) / 4) / 8 blocks per molecule; this construction meets that bound for A = 18 (11.25 blocks per molecule at G = 4:
// 85 % of the rows are pairs), A = 9 (3 blocks) and A = 25 (21 blocks).
#pragma once
#include <stdint.h>

#include <algorithm>
#include <array>
#include <stdexcept>
#include <vector>

namespace ti {

struct PairTemplate {
    int G = 0, nblk = 0;                     // molecules per group, row blocks per group
    std::vector<uint32_t> rows;              // [nblk * 16] row words
    std::vector<int32_t> slotnode;           // [nblk * 16] slot words (k = 0..3 I slots, 4..7 J slots, rest unused = -1)
    std::vector<int> pair_pos;               // [(m * A + src) * A + dst] -> (block * 2 + direction) * 16 + pair row, -1 = no such edge
    // The kernel writes, per block and slot, the partial sum of that slot's rows (plain stores, no atomics); plist names, per atom of the
    // group, the partial rows that belong to it, in walk order: [(m * A + atom) * kmax + k] = block * 8 + slot, -1 = end of the list
    std::vector<int32_t> plist; int kmax = 0;
    double fill = 0.0;                       // valid rows / all rows
};

namespace pair_detail {
struct Blk {
    int I[4], J[4];                          // slot keys mol * 32 + atom, -1 = empty
    Blk() { for (int k = 0; k < 4; ++k) I[k] = J[k] = -1; }
    static int count(const int* s) { int n = 0; for (int k = 0; k < 4; ++k) n += s[k] >= 0; return n; }
    static bool has(const int* s, int key) { for (int k = 0; k < 4; ++k) if (s[k] == key) return true; return false; }
    static void add(int* s, int key) { for (int k = 0; k < 4; ++k) if (s[k] < 0) { s[k] = key; return; } }
};
}  // namespace pair_detail

// src / dst / etype: the E directed edges of ONE molecule (A <= 32 atoms).  Returns false (no pair template) when the directed graph
// is not the symmetric closure of an undirected one with one type per pair.
// first_touch: mark (PAIR_SLOT_FIRST_TOUCH), per atom, the first slot that holds it in walk order -- inside a block the J slots
// (direction A) before the I slots (direction B): the order the atomic variant of the kernel issues its accumulator updates in.
constexpr int32_t PAIR_SLOT_FIRST_TOUCH = 1 << 30;      // == ti::SLOT_FIRST_TOUCH (ti_internal.hpp)
inline bool build_pair_template(int A, int E, const int32_t* src, const int32_t* dst, const int32_t* etype, PairTemplate& out, bool first_touch = false)
{
    using pair_detail::Blk;
    if (E <= 0 || (E & 1) || A > 32) return false;
    std::vector<int> pt((size_t)A * A, -1);                          // type of the directed edge src -> dst
    for (int k = 0; k < E; ++k) {
        if (src[k] == dst[k] || pt[(size_t)src[k] * A + dst[k]] >= 0) return false;        // self loop / duplicate edge
        pt[(size_t)src[k] * A + dst[k]] = etype[k];
    }
    for (int i = 0; i < A; ++i)
        for (int j = 0; j < A; ++j)
            if (pt[(size_t)i * A + j] != pt[(size_t)j * A + i]) return false;              // j -> i missing or of another type
    auto exists = [&](int i, int j) { return i != j && pt[(size_t)i * A + j] >= 0; };
    const int NT = (A + 3) / 4;

    int bestG = 0; std::vector<Blk> best;
    for (int G : {1, 2, 4, 8}) {
        std::vector<Blk> blocks;
        std::vector<char> cov((size_t)G * A * A, 0);
        auto covered = [&](int m, int i, int j) -> char& { return cov[((size_t)m * A + std::min(i, j)) * A + std::max(i, j)]; };
        // rows a block would newly cover if atom u of molecule m joined its I side (to_I) or J side
        auto gain = [&](const Blk& b, bool to_I, int m, int u) {
            const int* other = to_I ? b.J : b.I;
            int g = 0;
            for (int k = 0; k < 4; ++k)
                if (other[k] >= 0 && other[k] / 32 == m) { const int x = other[k] % 32; if (exists(u, x) && !covered(m, u, x)) ++g; }
            return g;
        };
        auto mark = [&](const Blk& b) {
            for (int a = 0; a < 4; ++a)
                for (int c = 0; c < 4; ++c)
                    if (b.I[a] >= 0 && b.J[c] >= 0 && b.I[a] / 32 == b.J[c] / 32 && exists(b.I[a] % 32, b.J[c] % 32))
                        covered(b.I[a] / 32, b.I[a] % 32, b.J[c] % 32) = 1;
        };
        for (int m = 0; m < G; ++m) {
            const size_t first_block = blocks.size();
            for (int ta = 0; ta < NT; ++ta)
                for (int tb = ta + 1; tb < NT; ++tb) {
                    Blk b;
                    for (int i = 4 * ta; i < std::min(A, 4 * ta + 4); ++i)
                        for (int j = 4 * tb; j < std::min(A, 4 * tb + 4); ++j)
                            if (exists(i, j)) {
                                if (!Blk::has(b.I, m * 32 + i)) Blk::add(b.I, m * 32 + i);
                                if (!Blk::has(b.J, m * 32 + j)) Blk::add(b.J, m * 32 + j);
                            }
                    if (Blk::count(b.I) == 0) continue;
                    mark(b);
                    blocks.push_back(b);
                }
            for (;;) {                                   // spare slots absorb uncovered pairs of this molecule
                int bg = 0, bb = -1, bu = -1; bool b_to_I = false;
                for (size_t bi = first_block; bi < blocks.size(); ++bi)
                    for (int side = 0; side < 2; ++side) {
                        const Blk& b = blocks[bi];
                        const int* mine = side ? b.I : b.J;
                        if (Blk::count(mine) == 4) continue;
                        for (int u = 0; u < A; ++u) {
                            if (Blk::has(mine, m * 32 + u)) continue;
                            const int g = gain(b, side != 0, m, u);
                            if (g > bg) { bg = g; bb = (int)bi; bu = u; b_to_I = side != 0; }
                        }
                    }
                if (bg == 0) break;
                Blk& b = blocks[bb];
                Blk::add(b_to_I ? b.I : b.J, m * 32 + bu);
                mark(b);
            }
        }
        std::vector<std::array<int, 3>> loose;           // pairs nothing covers yet, of all G molecules
        for (int m = 0; m < G; ++m)
            for (int i = 0; i < A; ++i)
                for (int j = i + 1; j < A; ++j)
                    if (exists(i, j) && !covered(m, i, j)) loose.push_back({m, i, j});
        while (!loose.empty()) {
            Blk b;
            for (;;) {
                int best_new = 99, best_k = -1; bool best_swap = false;
                for (size_t k = 0; k < loose.size(); ++k) {
                    if (covered(loose[k][0], loose[k][1], loose[k][2])) continue;
                    for (int sw = 0; sw < 2; ++sw) {
                        const int ki = loose[k][0] * 32 + loose[k][sw ? 2 : 1], kj = loose[k][0] * 32 + loose[k][sw ? 1 : 2];
                        const int ni = Blk::has(b.I, ki) ? 0 : 1, nj = Blk::has(b.J, kj) ? 0 : 1;
                        if (Blk::count(b.I) + ni > 4 || Blk::count(b.J) + nj > 4) continue;
                        if (ni + nj < best_new) { best_new = ni + nj; best_k = (int)k; best_swap = sw != 0; }
                    }
                }
                if (best_k < 0) break;
                const int ki = loose[best_k][0] * 32 + loose[best_k][best_swap ? 2 : 1], kj = loose[best_k][0] * 32 + loose[best_k][best_swap ? 1 : 2];
                if (!Blk::has(b.I, ki)) Blk::add(b.I, ki);
                if (!Blk::has(b.J, kj)) Blk::add(b.J, kj);
                mark(b);
            }
            blocks.push_back(b);
            std::vector<std::array<int, 3>> rest;
            for (auto& l : loose) if (!covered(l[0], l[1], l[2])) rest.push_back(l);
            loose.swap(rest);
        }
        // fewest blocks per molecule wins; a larger G must save at least 2 % to be preferred (fewer, longer waves otherwise)
        if (bestG == 0 || (double)blocks.size() / G < 0.98 * (double)best.size() / bestG) { bestG = G; best = blocks; }
    }
    const int G = bestG, nblk = (int)best.size();
    constexpr int RB = 16;
    out.G = G; out.nblk = nblk;
    out.rows.assign((size_t)nblk * RB, 0u); out.slotnode.assign((size_t)nblk * RB, -1);
    out.pair_pos.assign((size_t)G * A * A, -1);
    std::vector<char> done((size_t)G * A * A, 0), touched((size_t)G * 32, 0);
    size_t n_valid = 0;
    for (int bi = 0; bi < nblk; ++bi) {
        const Blk& b = best[bi];
        int safeI = -1, safeJ = -1;                      // rows of empty slots point at an atom that exists
        for (int k = 0; k < 4; ++k) { if (safeI < 0 && b.I[k] >= 0) safeI = b.I[k]; if (safeJ < 0 && b.J[k] >= 0) safeJ = b.J[k]; }
        for (int a = 0; a < 4; ++a)
            for (int c = 0; c < 4; ++c) {
                const int ki = b.I[a] >= 0 ? b.I[a] : safeI, kj = b.J[c] >= 0 ? b.J[c] : safeJ;
                const int m = ki / 32, i = ki % 32, j = kj % 32;
                bool valid = b.I[a] >= 0 && b.J[c] >= 0 && kj / 32 == m && exists(i, j);
                if (valid) {                             // an atom pair may appear twice when both atoms sit on both sides: first row wins
                    char& d = done[((size_t)m * A + std::min(i, j)) * A + std::max(i, j)];
                    if (d) valid = false; else d = 1;
                }
                const int type = valid ? pt[(size_t)i * A + j] : 0;
                out.rows[(size_t)bi * RB + 4 * a + c] = (valid ? 1u : 0u) | ((uint32_t)(ki / 32) << 1) | ((uint32_t)i << 4) |
                                                        ((uint32_t)(kj / 32) << 9) | ((uint32_t)j << 12) | ((uint32_t)type << 17);
                if (valid) {
                    ++n_valid;
                    out.pair_pos[((size_t)m * A + i) * A + j] = (bi * 2 + 0) * RB + 4 * a + c;       // direction A: I[a] -> J[c]
                    out.pair_pos[((size_t)m * A + j) * A + i] = (bi * 2 + 1) * RB + 4 * a + c;       // direction B: J[c] -> I[a]
                }
            }
        for (int side = 0; side < 2; ++side)               // J slots first (see first_touch above)
            for (int k = 0; k < 4; ++k) {
                const int key = side ? b.I[k] : b.J[k];
                if (key < 0) continue;
                const int32_t first = (first_touch && !touched[key]) ? PAIR_SLOT_FIRST_TOUCH : 0;
                touched[key] = 1;
                out.slotnode[(size_t)bi * RB + (side ? k : 4 + k)] = first | ((key / 32) << 8) | (key % 32);
            }
    }
    // partial-sum lists per atom: slot k of block bi (k < 4: the I slots = direction B's destinations; k >= 4: the J slots = direction
    // A's), in walk order -- the order the reduction adds them in
    std::vector<std::vector<int32_t>> lists((size_t)G * A);
    for (int bi = 0; bi < nblk; ++bi)
        for (int k = 0; k < 8; ++k) {
            const int key = k < 4 ? best[bi].I[k] : best[bi].J[k - 4];
            if (key >= 0) lists[(size_t)(key / 32) * A + key % 32].push_back(bi * 8 + k);
        }
    out.kmax = 1;
    for (auto& l : lists) out.kmax = std::max(out.kmax, (int)l.size());
    out.plist.assign((size_t)G * A * out.kmax, -1);
    for (size_t a = 0; a < lists.size(); ++a) std::copy(lists[a].begin(), lists[a].end(), out.plist.begin() + a * out.kmax);
    if (n_valid * 2 != (size_t)G * E) throw std::logic_error("pair template: not every edge was placed exactly once");
    out.fill = (double)n_valid / ((double)nblk * RB);
    return true;
}

}  // namespace ti
