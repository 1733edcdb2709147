// mfma_chain.hpp -- gfx950 building blocks shared by the drift kernels.
//
// Design: every dense layer of the drift networks is evaluated with the exact-f32 matrix
// instruction v_mfma_f32_32x32x2_f32.  A wave owns 32 "rows" (edges, atoms or particles) and keeps their
// activations in registers for the whole MLP chain:
//
//   * An activation set Act<NB> holds F = 32*NB features of 32 rows in NB*16 VGPRs per lane.  Lane l = (j, h) with
//     j = l & 31 (the row) and h = l >> 5; register (nb, i) holds feature  feat(nb,i,h) = 32*nb + 8*(i>>2) + 4*h + (i&3).
//     This is exactly the C/D accumulator layout of the 32x32 MFMA when the product is computed transposed,
//     D[n][row] = sum_k W[n][k] * X[row][k]  (weights as the A operand, activations as the B operand), so the output
//     of one layer is the B operand of the next with no data movement: k-step s uses register s of the input set.
//   * Because 4 consecutive registers hold 4 consecutive features, a lane loads/stores rows of row-major [row][F]
//     tensors and per-feature vectors (bias, gamma, beta) as float4 at offset 32*nb + 8*g + 4*h  (g = i>>2).
//   * Weights are pre-packed on the host (pack.cpp) into "chunks": 32 output features x F inputs, laid out
//     [k-step/4][lane][4] so that a wave reads one conflict-free ds_read_b128 per 4 MFMAs.  All waves of a workgroup
//     walk the same chunk stream, double-buffered through LDS (one barrier per chunk).
//   * The same chunk used with the operands swapped gives the flipped product D[row][n] (features on lanes, rows in
//     registers), used where a reduction over rows is needed.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ti {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NB>
struct Act {
    f32x16 b[NB];
};

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

// row of the 32x32 accumulator held in register i of lane-half h
__device__ __forceinline__ constexpr int acc_row(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }

// ---- cross-lane exchanges on the VALU (gfx950 v_permlane16_swap / v_permlane32_swap), never through the LDS crossbar.
//   swap16(a, b): the odd 16-lane rows of a <-> the even rows of b;   swap32(a, b): rows 2,3 of a <-> rows 0,1 of b.
// Why not __shfl_xor (ds_bpermute_b32): a ds_bpermute that is still outstanding when EXEC is narrowed returns wrong data on
// gfx950 -- lanes switched off after issue no longer contribute their value.  hipcc schedules exactly that
// (ds_bpermute; s_and_saveexec; ...; s_waitcnt lgkmcnt(0)) whenever a reduced value is first used inside a divergent `if`,
// and it goes wrong once another workgroup keeps the CU's LDS queue busy: the split-fp16 tangent readout kernel lost the
// (Vr . v) sum of whole 16-node tiles that way (140-170 of 256 molecules per evaluation at two waves per SIMD,
// 0 with the sums materialised before the branch, 0 with these swaps).  The swaps are in-order VALU instructions: no
// counter, no LDS round trip (~100 cycles each in every LayerNorm), same summation order as before.
// Inline asm, not __builtin_amdgcn_permlane{16,32}_swap: hipcc (ROCm 7.2) drops the second result of the builtin when both are
// bit-cast to float and used in arithmetic (it emits r0 + r0; seen in the .s).  `s_nop 1` = the two wait states between a VALU
// write of either operand and the swap (cdna_hip_programming.md T21); hipcc pads nothing inside asm.
__device__ __forceinline__ void lane_swap16(float& a, float& b) { asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void lane_swap32(float& a, float& b) { asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }

// XCD-aware workgroup order.  Workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share one, each XCD has its own
// 4 MiB L2; MI355X guide), so workgroups that read the SAME data -- the 54 seed directions of one primal molecule group in the
// tangent kernels -- should be neighbours on ONE XCD, not neighbours in blockIdx: logical index = (blocks of the XCDs before mine) +
// (my rank on my XCD).  A bijection of [0, n); purely a speed matter (no correctness depends on the placement).
__device__ __forceinline__ long long xcd_swizzle(unsigned b, unsigned n)
{
    const unsigned q = n >> 3, r = n & 7u, x = b & 7u, i = b >> 3;
    return (x < r ? (long long)x * (q + 1) : (long long)r * (q + 1) + (long long)(x - r) * q) + i;
}

// v + (the value of the lane 32 away), in every lane
// 1 / sqrt(x) for the LayerNorm scale: v_rsq_f32 (1 ulp) and one Newton step -- 5 instructions, within an ulp of the correctly rounded
// quotient; `1.0f / sqrtf(x)` compiles to ~25 (IEEE square root + IEEE division fix-ups) once per row and LayerNorm.
__device__ __forceinline__ float rsqrt_nr(float x)
{
    const float r = __builtin_amdgcn_rsqf(x);
    return r * __builtin_fmaf(-0.5f * x * r, r, 1.5f);
}
__device__ __forceinline__ float xhalf_sum(float v)
{
    float a = v, b = v;
    lane_swap32(a, b);              // a = [lo, lo], b = [hi, hi]
    return a + b;
}

// ------------------------------------------------------------------------------------------------ weight chunk pipe
// A "logical chunk" is 256*NB float4 (32 output features x F inputs).  SC logical chunks are staged per barrier interval
// (a superchunk): the gap between two MFMA bursts -- drain, ds_write, barrier, first ds_reads -- is paid once per SC chunks.
// T = threads per workgroup.  Usage: p = acquire(); ...GEMM on p...; release();   (uniform control flow only)
template <int NB, int T, int SC = 1>
struct Pipe {
    static constexpr int CH4 = 256 * NB;            // float4 per logical chunk
    static constexpr int SUP4 = CH4 * SC;           // float4 per superchunk
    static constexpr int PER = (SUP4 + T - 1) / T;
    const f32x4* __restrict__ g;
    f32x4* l[2];
    int nsup, idx, par, sub;
    f32x4 st[PER];

    __device__ __forceinline__ void init(const f32x4* stream, int n_chunks, f32x4* lds)
    {
        g = stream; nsup = n_chunks / SC; idx = 0; par = 0; sub = 0;
        l[0] = lds; l[1] = lds + SUP4;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int o = threadIdx.x + k * T;
            if (SUP4 % T == 0 || o < SUP4) l[0][o] = g[o];
        }
        __syncthreads();
    }
    // start of a superchunk: kick off the global load of the following one, return the LDS image of the current one
    __device__ __forceinline__ const f32x4* begin()
    {
        const int next = (idx + 1 == nsup) ? 0 : idx + 1;
        const f32x4* src = g + (size_t)next * SUP4;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int o = threadIdx.x + k * T;
            if (SUP4 % T == 0 || o < SUP4) st[k] = src[o];
        }
        return l[par];
    }
    __device__ __forceinline__ void end()
    {
        f32x4* dst = l[par ^ 1];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int o = threadIdx.x + k * T;
            if (SUP4 % T == 0 || o < SUP4) dst[o] = st[k];
        }
        __syncthreads();
        idx = (idx + 1 == nsup) ? 0 : idx + 1;
        par ^= 1;
    }
    // logical-chunk cursor on top of begin()/end()
    __device__ __forceinline__ const f32x4* acquire()
    {
        if (SC == 1) return begin();
        if (sub == 0) cur = begin();
        return cur + sub * CH4;
    }
    __device__ __forceinline__ void release()
    {
        if (SC == 1) { end(); return; }
        if (++sub == SC) { end(); sub = 0; }
    }
    const f32x4* cur;
};

// LDS-DMA variant (global_load_lds_dwordx4): the next superchunk goes straight from L2 into the other LDS buffer -- no
// staging registers, no ds_write.  The DMA is issued when the first logical chunk of the current superchunk is released
// (hipcc drains vmcnt(0) at the next use of an ordinary global load while a DMA is in flight, so it is kept away from
// the GEMM prologues), and is waited for at the superchunk's closing barrier.
// STAGGER (8-wave workgroups: waves w and w + 4 share a SIMD).  With one barrier per superchunk all eight waves run in lock step:
// both waves of a SIMD sit in their matrix phase together (the pipe is contended, the VALU idles) and then in their LayerNorm /
// product / reduction phase together (the VALU is contended, the matrix pipe idles) -- in-kernel stamps of the message kernel show a
// hidden layer taking 16 k cycles per pair of waves for 6 k cycles of matrix work and 9 k of vector work (profiles/r03c_*).  With
// STAGGER the first-dispatched half of the waves DEFERS the barrier that closes a superchunk to its next acquire(), i.e. to behind the
// vector phase that follows the products, while the second half keeps it right behind the products: between two barriers the early
// half runs [products k | vector phase k], the late half [vector phase k-1 | products k] -- the partners of a SIMD are half a phase
// apart, one on the matrix pipe while the other is on the VALU.  Every wave still executes exactly one barrier per superchunk, both
// halves read the SAME buffer between two barriers (two buffers suffice), and the prefetch of superchunk k+1 goes to the buffer
// every wave left before the previous barrier.
// NBUF buffers: the superchunk that is NBUF - 1 ahead is requested when the first chunk of the current one is released.  In-kernel
// stamps of the message kernels (profiles/r03c_*) showed every 4-chunk product phase lasting ~7.5 k cycles whatever its matrix work
// (3 k for the single products, 6 k for the lock-step pair): with two buffers the 64 KB a workgroup requests per superchunk have
// three chunk times to arrive, and at full occupancy an LDS-DMA of that size takes ~5 k cycles from issue to landed, so the interval
// between two barriers was the transfer's latency, not the products.  Closing superchunk k only needs superchunk k+1: the wait is
// `vmcnt((NBUF - 2) * PER)` -- the counter is in issue order, and the NBUF - 2 younger requests (and any younger loads / stores)
// may stay in flight.
template <int NB, int T, int SC, int CHUNK4 = 256 * NB, bool STAGGER = false, int NBUF = 2>      // CHUNK4: float4 per chunk (128 * NB for the hi-only chunks of the fp16 storage mode)
struct PipeDMA {
    static constexpr int CH4 = CHUNK4, SUP4 = CH4 * SC, PER = SUP4 / T;
    static_assert(SUP4 % T == 0, "superchunk must be a multiple of the workgroup's 16-byte lanes");
    static_assert(NBUF >= 2 && (NBUF - 2) * PER <= 48, "in-flight requests must fit the 6-bit vmcnt");
    const f32x4* __restrict__ g;
    f32x4* base;
    int nsup, idx, ahead, buf, sub, wave, lane;       // idx: superchunk being consumed, ahead: the next one to request, buf = idx % NBUF
    bool defer, pending;                    // STAGGER: this wave closes superchunks lazily / a close is outstanding

    __device__ __forceinline__ void dma(const f32x4* src, f32x4* dst) const
    {
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int o = k * T + wave * 64;                 // wave-uniform LDS base; lane i lands at base + 16*i bytes
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + o + lane),
                                             (__attribute__((address_space(3))) void*)(dst + o), 16, 0, 0);
        }
    }
    __device__ __forceinline__ void init(const f32x4* stream, int n_chunks, f32x4* lds, int wave_, int lane_)
    {
        g = stream; nsup = n_chunks / SC; idx = 0; buf = 0; sub = 0; wave = wave_; lane = lane_;
        base = lds;
        defer = STAGGER && wave_ < T / 128;               // the first half of the waves (SIMD partners are w and w + T/128)
        pending = false;
        ahead = 0;
#pragma unroll
        for (int b = 0; b < NBUF - 1; ++b) {               // superchunks 0 .. NBUF-2 (the stream is cyclic)
            dma(g + (size_t)ahead * SUP4, base + b * SUP4);
            ahead = (ahead + 1 == nsup) ? 0 : ahead + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    __device__ __forceinline__ void close()
    {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 2) * PER) : "memory");
        __syncthreads();
    }
    // Call once at the very end of a kernel: the last release() has a prefetch in flight that nobody will consume, and an
    // LDS-DMA still in flight when the workgroup retires lands in LDS that may already belong to the next workgroup.
    __device__ __forceinline__ void drain()
    {
        if (STAGGER && pending) { close(); pending = false; }          // every wave executes the same number of barriers
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __device__ __forceinline__ const f32x4* acquire()
    {
        if (STAGGER && pending) { close(); pending = false; }
        return base + buf * SUP4 + sub * CH4;
    }
    __device__ __forceinline__ void release()
    {
        if (sub == 0) {                                   // into the buffer of the superchunk every wave left before the last barrier
            dma(g + (size_t)ahead * SUP4, base + (buf == 0 ? NBUF - 1 : buf - 1) * SUP4);
            ahead = (ahead + 1 == nsup) ? 0 : ahead + 1;
        }
        if (++sub == SC) {
            if (STAGGER && defer) pending = true;
            else close();
            idx = (idx + 1 == nsup) ? 0 : idx + 1;
            buf = (buf + 1 == NBUF) ? 0 : buf + 1; sub = 0;
        }
    }
};

// The weight float4 of k-group s+1 is read from LDS before the 4 MFMAs of group s are issued, so the ds_read latency
// hides behind 256 cycles of matrix work instead of stalling every 4th MFMA.
// acc[n][row] += sum_k W[n][k] * in[row][k]      (transposed product; acc is one 32-feature output block)
template <int NB>
__device__ __forceinline__ void gemm_bt(f32x16& acc, const Act<NB>& in, const f32x4* wl, int lane)
{
    f32x4 w = wl[lane];
#pragma unroll
    for (int nbi = 0; nbi < NB; ++nbi)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int nxt = nbi * 4 + g + 1;
            const f32x4 wn = wl[(nxt < 4 * NB ? nxt : 0) * 64 + lane];
            acc = mfma32(w.x, in.b[nbi][4 * g + 0], acc);
            acc = mfma32(w.y, in.b[nbi][4 * g + 1], acc);
            acc = mfma32(w.z, in.b[nbi][4 * g + 2], acc);
            acc = mfma32(w.w, in.b[nbi][4 * g + 3], acc);
            w = wn;
        }
}

// acc[row][n] += sum_k in[row][k] * W[n][k]      (flipped: features n on lanes, rows in registers)
template <int NB>
__device__ __forceinline__ void gemm_fl(f32x16& acc, const Act<NB>& in, const f32x4* wl, int lane)
{
    f32x4 w = wl[lane];
#pragma unroll
    for (int nbi = 0; nbi < NB; ++nbi)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int nxt = nbi * 4 + g + 1;
            const f32x4 wn = wl[(nxt < 4 * NB ? nxt : 0) * 64 + lane];
            acc = mfma32(in.b[nbi][4 * g + 0], w.x, acc);
            acc = mfma32(in.b[nbi][4 * g + 1], w.y, acc);
            acc = mfma32(in.b[nbi][4 * g + 2], w.z, acc);
            acc = mfma32(in.b[nbi][4 * g + 3], w.w, acc);
            w = wn;
        }
}

// ------------------------------------------------------------------------------------------------ row <-> set moves
// one 32-feature block of a row-major row / per-feature vector: p points at feature 0
__device__ __forceinline__ f32x16 load_block(const float* __restrict__ p, int nb, int h)
{
    f32x16 r;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(p + 32 * nb + 8 * g + 4 * h);
        r[4 * g + 0] = t.x; r[4 * g + 1] = t.y; r[4 * g + 2] = t.z; r[4 * g + 3] = t.w;
    }
    return r;
}
__device__ __forceinline__ void store_block(float* __restrict__ p, int nb, int h, const f32x16& r)
{
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        f32x4 t = {r[4 * g + 0], r[4 * g + 1], r[4 * g + 2], r[4 * g + 3]};
        *reinterpret_cast<f32x4*>(p + 32 * nb + 8 * g + 4 * h) = t;
    }
}
// the same for state tensors that may be fp16 in HBM (H16: the fp16 storage mode); elem0 = element offset of feature 0 of the row
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
template <bool H16>
__device__ __forceinline__ f32x16 load_block_t(const float* base, size_t elem0, int nb, int h)
{
    if constexpr (!H16) return load_block(base + elem0, nb, h);
    f32x16 r;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const half4 t = *reinterpret_cast<const half4*>(reinterpret_cast<const _Float16*>(base) + elem0 + 32 * nb + 8 * g + 4 * h);
        r[4 * g + 0] = (float)t[0]; r[4 * g + 1] = (float)t[1]; r[4 * g + 2] = (float)t[2]; r[4 * g + 3] = (float)t[3];
    }
    return r;
}
template <bool H16>
__device__ __forceinline__ void store_block_t(float* base, size_t elem0, int nb, int h, const f32x16& r)
{
    if constexpr (!H16) { store_block(base + elem0, nb, h, r); return; }
#pragma unroll
    for (int g = 0; g < 4; ++g)
        *reinterpret_cast<half4*>(reinterpret_cast<_Float16*>(base) + elem0 + 32 * nb + 8 * g + 4 * h) =
            half4{(_Float16)r[4 * g + 0], (_Float16)r[4 * g + 1], (_Float16)r[4 * g + 2], (_Float16)r[4 * g + 3]};
}

template <int NB>
__device__ __forceinline__ void load_set(Act<NB>& a, const float* __restrict__ p, int h)
{
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) a.b[nb] = load_block(p, nb, h);
}
template <int NB>
__device__ __forceinline__ void store_set(float* __restrict__ p, int h, const Act<NB>& a)
{
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) store_block(p, nb, h, a.b[nb]);
}

// ------------------------------------------------------------------------------------------------ elementwise pieces
// x * sigmoid(x); v_exp_f32 / v_rcp_f32 are ~1 ulp, far below the 1e-5 parity bar
__device__ __forceinline__ float silu(float y) { return y * __builtin_amdgcn_rcpf(1.0f + __expf(-y)); }

// torch.nn.LayerNorm(F, eps=1e-5) + SiLU over the features of each row, in place; `a` already holds x W^T + b.
template <int NB>
__device__ __forceinline__ void ln_silu(Act<NB>& a, const float* __restrict__ gamma, const float* __restrict__ beta, int h)
{
    constexpr float invF = 1.0f / (32.0f * NB);
    float sum = 0.f;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int i = 0; i < 16; ++i) sum += a.b[nb][i];
    sum = xhalf_sum(sum);
    const float mean = sum * invF;
    float var = 0.f;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float d = a.b[nb][i] - mean;
            var = fmaf(d, d, var);
        }
    var = xhalf_sum(var);
    const float rstd = rsqrt_nr(var * invF + 1e-5f);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const f32x16 gm = load_block(gamma, nb, h);
        const f32x16 bt = load_block(beta, nb, h);
#pragma unroll
        for (int i = 0; i < 16; ++i) a.b[nb][i] = silu(fmaf((a.b[nb][i] - mean) * rstd, gm[i], bt[i]));
    }
}

// sin and cos of an fp32 angle, branch-free and accurate (~1 ulp) for any |a| < 2^30: the reduction a - n*pi/2 is done with
// two fp64 FMAs (53-bit pi/2 split), then the fdlibm k_sinf/k_cosf minimax polynomials on [-pi/4, pi/4] in fp32.  The ocml
// sincosf costs ~5x the instructions (Payne-Hanek path) and inlining it 16 times per row block blew the I-cache footprint.
__device__ __forceinline__ void sincos_cw(float a, float& s, float& c)
{
    const float n = rintf(a * 0.63661977236758134308f);
    const double nd = (double)n;
    const float r = (float)fma(-nd, 6.123233995736766036e-17, fma(-nd, 1.57079632679489655800e+00, (double)a));
    const float z = r * r;
    const float ps = fmaf(z, fmaf(z, fmaf(z, 2.7183114939898219064e-6f, -1.98393348360966317347e-4f), 8.3333293858894631756e-3f),
                          -1.66666666416265235595e-1f);
    const float pc = fmaf(z, fmaf(z, fmaf(z, 2.43904487962774090654e-5f, -1.38867637746099294692e-3f), 4.16666233237390631894e-2f),
                          -4.99999997251031003120e-1f);
    const float sr = fmaf(r * z, ps, r), cr = fmaf(z, pc, 1.0f);
    const int q = (int)n;
    const float ss = (q & 1) ? cr : sr, cc = (q & 1) ? sr : cr;
    s = (q & 2) ? -ss : ss;
    c = ((q + 1) & 2) ? -cc : cc;
}

// PositionalEncoder (/root/reference/mdqm9/thermo/ambient/models/embedding.py:127-160) of one scalar per row, written
// straight into the register layout: features 4m..4m+3 = cos(a(2m+1)), sin(a(2m+1)), cos(a(2m+2)), sin(a(2m+2)) with
// a(k) = ((x / max_length) * k) * pi evaluated left to right in fp32 like the reference (so the ARGUMENT is bit-identical).
template <int NB>
__device__ __forceinline__ void posenc_set(Act<NB>& a, float x_over_len, int h)
{
    constexpr float PI_F = 3.14159265358979323846f;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int m = 8 * nb + 2 * g + h;
            float s1, c1, s2, c2;
            sincos_cw((x_over_len * (float)(2 * m + 1)) * PI_F, s1, c1);
            sincos_cw((x_over_len * (float)(2 * m + 2)) * PI_F, s2, c2);
            a.b[nb][4 * g + 0] = c1; a.b[nb][4 * g + 1] = s1; a.b[nb][4 * g + 2] = c2; a.b[nb][4 * g + 3] = s2;
        }
}

// dot of a register set with a natural-order vector, summed over all F features of the row
template <int NB>
__device__ __forceinline__ float dot_set(const Act<NB>& a, const float* __restrict__ vec, int h)
{
    float acc = 0.f;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const f32x16 w = load_block(vec, nb, h);
#pragma unroll
        for (int i = 0; i < 16; ++i) acc = fmaf(a.b[nb][i], w[i], acc);
    }
    return xhalf_sum(acc);
}


// ================================================================================================================
// 16-row variant (namespace r16): the same chaining scheme on v_mfma_f32_16x16x4_f32 (same FLOP rate as 32x32x2).
// A wave owns 16 rows; an activation set of F = 16*NBK features is NBK*4 VGPRs per lane -- half the registers of the
// 32-row layout -- so several independent workgroups fit on a CU and hide each other's LayerNorm / wait phases.
//   lane l = (j, q), j = l & 15 (row), q = l >> 4;  register (nb, r) holds feature 16*nb + 4*q + r  (float4 at 16nb+4q).
//   accumulator of the flipped product: lane (n = l & 15, q) register r holds row 4*q + r.
// A weight chunk is still 32 output features x F inputs (two 16-feature blocks), packed [blk][nbi][lane][4] with
//   chunk[(blk*NBK + nbi)*64 + l][r] = W[row0 + 16*blk + (l&15)][col0 + 16*nbi + 4*(l>>4) + r]          (pack_chunk16).
namespace r16 {

template <int NBK>
struct Act {
    f32x4 b[NBK];
};

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// sum over the 4 lane-quarters that share a row (lanes j, j+16, j+32, j+48), fixed order
__device__ __forceinline__ float xquarters(float v)
{
    float a = v, b = v;
    lane_swap16(a, b);              // a = [v0, v0, v2, v2], b = [v1, v1, v3, v3]   (v_k = the value in lane row k)
    a += b; b = a;
    lane_swap32(a, b);              // a = [s01, s01, s01, s01], b = [s23, s23, s23, s23]
    return a + b;                   // (v0 + v1) + (v2 + v3) in every lane
}

// two 16-feature output blocks of one chunk; the two accumulator chains are independent and interleaved, which covers the
// 40-cycle dependent latency of the 32-cycle 16x16x4 MFMA
template <int NBK>
__device__ __forceinline__ void gemm_bt(f32x4& acc0, f32x4& acc1, const Act<NBK>& in, const f32x4* wl, int lane)
{
    f32x4 w0 = wl[lane], w1 = wl[NBK * 64 + lane];
#pragma unroll
    for (int nbi = 0; nbi < NBK; ++nbi) {
        const int nxt = nbi + 1 < NBK ? nbi + 1 : 0;
        const f32x4 n0 = wl[nxt * 64 + lane], n1 = wl[(NBK + nxt) * 64 + lane];
        acc0 = mfma16(w0.x, in.b[nbi].x, acc0); acc1 = mfma16(w1.x, in.b[nbi].x, acc1);
        acc0 = mfma16(w0.y, in.b[nbi].y, acc0); acc1 = mfma16(w1.y, in.b[nbi].y, acc1);
        acc0 = mfma16(w0.z, in.b[nbi].z, acc0); acc1 = mfma16(w1.z, in.b[nbi].z, acc1);
        acc0 = mfma16(w0.w, in.b[nbi].w, acc0); acc1 = mfma16(w1.w, in.b[nbi].w, acc1);
        w0 = n0; w1 = n1;
    }
}
template <int NBK>
__device__ __forceinline__ void gemm_fl(f32x4& acc0, f32x4& acc1, const Act<NBK>& in, const f32x4* wl, int lane)
{
    f32x4 w0 = wl[lane], w1 = wl[NBK * 64 + lane];
#pragma unroll
    for (int nbi = 0; nbi < NBK; ++nbi) {
        const int nxt = nbi + 1 < NBK ? nbi + 1 : 0;
        const f32x4 n0 = wl[nxt * 64 + lane], n1 = wl[(NBK + nxt) * 64 + lane];
        acc0 = mfma16(in.b[nbi].x, w0.x, acc0); acc1 = mfma16(in.b[nbi].x, w1.x, acc1);
        acc0 = mfma16(in.b[nbi].y, w0.y, acc0); acc1 = mfma16(in.b[nbi].y, w1.y, acc1);
        acc0 = mfma16(in.b[nbi].z, w0.z, acc0); acc1 = mfma16(in.b[nbi].z, w1.z, acc1);
        acc0 = mfma16(in.b[nbi].w, w0.w, acc0); acc1 = mfma16(in.b[nbi].w, w1.w, acc1);
        w0 = n0; w1 = n1;
    }
}

__device__ __forceinline__ f32x4 load_block(const float* p, int nb, int q) { return *reinterpret_cast<const f32x4*>(p + 16 * nb + 4 * q); }
__device__ __forceinline__ void store_block(float* p, int nb, int q, const f32x4& v) { *reinterpret_cast<f32x4*>(p + 16 * nb + 4 * q) = v; }
template <int NBK>
__device__ __forceinline__ void load_set(Act<NBK>& a, const float* p, int q)
{
#pragma unroll
    for (int nb = 0; nb < NBK; ++nb) a.b[nb] = load_block(p, nb, q);
}

template <int NBK>
__device__ __forceinline__ void ln_silu(Act<NBK>& a, const float* gamma, const float* beta, int q, float eps = 1e-5f)
{
    constexpr float invF = 1.0f / (16.0f * NBK);
    float sum = 0.f;
#pragma unroll
    for (int nb = 0; nb < NBK; ++nb) sum += (a.b[nb].x + a.b[nb].y) + (a.b[nb].z + a.b[nb].w);
    const float mean = xquarters(sum) * invF;
    float var = 0.f;
#pragma unroll
    for (int nb = 0; nb < NBK; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float d = a.b[nb][r] - mean;
            var = fmaf(d, d, var);
        }
    const float rstd = rsqrt_nr(xquarters(var) * invF + eps);
#pragma unroll
    for (int nb = 0; nb < NBK; ++nb) {
        const f32x4 gm = load_block(gamma, nb, q), bt = load_block(beta, nb, q);
#pragma unroll
        for (int r = 0; r < 4; ++r) a.b[nb][r] = silu(fmaf((a.b[nb][r] - mean) * rstd, gm[r], bt[r]));
    }
}

template <int NBK>
__device__ __forceinline__ void posenc_set(Act<NBK>& a, float x_over_len, int q)
{
    constexpr float PI_F = 3.14159265358979323846f;
#pragma unroll
    for (int nb = 0; nb < NBK; ++nb) {
        const int m = 4 * nb + q;
        float s1, c1, s2, c2;
        sincos_cw((x_over_len * (float)(2 * m + 1)) * PI_F, s1, c1);
        sincos_cw((x_over_len * (float)(2 * m + 2)) * PI_F, s2, c2);
        a.b[nb] = f32x4{c1, s1, c2, s2};
    }
}


// ---- forward-mode (dual number) twins: the primal result is bit-identical to ln_silu / posenc_set above
// LayerNorm + SiLU of a row and of its tangent:  n = (h - mean) rstd,  dn = rstd ((dh - mean(dh)) - n mean(n (dh - mean(dh)))),
// y = n g + b,  silu'(y) = sig(y) (1 + y (1 - sig(y))).
template <int NBK>
__device__ __forceinline__ void ln_silu_dual(Act<NBK>& a, Act<NBK>& da, const float* gamma, const float* beta, int q)
{
    constexpr float invF = 1.0f / (16.0f * NBK);
    float sum = 0.f, dsum = 0.f;
#pragma unroll
    for (int nb = 0; nb < NBK; ++nb) {
        sum += (a.b[nb].x + a.b[nb].y) + (a.b[nb].z + a.b[nb].w);
        dsum += (da.b[nb].x + da.b[nb].y) + (da.b[nb].z + da.b[nb].w);
    }
    const float mean = xquarters(sum) * invF, dmean = xquarters(dsum) * invF;
    float var = 0.f;
#pragma unroll
    for (int nb = 0; nb < NBK; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float d = a.b[nb][r] - mean;
            var = fmaf(d, d, var);
        }
    const float rstd = rsqrt_nr(xquarters(var) * invF + 1e-5f);
    float pr = 0.f;
#pragma unroll
    for (int nb = 0; nb < NBK; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) pr = fmaf((a.b[nb][r] - mean) * rstd, da.b[nb][r] - dmean, pr);
    const float proj = xquarters(pr) * invF;
#pragma unroll
    for (int nb = 0; nb < NBK; ++nb) {
        const f32x4 gm = load_block(gamma, nb, q), bt = load_block(beta, nb, q);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float n = (a.b[nb][r] - mean) * rstd;
            const float dn = rstd * ((da.b[nb][r] - dmean) - n * proj);
            const float y = fmaf(n, gm[r], bt[r]);
            const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-y));
            a.b[nb][r] = y * sg;
            da.b[nb][r] = (dn * gm[r]) * (sg * fmaf(y, 1.0f - sg, 1.0f));
        }
    }
}

// LayerNorm + SiLU that also returns what a tangent through it needs:  n = (h - mean) rstd  and  kk = rstd g silu'(y),
// so that later  da = kk ((dh - mean(dh)) - n mean(n (dh - mean(dh))))   (ln_tangent below).  `a` matches ln_silu bit for bit.
template <int NBK>
__device__ __forceinline__ void ln_silu_stats(Act<NBK>& a, Act<NBK>& n_out, Act<NBK>& k_out, const float* gamma, const float* beta, int q)
{
    constexpr float invF = 1.0f / (16.0f * NBK);
    float sum = 0.f;
#pragma unroll
    for (int nb = 0; nb < NBK; ++nb) sum += (a.b[nb].x + a.b[nb].y) + (a.b[nb].z + a.b[nb].w);
    const float mean = xquarters(sum) * invF;
    float var = 0.f;
#pragma unroll
    for (int nb = 0; nb < NBK; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float d = a.b[nb][r] - mean;
            var = fmaf(d, d, var);
        }
    const float rstd = rsqrt_nr(xquarters(var) * invF + 1e-5f);
#pragma unroll
    for (int nb = 0; nb < NBK; ++nb) {
        const f32x4 gm = load_block(gamma, nb, q), bt = load_block(beta, nb, q);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float n = (a.b[nb][r] - mean) * rstd;
            const float y = fmaf(n, gm[r], bt[r]);
            const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-y));
            a.b[nb][r] = y * sg;
            n_out.b[nb][r] = n;
            k_out.b[nb][r] = (rstd * gm[r]) * (sg * fmaf(y, 1.0f - sg, 1.0f));
        }
    }
}

// tangent through LayerNorm + SiLU from the stored (n, kk) of the primal row; in place on the tangent pre-activation
template <int NBK>
__device__ __forceinline__ void ln_tangent(Act<NBK>& da, const Act<NBK>& n, const Act<NBK>& kk)
{
    constexpr float invF = 1.0f / (16.0f * NBK);
    float dsum = 0.f;
#pragma unroll
    for (int nb = 0; nb < NBK; ++nb) dsum += (da.b[nb].x + da.b[nb].y) + (da.b[nb].z + da.b[nb].w);
    const float dmean = xquarters(dsum) * invF;
    float pr = 0.f;
#pragma unroll
    for (int nb = 0; nb < NBK; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) pr = fmaf(n.b[nb][r], da.b[nb][r] - dmean, pr);
    const float proj = xquarters(pr) * invF;
#pragma unroll
    for (int nb = 0; nb < NBK; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) da.b[nb][r] = kk.b[nb][r] * ((da.b[nb][r] - dmean) - n.b[nb][r] * proj);
}

// posenc_set and its derivative along dx_over_len:  d cos(a) = -sin(a) da,  d sin(a) = cos(a) da,  da = (dx_over_len k) pi
template <int NBK>
__device__ __forceinline__ void posenc_dual(Act<NBK>& a, Act<NBK>& da, float x_over_len, float dx_over_len, int q)
{
    constexpr float PI_F = 3.14159265358979323846f;
#pragma unroll
    for (int nb = 0; nb < NBK; ++nb) {
        const int m = 4 * nb + q;
        float s1, c1, s2, c2;
        sincos_cw((x_over_len * (float)(2 * m + 1)) * PI_F, s1, c1);
        sincos_cw((x_over_len * (float)(2 * m + 2)) * PI_F, s2, c2);
        const float d1 = (dx_over_len * (float)(2 * m + 1)) * PI_F, d2 = (dx_over_len * (float)(2 * m + 2)) * PI_F;
        a.b[nb] = f32x4{c1, s1, c2, s2};
        da.b[nb] = f32x4{-s1 * d1, c1 * d1, -s2 * d2, c2 * d2};
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Split-fp16 operands for the fp16 matrix rate (16x the f32 MFMA rate) at fp32-like accuracy.
//   x = xh + 2^-11 * xl,   xh = fp16(x),  xl = fp16((x - xh) * 2^11)          (|x| < 65504)
//   With round-to-nearest |x - xh| <= 2^-12 |x| and xl captures that residual to 2^-12 again, so the pair carries ~24
//   significand bits -- fp32's.  x*w = xh*wh + 2^-11 (xh*wl + xl*wh) + O(2^-24 |x w|): three products, fp32 accumulation;
//   the dropped lo*lo product is at fp32 rounding level (measured: 4 products change the drift error by nothing).
// on v_mfma_f32_16x16x32_f16.  One instruction covers a PAIR of 16-feature blocks (k = 32): lane (j, q) supplies the 8
// k-slots (q, i): i < 4 -> feature 16*(2m) + 4q + i, i >= 4 -> feature 16*(2m+1) + 4q + (i-4), i.e. exactly the 8 values
// it already holds in registers b[2m], b[2m+1] -- the chaining property of the fp32 layout is kept.  Weights are split
// on the host (pack_chunk16_split) with the same k-slot order; a chunk is still 16 KB at F = 128.
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x4 mfma16h(h8 a, h8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

// exact reciprocal of a power of two (the row scales of Opnd::set_scaled)
__device__ __forceinline__ float pow2_inverse(float s) { return __builtin_bit_cast(float, (254u << 23) - __builtin_bit_cast(unsigned, s)); }

// max over the 4 lane rows (quarters) that share an edge / node row, in every lane
__device__ __forceinline__ float xquarters_max(float v)
{
    float a = v, b = v;
    lane_swap16(a, b);
    a = fmaxf(a, b); b = a;
    lane_swap32(a, b);
    return fmaxf(a, b);
}

// The split of four values in 8 vector instructions: hi = fp16(v) is one v_cvt_pk_f16_f32 per pair, and each scaled residual
// fp16(2^11 (v - hi)) = fp16(fma(f32(hi), -2^11, 2^11 v)) is ONE v_fma_mix{lo,hi}_f16 that reads its half of the packed hi pair as it
// is and writes its half of the packed lo pair.  2^11 v and 2^11 hi are exact and v - hi is exactly representable, so the value that
// is rounded to fp16 is the one of "convert hi back, subtract, scale, pack" (12 instructions per four values) -- the same bits,
// fp16 subnormals included (tools/micro/split_probe.hip compares the two forms on 2^20 values).  hipcc forms the mix instructions
// for one pair in four and SLP-packs the rest into cvt / v_pk_fma_f32 / cvt_pk sequences, hence the inline asm.
// Hazard: v_fma_mixlo / mixhi write HALF a register; on gfx940/950 a vector instruction that reads such a register needs a wait
// state behind the write (hipcc inserts it for its own instructions, not inside inline asm).  A first version with mixlo / mixhi of
// one pair back to back lost low halves now and then (tangent taps off by 4e-5; the probe, with other instructions in between, saw
// nothing).  Here two pairs are interleaved and the block begins and ends with s_nop: at least two issue slots between every half
// write and the next read of that register, one in front of the first read of the packed hi pairs.  With it the tangent taps meet
// their 1e-5 bar with this form in every kernel (tests/test_gpu_divergence.py).  -DTI_SPLIT_REFERENCE=1 builds the 12-instruction
// form; its drift differs from this one's by 2.5e-7 (|b| ~ 0.1) although the probe finds the same bits for the same input: under
// hipcc's default -ffp-contract=fast the reference form's `v - hi` fuses with the multiplication that produced v (SiLU's y * rcp),
// i.e. it splits the unrounded product, while this form splits the rounded fp32 value it is given.
#ifndef TI_SPLIT_REFERENCE
#define TI_SPLIT_REFERENCE 0
#endif
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// v[0..3] -> packed fp16 pairs (hi01, hi23) and their scaled residuals (lo01, lo23), as dwords: the operand registers are assembled
// from whole dwords, never from single fp16 elements.  (A first form returned 4-element fp16 vectors and inserted them element by
// element into the 8-element operand registers: correct in every kernel that consumes the registers whole, but a test harness that
// read single elements back out of them got zeros and copies of element 0 from hipcc -- tools/micro/split_probe.hip stores whole
// registers for that reason.)
__device__ __forceinline__ void split_quad(const f32x4& v, unsigned& hi01, unsigned& hi23, unsigned& lo01, unsigned& lo23)
{
    hi01 = __builtin_bit_cast(unsigned, h2{(_Float16)v[0], (_Float16)v[1]});
    hi23 = __builtin_bit_cast(unsigned, h2{(_Float16)v[2], (_Float16)v[3]});
    if constexpr (TI_SPLIT_REFERENCE) {
        const h2 a = __builtin_bit_cast(h2, hi01), b = __builtin_bit_cast(h2, hi23);
        lo01 = __builtin_bit_cast(unsigned, h2{(_Float16)((v[0] - (float)a[0]) * 2048.0f), (_Float16)((v[1] - (float)a[1]) * 2048.0f)});
        lo23 = __builtin_bit_cast(unsigned, h2{(_Float16)((v[2] - (float)b[0]) * 2048.0f), (_Float16)((v[3] - (float)b[1]) * 2048.0f)});
    } else {
        const f32x4 s = v * 2048.0f;
        const float c = -2048.0f;
        asm("s_nop 0\n\t"
            "v_fma_mixlo_f16 %0, %2, %4, %5 op_sel_hi:[1,0,0]\n\t"
            "v_fma_mixlo_f16 %1, %3, %4, %7 op_sel_hi:[1,0,0]\n\t"
            "s_nop 0\n\t"
            "v_fma_mixhi_f16 %0, %2, %4, %6 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mixhi_f16 %1, %3, %4, %8 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
            "s_nop 1"
            : "=&v"(lo01), "=&v"(lo23)
            : "v"(hi01), "v"(hi23), "s"(c), "v"(s[0]), "v"(s[1]), "v"(s[2]), "v"(s[3]));
    }
}

template <int NBK, bool SPLIT>
struct Opnd {                                   // fp32 operand: the activation set itself
    Act<NBK> a;
    __device__ __forceinline__ void set(const Act<NBK>& x) { a = x; }
    __device__ __forceinline__ float set_scaled(const Act<NBK>& x) { a = x; return 1.0f; }
    __device__ __forceinline__ f32x4 value(int nb, float) const { return a.b[nb]; }          // the block the operand was set from (exact)
};
template <int NBK>
struct Opnd<NBK, true> {                        // split operand: hi and scaled-lo halves, NBK/2 k-steps
    h8 hi[NBK / 2], lo[NBK / 2];
    __device__ __forceinline__ void set(const Act<NBK>& x)
    {
#pragma unroll
        for (int m = 0; m < NBK / 2; ++m)
        {
            unsigned h0, h1, h2_, h3, l0, l1, l2, l3;     // element i of the k-step = half i: b[2m][0..3] | b[2m+1][0..3]
            split_quad(x.b[2 * m], h0, h1, l0, l1);
            split_quad(x.b[2 * m + 1], h2_, h3, l2, l3);
            hi[m] = __builtin_bit_cast(h8, u32x4{h0, h1, h2_, h3}); lo[m] = __builtin_bit_cast(h8, u32x4{l0, l1, l2, l3});
        }
    }
    // For operand sets that are NOT the output of a LayerNorm (the residual streams e, s, v, |Vv|: any magnitude fp32 holds):
    // the row (this lane's 16*NBK values and those of the 3 lanes that share its row) is divided by 2^k, k = exponent of the
    // row's largest |value|, before the split -- exact, and the halves then sit in [2^-24, 2) whatever the row's scale -- and
    // 2^k is returned; the caller multiplies the product of this operand by it (gemm_*_scaled), again exactly.  Without it a
    // value >= 65504 overflows the hi half (inf -> NaN downstream) and rows of magnitude < 1e-7 lose their lo half to fp16
    // subnormals (tests/golden/range_*.npz).
    __device__ __forceinline__ float set_scaled(const Act<NBK>& x)
    {
        float m = 0.f;
#pragma unroll
        for (int nb = 0; nb < NBK; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) m = fmaxf(m, fabsf(x.b[nb][r]));
        m = xquarters_max(m);
        const unsigned e = (__builtin_bit_cast(unsigned, m) >> 23) & 0xffu;            // biased exponent of the row maximum
        // rows below 2^-63 (incl. all-zero rows) are left unscaled -- their products vanish beside any bias -- and 2^k stops at
        // 2^63, so that scale, 1 / scale and (accumulator init) / scale stay far inside the fp32 range
        const unsigned ec = e < 64u ? 127u : e > 190u ? 190u : e;
        const float inv = __builtin_bit_cast(float, (254u - ec) << 23), scale = __builtin_bit_cast(float, ec << 23);
#pragma unroll
        for (int m2 = 0; m2 < NBK / 2; ++m2)
        {
            unsigned h0, h1, h2_, h3, l0, l1, l2, l3;
            split_quad(x.b[2 * m2] * inv, h0, h1, l0, l1);
            split_quad(x.b[2 * m2 + 1] * inv, h2_, h3, l2, l3);
            hi[m2] = __builtin_bit_cast(h8, u32x4{h0, h1, h2_, h3}); lo[m2] = __builtin_bit_cast(h8, u32x4{l0, l1, l2, l3});
        }
        return scale;
    }
    // Block nb of the set the operand was made from, rebuilt from its halves: (hi + 2^-11 lo) * scale, every step exact, i.e. the
    // original to ~23 bits (|error| <= 2^-23 |x| for values within 2^-13 of their row's maximum, <= 2^-36 of that maximum below).
    // Lets a kernel that already holds a row as an operand use it again as a value instead of reading it from HBM a second time.
    __device__ __forceinline__ f32x4 value(int nb, float scale) const
    {
        const int m = nb >> 1, o = 4 * (nb & 1);
        f32x4 r;
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = ((float)hi[m][o + i] + (float)lo[m][o + i] * 4.8828125e-4f) * scale;
        return r;
    }
};

// chunk image: [(blk*(NBK/2) + m)*2 + {0: hi, 1: lo}][lane] of 16-byte h8, i.e. 2*KS "steps" of two fragments each, contiguous.
// One step = 3 MFMAs (48 matrix cycles) on one (hi, lo) weight fragment pair.  The fragments of step s + TI_FRAG_AHEAD are read
// from LDS before the MFMAs of step s issue; the scheduling barriers pin that order (hipcc otherwise sinks each read to just
// in front of its own MFMAs and waits lgkmcnt(0) there: the wave then sits through a full LDS latency every 48 matrix
// cycles).  The barrier mask lets VALU / SALU / VMEM instructions cross, DS reads and MFMAs not.
#ifndef TI_FRAG_AHEAD
#define TI_FRAG_AHEAD 1
#endif
template <int NBK, bool FLIP>
__device__ __forceinline__ void gemm_split_chunk(f32x4& acc0, f32x4& acc1, const Opnd<NBK, true>& in, const h8* wl, int lane)
{
    constexpr int KS = NBK / 2, STEPS = 2 * KS, AH = TI_FRAG_AHEAD < STEPS ? TI_FRAG_AHEAD : STEPS;
    f32x4 x0 = {0, 0, 0, 0}, x1 = {0, 0, 0, 0};
    h8 fh[AH + 1], fl[AH + 1];                       // ring of fragment pairs, statically indexed after unrolling
#pragma unroll
    for (int s = 0; s < AH; ++s) { fh[s] = wl[(2 * s) * 64 + lane]; fl[s] = wl[(2 * s + 1) * 64 + lane]; }
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        if (s + AH < STEPS) {
            fh[(s + AH) % (AH + 1)] = wl[(2 * (s + AH)) * 64 + lane];
            fl[(s + AH) % (AH + 1)] = wl[(2 * (s + AH) + 1) * 64 + lane];
        }
        __builtin_amdgcn_sched_barrier(0x16);
        const h8 wh = fh[s % (AH + 1)], wlo = fl[s % (AH + 1)];
        const int m = s % KS;
        f32x4& acc = s < KS ? acc0 : acc1;
        f32x4& x = s < KS ? x0 : x1;
        if (FLIP) { acc = mfma16h(in.hi[m], wh, acc); x = mfma16h(in.lo[m], wh, x); x = mfma16h(in.hi[m], wlo, x); }
        else      { acc = mfma16h(wh, in.hi[m], acc); x = mfma16h(wh, in.lo[m], x); x = mfma16h(wlo, in.hi[m], x); }
        __builtin_amdgcn_sched_barrier(0x16);
    }
    acc0 += x0 * 4.8828125e-4f;                                          // 2^-11
    acc1 += x1 * 4.8828125e-4f;
}
template <int NBK>
__device__ __forceinline__ void gemm_bt(f32x4& acc0, f32x4& acc1, const Opnd<NBK, true>& in, const f32x4* wl4, int lane)
{
    gemm_split_chunk<NBK, false>(acc0, acc1, in, reinterpret_cast<const h8*>(wl4), lane);
}
template <int NBK>
__device__ __forceinline__ void gemm_fl(f32x4& acc0, f32x4& acc1, const Opnd<NBK, true>& in, const f32x4* wl4, int lane)
{
    gemm_split_chunk<NBK, true>(acc0, acc1, in, reinterpret_cast<const h8*>(wl4), lane);
}
// ---------------------------------------------------------------------------------------------------------------------
// One accumulator chain (message kernel, TI_PREC_F16X2).  The 2^11 on the residual halves of Opnd<NBK, true> is there to keep them out
// of the fp16 subnormal range, and it is why the cross terms need their own accumulator and a v_fma per output element to fold it
// back.  v_mfma_f32_16x16x32_f16 takes subnormal inputs at face value (tools/micro/mfma_denorm.hip), so here
//   * the WEIGHTS of each matrix are scaled on the host by a power of two S to the top of the fp16 range (max |S w| in [2^13, 2^14):
//     hi = fp16(S w), lo = fp16(S w - hi) is a normal number for every weight within 2^-16 of the matrix's largest), ti_api.hip;
//   * ACTIVATIONS are split unscaled: hi = fp16(x), lo = fp16(x - hi).  Their rows are LayerNorm / SiLU outputs, encodings or rows
//     normalised by set_scaled: a residual below 2^-14 is resolved to 2^-24, fp32 rounding level for such a row;
//   * hi.hi, hi.lo and lo.hi accumulate into ONE register set, which then holds S times the product: S cancels in the LayerNorm that
//     follows (epsilon scaled by S^2, biases by S) or is folded into factors the consumer multiplies with anyway.
template <int NBK>
struct Opnd1 {
    h8 hi[NBK / 2], lo[NBK / 2];
    __device__ __forceinline__ static void quad(const f32x4& v, unsigned& hi01, unsigned& hi23, unsigned& lo01, unsigned& lo23)
    {
        hi01 = __builtin_bit_cast(unsigned, h2{(_Float16)v[0], (_Float16)v[1]});
        hi23 = __builtin_bit_cast(unsigned, h2{(_Float16)v[2], (_Float16)v[3]});
        // fp16(fma(f32(hi), -1, v)) = fp16(v - hi): exact difference, one rounding; half-register writes fenced as in split_quad
        asm("s_nop 0\n\t"
            "v_fma_mixlo_f16 %0, %2, -1.0, %4 op_sel_hi:[1,0,0]\n\t"
            "v_fma_mixlo_f16 %1, %3, -1.0, %6 op_sel_hi:[1,0,0]\n\t"
            "s_nop 0\n\t"
            "v_fma_mixhi_f16 %0, %2, -1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mixhi_f16 %1, %3, -1.0, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
            "s_nop 1"
            : "=&v"(lo01), "=&v"(lo23)
            : "v"(hi01), "v"(hi23), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
    }
    __device__ __forceinline__ void set(const Act<NBK>& x)
    {
#pragma unroll
        for (int m = 0; m < NBK / 2; ++m) {
            unsigned h0, h1, h2_, h3, l0, l1, l2, l3;
            quad(x.b[2 * m], h0, h1, l0, l1);
            quad(x.b[2 * m + 1], h2_, h3, l2, l3);
            hi[m] = __builtin_bit_cast(h8, u32x4{h0, h1, h2_, h3}); lo[m] = __builtin_bit_cast(h8, u32x4{l0, l1, l2, l3});
        }
    }
    // un-normalised rows (e): divided by the power of two of the row maximum first (as Opnd<NBK, true>::set_scaled), which is returned
    __device__ __forceinline__ float set_scaled(const Act<NBK>& x)
    {
        float m = 0.f;
#pragma unroll
        for (int nb = 0; nb < NBK; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) m = fmaxf(m, fabsf(x.b[nb][r]));
        m = xquarters_max(m);
        const unsigned e = (__builtin_bit_cast(unsigned, m) >> 23) & 0xffu;
        const unsigned ec = e < 64u ? 127u : e > 190u ? 190u : e;
        const float inv = __builtin_bit_cast(float, (254u - ec) << 23), scale = __builtin_bit_cast(float, ec << 23);
#pragma unroll
        for (int m2 = 0; m2 < NBK / 2; ++m2) {
            unsigned h0, h1, h2_, h3, l0, l1, l2, l3;
            quad(x.b[2 * m2] * inv, h0, h1, l0, l1);
            quad(x.b[2 * m2 + 1] * inv, h2_, h3, l2, l3);
            hi[m2] = __builtin_bit_cast(h8, u32x4{h0, h1, h2_, h3}); lo[m2] = __builtin_bit_cast(h8, u32x4{l0, l1, l2, l3});
        }
        return scale;
    }
};
// chunk image as for gemm_split_chunk ([step][hi | lo][lane]); three products per step into the one accumulator of the step's output block
template <int NBK, bool FLIP>
__device__ __forceinline__ void gemm_split_chunk1(f32x4& acc0, f32x4& acc1, const Opnd1<NBK>& in, const h8* wl, int lane)
{
    constexpr int KS = NBK / 2, STEPS = 2 * KS, AH = TI_FRAG_AHEAD < STEPS ? TI_FRAG_AHEAD : STEPS;
    h8 fh[AH + 1], fl[AH + 1];
#pragma unroll
    for (int s = 0; s < AH; ++s) { fh[s] = wl[(2 * s) * 64 + lane]; fl[s] = wl[(2 * s + 1) * 64 + lane]; }
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        if (s + AH < STEPS) {
            fh[(s + AH) % (AH + 1)] = wl[(2 * (s + AH)) * 64 + lane];
            fl[(s + AH) % (AH + 1)] = wl[(2 * (s + AH) + 1) * 64 + lane];
        }
        __builtin_amdgcn_sched_barrier(0x16);
        const h8 wh = fh[s % (AH + 1)], wlo = fl[s % (AH + 1)];
        const int m = s % KS;
        f32x4& acc = s < KS ? acc0 : acc1;
        if (FLIP) { acc = mfma16h(in.hi[m], wh, acc); acc = mfma16h(in.lo[m], wh, acc); acc = mfma16h(in.hi[m], wlo, acc); }
        else      { acc = mfma16h(wh, in.hi[m], acc); acc = mfma16h(wh, in.lo[m], acc); acc = mfma16h(wlo, in.hi[m], acc); }
        __builtin_amdgcn_sched_barrier(0x16);
    }
}
template <int NBK>
__device__ __forceinline__ void gemm_bt(f32x4& acc0, f32x4& acc1, const Opnd1<NBK>& in, const f32x4* wl4, int lane) { gemm_split_chunk1<NBK, false>(acc0, acc1, in, reinterpret_cast<const h8*>(wl4), lane); }
template <int NBK>
__device__ __forceinline__ void gemm_fl(f32x4& acc0, f32x4& acc1, const Opnd1<NBK>& in, const f32x4* wl4, int lane) { gemm_split_chunk1<NBK, true>(acc0, acc1, in, reinterpret_cast<const h8*>(wl4), lane); }
// Two operand sets against ONE weight chunk (pair-major message kernel, painn_pair_kernel.hpp: the two directions of 16 atom pairs):
// every (hi, lo) fragment pair is read from LDS once and feeds six products, three per operand set -- half the LDS fragment
// reads, weight DMA and barriers per row of the one-set form, and two independent accumulator chains per output block.
template <int NBK, bool FLIP>
__device__ __forceinline__ void gemm_split_chunk1_x2(f32x4& a0, f32x4& a1, f32x4& b0, f32x4& b1, const Opnd1<NBK>& inA, const Opnd1<NBK>& inB,
                                                     const h8* wl, int lane)
{
    constexpr int KS = NBK / 2, STEPS = 2 * KS, AH = TI_FRAG_AHEAD < STEPS ? TI_FRAG_AHEAD : STEPS;
    h8 fh[AH + 1], fl[AH + 1];
#pragma unroll
    for (int s = 0; s < AH; ++s) { fh[s] = wl[(2 * s) * 64 + lane]; fl[s] = wl[(2 * s + 1) * 64 + lane]; }
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        if (s + AH < STEPS) {
            fh[(s + AH) % (AH + 1)] = wl[(2 * (s + AH)) * 64 + lane];
            fl[(s + AH) % (AH + 1)] = wl[(2 * (s + AH) + 1) * 64 + lane];
        }
        __builtin_amdgcn_sched_barrier(0x16);
        const h8 wh = fh[s % (AH + 1)], wlo = fl[s % (AH + 1)];
        const int m = s % KS;
        f32x4& accA = s < KS ? a0 : a1;
        f32x4& accB = s < KS ? b0 : b1;
        if (FLIP) {
            accA = mfma16h(inA.hi[m], wh, accA); accB = mfma16h(inB.hi[m], wh, accB);
            accA = mfma16h(inA.lo[m], wh, accA); accB = mfma16h(inB.lo[m], wh, accB);
            accA = mfma16h(inA.hi[m], wlo, accA); accB = mfma16h(inB.hi[m], wlo, accB);
        } else {
            accA = mfma16h(wh, inA.hi[m], accA); accB = mfma16h(wh, inB.hi[m], accB);
            accA = mfma16h(wh, inA.lo[m], accA); accB = mfma16h(wh, inB.lo[m], accB);
            accA = mfma16h(wlo, inA.hi[m], accA); accB = mfma16h(wlo, inB.hi[m], accB);
        }
        __builtin_amdgcn_sched_barrier(0x16);
    }
}
// ---------------------------------------------------------------------------------------------------------------------
// fp16-storage mode (include/ti_hip.h TI_PREC_F16; BASELINE.json configs[4] "fp16 node features with MFMA linears"): the state
// tensors s, v, P, e live in HBM as fp16, every matrix product is ONE v_mfma_f32_16x16x32_f16 per 32-wide k-step on the fp16
// rounding of its operands (weights: a hi-only image, half the bytes per chunk through LDS), accumulation,
// LayerNorm, SiLU, sin / cos and the per-atom sums stay fp32.  A separately labelled precision: its drift error against the
// reference is ~1e-3, not the 1e-5 of the other two paths (tests/test_gpu_parity.py reports it).
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

template <int NBK>
struct OpndH {                                  // fp16 operand: NBK/2 k-steps of 8 halves, same k-slot order as Opnd<NBK, true>
    h8 hi[NBK / 2];
    __device__ __forceinline__ void set(const Act<NBK>& x)
    {
#pragma unroll
        for (int m = 0; m < NBK / 2; ++m)
#pragma unroll
            for (int i = 0; i < 8; ++i) hi[m][i] = (_Float16)(i < 4 ? x.b[2 * m][i] : x.b[2 * m + 1][i - 4]);
    }
    __device__ __forceinline__ float set_scaled(const Act<NBK>& x) { set(x); return 1.0f; }
    // a row of a fp16 [row][F] tensor IS the operand: lane (j, q) owns features 16 nb + 4 q .. + 3 of its row
    __device__ __forceinline__ void load_row(const _Float16* row, int q)
    {
#pragma unroll
        for (int m = 0; m < NBK / 2; ++m) {
            const h4 a = *reinterpret_cast<const h4*>(row + 16 * (2 * m) + 4 * q), b = *reinterpret_cast<const h4*>(row + 16 * (2 * m + 1) + 4 * q);
            hi[m] = h8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        }
    }
};

template <int NBK, bool FLIP>
__device__ __forceinline__ void gemm_half_chunk(f32x4& acc0, f32x4& acc1, const OpndH<NBK>& in, const h8* wl, int lane)
{
    constexpr int KS = NBK / 2, STEPS = 2 * KS;
    h8 f[2];
    f[0] = wl[lane];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        if (s + 1 < STEPS) f[(s + 1) & 1] = wl[(s + 1) * 64 + lane];                // the chunk holds hi fragments only (8 KB at F = 128)
        __builtin_amdgcn_sched_barrier(0x16);
        const int m = s % KS;
        f32x4& acc = s < KS ? acc0 : acc1;
        if (FLIP) acc = mfma16h(in.hi[m], f[s & 1], acc); else acc = mfma16h(f[s & 1], in.hi[m], acc);
        __builtin_amdgcn_sched_barrier(0x16);
    }
}
template <int NBK>
__device__ __forceinline__ void gemm_bt(f32x4& acc0, f32x4& acc1, const OpndH<NBK>& in, const f32x4* wl4, int lane) { gemm_half_chunk<NBK, false>(acc0, acc1, in, reinterpret_cast<const h8*>(wl4), lane); }
template <int NBK>
__device__ __forceinline__ void gemm_fl(f32x4& acc0, f32x4& acc1, const OpndH<NBK>& in, const f32x4* wl4, int lane) { gemm_half_chunk<NBK, true>(acc0, acc1, in, reinterpret_cast<const h8*>(wl4), lane); }

// product of the pipe's current chunk with operand set `in` (FLIP: features on lanes).  The caller releases the chunk.
// The product runs at raised issue priority (s_setprio): the wave's matrix instructions go ahead of the SIMD partner's vector work --
// LayerNorm, operand splits, the output stage's sums -- which fills the slots in between instead of delaying them.  Measured on the
// pair-major message kernel: 28.13 -> 27.33 ms, levels 1, 2, 3 alike; directed layout 31.24 -> 31.03 (profiles/r03i_setprio_timing.txt).
// (Only here and in gemm_x2_on_pipe, i.e. in the message kernels: s_setprio is a scheduling boundary for hipcc, and placed in the
// pipe's acquire / release it cost the tangent edge kernel 680 spilled registers.)
template <bool FLIP, class OP, class PIPE>
__device__ __forceinline__ void gemm_on_pipe(f32x4& acc0, f32x4& acc1, const OP& in, PIPE& pipe, int lane)
{
    const f32x4* wl = pipe.acquire();
    __builtin_amdgcn_s_setprio(1);
    if constexpr (FLIP) gemm_fl(acc0, acc1, in, wl, lane);
    else gemm_bt(acc0, acc1, in, wl, lane);
    __builtin_amdgcn_s_setprio(0);
}
// An operand set as the 16-byte registers it is made of, parked in / fetched from HBM in register order ([register][lane]: every
// instruction moves 1 KB contiguous).  dst / src already point at this lane's slot.
template <typename OP>
__device__ __forceinline__ void opnd_store(const OP& o, f32x4* dst)
{
    constexpr int N = sizeof(OP) / 16;
    f32x4 r[N];
    __builtin_memcpy(r, &o, sizeof(OP));
#pragma unroll
    for (int k = 0; k < N; ++k) dst[k * 64] = r[k];
}
template <typename OP>
__device__ __forceinline__ void opnd_load(OP& o, const f32x4* src)
{
    constexpr int N = sizeof(OP) / 16;
    f32x4 r[N];
#pragma unroll
    for (int k = 0; k < N; ++k) r[k] = src[k * 64];
    __builtin_memcpy(&o, r, sizeof(OP));
}
// operand type of a matrix path: PREC 0 = f32 MFMA, 1 = split fp16 (hi + 2^-11 lo, 3 products), 2 = fp16 storage mode (1 product)
template <int NBK, int PREC> struct OpSel { using type = Opnd<NBK, PREC == 1>; };
template <int NBK> struct OpSel<NBK, 2> { using type = OpndH<NBK>; };

// state tensors: fp32, or fp16 in the storage mode.  One 16-feature block of a row: features 16 nb + 4 q .. + 3
template <bool H16>
__device__ __forceinline__ f32x4 load_state(const float* base, size_t row_elem0, int nb, int q)
{
    if constexpr (H16) {
        const h4 v = *reinterpret_cast<const h4*>(reinterpret_cast<const _Float16*>(base) + row_elem0 + 16 * nb + 4 * q);
        return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    } else {
        return *reinterpret_cast<const f32x4*>(base + row_elem0 + 16 * nb + 4 * q);
    }
}
template <bool H16>
__device__ __forceinline__ void store_state(float* base, size_t row_elem0, int nb, int q, const f32x4& v)
{
    if constexpr (H16) {
        *reinterpret_cast<h4*>(reinterpret_cast<_Float16*>(base) + row_elem0 + 16 * nb + 4 * q) = h4{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
    } else {
        *reinterpret_cast<f32x4*>(base + row_elem0 + 16 * nb + 4 * q) = v;
    }
}

// fp32 operands through the same interface
template <int NBK>
__device__ __forceinline__ void gemm_bt(f32x4& acc0, f32x4& acc1, const Opnd<NBK, false>& in, const f32x4* wl, int lane) { gemm_bt(acc0, acc1, in.a, wl, lane); }
template <int NBK>
__device__ __forceinline__ void gemm_fl(f32x4& acc0, f32x4& acc1, const Opnd<NBK, false>& in, const f32x4* wl, int lane) { gemm_fl(acc0, acc1, in.a, wl, lane); }

// ---- per-slot row sums of a flipped-layout block as a 16x16 selection product  S[slot][n] = sum_row Sel[slot][row] val[row][n].
// sel[r] (A operand: lane (slot, q) holds row 4q + r) is 1 where the row's destination is that slot; val (B operand) is the
// accumulator layout itself (lane (n, q) holds rows 4q + r).  Exact products, fixed summation order -> deterministic.
// f32: four 16x16x4 products (32 matrix cycles each).  Split mode: the values go through the fp16 pipe as hi + 2^-11 lo, the
// 0/1 selector is exact in fp16: two 16x16x16 products instead -- the selection sums were a quarter of the split kernel's
// matrix cycles.
template <bool SPLIT>
__device__ __forceinline__ f32x4 select_sum(const f32x4& sel, const f32x4& v)
{
    if (SPLIT) {
        h4 sh, hi, lo;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            sh[r] = (_Float16)sel[r];
            const _Float16 h = (_Float16)v[r];
            hi[r] = h;
            lo[r] = (_Float16)((v[r] - (float)h) * 2048.0f);
        }
        const f32x4 z = {0, 0, 0, 0};
        const f32x4 a = __builtin_amdgcn_mfma_f32_16x16x16f16(sh, hi, z, 0, 0, 0);
        const f32x4 b = __builtin_amdgcn_mfma_f32_16x16x16f16(sh, lo, z, 0, 0, 0);
        return a + b * 4.8828125e-4f;
    }
    f32x4 s = {0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < 4; ++r) s = mfma16(sel[r], v[r], s);
    return s;
}

// ---- per-slot row sums of flipped-layout blocks without the matrix core.  A block of 16 edge rows holds at most NS destination
// atoms ("slots" 0..NS-1; the template builder guarantees NS <= 4).  Lane (n, q) holds rows 4q + r of feature n in v[r].
//   partial  P_k = sum_r msk[k][r] v[r]                              (rows of slot k among this lane's four; msk is 0 / 1)
//   v_permlane16_swap(a, b): odd 16-lane rows of a <-> even rows of b;  a + b afterwards = [a0+a1, b0+b1, a2+a3, b2+b3]
//   v_permlane32_swap(a, b): rows 2,3 of a <-> rows 0,1 of b;           a + b afterwards = [a0+a2, a1+a3, b0+b2, b1+b3]
// so two exchange levels leave, in lane row (quarter) q, the full 16-row sum of ONE of four inputs: NS = 4: the four slots of
// one value; NS = 2: slots {0,1} of v0 in quarters {0,1} and of v1 in quarters {2,3}.  Fixed order -> deterministic; one atomic
// instruction then adds every slot of the block.  Replaces the 16x16 selection product (two fp16 MFMAs + a hi/lo conversion of
// the values per sum in split mode, four f32 MFMAs otherwise) and its four sparsely populated atomics per sum.
template <int NS>
struct QuarterSum {
    static_assert(NS == 2 || NS == 4, "2 or 4 slots per row block");
    float msk[NS][4];
    __device__ __forceinline__ static constexpr int slot_of_quarter(int q) { return NS == 2 ? (q & 1) : q; }
    __device__ __forceinline__ void set_row(int r, int slot, float one = 1.0f)     // `one`: a common factor of the summed values, applied here for free
    {
#pragma unroll
        for (int k = 0; k < NS; ++k) msk[k][r] = slot == k ? one : 0.0f;
    }
    __device__ __forceinline__ float partial(int k, const f32x4& v) const
    {
        return fmaf(msk[k][3], v[3], fmaf(msk[k][2], v[2], fmaf(msk[k][1], v[1], msk[k][0] * v[0])));
    }
    __device__ __forceinline__ static float swap16_add(float a, float b) { lane_swap16(a, b); return a + b; }
    __device__ __forceinline__ static float swap32_add(float a, float b) { lane_swap32(a, b); return a + b; }
    // NS == 4: quarter q of the result = sum over the block's rows of slot q
    __device__ __forceinline__ float sum(const f32x4& v) const
    {
        static_assert(NS == 4 || NS == 2, "");
        if (NS == 4) return swap32_add(swap16_add(partial(0, v), partial(1, v)), swap16_add(partial(2 % NS, v), partial(3 % NS, v)));
        const float x = swap16_add(partial(0, v), partial(1, v));       // NS == 2: slots {0,1} in quarters {0,1} and again in {2,3}
        return swap32_add(x, x);
    }
    // NS == 2: quarters {0,1} = slots {0,1} of v0, quarters {2,3} = slots {0,1} of v1
    __device__ __forceinline__ float sum_pair(const f32x4& v0, const f32x4& v1) const
    {
        return swap32_add(swap16_add(partial(0, v0), partial(1, v0)), swap16_add(partial(0, v1), partial(1, v1)));
    }
};

// ---- (value, tangent) operand pairs against one weight chunk: every LDS fragment is read once for both products
template <int NBK, bool FLIP>
__device__ __forceinline__ void gemm_pair_f32(f32x4& a0, f32x4& a1, f32x4& t0, f32x4& t1, const Act<NBK>& in, const Act<NBK>& tin,
                                              const f32x4* wl, int lane)
{
    f32x4 w0 = wl[lane], w1 = wl[NBK * 64 + lane];
#pragma unroll
    for (int nbi = 0; nbi < NBK; ++nbi) {
        const int nxt = nbi + 1 < NBK ? nbi + 1 : 0;
        const f32x4 n0 = wl[nxt * 64 + lane], n1 = wl[(NBK + nxt) * 64 + lane];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (FLIP) {
                a0 = mfma16(in.b[nbi][r], w0[r], a0); a1 = mfma16(in.b[nbi][r], w1[r], a1);
                t0 = mfma16(tin.b[nbi][r], w0[r], t0); t1 = mfma16(tin.b[nbi][r], w1[r], t1);
            } else {
                a0 = mfma16(w0[r], in.b[nbi][r], a0); a1 = mfma16(w1[r], in.b[nbi][r], a1);
                t0 = mfma16(w0[r], tin.b[nbi][r], t0); t1 = mfma16(w1[r], tin.b[nbi][r], t1);
            }
        }
        w0 = n0; w1 = n1;
    }
}
template <int NBK, bool FLIP>
__device__ __forceinline__ void gemm_pair_split_block(f32x4& acc, f32x4& tacc, const Opnd<NBK, true>& in, const Opnd<NBK, true>& tin,
                                                      const h8* wl, int lane)
{
    constexpr int KS = NBK / 2;
    f32x4 x = {0, 0, 0, 0}, tx = {0, 0, 0, 0};
    h8 wh = wl[lane], wlo = wl[64 + lane];
#pragma unroll
    for (int m = 0; m < KS; ++m) {
        const int nx = m + 1 < KS ? m + 1 : m;
        const h8 nh = wl[(nx * 2 + 0) * 64 + lane], nl = wl[(nx * 2 + 1) * 64 + lane];
        asm volatile("" ::: "memory");
        if (FLIP) {
            acc = mfma16h(in.hi[m], wh, acc); tacc = mfma16h(tin.hi[m], wh, tacc);
            x = mfma16h(in.lo[m], wh, x); tx = mfma16h(tin.lo[m], wh, tx);
            x = mfma16h(in.hi[m], wlo, x); tx = mfma16h(tin.hi[m], wlo, tx);
        } else {
            acc = mfma16h(wh, in.hi[m], acc); tacc = mfma16h(wh, tin.hi[m], tacc);
            x = mfma16h(wh, in.lo[m], x); tx = mfma16h(wh, tin.lo[m], tx);
            x = mfma16h(wlo, in.hi[m], x); tx = mfma16h(wlo, tin.hi[m], tx);
        }
        wh = nh; wlo = nl;
    }
    acc += x * 4.8828125e-4f;
    tacc += tx * 4.8828125e-4f;
}
template <int NBK>
__device__ __forceinline__ void gemm_bt2(f32x4& a0, f32x4& a1, f32x4& t0, f32x4& t1, const Opnd<NBK, false>& in, const Opnd<NBK, false>& tin,
                                         const f32x4* wl, int lane) { gemm_pair_f32<NBK, false>(a0, a1, t0, t1, in.a, tin.a, wl, lane); }
template <int NBK>
__device__ __forceinline__ void gemm_fl2(f32x4& a0, f32x4& a1, f32x4& t0, f32x4& t1, const Opnd<NBK, false>& in, const Opnd<NBK, false>& tin,
                                         const f32x4* wl, int lane) { gemm_pair_f32<NBK, true>(a0, a1, t0, t1, in.a, tin.a, wl, lane); }
template <int NBK>
__device__ __forceinline__ void gemm_bt2(f32x4& a0, f32x4& a1, f32x4& t0, f32x4& t1, const Opnd<NBK, true>& in, const Opnd<NBK, true>& tin,
                                         const f32x4* wl4, int lane)
{
    const h8* wl = reinterpret_cast<const h8*>(wl4);
    gemm_pair_split_block<NBK, false>(a0, t0, in, tin, wl, lane);
    gemm_pair_split_block<NBK, false>(a1, t1, in, tin, wl + NBK * 64, lane);
}
template <int NBK>
__device__ __forceinline__ void gemm_fl2(f32x4& a0, f32x4& a1, f32x4& t0, f32x4& t1, const Opnd<NBK, true>& in, const Opnd<NBK, true>& tin,
                                         const f32x4* wl4, int lane)
{
    const h8* wl = reinterpret_cast<const h8*>(wl4);
    gemm_pair_split_block<NBK, true>(a0, t0, in, tin, wl, lane);
    gemm_pair_split_block<NBK, true>(a1, t1, in, tin, wl + NBK * 64, lane);
}

// ---- two operand sets (the two directions of a pair block) against the pipe's current chunk, every matrix path
template <int NBK, bool FLIP>
__device__ __forceinline__ void gemm_half_chunk_x2(f32x4& a0, f32x4& a1, f32x4& b0, f32x4& b1, const OpndH<NBK>& inA, const OpndH<NBK>& inB,
                                                   const h8* wl, int lane)
{
    constexpr int KS = NBK / 2, STEPS = 2 * KS;
    h8 f[2];
    f[0] = wl[lane];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        if (s + 1 < STEPS) f[(s + 1) & 1] = wl[(s + 1) * 64 + lane];
        __builtin_amdgcn_sched_barrier(0x16);
        const int m = s % KS;
        f32x4& accA = s < KS ? a0 : a1;
        f32x4& accB = s < KS ? b0 : b1;
        if (FLIP) { accA = mfma16h(inA.hi[m], f[s & 1], accA); accB = mfma16h(inB.hi[m], f[s & 1], accB); }
        else      { accA = mfma16h(f[s & 1], inA.hi[m], accA); accB = mfma16h(f[s & 1], inB.hi[m], accB); }
        __builtin_amdgcn_sched_barrier(0x16);
    }
}
template <bool FLIP, int NBK>
__device__ __forceinline__ void gemm_x2(f32x4& a0, f32x4& a1, f32x4& b0, f32x4& b1, const Opnd1<NBK>& inA, const Opnd1<NBK>& inB, const f32x4* wl, int lane)
{
    gemm_split_chunk1_x2<NBK, FLIP>(a0, a1, b0, b1, inA, inB, reinterpret_cast<const h8*>(wl), lane);
}
template <bool FLIP, int NBK>
__device__ __forceinline__ void gemm_x2(f32x4& a0, f32x4& a1, f32x4& b0, f32x4& b1, const Opnd<NBK, false>& inA, const Opnd<NBK, false>& inB, const f32x4* wl, int lane)
{
    gemm_pair_f32<NBK, FLIP>(a0, a1, b0, b1, inA.a, inB.a, wl, lane);
}
template <bool FLIP, int NBK>
__device__ __forceinline__ void gemm_x2(f32x4& a0, f32x4& a1, f32x4& b0, f32x4& b1, const Opnd<NBK, true>& inA, const Opnd<NBK, true>& inB, const f32x4* wl4, int lane)
{
    const h8* wl = reinterpret_cast<const h8*>(wl4);
    gemm_pair_split_block<NBK, FLIP>(a0, b0, inA, inB, wl, lane);
    gemm_pair_split_block<NBK, FLIP>(a1, b1, inA, inB, wl + NBK * 64, lane);
}
template <bool FLIP, int NBK>
__device__ __forceinline__ void gemm_x2(f32x4& a0, f32x4& a1, f32x4& b0, f32x4& b1, const OpndH<NBK>& inA, const OpndH<NBK>& inB, const f32x4* wl, int lane)
{
    gemm_half_chunk_x2<NBK, FLIP>(a0, a1, b0, b1, inA, inB, reinterpret_cast<const h8*>(wl), lane);
}
template <bool FLIP, class OP, class PIPE>
__device__ __forceinline__ void gemm_x2_on_pipe(f32x4& a0, f32x4& a1, f32x4& b0, f32x4& b1, const OP& inA, const OP& inB, PIPE& pipe, int lane)
{
    const f32x4* wl = pipe.acquire();
    __builtin_amdgcn_s_setprio(1);                  // see gemm_on_pipe
    gemm_x2<FLIP>(a0, a1, b0, b1, inA, inB, wl, lane);
    __builtin_amdgcn_s_setprio(0);
}

}  // namespace r16

}  // namespace ti
