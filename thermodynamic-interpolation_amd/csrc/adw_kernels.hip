// adw_kernels.hip -- FCNetMultiBeta drift (adw double well) on gfx950, plus the elementwise integrator kernels.
//
// Restates /root/reference/adw/thermo/models/simple.py:22-41.  Both MLPs of the model have the same shape
//   Linear(3 -> H), SiLU, [Linear(H -> H), SiLU] x n_hidden, Linear(H -> 1)
// (beta_embed: inputs [beta0, beta1, t], n_hidden = 1;  net: inputs [x, t, beta_embed], n_hidden = num_layers-1),
// so one kernel serves both: hidden activations stay in registers, H x H layers on the matrix cores.
#include "mfma_chain.hpp"
#include "ti_internal.hpp"

namespace ti {

// Per-MLP vector block in LDS (floats): w_in [H][3] | b_in [H] | b_hidden [n_hidden][H] | w_out [H]
//
// 16 rows per wave on the r16 primitives (mfma_chain.hpp), f32 or split-fp16 matrix path.  TAN additionally propagates the
// tangent d/d(a0) through the network (forward mode): out_div = d out / d a0, which for `net` is the divergence of the 1-D
// drift, ODEWrapper.compute_divergence (/root/reference/adw/thermo/models/ode_wrapper.py:55-67) without its 1e-2 factor.
template <int NBK, bool SPLIT, bool TAN>
__global__ __launch_bounds__(256, (NBK <= 8 && !TAN) ? 2 : 1) void adw_mlp_kernel(const AdwParams p)
{
    constexpr int H = 16 * NBK, NB = (H + 31) / 32, WAVES = 4, T = 64 * WAVES, CH4 = 256 * NB;
    using A16 = r16::Act<NBK>;
    using OP = r16::Opnd<NBK, SPLIT>;
    extern __shared__ f32x4 lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), j = lane & 15, q = lane >> 4;
    float* vec = reinterpret_cast<float*>(lds + 2 * CH4);
    const int nvec4 = (5 + p.n_hidden) * H / 4;
    for (int i = threadIdx.x; i < nvec4; i += T) reinterpret_cast<f32x4*>(vec)[i] = reinterpret_cast<const f32x4*>(p.vecs)[i];
    PipeDMA<NB, T, 1> pipe;
    if (p.nch > 0) pipe.init(reinterpret_cast<const f32x4*>(p.stream), p.nch, lds, wave, lane);
    else __syncthreads();
    const float* w_in = vec;
    const float* b_in = vec + 3 * H;
    const float* b_hid = vec + 4 * H;
    const float* w_out = vec + (4 + p.n_hidden) * H;

    const long long row = ((long long)blockIdx.x * WAVES + wave) * 16 + j;
    const bool ok = row < p.B;
    const long long r = ok ? row : p.B - 1;
    const float a0 = p.x[r];
    const float a1 = p.in1 ? p.in1[r] : p.t;
    const float a2 = p.idx ? p.emb[p.idx[r]] : (p.emb ? p.emb[r] : p.t);

    // silu(z) = z * sig(z);  silu'(z) = sig(z) + silu(z) * (1 - sig(z))
    auto act = [](float z, float& y, float& dy) {
        const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-z));
        y = z * sg;
        dy = fmaf(y, 1.0f - sg, sg);
    };
    // input layer (K = 3): plain FMAs straight into the register layout
    A16 cur, tan;
#pragma unroll
    for (int nb = 0; nb < NBK; ++nb) {
        const float* w = w_in + (16 * nb + 4 * q) * 3;                     // rows f..f+3 of W_in[H][3]
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(w), w1 = *reinterpret_cast<const f32x4*>(w + 4),
                    w2 = *reinterpret_cast<const f32x4*>(w + 8);
        const float ww[12] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w};
        const f32x4 bb = r16::load_block(b_in, nb, q);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float z = fmaf(ww[3 * k + 2], a2, fmaf(ww[3 * k + 1], a1, fmaf(ww[3 * k], a0, bb[k])));
            float y, dy;
            act(z, y, dy);
            cur.b[nb][k] = y;
            if (TAN) tan.b[nb][k] = dy * ww[3 * k];
        }
    }
    // hidden layers
    for (int l = 0; l < p.n_hidden; ++l) {
        OP in, tin;
        in.set(cur);
        if (TAN) tin.set(tan);
        const float* bias = b_hid + (size_t)l * H;
#pragma unroll
        for (int ch = 0; ch < NB; ++ch) {
            const f32x4* wl = pipe.acquire();
            f32x4 z0 = r16::load_block(bias, 2 * ch, q), z1 = r16::load_block(bias, 2 * ch + 1, q);
            r16::gemm_bt(z0, z1, in, wl, lane);
            f32x4 t0 = {0, 0, 0, 0}, t1 = {0, 0, 0, 0};
            if (TAN) r16::gemm_bt(t0, t1, tin, wl, lane);
            pipe.release();
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float y, dy;
                act(z0[k], y, dy); cur.b[2 * ch][k] = y;     if (TAN) tan.b[2 * ch][k] = dy * t0[k];
                act(z1[k], y, dy); cur.b[2 * ch + 1][k] = y; if (TAN) tan.b[2 * ch + 1][k] = dy * t1[k];
            }
        }
    }
    float o = 0.f, d = 0.f;
#pragma unroll
    for (int nb = 0; nb < NBK; ++nb) {
        const f32x4 w = r16::load_block(w_out, nb, q);
#pragma unroll
        for (int k = 0; k < 4; ++k) { o = fmaf(cur.b[nb][k], w[k], o); if (TAN) d = fmaf(tan.b[nb][k], w[k], d); }
    }
    o = r16::xquarters(o) + p.b_out;
    if (TAN) d = r16::xquarters(d);
    if (ok && q == 0) { p.out[row] = o; if (TAN) p.out_div[row] = d; }
    pipe.drain();
}

static size_t adw_lds_bytes(int NB, int n_hidden) { return 2 * (size_t)256 * NB * 16 + (size_t)(5 + n_hidden) * 32 * NB * 4; }

#define TI_DISPATCH_NB(NBv, ...) \
    switch (NBv) {                                                            \
        case 1: { constexpr int NB = 1; __VA_ARGS__; } break;                 \
        case 2: { constexpr int NB = 2; __VA_ARGS__; } break;                 \
        case 4: { constexpr int NB = 4; __VA_ARGS__; } break;                 \
        case 8: { constexpr int NB = 8; __VA_ARGS__; } break;                 \
        default: return hipErrorInvalidValue;                                 \
    }

template <int NBK>
static hipError_t adw_set_attrs(size_t bytes)
{
    hipError_t e;
#define TI_SET(k) if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes)) != hipSuccess) return e
    TI_SET((adw_mlp_kernel<NBK, false, false>)); TI_SET((adw_mlp_kernel<NBK, true, false>));
    TI_SET((adw_mlp_kernel<NBK, false, true>)); TI_SET((adw_mlp_kernel<NBK, true, true>));
#undef TI_SET
    return hipSuccess;
}

hipError_t configure_adw_kernels(int NBv, int max_hidden)
{
    TI_DISPATCH_NB(NBv, return (adw_set_attrs<2 * NB>(adw_lds_bytes(NB, max_hidden))));
    return hipSuccess;
}

hipError_t launch_adw(int NBv, bool split, const AdwParams& p, hipStream_t st)
{
    TI_DISPATCH_NB(NBv, {
        const dim3 g((unsigned)((p.B + 63) / 64));
        const size_t l = adw_lds_bytes(NB, p.n_hidden);
        const bool tanv = p.out_div != nullptr;
        if (split) {
            if (tanv) hipLaunchKernelGGL((adw_mlp_kernel<2 * NB, true, true>), g, dim3(256), l, st, p);
            else hipLaunchKernelGGL((adw_mlp_kernel<2 * NB, true, false>), g, dim3(256), l, st, p);
        } else {
            if (tanv) hipLaunchKernelGGL((adw_mlp_kernel<2 * NB, false, true>), g, dim3(256), l, st, p);
            else hipLaunchKernelGGL((adw_mlp_kernel<2 * NB, false, false>), g, dim3(256), l, st, p);
        }
    });
    return hipGetLastError();
}

// ================================================================================================== integrator
// Unfused multiply/add on purpose: the reference state update is `x + dt * b` in two roundings.
__global__ void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, float a, const float* __restrict__ b, long long n)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = __fadd_rn(x[i], __fmul_rn(a, b[i]));
}
__global__ void heun_kernel(float* __restrict__ x, float hdt, const float* __restrict__ b1, const float* __restrict__ b2, long long n)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = __fadd_rn(x[i], __fmul_rn(hdt, __fadd_rn(b1[i], b2[i])));
}

// Philox4x32-10, same function as oracle/ti_oracle.c:ti_normal (build-defined, include/ti_hip.h TI_SCHEME_EM)
__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
__device__ __forceinline__ float ti_normal(uint64_t seed, long long traj, int step, int comp)
{
    uint32_t c[4] = {(uint32_t)traj, (uint32_t)((uint64_t)traj >> 32), (uint32_t)step, (uint32_t)(comp >> 2)};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    const int pair = (comp & 3) >> 1;
    const float u1 = ((float)(c[2 * pair] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float u2 = ((float)(c[2 * pair + 1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float r = sqrtf(-2.0f * logf(u1)), a = 6.283185307179586f * u2;
    return (comp & 1) ? r * sinf(a) : r * cosf(a);
}

// one thread per trajectory: x[traj][c] += sigma * (xi_c - COM_c)
__global__ void noise_kernel(float* __restrict__ x, float sigma, uint64_t seed, long long traj0, int step, long long B, int comps,
                             int atoms_for_com)
{
    const long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= B) return;
    float com[3] = {0.f, 0.f, 0.f};
    if (atoms_for_com > 0) {
        for (int c = 0; c < comps; ++c) com[c % 3] += ti_normal(seed, traj0 + m, step, c);
        for (int k = 0; k < 3; ++k) com[k] = com[k] / (float)atoms_for_com;
    }
    for (int c = 0; c < comps; ++c) {
        float z = ti_normal(seed, traj0 + m, step, c);
        if (atoms_for_com > 0) z -= com[c % 3];
        x[m * comps + c] = __fadd_rn(x[m * comps + c], __fmul_rn(sigma, z));
    }
}

__global__ void scale_kernel(float* __restrict__ y, const float* __restrict__ x, float a, long long n)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = x[i] * a;
}

__global__ void nan_check_kernel(const float* __restrict__ x, long long n, int* flag)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && !isfinite(x[i])) *flag = 1;
}

// MFMA lane-map self-test: D = A * B with A[i][k] = 1 + i + 100k, B[k][j] = 1000 + j - 7k; also returns the
// accumulator-layout ids so the host can check (reg, lane) -> (row, col) independently.
__global__ void selftest_kernel(float* out)
{
    const int l = threadIdx.x;
    const float a = 1.0f + (float)(l & 31) + 100.0f * (float)(l >> 5);
    const float b = 1000.0f + (float)(l & 31) - 7.0f * (float)(l >> 5);
    f32x16 acc = {0};
    acc = mfma32(a, b, acc);
    for (int i = 0; i < 16; ++i) out[l * 16 + i] = acc[i];
}

// Operand-split self-test: Opnd<8, true>::set -- the 8-instruction form with its half-register writes (mfma_chain.hpp: split_quad),
// eight quads back to back as in the kernels -- against the plain arithmetic, bit for bit, on values spread over 45 binades (exact
// fp16 values, zeros and fp16-subnormal residuals included).  out[0] counts the halves that differ.
__global__ void split_selftest_kernel(unsigned* out)
{
    const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
    r16::Act<8> x;
#pragma unroll
    for (int nb = 0; nb < 8; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            unsigned h = (gid * 32u + nb * 4u + r) * 2654435761u;
            h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
            const int e = (int)(h % 45u) - 30;                                   // 2^-30 .. 2^14
            const unsigned keep = (h >> 8) % 5u == 0 ? 0x7fe000u : (h >> 8) % 7u == 0 ? 0u : 0x7fffffu;   // some exact fp16 values, some powers of two
            float v = __builtin_bit_cast(float, (unsigned)((e + 127) << 23) | ((h >> 9) & keep));
            if ((h >> 3) % 11u == 0) v = 0.f;
            x.b[nb][r] = (h & 1u) ? -v : v;
        }
    r16::Opnd<8, true> o;
    o.set(x);
    unsigned bad = 0;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const r16::u32x4 hw = __builtin_bit_cast(r16::u32x4, o.hi[m]), lw = __builtin_bit_cast(r16::u32x4, o.lo[m]);     // registers as dwords
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float v0 = x.b[2 * m + (k >> 1)][2 * (k & 1)], v1 = x.b[2 * m + (k >> 1)][2 * (k & 1) + 1];
            const _Float16 h0 = (_Float16)v0, h1 = (_Float16)v1;
            const r16::h2 hr{h0, h1}, lr{(_Float16)((v0 - (float)h0) * 2048.0f), (_Float16)((v1 - (float)h1) * 2048.0f)};
            const unsigned dh = hw[k] ^ __builtin_bit_cast(unsigned, hr), dl = lw[k] ^ __builtin_bit_cast(unsigned, lr);
            bad += ((dh & 0xffffu) != 0) + ((dh >> 16) != 0) + ((dl & 0xffffu) != 0) + ((dl >> 16) != 0);
        }
    }
    // the one-accumulator format's split of the same values (Opnd1::quad: unscaled residual, v_fma_mix with the literal -1.0): its own
    // hand-written sequence with the same half-register-write hazard, so its own bit-for-bit check
    r16::Opnd1<8> o1;
    o1.set(x);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const r16::u32x4 hw = __builtin_bit_cast(r16::u32x4, o1.hi[m]), lw = __builtin_bit_cast(r16::u32x4, o1.lo[m]);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float v0 = x.b[2 * m + (k >> 1)][2 * (k & 1)], v1 = x.b[2 * m + (k >> 1)][2 * (k & 1) + 1];
            const _Float16 h0 = (_Float16)v0, h1 = (_Float16)v1;
            const r16::h2 hr{h0, h1}, lr{(_Float16)(v0 - (float)h0), (_Float16)(v1 - (float)h1)};      // fp16-subnormal residuals included
            const unsigned dh = hw[k] ^ __builtin_bit_cast(unsigned, hr), dl = lw[k] ^ __builtin_bit_cast(unsigned, lr);
            bad += ((dh & 0xffffu) != 0) + ((dh >> 16) != 0) + ((dl & 0xffffu) != 0) + ((dl >> 16) != 0);
        }
    }
    if (bad) atomicAdd(out, bad);
    // The one-accumulator format relies on v_mfma_f32_16x16x32_f16 taking fp16 SUBNORMAL inputs at face value (tools/micro/mfma_denorm.hip).
    // One wave checks it: A = 2^-24 (the smallest subnormal) everywhere, B = 1 -> every output element must be 32 * 2^-24 = 2^-19 exactly.
    if (gid < 64) {
        r16::h8 a, b;
#pragma unroll
        for (int i = 0; i < 8; ++i) { a[i] = __builtin_bit_cast(_Float16, (unsigned short)1); b[i] = (_Float16)1.0f; }
        const f32x4 d = r16::mfma16h(a, b, f32x4{0, 0, 0, 0});
        unsigned wrong = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) wrong += d[r] != 1.9073486328125e-06f;
        if (wrong) atomicAdd(out + 1, wrong);
    }
}

static inline dim3 grid1(long long n, int bs) { return dim3((unsigned)((n + bs - 1) / bs)); }

hipError_t launch_axpy(float* y, const float* x, float a, const float* b, long long n, hipStream_t st)
{
    if (n > 0) hipLaunchKernelGGL(axpy_kernel, grid1(n, 256), dim3(256), 0, st, y, x, a, b, n);
    return hipGetLastError();
}
hipError_t launch_heun(float* x, float hdt, const float* b1, const float* b2, long long n, hipStream_t st)
{
    if (n > 0) hipLaunchKernelGGL(heun_kernel, grid1(n, 256), dim3(256), 0, st, x, hdt, b1, b2, n);
    return hipGetLastError();
}
hipError_t launch_noise(float* x, float sigma, uint64_t seed, long long traj0, int step, long long B, int comps, int atoms_for_com,
                        hipStream_t st)
{
    if (B > 0) hipLaunchKernelGGL(noise_kernel, grid1(B, 128), dim3(128), 0, st, x, sigma, seed, traj0, step, B, comps, atoms_for_com);
    return hipGetLastError();
}
hipError_t launch_scale(float* y, const float* x, float a, long long n, hipStream_t st)
{
    if (n > 0) hipLaunchKernelGGL(scale_kernel, grid1(n, 256), dim3(256), 0, st, y, x, a, n);
    return hipGetLastError();
}
hipError_t launch_selftest(float* out, hipStream_t st)
{
    hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(64), 0, st, out);
    return hipGetLastError();
}
hipError_t launch_split_selftest(unsigned* out, hipStream_t st)
{
    hipLaunchKernelGGL(split_selftest_kernel, dim3(2048), dim3(256), 0, st, out);      // two workgroups' worth of waves on every CU
    return hipGetLastError();
}
hipError_t launch_nan_check(const float* x, long long n, int* flag, hipStream_t st)
{
    if (n > 0) hipLaunchKernelGGL(nan_check_kernel, grid1(n, 256), dim3(256), 0, st, x, n, flag);
    return hipGetLastError();
}

}  // namespace ti
