// split_probe.hip -- the product's operand split (mfma_chain.hpp: Opnd<8, true>::set -> split_quad, 8 vector instructions per four
// values, half-register writes) against the plain arithmetic, bit for bit.  The two forms run in SEPARATE kernels on the same
// generated values (nothing of one can be scheduled into or folded with the other) and the host compares the fp16 halves.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I thermodynamic-interpolation_amd/csrc -o tools/micro/split_probe tools/micro/split_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#include "mfma_chain.hpp"

using namespace ti;

__device__ __forceinline__ void make_values(unsigned gid, r16::Act<8>& x)
{
#pragma unroll
    for (int nb = 0; nb < 8; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            unsigned h = (gid * 32u + nb * 4u + r) * 2654435761u;
            h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
            const int e = (int)(h % 45u) - 30;                                   // 2^-30 .. 2^14
            const unsigned keep = (h >> 8) % 5u == 0 ? 0x7fe000u : (h >> 8) % 7u == 0 ? 0u : 0x7fffffu;   // exact fp16 values, powers of two
            float v = __builtin_bit_cast(float, (unsigned)((e + 127) << 23) | ((h >> 9) & keep));
            if ((h >> 3) % 11u == 0) v = 0.f;
            x.b[nb][r] = (h & 1u) ? -v : v;
        }
}

__global__ void split_product(unsigned short* hi, unsigned short* lo)
{
    const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
    r16::Act<8> x;
    make_values(gid, x);
    r16::Opnd<8, true> o;
    o.set(x);
    // whole registers, as the matrix instructions consume them (16-byte stores; element i of k-step m is half i of register m)
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        reinterpret_cast<r16::h8*>(hi)[(size_t)gid * 4 + m] = o.hi[m];
        reinterpret_cast<r16::h8*>(lo)[(size_t)gid * 4 + m] = o.lo[m];
    }
}

__global__ void split_plain(unsigned short* hi, unsigned short* lo, float* vals)
{
    const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
    r16::Act<8> x;
    make_values(gid, x);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float v = i < 4 ? x.b[2 * m][i] : x.b[2 * m + 1][i - 4];
            const _Float16 h = (_Float16)v, l = (_Float16)((v - (float)h) * 2048.0f);
            hi[(size_t)gid * 32 + m * 8 + i] = __builtin_bit_cast(unsigned short, h);
            lo[(size_t)gid * 32 + m * 8 + i] = __builtin_bit_cast(unsigned short, l);
            vals[(size_t)gid * 32 + m * 8 + i] = v;
        }
}

int main()
{
    const int blocks = 2048, threads = 256;
    const size_t n = (size_t)blocks * threads * 32;
    unsigned short *h1, *l1, *h2, *l2; float* vals;
    hipMalloc((void**)&h1, n * 2); hipMalloc((void**)&l1, n * 2); hipMalloc((void**)&h2, n * 2); hipMalloc((void**)&l2, n * 2); hipMalloc((void**)&vals, n * 4);
    split_product<<<blocks, threads>>>(h1, l1);
    split_plain<<<blocks, threads>>>(h2, l2, vals);
    std::vector<unsigned short> a(n), b(n), c(n), d(n); std::vector<float> v(n);
    hipMemcpy(a.data(), h1, n * 2, hipMemcpyDeviceToHost); hipMemcpy(b.data(), l1, n * 2, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), h2, n * 2, hipMemcpyDeviceToHost); hipMemcpy(d.data(), l2, n * 2, hipMemcpyDeviceToHost);
    hipMemcpy(v.data(), vals, n * 4, hipMemcpyDeviceToHost);
    size_t bad_hi = 0, bad_lo = 0, sub = 0, shown = 0;
    for (size_t i = 0; i < n; ++i) {
        bad_hi += a[i] != c[i];
        if (b[i] != d[i]) { ++bad_lo; if (shown++ < 8) printf("  v = %a: lo product %04x plain %04x (hi %04x / %04x), element %zu of its lane\n", v[i], b[i], d[i], a[i], c[i], i % 32); }
        sub += (d[i] & 0x7c00) == 0 && (d[i] & 0x3ff) != 0;
    }
    printf("values %zu: hi halves that differ %zu, lo halves that differ %zu (plain lo fp16-subnormal: %zu)\n", n, bad_hi, bad_lo, sub);
    return bad_hi || bad_lo ? 1 : 0;
}
