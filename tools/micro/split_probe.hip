// split_probe.hip -- do the two forms of the fp32 -> (hi, 2^11 lo) fp16 split (mfma_chain.hpp: split_pair) give the same bits?
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/split_probe tools/micro/split_probe.hip && tools/micro/split_probe
// Values span 2^-40 .. 2^15 with random mantissas; reports, per binade of |2^11 (v - hi)|, how many lo halves differ and the
// largest |difference| of the reconstructed value hi + 2^-11 lo.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void probe(const float* v, unsigned short* hi, unsigned short* lo_ref, unsigned short* lo_mix, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const float v0 = v[2 * i], v1 = v[2 * i + 1];
    const h2 h = h2{(_Float16)v0, (_Float16)v1};
    const h2 lr = h2{(_Float16)((v0 - (float)h[0]) * 2048.0f), (_Float16)((v1 - (float)h[1]) * 2048.0f)};
    const unsigned hb = __builtin_bit_cast(unsigned, h);
    const float s0 = v0 * 2048.0f, s1 = v1 * 2048.0f, c = -2048.0f;
    unsigned d;
    asm volatile("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(hb), "s"(c), "v"(s0));
    asm volatile("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(d) : "v"(hb), "s"(c), "v"(s1));
    const unsigned r = __builtin_bit_cast(unsigned, lr);
    hi[2 * i] = hb & 0xffff; hi[2 * i + 1] = hb >> 16;
    lo_ref[2 * i] = r & 0xffff; lo_ref[2 * i + 1] = r >> 16;
    lo_mix[2 * i] = d & 0xffff; lo_mix[2 * i + 1] = d >> 16;
}

static float h2f(unsigned short h)
{
    const int s = h >> 15, e = (h >> 10) & 31, m = h & 1023;
    float x = e == 0 ? std::ldexp((float)m, -24) : e == 31 ? INFINITY : std::ldexp((float)(m + 1024), e - 25);
    return s ? -x : x;
}

int main()
{
    const int n = 1 << 20;
    std::vector<float> v(n);
    srand(1);
    for (int i = 0; i < n; ++i) {
        const int e = rand() % 56 - 40;
        v[i] = std::ldexp(1.0f + (rand() & 0x7fffff) / 8388608.0f, e) * ((rand() & 1) ? -1.f : 1.f);
    }
    float* dv; unsigned short *dh, *dr, *dm;
    hipMalloc(&dv, n * 4); hipMalloc(&dh, n * 2); hipMalloc(&dr, n * 2); hipMalloc(&dm, n * 2);
    hipMemcpy(dv, v.data(), n * 4, hipMemcpyHostToDevice);
    probe<<<n / 2 / 256, 256>>>(dv, dh, dr, dm, n);
    std::vector<unsigned short> hi(n), lr(n), lm(n);
    hipMemcpy(hi.data(), dh, n * 2, hipMemcpyDeviceToHost); hipMemcpy(lr.data(), dr, n * 2, hipMemcpyDeviceToHost);
    hipMemcpy(lm.data(), dm, n * 2, hipMemcpyDeviceToHost);
    long differ = 0, differ_normal = 0, mix_zero_ref_sub = 0, ref_sub = 0;
    double worst = 0, worst_rel = 0;
    for (int i = 0; i < n; ++i) {
        const float a = h2f(lr[i]), b = h2f(lm[i]);
        const bool sub = (lr[i] & 0x7c00) == 0 && (lr[i] & 0x3ff) != 0;
        ref_sub += sub;
        if (lr[i] != lm[i]) {
            ++differ;
            if (!sub) ++differ_normal;
            if (sub && (lm[i] & 0x7fff) == 0) ++mix_zero_ref_sub;
            const double dabs = std::fabs((double)a - b) / 2048.0;
            worst = std::max(worst, dabs); worst_rel = std::max(worst_rel, dabs / std::fabs(v[i]));
        }
    }
    printf("values %d: lo halves that differ %ld (of them with a NORMAL reference lo: %ld); reference lo subnormal %ld, mix form 0 there %ld\n", n,
           differ, differ_normal, ref_sub, mix_zero_ref_sub);
    printf("largest |difference| of hi + 2^-11 lo: %.3e absolute, %.3e relative to |v|\n", worst, worst_rel);
    return 0;
}
