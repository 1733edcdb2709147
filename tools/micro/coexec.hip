// Microbenchmark: do the matrix pipe and the VALU of one SIMD run side by side on gfx950, (a) from two different waves, (b) from one wave's
// instruction stream?  Decides whether the serial "GEMM burst, then LayerNorm / conversion" phases of the edge kernel can be overlapped
// by skewing the waves of a SIMD, or only by interleaving inside a wave.   hipcc --offload-arch=gfx950 -O3 -o coexec coexec.hip && ./coexec
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x4 mf(h8 a, h8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
#define VFMA(x) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x) : "v"(ca), "v"(cb))

// one "unit" = 24 MFMAs (three chains, as gemm_split_chunk) and/or 96 VALU fmas (4 per MFMA, the edge kernel's measured ratio)
template <bool DO_M, bool DO_V, bool INTERLEAVE>
__device__ __forceinline__ void unit(f32x4 (&acc)[4], float (&x)[8], const h8 (&a)[4], const h8 (&b)[4], float ca, float cb)
{
    if (INTERLEAVE) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int m = s & 3;
            acc[0] = mf(a[m], b[m], acc[0]);
            VFMA(x[0]); VFMA(x[1]); VFMA(x[2]); VFMA(x[3]);
            acc[1] = mf(a[m], b[(m + 1) & 3], acc[1]);
            VFMA(x[4]); VFMA(x[5]); VFMA(x[6]); VFMA(x[7]);
            acc[2] = mf(a[(m + 1) & 3], b[m], acc[2]);
            VFMA(x[0]); VFMA(x[1]); VFMA(x[2]); VFMA(x[3]);
        }
        return;
    }
    if (DO_M) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int m = s & 3;
            acc[0] = mf(a[m], b[m], acc[0]); acc[1] = mf(a[m], b[(m + 1) & 3], acc[1]); acc[2] = mf(a[(m + 1) & 3], b[m], acc[2]);
        }
    }
    if (DO_V) {
#pragma unroll
        for (int s = 0; s < 12; ++s) { VFMA(x[0]); VFMA(x[1]); VFMA(x[2]); VFMA(x[3]); VFMA(x[4]); VFMA(x[5]); VFMA(x[6]); VFMA(x[7]); }
    }
}

// MODE 0 all waves matrix only | 1 all waves VALU only | 2 waves 0-3 matrix, waves 4-7 VALU (one of each per SIMD) | 3 every wave interleaves
// both in one stream | 4 every wave alternates a matrix phase and a VALU phase of PH units, all waves in phase | 5 same, waves 4-7 start with
// the VALU phase (skewed by one phase)
template <int MODE, int WPS>
__global__ __launch_bounds__(64 * 4 * WPS, 1) void k(float* out, const h8* in, unsigned long long* cyc, int iters, int PH)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    h8 a[4], b[4];
    for (int m = 0; m < 4; ++m) { a[m] = in[m * 64 + lane]; b[m] = in[256 + m * 64 + lane]; }
    f32x4 acc[4];
    float x[8];
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0, 0, 0, 0};
    for (int i = 0; i < 8; ++i) x[i] = lane * 0.001f + i;
    const float ca = 0.999f, cb = 0.001f;
    const bool second = wave >= 4;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (MODE == 0) for (int it = 0; it < iters; ++it) unit<true, false, false>(acc, x, a, b, ca, cb);
    if (MODE == 1) for (int it = 0; it < iters; ++it) unit<false, true, false>(acc, x, a, b, ca, cb);
    if (MODE == 2) {
        if (!second) for (int it = 0; it < iters; ++it) unit<true, false, false>(acc, x, a, b, ca, cb);
        else         for (int it = 0; it < iters; ++it) unit<false, true, false>(acc, x, a, b, ca, cb);
    }
    if (MODE == 3) for (int it = 0; it < iters; ++it) unit<true, true, true>(acc, x, a, b, ca, cb);
    if (MODE == 4 || MODE == 5) {
        const bool skew = MODE == 5 && second;
        for (int it = 0; it < iters; it += PH) {
            if (!skew) { for (int u = 0; u < PH; ++u) unit<true, false, false>(acc, x, a, b, ca, cb); for (int u = 0; u < PH; ++u) unit<false, true, false>(acc, x, a, b, ca, cb); }
            else       { for (int u = 0; u < PH; ++u) unit<false, true, false>(acc, x, a, b, ca, cb); for (int u = 0; u < PH; ++u) unit<true, false, false>(acc, x, a, b, ca, cb); }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    f32x4 r = acc[0] + acc[1] + acc[2] + acc[3];
    float s = r[0] + r[1] + r[2] + r[3];
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

template <int MODE, int WPS>
static void run(const char* name, float* out, const h8* in, unsigned long long* cyc, int PH = 1)
{
    const int iters = 4096, blocks = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, WPS>), dim3(blocks), dim3(64 * 4 * WPS), 0, 0, out, in, cyc, 64, PH);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, WPS>), dim3(blocks), dim3(64 * 4 * WPS), 0, 0, out, in, cyc, iters, PH);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[8]; hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost);
    printf("%-78s waves/SIMD %d PH %3d: wall %7.3f ms; s_memtime ticks per unit: wave0 %.0f  wave%d %.0f\n", name, WPS, PH, ms, (double)c[0] / iters, 4 * WPS - 1,
           (double)c[4 * WPS - 1] / iters);
}

int main()
{
    float* out; h8* in; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&in, 512 * 16); hipMalloc(&cyc, 64);
    std::vector<_Float16> h(512 * 8);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (_Float16)(((int)(i * 2654435761u >> 20) % 200 - 100) / 128.0f);
    hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    printf("unit = 24 fp16 MFMAs (16x16x32) and / or 96 v_fma_f32 per wave; 4096 units per wave\n");
    run<0, 1>("matrix only", out, in, cyc);
    run<0, 2>("matrix only", out, in, cyc);
    run<1, 1>("VALU only", out, in, cyc);
    run<1, 2>("VALU only", out, in, cyc);
    run<2, 2>("one wave matrix, the other VALU on each SIMD (half the work of the rows below)", out, in, cyc);
    run<3, 1>("matrix + VALU interleaved in one stream", out, in, cyc);
    run<3, 2>("matrix + VALU interleaved in one stream", out, in, cyc);
    for (int ph : {1, 8, 64}) {
        run<4, 2>("phases: matrix then VALU, both waves of a SIMD in phase", out, in, cyc, ph);
        run<5, 2>("phases: matrix then VALU, second wave of a SIMD skewed by one phase", out, in, cyc, ph);
    }
    run<4, 1>("phases: matrix then VALU", out, in, cyc, 8);
    return 0;
}
