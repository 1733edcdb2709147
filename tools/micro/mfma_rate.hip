// Microbenchmark: cycles per v_mfma_f32_16x16x32_f16 for the issue patterns of the split-fp16 GEMM (tools/micro/mfma_rate.hip)
//   hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip && ./mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x4 mf(h8 a, h8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

// MODE 0: three chains per step as in gemm_split_chunk (acc, x, x) ; MODE 1: 8 independent accumulators; MODE 2: one chain
template <int MODE, int WPS>
__global__ __launch_bounds__(64 * 4 * WPS, WPS) void k(float* out, const h8* in, unsigned long long* cyc, int iters)
{
    const int lane = threadIdx.x & 63;
    h8 a[4], b[4];
    for (int m = 0; m < 4; ++m) { a[m] = in[m * 64 + lane]; b[m] = in[256 + m * 64 + lane]; }
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int m = s & 3;
            if (MODE == 0) { acc[0] = mf(a[m], b[m], acc[0]); acc[1] = mf(a[m], b[(m + 1) & 3], acc[1]); acc[1] = mf(a[(m + 1) & 3], b[m], acc[1]); }
            if (MODE == 1) { acc[(3 * s) & 7] = mf(a[m], b[m], acc[(3 * s) & 7]); acc[(3 * s + 1) & 7] = mf(a[m], b[(m + 1) & 3], acc[(3 * s + 1) & 7]); acc[(3 * s + 2) & 7] = mf(a[(m + 1) & 3], b[m], acc[(3 * s + 2) & 7]); }
            if (MODE == 2) { acc[0] = mf(a[m], b[m], acc[0]); acc[0] = mf(a[m], b[(m + 1) & 3], acc[0]); acc[0] = mf(a[(m + 1) & 3], b[m], acc[0]); }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    f32x4 r = acc[0];
    for (int i = 1; i < 8; ++i) r += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r[0] + r[1] + r[2] + r[3];
    if (lane == 0 && blockIdx.x == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

template <int MODE, int WPS>
static void run(const char* name, float* out, const h8* in, unsigned long long* cyc)
{
    const int iters = 20000, blocks = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, WPS>), dim3(blocks), dim3(64 * 4 * WPS), 0, 0, out, in, cyc, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, WPS>), dim3(blocks), dim3(64 * 4 * WPS), 0, 0, out, in, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[8]; hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost);
    const double n = 24.0 * iters;                     // MFMAs per wave
    printf("%-34s waves/SIMD %d: %6.2f cycles per MFMA per wave (s_memtime), %6.2f per SIMD; wall %.3f ms -> %.1f TFLOP/s chip, implied clock %.2f GHz\n", name, WPS,
           c[0] / n, c[0] / n / WPS, ms, n * WPS * 4 * blocks * 16384.0 / (ms * 1e-3) / 1e12, c[0] / (ms * 1e-3) / 1e9);
}

int main()
{
    float* out; h8* in; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&in, 512 * 16); hipMalloc(&cyc, 64);
    std::vector<_Float16> h(512 * 8);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (_Float16)(((int)(i * 2654435761u >> 20) % 200 - 100) / 128.0f);
    hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    run<0, 1>("split pattern (acc, x, x)", out, in, cyc);
    run<0, 2>("split pattern (acc, x, x)", out, in, cyc);
    run<1, 1>("8 independent accumulators", out, in, cyc);
    run<1, 2>("8 independent accumulators", out, in, cyc);
    run<2, 1>("single chain", out, in, cyc);
    run<2, 2>("single chain", out, in, cyc);
    return 0;
}
