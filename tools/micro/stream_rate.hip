// Microbenchmark: how fast can ONE workgroup (the latency regime: a handful of workgroups on the whole chip) pull its weight-chunk
// stream through LDS-DMA, (a) when the stream is hot in its XCD's L2, (b) after the L2 has been flushed by other traffic?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../thermodynamic-interpolation_amd/csrc -o stream_rate stream_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "mfma_chain.hpp"
using namespace ti;

template <int WAVES, int SC, bool GEMM>
__global__ __launch_bounds__(64 * WAVES, 1) void k(float* out, const float4* stream, int nch)
{
    constexpr int NB = 4, NBK = 8, T = 64 * WAVES;
    extern __shared__ f32x4 lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), q = lane >> 4;
    PipeDMA<NB, T, SC> pipe;
    pipe.init(reinterpret_cast<const f32x4*>(stream), nch, lds, wave, lane);
    r16::Act<NBK> a;
    for (int nb = 0; nb < NBK; ++nb) a.b[nb] = f32x4{0.01f * lane + nb, 0.5f - 0.001f * lane, 0.25f * q, 1.0f / (1 + nb)};
    r16::Opnd<NBK, true> op;
    op.set(a);
    f32x4 acc[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
    for (int c = 0; c < nch; ++c) {
        const f32x4* wl = pipe.acquire();
        if (GEMM) r16::gemm_bt(acc[0], acc[1], op, wl, lane);
        else acc[0] += wl[lane];
        pipe.release();
    }
    pipe.drain();
    const f32x4 r = acc[0] + acc[1];
    out[blockIdx.x * T + threadIdx.x] = r[0] + r[1] + r[2] + r[3];
}
__global__ void flush(float4* p, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = float4{1, 2, 3, 4}; }

template <int WAVES, int SC, bool GEMM>
static void run(const char* name, float* out, const float4* stream, float4* junk, int nch, int blocks)
{
    const size_t l = 2 * SC * (size_t)1024 * 16 + 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<WAVES, SC, GEMM>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float hot = 1e9f, cold = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipLaunchKernelGGL((k<WAVES, SC, GEMM>), dim3(blocks), dim3(64 * WAVES), l, 0, out, stream, nch);          // warms the L2s of the XCDs it lands on
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<WAVES, SC, GEMM>), dim3(blocks), dim3(64 * WAVES), l, 0, out, stream, nch);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); hot = ms < hot ? ms : hot;
        hipLaunchKernelGGL(flush, dim3(2048), dim3(256), 0, 0, junk, (size_t)(512u << 20) / 16);                 // 512 MB of writes: L2 and most of the MALL turned over
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<WAVES, SC, GEMM>), dim3(blocks), dim3(64 * WAVES), l, 0, out, stream, nch);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); cold = ms < cold ? ms : cold;
    }
    const double mb = nch * 16384.0 / 1e6;
    printf("%-34s %3d WG(s), %3d chunks (%.2f MB): hot L2 %7.1f us = %6.1f GB/s per WG | after a flush %7.1f us = %6.1f GB/s per WG\n", name, blocks, nch, mb,
           hot * 1e3, mb / hot, cold * 1e3, mb / cold);
}

int main()
{
    float* out; float4* stream; float4* junk;
    const int nch = 116;
    hipMalloc(&out, 512 * 512 * 4); hipMalloc(&stream, nch * 16384); hipMalloc(&junk, 512u << 20);
    std::vector<_Float16> h((size_t)nch * 8192);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (_Float16)(((int)(i * 2654435761u >> 20) % 200 - 100) / 1280.0f);
    hipMemcpy(stream, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    for (int blocks : {1, 2, 8, 16, 64}) {
        run<4, 2, true>("4 waves, SC=2, one GEMM per chunk", out, stream, junk, nch, blocks);
        run<4, 4, true>("4 waves, SC=4, one GEMM per chunk", out, stream, junk, nch, blocks);
    }
    run<4, 2, false>("4 waves, SC=2, no GEMM", out, stream, junk, nch, 2);
    run<4, 4, false>("4 waves, SC=4, no GEMM", out, stream, junk, nch, 2);
    run<8, 4, false>("8 waves, SC=4, no GEMM", out, stream, junk, nch, 2);
    return 0;
}
