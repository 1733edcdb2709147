// Microbenchmark of the weight-chunk stream of the edge kernel alone: LDS-DMA superchunks, barriers, fragment reads and the
// 3-product split-fp16 MFMA steps of mfma_chain.hpp -- no LayerNorm, no gathers, no atomics.  Reports cycles per chunk.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../thermodynamic-interpolation_amd/csrc [-DTI_FRAG_AHEAD=n] -o chunk_stream chunk_stream.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "mfma_chain.hpp"
using namespace ti;

template <int WAVES, int SC, bool FLIP, int MODE>
__global__ __launch_bounds__(64 * WAVES, 8 / WAVES) void k(float* out, const float4* stream, int nch, int iters)
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
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {          // 8 chunks: two superchunks of 4, or four of 2
            // MODE 0: the product pipe; 1: no DMA (the LDS image is reused), barrier kept; 2: no DMA, no barrier; 3: DMA + barrier, no MFMA
            const f32x4* wl = MODE == 0 || MODE == 3 ? pipe.acquire() : lds + (c % (2 * SC)) * 1024;
            if (MODE != 3) { if (FLIP) r16::gemm_fl(acc[c & 3], acc[4 + (c & 3)], op, wl, lane); else r16::gemm_bt(acc[c & 3], acc[4 + (c & 3)], op, wl, lane); }
            if (MODE == 0 || MODE == 3) pipe.release();
            else if (MODE == 1 && c % SC == SC - 1) __syncthreads();
        }
    }
    pipe.drain();
    f32x4 r = acc[0];
    for (int i = 1; i < 8; ++i) r += acc[i];
    out[blockIdx.x * T + threadIdx.x] = r[0] + r[1] + r[2] + r[3];
}

template <int WAVES, int SC, bool FLIP, int MODE = 0>
static void run(const char* name, float* out, const float4* stream)
{
    const int iters = 2000, nch = 56, per_cu = 8 / WAVES, blocks = 256 * per_cu;
    const size_t l = 2 * SC * (size_t)1024 * 16;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<WAVES, SC, FLIP, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<WAVES, SC, FLIP, MODE>), dim3(blocks), dim3(64 * WAVES), l, 0, out, stream, nch, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<WAVES, SC, FLIP, MODE>), dim3(blocks), dim3(64 * WAVES), l, 0, out, stream, nch, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double chunks_per_simd = 8.0 * iters * 2;             // two waves per SIMD, 8 chunks per iteration each
    printf("%-28s %7.3f ms  %7.1f ns per chunk per SIMD = %6.0f cycles at 2.0 GHz (24 MFMAs = 384 matrix cycles)  -> %.0f TFLOP/s of fp16 MFMA\n", name, ms,
           ms * 1e6 / chunks_per_simd, ms * 1e6 / chunks_per_simd * 2.0, chunks_per_simd * 1024 * 24 * 16384.0 / (ms * 1e-3) / 1e12);
}

int main()
{
    float* out; float4* stream;
    hipMalloc(&out, 512 * 512 * 4); hipMalloc(&stream, 56 * 16384);
    std::vector<_Float16> h(56 * 8192);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (_Float16)(((int)(i * 2654435761u >> 20) % 200 - 100) / 1280.0f);
    hipMemcpy(stream, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    run<8, 4, false>("8 waves, SC=4, bt", out, stream);
    run<8, 4, true>("8 waves, SC=4, fl", out, stream);
    run<8, 2, false>("8 waves, SC=2, bt", out, stream);
    run<4, 2, false>("2 x 4 waves, SC=2, bt", out, stream);
    run<8, 4, false, 1>("8w SC=4 bt, no DMA", out, stream);
    run<8, 4, false, 2>("8w SC=4 bt, no DMA no barrier", out, stream);
    run<8, 4, false, 3>("8w SC=4, DMA+barrier only", out, stream);
    run<4, 2, false, 2>("2x4w SC=2 bt, no DMA no bar", out, stream);
    return 0;
}
