// coexec_classes.hip (round 3; coexec.hip is the round-2 scalar-fma version) -- do matrix (MFMA) and vector (VALU) instructions of TWO waves on one SIMD overlap on gfx950, or only
// within one wave's own instruction stream?  (the message kernels' two waves per SIMD add up instead of overlapping.)
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o tools/micro/coexec_classes tools/micro/coexec_classes.hip   (here), then on the GPU box: tools/micro/coexec_classes
// One 512-thread workgroup per CU (LDS-padded): waves w and w + 4 share a SIMD.  Roles by wave index:
//   mode 0: waves 0-3 issue N fp16 16x16x32 MFMAs (chains of 3 on one accumulator, like the split products), waves 4-7 exit
//   mode 1: waves 4-7 issue V dependent-free v_fma_f32, waves 0-3 exit
//   mode 2: both (MFMA wave and VALU wave on every SIMD)
//   mode 3: all 8 waves MFMA;   mode 4: all 8 waves VALU
//   mode 5: waves 0-3 issue the MFMAs with the VALU work of mode 1 interleaved in their own stream (3 MFMA : K VALU), waves 4-7 exit
//   mode 6: as 2, the MFMA waves at s_setprio 1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int ITER = 4096;      // outer iterations
constexpr int VPER = 7;         // VALU instructions per 3 MFMAs in the interleaved mode (the message kernel: 2.2 per MFMA)

template <bool PRIO>
__device__ __forceinline__ f32x4 mfma_loop(h8 a, h8 b, f32x4 acc)
{
    if (PRIO) __builtin_amdgcn_s_setprio(1);
    f32x4 c0 = acc, c1 = acc, c2 = acc, c3 = acc;
    for (int i = 0; i < ITER; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, a, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, c1, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, a, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, c2, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, a, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c3, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, c3, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, a, c3, 0, 0, 0);
    }
    if (PRIO) __builtin_amdgcn_s_setprio(0);
    return (c0 + c1) + (c2 + c3);
}

__device__ __forceinline__ float valu_loop(float x, float y)
{
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = x + k;
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int r = 0; r < 4 * VPER / 8 + 1; ++r)          // ~ 4 * VPER per iteration: the VALU work that goes with 12 MFMAs
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = __builtin_fmaf(v[k], y, x);
    }
    float s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += v[k];
    return s;
}

__device__ __forceinline__ float both_loop(h8 a, h8 b, f32x4 acc, float x, float y, f32x4& out)
{
    f32x4 c0 = acc, c1 = acc, c2 = acc, c3 = acc;
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = x + k;
    for (int i = 0; i < ITER; ++i) {
#define M3(c) c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); c = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, c, 0, 0, 0); c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, a, c, 0, 0, 0);
#define V8 _Pragma("unroll") for (int k = 0; k < 8; ++k) v[k] = __builtin_fmaf(v[k], y, x);
        M3(c0) V8 M3(c1) V8 M3(c2) V8 M3(c3) V8
        // pin the pattern: 1 MFMA, then 2-3 VALU in its shadow
#define G3 __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 3, 0); __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 3, 0); __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
        G3 G3 G3 G3
    }
    float s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += v[k];
    out = (c0 + c1) + (c2 + c3);
    return s;
}

// ---- the same with v_mfma_f32_32x32x16_f16 (8 passes; same FLOP rate, half as many instructions): does a longer matrix instruction leave
// the VALU port to the other wave / to the wave's own vector instructions?
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ f32x16 mfma32_loop(h8 a, h8 b, float x)
{
    f32x16 c0, c1;
#pragma unroll
    for (int k = 0; k < 16; ++k) { c0[k] = x + k; c1[k] = x - k; }
    for (int i = 0; i < ITER; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, a, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c1, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, a, c1, 0, 0, 0);
    }
    return c0 + c1;
}
__device__ __forceinline__ float both32_loop(h8 a, h8 b, float x, float y, f32x16& out)
{
    f32x16 c0, c1;
#pragma unroll
    for (int k = 0; k < 16; ++k) { c0[k] = x + k; c1[k] = x - k; }
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = x + k;
    for (int i = 0; i < ITER; ++i) {
#define N3(c) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); c = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c, 0, 0, 0); c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, a, c, 0, 0, 0);
        N3(c0) V8 V8 N3(c1) V8 V8
#define H3 __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 6, 0); __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 5, 0); __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
        H3 H3
    }
    float s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += v[k];
    out = c0 + c1;
    return s;
}

__global__ __launch_bounds__(512, 1) void coexec(int mode, float* sink, float x, float y)
{
    extern __shared__ float pad[];
    const int wave = threadIdx.x >> 6;
    const bool lo = wave < 4;
    h8 a, b;
#pragma unroll
    for (int k = 0; k < 8; ++k) { a[k] = (_Float16)(x + k); b[k] = (_Float16)(y - k); }
    f32x4 acc = {x, y, x, y};
    float r = 0;
    const bool do_m = mode == 3 || mode == 5 || ((mode == 0 || mode == 2 || mode == 6) && lo);
    const bool do_v = mode == 4 || ((mode == 1 || mode == 2 || mode == 6) && !lo);
    if (mode >= 7) {                 // 7: 32x32x16 waves only; 8: + VALU wave per SIMD; 9: interleaved in one wave; 10: two 32x32x16 waves per SIMD
        const bool m32 = mode == 10 || ((mode == 7 || mode == 8) && lo);
        if (mode == 9) { if (lo) { f32x16 o; r = both32_loop(a, b, x, y, o); for (int k = 0; k < 16; ++k) r += o[k]; } }
        else if (m32) { f32x16 o = mfma32_loop(a, b, x); for (int k = 0; k < 16; ++k) r += o[k]; }
        else if (mode == 8 && !lo) r = valu_loop(x, y);
    } else
    if (mode == 5) { if (lo) { f32x4 o; r = both_loop(a, b, acc, x, y, o); r += o[0] + o[1] + o[2] + o[3]; } }
    else if (do_m) { f32x4 o = mode == 6 ? mfma_loop<true>(a, b, acc) : mfma_loop<false>(a, b, acc); r = o[0] + o[1] + o[2] + o[3]; }
    else if (do_v) r = valu_loop(x, y);
    if (r == 12345.678f) sink[threadIdx.x] = r + pad[threadIdx.x];
}

// ---- which vector instruction classes run in the shadow of another wave's matrix instructions?  The VALU wave issues 32 independent
// instructions of ONE class per iteration (inline asm, so that hipcc neither packs nor folds them).
template <int CLS>
__device__ __forceinline__ float class_loop(float x, float y)
{
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = x + k;
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int k = 0; k < 8; k += 2) {
                if (CLS == 0) { asm volatile("v_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %2, %3" : "+v"(v[k]), "+v"(v[k + 1]) : "v"(y), "v"(x)); }
                else if (CLS == 1) {          // packed fp32: one instruction on the register pair (counts as one of the 16 per iteration)
                    typedef float f2 __attribute__((ext_vector_type(2)));
                    f2 t = {v[k], v[k + 1]}, yy = {y, y}, xx = {x, x};
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(t) : "v"(yy), "v"(xx));
                    v[k] = t.x; v[k + 1] = t.y;
                }
                else if (CLS == 2) { asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1" : "+v"(v[k]), "+v"(v[k + 1])); }
                else if (CLS == 3) { asm volatile("v_cvt_pk_f16_f32 %0, %0, %1\n\tv_cvt_pk_f16_f32 %1, %1, %0" : "+v"(v[k]), "+v"(v[k + 1])); }
                else if (CLS == 4) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(v[k]), "+v"(v[k + 1])); }
                else if (CLS == 5) { asm volatile("v_fma_mixlo_f16 %0, %0, %2, %1 op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %1, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(v[k]), "+v"(v[k + 1]) : "v"(y)); }
            }
    }
    float s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += v[k];
    return s;
}

__global__ __launch_bounds__(512, 1) void coexec_cls(int cls, int with_mfma, float* sink, float x, float y)
{
    extern __shared__ float pad[];
    const bool lo = (threadIdx.x >> 6) < 4;
    h8 a, b;
#pragma unroll
    for (int k = 0; k < 8; ++k) { a[k] = (_Float16)(x + k); b[k] = (_Float16)(y - k); }
    f32x4 acc = {x, y, x, y};
    float r = 0;
    if (lo) { if (with_mfma) { f32x4 o = mfma_loop<false>(a, b, acc); r = o[0] + o[1] + o[2] + o[3]; } }
    else switch (cls) {
        case 0: r = class_loop<0>(x, y); break;
        case 1: r = class_loop<1>(x, y); break;
        case 2: r = class_loop<2>(x, y); break;
        case 3: r = class_loop<3>(x, y); break;
        case 4: r = class_loop<4>(x, y); break;
        default: r = class_loop<5>(x, y); break;
    }
    if (r == 12345.678f) sink[threadIdx.x] = r + pad[threadIdx.x];
}

// ---- closer to the message kernel: the matrix wave reads its B fragments from LDS (two ds_read_b128 per six MFMAs, one step ahead,
// waited for with lgkmcnt), the vector wave runs a LayerNorm-like mix (per 32 instructions: 16 fma, 4 exp, 4 rcp, 4 cvt_pk, 4 fma_mix)
__device__ __forceinline__ f32x4 mfma_lds_loop(const h8* frag, int lane, h8 a, f32x4 acc)
{
    f32x4 c0 = acc, c1 = acc;
    h8 f0 = frag[lane], f1 = frag[64 + lane];
    for (int i = 0; i < 2 * ITER; ++i) {
        const int nx = ((i + 1) & 63) * 128;
        const h8 n0 = frag[nx + lane], n1 = frag[nx + 64 + lane];
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, f0, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(f0, a, c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(f1, f0, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, f1, c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, f1, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(f1, a, c1, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        f0 = n0; f1 = n1;
    }
    return c0 + c1;
}
// the same FLOPs on v_mfma_f32_32x32x16_f16: two fragments feed three instructions (hi.hi, lo.hi, hi.lo of one 32 x 32 tile), two tiles in flight
__device__ __forceinline__ f32x16 mfma32_lds_loop(const h8* frag, int lane, h8 a, float x)
{
    f32x16 c0, c1;
#pragma unroll
    for (int k = 0; k < 16; ++k) { c0[k] = x + k; c1[k] = x - k; }
    h8 f0 = frag[lane], f1 = frag[64 + lane];
    for (int i = 0; i < ITER; ++i) {
        const int nx = ((2 * i + 1) & 63) * 128, ny = ((2 * i + 2) & 63) * 128;
        const h8 n0 = frag[nx + lane], n1 = frag[nx + 64 + lane];
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, f0, c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f1, f0, c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, f1, c0, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        const h8 m0 = frag[ny + lane], m1 = frag[ny + 64 + lane];
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, n0, c1, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(n1, n0, c1, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, n1, c1, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        f0 = m0; f1 = m1;
    }
    return c0 + c1;
}
__device__ __forceinline__ float mix_loop(float x, float y)
{
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = x + 0.01f * k;
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int k = 0; k < 8; k += 2) {
            asm volatile("v_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %2, %3\n\tv_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %2, %3" : "+v"(v[k]), "+v"(v[k + 1]) : "v"(y), "v"(x));
            asm volatile("v_exp_f32 %0, %0\n\tv_rcp_f32 %1, %1" : "+v"(v[k]), "+v"(v[k + 1]));
            asm volatile("v_cvt_pk_f16_f32 %0, %0, %1\n\tv_fma_mixlo_f16 %1, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(v[k]), "+v"(v[k + 1]) : "v"(y));
        }
    }
    float s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += v[k];
    return s;
}
__global__ __launch_bounds__(512, 1) void coexec_lds(int mode, float* sink, float x, float y)
{
    extern __shared__ float pad[];
    h8* frag = reinterpret_cast<h8*>(pad);
    for (int i = threadIdx.x; i < 64 * 128; i += 512) { h8 t; for (int k = 0; k < 8; ++k) t[k] = (_Float16)(0.001f * ((i + k) & 15)); frag[i] = t; }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool lo = wave < 4;
    h8 a;
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = (_Float16)(x + k);
    f32x4 acc = {x, y, x, y};
    float r = 0;
    // 0: LDS-fed matrix waves only; 1: mix waves only; 2: both; 3: both, matrix wave at priority 1; 4: two LDS-fed matrix waves; 5: two mix waves
    // 6 / 7 / 8: as 0 / 2 / 4 with the 32x32x16 instruction
    if (mode >= 6) {
        const bool m32 = mode == 8 || lo;
        if (m32) { f32x16 o = mfma32_lds_loop(frag, lane, a, x); for (int k = 0; k < 16; ++k) r += o[k]; }
        else if (mode == 7) r = mix_loop(x, y);
        if (r == 12345.678f) sink[threadIdx.x] = r;
        return;
    }
    const bool do_m = mode == 4 || ((mode == 0 || mode == 2 || mode == 3) && lo);
    const bool do_v = mode == 5 || ((mode == 1 || mode == 2 || mode == 3) && !lo);
    if (do_m) { if (mode == 3) __builtin_amdgcn_s_setprio(1); f32x4 o = mfma_lds_loop(frag, lane, a, acc); r = o[0] + o[1] + o[2] + o[3]; }
    else if (do_v) r = mix_loop(x, y);
    if (r == 12345.678f) sink[threadIdx.x] = r;
}

int main()
{
    float* sink;
    hipMalloc(&sink, 4096);
    hipFuncSetAttribute((const void*)coexec, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[] = {"MFMA waves only (one per SIMD)", "VALU waves only (one per SIMD)", "MFMA wave + VALU wave per SIMD", "two MFMA waves per SIMD",
                           "two VALU waves per SIMD", "one wave per SIMD, MFMA and VALU interleaved in its stream", "MFMA wave (s_setprio 1) + VALU wave per SIMD",
                           "32x32x16 MFMA waves only (one per SIMD; 6 per iteration = the same FLOPs)", "32x32x16 MFMA wave + VALU wave per SIMD", "one wave per SIMD, 32x32x16 MFMA and VALU interleaved", "two 32x32x16 MFMA waves per SIMD"};
    for (int mode = 0; mode < 11; ++mode) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(coexec, dim3(256), dim3(512), 100 * 1024, 0, mode, sink, 1.0f, 0.5f);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep && ms < best) best = ms;
        }
        printf("mode %2d  %-74s %8.3f ms\n", mode, names[mode], best);
    }
    hipFuncSetAttribute((const void*)coexec_cls, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    const char* cls[] = {"v_fma_f32 x32", "v_pk_fma_f32 x16", "v_exp_f32 x32", "v_cvt_pk_f16_f32 x32", "v_permlane16/32_swap x32 (+ s_nop 1 each)", "v_fma_mixlo/hi_f16 x32"};
    printf("vector class per iteration (alone | next to a 16x16x32 MFMA wave on the same SIMD, which alone takes the mode-0 time):\n");
    for (int c = 0; c < 6; ++c) {
        float t[2];
        for (int w = 0; w < 2; ++w) {
            float best = 1e30f;
            for (int rep = 0; rep < 4; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(coexec_cls, dim3(256), dim3(512), 100 * 1024, 0, c, w, sink, 1.0f, 0.5f);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep && ms < best) best = ms;
            }
            t[w] = best;
        }
        printf("  %-44s %8.3f ms | %8.3f ms\n", cls[c], t[0], t[1]);
    }
    hipFuncSetAttribute((const void*)coexec_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
    const char* ln[] = {"LDS-fed matrix waves only (12 MFMA + 4 ds_read_b128 per iteration)", "LayerNorm-like vector waves only (16 fma, 4 exp, 4 rcp, 4 cvt_pk, 4 fma_mix)",
                        "LDS-fed matrix wave + LayerNorm-like wave", "the same, matrix wave at s_setprio 1", "two LDS-fed matrix waves", "two LayerNorm-like waves",
                        "LDS-fed 32x32x16 matrix waves only (6 MFMA + 4 ds_read_b128 per iteration: the same FLOPs and bytes)", "LDS-fed 32x32x16 matrix wave + LayerNorm-like wave", "two LDS-fed 32x32x16 matrix waves"};
    printf("closer to the message kernel:\n");
    for (int mode = 0; mode < 9; ++mode) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(coexec_lds, dim3(256), dim3(512), 140 * 1024, 0, mode, sink, 1.0f, 0.5f);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep && ms < best) best = ms;
        }
        printf("  lds mode %d  %-84s %8.3f ms\n", mode, ln[mode], best);
    }
    if (hipGetLastError() != hipSuccess) { printf("HIP error\n"); return 1; }
    return 0;
}
