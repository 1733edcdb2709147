// mfma_denorm.hip -- does v_mfma_f32_16x16x32_f16 take fp16 SUBNORMAL inputs at face value or flush them to zero?
// (Decides whether the operand split could carry unscaled residuals -- one accumulator chain instead of two)
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/mfma_denorm tools/micro/mfma_denorm.hip && tools/micro/mfma_denorm
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float a_val, float b_val, float* out)
{
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)a_val; b[i] = (_Float16)b_val; }
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
}
int main()
{
    float* d; hipMalloc((void**)&d, 4);
    const float cases[][2] = {{0x1p-20f, 1.0f}, {1.0f, 0x1p-20f}, {0x1p-24f, 1.0f}, {0x1p-20f, 0x1p-4f}, {0x1p-14f, 1.0f}, {0x1.8p-16f, 2.0f}};
    for (auto& cs : cases) {
        k<<<1, 64>>>(cs[0], cs[1], d);
        float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        printf("a = %a (fp16 %s), b = %a: D[0][0] = %a, exact 32 a b = %a  -> %s\n", cs[0], cs[0] < 0x1p-14f ? "subnormal" : "normal", cs[1], h,
               32.0f * cs[0] * cs[1], h == 32.0f * cs[0] * cs[1] ? "kept" : h == 0.f ? "FLUSHED" : "other");
    }
    return 0;
}
