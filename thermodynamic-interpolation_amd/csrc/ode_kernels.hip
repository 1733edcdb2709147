// ode_kernels.hip -- elementwise pieces of the Runge-Kutta drivers (ti_api.hip): stage combinations, the scaled error /
// step-size norms of the adaptive solver (deterministic two-pass reductions), the quartic dense-output fit and evaluation.
//
// The adaptive solver restates torchdiffeq 0.2.5 (`dopri5`, /root/reference/ti_env.yml:14 -- third-party, not in the
// reference checkout): rk_common.py `_runge_kutta_step`, `_compute_error_ratio`, `_interp_fit`, `_interp_evaluate` and
// misc.py `_select_initial_step`, `_rms_norm`.  State arrays are fp32 like the reference's tensors; sums of squares are
// accumulated in fp64 and reduced in a fixed order, so accept / reject decisions repeat bit for bit.
#include "ti_internal.hpp"

namespace ti {

namespace {

constexpr int RED_BLOCK = 256;

__device__ __forceinline__ float comb(const RkComb& c, long long i)
{
    float acc = c.c[0] * c.k[0][i];
    for (int j = 1; j < c.nk; ++j) acc = fmaf(c.c[j], c.k[j][i], acc);
    return acc;
}

// block sum in a fixed tree order; result valid in thread 0
__device__ __forceinline__ double block_sum(double v)
{
    __shared__ double sm[RED_BLOCK];
    sm[threadIdx.x] = v;
    __syncthreads();
    for (int s = RED_BLOCK / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s];
        __syncthreads();
    }
    return sm[0];
}

}  // namespace

// y = y0 + sum_j c_j k_j
__global__ void rk_combo_kernel(float* __restrict__ y, const float* __restrict__ y0, RkComb c, long long n)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = y0[i] + comb(c, i);
}

// partial[b] = sum_i ((sum_j c_j k_j[i]) / (atol + rtol max(|y0_i|, |y1_i|)))^2      (_compute_error_ratio)
__global__ __launch_bounds__(RED_BLOCK) void rk_ratio_partial_kernel(double* __restrict__ partial, const float* __restrict__ y0,
                                                                     const float* __restrict__ y1, RkComb c, float rtol, float atol, long long n)
{
    double acc = 0.0;
    for (long long i = (long long)blockIdx.x * RED_BLOCK + threadIdx.x; i < n; i += (long long)gridDim.x * RED_BLOCK) {
        const float tol = atol + rtol * fmaxf(fabsf(y0[i]), fabsf(y1[i]));
        const float r = comb(c, i) / tol;
        acc += (double)r * (double)r;
    }
    acc = block_sum(acc);
    if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}

// partial[b] = sum_i ((a_i - b_i) / (atol + rtol |y0_i|))^2   (b may be NULL)                (_select_initial_step)
__global__ __launch_bounds__(RED_BLOCK) void scaled_sq_partial_kernel(double* __restrict__ partial, const float* __restrict__ a,
                                                                      const float* __restrict__ b, const float* __restrict__ y0, float rtol,
                                                                      float atol, long long n)
{
    double acc = 0.0;
    for (long long i = (long long)blockIdx.x * RED_BLOCK + threadIdx.x; i < n; i += (long long)gridDim.x * RED_BLOCK) {
        const float scale = atol + fabsf(y0[i]) * rtol;
        const float r = (b ? a[i] - b[i] : a[i]) / scale;
        acc += (double)r * (double)r;
    }
    acc = block_sum(acc);
    if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}

__global__ __launch_bounds__(RED_BLOCK) void reduce_partials_kernel(double* __restrict__ out, const double* __restrict__ partial, int nb)
{
    double acc = 0.0;
    for (int i = threadIdx.x; i < nb; i += RED_BLOCK) acc += partial[i];
    acc = block_sum(acc);
    if (threadIdx.x == 0) *out = acc;
}

// coefficients [5][n] of the quartic through (y0, f0), (y_mid), (y1, f1) on [t0, t0 + dt]   (_interp_fit)
__global__ void interp_fit_kernel(float* __restrict__ coef, const float* __restrict__ y0, const float* __restrict__ y1,
                                  const float* __restrict__ f0, const float* __restrict__ f1, RkComb mid, float dt, long long n)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float a0 = y0[i], a1 = y1[i], g0 = f0[i], g1 = f1[i];
    const float ym = a0 + comb(mid, i);
    coef[i] = a0;
    coef[n + i] = dt * g0;
    coef[2 * n + i] = dt * (g1 - 4.0f * g0) - 11.0f * a0 - 5.0f * a1 + 16.0f * ym;
    coef[3 * n + i] = dt * (5.0f * g0 - 3.0f * g1) + 18.0f * a0 + 14.0f * a1 - 32.0f * ym;
    coef[4 * n + i] = 2.0f * dt * (g1 - g0) - 8.0f * (a1 + a0) + 16.0f * ym;
}

// total = c0 + x c1 + x^2 c2 + x^3 c3 + x^4 c4 in torchdiffeq's evaluation order   (_interp_evaluate)
__global__ void interp_eval_kernel(float* __restrict__ out, const float* __restrict__ coef, float x, long long n)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float total = coef[i] + x * coef[n + i];
    float xp = x;
#pragma unroll
    for (int k = 2; k < 5; ++k) { xp = xp * x; total = total + xp * coef[k * n + i]; }
    out[i] = total;
}

static inline dim3 grid1(long long n, int bs) { return dim3((unsigned)((n + bs - 1) / bs)); }

hipError_t launch_rk_combo(float* y, const float* y0, const RkComb& c, long long n, hipStream_t st)
{
    if (n > 0) hipLaunchKernelGGL(rk_combo_kernel, grid1(n, 256), dim3(256), 0, st, y, y0, c, n);
    return hipGetLastError();
}
static int red_blocks(long long n) { return (int)std::min<long long>(RED_PARTIALS, std::max<long long>(1, (n + RED_BLOCK - 1) / RED_BLOCK)); }
hipError_t launch_rk_ratio_sumsq(double* out, double* partial, const float* y0, const float* y1, const RkComb& c, float rtol, float atol,
                                 long long n, hipStream_t st)
{
    const int nb = red_blocks(n);
    hipLaunchKernelGGL(rk_ratio_partial_kernel, dim3(nb), dim3(RED_BLOCK), 0, st, partial, y0, y1, c, rtol, atol, n);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(RED_BLOCK), 0, st, out, partial, nb);
    return hipGetLastError();
}
hipError_t launch_scaled_sumsq(double* out, double* partial, const float* a, const float* b, const float* y0, float rtol, float atol,
                               long long n, hipStream_t st)
{
    const int nb = red_blocks(n);
    hipLaunchKernelGGL(scaled_sq_partial_kernel, dim3(nb), dim3(RED_BLOCK), 0, st, partial, a, b, y0, rtol, atol, n);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(RED_BLOCK), 0, st, out, partial, nb);
    return hipGetLastError();
}
hipError_t launch_interp_fit(float* coef, const float* y0, const float* y1, const float* f0, const float* f1, const RkComb& mid, float dt,
                             long long n, hipStream_t st)
{
    if (n > 0) hipLaunchKernelGGL(interp_fit_kernel, grid1(n, 256), dim3(256), 0, st, coef, y0, y1, f0, f1, mid, dt, n);
    return hipGetLastError();
}
hipError_t launch_interp_eval(float* out, const float* coef, float x, long long n, hipStream_t st)
{
    if (n > 0) hipLaunchKernelGGL(interp_eval_kernel, grid1(n, 256), dim3(256), 0, st, out, coef, x, n);
    return hipGetLastError();
}

}  // namespace ti
