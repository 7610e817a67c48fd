// v_rsq_f64 against 1 / sqrt(v) in correctly rounded float64: largest relative error over 2^26 arguments spread over [1e-6, 1e9]
// (refine64's error bound multiplies by the raw instruction with 0.36 % of slack).   hipcc --offload-arch=gfx950 -O2 rsq64.hip -o rsq64 && ./rsq64
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
__global__ void k(double* out, unsigned n)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    double worst = 0.0;
    for (unsigned j = i; j < n; j += gridDim.x * blockDim.x) {
        const double t = (double)j / (double)n;                  // [0, 1)
        const double v = exp2(-20.0 + 50.0 * t) * (1.0 + 0.37 * (double)(j & 1023u) / 1024.0);
        const double a = __builtin_amdgcn_rsq(v), b = 1.0 / sqrt(v);
        worst = fmax(worst, fabs(a - b) / b);
    }
    out[i] = worst;
}
int main()
{
    const unsigned threads = 256 * 1024, n = 1u << 26;
    double* d;
    hipMalloc(&d, threads * sizeof(double));
    hipLaunchKernelGGL(k, dim3(threads / 256), dim3(256), 0, 0, d, n);
    double* h = new double[threads];
    hipMemcpy(h, d, threads * sizeof(double), hipMemcpyDeviceToHost);
    double w = 0;
    for (unsigned i = 0; i < threads; ++i) w = fmax(w, h[i]);
    printf("{\"v_rsq_f64_max_relative_error\": %.3e, \"log2\": %.2f, \"samples\": %u}\n", w, log2(w), n);
    return 0;
}
