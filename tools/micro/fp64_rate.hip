// Issue rate of fp64 vector instructions on one GPU: 8 independent chains per lane, 4 waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/micro/fp64_rate.hip -o gpurun_out/fp64_rate
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP>
__global__ __launch_bounds__(256) void k(double *out, double b, double c, int n)
{
    double a[8];
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x + i;
    for (int it = 0; it < n; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 0) a[i] = fma(a[i], b, c);
            else if (OP == 1) a[i] = a[i] + b;
            else a[i] = a[i] * b;
        }
    }
    double s = 0;
    for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main()
{
    double *out;
    const int blocks = 256 * 4;  // 4 workgroups of 4 waves per CU: 4 waves per SIMD
    hipMalloc(&out, blocks * 256 * sizeof(double));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int n = 20000;
    const char *names[] = {"v_fma_f64", "v_add_f64", "v_mul_f64"};
    for (int op = 0; op < 3; op++) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (op == 0) k<0><<<blocks, 256>>>(out, 1.0000001, 1e-9, n);
            else if (op == 1) k<1><<<blocks, 256>>>(out, 1.0000001, 1e-9, n);
            else k<2><<<blocks, 256>>>(out, 1.0000001, 1e-9, n);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double wave_instr = (double)blocks * 4 * n * 8;  // wave-level instructions
        const double per_simd = wave_instr / (256.0 * 4.0);
        printf("%s: %.3f ms, %.2f ns per wave instruction per SIMD (= %.1f cycles at 2.4 GHz), %.1f T instr-lanes/s\n",
               names[op], ms, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4, wave_instr * 64 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
