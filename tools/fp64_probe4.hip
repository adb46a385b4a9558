// Scratch micro-benchmark 4: fp64 VALU issue rate of the instruction pairs the max-product kernel (K5) is made of.
//   mode 0: v_fma_f64 (the 78.6 TFLOP/s vector figure)   mode 1: v_mul_f64 + v_max_f64, register operands
//   mode 2: v_mul_f64 with an SGPR operand + v_max_f64    mode 3: v_max_f64 only   mode 4: v_mul_f64 only
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ inline double vmax(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, const double* __restrict__ seed, int iters) {
    double acc[16];
    for (int t = 0; t < 16; ++t) acc[t] = seed[threadIdx.x + t] * 1e-3;
    double b = seed[threadIdx.x & 31];
    const double* sp = seed + __builtin_amdgcn_readfirstlane(blockIdx.x & 7) * 16;
    double s[16];
    for (int t = 0; t < 16; ++t) s[t] = sp[t];                 // uniform -> SGPRs
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            if (MODE == 0) acc[t] = __builtin_fma(acc[t], b, b);
            if (MODE == 1) acc[t] = vmax(acc[t], b * acc[(t + 5) & 15]);
            if (MODE == 2) acc[t] = vmax(acc[t], b * s[t]);
            if (MODE == 3) acc[t] = vmax(acc[t], b);
            if (MODE == 4) acc[t] = acc[t] * b;
        }
        if (MODE == 2) b += 1e-9;
    }
    double r = 0;
    for (int t = 0; t < 16; ++t) r += acc[t];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int MODE>
void run(const char* name, double ops_per_elem, double* d_out, double* d_seed) {
    const int blocks = 256 * 8, iters = 20000;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, d_seed, 100);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, d_seed, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double lane_ops = (double)blocks * 256 * iters * 16 * ops_per_elem;
    std::printf("%-40s %8.3f ms  %7.2f T lane-ops/s\n", name, ms, lane_ops / (ms * 1e-3) / 1e12);
}

int main() {
    double *d_out, *d_seed;
    hipMalloc(&d_out, sizeof(double) * 256 * 8 * 256);
    std::vector<double> seed(1024);
    for (int i = 0; i < 1024; ++i) seed[i] = 0.5 + 1e-3 * i;
    hipMalloc(&d_seed, sizeof(double) * 1024);
    hipMemcpy(d_seed, seed.data(), sizeof(double) * 1024, hipMemcpyHostToDevice);
    run<0>("v_fma_f64", 1, d_out, d_seed);
    run<1>("v_mul_f64 + v_max_f64 (vgpr)", 2, d_out, d_seed);
    run<2>("v_mul_f64 (sgpr operand) + v_max_f64", 2, d_out, d_seed);
    run<3>("v_max_f64", 1, d_out, d_seed);
    run<4>("v_mul_f64", 1, d_out, d_seed);
    return 0;
}
