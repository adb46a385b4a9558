// Scratch micro-benchmark: fp64 MFMA / VALU issue rates on gfx950 (roofline denominator study).
// hipcc --offload-arch=gfx950 -O3 tools/fp64_probe.hip -o tools/fp64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double double4_t __attribute__((ext_vector_type(4)));

template <int NACC, int NFMA>
__global__ __launch_bounds__(256) void k_mix(double* out, long long* cyc, int iters) {
    double4_t acc[NACC > 0 ? NACC : 1];
    for (int i = 0; i < (NACC > 0 ? NACC : 1); ++i) acc[i] = double4_t{0, 0, 0, 0};
    double f[NFMA > 0 ? NFMA : 1];
    for (int i = 0; i < (NFMA > 0 ? NFMA : 1); ++i) f[i] = 1.0 + threadIdx.x * 1e-9 + i;
    double x = 1.0 + 1e-9 * threadIdx.x, y = 1.0 - 1e-9 * threadIdx.x;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NFMA; ++i) f[i] = __builtin_fma(f[i], y, x);
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < (NACC > 0 ? NACC : 1); ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < (NFMA > 0 ? NFMA : 1); ++i) s += f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// role split by wave: waves with (wave & 1) do MFMA, others VALU (two waves per SIMD when 512 threads)
template <int NACC, int NFMA>
__global__ __launch_bounds__(512) void k_split(double* out, long long* cyc, int iters) {
    const int wave = threadIdx.x >> 6;
    double4_t acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = double4_t{0, 0, 0, 0};
    double f[NFMA];
    for (int i = 0; i < NFMA; ++i) f[i] = 1.0 + threadIdx.x * 1e-9 + i;
    double x = 1.0 + 1e-9 * threadIdx.x, y = 1.0 - 1e-9 * threadIdx.x;
    long long t0 = __builtin_amdgcn_s_memtime();
    if (wave >= 4) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[i], 0, 0, 0);
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NFMA; ++i) f[i] = __builtin_fma(f[i], y, x);
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < NFMA; ++i) s += f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <typename F>
void run(const char* name, F launch, int blocks, int threads, int iters, double flops_per_thread_iter_mfma, double flops_valu, int ncyc) {
    double* d; long long* c;
    hipMalloc(&d, sizeof(double) * blocks * threads);
    hipMalloc(&c, sizeof(long long) * blocks * 8);
    hipMemset(c, 0, sizeof(long long) * blocks * 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch(d, c, iters / 10);
    hipDeviceSynchronize();
    hipEventRecord(a);
    launch(d, c, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<long long> hc(blocks * 8);
    hipMemcpy(hc.data(), c, sizeof(long long) * blocks * 8, hipMemcpyDeviceToHost);
    double cy = 0; int n = 0;
    for (int i = 0; i < ncyc; ++i) if (hc[i] > 0) { cy += hc[i]; ++n; }
    cy /= (n ? n : 1);
    double waves = (double)blocks * threads / 64;
    double tf_m = waves * iters * flops_per_thread_iter_mfma / (ms * 1e-3) / 1e12;
    double tf_v = waves * iters * flops_valu / (ms * 1e-3) / 1e12;
    printf("%-44s ms %8.3f  cycles/iter(s_memtime 100MHz ticks?) %9.2f  MFMA TF %7.2f  VALU TF %7.2f  clk(GHz est) %.3f\n", name, ms, cy / iters, tf_m, tf_v,
           cy / (ms * 1e-3) / 1e9);
    hipFree(d); hipFree(c);
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    int cus = p.multiProcessorCount;
    printf("CUs %d clock %d kHz\n", cus, p.clockRate);
    const int it = 20000;
#define MIX(NA, NF, BPC, name) run(name, [&](double* d, long long* c, int n) { hipLaunchKernelGGL((k_mix<NA, NF>), dim3(cus * BPC), dim3(256), 0, 0, d, c, n); }, cus * BPC, 256, it, NA * 2048.0, NF * 128.0, cus * BPC)
    MIX(1, 0, 1, "mfma x1 acc, 1 wave/SIMD");
    MIX(1, 0, 2, "mfma x1 acc, 2 waves/SIMD");
    MIX(1, 0, 4, "mfma x1 acc, 4 waves/SIMD");
    MIX(1, 0, 8, "mfma x1 acc, 8 waves/SIMD");
    MIX(2, 0, 2, "mfma x2 acc, 2 waves/SIMD");
    MIX(2, 0, 4, "mfma x2 acc, 4 waves/SIMD");
    MIX(2, 0, 8, "mfma x2 acc, 8 waves/SIMD");
    MIX(4, 0, 2, "mfma x4 acc, 2 waves/SIMD");
    MIX(4, 0, 3, "mfma x4 acc, 3 waves/SIMD");
    MIX(4, 0, 4, "mfma x4 acc, 4 waves/SIMD");
    MIX(4, 0, 6, "mfma x4 acc, 6 waves/SIMD");
    MIX(4, 0, 8, "mfma x4 acc, 8 waves/SIMD");
    MIX(8, 0, 1, "mfma x8 acc, 1 wave/SIMD");
    MIX(8, 0, 2, "mfma x8 acc, 2 waves/SIMD");
    MIX(8, 0, 3, "mfma x8 acc, 3 waves/SIMD");
    MIX(8, 0, 4, "mfma x8 acc, 4 waves/SIMD");
    MIX(0, 8, 2, "valu fma x8, 2 waves/SIMD");
    MIX(0, 8, 4, "valu fma x8, 4 waves/SIMD");
    MIX(8, 4, 2, "same wave: 8 mfma + 4 fma, 2 w/SIMD");
    MIX(8, 8, 4, "same wave: 8 mfma + 8 fma, 4 w/SIMD");
    run("split waves: 8 mfma | 32 fma (2 w/SIMD)", [&](double* d, long long* c, int n) { hipLaunchKernelGGL((k_split<8, 32>), dim3(cus), dim3(512), 0, 0, d, c, n); }, cus, 512, it, 0.5 * 8 * 2048.0, 0.5 * 32 * 128.0, cus * 8);
    run("split waves: 8 mfma | 64 fma (2 w/SIMD)", [&](double* d, long long* c, int n) { hipLaunchKernelGGL((k_split<8, 64>), dim3(cus), dim3(512), 0, 0, d, c, n); }, cus, 512, it, 0.5 * 8 * 2048.0, 0.5 * 64 * 128.0, cus * 8);
    run("split waves x2 blocks: 8 mfma | 64 fma (4 w/SIMD)", [&](double* d, long long* c, int n) { hipLaunchKernelGGL((k_split<8, 64>), dim3(cus * 2), dim3(512), 0, 0, d, c, n); }, cus * 2, 512, it, 0.5 * 8 * 2048.0, 0.5 * 64 * 128.0, cus * 16);
    return 0;
}
