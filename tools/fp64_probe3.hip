// Scratch micro-benchmark 3: fp64 MFMA issue rate when every k-step's operands come from LDS
// (GEMM-like: 9 A + 2 B fragment reads per 18 MFMAs), fixed wall-time window.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double double4_t __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: registers only, 1: ds_read_b64 x11, 2: ds_read2_b64 (compiler choice), 3: reads issued one step ahead
__global__ __launch_bounds__(256, 2) void k_lds(double* out, unsigned long long* iters_out, int window, const double* seed) {
    __shared__ double lds[2 * 16 * 144 * 2];
    for (int i = threadIdx.x; i < 2 * 16 * 144 * 2; i += 256) lds[i] = seed[i & 1023];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    double4_t acc[9][2];
    for (int i = 0; i < 9; ++i) { acc[i][0] = double4_t{0, 0, 0, 0}; acc[i][1] = double4_t{0, 0, 0, 0}; }
    unsigned long long n = 0;
    const unsigned long long t_end = __builtin_amdgcn_s_memrealtime() + window;
    const double* As0 = lds + l4 * 144 + l15;
    const double* Bs0 = lds + 16 * 144 + l4 * 144 + wave * 32 + l15;
    double af[2][9], bf[2][2];
    for (int i = 0; i < 9; ++i) { af[0][i] = seed[lane + i]; af[1][i] = seed[lane + i + 64]; }
    bf[0][0] = seed[lane]; bf[0][1] = seed[lane + 1]; bf[1][0] = seed[lane + 2]; bf[1][1] = seed[lane + 3];
    while (__builtin_amdgcn_s_memrealtime() < t_end) {
        for (int rep = 0; rep < 8; ++rep) {
            const double* As = As0 + (rep & 1) * (2 * 16 * 144);
            const double* Bs = Bs0 + (rep & 1) * (2 * 16 * 144);
            asm volatile("" ::: "memory");
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                if (MODE == 1 || MODE == 2) {
#pragma unroll
                    for (int i = 0; i < 9; ++i) af[s4 & 1][i] = As[s4 * 4 * 144 + i * 16];
                    bf[s4 & 1][0] = Bs[s4 * 4 * 144];
                    bf[s4 & 1][1] = Bs[s4 * 4 * 144 + 16];
                }
                if (MODE == 3) {
#pragma unroll
                    for (int i = 0; i < 9; ++i) af[(s4 + 1) & 1][i] = As[((s4 + 1) & 3) * 4 * 144 + i * 16];
                    bf[(s4 + 1) & 1][0] = Bs[((s4 + 1) & 3) * 4 * 144];
                    bf[(s4 + 1) & 1][1] = Bs[((s4 + 1) & 3) * 4 * 144 + 16];
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int i = 0; i < 9; ++i) {
                    acc[i][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[s4 & 1][i], bf[s4 & 1][0], acc[i][0], 0, 0, 0);
                    acc[i][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[s4 & 1][i], bf[s4 & 1][1], acc[i][1], 0, 0, 0);
                }
                if (MODE == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
        n += 8 * 4 * 18;
    }
    double s = 0;
    for (int i = 0; i < 9; ++i) s += acc[i][0][0] + acc[i][1][1] + acc[i][0][2] + acc[i][1][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) iters_out[blockIdx.x * 4 + wave] = n;
}

int main() {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const int maxblocks = cus * 2;
    double* d; unsigned long long* it; double* seed;
    (void)hipMalloc(&d, sizeof(double) * maxblocks * 256);
    (void)hipMalloc(&it, 8 * maxblocks * 4);
    std::vector<double> hs(2048);
    unsigned long long r = 88172645463325252ull;
    for (auto& v : hs) { r ^= r << 13; r ^= r >> 7; r ^= r << 17; v = 1e-3 * (double)(r >> 11) / 9007199254740992.0; }
    (void)hipMalloc(&seed, sizeof(double) * 2048);
    (void)hipMemcpy(seed, hs.data(), sizeof(double) * 2048, hipMemcpyHostToDevice);
    const int window = 2000000;
    for (int bpc = 1; bpc <= 2; ++bpc)
        for (int mode = 0; mode < 4; ++mode) {
            const int blocks = cus * bpc;
            (void)hipMemset(it, 0, 8 * maxblocks * 4);
            switch (mode) {
                case 0: hipLaunchKernelGGL(k_lds<0>, dim3(blocks), dim3(256), 0, 0, d, it, window, seed); break;
                case 1: hipLaunchKernelGGL(k_lds<1>, dim3(blocks), dim3(256), 0, 0, d, it, window, seed); break;
                case 2: hipLaunchKernelGGL(k_lds<2>, dim3(blocks), dim3(256), 0, 0, d, it, window, seed); break;
                case 3: hipLaunchKernelGGL(k_lds<3>, dim3(blocks), dim3(256), 0, 0, d, it, window, seed); break;
            }
            (void)hipDeviceSynchronize();
            std::vector<unsigned long long> hi(maxblocks * 4);
            (void)hipMemcpy(hi.data(), it, 8 * maxblocks * 4, hipMemcpyDeviceToHost);
            double nm = 0;
            for (int i = 0; i < blocks * 4; ++i) nm += hi[i];
            printf("waves/SIMD %d  mode %d : MFMA %6.2f TF\n", bpc, mode, nm * 2048.0 / (window / 100e6) / 1e12);
        }
    return 0;
}
