// Scratch micro-benchmark 2: co-issue of fp64 MFMA waves and fp64 VALU-FMA waves on one SIMD,
// measured over a fixed wall-time window (every wave loops until a common s_memrealtime deadline).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void k_roles(double* out, unsigned long long* iters_out, unsigned long long* cyc_out,
                                               unsigned mask, int window_100mhz_ticks, const double* seed) {
    const int wave = threadIdx.x >> 6;
    const bool mfma_role = (mask >> wave) & 1u;
    double4_t acc[8];
    double f[16];
    double x = seed[threadIdx.x], y = seed[512 + threadIdx.x];
    for (int i = 0; i < 8; ++i) acc[i] = double4_t{x, y, x, y};
    for (int i = 0; i < 16; ++i) f[i] = x + i;
    unsigned long long n = 0;
    const unsigned long long t_end = __builtin_amdgcn_s_memrealtime() + window_100mhz_ticks;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    if (mfma_role) {
        while (__builtin_amdgcn_s_memrealtime() < t_end) {
            for (int rep = 0; rep < 8; ++rep) {
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[i], 0, 0, 0);
            }
            n += 64;
        }
    } else {
        while (__builtin_amdgcn_s_memrealtime() < t_end) {
            for (int rep = 0; rep < 8; ++rep) {
#pragma unroll
                for (int i = 0; i < 16; ++i) f[i] = __builtin_fma(f[i], y, x);
            }
            n += 128;
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 16; ++i) s += f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        iters_out[blockIdx.x * 8 + wave] = n;
        cyc_out[blockIdx.x * 8 + wave] = c1 - c0;
    }
}

int main() {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    double* d; unsigned long long *it, *cy; double* seed;
    const int maxblocks = cus * 4;
    (void)hipMalloc(&d, sizeof(double) * maxblocks * 512);
    (void)hipMalloc(&it, 8 * maxblocks * 8); (void)hipMalloc(&cy, 8 * maxblocks * 8);
    std::vector<double> hs(1024);
    unsigned long long r = 88172645463325252ull;
    for (auto& v : hs) { r ^= r << 13; r ^= r >> 7; r ^= r << 17; v = 0.5 + (double)(r >> 11) / 9007199254740992.0; }   // [0.5,1.5)
    (void)hipMalloc(&seed, sizeof(double) * 1024);
    (void)hipMemcpy(seed, hs.data(), sizeof(double) * 1024, hipMemcpyHostToDevice);
    const int window = 2000000;   // 20 ms at 100 MHz
    struct Cfg { const char* name; unsigned mask; int bpc; };
    // waves 0..3 -> SIMD 0..3, waves 4..7 -> SIMD 0..3 (second wave of each SIMD)
    Cfg cfgs[] = {
        {"1 MFMA wave/SIMD (other idle-exit)", 0x0Fu | 0x100u, 1},   // special: handled below (waves 4-7 exit)
        {"2 MFMA waves/SIMD", 0xFFu, 1},
        {"4 MFMA waves/SIMD", 0xFFu, 2},
        {"1 MFMA + 1 VALU /SIMD", 0x0Fu, 1},
        {"2 MFMA + 2 VALU /SIMD", 0x0Fu, 2},
        {"3 MFMA + 1 VALU /SIMD (2 blocks: F,F0)", 0x0Fu, 2},        // placeholder, see below
        {"2 VALU waves/SIMD", 0x00u, 1},
        {"4 VALU waves/SIMD", 0x00u, 2},
    };
    for (int rep = 0; rep < 2; ++rep)
    for (auto& c : cfgs) {
        const int blocks = cus * c.bpc;
        (void)hipMemset(it, 0, 8 * maxblocks * 8); (void)hipMemset(cy, 0, 8 * maxblocks * 8);
        hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        (void)hipEventRecord(a);
        hipLaunchKernelGGL(k_roles, dim3(blocks), dim3(c.mask & 0x100u ? 256 : 512), 0, 0, d, it, cy, c.mask & 0xFFu, window, seed);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        std::vector<unsigned long long> hi(maxblocks * 8), hc(maxblocks * 8);
        (void)hipMemcpy(hi.data(), it, 8 * maxblocks * 8, hipMemcpyDeviceToHost);
        (void)hipMemcpy(hc.data(), cy, 8 * maxblocks * 8, hipMemcpyDeviceToHost);
        double nm = 0, nv = 0, cm = 0, cv = 0; int wm = 0, wv = 0;
        const int wpb = (c.mask & 0x100u) ? 4 : 8;
        for (int bl = 0; bl < blocks; ++bl) for (int w = 0; w < wpb; ++w) {
            bool m = (c.mask >> w) & 1u;
            if (m) { nm += hi[bl * 8 + w]; cm += hc[bl * 8 + w]; ++wm; } else { nv += hi[bl * 8 + w]; cv += hc[bl * 8 + w]; ++wv; }
        }
        const double secs = window / 100e6;
        printf("%-42s wall %6.2f ms | MFMA %6.2f TF (%.1f cyc/mfma/wave, clk %.2f GHz) | VALU %6.2f TF (%.1f cyc/fma/wave)\n", c.name, ms,
               nm * 2048.0 / secs / 1e12, wm ? cm / nm : 0.0, wm ? cm / wm / secs / 1e9 : (wv ? cv / wv / secs / 1e9 : 0.0),
               nv * 128.0 / secs / 1e12, wv ? cv / nv : 0.0);
    }
    return 0;
}
