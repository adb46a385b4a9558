// Probe: layout of v_mfma_f64_16x16x4_f64's D with the operands in either order (is D(b, a) = D(a, b)^T ?).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ void probe(double* out) {
    const int lane = threadIdx.x, l15 = lane & 15, l4 = lane >> 4;
    // A[i][k] = 1 + i + 0.01 k (16 x 4), B[k][j] = 2 + 0.1 j + 0.001 k (4 x 16); fragments: (lane & 15, lane >> 4)
    const double a = 1.0 + l15 + 0.01 * l4, b = 2.0 + 0.1 * l15 + 0.001 * l4;
    double4_t z = {0, 0, 0, 0};
    double4_t d1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, z, 0, 0, 0);
    double4_t d2 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, z, 0, 0, 0);
    for (int r = 0; r < 4; ++r) { out[(lane * 4 + r) * 2] = d1[r]; out[(lane * 4 + r) * 2 + 1] = d2[r]; }
}
int main() {
    double* d; hipMalloc(&d, 64 * 4 * 2 * sizeof(double));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    double h[512]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    auto C = [](int i, int j) { double s = 0; for (int k = 0; k < 4; ++k) s += (1.0 + i + 0.01 * k) * (2.0 + 0.1 * j + 0.001 * k); return s; };
    int ok1 = 0, ok2 = 0, ok2b = 0;
    for (int lane = 0; lane < 64; ++lane) for (int r = 0; r < 4; ++r) {
        const int l15 = lane & 15, l4 = lane >> 4;
        const double d1 = h[(lane * 4 + r) * 2], d2 = h[(lane * 4 + r) * 2 + 1];
        ok1 += fabs(d1 - C(4 * r + l4, l15)) < 1e-9;          // D(a,b): row = l4 + 4 reg, col = l15
        ok2 += fabs(d2 - C(l15, 4 * r + l4)) < 1e-9;          // D(b,a): C^T with the same mapping?
        ok2b += fabs(d2 - C(l15, 4 * l4 + r)) < 1e-9;         // or col = 4 l4 + reg ?
    }
    printf("D(a,b) row=l4+4r,col=l15: %d/256   D(b,a)=C[l15][l4+4r]: %d/256   D(b,a)=C[l15][4*l4+r]: %d/256\n", ok1, ok2, ok2b);
    for (int r = 0; r < 4; ++r) printf("lane 17 reg %d: d1 %.6f d2 %.6f\n", r, h[(17 * 4 + r) * 2], h[(17 * 4 + r) * 2 + 1]);
    return 0;
}
