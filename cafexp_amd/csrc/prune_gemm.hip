// K2 prune_gemm: one interior branch of the post-order prune for every family (and gamma
// category) of a chunk at once.
//
// Replaces, for interior children, lambda::calculate_child_factor (src/lambda.cpp:13,32) ->
// matrix::multiply (src/matrix_cache.cpp:28-57) inside compute_node_probability
// (src/probability.cpp:201-241): the reference does one (M+1)-long mat-vec per child per
// family per category.  Here the families x categories of a chunk are the columns of the
// child's likelihood panel, so the branch is ONE dense fp64 GEMM per category
//     C[s, f] = sum_{c=0..M} P_child[s + row_off][c] * L_child[c, f]
// and the child product (probability.cpp:211-218, 233-240) is the epilogue: the first child of a
// parent stores C, later children multiply into the parent panel.
//
// Tiling: 128 x 128 block tile, K step 16, 256 threads = 4 waves in a 2 x 2 grid of 64 x 64
// wave tiles, v_mfma_f64_16x16x4_f64 (16 accumulator tiles = 128 VGPRs per lane).  The next
// K-tile is fetched global->registers while the current one is consumed from LDS.  LDS images:
// A as [128][16+2] and B as [16][128+16] doubles -- both fragment reads (ds_read_b64) are
// bank-conflict-free (row stride 18 doubles = 36 dwords walks all even banks; row stride 144
// doubles = 32 (mod 64) dwords puts k and k+1 on opposite bank halves).
// Bound: fp64 MFMA (intensity ~ 2*128*128*16 flop per 32 KB staged = 64 flop/B against L2).
#include "cafe_kernels.h"

namespace cafe {

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int kAStride = kBK + 2;      // 18
constexpr int kBStride = kBN + 16;     // 144

__global__ __launch_bounds__(256, 2) void prune_gemm_kernel(const GemmArgs a) {
    __shared__ double As[kBM * kAStride];
    __shared__ double Bs[kBK * kBStride];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int cat = blockIdx.z;

    // XCD-aware tile order: blocks b and b+8 share an XCD, so give each XCD a contiguous run of
    // tiles; inside a run the row tiles of one column tile are adjacent and share the B panel in L2.
    const int nblk = a.n_row_tiles * a.n_col_tiles;
    int bid = blockIdx.x;
    if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);
    const int row_tile = bid % a.n_row_tiles;
    const int col_tile = bid / a.n_row_tiles;
    const int row0 = row_tile * kBM;
    const int col0 = col_tile * kBN;

    const double* __restrict__ A = a.pool.base + (int64_t)a.slot[cat] * a.pool.stride;
    const double* __restrict__ B = a.src + (int64_t)cat * a.panel_kstride;
    double* __restrict__ C = a.dst + (int64_t)cat * a.panel_kstride;
    const int lda = a.pool.ld;
    const int ldb = a.ld;

    // staging assignment: 4 double2 per thread for each operand
    int a_row[4], a_c2[4], b_row[4], b_c2[4];
    bool a_ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + i * 256;
        a_row[i] = idx >> 3;
        a_c2[i] = (idx & 7) * 2;
        b_row[i] = idx >> 6;
        b_c2[i] = (idx & 63) * 2;
        a_ok[i] = (row0 + a_row[i] + a.row_off) < a.pool.n;
    }

    double2 ra[4], rb[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (a_ok[i])
                ra[i] = *reinterpret_cast<const double2*>(A + (int64_t)(row0 + a_row[i] + a.row_off) * lda + k0 + a_c2[i]);
            else
                ra[i] = make_double2(0.0, 0.0);
            rb[i] = *reinterpret_cast<const double2*>(B + (int64_t)(k0 + b_row[i]) * ldb + col0 + b_c2[i]);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<double2*>(&As[a_row[i] * kAStride + a_c2[i]]) = ra[i];
            *reinterpret_cast<double2*>(&Bs[b_row[i] * kBStride + b_c2[i]]) = rb[i];
        }
    };

    double4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = double4_t{0.0, 0.0, 0.0, 0.0};

    const int l15 = lane & 15, l4 = lane >> 4;
    const double* a_frag = &As[(wr * 64 + l15) * kAStride + l4];
    const double* b_frag = &Bs[l4 * kBStride + wc * 64 + l15];

    fetch(0);
    for (int k0 = 0; k0 < a.kc; k0 += kBK) {
        commit();
        __syncthreads();
        if (k0 + kBK < a.kc) fetch(k0 + kBK);
#pragma unroll
        for (int kk = 0; kk < kBK; kk += 4) {
            double af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = a_frag[i * 16 * kAStride + kk];
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[j] = b_frag[kk * kBStride + j * 16];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // epilogue: C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = row0 + wr * 64 + i * 16 + l4 + 4 * r;
            if (row >= a.rows_store) continue;
            double* crow = C + (int64_t)row * ldb + col0 + wc * 64 + l15;
            if (row < a.rows) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    double v = acc[i][j][r];
                    if (a.mode) v *= crow[j * 16];
                    crow[j * 16] = v;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) crow[j * 16] = 0.0;
            }
        }
    }
}

hipError_t launch_prune_gemm(const GemmArgs& a, int n_categories, hipStream_t stream) {
    dim3 grid(a.n_row_tiles * a.n_col_tiles, 1, n_categories), block(256);
    (void)hipGetLastError();
    hipLaunchKernelGGL(prune_gemm_kernel, grid, block, 0, stream, a);
    return hipGetLastError();
}

// ---- fp64 MFMA issue-rate probe (roofline denominator check) -----------------
__global__ __launch_bounds__(256) void mfma_probe_kernel(double* out, int iters) {
    double4_t acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = double4_t{0.0, 0.0, 0.0, 0.0};
    double x = 1.0 + 1e-9 * threadIdx.x, y = 1.0 - 1e-9 * threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[i], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

hipError_t launch_mfma_probe(double* d_out, int iters, int blocks, hipStream_t stream) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(mfma_probe_kernel, dim3(blocks), dim3(256), 0, stream, d_out, iters);
    return hipGetLastError();
}

}  // namespace cafe
