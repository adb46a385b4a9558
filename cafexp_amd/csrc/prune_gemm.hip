// K2 prune_gemm: one interior branch of the post-order prune for every family (and gamma
// category) of a chunk at once.
//
// Replaces, for interior children, lambda::calculate_child_factor (src/lambda.cpp:13,32) ->
// matrix::multiply (src/matrix_cache.cpp:28-57) inside compute_node_probability
// (src/probability.cpp:201-241): the reference does one (M+1)-long mat-vec per child per
// family per category.  Here the families x categories of a chunk are the columns of the
// child's likelihood panel, so the branch is ONE dense fp64 GEMM per category
//     C[s, f] = sum_{c=0..M} P_child[s][c] * L_child[c, f],     s = 1..M  (1..R under the root)
// and the child product (probability.cpp:211-218, 233-240) is the epilogue: the first child of a
// parent stores C, later children multiply into the parent panel.  Parent size 0 is not part of
// the GEMM: P[0][c] = delta(c,0) (matrix_cache.cpp:70-77), so that row is a copy of the child's
// row 0, done by the blocks of the first row tile.
//
// Operands: A = the branch's k-major matrix Pt[c][s-1] (bd_matrix.hip), B = the child panel
// [c][family]; both tiles are [16 k][row/col] images filled by LDS-DMA (global_load_lds_dwordx4,
// one contiguous 1 KB piece per wave instruction, no VGPR staging), double-buffered, one barrier
// per K step.  Fragment reads are ds_read_b64 at [k = lane>>4][16*blk + (lane&15)]: with a row
// stride = 16 (mod 32) doubles, k and k+1 fall on opposite bank halves: conflict-free.
// Tiling: block tile (16*MI) x 128, 4 waves side by side along the family axis, each MI x 2
// accumulator tiles of v_mfma_f64_16x16x4_f64 in VGPRs (AGPR accumulators halve the issue rate of
// the f64 MFMA on gfx950).  MI is chosen per launch to minimise row padding
// (M = 720 -> MI = 9: 5 tiles of 144 rows, no padding).  Bound: fp64 MFMA.
#include "cafe_kernels.h"

namespace cafe {

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int kBStride = kBN + 16;                        // 144 doubles
constexpr int a_stride(int bm) { return (bm % 32 == 16) ? bm : bm + 16; }

template <int MI>
__global__ __launch_bounds__(256, 2) void prune_gemm_kernel(const GemmArgs a) {
    constexpr int BM = 16 * MI;
    constexpr int SA = a_stride(BM);
    constexpr int A_TILE = kBK * SA, B_TILE = kBK * kBStride;
    __shared__ double lds[2 * (A_TILE + B_TILE)];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int cat = blockIdx.z;

    // XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch, a speed-only
    // assumption), so XCD x takes the column tiles x, x+8, ... and runs their row tiles back to back:
    // the row tiles of one column tile then share the child panel (B) in that XCD's L2.
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int row_tile = idx % a.n_row_tiles;
    const int col_tile = xcd + 8 * (idx / a.n_row_tiles);
    if (col_tile >= a.n_col_tiles) return;
    const int row0 = row_tile * BM;                        // parent size row0 + 1 is the tile's first row
    const int col0 = col_tile * kBN;

    const double* __restrict__ A = a.pool.base + (int64_t)a.slot[cat] * a.pool.stride + row0;
    const double* __restrict__ B = a.src + (int64_t)cat * a.panel_kstride + col0;
    double* __restrict__ C = a.dst + (int64_t)cat * a.panel_kstride + col0;
    const int lda = a.pool.ld;
    const int ldb = a.ld;

    // LDS-DMA: wave w fills k-rows w, w+4, w+8, w+12 of both tiles; a row is 1 KB pieces of 64 x 16 B
    auto stage = [&](int k0, int buf) {
        double* As = lds + buf * (A_TILE + B_TILE);
        double* Bs = As + A_TILE;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int krow = wave + 4 * r;
            const double* ga = A + (int64_t)(k0 + krow) * lda;
#pragma unroll
            for (int off = 0; off < BM; off += 128) {
                const int nl = (BM - off) >= 128 ? 64 : (BM - off) / 2;
                if (lane < nl)
                    __builtin_amdgcn_global_load_lds((gptr_t)(ga + off + lane * 2), (lptr_t)(As + krow * SA + off), 16, 0, 0);
            }
            const double* gb = B + (int64_t)(k0 + krow) * ldb;
            __builtin_amdgcn_global_load_lds((gptr_t)(gb + lane * 2), (lptr_t)(Bs + krow * kBStride), 16, 0, 0);
        }
    };

    double4_t acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        acc[i][0] = double4_t{0.0, 0.0, 0.0, 0.0};
        acc[i][1] = double4_t{0.0, 0.0, 0.0, 0.0};
    }

    const int l15 = lane & 15, l4 = lane >> 4;
    const int n_k = (a.k_valid + kBK - 1) / kBK;

    stage(0, 0);
    __syncthreads();                                       // vmcnt(0) + barrier: tile 0 has landed
    for (int kt = 0; kt < n_k; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < n_k) stage((kt + 1) * kBK, buf ^ 1);
        const double* As = lds + buf * (A_TILE + B_TILE) + l4 * SA + l15;
        const double* Bs = lds + buf * (A_TILE + B_TILE) + A_TILE + l4 * kBStride + wave * 32 + l15;
        // every k-step of a tile is executed: rows c > M of the k-major matrix are zero, so the
        // padded steps of the last tile add exact zeros (and keep the loop free of branches)
#pragma unroll
        for (int kk = 0; kk < kBK; kk += 4) {
            double af[MI], bf[2];
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = As[kk * SA + i * 16];
            bf[0] = Bs[kk * kBStride];
            bf[1] = Bs[kk * kBStride + 16];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                acc[i][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[0], acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[1], acc[i][1], 0, 0, 0);
            }
        }
        __syncthreads();                                   // next tile landed, this one fully read
    }

    // epilogue: C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = row0 + i * 16 + l4 + 4 * r;    // parent size row + 1
            if (row < a.rows) {
                double* crow = C + (int64_t)(row + a.out_off) * ldb + wave * 32 + l15;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    double v = acc[i][j][r];
                    if (a.mode) v *= crow[j * 16];
                    crow[j * 16] = v;
                }
            }
        }
    }
    // parent size 0 only reaches child size 0 with probability 1: copy the child's row 0
    if (a.out_off == 1 && row_tile == 0 && tid < kBN) {
        double v = B[tid];
        if (a.mode) v *= C[tid];
        C[tid] = v;
    }
}

int prune_gemm_pick_mi(int rows) {
    int best = 9, best_cost = 1 << 30;
    for (int mi = 9; mi >= 4; --mi) {
        const int bm = 16 * mi;
        const int cost = (rows + bm - 1) / bm * bm;
        if (cost < best_cost) { best_cost = cost; best = mi; }
    }
    return best;
}

hipError_t launch_prune_gemm(const GemmArgs& a, int n_categories, hipStream_t stream) {
    dim3 grid(8 * ((a.n_col_tiles + 7) / 8) * a.n_row_tiles, 1, n_categories), block(256);
    (void)hipGetLastError();
    switch (a.mi) {
        case 4: hipLaunchKernelGGL(prune_gemm_kernel<4>, grid, block, 0, stream, a); break;
        case 5: hipLaunchKernelGGL(prune_gemm_kernel<5>, grid, block, 0, stream, a); break;
        case 6: hipLaunchKernelGGL(prune_gemm_kernel<6>, grid, block, 0, stream, a); break;
        case 7: hipLaunchKernelGGL(prune_gemm_kernel<7>, grid, block, 0, stream, a); break;
        case 8: hipLaunchKernelGGL(prune_gemm_kernel<8>, grid, block, 0, stream, a); break;
        case 9: hipLaunchKernelGGL(prune_gemm_kernel<9>, grid, block, 0, stream, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ---- fp64 MFMA issue-rate probe (roofline denominator check) -----------------
// 512-thread blocks keep the accumulators in VGPRs; with AGPR accumulators the same loop runs at
// about half the rate on gfx950.
__global__ __launch_bounds__(512) void mfma_probe_kernel(double* out, int iters) {
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
    hipLaunchKernelGGL(mfma_probe_kernel, dim3(blocks), dim3(512), 0, stream, d_out, iters);
    return hipGetLastError();
}

}  // namespace cafe
