// Zero extents of the likelihood panels (interval propagation up the tree, once per scorer call, right after K1).
//
// A likelihood column is exactly zero far from the observed sizes: a leaf with count x contributes P_leaf[s][x], which
// underflows to 0 once s is many standard deviations from x; an interior child contributes sum_c P[s][c] L[c], zero
// wherever row s of the matrix has no non-zero entry inside the support of L; a node's vector is the product of its
// children's factors, zero wherever one of them is.  K1 publishes the non-zero extents of the matrices (per column of a
// leaf branch's matrix, per block of 16 rows of an interior branch's); this kernel turns them, per node, category and
// panel column, into an interval [lo, hi] of panel rows outside which the column is EXACTLY zero -- conservative (it may
// contain zeros, it never excludes a non-zero) -- and reduces the intervals of a 128-column tile to their hull.  K2 then
// runs, for a (row tile, column tile) pair, only the K tiles inside the intersection of the matrix extent and the panel
// extent: every product it leaves out has an exact zero in it, so no bit of any result changes (the reference has no
// counterpart: it multiplies all (M+1)^2 entries for every family, matrix_cache.cpp:28-57).
#include "cafe_kernels.h"

namespace cafe {

__global__ __launch_bounds__(kBN) void node_extent_kernel(const ExtArgs a) {
    const ExtNode& nd = a.nodes[a.first + blockIdx.z];
    const int col = blockIdx.x * kBN + threadIdx.x;
    if (blockIdx.x * kBN >= nd.cols) return;               // (levels are launched as wide as their widest node)
    const int k = blockIdx.y;
    int lo = 0, hi = a.M;
    const int half = a.n_dev > 0 ? (a.n_dev - 1) / 2 : 0;
    for (int l = 0; l < nd.n_leaf; ++l) {
        const int x = nd.cnt[(int64_t)nd.leaf_row[l] * nd.cnt_ld + col];
        const int32_t* e = a.leaf_ext + (int64_t)(k * a.n_pairs_leaf + nd.leaf_pair[l]) * a.leaf_ext_blocks * 2;
        int flo = 0x7fffffff, fhi = -1;
        if (a.n_dev <= 0) {
            flo = e[2 * x]; fhi = e[2 * x + 1];
        } else {
            for (int i = 0; i < a.n_dev; ++i) {             // sum_i err[x][i] P[s][x - half + i]: the hull of the taps that count
                const int c = x - half + i;
                if (c < 0 || c > a.M || a.err[(int64_t)x * a.n_dev + i] == 0.0) continue;
                flo = min(flo, e[2 * c]); fhi = max(fhi, e[2 * c + 1]);
            }
        }
        lo = max(lo, flo); hi = min(hi, fhi);
    }
    for (int j = 0; j < nd.n_inner; ++j) {
        const int cc = nd.inner_map[j] ? nd.inner_map[j][col] : col;
        const int32_t* ce = nd.inner_colext[j] + ((int64_t)k * nd.inner_cols[j] + cc) * 2;
        const int clo = ce[0], chi = ce[1];
        int flo = 0x7fffffff, fhi = -1;
        if (chi >= clo) {
            // parent sizes s = 16 b + 1 .. 16 b + 16 (panel rows s) of block b have non-zero matrix entries for the child
            // sizes [e[2b], e[2b+1]]; size 0 only reaches child size 0 (P[0][c] = delta(c, 0))
            const int32_t* e = a.kext + (int64_t)(k * a.n_pairs_inner + nd.inner_pair[j]) * a.kext_blocks * 2;
            for (int b = 0; b < a.kext_blocks; ++b)
                if (e[2 * b + 1] >= clo && e[2 * b] <= chi) { flo = min(flo, 16 * b + 1); fhi = max(fhi, 16 * b + 16); }
            if (clo == 0) { flo = 0; fhi = max(fhi, 0); }
            fhi = min(fhi, a.M);
        }
        lo = max(lo, flo); hi = min(hi, fhi);
    }
    if (hi < lo) { lo = 0x7fffffff; hi = -1; }
    int32_t* out = nd.colext + ((int64_t)k * nd.cols + col) * 2;
    out[0] = lo; out[1] = hi;
    // hull over the 128 columns of the tile
    __shared__ int s_lo[2], s_hi[2];
    for (int off = 32; off > 0; off >>= 1) { lo = min(lo, __shfl_down(lo, off)); hi = max(hi, __shfl_down(hi, off)); }
    if ((threadIdx.x & 63) == 0) { s_lo[threadIdx.x >> 6] = lo; s_hi[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int32_t* t = nd.tileext + ((int64_t)k * (nd.cols / kBN) + blockIdx.x) * 2;
        t[0] = min(s_lo[0], s_lo[1]); t[1] = max(s_hi[0], s_hi[1]);
    }
}

hipError_t launch_node_extents(const ExtArgs& a, int max_col_tiles, int n_categories, hipStream_t stream) {
    if (a.count <= 0) return hipSuccess;
    (void)hipGetLastError();
    hipLaunchKernelGGL(node_extent_kernel, dim3(max_col_tiles, n_categories, a.count), dim3(kBN), 0, stream, a);
    return hipGetLastError();
}

}  // namespace cafe
