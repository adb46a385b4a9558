// K3 leaf_gather and K4 root_reduce (+ the final -lnL sum).
//
// K3 replaces the leaf branch of compute_node_probability (src/probability.cpp:179-199) together
// with the mat-vec the reference then runs on the leaf's one-hot vector (src/matrix_cache.cpp:28):
// P . e_x is column x of P, so a leaf child contributes factor[s][f] = P_leaf[s][x_f], or with an
// error model sum_i err[x_f][i] * P_leaf[s][x_f - (n_dev-1)/2 + i] (taps outside [0,M] dropped).
// The product over a node's leaf children is written (or multiplied) into the parent panel.
// Bound: L2/HBM traffic (one 8-byte gather per tap and one 8-byte store per panel element).
//
// K4 replaces the per-family tail of base_model::infer_family_likelihoods (src/base_model.cpp:89-106)
// and gamma_model::prune / infer_family_likelihoods (src/gamma_core.cpp:144-166, 201-225).
#include <algorithm>

#include "cafe_kernels.h"

namespace cafe {

constexpr int kGatherRows = 16;   // rows of the panel per block

// A K3 launch carries a GROUP of ops of one variant (the ready nodes of a tree level): blockIdx.z = op * K + category; the
// grid is as wide and as tall as the group's largest op, blocks beyond an op's own extent return at once.  The op
// descriptors live in a device array, read-only for the launch (constant address space: wave-uniform scalar loads).
typedef const __attribute__((address_space(4))) GatherArgs* gop_cptr_t;
#define CAFE_GATHER_OP()                                                                   \
    const int op_index = blockIdx.z / g.n_categories, cat = blockIdx.z - op_index * g.n_categories; \
    const gop_cptr_t a = (gop_cptr_t)(unsigned long long)g.ops + op_index;              \
    const int ld = g.uniform_ld > 0 ? g.uniform_ld : a->ld

__global__ __launch_bounds__(256) void leaf_gather_kernel(const GatherGroup g) {
    CAFE_GATHER_OP();
    const int f = blockIdx.x * 256 + threadIdx.x;           // column inside the chunk (ld is a multiple of 128)
    if (f >= ld) return;
    const int r0 = blockIdx.y * kGatherRows;
    if (r0 >= a->rows_store) return;
    double* __restrict__ dst = a->dst + (int64_t)cat * a->panel_kstride + f;
    const int ldp = g.pool.ld;
    const int n_leaf = a->n_leaf, n_src = a->n_src, rows = a->rows, rows_store = a->rows_store, row_off = a->row_off, mode = a->mode;

    int x[kMaxLeafPerOp];
    const double* P[kMaxLeafPerOp];
#pragma unroll
    for (int l = 0; l < kMaxLeafPerOp; ++l) {
        if (l < n_leaf) {
            x[l] = a->counts[(int64_t)a->taxon[l] * a->counts_ld + g.f0 + f];
            P[l] = g.pool.base + (int64_t)a->slot[l][cat] * g.pool.stride;
        } else {
            x[l] = 0;
            P[l] = g.pool.base;
        }
    }
    const int half = (g.n_dev - 1) / 2;

    for (int rr = 0; rr < kGatherRows; ++rr) {
        const int r = r0 + rr;
        if (r >= rows_store) break;
        double v = 0.0;
        if (r < rows) {
            const int64_t srow = (int64_t)(r + row_off) * ldp;
            v = 1.0;
#pragma unroll
            for (int l = 0; l < kMaxLeafPerOp; ++l) {
                if (l < n_leaf) {
                    double fac;
                    if (g.err == nullptr) {
                        fac = P[l][srow + x[l]];
                    } else {
                        fac = 0.0;
                        for (int i = 0; i < g.n_dev; ++i) {
                            const int c = x[l] - half + i;
                            if (c < 0 || c > g.max_family_size) continue;
                            fac += P[l][srow + c] * g.err[(int64_t)x[l] * g.n_dev + i];
                        }
                    }
                    v *= fac;
                }
            }
            for (int j = 0; j < n_src; ++j)             // transposed factors: [column][15 + row_off + panel row]
                v *= a->src[j][(int64_t)cat * a->kstride_src[j] + (int64_t)a->map[j][f] * a->ld_src[j] + 15 + row_off + r];
            if (mode) v *= dst[(int64_t)r * ld];
        }
        dst[(int64_t)r * ld] = v;
    }
}

// Fast path for the common shapes (one or two leaves; no error model or a 3-tap one; store or multiply): a thread
// owns two adjacent families (16-byte accesses, 256 threads = 4 KB contiguous per row) and kFastRows rows; the
// gathers of all its rows are issued before the first store.  Error-model taps outside [0, M] get weight 0 and a
// clamped column instead of a branch: fac = sum_i err[x][i] * P[s][x - half + i] in the reference's tap order
// (probability.cpp:187-196 builds the leaf vector, matrix_cache.cpp:28 multiplies it).
constexpr int kFastRows = 8;
template <int NLEAF, int NDEV, bool MUL, int NSRC>
__global__ __launch_bounds__(256) void leaf_gather_fast_kernel(const GatherGroup g) {
    CAFE_GATHER_OP();
    const int f = (blockIdx.x * 256 + threadIdx.x) * 2;
    if (f >= ld) return;
    const int r0 = blockIdx.y * kFastRows;
    const int rows = a->rows, rows_store = a->rows_store;
    if (r0 >= rows_store) return;
    if (a->tileext) {
        // rows outside the zero extents of the workgroup's (up to four) 128-column tiles are never staged by K2 (see the
        // assemble pass below): leave them as they are
        const int n_ct = ld / kBN, ct0 = (blockIdx.x * 512) / kBN;
        int lo = 0x7fffffff, hi = -1;
        for (int ct = ct0; ct < ct0 + 4 && ct < n_ct; ++ct) {
            const int32_t* te = a->tileext + ((int64_t)cat * n_ct + ct) * 2;
            const bool some = te[1] >= te[0];
            lo = min(lo, some ? te[0] : 0);
            hi = max(hi, some ? te[1] : 0);
        }
        if (r0 + kFastRows - 1 < (lo & ~15) || r0 > (hi | 15)) return;
    }
    const unsigned ldp = (unsigned)g.pool.ld;
    double* __restrict__ dst = a->dst + (int64_t)cat * a->panel_kstride + (int64_t)r0 * ld + f;
    constexpr int NL = NLEAF > 0 ? NLEAF : 1;               // (zero-length arrays are not allowed)
    unsigned o0[NL][NDEV], o1[NL][NDEV];
    double w0[NL][NDEV], w1[NL][NDEV];
    const double* P[NL];
    const double* S[NSRC > 0 ? NSRC : 1];                   // factor panels, row r0 of this category
    unsigned m0[NSRC > 0 ? NSRC : 1], m1[NSRC > 0 ? NSRC : 1];
    int64_t lds_[NSRC > 0 ? NSRC : 1];
#pragma unroll
    for (int j = 0; j < NSRC; ++j) {
        lds_[j] = a->ld_src[j];
        S[j] = a->src[j] + (int64_t)cat * a->kstride_src[j] + (int64_t)r0 * lds_[j];
        m0[j] = (unsigned)a->map[j][f];
        m1[j] = (unsigned)a->map[j][f + 1];
    }
    constexpr int half = (NDEV - 1) / 2;
    const int row_off = a->row_off;
#pragma unroll
    for (int l = 0; l < NLEAF; ++l) {
        const int32_t* cnt = a->counts + (int64_t)a->taxon[l] * a->counts_ld + g.f0 + f;
        const int x0 = cnt[0], x1 = cnt[1];
#pragma unroll
        for (int i = 0; i < NDEV; ++i) {
            const int c0 = x0 - half + i, c1 = x1 - half + i;
            const bool ok0 = c0 >= 0 && c0 <= g.max_family_size, ok1 = c1 >= 0 && c1 <= g.max_family_size;
            o0[l][i] = (unsigned)(ok0 ? c0 : x0);
            o1[l][i] = (unsigned)(ok1 ? c1 : x1);
            w0[l][i] = NDEV == 1 ? 1.0 : (ok0 ? g.err[(int64_t)x0 * NDEV + i] : 0.0);
            w1[l][i] = NDEV == 1 ? 1.0 : (ok1 ? g.err[(int64_t)x1 * NDEV + i] : 0.0);
        }
        P[l] = g.pool.base + (int64_t)a->slot[l][cat] * g.pool.stride + (int64_t)(r0 + row_off) * ldp;
    }
    double2 v[kFastRows];
#pragma unroll
    for (int rr = 0; rr < kFastRows; ++rr) {
        v[rr] = make_double2(0.0, 0.0);
        if (r0 + rr < rows) {
            double x = 1.0, y = 1.0;
#pragma unroll
            for (int l = 0; l < NLEAF; ++l) {
                const double* row = P[l] + (unsigned)rr * ldp;
                if (NDEV == 1) {
                    x *= row[o0[l][0]];
                    y *= row[o1[l][0]];
                } else {
                    double fx = 0.0, fy = 0.0;
#pragma unroll
                    for (int i = 0; i < NDEV; ++i) {
                        fx += row[o0[l][i]] * w0[l][i];
                        fy += row[o1[l][i]] * w1[l][i];
                    }
                    x *= fx;
                    y *= fy;
                }
            }
#pragma unroll
            for (int j = 0; j < NSRC; ++j) {
                const double* row = S[j] + (int64_t)rr * lds_[j];
                x *= row[m0[j]];
                y *= row[m1[j]];
            }
            if (MUL) {
                const double2 old = *reinterpret_cast<const double2*>(dst + (int64_t)rr * ld);
                x *= old.x;
                y *= old.y;
            }
            v[rr] = make_double2(x, y);
        }
    }
#pragma unroll
    for (int rr = 0; rr < kFastRows; ++rr)
        if (r0 + rr < rows_store) {
            // streamed once, read back by the next GEMM from HBM anyway: do not displace the matrices in L2
            __builtin_nontemporal_store(v[rr].x, dst + (int64_t)rr * ld);
            __builtin_nontemporal_store(v[rr].y, dst + (int64_t)rr * ld + 1);
        }
}

template <int NLEAF, int NDEV, int NSRC>
static void launch_fast3(const GatherGroup& g, int mode, dim3 grid, hipStream_t stream) {
    if (mode) hipLaunchKernelGGL((leaf_gather_fast_kernel<NLEAF, NDEV, true, NSRC>), grid, dim3(256), 0, stream, g);
    else hipLaunchKernelGGL((leaf_gather_fast_kernel<NLEAF, NDEV, false, NSRC>), grid, dim3(256), 0, stream, g);
}
template <int NLEAF, int NDEV>
static void launch_fast(const GatherGroup& g, int mode, dim3 grid, hipStream_t stream) {
    launch_fast3<NLEAF, NDEV, 0>(g, mode, grid, stream);
}

// Assemble pass: the parent's panel from the TRANSPOSED factor panels of its de-duplicated interior children (and its
// leaf children): P[r][c] = prod_j F_j[map_j[c]][r] * prod_leaves P_leaf[r][x_leaf(c)].  A factor column is contiguous
// along the rows, so a workgroup fetches, per mapped column, whole 128-byte lines (32 lanes x 16 bytes per column and
// instruction), multiplies the factors in that orientation, turns the 64 x 64 tile through LDS and writes the panel rows
// 512 contiguous bytes per wave instruction.  Bound: HBM (factors read once, panel written once); the row-major gather
// this replaces issued one 8-byte element per 128-byte line and was bound by the texture addresser (3.8 TB/s of L2
// traffic, profiles/r01).
constexpr int kAsmT = 64;                   // tile: 64 columns x 64 rows of the transposed index space
constexpr int kAsmS = kAsmT + 1;            // LDS row stride (doubles): column-wise reads of a row-major tile without conflicts
// LEAFT: the leaf children come as transposed copies of their matrices (GatherArgs::lt, no error model): they are
// multiplied in phase A like further factors -- the same products in the same order as the gather of phase B, so the panel
// has the same bits either way.
template <int NSRC, int NLEAF, int NDEV, bool MUL, bool LEAFT = false>
__global__ __launch_bounds__(256) void assemble_t_kernel(const GatherGroup g) {
    __shared__ double tile[kAsmT * kAsmS];
    CAFE_GATHER_OP();
    const int c0 = blockIdx.x * kAsmT;                       // first parent column of the tile (ld is a multiple of 128)
    if (c0 >= ld) return;
    const int i0 = blockIdx.y * kAsmT;                       // first transposed row index: panel row = index - toff
    const int row_off = a->row_off, rows = a->rows, rows_store = a->rows_store;
    const int toff = 15 + row_off;
    if (i0 - toff >= rows_store) return;
    const int tid = threadIdx.x;
    if (a->tileext) {
        // Rows outside the zero extent of the 128-column tile (extents.hip) are exact zeros that K2 never reads: it stages only
        // the K tiles (16 rows each) that meet the extent, or the tile at its first row when nothing does.  Leave them alone.
        const int32_t* te = a->tileext + ((int64_t)cat * (ld / kBN) + (c0 / kBN)) * 2;
        int lo = te[0], hi = te[1];
        if (hi < lo) { lo = 0; hi = 0; }
        if (i0 + kAsmT - 1 - toff < (lo & ~15) || i0 - toff > (hi | 15)) return;
    }
    // ---- phase A: 64 columns x 64 indices of each factor, 16 bytes per lane, 32 lanes per column
    {
        const int seg = tid & 31, cl = tid >> 5;             // column cl + 8 * i of the tile
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = cl + 8 * i;
            double2 v = make_double2(1.0, 1.0);
#pragma unroll
            for (int j = 0; j < NSRC; ++j) {
                const double2 f = *reinterpret_cast<const double2*>(a->src[j] + (int64_t)cat * a->kstride_src[j] +
                                                                    (int64_t)a->map[j][c0 + c] * a->ld_src[j] + i0 + 2 * seg);
                v.x *= f.x;
                v.y *= f.y;
            }
            if (LEAFT) {
#pragma unroll
                for (int l = 0; l < NLEAF; ++l) {
                    const int x = a->counts[(int64_t)a->taxon[l] * a->counts_ld + g.f0 + c0 + c];
                    const double2 f = *reinterpret_cast<const double2*>(a->lt[l] + (int64_t)cat * a->lt_kstride +
                                                                        (int64_t)x * a->ld_src[0] + i0 + 2 * seg);
                    v.x *= f.x;
                    v.y *= f.y;
                }
            }
            tile[c * kAsmS + 2 * seg] = v.x;
            tile[c * kAsmS + 2 * seg + 1] = v.y;
        }
    }
    __syncthreads();
    // ---- phase B: wave w writes panel rows (indices) i0 + 16 w .. + 15, lane = column
    const int lane = tid & 63, w = tid >> 6;
    const int f = c0 + lane;
    constexpr int NG = LEAFT ? 0 : NLEAF;                   // leaves still to gather on the way out
    constexpr int NL = NG > 0 ? NG : 1;
    constexpr int half = (NDEV - 1) / 2;
    unsigned o[NL][NDEV];
    double wgt[NL][NDEV];
    const double* P[NL];
#pragma unroll
    for (int l = 0; l < NG; ++l) {
        const int x = a->counts[(int64_t)a->taxon[l] * a->counts_ld + g.f0 + f];
#pragma unroll
        for (int i = 0; i < NDEV; ++i) {
            const int c = x - half + i;
            const bool ok = c >= 0 && c <= g.max_family_size;
            o[l][i] = (unsigned)(ok ? c : x);
            wgt[l][i] = NDEV == 1 ? 1.0 : (ok ? g.err[(int64_t)x * NDEV + i] : 0.0);
        }
        P[l] = g.pool.base + (int64_t)a->slot[l][cat] * g.pool.stride;
    }
    double* __restrict__ dst = a->dst + (int64_t)cat * a->panel_kstride + f;
#pragma unroll 4
    for (int rr = 0; rr < 16; ++rr) {
        const int r = i0 + 16 * w + rr - toff;               // panel row
        if (r < 0 || r >= rows_store) continue;
        double v = 0.0;
        if (r < rows) {
            v = tile[lane * kAsmS + 16 * w + rr];
#pragma unroll
            for (int l = 0; l < NG; ++l) {
                const double* row = P[l] + (int64_t)(r + row_off) * g.pool.ld;
                if (NDEV == 1) {
                    v *= row[o[l][0]];
                } else {
                    double fx = 0.0;
#pragma unroll
                    for (int i = 0; i < NDEV; ++i) fx += row[o[l][i]] * wgt[l][i];
                    v *= fx;
                }
            }
            if (MUL) v *= dst[(int64_t)r * ld];
        }
        __builtin_nontemporal_store(v, dst + (int64_t)r * ld);
    }
}

template <int NSRC, int NLEAF, int NDEV, bool LEAFT = false>
static void launch_asm3(const GatherGroup& g, int mode, dim3 grid, hipStream_t stream) {
    if (mode) hipLaunchKernelGGL((assemble_t_kernel<NSRC, NLEAF, NDEV, true, LEAFT>), grid, dim3(256), 0, stream, g);
    else hipLaunchKernelGGL((assemble_t_kernel<NSRC, NLEAF, NDEV, false, LEAFT>), grid, dim3(256), 0, stream, g);
}
template <int NSRC>
static void launch_asm(const GatherGroup& g, int n_leaf, int mode, int ndev, bool leaf_t, dim3 grid, hipStream_t stream) {
    if (leaf_t && ndev == 1 && n_leaf == 1) { launch_asm3<NSRC, 1, 1, true>(g, mode, grid, stream); return; }
    if (leaf_t && ndev == 1 && n_leaf == 2) { launch_asm3<NSRC, 2, 1, true>(g, mode, grid, stream); return; }
    if (n_leaf == 0) launch_asm3<NSRC, 0, 1>(g, mode, grid, stream);
    else if (n_leaf == 1) { if (ndev == 1) launch_asm3<NSRC, 1, 1>(g, mode, grid, stream); else launch_asm3<NSRC, 1, 3>(g, mode, grid, stream); }
    else { if (ndev == 1) launch_asm3<NSRC, 2, 1>(g, mode, grid, stream); else launch_asm3<NSRC, 2, 3>(g, mode, grid, stream); }
}

// dst[pair][category][x][15 + s] = P[s][x]: a 64 x 64 tile through LDS, 512-byte rows in, 512-byte rows out.  Entries the
// kernel does not write (the first 15 of a count's run, those behind the matrix) keep the zeros of cafe_create.
__global__ __launch_bounds__(256) void leaf_transpose_kernel(const LeafTArgs a) {
    __shared__ double tile[kAsmT * kAsmS];
    const int x0 = blockIdx.x * kAsmT, s0 = blockIdx.y * kAsmT;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int pi = blockIdx.z % a.n_list, cat = blockIdx.z / a.n_list;
    const double* __restrict__ P = a.pool.base + ((int64_t)cat * a.pool_pairs + a.pairs[pi]) * a.pool.stride;
    double* __restrict__ D = a.dst + (int64_t)pi * a.pair_stride + (int64_t)cat * a.kstride;
    const int n = a.pool.n;
    // (all 16 loads of a thread in flight before the first LDS store, all 16 LDS reads before the first global store)
    double v[16];
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
        const int s = s0 + 16 * w + rr, x = x0 + lane;
        v[rr] = (s < n && x < a.n_x) ? P[(int64_t)s * a.pool.ld + x] : 0.0;
    }
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) tile[(16 * w + rr) * kAsmS + lane] = v[rr];
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) v[rr] = tile[lane * kAsmS + 16 * w + rr];
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
        const int x = x0 + 16 * w + rr, s = s0 + lane;
        if (x < a.n_x && s < n) D[(int64_t)x * a.ld_t + 15 + s] = v[rr];
    }
}

hipError_t launch_leaf_transpose(const LeafTArgs& a, int n_pairs, int n_categories, hipStream_t stream) {
    if (n_pairs <= 0) return hipSuccess;
    (void)hipGetLastError();
    if ((unsigned)(n_pairs * n_categories) > 65535u || a.n_list != n_pairs) return hipErrorInvalidValue;
    dim3 grid((a.n_x + kAsmT - 1) / kAsmT, (a.pool.n + kAsmT - 1) / kAsmT, n_pairs * n_categories);
    hipLaunchKernelGGL(leaf_transpose_kernel, grid, dim3(256), 0, stream, a);
    return hipGetLastError();
}

// h_ops: the group's descriptors on the host (grid extents and the variant: every op of a group has the same n_leaf, n_src
// and mode -- the schedule groups by them)
hipError_t launch_leaf_gather_group(const GatherGroup& g, const GatherArgs* h_ops, hipStream_t stream) {
    (void)hipGetLastError();
    if (g.n_ops < 1 || !g.ops || !h_ops) return hipErrorInvalidValue;
    const int n_leaf = h_ops[0].n_leaf, n_src = h_ops[0].n_src, mode = h_ops[0].mode;
    if (n_leaf == 0 && n_src == 0) return hipErrorInvalidValue;
    int max_ld = 0, max_rows_store = 0, max_row_off = 0;
    for (int i = 0; i < g.n_ops; ++i) {
        if (h_ops[i].n_leaf != n_leaf || h_ops[i].n_src != n_src || h_ops[i].mode != mode) return hipErrorInvalidValue;
        max_ld = std::max(max_ld, g.uniform_ld > 0 ? g.uniform_ld : h_ops[i].ld);
        max_rows_store = std::max(max_rows_store, h_ops[i].rows_store);
        max_row_off = std::max(max_row_off, h_ops[i].row_off);
    }
    const unsigned z = (unsigned)(g.n_ops * g.n_categories);
    if (z > 65535u) return hipErrorInvalidValue;
    const int ndev = g.err == nullptr ? 1 : g.n_dev;
    if ((ndev == 1 || ndev == 3) && n_leaf <= 2 && n_src >= 1 && n_src <= 2) {
        dim3 grid(max_ld / kAsmT, (max_rows_store + 15 + max_row_off + kAsmT - 1) / kAsmT, z);
        bool leaf_t = n_leaf > 0 && g.leaf_t;              // every leaf of every op has its transposed copy, and this call filled them
        for (int i = 0; i < g.n_ops && leaf_t; ++i)
            for (int l = 0; l < n_leaf; ++l) leaf_t = leaf_t && h_ops[i].lt[l] != nullptr;
        if (n_src == 1) launch_asm<1>(g, n_leaf, mode, ndev, leaf_t, grid, stream);
        else launch_asm<2>(g, n_leaf, mode, ndev, leaf_t, grid, stream);
        return hipGetLastError();
    }
    if ((ndev == 1 || ndev == 3) && n_leaf <= 2 && n_src == 0) {
        dim3 grid((max_ld / 2 + 255) / 256, (max_rows_store + kFastRows - 1) / kFastRows, z);
        if (n_leaf == 1) { if (ndev == 1) launch_fast<1, 1>(g, mode, grid, stream); else launch_fast<1, 3>(g, mode, grid, stream); }
        else { if (ndev == 1) launch_fast<2, 1>(g, mode, grid, stream); else launch_fast<2, 3>(g, mode, grid, stream); }
        return hipGetLastError();
    }
    dim3 grid((max_ld + 255) / 256, (max_rows_store + kGatherRows - 1) / kGatherRows, z), block(256);
    hipLaunchKernelGGL(leaf_gather_kernel, grid, block, 0, stream, g);
    return hipGetLastError();
}

// kRedFam families x kRedSeg row segments per workgroup: thread (family, segment) walks the rows j = segment,
// segment + kRedSeg, ... of its family's root column (a wave instruction reads kRedFam consecutive families of 64 / kRedFam
// rows: full 128-byte lines), the segments are then combined in LDS.  Maxima do not depend on the order they are combined
// in; the category sum is only tested against zero (all terms are non-negative).  NaN follows the reference's scan
// (std::max_element / the `>` scan of gamma_core.cpp:158): a NaN at j = 0 is returned, a NaN elsewhere never wins a
// comparison.  (One thread per family, 750 x 8 dependent strided loads, took 2.5 ms whatever the number of families.)
constexpr int kRedFam = 16, kRedSeg = 16;
__device__ inline double combine_max(double* sh, int fam, int seg, double v) {
    sh[seg * kRedFam + fam] = v;
    __syncthreads();
    double best = sh[fam];
    if (seg == 0)
        for (int s2 = 1; s2 < kRedSeg; ++s2) { const double p = sh[s2 * kRedFam + fam]; if (p > best) best = p; }
    __syncthreads();
    return best;
}
__device__ inline double combine_sum(double* sh, int fam, int seg, double v) {
    sh[seg * kRedFam + fam] = v;
    __syncthreads();
    double t = 0.0;
    if (seg == 0)
        for (int s2 = 0; s2 < kRedSeg; ++s2) t += sh[s2 * kRedFam + fam];
    __syncthreads();
    return t;
}

__global__ __launch_bounds__(256) void root_reduce_kernel(const ReduceArgs a) {
    __shared__ double sh[kRedFam * kRedSeg];
    const int fam = threadIdx.x % kRedFam, seg = threadIdx.x / kRedFam;
    const int64_t fl = (int64_t)blockIdx.x * kRedFam + fam;
    const bool live = fl < a.nf;                            // (every thread reaches the barriers)
    const int64_t f = a.f0 + fl;
    const double ninf = -__builtin_huge_val();
    if (a.model == 2) {
        // p-value path: the observed / simulated "max likelihood" is max_j L_root[j], no prior
        // (probability.cpp:313, :399)
        const double* col = a.root + fl;
        double best = ninf;
        if (live)
            for (int j = seg; j < a.R; j += kRedSeg) {
                const double L = col[(int64_t)j * a.ld];
                if (j == 0 || L > best) best = L;
            }
        best = combine_max(sh, fam, seg, best);
        if (live && seg == 0) { a.fam_out[f] = best; a.failed[f] = 0; }
        return;
    }
    if (a.model == 0) {
        // lnL_f = max_j( log L_j + log prior_j ), first maximum like std::max_element (base_model.cpp:94-101)
        const double* col = a.root + fl;
        double best = ninf;
        if (live)
            for (int j = seg; j < a.R; j += kRedSeg) {
                const double full = log(col[(int64_t)j * a.ld]) + a.log_prior[j];
                if (j == 0 || full > best) best = full;
            }
        best = combine_max(sh, fam, seg, best);
        if (live && seg == 0) { a.fam_out[f] = best; a.failed[f] = 0; }
        return;
    }
    double lik = 0.0;
    int fail = 0;
    for (int k = 0; k < a.K; ++k) {
        const double* col = a.root + (int64_t)k * a.panel_kstride + fl;
        double sum = 0.0, best = ninf;
        if (live)
            for (int j = seg; j < a.R; j += kRedSeg) {
                const double L = col[(int64_t)j * a.ld];
                sum += L;
                const double full = L * a.prior[j];
                if (j == 0 || full > best) best = full;
            }
        sum = combine_sum(sh, fam, seg, sum);
        best = combine_max(sh, fam, seg, best);
        if (live && seg == 0) {
            if (sum == 0.0) fail = 1;                      // "saturation", gamma_core.cpp:152
            const double cl = best * a.cat_probs[k];        // gamma_core.cpp:162
            a.cat_out[f * a.K + k] = cl;
            lik += cl;                                      // gamma_core.cpp:207
        }
    }
    if (live && seg == 0) {
        a.fam_lik[f] = lik;
        a.fam_out[f] = log(lik);
        a.failed[f] = fail;
    }
}

hipError_t launch_root_reduce(const ReduceArgs& a, hipStream_t stream) {
    if (a.nf <= 0) return hipSuccess;
    dim3 grid((unsigned)((a.nf + kRedFam - 1) / kRedFam)), block(256);
    (void)hipGetLastError();
    hipLaunchKernelGGL(root_reduce_kernel, grid, block, 0, stream, a);
    return hipGetLastError();
}

// Deterministic two-stage sum: fixed block partials, then one block folds them in index order.
__device__ inline double block_sum(double v, double* sh) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) sh[w] = v;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0)
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh[i];
    __syncthreads();
    return t;
}

// ... and the block that finishes last folds the partials, in index order whichever block that is (one launch instead of two:
// in the small-matrix regime a call is the sum of its kernels' latencies).  scratch: [2 * blocks] partials, then one
// unsigned counter (zero between calls: the folding block resets it).
// out2: optionally a second copy, in host memory the device can write (the scorer's return value without a copy engine hop)
__global__ __launch_bounds__(256) void partial_sum_kernel(const double* __restrict__ fam_out, const double* __restrict__ w,
                                                          const int32_t* __restrict__ failed, int64_t n, double* scratch, double* out, double* out2) {
    __shared__ double sh[4];
    __shared__ bool last;
    double s = 0.0, bad = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        s += w[i] * fam_out[i];
        bad += failed[i] ? w[i] : 0.0;
    }
    const double ts = block_sum(s, sh);
    const double tb = block_sum(bad, sh);
    unsigned* counter = reinterpret_cast<unsigned*>(scratch + 2 * gridDim.x);
    if (threadIdx.x == 0) {
        scratch[2 * blockIdx.x] = ts;
        scratch[2 * blockIdx.x + 1] = tb;
        __threadfence();                                     // the partials are visible before the ticket is taken
        last = atomicAdd(counter, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!last || threadIdx.x != 0) return;
    __threadfence();
    double fs = 0.0, fb = 0.0;
    const volatile double* sc = scratch;                     // (written by other blocks of this launch)
    for (unsigned i = 0; i < gridDim.x; ++i) {
        fs += sc[2 * i];
        fb += sc[2 * i + 1];
    }
    out[0] = fs;
    out[1] = fb;
    if (out2) { out2[0] = fs; out2[1] = fb; }
    *counter = 0u;
}

hipError_t launch_final_sum(const double* fam_out, const double* weights, const int32_t* failed, int64_t n,
                            double* scratch, int n_scratch, double* out, double* out_host, hipStream_t stream) {
    int blocks = (int)((n + 255) / 256);
    if (blocks > n_scratch) blocks = n_scratch;
    if (blocks < 1) blocks = 1;
    (void)hipGetLastError();
    hipLaunchKernelGGL(partial_sum_kernel, dim3(blocks), dim3(256), 0, stream, fam_out, weights, failed, n, scratch, out, out_host);
    return hipGetLastError();
}

}  // namespace cafe
