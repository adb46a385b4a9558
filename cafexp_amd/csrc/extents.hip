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
#include <algorithm>

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

constexpr int kPlanTail = 2;
// Tile lists of the K2 launches of a call (PlanLaunch, cafe_kernels.h): one wave per (XCD, launch), lane = workgroup.
// Cost of a tile = its K tiles + PlanLaunch::fixed.  With the full grid (64 workgroups per XCD) workgroups j and j + 32 share
// a CU, and the one dispatched first (j < 32) wins the MFMA arbitration: given equal lists it ALWAYS finishes first, 8 % of
// the launch earlier, and the CU runs one workgroup to the end (tools/gemm_timeline.py).  The planner therefore charges a
// tile `bias` percent less to j < 32 and as much more to j >= 32, so that the favoured workgroup takes more of the work.
// What a K tile costs workgroup w of an XCD, in percent of the mean: the workgroups of a CU do not share its matrix pipe
// evenly -- the one dispatched first wins the arbitration (with three per CU and equal lists they finish at 2 672 / 3 200 /
// 3 503 us of a 3 563 us launch: tools/k2_tile_costs.py) -- so the planner charges the favoured ones less and they take more
// of the work.  Full grids only (local indices j, j + 32, ... share a CU).
__device__ inline int plan_weight(const PlanLaunch& L, int nlb, int w) {
    if (nlb == 64 && L.bias) return w < 32 ? 100 - L.bias : 100 + L.bias;
    if (nlb == 96 || nlb == 128) return L.bias3[w >> 5];
    return 100;
}

// Tile t of XCD `xcd`'s list -- the concatenation of the lists of the launch's ops, each pair-major with the row tile
// fastest (the order prune_gemm.hip's decode assumes) -- as a plan entry: x = op << 24 | index in the op's own list,
// y = first K tile << 16 | K tiles.  s_first[o]: where op o's tiles start in the XCD's list (s_first[n_ops] = all).
__device__ inline int2 plan_tile_entry(const PlanLaunch& L, const int* s_first, int xcd, int t) {
    int op = 0;
    while (op + 1 < L.n_ops && t >= s_first[op + 1]) ++op;
    const GemmOp& o = L.ops[op];
    const int tl = t - s_first[op];
    const int nrt = o.n_row_tiles, nct = L.uniform_ld > 0 ? L.uniform_ld / kBN : o.n_col_tiles;
    const int row_tile = tl % nrt;
    const int pair = xcd + 8 * (tl / nrt);
    const int ct = pair % nct, cat = pair / nct;
    int lo = 0, hi = L.k_valid - 1, zlo = 0;
    if (L.aext) {
        const int b0 = row_tile * L.mi;
        const int32_t* e = L.aext + ((int64_t)o.slot[cat] * L.ext_blocks + b0) * 2;
        lo = 0x7fffffff; hi = -1;
        for (int b = 0; b < L.mi; ++b)
            if (b0 + b < L.ext_blocks) { lo = min(lo, e[2 * b]); hi = max(hi, e[2 * b + 1]); }
        if (o.bext) {
            // (rows of B outside its tile extent may never have been written: the assemble pass leaves them out, leaf_reduce.hip)
            const int32_t* be = o.bext + ((int64_t)cat * nct + ct) * 2;
            lo = max(lo, be[0]);
            hi = min(hi, be[1]);
            if (be[1] >= be[0]) zlo = be[0];
        }
        if (hi < lo) { lo = zlo; hi = zlo; }               // an all-zero tile still runs one K tile: the panel must receive its zeros
        hi = min(hi, L.k_valid - 1);
    }
    return make_int2((op << 24) | tl, ((lo / L.kb) << 16) | (hi / L.kb - lo / L.kb + 1));
}

// The last kPlanTail rounds are dealt as ONE batch, longest tile first, each to the workgroup with the least load at that
// moment (a workgroup may then end up with a tile more or less than its neighbours: lists have kPlanSlack spare entries).
// The rounds before them are dealt in chunks of kPlanChunk rounds, one wave per chunk (blockIdx.z), each balancing its own
// rounds from a load of zero -- a launch of a whole tree level has hundreds of rounds, and one wave dealing them one after
// the other took 1.8 ms in front of the first K2 launch of a config-4 call.
constexpr int kPlanChunk = 8;
__global__ __launch_bounds__(kPlanLanes) void tile_plan_kernel(const PlanLaunch* __restrict__ launches) {
    // (a copy in registers: the lists written below might alias the descriptor for all the compiler knows, and every field
    // would be re-read from memory at each use -- the serial tail loop then paid a memory round trip per item)
    const PlanLaunch L = launches[blockIdx.y];
    const int xcd = blockIdx.x, lane = threadIdx.x;         // lane = workgroup of the XCD (< blocks_per_xcd <= kPlanLanes)
    const int nlb = L.blocks_per_xcd;
    __shared__ int s_first[kMaxGroupOps + 1];
    if (lane == 0) {
        int acc = 0;
        for (int o = 0; o < L.n_ops; ++o) {
            s_first[o] = acc;
            const int nct = L.uniform_ld > 0 ? L.uniform_ld / kBN : L.ops[o].n_col_tiles;
            acc += ((L.n_categories * nct - xcd + 7) >> 3) * L.ops[o].n_row_tiles;
        }
        s_first[L.n_ops] = acc;
    }
    __syncthreads();
    const int n_tiles = s_first[L.n_ops];
    // (two workgroups per CU, 64 per XCD: j and j + 32 share a CU and the first-dispatched one runs `bias` percent faster)
    __shared__ int s_load[kPlanLanes], s_cost[kPlanLanes], s_who[kPlanLanes];
    __shared__ int t_cost[kPlanTail * kPlanLanes], t_x[kPlanTail * kPlanLanes], t_y[kPlanTail * kPlanLanes], t_by_rank[kPlanTail * kPlanLanes];
    s_load[lane] = 0;
    __syncthreads();
    const int head = max(0, (n_tiles + nlb - 1) / nlb - kPlanTail);      // full rounds dealt one by one
    const int n_chunks = max(1, (head + kPlanChunk - 1) / kPlanChunk), chunk = blockIdx.z;
    if (chunk >= n_chunks) return;                          // (block-uniform: the grid is as deep as the launch with most rounds)
    // the chunk's entries first, all at once (their extent look-ups are chains of dependent loads: one after the other they cost
    // a memory round trip per round), then the rounds from LDS
    __shared__ int c_x[kPlanChunk][kPlanLanes], c_y[kPlanChunk][kPlanLanes];
    const int r_begin = chunk * kPlanChunk, r_end = min(head, (chunk + 1) * kPlanChunk);
    const bool valid = lane < nlb;
#pragma unroll
    for (int i = 0; i < kPlanChunk; ++i)
        if (valid && r_begin + i < r_end) {
            const int2 en = plan_tile_entry(L, s_first, xcd, (r_begin + i) * nlb + lane);
            c_x[i][lane] = en.x; c_y[i][lane] = en.y;
        }
    for (int r = r_begin; r < r_end; ++r) {
        const int2 en = valid ? make_int2(c_x[r - r_begin][lane], c_y[r - r_begin][lane]) : make_int2(0, 0);
        const int cst = valid ? (en.y & 0xFFFF) + L.fixed : -1;
        const int mine = s_load[lane];
        s_cost[lane] = cst;
        __syncthreads();
        int crank = 0, lrank = 0;                          // longest tile first; least-loaded workgroup first (ties: by index)
        for (int k = 0; k < nlb; ++k) {
            const int ck = s_cost[k], lk = s_load[k];
            crank += (ck > cst || (ck == cst && k < lane)) ? 1 : 0;
            lrank += (lk < mine || (lk == mine && k < lane)) ? 1 : 0;
        }
        if (valid) s_who[lrank] = lane;
        __syncthreads();
        if (valid) {
            const int w = s_who[crank];
            L.plan[((int64_t)xcd * nlb + w) * L.rounds + r] = en;
            s_load[w] += cst * plan_weight(L, nlb, w);
        }
        __syncthreads();
    }
    if (chunk != n_chunks - 1) return;
    // ---- the rest as one batch, by the block that dealt the last chunk (on top of that chunk's loads)
    const int t0 = head * nlb, n_tail = n_tiles - t0;
    for (int i = lane; i < n_tail; i += kPlanLanes) {
        const int2 en = plan_tile_entry(L, s_first, xcd, t0 + i);
        t_cost[i] = (en.y & 0xFFFF) + L.fixed;
        t_x[i] = en.x;
        t_y[i] = en.y;
    }
    __syncthreads();
    for (int i = lane; i < n_tail; i += kPlanLanes) {
        const int c = t_cost[i];
        int rank = 0;
        for (int j = 0; j < n_tail; ++j) rank += (t_cost[j] > c || (t_cost[j] == c && j < i)) ? 1 : 0;
        t_by_rank[rank] = i;
    }
    __syncthreads();
    // each item, longest first, to the workgroup with the least load at that moment: a serial loop, run by the first wave alone
    // (shuffles, no block barrier), lane j looking after workgroups j and j + 64
    if (lane >= 64) return;
    const int wg[2] = {lane, lane + 64};
    int pos[2] = {head, head};
    unsigned load[2] = {(unsigned)s_load[wg[0]], (unsigned)s_load[wg[1]]};
    int wt[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) wt[h] = plan_weight(L, nlb, wg[h]);
    for (int q = 0; q < n_tail; ++q) {
        const int item = t_by_rank[q];
        unsigned key = 0xFFFFFFFFu;
#pragma unroll
        for (int h = 0; h < 2; ++h)
            if (wg[h] < nlb && pos[h] < L.rounds) key = min(key, (min(load[h], 0x00FFFFFFu) << 7) | (unsigned)wg[h]);
        for (int off = 32; off > 0; off >>= 1) key = min(key, (unsigned)__shfl_xor((int)key, off));
        if (key == 0xFFFFFFFFu) break;                      // (no list has room: cannot happen, the lists have kPlanSlack spare entries)
        const int w = (int)(key & 127u);
#pragma unroll
        for (int h = 0; h < 2; ++h)
            if (w == wg[h]) {
                L.plan[((int64_t)xcd * nlb + w) * L.rounds + pos[h]++] = make_int2(t_x[item], t_y[item]);
                load[h] += (unsigned)(t_cost[item] * wt[h]);
            }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
        if (wg[h] < nlb)
            for (; pos[h] < L.rounds; ++pos[h]) L.plan[((int64_t)xcd * nlb + wg[h]) * L.rounds + pos[h]] = make_int2(0, 0);
}

hipError_t launch_tile_plan(const PlanLaunch* d_launches, int n_launches, int max_rounds, hipStream_t stream) {
    if (n_launches <= 0) return hipSuccess;
    (void)hipGetLastError();
    const int chunks = std::max(1, (std::max(0, max_rounds - kPlanSlack - kPlanTail) + kPlanChunk - 1) / kPlanChunk);
    hipLaunchKernelGGL(tile_plan_kernel, dim3(8, n_launches, chunks), dim3(kPlanLanes), 0, stream, d_launches);
    return hipGetLastError();
}

}  // namespace cafe
