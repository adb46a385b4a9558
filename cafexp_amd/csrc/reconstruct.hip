// Pupko joint ancestral reconstruction and Viterbi branch probabilities.
//
// Replaces, per family and gamma category:
//   reconstruct_leaf_node / reconstruct_internal_node / reconstruct_root_node / reconstruct_gene_family
//                                                       src/gene_family_reconstructor.cpp:13-165
//   compute_viterbi_sum                                 src/gene_family_reconstructor.cpp:361-400
//
// The reference keeps, per node v, L_v[i] = max_j P_v[i][j] * prod_{children c} L_c[j] and C_v[i] = the
// arg max (i = size of v's parent, j = size of v), then walks down from the root's choice.  Only one
// entry of every C_v is ever read (i = the state chosen for the parent), so the device keeps the product
// panels B_v[j][family] = prod_c L_c[j] instead of the argmax tables:
//   forward   (K5 maxprod)  B_parent (op)= max_j P_v[.][j] * B_v[j]     a (max,x) "GEMM": no MFMA form, fp64 VALU,
//                                                                        2 instructions per (i, j, family)
//   leaves    (recon_leaf)  B_parent (op)= P_leaf[j][x_f], 0 at j = 0   (gene_family_reconstructor.cpp:28-32: L[0] stays 0)
//   root      (root_select) state = first arg max_{j=1..min(M,R)} B_root[j] * prior(j)        (:47-62)
//   backward  (backtrack)   state_v = first arg max_j B_v[j] * P_v[state_parent][j]            (:96-111, :147-152)
// "first arg max" = the reference's strict `val > max_val` scan from max_val = -1.
// Families are the fastest axis of every panel, as in the scorer path; the interior matrices are the
// k-major ones K1 already builds for K2 (Pt[j][i-1] = P[i][j]), read through the scalar unit.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <memory>
#include <vector>

#include "cafe_ctx.h"

namespace cafe {

namespace {

constexpr int kTI = 16;                 // parent sizes per wave in K5
constexpr int kRowsPerBlock = 4 * kTI;

// Row groups of a product panel: group 0 = rows 0..16, group g >= 1 = rows 16g+1 .. 16g+16 (what one K5 wave writes).  Whoever
// writes a panel -- the store of the first child, the multiplies of the others -- records per (64-family tile, row group)
// whether any value it wrote is non-zero; after the last child the flags describe the finished panel, and the K5 that reads it
// walks only from the first to the last flagged group: every product it leaves out has an exact zero in it, and
// max(acc, 0) = acc (all terms are >= 0), so the result has the same bits.
__device__ inline int group_first_row(int g) { return g == 0 ? 0 : 16 * g + 1; }

// B_dst[j][f] (op)= (j >= 1) ? P[j][x_f] : 0 for j = 0..M; one thread per family, one row group per block row
template <bool MUL>
__global__ __launch_bounds__(256) void recon_leaf_kernel(const double* __restrict__ P, int ldp, const int32_t* __restrict__ counts,
                                                         double* __restrict__ dst, int64_t ld, int M, int32_t* __restrict__ flags, int n_groups) {
    const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (f >= ld) return;                                         // (ld is a multiple of 128: whole waves)
    const int x = counts[f];
    const int g = blockIdx.y, j0 = group_first_row(g), j1 = min(M, 16 * g + 16);
    bool nz = false;
    for (int j = j0; j <= j1; ++j) {
        double v = j >= 1 ? P[(int64_t)j * ldp + x] : 0.0;
        double* o = dst + (int64_t)j * ld + f;
        if (MUL) v *= *o;
        *o = v;
        nz = nz || v != 0.0;
    }
    const bool any = __any(nz);
    if ((threadIdx.x & 63) == 0) flags[(f >> 6) * n_groups + g] = any ? 1 : 0;
}

// v_max_f64 without the canonicalising self-max the compiler adds in front of llvm.maxnum on a loop-carried value
// (all operands here are products of finite non-negative numbers)
__device__ inline double vmax(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// K5: dst[i][f] (op)= max_j Pt[j][i-1] * B[j][f], i = 1..M; dst[0][f] (op)= B[0][f] (P[0][j] = delta(j,0)).
// A wave owns 64 families (lanes) x kTI parent sizes (accumulators); the matrix entries of a step are wave-uniform
// and come through scalar loads, so a step is one coalesced 512 B load + 2*kTI fp64 VALU instructions.  The four
// waves of a block share the family columns (L1 hits) and split 64 parent sizes.  Steps run in groups of four
// with the next group's panel values already in flight; rows M+1.. of the group padding multiply zero matrix rows
// (the k-major matrix is zero beyond row M, the panel workspace beyond row M is zeroed once).
// ext (or nullptr): K1's non-zero extents of this matrix -- per block of 16 parent sizes the first / last child size j with a
// non-zero entry.  A wave's 16 parent sizes are exactly one such block, so its j loop runs over that range only: every
// product left out is b * 0 = 0, and max(acc, 0) = acc (acc starts at 0, all terms are >= 0): the same bits.
// bflags: the row-group flags of B (see above); dflags: those of dst, written here.
template <bool MUL>
__global__ __launch_bounds__(256) void maxprod_kernel(const double* __restrict__ Pt, int ldp, const double* __restrict__ B,
                                                      double* __restrict__ dst, int64_t ld, int M, int kc_rows, const int32_t* __restrict__ ext,
                                                      const int32_t* __restrict__ bflags, int32_t* __restrict__ dflags, int n_groups) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // XCD-aware order: workgroups b, b+8, ... share an XCD; consecutive ones there take the row groups of ONE
    // 64-family column tile, so the panel column is fetched into that L2 once and reused by all of them
    const int n_rg = (M + kRowsPerBlock - 1) / kRowsPerBlock;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int ct = xcd + 8 * (slot / n_rg), rg = slot % n_rg;
    if ((int64_t)ct * 64 >= ld) return;
    const int64_t f = (int64_t)ct * 64 + lane;                  // ld is a multiple of 128: always in range
    const int i0 = __builtin_amdgcn_readfirstlane((rg * 4 + wave) * kTI);       // Pt columns i0 .. i0+kTI-1 <-> sizes i0+1 ..
    if (i0 >= M) return;
    double acc[kTI];
#pragma unroll
    for (int t = 0; t < kTI; ++t) acc[t] = 0.0;                  // all products are >= 0: same result as the reference's -1 start
    const double* bp = B + f;
    const double* pp = Pt + i0;
    int groups = (M + 4) / 4;                                    // rows 0 .. 4*groups-1 cover 0..M
    int g0 = 0;
    {
        int lo = 0, hi = M;
        if (ext) { lo = ext[2 * (i0 / kTI)]; hi = ext[2 * (i0 / kTI) + 1]; }     // hi < lo: an all-zero block of the matrix, acc stays 0
        if (bflags) {                                            // ... and the rows of B's tile between its first and last flagged group
            const unsigned long long m = __ballot(lane < n_groups && bflags[(int64_t)ct * n_groups + lane] != 0);     // n_groups <= 64
            if (m == 0) { lo = 1; hi = 0; }
            else {
                lo = max(lo, group_first_row(__builtin_ctzll(m)));
                hi = min(hi, 16 * (63 - __builtin_clzll(m)) + 16);
            }
        }
        if (hi < lo) { g0 = 0; groups = 0; }
        else { g0 = __builtin_amdgcn_readfirstlane(lo / 4); groups = __builtin_amdgcn_readfirstlane(min(groups, hi / 4 + 1)); }
    }
    double bn[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) bn[u] = bp[(int64_t)(4 * g0 + u) * ld];
    double pn[kTI];                                              // the matrix row of the NEXT step, already requested
    {
        const double* prow0 = pp + (int64_t)min(4 * g0, kc_rows - 1) * ldp;
#pragma unroll
        for (int t = 0; t < kTI; ++t) pn[t] = prow0[t];
    }
    for (int g = g0; g < groups; ++g) {
        double b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) b[u] = bn[u];
        if (g + 1 < groups) {
#pragma unroll
            for (int u = 0; u < 4; ++u) bn[u] = bp[(int64_t)(4 * g + 4 + u) * ld];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            double pc[kTI];
#pragma unroll
            for (int t = 0; t < kTI; ++t) pc[t] = pn[t];
            const int next_row = min(4 * g + u + 1, kc_rows - 1);       // stay inside this matrix (rows M+1 .. kc_rows-1 are zero)
            const double* prow = pp + (int64_t)next_row * ldp;          // uniform address: s_load
#pragma unroll
            for (int t = 0; t < kTI; ++t) pn[t] = prow[t];
#pragma unroll
            for (int t = 0; t < kTI; ++t) acc[t] = vmax(acc[t], b[u] * pc[t]);
        }
    }
    bool nz = false;
#pragma unroll
    for (int t = 0; t < kTI; ++t) {
        const int i = i0 + 1 + t;
        if (i <= M) {
            double* o = dst + (int64_t)i * ld + f;
            const double v = MUL ? *o * acc[t] : acc[t];
            *o = v;
            nz = nz || v != 0.0;
        }
    }
    if (i0 == 0) {
        double* o = dst + f;
        const double v = MUL ? *o * bp[0] : bp[0];
        *o = v;
        nz = nz || v != 0.0;
    }
    const bool any = __any(nz);
    if (lane == 0) dflags[(int64_t)ct * n_groups + i0 / kTI] = any ? 1 : 0;
}

// root: C[0] = first arg max over j = 1..jmax of B[j] * prior(j), scanning from max_val = -1
__global__ __launch_bounds__(256) void root_select_kernel(const double* __restrict__ B, int64_t ld, const double* __restrict__ prior,
                                                          int jmax, int32_t* __restrict__ state) {
    const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (f >= ld) return;
    double best = -1.0;
    int arg = 0;
    for (int j = 1; j <= jmax; ++j) {
        const double val = B[(int64_t)j * ld + f] * prior[j];
        if (val > best) { best = val; arg = j; }
    }
    state[f] = arg;
}

// interior node: state = first arg max_j B[j] * P[i][j], i = the parent's state; P[i][j] = Pt[j][i-1], row 0 = e_0.
// 64 families per block; the j range is cut into four consecutive segments scanned by four threads per family and
// combined in segment order with the same strict comparison, which keeps the reference's "first maximum".
// ext (or nullptr): K1's extents of the matrix -- row i is exactly zero outside [lo, hi] of its block of 16 parent sizes, so
// the scan covers that range only: outside it every value is 0, which the reference's scan takes at j = 0 (0 > -1) and never
// again (0 > 0 is false) -- hence the start (0, j = 0) when the range begins behind j = 0.
__global__ __launch_bounds__(256) void backtrack_kernel(const double* __restrict__ Pt, int ldp, const double* __restrict__ B, int64_t ld,
                                                        int M, const int32_t* __restrict__ parent_state, int32_t* __restrict__ state,
                                                        const int32_t* __restrict__ ext) {
    __shared__ double s_best[4][64];
    __shared__ int s_arg[4][64];
    const int fl = threadIdx.x & 63, seg = threadIdx.x >> 6;
    const int64_t f = (int64_t)blockIdx.x * 64 + fl;          // ld is a multiple of 128: always in range
    const int i = parent_state[f];
    int lo = 0, hi = M;
    if (i == 0) hi = 0;                                        // row 0 = e_0
    else if (ext) { lo = ext[2 * ((i - 1) >> 4)]; hi = min(M, ext[2 * ((i - 1) >> 4) + 1]); }
    const int per = (max(0, hi - lo + 1) + 3) / 4;
    const int j0 = lo + seg * per, j1 = min(hi + 1, j0 + per);
    double best = -1.0;
    int arg = 0;
    for (int j = j0; j < j1; ++j) {
        const double p = i == 0 ? (j == 0 ? 1.0 : 0.0) : Pt[(int64_t)j * ldp + (i - 1)];
        const double val = B[(int64_t)j * ld + f] * p;
        if (val > best) { best = val; arg = j; }
    }
    s_best[seg][fl] = best;
    s_arg[seg][fl] = arg;
    __syncthreads();
    if (seg == 0) {
        best = lo > 0 ? 0.0 : -1.0;
        arg = 0;
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2)
            if (s_best[s2][fl] > best) { best = s_best[s2][fl]; arg = s_arg[s2][fl]; }
        state[f] = arg;
    }
}

// out[f][v] = state[v][f]: the host wants a family's nodes side by side
__global__ __launch_bounds__(256) void state_transpose_kernel(const int32_t* __restrict__ state, int n, int64_t cols, int64_t ld, int32_t* __restrict__ out) {
    const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (f >= ld) return;
    const int v0 = blockIdx.y * 16;
#pragma unroll 4
    for (int v = v0; v < min(n, v0 + 16); ++v) out[f * n + v] = state[(int64_t)v * cols + f];
}

// compute_viterbi_sum: one thread per (family, node); NaN = "invalid" (root, or parent size == child size)
__global__ __launch_bounds__(256) void viterbi_kernel(const int32_t* __restrict__ sizes, int64_t F, int n_nodes, const int32_t* __restrict__ parent,
                                                      const int32_t* __restrict__ is_leaf, const int32_t* __restrict__ slot,
                                                      MatrixPool pool, MatrixPool kpool, int M, double* __restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= F * n_nodes) return;
    const int64_t f = idx / n_nodes;
    const int v = (int)(idx - f * n_nodes);
    const int par = parent[v];
    double result = __builtin_nan("");
    if (par >= 0) {
        const int ps = sizes[f * n_nodes + par], cs = sizes[idx];
        if (ps != cs) {
            // row ps of the branch's matrix, in whichever layout the branch uses
            const double* base;
            int64_t step;
            if (is_leaf[v]) { base = pool.base + (int64_t)slot[v] * pool.stride + (int64_t)ps * pool.ld; step = 1; }
            else { base = kpool.base + (int64_t)slot[v] * kpool.stride + (ps - 1); step = kpool.ld; }
            auto P = [&](int m) -> double {
                if (ps == 0) return m == 0 ? 1.0 : 0.0;
                return base[(int64_t)m * step];
            };
            const double calc = P(cs);
            result = 0.0;
            for (int m = 0; m < M; ++m) {                      // m < max_family_size, as in the reference
                const double pm = P(m);
                if (pm == calc) result += pm / 2.0;
                else if (pm < calc) result += pm;
            }
        }
    }
    out[idx] = result;
}

struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
};

}  // namespace

// (The reference's "Zero matrix found" throw, gene_family_reconstructor.cpp:87, cannot fire: matrix::is_zero tests the
// largest entry, and precalculate_matrices always sets P[0][0] = 1, matrix_cache.cpp:148 -- saturated matrices included.
// A saturated branch reconstructs through its all-zero rows exactly as the reference does: every product is 0 and the
// first argmax wins.)
int reconstruct_impl(cafe_ctx* c, const cafe_params* pr, const float* root_prior, int32_t* states) {
    if (!pr || !pr->lambdas || !root_prior || !states) { set_err(c, "cafe_reconstruct: lambdas, root_prior and states are required"); return CAFE_ERR_ARGUMENT; }
    const bool gamma = pr->model == CAFE_MODEL_GAMMA;
    const int K = gamma ? pr->n_categories : 1;
    if (gamma && (K < 1 || K > c->Kmax || !pr->multipliers)) { set_err(c, "cafe_reconstruct: gamma model needs 1..%d categories with multipliers", c->Kmax); return CAFE_ERR_ARGUMENT; }
    if (!lambdas_valid(c, pr->lambdas)) { set_err(c, "cafe_reconstruct: invalid lambda"); return CAFE_ERR_ARGUMENT; }
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    if (c->upload_pending) { HIP_TRY(c, hipEventSynchronize(c->ev_upload)); c->upload_pending = false; }
    c->have_results = false;
    c->last_stream = s;
    c->K_last = K;
    { const int rc = prepare_matrices(c, pr->lambdas, gamma ? pr->multipliers : nullptr, K, s); if (rc != CAFE_OK) return rc; }
    HIP_TRY(c, hipEventRecord(c->ev_upload, s));
    c->upload_pending = true;

    const int M = c->M, jmax = std::min(c->M, c->R), n = c->n_nodes;
    std::vector<int> interior, bidx(n, -1);
    for (int v = 0; v < n; ++v) if (c->leaf_taxon[v] < 0) { bidx[v] = (int)interior.size(); interior.push_back(v); }
    const int nI = (int)interior.size();
    const int rows = round_up(M + 1, 4);                    // K5 walks the child sizes in groups of four

    // workspace: one product panel per interior node + the state table, sized to the free memory
    size_t free_b = 0, total_b = 0;
    HIP_TRY(c, hipMemGetInfo(&free_b, &total_b));
    // (panels, the state table both ways round, the row-group flags: one int per 64 columns and 16 rows)
    const size_t per_col = (size_t)nI * rows * sizeof(double) + (size_t)2 * n * sizeof(int32_t) + (size_t)nI * ((M + 15) / 16) * sizeof(int32_t) / 64 + 1;
    const size_t budget = c->workspace_limit ? c->workspace_limit : (size_t)(free_b * 0.8);
    int64_t cols = std::min<int64_t>(c->Fp, (int64_t)(budget / per_col) / kBN * kBN);
    if (cols < kBN) { set_err(c, "cafe_reconstruct: not enough device memory for %d product panels", nI); return CAFE_ERR_MEMORY; }
    const int n_groups = (M + 15) / 16;                     // row groups of a panel (flags, see recon_leaf_kernel)
    const int64_t n_tiles = cols / 64;
    DevBuf panels, st, prior, flg;
    if (hipMalloc(&panels.p, (size_t)nI * rows * cols * sizeof(double)) != hipSuccess || hipMalloc(&st.p, (size_t)2 * n * cols * sizeof(int32_t)) != hipSuccess ||
        hipMalloc(&prior.p, sizeof(double) * (jmax + 1)) != hipSuccess || hipMalloc(&flg.p, sizeof(int32_t) * (size_t)nI * n_tiles * n_groups) != hipSuccess) {
        (void)hipGetLastError();
        set_err(c, "cafe_reconstruct: cannot allocate the workspace (%lld columns)", (long long)cols);
        return CAFE_ERR_MEMORY;
    }
    double* d_B = static_cast<double*>(panels.p);
    int32_t* d_state = static_cast<int32_t*>(st.p);
    int32_t* d_state_t = d_state + (size_t)n * cols;         // [column][node] for the way back
    int32_t* d_flags = static_cast<int32_t*>(flg.p);
    const bool use_flags = n_groups <= 64 && !std::getenv("CAFE_NO_RECON_FLAGS");
    // rows 0..M of a panel are written by its first child; the padding rows (K5 reads in groups of four) must not hold NaN patterns
    if (rows > M + 1)
        HIP_TRY(c, hipMemset2DAsync(d_B + (int64_t)(M + 1) * cols, sizeof(double) * (size_t)rows * cols, 0, sizeof(double) * (size_t)(rows - M - 1) * cols, nI, s));
    {
        std::vector<double> hp(jmax + 1);
        for (int j = 0; j <= jmax; ++j) hp[j] = (double)root_prior[j];         // compute() returns a float
        HIP_TRY(c, hipMemcpyAsync(prior.p, hp.data(), sizeof(double) * (jmax + 1), hipMemcpyHostToDevice, s));
        HIP_TRY(c, hipStreamSynchronize(s));
    }
    const int64_t pstride = (int64_t)rows * cols;
    std::unique_ptr<int32_t[]> h_state(new int32_t[(size_t)n * cols]);      // [column][node]

    for (int k = 0; k < K; ++k)
        for (int64_t f0 = 0; f0 < c->Fp; f0 += cols) {
            const int64_t ld = std::min<int64_t>(cols, c->Fp - f0);
            // ---- forward: children before parents (node order of the problem)
            std::vector<char> started(n, 0);
            for (int v = 0; v < n; ++v) {
                if (v == c->root) continue;
                const int par = c->parent[v];
                double* dst = d_B + (int64_t)bidx[par] * pstride;
                const bool mul = started[par];
                started[par] = 1;
                const int slot = c->slot_of[(size_t)v * c->Kmax + k];
                (void)hipGetLastError();
                if (c->leaf_taxon[v] >= 0) {
                    const double* P = c->pool.base + (int64_t)slot * c->pool.stride;
                    const int32_t* cnt = c->d_counts + (int64_t)c->leaf_taxon[v] * c->Fp + f0;
                    dim3 grid((unsigned)((ld + 255) / 256), (unsigned)n_groups);
                    int32_t* dfl = d_flags + (int64_t)bidx[par] * n_tiles * n_groups;
                    if (mul) hipLaunchKernelGGL(recon_leaf_kernel<true>, grid, dim3(256), 0, s, P, c->pool.ld, cnt, dst, ld, M, dfl, n_groups);
                    else hipLaunchKernelGGL(recon_leaf_kernel<false>, grid, dim3(256), 0, s, P, c->pool.ld, cnt, dst, ld, M, dfl, n_groups);
                } else {
                    const double* Pt = c->kpool.base + (int64_t)slot * c->kpool.stride;
                    const double* B = d_B + (int64_t)bidx[v] * pstride;
                    const int n_ct = (int)(ld / 64), n_rg = (M + kRowsPerBlock - 1) / kRowsPerBlock;
                    dim3 grid((unsigned)(8 * ((n_ct + 7) / 8) * n_rg));
                    const int32_t* ext = c->kpool.ext ? c->kpool.ext + (size_t)slot * c->kpool.ext_blocks * 2 : nullptr;
                    const int32_t* bfl = use_flags ? d_flags + (int64_t)bidx[v] * n_tiles * n_groups : nullptr;
                    int32_t* dfl = d_flags + (int64_t)bidx[par] * n_tiles * n_groups;
                    if (mul) hipLaunchKernelGGL(maxprod_kernel<true>, grid, dim3(256), 0, s, Pt, c->kpool.ld, B, dst, ld, M, c->kpool.rows, ext, bfl, dfl, n_groups);
                    else hipLaunchKernelGGL(maxprod_kernel<false>, grid, dim3(256), 0, s, Pt, c->kpool.ld, B, dst, ld, M, c->kpool.rows, ext, bfl, dfl, n_groups);
                }
                HIP_TRY(c, hipGetLastError());
            }
            // ---- root choice, then parents before children
            const unsigned gb = (unsigned)((ld + 255) / 256);
            hipLaunchKernelGGL(root_select_kernel, dim3(gb), dim3(256), 0, s, d_B + (int64_t)bidx[c->root] * pstride, ld,
                               static_cast<const double*>(prior.p), jmax, d_state + (int64_t)c->root * cols);
            HIP_TRY(c, hipGetLastError());
            for (int v = n - 1; v >= 0; --v) {
                if (v == c->root) continue;
                if (c->leaf_taxon[v] >= 0) {
                    HIP_TRY(c, hipMemcpyAsync(d_state + (int64_t)v * cols, c->d_counts + (int64_t)c->leaf_taxon[v] * c->Fp + f0, sizeof(int32_t) * ld,
                                              hipMemcpyDeviceToDevice, s));
                    continue;
                }
                const int slot = c->slot_of[(size_t)v * c->Kmax + k];
                hipLaunchKernelGGL(backtrack_kernel, dim3((unsigned)(ld / 64)), dim3(256), 0, s, c->kpool.base + (int64_t)slot * c->kpool.stride, c->kpool.ld,
                                   d_B + (int64_t)bidx[v] * pstride, ld, M, d_state + (int64_t)c->parent[v] * cols, d_state + (int64_t)v * cols,
                                   c->kpool.ext ? c->kpool.ext + (size_t)slot * c->kpool.ext_blocks * 2 : nullptr);
                HIP_TRY(c, hipGetLastError());
            }
            hipLaunchKernelGGL(state_transpose_kernel, dim3(gb, (unsigned)((n + 15) / 16)), dim3(256), 0, s, d_state, n, cols, ld, d_state_t);
            HIP_TRY(c, hipGetLastError());
            HIP_TRY(c, hipMemcpyAsync(h_state.get(), d_state_t, sizeof(int32_t) * (size_t)n * ld, hipMemcpyDeviceToHost, s));
            HIP_TRY(c, hipStreamSynchronize(s));
            // unique column -> every family that shares it
            for (int64_t f = 0; f < c->F_all; ++f) {
                const int64_t u = c->ref_of[f];
                if (u < f0 || u >= f0 + ld) continue;
                std::memcpy(states + ((int64_t)k * c->F_all + f) * n, h_state.get() + (size_t)(u - f0) * n, sizeof(int32_t) * n);
            }
        }
    c->upload_pending = false;
    return CAFE_OK;
}

int branch_probabilities_impl(cafe_ctx* c, const cafe_params* pr, const int32_t* sizes, double* out) {
    if (!pr || !pr->lambdas || !sizes || !out) { set_err(c, "cafe_branch_probabilities: lambdas, sizes and out are required"); return CAFE_ERR_ARGUMENT; }
    if (!lambdas_valid(c, pr->lambdas)) { set_err(c, "cafe_branch_probabilities: invalid lambda"); return CAFE_ERR_ARGUMENT; }
    const int n = c->n_nodes;
    const int64_t F = c->F_all;
    for (int64_t i = 0; i < F * n; ++i)
        if (sizes[i] < 0 || sizes[i] > c->M) { set_err(c, "cafe_branch_probabilities: size outside [0, %d] for family %lld", c->M, (long long)(i / n)); return CAFE_ERR_ARGUMENT; }
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    if (c->upload_pending) { HIP_TRY(c, hipEventSynchronize(c->ev_upload)); c->upload_pending = false; }
    c->have_results = false;
    c->last_stream = s;
    c->K_last = 1;
    { const int rc = prepare_matrices(c, pr->lambdas, nullptr, 1, s); if (rc != CAFE_OK) return rc; }     // the plain lambda (execute.cpp:158)
    HIP_TRY(c, hipEventRecord(c->ev_upload, s));
    c->upload_pending = true;
    std::vector<int32_t> h_parent(n), h_leaf(n), h_slot(n, 0);
    for (int v = 0; v < n; ++v) {
        h_parent[v] = c->parent[v];
        h_leaf[v] = c->leaf_taxon[v] >= 0;
        if (v != c->root) h_slot[v] = c->slot_of[(size_t)v * c->Kmax];
    }
    DevBuf d_sizes, d_out, d_meta;
    if (hipMalloc(&d_sizes.p, sizeof(int32_t) * F * n) != hipSuccess || hipMalloc(&d_out.p, sizeof(double) * F * n) != hipSuccess ||
        hipMalloc(&d_meta.p, sizeof(int32_t) * 3 * n) != hipSuccess) {
        (void)hipGetLastError();
        set_err(c, "cafe_branch_probabilities: cannot allocate %lld x %d entries", (long long)F, n);
        return CAFE_ERR_MEMORY;
    }
    int32_t* meta = static_cast<int32_t*>(d_meta.p);
    HIP_TRY(c, hipMemcpyAsync(d_sizes.p, sizes, sizeof(int32_t) * F * n, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(meta, h_parent.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(meta + n, h_leaf.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(meta + 2 * n, h_slot.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice, s));
    (void)hipGetLastError();
    hipLaunchKernelGGL(viterbi_kernel, dim3((unsigned)((F * n + 255) / 256)), dim3(256), 0, s, static_cast<const int32_t*>(d_sizes.p), F, n, meta, meta + n,
                       meta + 2 * n, c->pool, c->kpool, c->M, static_cast<double*>(d_out.p));
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(out, d_out.p, sizeof(double) * F * n, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    c->upload_pending = false;
    return CAFE_OK;
}

}  // namespace cafe
