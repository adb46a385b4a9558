// compute_pvalues (src/probability.cpp:255-454) entirely on the device.
//
// The reference simulates, for every root size i < R, `n` families down the tree (each child size drawn from row
// `parent size` of the branch's transition matrix restricted to sizes 0..M-1: set_weighted_random_family_size,
// :320-351), prunes every simulated family and keeps max_j L_root[j] (:273-317), sorts the n values per root size and
// reports for each observed family max_i upper_bound(conditional[i], observed) / n (:379-444).
//
// cafexp_amd/host/pvalues.cpp follows that draw for draw on the host (same engine, same libstdc++ distributions) and is
// what the parity tests compare with the reference at a fixed seed.  This file is the same computation for shapes where
// one host thread and a host copy of every matrix would dominate (100 taxa, N = 751: 148 M draws, 0.9 GB): the draws
// come from a counter-based generator (Philox4x32-10 keyed by the seed, counter = (family, node)), so the simulated
// families are a DIFFERENT sample of the same distribution: p-values agree with the reference statistically (Monte
// Carlo error ~ sqrt(p(1-p)/n)), not draw for draw.
//   row_cdf       prefix sums of every branch's row-major matrix rows over c = 0..M-1 (inverse-CDF sampling)
//   simulate      one thread per simulated family, nodes parents first; sizes in a [node][family] scratch, leaves
//                 written straight into the child context's taxon-major count table
//   (prune)       cafe_ctx.hip's root-maximum schedule on the child context and on the observed families
//   sort_rows     bitonic sort of each root size's n values in LDS
//   tree_pvalue   per observed family: max over root sizes of the upper_bound position
#include <algorithm>
#include <cstdint>
#include <map>
#include <vector>

#include "cafe_ctx.h"

namespace cafe {

namespace {

struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
};

// inclusive prefix sums along c of rows 1..n-1 of `count` row-major matrices; one wave per row
__global__ __launch_bounds__(64) void row_cdf_kernel(double* __restrict__ base, int64_t stride, int ld, int n, int m_cols) {
    const int row = blockIdx.x + 1, slot = blockIdx.y, lane = threadIdx.x;
    double* r = base + (int64_t)slot * stride + (int64_t)row * ld;
    double carry = 0.0;
    for (int c0 = 0; c0 < m_cols; c0 += 64) {
        const int c = c0 + lane;
        double v = c < m_cols ? r[c] : 0.0;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const double up = __shfl_up(v, d);
            if (lane >= d) v += up;
        }
        v += carry;
        if (c < m_cols) r[c] = v;
        carry = __shfl(v, 63);
    }
    (void)n;
}

__device__ inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

struct SimArgs {
    const double* cdf;          // [slot][N][ld] row-major prefix sums
    int64_t stride;
    int32_t ld, M, n_nodes, n_sim;
    const int32_t* order;       // nodes, parents before children (root first)
    const int32_t* parent;      // [n_nodes]
    const int32_t* slot;        // [n_nodes] cdf slot of the branch above the node
    const int32_t* leaf_taxon;  // [n_nodes]
    int32_t* sizes;             // [n_nodes][ld_f] scratch
    int32_t* counts;            // child context's [taxon][counts_ld]
    int64_t counts_ld, ld_f, n_families;
    uint32_t k0, k1;
};

__global__ __launch_bounds__(256) void simulate_kernel(const SimArgs a) {
    const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (f >= a.n_families) return;
    const int root_size = (int)(f / a.n_sim);
    for (int t = 0; t < a.n_nodes; ++t) {
        const int v = a.order[t];
        const int par = a.parent[v];
        int size;
        if (par < 0) {
            size = root_size;
        } else {
            const int ps = a.sizes[(int64_t)par * a.ld_f + f];
            size = 0;
            if (ps > 0) {                                   // an extinct lineage stays extinct, no draw (:328)
                const double* row = a.cdf + (int64_t)a.slot[v] * a.stride + (int64_t)ps * a.ld;
                uint32_t r[4];
                philox4x32_10((uint32_t)f, (uint32_t)(f >> 32), (uint32_t)v, 0u, a.k0, a.k1, r);
                const double u = ((double)(((uint64_t)r[0] << 21) ^ (r[1] >> 11)) + 0.5) * (1.0 / 9007199254740992.0);   // (0,1)
                // A saturated / degenerate branch has an all-zero row (matrix_cache.cpp:153): target = 0 and the search
                // returns size 0.  The reference draws from std::discrete_distribution over all-zero weights there
                // (probability.cpp:333-344, after a uniform draw it then discards) -- outside that distribution's
                // precondition (sum of weights > 0); libstdc++ returns index 0, so does this path, by construction.
                const double target = u * row[a.M - 1];     // sizes 0..M-1 carry the weights (:338-341)
                int lo = 0, hi = a.M - 1;                   // first c with cdf[c] >= target
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (row[mid] >= target) hi = mid; else lo = mid + 1;
                }
                size = lo;
            }
        }
        a.sizes[(int64_t)v * a.ld_f + f] = size;
        const int tx = a.leaf_taxon[v];
        if (tx >= 0) a.counts[(int64_t)tx * a.counts_ld + f] = size;
    }
}

// one block per root size: sorts its n values ascending (n <= 2048, padded with +inf)
__global__ __launch_bounds__(256) void sort_rows_kernel(double* __restrict__ v, int n) {
    extern __shared__ double sh[];
    int P2 = 1;
    while (P2 < n) P2 <<= 1;
    double* row = v + (int64_t)blockIdx.x * n;
    for (int i = threadIdx.x; i < P2; i += 256) sh[i] = i < n ? row[i] : __builtin_inf();
    __syncthreads();
    for (int k = 2; k <= P2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < P2; i += 256) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const bool up = (i & k) == 0;
                    const double x = sh[i], y = sh[ixj];
                    if ((x > y) == up) { sh[i] = y; sh[ixj] = x; }
                }
            }
            __syncthreads();
        }
    for (int i = threadIdx.x; i < n; i += 256) row[i] = sh[i];
}

// compute_tree_pvalue: max over root sizes of pvalue(observed, conditional[s]) (probability.cpp:379-407)
__global__ __launch_bounds__(256) void tree_pvalue_kernel(const double* __restrict__ observed, int64_t F, const double* __restrict__ cond, int R, int n,
                                                          double* __restrict__ out) {
    const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (f >= F) return;
    const double v = observed[f];
    double best = 0.0;
    for (int s = 0; s < R; ++s) {
        const double* c = cond + (int64_t)s * n;
        int lo = 0, hi = n;                                 // first element > v
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (!(v < c[mid])) lo = mid + 1; else hi = mid;
        }
        const int idx = lo != n ? lo : n - 1;
        const double p = idx / (double)n;
        if (s == 0 || p > best) best = p;
    }
    out[f] = best;
}

}  // namespace

int pvalues_impl(cafe_ctx* c, const cafe_params* pr, int32_t n_sim, uint64_t seed, double* pvalues) {
    if (!pr || !pr->lambdas || !pvalues) { set_err(c, "cafe_pvalues: lambdas and pvalues are required"); return CAFE_ERR_ARGUMENT; }
    if (n_sim < 1 || n_sim > 2048) { set_err(c, "cafe_pvalues: 1..2048 simulations per root size"); return CAFE_ERR_ARGUMENT; }
    if (!lambdas_valid(c, pr->lambdas)) { set_err(c, "cafe_pvalues: invalid lambda"); return CAFE_ERR_ARGUMENT; }
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const int n = c->n_nodes, N = c->N, M = c->M, R = c->R;
    const int64_t Fs = (int64_t)R * n_sim;

    // ---- observed families: max_j L_root[j] (also builds this call's matrices for the plain lambdas)
    { const int rc = enqueue_rootmax(c, pr->lambdas, s); if (rc != CAFE_OK) return rc; }

    // ---- row-major matrix + CDF of EVERY branch (the scorer keeps interior branches k-major only)
    MatrixPool sp{};
    sp.n = N; sp.ld = round_up(N, 16); sp.stride = (int64_t)N * sp.ld; sp.kmajor = 0; sp.rows = N; sp.k_valid = N;
    std::map<std::pair<long, long>, int> key_slot;
    std::vector<SlotParam> slots;
    std::vector<int32_t> h_slot(n, 0), h_parent(n), h_leaf(n), h_order;
    for (int v = 0; v < n; ++v) {
        h_parent[v] = c->parent[v];
        h_leaf[v] = c->leaf_taxon[v];
        if (v == c->root) continue;
        const long lq = long(pr->lambdas[c->lam_idx[v]] * 1000000000), tq = long(c->blen[v] * 1000);     // matrix_cache.h:47-50
        auto it = key_slot.find({tq, lq});
        if (it == key_slot.end()) {
            const double lambda_q = double(lq) / 1000000000.0, t_q = double(tq) / 1000.0;
            const double alpha = lambda_q * t_q / (1 + lambda_q * t_q), coeff = 1 - 2 * alpha;
            SlotParam p0; p0.alpha = alpha; p0.oma2 = (1 - alpha) * (1 - alpha); p0.zero = !(coeff > 0 && coeff != 1); p0.pad = 0;
            it = key_slot.emplace(std::make_pair(tq, lq), (int)slots.size()).first;
            slots.push_back(p0);
        }
        h_slot[v] = it->second;
    }
    for (int v = n - 1; v >= 0; --v) h_order.push_back(v);          // parents have larger indices: descending = parents first
    DevBuf d_pool, d_sp, d_meta, d_sizes, d_cond, d_pv;
    const size_t pool_bytes = sizeof(double) * (size_t)sp.stride * slots.size();
    if (hipMalloc(&d_pool.p, pool_bytes) != hipSuccess || hipMalloc(&d_sp.p, sizeof(SlotParam) * slots.size()) != hipSuccess ||
        hipMalloc(&d_meta.p, sizeof(int32_t) * 4 * n) != hipSuccess || hipMalloc(&d_cond.p, sizeof(double) * Fs) != hipSuccess ||
        hipMalloc(&d_pv.p, sizeof(double) * c->F_uniq) != hipSuccess) {
        (void)hipGetLastError();
        set_err(c, "cafe_pvalues: cannot allocate the simulation workspace");
        return CAFE_ERR_MEMORY;
    }
    sp.base = static_cast<double*>(d_pool.p);
    HIP_TRY(c, hipMemsetAsync(d_pool.p, 0, pool_bytes, s));
    HIP_TRY(c, hipMemcpyAsync(d_sp.p, slots.data(), sizeof(SlotParam) * slots.size(), hipMemcpyHostToDevice, s));
    int32_t* meta = static_cast<int32_t*>(d_meta.p);
    HIP_TRY(c, hipMemcpyAsync(meta, h_order.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(meta + n, h_parent.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(meta + 2 * n, h_slot.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(meta + 3 * n, h_leaf.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice, s));
    HIP_TRY(c, launch_bd_matrix_build(sp, static_cast<const SlotParam*>(d_sp.p), (int)slots.size(), s));
    (void)hipGetLastError();
    hipLaunchKernelGGL(row_cdf_kernel, dim3(N - 1, (unsigned)slots.size()), dim3(64), 0, s, sp.base, sp.stride, sp.ld, N, M);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(s));                            // the host vectors above go out of use

    // ---- simulate into a child context over the same tree, prune, keep the root maxima
    cafe_ctx* child = create_child_for_device_counts(c, Fs);
    if (!child) { set_err(c, "cafe_pvalues: cannot create the context of %lld simulated families", (long long)Fs); return CAFE_ERR_MEMORY; }
    struct ChildGuard { cafe_ctx* p; ~ChildGuard() { destroy_child(p); } } guard{child};
    if (hipMalloc(&d_sizes.p, sizeof(int32_t) * (size_t)n * Fs) != hipSuccess) {
        (void)hipGetLastError();
        set_err(c, "cafe_pvalues: cannot allocate the simulation scratch");
        return CAFE_ERR_MEMORY;
    }
    SimArgs a{};
    a.cdf = sp.base; a.stride = sp.stride; a.ld = sp.ld; a.M = M; a.n_nodes = n; a.n_sim = n_sim;
    a.order = meta; a.parent = meta + n; a.slot = meta + 2 * n; a.leaf_taxon = meta + 3 * n;
    a.sizes = static_cast<int32_t*>(d_sizes.p); a.counts = child->d_counts; a.counts_ld = child->Fp; a.ld_f = Fs; a.n_families = Fs;
    a.k0 = (uint32_t)seed; a.k1 = (uint32_t)(seed >> 32);
    (void)hipGetLastError();
    hipLaunchKernelGGL(simulate_kernel, dim3((unsigned)((Fs + 255) / 256)), dim3(256), 0, s, a);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(s));                            // the child context runs on its own stream
    { const int rc = enqueue_rootmax(child, pr->lambdas, child->stream); if (rc != CAFE_OK) { set_err(c, "cafe_pvalues: %s", child->err.c_str()); return rc; } }
    HIP_TRY(c, hipMemcpyAsync(d_cond.p, child->d_fam_out, sizeof(double) * Fs, hipMemcpyDeviceToDevice, child->stream));
    int P2 = 1;
    while (P2 < n_sim) P2 <<= 1;
    hipLaunchKernelGGL(sort_rows_kernel, dim3(R), dim3(256), sizeof(double) * P2, child->stream, static_cast<double*>(d_cond.p), n_sim);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(child->stream));
    child->upload_pending = false;

    // ---- p-value of every (distinct) observed family, spread to the families that share a column
    hipLaunchKernelGGL(tree_pvalue_kernel, dim3((unsigned)((c->F_uniq + 255) / 256)), dim3(256), 0, s, c->d_fam_out, c->F_uniq,
                       static_cast<const double*>(d_cond.p), R, n_sim, static_cast<double*>(d_pv.p));
    HIP_TRY(c, hipGetLastError());
    std::vector<double> tmp((size_t)c->F_uniq);
    HIP_TRY(c, hipMemcpyAsync(tmp.data(), d_pv.p, sizeof(double) * c->F_uniq, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    c->upload_pending = false;
    for (int64_t f = 0; f < c->F_all; ++f) pvalues[f] = tmp[c->ref_of[f]];
    return CAFE_OK;
}

}  // namespace cafe
