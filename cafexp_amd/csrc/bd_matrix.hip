// K1 bd_matrix_build: all birth-death transition matrices of one scorer call.
//
// Replaces matrix_cache::precalculate_matrices (src/matrix_cache.cpp:121-171) and, entry by
// entry, the_probability_of_going_from_parent_fam_size_to_c (src/probability.cpp:147) /
// birthdeath_rate_with_log_alpha (src/probability.cpp:101).  The reference evaluates every
// entry as a log-space sum of min(s,c)+1 terms (O(N^3) per matrix).  The closed form it sums is
// the s-fold convolution of the single-lineage law of the critical linear birth-death process,
//     p1(0) = a,   p1(k) = (1-a)^2 a^(k-1)  (k >= 1),    a = lambda t / (1 + lambda t),
// so row s = row s-1 (*) p1, which is a first-order linear recurrence along the row:
//     h(c) = P[s-1][c-1] + a h(c-1),        P[s][c] = a P[s-1][c] + (1-a)^2 h(c).
// All terms are non-negative (no cancellation), O(N^2) per matrix.  The reference's special
// cases are kept: row 0 = e_0 (matrix_cache.cpp:70-77), saturated or degenerate coeff => rows
// s >= 1 are zero (matrix_cache.cpp:153, probability.cpp:154), values clamped to [0,1]
// (probability.cpp:145).
//
// Mapping: ONE 64-lane wave per matrix.  Lane l owns E consecutive columns of the current row
// in registers; a row step is E local FMAs, a scan over the 64 lane aggregates with the constant
// ratio a^E, and E fix-up FMAs.  The scan runs on DPP moves only (no LDS crossbar, no barrier):
// four Kogge-Stone steps inside each row of 16 lanes (row_shr:1,2,4,8), then the row totals are
// carried over with row_bcast:15 and row_bcast:31 times a per-lane power of the ratio; the
// neighbour values (last column of the lane to the left, the carry) are wave_shr:1.  A row step is
// a dependent chain, so its latency is the kernel's time until the HBM write of the pool takes
// over (__shfl_up, which compiles to ds_bpermute, made a step of N = 751 take 2.5 us).  The row steps are sequential; the grid has one wave per (branch, category)
// matrix, so a call with hundreds of matrices fills the chip.  Measured at N = 751, 1320 matrices (tools/k1_time.py): 1.13 ms, of
// which 0.69 ms is the chain of row steps (the build without its stores) and 0.44 ms the LDS turn + the write of 5.8 GB.
//
// Two output layouts:
//   row-major  P[s][c]              -- leaf branches: K3 reads column x of P (P . e_x)
//   k-major    Pt[c][j] = P[j+1][c] -- interior branches: the A operand of K2, contraction index c
//                                      outermost so that an A tile row is contiguous in LDS-DMA order.
// The k-major layout is written without a transpose: the process is reversible with respect to
// pi(n) = 1/n, so P[s][c] = (s/c) P[c][s] for s,c >= 1; Pt's row c is P's row c scaled by s/c
// (one extra rounding), Pt's row 0 is P[s][0] = a^s, and P's row 0 (e_0) is never stored: K2
// copies that row (prune_gemm.hip).
#include "cafe_kernels.h"

// diagnostic: -D'CAFE_EXPERIMENT_K1_STORE_IF=&& n < 0' builds K1 without its global stores (the matrices are wrong): what is
// left is the latency chain of the row steps, DESIGN.md section 3 K1
#ifndef CAFE_EXPERIMENT_K1_STORE_IF
#define CAFE_EXPERIMENT_K1_STORE_IF
#endif

namespace cafe {

// 1 / r for the k-major scaling P[s][c] = (s/c) P[c][s]: the division (v_rcp_f64 + two scales + four FMAs + fix-up) sat in
// every row step; the correctly rounded quotient is the same number whoever divides, so it is folded at compile time and read
// with a scalar load (r is uniform).  Orders go up to bd_matrix_max_order() = 2048.
struct InvTable {
    double v[2048];
    constexpr InvTable() : v() {
        for (int i = 1; i < 2048; ++i) v[i] = 1.0 / (double)i;
    }
};
__constant__ InvTable kInvR = InvTable();

// DPP move of a double (two 32-bit halves); lanes without a source read 0
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ double dpp_move(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, true);
    return __hiloint2double(hi, lo);
}
constexpr int kDppRowShr1 = 0x111, kDppRowShr2 = 0x112, kDppRowShr4 = 0x114, kDppRowShr8 = 0x118;
constexpr int kDppWaveShr1 = 0x138, kDppRowBcast15 = 0x142, kDppRowBcast31 = 0x143;

template <int E, bool KMAJOR>
__device__ __forceinline__ void bd_matrix_build_one(const MatrixPool& pool, const SlotParam sp, int slot) {
    const int lane = threadIdx.x;
    double* __restrict__ P = pool.base + (int64_t)slot * pool.stride;
    const int ld = pool.ld;
    const int n = pool.n;                             // matrix order N (sizes 0..N-1)
    const int n_rows = KMAJOR ? pool.rows : n;        // rows to write
    const int k_valid = KMAJOR ? pool.k_valid : n;    // recurrence rows that are ever read
    constexpr int e_base = KMAJOR ? 1 : 0;            // first owned column of lane 0
    const int c0 = e_base + lane * E;                 // owned columns c0 .. c0+E-1 of the current P row
    const int j0 = lane * E;                          // where they are stored
    const double a = sp.alpha, q = sp.oma2;

    double apow[E];                      // a^(i+1)
    apow[0] = a;
#pragma unroll
    for (int i = 1; i < E; ++i) apow[i] = apow[i - 1] * a;
    double ratio[4];                     // (a^E)^(2^d): the in-row scan steps
    ratio[0] = apow[E - 1];
#pragma unroll
    for (int d = 1; d < 4; ++d) ratio[d] = ratio[d - 1] * ratio[d - 1];
    // what a lane of rows 1, 3 (rows 2, 3) adds of the total that lane 15 of the row before (lane 31) holds
    const double w15 = pow(apow[E - 1], (double)((lane & 15) + 1));
    const double w31 = lane >= 32 ? pow(apow[E - 1], (double)(lane - 31)) : 0.0;

    // Columns past the matrix (c0 + i >= n: the tail of the last lanes) must be stored as zeros.  Up to E = 16 they ARE zeros: the
    // lane multiplies h by a per-element (1-a)^2 that is 0 there, so p stays exactly 0 (a p + 0 h) and nothing is masked per
    // row step -- 2 E selects less of its ~200 instructions; columns inside the matrix see the same operands as before.  Wider
    // E keeps the selects (E more live doubles would spill).
    constexpr bool QM = E <= 16;
    double qm[QM ? E : 1];
#pragma unroll
    for (int i = 0; i < (QM ? E : 1); ++i) qm[i] = (c0 + i < n) ? q : 0.0;
    double p[E];                         // P[row][c0 + i]
    double p0 = 1.0;                     // P[row][0] = a^row (k-major only: lane 0's left neighbour)
#pragma unroll
    for (int i = 0; i < E; ++i) p[i] = (c0 + i == 0) ? 1.0 : 0.0;

    // A lane owns E consecutive columns (what the scan needs); stored from there, an instruction would write 64 pieces of
    // 16 bytes 8*E bytes apart.  The row is turned through LDS instead (one wave per block: a wait on the LDS counter is
    // the only synchronisation) and leaves as 1 KB contiguous per store instruction.
    __shared__ double2 rowbuf[64 * E / 2];
    auto store_row = [&](int r, const double* v) {
        if constexpr (E <= 4) {
            // small orders: a matrix is a chain of short row steps, not a stream of lines, and the LDS round trip with its two
            // waits is a seventh of a step (mammals K1 59.5 -> 51 us).  The lane stores its own columns (pieces 8*E bytes apart: the row is 1-2 KB).
            double2* row = reinterpret_cast<double2*>(P + (int64_t)r * ld);
#pragma unroll
            for (int i = 0; i < E; i += 2) {
                double2 w;
                w.x = (QM || c0 + i < n) ? v[i] : 0.0;
                w.y = (QM || c0 + i + 1 < n) ? v[i + 1] : 0.0;
                if (j0 + i < ld CAFE_EXPERIMENT_K1_STORE_IF) row[(j0 + i) >> 1] = w;
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < E; i += 2) {
            double2 w;
            w.x = (QM || c0 + i < n) ? v[i] : 0.0;
            w.y = (QM || c0 + i + 1 < n) ? v[i + 1] : 0.0;
            rowbuf[(j0 + i) >> 1] = w;
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0): every lane's part of the row is in LDS
        __asm__ volatile("" ::: "memory");
        double2* row = reinterpret_cast<double2*>(P + (int64_t)r * ld);
#pragma unroll
        for (int i = 0; i < E / 2; ++i) {
            const int q = lane + 64 * i;                 // 16-byte piece of the row
            if (2 * q < ld CAFE_EXPERIMENT_K1_STORE_IF) row[q] = rowbuf[q];      // (non-temporal stores measured: 1.24 -> 1.29-1.33 ms at order 751)
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);          // the pieces are in registers before the next row overwrites the buffer
        __asm__ volatile("" ::: "memory");
    };

    double z[E];
#pragma unroll
    for (int i = 0; i < E; ++i) z[i] = 0.0;

    // k-major: the non-zero extent (first / last contraction index) of every block of 16 stored columns, for K2
    __shared__ int ext_lo[128], ext_hi[128];
    int first_nz = 0x7fffffff, last_nz = -1;          // over this lane's columns
    int col_first[KMAJOR ? 1 : E], col_last[KMAJOR ? 1 : E];   // row-major: first / last row (parent size) with a non-zero entry, per owned column
#pragma unroll
    for (int i = 0; i < (KMAJOR ? 1 : E); ++i) { col_first[i] = 0x7fffffff; col_last[i] = -1; }
    auto note = [&](int r, const double* v) {
        if (!pool.ext) return;                         // (uniform) orders below 256 publish no extents: nothing to track in the row step
        if (KMAJOR) {
            bool any = false;
#pragma unroll
            for (int i = 0; i < E; ++i) any = any || ((QM || c0 + i < n) && v[i] != 0.0);
            if (any) { first_nz = first_nz < r ? first_nz : r; last_nz = r; }
        } else {
#pragma unroll
            for (int i = 0; i < (KMAJOR ? 1 : E); ++i)
                if (v[i] != 0.0) { col_first[i] = col_first[i] < r ? col_first[i] : r; col_last[i] = r; }
        }
    };
    auto publish_extents = [&]() {
        if (!pool.ext) return;
        if (!KMAJOR) {                                 // per column x of P: rows s with P[s][x] != 0 (the support of a leaf's factor)
            int32_t* out = pool.ext + (int64_t)slot * pool.ext_blocks * 2;
#pragma unroll
            for (int i = 0; i < (KMAJOR ? 1 : E); ++i)
                if (c0 + i < n) { out[2 * (c0 + i)] = col_first[i]; out[2 * (c0 + i) + 1] = col_last[i]; }
            return;
        }
        const int nb = pool.ext_blocks;
        for (int b = lane; b < nb; b += 64) { ext_lo[b] = 0x7fffffff; ext_hi[b] = -1; }
        __syncthreads();                               // one wave per block: orders the LDS initialisation
        if (last_nz >= 0 && j0 < n - 1) {
            const int b_lo = j0 >> 4, b_hi = min(j0 + E - 1, n - 2) >> 4;
            for (int b = b_lo; b <= b_hi && b < nb; ++b) { atomicMin(&ext_lo[b], first_nz); atomicMax(&ext_hi[b], last_nz); }
        }
        __syncthreads();
        int32_t* out = pool.ext + (int64_t)slot * nb * 2;
        for (int b = lane; b < nb; b += 64) { out[2 * b] = ext_lo[b]; out[2 * b + 1] = ext_hi[b]; }
    };

    if (sp.zero) {                       // saturated / degenerate: every entry with parent size >= 1 is 0
        if (!KMAJOR) { store_row(0, p); note(0, p); }    // row-major keeps P's row 0 = e_0; k-major never holds it
        for (int r = KMAJOR ? 0 : 1; r < n_rows; ++r) store_row(r, z);
        publish_extents();               // all blocks empty
        return;
    }

    if (KMAJOR) {
        double v[E];                     // Pt[0][j] = P[j+1][0] = a^(j+1)
#pragma unroll
        for (int i = 0; i < E; ++i) v[i] = (!QM || c0 + i < n) ? pow(a, (double)(c0 + i)) : 0.0;
        store_row(0, v);
        note(0, v);
    } else {
        store_row(0, p);
        note(0, p);
    }

    for (int r = 1; r < n_rows; ++r) {
        if (r >= k_valid || r >= n) {    // contraction rows past M (or past the matrix) are never read: keep them 0
            store_row(r, z);
            continue;
        }
        double left = dpp_move<kDppWaveShr1>(p[E - 1]);
        if (lane == 0) left = KMAJOR ? p0 : 0.0;
        double h[E];
        h[0] = left;
#pragma unroll
        for (int i = 1; i < E; ++i) h[i] = fma(a, h[i - 1], p[i - 1]);
        double S = h[E - 1];             // inclusive scan of the lane totals with ratio a^E
        S = fma(ratio[0], dpp_move<kDppRowShr1>(S), S);          // lanes without a source add ratio * 0
        S = fma(ratio[1], dpp_move<kDppRowShr2>(S), S);
        S = fma(ratio[2], dpp_move<kDppRowShr4>(S), S);
        S = fma(ratio[3], dpp_move<kDppRowShr8>(S), S);
        S = fma(w15, dpp_move<kDppRowBcast15, 0xa>(S), S);       // rows 1 and 3 take the total of rows 0 and 2
        S = fma(w31, dpp_move<kDppRowBcast31, 0xc>(S), S);       // rows 2 and 3 take the total of rows 0..1
        const double carry = dpp_move<kDppWaveShr1>(S);          // lane 0: 0
#pragma unroll
        for (int i = 0; i < E; ++i) {
            double hh = fma(apow[i], carry, h[i]);
            double v = fma(a, p[i], (QM ? qm[i] : q) * hh);
            v = v < 1.0 ? v : 1.0;
            p[i] = v > 0.0 ? v : 0.0;
        }
        p0 *= a;
        if (KMAJOR) {
            const double inv_r = kInvR.v[r];               // = 1.0 / (double)r, bit for bit
            double v[E];
#pragma unroll
            for (int i = 0; i < E; ++i) {
                double t = p[i] * ((double)(c0 + i) * inv_r);     // P[s][c] = (s/c) P[c][s]
                v[i] = t < 1.0 ? t : 1.0;
            }
            store_row(r, v);
            note(r, v);
        } else {
            store_row(r, p);
            note(r, p);
        }
    }
    publish_extents();
}

template <int E, bool KMAJOR>
__global__ __launch_bounds__(64) void bd_matrix_build_kernel(MatrixPool pool, const SlotParam* __restrict__ slots, int n_slots) {
    const int slot = blockIdx.x;
    if (slot >= n_slots) return;
    bd_matrix_build_one<E, KMAJOR>(pool, slots[slot], slot);
}

// both pools of a scorer call in one launch: each matrix is a latency-bound chain of N row steps on one wave, so the
// row-major and the k-major matrices should be in flight together rather than one launch after the other
template <int E>
__global__ __launch_bounds__(64) void bd_matrix_build_both_kernel(MatrixPool pool, MatrixPool kpool, const SlotParam* __restrict__ slots,
                                                                  const SlotParam* __restrict__ kslots, int n_slots, int n_kslots) {
    const int b = blockIdx.x;                          // uniform per wave: no divergence
    if (b < n_kslots) bd_matrix_build_one<E, true>(kpool, kslots[b], b);          // the longer chains first
    else if (b - n_kslots < n_slots) bd_matrix_build_one<E, false>(pool, slots[b - n_kslots], b - n_kslots);
}

int bd_matrix_max_order() { return 64 * 32; }

template <bool KMAJOR>
static hipError_t launch_layout(const MatrixPool& pool, const SlotParam* d_slots, int n_slots, hipStream_t stream) {
    if (n_slots <= 0) return hipSuccess;
    const int cols = KMAJOR ? pool.n - 1 : pool.n;      // owned columns needed: c = e_base .. n-1
    if (cols > bd_matrix_max_order() || (pool.ld & 1)) return hipErrorInvalidValue;
    dim3 grid(n_slots), block(64);
#define CAFE_BD_CASE(EV)                                                                                          \
    if (cols <= 64 * EV) {                                                                                        \
        (void)hipGetLastError();                                                                                  \
        hipLaunchKernelGGL((bd_matrix_build_kernel<EV, KMAJOR>), grid, block, 0, stream, pool, d_slots, n_slots); \
        return hipGetLastError();                                                                                 \
    }
    CAFE_BD_CASE(2)
    CAFE_BD_CASE(4)
    CAFE_BD_CASE(6)
    CAFE_BD_CASE(8)
    CAFE_BD_CASE(10)
    CAFE_BD_CASE(12)
    CAFE_BD_CASE(14)
    CAFE_BD_CASE(16)
    CAFE_BD_CASE(20)
    CAFE_BD_CASE(24)
    CAFE_BD_CASE(28)
    CAFE_BD_CASE(32)
#undef CAFE_BD_CASE
    return hipErrorInvalidValue;
}

hipError_t launch_bd_matrix_build_both(const MatrixPool& pool, const MatrixPool& kpool, const SlotParam* d_slots, const SlotParam* d_kslots,
                                       int n_slots, int n_kslots, hipStream_t stream) {
    if (n_slots <= 0 || n_kslots <= 0) {               // one layout only: the single-pool launch
        hipError_t e = launch_bd_matrix_build(pool, d_slots, n_slots, stream);
        return e != hipSuccess ? e : launch_bd_matrix_build(kpool, d_kslots, n_kslots, stream);
    }
    const int cols = pool.n;
    if (cols > bd_matrix_max_order() || (pool.ld & 1) || (kpool.ld & 1) || pool.n != kpool.n) return hipErrorInvalidValue;
    dim3 grid(n_slots + n_kslots), block(64);
#define CAFE_BD_CASE(EV)                                                                                                        \
    if (cols <= 64 * EV) {                                                                                                      \
        (void)hipGetLastError();                                                                                                \
        hipLaunchKernelGGL((bd_matrix_build_both_kernel<EV>), grid, block, 0, stream, pool, kpool, d_slots, d_kslots, n_slots, n_kslots); \
        return hipGetLastError();                                                                                               \
    }
    CAFE_BD_CASE(2)
    CAFE_BD_CASE(4)
    CAFE_BD_CASE(6)
    CAFE_BD_CASE(8)
    CAFE_BD_CASE(10)
    CAFE_BD_CASE(12)
    CAFE_BD_CASE(14)
    CAFE_BD_CASE(16)
    CAFE_BD_CASE(20)
    CAFE_BD_CASE(24)
    CAFE_BD_CASE(28)
    CAFE_BD_CASE(32)
#undef CAFE_BD_CASE
    return hipErrorInvalidValue;
}

hipError_t launch_bd_matrix_build(const MatrixPool& pool, const SlotParam* d_slots, int n_slots, hipStream_t stream) {
    return pool.kmajor ? launch_layout<true>(pool, d_slots, n_slots, stream) : launch_layout<false>(pool, d_slots, n_slots, stream);
}

}  // namespace cafe
