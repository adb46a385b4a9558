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
// parent stores C, later children multiply into the parent panel, and ONE leaf sibling can be
// folded in as well (its factor is a gathered column of its matrix, K3's job otherwise).
// Parent size 0 is not part of the GEMM: P[0][c] = delta(c,0) (matrix_cache.cpp:70-77), so that
// row is a copy of the child's row 0, done by the blocks of the first row tile.
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
//
// Two measured facts shape the code (tools/gemm_timeline.py, tools/fp64_probe2.hip):
//  * a wave whose next MFMA is waiting for the matrix pipe holds the SIMD's vector issue port: the
//    co-resident workgroup's VALU instructions crawl (one per ~16 MFMAs).  Everything outside the
//    MFMA stream is therefore written to need almost no VALU: uniform bases in SGPRs + one 32-bit
//    per-lane offset, compile-time specialised epilogues, no per-element branches;
//  * the accumulator array must only be indexed by compile-time constants, or it is demoted to
//    scratch memory (5x slower).  `make check` fails the build if a K2 instantiation uses scratch.
#include <type_traits>

#include "cafe_kernels.h"

namespace cafe {

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int kBStride = kBN + 16;                        // 144 doubles
constexpr int a_stride(int bm) { return (bm % 32 == 16) ? bm : bm + 16; }

// MI: row tile = 16*MI.  MUL: multiply into the parent panel instead of storing.  LEAF: one leaf
// sibling (no error model) is folded into the epilogue.
template <int MI, bool MUL, bool LEAF>
__global__ __launch_bounds__(256, 2) void prune_gemm_kernel(const GemmArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)   // the body uses amdgcn-only types (buffer resource); hipcc's host pass only needs the stub
    constexpr int BM = 16 * MI;
    constexpr int SA = a_stride(BM);
    constexpr int A_TILE = kBK * SA, B_TILE = kBK * kBStride;
    __shared__ double lds[2 * (A_TILE + B_TILE)];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // scalar: LDS-DMA bases (M0) need no VALU
    const int cat = blockIdx.z;
    unsigned long long st0 = 0, st1 = 0, st2 = 0;
    if (a.stamps) st0 = __builtin_amdgcn_s_memrealtime();

    // XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch, a speed-only
    // assumption), so XCD x takes the column tiles x, x+8, ... and runs their row tiles back to back:
    // the row tiles of one column tile then share the child panel (B) in that XCD's L2.
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int row_tile = idx % a.n_row_tiles;
    const int col_tile = xcd + 8 * (idx / a.n_row_tiles);
    if (col_tile >= a.n_col_tiles) return;
    const int row0 = row_tile * BM;                        // parent size row0 + 1 is the tile's first row
    const int col0 = col_tile * kBN;

    const double* __restrict__ B = a.src + (int64_t)cat * a.panel_kstride + col0;
    double* __restrict__ C = a.dst + (int64_t)cat * a.panel_kstride + col0;
    const int lda = a.pool.ld;
    const int ldb = a.ld;

    // LDS-DMA fill of one tile pair, issued in four quarters (one per k-step) so that the issue slots
    // fall in the shadow of the MFMAs.  16 slots = 4 quarters x 4 waves.  B: k-row `slot` is one 1 KB piece.
    // A: with SA == BM the [16][BM] image is contiguous, 2*MI pieces of 1 KB laid end to end (a piece may
    // span two k-rows; the per-lane source address makes that free); slot s moves pieces s, s+16, ...
    // and wraps, re-writing an identical piece rather than branching.  With a padded row (even MI) each
    // k-row is moved on its own with the tail lanes masked off.
    constexpr bool A_CONTIG = (SA == BM);
    constexpr int NP = 2 * MI, PER = (NP + 15) / 16;
    // Buffer addressing (buffer_load_dwordx4 ... lds): resource descriptor in SGPRs, one 32-bit per-lane
    // byte offset that never changes, and a scalar byte offset per piece -- the DMA issue needs no VALU at all.
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.pool.base + (int64_t)a.slot[cat] * a.pool.stride), 0, (int)(a.pool.stride * 8), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.src + (int64_t)cat * a.panel_kstride), 0, (int)(a.panel_kstride * 8 > 0xFFFFFFF0ll ? 0xFFFFFFF0ll : a.panel_kstride * 8), 0x00020000);
    unsigned a_voff[4][PER > 0 ? PER : 1];       // bytes
    int a_dst[4][PER > 0 ? PER : 1];
    if (A_CONTIG) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                int piece = q * 4 + wave + 16 * j;             // scalar
                if (piece >= NP) piece -= NP;
                // element e = 128*piece + 2*lane of the [16][BM] image -> (k-row, column).  128*piece splits on
                // the scalar unit; adding 2*lane (<= 126 < 2*BM as BM >= 64) wraps at most twice: a few VALU
                // instead of a vector div/mod
                const int r0s = (piece * 128) / BM, c0s = (piece * 128) % BM;
                int c = c0s + lane * 2, r = r0s;
                if (c >= BM) { c -= BM; r += 1; }
                if (c >= BM) { c -= BM; r += 1; }
                a_voff[q][j] = (unsigned)((r * lda + c + row0) * 8);
                a_dst[q][j] = piece * 128;
            }
    }
    const unsigned a_voff_row = (unsigned)((row0 + lane * 2) * 8);
    const unsigned b_voff = (unsigned)((col0 + lane * 2) * 8);
    auto stage_quarter = [&](int k0, int buf, int q) {
        double* As = lds + buf * (A_TILE + B_TILE);
        double* Bs = As + A_TILE;
        const int krow = q * 4 + wave;
        if (A_CONTIG) {
            const int soff = k0 * lda * 8;
#pragma unroll
            for (int j = 0; j < PER; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lptr_t)(As + a_dst[q][j]), 16, a_voff[q][j], soff, 0, 0);
        } else {
            constexpr int nl = BM >= 128 ? 64 : BM / 2;
            if (nl == 64 || lane < nl)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lptr_t)(As + krow * SA), 16, a_voff_row, (k0 + krow) * lda * 8, 0, 0);
        }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lptr_t)(Bs + krow * kBStride), 16, b_voff, (k0 + krow) * ldb * 8, 0, 0);
    };

    const int l15 = lane & 15, l4 = lane >> 4;
    const int n_k = (a.k_valid + kBK - 1) / kBK;
    const int a_off = l4 * SA + l15;
    const int b_off = A_TILE + l4 * kBStride + wave * 32 + l15;

#pragma unroll
    for (int q = 0; q < 4; ++q) stage_quarter(0, 0, q);
    double4_t acc[MI][2];                                  // zeroed while the first tile is in flight
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        acc[i][0] = double4_t{0.0, 0.0, 0.0, 0.0};
        acc[i][1] = double4_t{0.0, 0.0, 0.0, 0.0};
    }
    __syncthreads();                                       // vmcnt(0) + barrier: tile 0 has landed
    if (a.stamps) st1 = __builtin_amdgcn_s_memrealtime();

    // One K tile (rows c > M of the k-major matrix are zero, so padded k inside a step adds exact zeros;
    // whole padded k-steps of the last tile are skipped).  Before the MFMAs of step s are issued, the DMA quarter of
    // the next tile and the fragment reads of step s+1 (second register set) are already in flight, so a
    // workgroup that has the SIMD to itself (its partner in prologue/epilogue) does not expose the LDS
    // latency four times per tile.
    double af[2][MI], bf[2][2];
    auto load_frags = [&](const double* base, int s4, int set) {
#pragma unroll
        for (int i = 0; i < MI; ++i) af[set][i] = base[a_off + s4 * 4 * SA + i * 16];
        bf[set][0] = base[b_off + s4 * 4 * kBStride];
        bf[set][1] = base[b_off + s4 * 4 * kBStride + 16];
    };
    auto tile = [&](int kt, auto stage_next, int n_steps) {
        const int buf = kt & 1;
        const double* cur = lds + buf * (A_TILE + B_TILE);
        load_frags(cur, 0, 0);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            if (!decltype(stage_next)::value && s4 >= n_steps) break;      // last tile: skip all-padding k-steps
            if (decltype(stage_next)::value) stage_quarter((kt + 1) * kBK, buf ^ 1, s4);
            if (s4 < 3) load_frags(cur, s4 + 1, (s4 + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                acc[i][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[s4 & 1][i], bf[s4 & 1][0], acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[s4 & 1][i], bf[s4 & 1][1], acc[i][1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();                                   // next tile landed, this one fully read
    };
    for (int kt = 0; kt + 1 < n_k; ++kt) tile(kt, std::true_type{}, 4);
    tile(n_k - 1, std::false_type{}, (a.k_valid - (n_k - 1) * kBK + 3) / 4);
    if (a.stamps) st2 = __builtin_amdgcn_s_memrealtime();

    // ---- epilogue.  C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg,
    // i.e. a lane holds single columns.  Each 16-row block of the wave's 16*MI x 32 sub-tile goes through a
    // small private LDS image so that a lane then owns two adjacent columns of one row: stores (and the
    // loads of the multiply mode) are 16 B per lane, 256 B contiguous per row.  All addresses are
    // `uniform base + per-lane 32-bit offset`; the old panel values (MUL) and the leaf factor (LEAF) of
    // row block i+1 are loaded before block i is staged and stored.
    constexpr int SS = 34;                                  // staging row stride (doubles)
    double* stg = lds + wave * (16 * SS);
    const int e_row = lane >> 4, e_col = (lane & 15) * 2;   // row inside a group of 4, first of two columns
    const int gcol = wave * 32 + e_col;                     // column inside the block tile
    typedef int int4_t __attribute__((ext_vector_type(4)));
    typedef int int2_t __attribute__((ext_vector_type(2)));
    const int kbytes = (int)(a.panel_kstride * 8 > 0xFFFFFFF0ll ? 0xFFFFFFF0ll : a.panel_kstride * 8);
    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc((void*)(a.dst + (int64_t)cat * a.panel_kstride), 0, kbytes, 0x00020000);
    const unsigned c_voff = (unsigned)((e_row * ldb + col0 + gcol) * 8);       // per-lane byte offset, fixed
    const int c_soff0 = (row0 + a.out_off) * ldb * 8;                          // scalar: tile's first output row
    const int ldl = a.lpool.ld;
    __amdgpu_buffer_rsrc_t rsL = rsC;
    unsigned l_voff0 = 0, l_voff1 = 0;
    int l_soff0 = 0;
    if (LEAF) {
        const int32_t* cnt = a.counts + (int64_t)a.taxon[0] * a.counts_ld + a.f0 + col0 + gcol;
        rsL = __builtin_amdgcn_make_buffer_rsrc((void*)(a.lpool.base + (int64_t)a.leaf_slot[0][cat] * a.lpool.stride), 0,
                                                (int)(a.lpool.stride * 8), 0x00020000);
        l_voff0 = (unsigned)((e_row * ldl + cnt[0]) * 8);
        l_voff1 = (unsigned)((e_row * ldl + cnt[1]) * 8);
        l_soff0 = (row0 + 1) * ldl * 8;                                        // parent size row0 + 1
    }
    const int rows_here = a.rows - row0;                    // valid rows of this tile (>= BM for interior tiles)
    struct Pre { double2 f[4]; };
    // FULL: the tile has all BM rows (always true when 16*MI divides the row count): no per-row masks
    auto prefetch = [&](int i, Pre& p, auto full) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int step = i * 16 + q * 4;                // uniform
            double2 f = make_double2(1.0, 1.0);
            if (decltype(full)::value || step + e_row < rows_here) {
                if (LEAF) {
                    f.x = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rsL, l_voff0, l_soff0 + step * ldl * 8, 0));
                    f.y = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rsL, l_voff1, l_soff0 + step * ldl * 8, 0));
                }
                if (MUL) {
                    const double2 old = __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(rsC, c_voff, c_soff0 + step * ldb * 8, 0));
                    f.x *= old.x;
                    f.y *= old.y;
                }
            }
            p.f[q] = f;
        }
    };
    auto flush = [&](int i, const Pre& p, auto full) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) stg[(l4 + 4 * r) * SS + j * 16 + l15] = acc[i][j][r];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            double2 v = *reinterpret_cast<const double2*>(&stg[(q * 4 + e_row) * SS + e_col]);
            const int step = i * 16 + q * 4;
            if (decltype(full)::value || step + e_row < rows_here) {
                if (MUL || LEAF) {
                    v.x *= p.f[q].x;
                    v.y *= p.f[q].y;
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(int4_t, v), rsC, c_voff, c_soff0 + step * ldb * 8, 0);
            }
        }
    };
    auto epilogue = [&](auto full) {
        Pre pre[2];
        if (MUL || LEAF) prefetch(0, pre[0], full);
        // full unroll is mandatory: a runtime i would index the accumulator array dynamically and demote
        // it to scratch memory
#pragma clang loop unroll(full)
        for (int i = 0; i < MI; ++i) {
            if ((MUL || LEAF) && i + 1 < MI) prefetch(i + 1, pre[(i + 1) & 1], full);
            flush(i, pre[i & 1], full);
        }
    };
    if (rows_here >= BM) epilogue(std::true_type{});
    else epilogue(std::false_type{});
    // parent size 0 only reaches child size 0 (P[0][c] = delta(c,0)): that panel row is the child's row 0,
    // times the leaf sibling's P_leaf[0][x] = delta(x,0)
    if (a.out_off == 1 && row_tile == 0 && tid < kBN / 2) {
        const int c2 = tid * 2;
        double2 v = *reinterpret_cast<const double2*>(B + c2);
        if (LEAF) {
            const int32_t* cnt = a.counts + (int64_t)a.taxon[0] * a.counts_ld + a.f0 + col0 + c2;
            v.x = cnt[0] == 0 ? v.x : 0.0;
            v.y = cnt[1] == 0 ? v.y : 0.0;
        }
        double2* dst = reinterpret_cast<double2*>(C + c2);
        if (MUL) {
            const double2 old = *dst;
            v.x *= old.x;
            v.y *= old.y;
        }
        *dst = v;
    }
    if (a.stamps && tid == 0) {             // diagnostic build only: per-block timeline (100 MHz ticks) + placement
        const unsigned long long st3 = __builtin_amdgcn_s_memrealtime();
        unsigned long long* o = a.stamps + 6 * ((size_t)blockIdx.z * gridDim.x + blockIdx.x);
        o[0] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));      // HW_REG_HW_ID
        o[1] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));     // HW_REG_XCC_ID
        o[2] = st0; o[3] = st1; o[4] = st2; o[5] = st3;
    }
#else
    (void)a;
#endif
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

template <int MI>
static void launch_mi(const GemmArgs& a, dim3 grid, hipStream_t stream) {
    const dim3 block(256);
    if (a.mode) {
        if (a.n_leaf) hipLaunchKernelGGL((prune_gemm_kernel<MI, true, true>), grid, block, 0, stream, a);
        else hipLaunchKernelGGL((prune_gemm_kernel<MI, true, false>), grid, block, 0, stream, a);
    } else {
        if (a.n_leaf) hipLaunchKernelGGL((prune_gemm_kernel<MI, false, true>), grid, block, 0, stream, a);
        else hipLaunchKernelGGL((prune_gemm_kernel<MI, false, false>), grid, block, 0, stream, a);
    }
}

hipError_t launch_prune_gemm(const GemmArgs& a, int n_categories, hipStream_t stream) {
    if (a.n_leaf > 1 || (a.n_leaf == 1 && a.err != nullptr)) return hipErrorInvalidValue;   // the schedule never asks for it
    dim3 grid(8 * ((a.n_col_tiles + 7) / 8) * a.n_row_tiles, 1, n_categories);
    (void)hipGetLastError();
    switch (a.mi) {
        case 4: launch_mi<4>(a, grid, stream); break;
        case 5: launch_mi<5>(a, grid, stream); break;
        case 6: launch_mi<6>(a, grid, stream); break;
        case 7: launch_mi<7>(a, grid, stream); break;
        case 8: launch_mi<8>(a, grid, stream); break;
        case 9: launch_mi<9>(a, grid, stream); break;
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
