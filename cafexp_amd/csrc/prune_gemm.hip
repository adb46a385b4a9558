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
#include <algorithm>
#include <type_traits>

#include <hip/hip_ext.h>

#include "cafe_kernels.h"

// Measurement only: built with -D'CAFE_EXPERIMENT_B_COLUMN(c)=0' every workgroup stages column tile 0 of
// the child panel, so that every B fetch after the first is an L2 hit -- wrong results, the same work: what the refetches of
// the child panel cost (0.9 % of an every-K-tile call).
// Likewise for the matrix tiles: -D'CAFE_EXPERIMENT_A_SLOT(s)=0' -D'CAFE_EXPERIMENT_A_ROW(r)=0' makes every workgroup stage rows of
// ONE matrix at row tile 0 (wrong results, same work): what the L2 misses of the A stream cost.
#ifndef CAFE_EXPERIMENT_A_SLOT
#define CAFE_EXPERIMENT_A_SLOT(s) (s)
#endif
#ifndef CAFE_EXPERIMENT_A_ROW
#define CAFE_EXPERIMENT_A_ROW(r) (r)
#endif
#ifndef CAFE_EXPERIMENT_B_COLUMN
#define CAFE_EXPERIMENT_B_COLUMN(c) (c)
#endif
// Cache policy of K2's three streams (the aux operand of the buffer instructions: 1 = sc0, 2 = nt, 16 = sc1): the k-major
// matrix tiles (A), the child panel's K tiles (B, LDS-DMA both) and the stores of the output panel (C).  Measured at config 4
// (one process per build, K2 per call): everything 0 117.4-117.8 ms; B loads nt 122.3; A loads nt
// 128.1 (both streams live on their L2 hits); C stores nt 117.0-117.5; C stores nt + sc1 116.9; C stores sc1 117.7.  The
// output panel is written once and read again a launch later from HBM whatever the policy, so the variants with 8-deep
// K tiles (matrix orders >= 256, where a panel is far larger than the L2s) store it non-temporal and leave the L2s to A and
// B; the small-order variants keep the default (their whole panel fits the L2s and the next launch reads it from there).
#ifndef CAFE_K2_A_LOAD_AUX
#define CAFE_K2_A_LOAD_AUX 0
#endif
#ifndef CAFE_K2_B_LOAD_AUX
#define CAFE_K2_B_LOAD_AUX 0
#endif
#ifndef CAFE_K2_MUL_LOAD_AUX
#define CAFE_K2_MUL_LOAD_AUX (KB == 8 ? 2 : 0)   // the parent panel's old values in multiply mode: read once, then overwritten (116.62 -> 116.40 ms, three alternating runs)
#endif
#ifndef CAFE_K2_C_STORE_AUX
#define CAFE_K2_C_STORE_AUX (KB == 8 ? 2 : 0)
#endif

namespace cafe {

typedef double double4_t __attribute__((ext_vector_type(4)));
// the op descriptors and tile lists are read-only for the launch: constant address space, so that a wave-uniform access is
// a scalar load whatever the stores around it
typedef const __attribute__((address_space(4))) GemmOp* op_cptr_t;
typedef const __attribute__((address_space(4))) long long* plan_cptr_t;
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// Row stride of the B tile in LDS.  144 doubles (= 16 mod 32) puts k and k+1 on opposite bank halves: conflict-free fragment
// reads.  Row tiles of up to 80 rows drop the padding: the stage shrinks from 28 to 26 KB, THREE workgroups fit a CU's 160 KB,
// and the 4-way conflict on the two B fragment reads of a k-step (16 instead of 4 LDS clocks each, against 640 clocks of
// MFMA issue) is hidden behind the third wave of every SIMD.
constexpr int b_stride(int mi) { return mi <= 5 ? kBN : kBN + 16; }
constexpr int a_stride(int bm) { return (bm % 32 == 16) ? bm : bm + 16; }

// MI: row tile = 16*MI.  MUL: multiply into the parent panel instead of storing.  LEAF: what else the epilogue folds in --
// 1 or 3 (= taps): one leaf sibling (its factor is a gathered column of its matrix, or the three error-model taps);
// 2: the factor panel of an interior sibling that has fewer distinct columns than the parent, gathered through the
// parent->sibling column map (subtree-level de-duplication): the parent's panel is then complete after this launch,
// without an assemble pass.  TRANS: a factor GEMM (plain store over the CHILD's columns) writes its result transposed,
// F_T[column][16 - out_off + panel row] with the rows contiguous: whoever spreads that factor over the parent's columns
// (K3's assemble pass, the LEAF == 2 epilogue) then fetches whole 128-byte lines per mapped column instead of one
// 8-byte element per line.  The +16 puts every 16-row block of the tile on a line boundary.  Such a launch issues its
// MFMAs with the two operands swapped: the fragments of v_mfma_f64_16x16x4_f64 are symmetric (16 along lane & 15, 4
// along lane >> 4), so D' = B^T A^T = C^T lands in the accumulators -- lane & 15 is then the ROW of C and (lane >> 4) + 4
// reg its column, and a store instruction writes four columns x 16 consecutive rows = four full 128-byte lines of the
// transposed factor, as wide as the row-major store of the other variants (32-byte pieces measured 8 % slower).
template <int KB, int MI, bool MUL, int LEAF, bool TRANS = false>
__global__ __launch_bounds__(256, prune_gemm_wg_per_cu(MI, KB)) void prune_gemm_kernel(const GemmArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)   // the body uses amdgcn-only types (buffer resource); hipcc's host pass only needs the stub
    constexpr int BM = 16 * MI;
    constexpr int SA = a_stride(BM);
    constexpr int kBStride = b_stride(MI);
    constexpr int A_TILE = KB * SA, B_TILE = KB * kBStride, STAGE = A_TILE + B_TILE;
    __shared__ double lds[2 * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // scalar: LDS-DMA bases (M0) need no VALU
    const int l15 = lane & 15, l4 = lane >> 4;
    constexpr bool GATH = LEAF == 2;
    const int lda = a.pool.ld;
    const op_cptr_t ops = (op_cptr_t)(unsigned long long)a.ops;
    unsigned long long st0 = 0, ep_ticks = 0, n_done = 0, loop_ticks = 0, kt_done = 0, t_tile = 0;
    if (a.stamps) st0 = __builtin_amdgcn_s_memrealtime();

    // ---- persistent tile loop.  The grid is 2 workgroups per CU; a workgroup walks a fixed list of output
    // tiles so that the first K tile of the next output tile is already being fetched while the current one
    // is finished and stored: no per-tile launch gap or cold prologue.  XCD-aware order (speed only): blocks
    // b and b+8 share an XCD under round-robin dispatch, so XCD x owns the (category, column tile) pairs
    // x, x+8, ... (balanced to within one pair whatever the shard size) and its 64 blocks take consecutive
    // (pair, row tile) couples, row tile fastest: the row tiles of one column tile run together and share the
    // child panel (B) in that L2.
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3, n_local_blocks = gridDim.x >> 3;

    // per-lane constants that do not depend on the tile (tile origins travel in the scalar offsets)
    constexpr bool A_CONTIG = (SA == BM);
    constexpr int NS = KB / 4;                    // k-steps of a K tile = DMA "quarters" of a stage (4 k-rows each, one per wave)
    constexpr int SLOTS = 4 * NS;
    constexpr int NP = KB * BM / 128, PER = (NP + SLOTS - 1) / SLOTS;     // 1 KB pieces of the contiguous [KB][BM] A image
    unsigned a_voff[NS][PER > 0 ? PER : 1];      // bytes
    int a_dst[NS][PER > 0 ? PER : 1];
    if (A_CONTIG) {
#pragma unroll
        for (int q = 0; q < NS; ++q)
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                int piece = q * 4 + wave + SLOTS * j;          // scalar
                while (piece >= NP) piece -= NP;               // (scalar; at most twice: NP >= 6)
                // element e = 128*piece + 2*lane of the [16][BM] image -> (k-row, column).  128*piece splits on
                // the scalar unit; adding 2*lane (<= 126 < 2*BM as BM >= 64) wraps at most twice
                const int r0s = (piece * 128) / BM, c0s = (piece * 128) % BM;
                int c = c0s + lane * 2, r = r0s;
                if (c >= BM) { c -= BM; r += 1; }
                if (c >= BM) { c -= BM; r += 1; }
                if (BM < 64 && c >= BM) { c -= BM; r += 1; }        // (48-row tiles: 126 < 3 * 48)
                a_voff[q][j] = (unsigned)((r * lda + c) * 8);
                a_dst[q][j] = piece * 128;
            }
    }
    const unsigned lane16 = (unsigned)(lane * 16);
    const int a_off = l4 * SA + l15;
    const int b_off = A_TILE + l4 * kBStride + wave * 32 + l15;
    const int n_k = (a.k_valid + KB - 1) / KB;
    const int last_steps = (a.k_valid - (n_k - 1) * KB + 3) / 4;
    auto span_bytes = [](int64_t doubles) -> int { return (int)(doubles * 8 > 0xFFFFFFF0ll ? 0xFFFFFFF0ll : doubles * 8); };

    // Tile descriptor (all scalar): the op it belongs to, buffer resources of the A matrix / child panel / parent panel and
    // origins.  K tiles [kt0, kt0 + nkt) are the ones inside (matrix extent of the row tile) x (panel extent of the column
    // tile) -- worked out by the tile planner: far from the diagonal the matrix entries underflow to exact zeros, a likelihood
    // column is exactly zero far from the observed sizes, and a K tile of zeros adds nothing: skipping it leaves every
    // accumulator bit as it is and saves its MFMAs and the B rows it would have staged.  (An all-zero tile still runs one
    // K tile: the panel must receive its zeros.)
    struct Tile {
        __amdgpu_buffer_rsrc_t rsA, rsB, rsC;
        op_cptr_t o;
        int cat, row0, col0, row_tile, ldb;
        int kt0, nkt;
    };
    // The workgroup's tiles: entry i of its planned list (tile_plan_kernel: x = op << 24 | index in the op's list for this
    // XCD, y = first K tile << 16 | K tiles); y == 0 ends the list.  All scalar.
    const plan_cptr_t mylist = (plan_cptr_t)(unsigned long long)(a.plan + ((int64_t)xcd * n_local_blocks + local) * a.plan_rounds);
    auto entry = [&](int i) -> int2 {
        if (i >= a.plan_rounds) return make_int2(0, 0);
        const long long e = mylist[i];                      // int2 {x, y}: x in the low word
        return make_int2(__builtin_amdgcn_readfirstlane((int)(e & 0xFFFFFFFFll)), __builtin_amdgcn_readfirstlane((int)(e >> 32)));
    };
    auto decode = [&](int2 en) -> Tile {
        Tile x;
        x.o = ops + (en.x >> 24);
        const int t = en.x & 0xFFFFFF;
        const int nrt = x.o->n_row_tiles;
        x.ldb = a.uniform_ld > 0 ? a.uniform_ld : x.o->ld;
        const int nct = x.ldb / kBN;
        x.row_tile = t % nrt;
        const int pair = xcd + 8 * (t / nrt);
        const int ct = pair % nct;
        x.cat = pair / nct;
        x.row0 = x.row_tile * BM;
        x.col0 = ct * kBN;
        x.rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(a.pool.base + (int64_t)CAFE_EXPERIMENT_A_SLOT(x.o->slot[x.cat]) * a.pool.stride), 0, (int)(a.pool.stride * 8), 0x00020000);
        x.rsB = __builtin_amdgcn_make_buffer_rsrc((void*)(x.o->src + (int64_t)x.cat * x.o->src_kstride), 0, span_bytes(x.o->src_kstride), 0x00020000);
        x.rsC = __builtin_amdgcn_make_buffer_rsrc((void*)(x.o->dst + (int64_t)x.cat * x.o->dst_kstride), 0, span_bytes(x.o->dst_kstride), 0x00020000);
        x.kt0 = en.y >> 16;
        x.nkt = en.y & 0xFFFF;
        return x;
    };
    // LDS-DMA fill of K tile k0 of output tile x into stage `buf`, quarter q (16 slots = 4 quarters x 4 waves).
    // B: k-row `slot` is one 1 KB piece.  A: with SA == BM the [16][BM] image is contiguous, 2*MI pieces of 1 KB
    // laid end to end (a piece may span two k-rows); slot s moves pieces s, s+16, ... and wraps, re-writing an
    // identical piece rather than branching.  With a padded row (even MI) each k-row is moved on its own with
    // the tail lanes masked off.  buffer_load ... lds: descriptor + scalar offset + fixed lane offset, no VALU.
    auto stage_quarter = [&](const Tile& x, int k0, int buf, int q) {
        double* As = lds + buf * STAGE;
        double* Bs = As + A_TILE;
        const int krow = q * 4 + wave;
        if (A_CONTIG) {
            const int soff = (k0 * lda + CAFE_EXPERIMENT_A_ROW(x.row0)) * 8;
#pragma unroll
            for (int j = 0; j < PER; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(x.rsA, (lptr_t)(As + a_dst[q][j]), 16, a_voff[q][j], soff, 0, CAFE_K2_A_LOAD_AUX);
        } else {
            constexpr int nl = BM >= 128 ? 64 : BM / 2;
            if (nl == 64 || lane < nl)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(x.rsA, (lptr_t)(As + krow * SA), 16, lane16, ((k0 + krow) * lda + CAFE_EXPERIMENT_A_ROW(x.row0)) * 8, 0, CAFE_K2_A_LOAD_AUX);
        }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(x.rsB, (lptr_t)(Bs + krow * kBStride), 16, lane16, ((k0 + krow) * x.ldb + CAFE_EXPERIMENT_B_COLUMN(x.col0)) * 8, 0, CAFE_K2_B_LOAD_AUX);
    };

    // Which of the tile's MI row blocks take part in K tile kt of its range: block i only inside ITS OWN matrix extent (the tile's
    // range is the hull of them, cut by the panel's extent).  The band of a transition matrix moves down by 16 rows per block, so
    // the first K tiles of a range meet only the upper blocks and the last ones only the lower blocks -- a tenth of the MFMAs of
    // a config-4 call multiply a zero block of A; leaving them out changes no accumulator bit.  All scalar.
    typedef const __attribute__((address_space(4))) int32_t* ext_cptr_t;
    struct Blocks { int lo[MI], hi[MI]; };                  // K tiles relative to the tile's first one; lo > hi: never
    auto block_ranges = [&](const Tile& x) -> Blocks {
        Blocks b;
        if (!a.pool.ext) {
#pragma unroll
            for (int i = 0; i < MI; ++i) { b.lo[i] = 0; b.hi[i] = 0x7fff; }
            return b;
        }
        const ext_cptr_t e = (ext_cptr_t)(unsigned long long)(a.pool.ext + ((int64_t)x.o->slot[x.cat] * a.pool.ext_blocks + x.row_tile * MI) * 2);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            int lo = 1, hi = 0;
            if (x.row_tile * MI + i < a.pool.ext_blocks) { lo = e[2 * i]; hi = e[2 * i + 1]; }
            const bool some = hi >= lo;
            b.lo[i] = some ? lo / KB - x.kt0 : 1;
            b.hi[i] = some ? hi / KB - x.kt0 : 0;
        }
        return b;
    };
    auto active = [&](const Blocks& b, int kt) -> unsigned {
        unsigned m = 0;
#pragma unroll
        for (int i = 0; i < MI; ++i) m |= (unsigned)(kt >= b.lo[i] && kt <= b.hi[i]) << i;
        return m;
    };

    int2 e_cur = entry(0);
    if (e_cur.y == 0) return;
    {
        const Tile first = decode(e_cur);
#pragma unroll
        for (int q = 0; q < NS; ++q) stage_quarter(first, first.kt0 * KB, 0, q);
    }
    int g = 0;                                              // running K-tile count: stage parity
    __syncthreads();                                        // vmcnt(0) + barrier: the first K tile has landed (later
                                                            // ones land behind the barrier that ends each K tile)

    for (int ti = 0;; ++ti) {
        const int2 e_nxt = entry(ti + 1);
        const bool has_next = e_nxt.y != 0;
        const Tile cur = decode(e_cur);                     // scalar work, once per output tile; cheaper than carrying it
        const Tile nxt = decode(has_next ? e_nxt : e_cur);  // only its A/B descriptors and origins are used (DMA)

        if (a.stamps) t_tile = __builtin_amdgcn_s_memrealtime();
        double4_t acc[MI][2];                               // zeroed while the first K tile is in flight
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            acc[i][0] = double4_t{0.0, 0.0, 0.0, 0.0};
            acc[i][1] = double4_t{0.0, 0.0, 0.0, 0.0};
        }

        // ---- main loop.  Rows c > M of the k-major matrix are zero, so padded k inside a step adds exact zeros; whole
        // padded k-steps of the last K tile are skipped.  The full K tiles are software-pipelined at the k-step level
        // with ONE fragment register set: the A fragment of row block i for step s+1 is read right after the two MFMAs
        // that consume it in step s (MFMAs read their operands at issue), the two B fragments alternate between two
        // pairs.  A wave alone on its SIMD then keeps the matrix pipe fed (tools/fp64_probe3: 78 TFLOP/s), where a
        // "read everything, wait, 18 MFMAs" step leaves a bubble of one LDS latency per step (an ablated copy of this
        // kernel without DMA, barriers and epilogue: 71.8 TFLOP/s).  The barrier that hands over a stage sits in front
        // of a K tile's last step, after that step's fragments were read, so that the same step can already read the
        // next tile's first fragments; the DMA quarters of a K tile go out in the four steps before that barrier.
        double af[MI], bfr[2][2];
        auto read_b = [&](const double* base, int s4, double (&b)[2]) {
            b[0] = base[b_off + s4 * 4 * kBStride];
            b[1] = base[b_off + s4 * 4 * kBStride + 16];
        };
        {   // cold start of the output tile: its K tile 0 has landed (prologue / the barrier that ended the previous tile)
            const double* base = lds + (g & 1) * STAGE;
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = base[a_off + i * 16];
            read_b(base, 0, bfr[0]);
        }
        const Blocks blk = block_ranges(cur);
        const int nkt = cur.nkt, kbase = cur.kt0 * KB;     // this tile's K tiles: kbase, kbase + 16, ...
        const int steps_last = cur.kt0 + nkt == n_k ? last_steps : NS;  // only the matrix's last K tile is ragged
        for (int kt = 0; kt + 1 < nkt; ++kt) {
            const int buf = g & 1;
            const double* base = lds + buf * STAGE;
            const double* nbase = lds + (buf ^ 1) * STAGE;
            const int k1 = kbase + (kt + 1) * KB;          // K tile being staged into the other stage
            const bool next_full = kt + 2 < nkt;            // K tile kt+1 is another pipelined one (not this tile's last)
            const unsigned on = active(blk, kt);
#pragma unroll
            for (int s4 = 0; s4 < NS; ++s4) {
                if (s4 < NS - 1) {
                    if (s4 == 0 && kt == 0) {               // K tile 1 of this output tile: no earlier barrier to carry it
#pragma unroll
                        for (int q = 0; q < NS; ++q) stage_quarter(cur, k1, buf ^ 1, q);
                    }
                } else {
                    // K tile kt+1 has landed; every wave has read all of K tile kt (the fragments of its last step are in
                    // registers), so ALL of K tile kt+2 goes out here, a whole K tile ahead of the barrier that needs it.
                    // (Until the middle of round 3 only its first quarter left here and the others one step before they
                    // were needed: with 8-deep K tiles that was a lead of one k-step -- 0.4 us of a wave's time where only two
                    // of a tile's five row blocks are live, less than a fetch from the Infinity Cache takes.)
                    __syncthreads();
                    if (next_full) {
#pragma unroll
                        for (int q = 0; q < NS; ++q) stage_quarter(cur, k1 + KB, buf, q);
                    } else if (has_next) {
#pragma unroll
                        for (int q = 0; q < NS; ++q) stage_quarter(nxt, nxt.kt0 * KB, buf, q);
                    }
                }
                const bool pre = s4 < NS - 1 || next_full;  // uniform
                const double* src = s4 < NS - 1 ? base : nbase;
                const int ns = (s4 + 1) % NS;
                if (pre) read_b(src, ns, bfr[(s4 + 1) & 1]);
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    if (on & (1u << i)) {
                        if (TRANS) {                        // operands swapped: the accumulators hold C^T
                            acc[i][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(bfr[s4 & 1][0], af[i], acc[i][0], 0, 0, 0);
                            acc[i][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(bfr[s4 & 1][1], af[i], acc[i][1], 0, 0, 0);
                        } else {
                            acc[i][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bfr[s4 & 1][0], acc[i][0], 0, 0, 0);
                            acc[i][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bfr[s4 & 1][1], acc[i][1], 0, 0, 0);
                        }
                    }
                    if (pre) af[i] = src[a_off + ns * 4 * SA + i * 16];
                    __builtin_amdgcn_sched_barrier(0);      // keep "two MFMAs, then the read that reuses their register"
                }
            }
            ++g;
        }
        {   // last K tile: last_steps of its four steps hold valid k; plain read -> MFMA steps.  The next output tile's
            // first K tile is in flight (it went out at the barrier of the K tile before, or goes out here if this is the only one).
            const int buf = g & 1;
            const double* base = lds + buf * STAGE;
            if (has_next && nkt == 1) {                     // (otherwise it went out at the barrier of the K tile before)
                const int nk0 = nxt.kt0 * KB;
#pragma unroll
                for (int q = 0; q < NS; ++q) stage_quarter(nxt, nk0, buf ^ 1, q);
            }
            const unsigned on = active(blk, nkt - 1);
#pragma unroll
            for (int s4 = 0; s4 < NS; ++s4) {
                if (s4 < steps_last) {
#pragma unroll
                    for (int i = 0; i < MI; ++i) af[i] = base[a_off + s4 * 4 * SA + i * 16];
                    read_b(base, s4, bfr[0]);
#pragma unroll
                    for (int i = 0; i < MI; ++i) {
                        if (!(on & (1u << i))) continue;
                        if (TRANS) {
                            acc[i][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(bfr[0][0], af[i], acc[i][0], 0, 0, 0);
                            acc[i][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(bfr[0][1], af[i], acc[i][1], 0, 0, 0);
                        } else {
                            acc[i][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bfr[0][0], acc[i][0], 0, 0, 0);
                            acc[i][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bfr[0][1], acc[i][1], 0, 0, 0);
                        }
                    }
                }
            }
            __syncthreads();                                // the next output tile's K tile 0 landed, this one fully read
            ++g;
        }

        // ---- epilogue.  C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg.  A store
        // instruction of one accumulator register therefore writes 4 rows x 16 columns = four full 128-byte lines.
        // All accesses are buffer operations: descriptor + SCALAR offset (row block, register row group, column
        // half) + one fixed per-lane offset -- no address VALU and no LDS round trip; the old panel values (MUL) and
        // the leaf factors (LEAF) of row block i+1 are loaded before block i is stored.
        // TRANS (accumulators hold C^T): lane & 15 = row inside the 16-row block, (lane >> 4) + 4 reg = column inside the 16
        const op_cptr_t o = cur.o;
        const int ldb = cur.ldb, ldt = o->dst_ldt;          // ldt (TRANS): rows per column of the transposed factor
        const int ldl = GATH ? (int)o->gath_ld : a.lpool.ld;   // stride of what the epilogue gathers
        const int out_off = o->out_off;
        const unsigned c_voff = TRANS ? (unsigned)(((wave * 32 + l4) * ldt + l15) * 8) : (unsigned)((l4 * ldb + wave * 32 + l15) * 8);
        const int c_soff0 = TRANS ? (cur.col0 * ldt + cur.row0 + 16) * 8 : ((cur.row0 + out_off) * ldb + cur.col0) * 8;
        __amdgpu_buffer_rsrc_t rsL = cur.rsC;
        constexpr int NT = LEAF == 3 ? 3 : 1;               // taps of the leaf sibling: 1, or the 3 of an error model
        unsigned l_voff[2][NT];
        double l_w[2][NT];                                  // error-model weights (taps outside [0, M]: weight 0, clamped column)
        int l_soff0 = 0;
        if (GATH) {
            // sibling factor F[s][map[column]]: the same access shape as a leaf's matrix column, another base and stride
            const int32_t* mp = o->gath_map + cur.col0 + wave * 32 + l15;
            rsL = __builtin_amdgcn_make_buffer_rsrc((void*)(o->gath_src + (int64_t)cur.cat * o->gath_kstride), 0, span_bytes(o->gath_kstride), 0x00020000);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                l_voff[j][0] = (unsigned)((mp[16 * j] * ldl + l4) * 8);     // transposed factor: [column][16 - out_off + row]
                l_w[j][0] = 1.0;
            }
            l_soff0 = (cur.row0 + 16) * 8;
        } else if (LEAF) {
            const int32_t* cnt = o->counts + (int64_t)o->taxon * o->counts_ld + a.f0 + cur.col0 + wave * 32 + l15;
            rsL = __builtin_amdgcn_make_buffer_rsrc((void*)(a.lpool.base + (int64_t)o->leaf_slot[cur.cat] * a.lpool.stride), 0,
                                                    (int)(a.lpool.stride * 8), 0x00020000);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int x = cnt[16 * j];
#pragma unroll
                for (int i = 0; i < NT; ++i) {
                    const int cc = LEAF == 3 ? x - 1 + i : x;
                    const bool ok = LEAF != 3 || (cc >= 0 && cc <= a.max_family_size);
                    l_voff[j][i] = (unsigned)((l4 * ldl + (ok ? cc : x)) * 8);
                    l_w[j][i] = LEAF == 3 ? (ok ? a.err[(int64_t)x * 3 + i] : 0.0) : 1.0;
                }
            }
            l_soff0 = (cur.row0 + 1) * ldl * 8;             // parent size row0 + 1
        }
        const int rows_here = o->rows - cur.row0;           // valid rows of this tile (>= BM for interior tiles)
        struct Pre { double f[2][4]; };
        // FULL: the tile has all BM rows (always true when 16*MI divides the row count): no per-row masks
        auto prefetch = [&](int i, Pre& p, auto full) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int step = i * 16 + r * 4;            // uniform
                const bool ok = decltype(full)::value || step + l4 < rows_here;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    double f = 1.0;
                    if (ok) {
                        if (LEAF == 1) f = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rsL, l_voff[j][0], l_soff0 + step * ldl * 8, 0));
                        if (GATH) f = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rsL, l_voff[j][0], l_soff0 + step * 8, 0));
                        if (LEAF == 3) {                // sum_i err[x][i] * P_leaf[s][x - 1 + i], taps in order (leaf_reduce.hip)
                            f = 0.0;
#pragma unroll
                            for (int i = 0; i < 3; ++i)
                                f += __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rsL, l_voff[j][i], l_soff0 + step * ldl * 8, 0)) * l_w[j][i];
                        }
                        if (MUL) f *= __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(cur.rsC, c_voff, c_soff0 + (step * ldb + j * 16) * 8, CAFE_K2_MUL_LOAD_AUX));
                    }
                    p.f[j][r] = f;
                }
            }
        };
        typedef int int2_t __attribute__((ext_vector_type(2)));
        auto flush = [&](int i, const Pre& p, auto full) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int step = i * 16 + r * 4;
                if (TRANS) {                                // C^T: register r = columns 4r + (lane >> 4), lanes = 16 rows
                    if (decltype(full)::value || i * 16 + l15 < rows_here) {
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const double v = acc[i][j][r];      // (a bit_cast straight from the vector element stores element 0)
                            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(int2_t, v), cur.rsC, c_voff,
                                                                  c_soff0 + ((j * 16 + r * 4) * ldt + i * 16) * 8, CAFE_K2_C_STORE_AUX);
                        }
                    }
                } else if (decltype(full)::value || step + l4 < rows_here) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        double v = acc[i][j][r];
                        if (MUL || LEAF) v *= p.f[j][r];
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(int2_t, v), cur.rsC, c_voff, c_soff0 + (step * ldb + j * 16) * 8, CAFE_K2_C_STORE_AUX);
                    }
                }
            }
        };
        auto epilogue = [&](auto full) {
            // (a second set of prefetched operands only where the register budget has room: not with four waves per SIMD)
            constexpr int NB = (LEAF == 3 || prune_gemm_wg_per_cu(MI, KB) >= 4) ? 1 : 2;
            Pre pre[NB];
            if (MUL || LEAF) prefetch(0, pre[0], full);
            // full unroll is mandatory: a runtime i would index the accumulator array dynamically and demote
            // it to scratch memory
#pragma clang loop unroll(full)
            for (int i = 0; i < MI; ++i) {
                if (NB == 2) {
                    if ((MUL || LEAF) && i + 1 < MI) prefetch(i + 1, pre[(i + 1) & 1], full);
                    flush(i, pre[i & (NB - 1)], full);
                } else {
                    flush(i, pre[0], full);
                    if (i + 1 < MI) prefetch(i + 1, pre[0], full);
                }
            }
        };
        unsigned long long e0 = 0;
        if (a.stamps) { e0 = __builtin_amdgcn_s_memrealtime(); loop_ticks += e0 - t_tile; kt_done += (unsigned long long)cur.nkt; }
        if (rows_here >= BM) epilogue(std::true_type{});
        else epilogue(std::false_type{});
        if (a.stamps) { ep_ticks += __builtin_amdgcn_s_memrealtime() - e0; n_done += 1; }
        // parent size 0 only reaches child size 0 (P[0][c] = delta(c,0)): that panel row is the child's row 0,
        // times the leaf sibling's P_leaf[0][x] = delta(x,0) (with an error model: the weight of the tap at size 0)
        if (out_off == 1 && cur.row_tile == 0 && tid < kBN / 2) {
            const int c2 = tid * 2;
            const double* Bp = o->src + (int64_t)cur.cat * o->src_kstride + cur.col0 + c2;
            double2 v = *reinterpret_cast<const double2*>(Bp);
            if (GATH) {                                     // the sibling's factor at parent size 0 (transposed: row index 15)
                const double* G = o->gath_src + (int64_t)cur.cat * o->gath_kstride + 15;
                v.x *= G[(int64_t)o->gath_map[cur.col0 + c2] * ldl];
                v.y *= G[(int64_t)o->gath_map[cur.col0 + c2 + 1] * ldl];
            } else if (LEAF) {
                const int32_t* cnt = o->counts + (int64_t)o->taxon * o->counts_ld + a.f0 + cur.col0 + c2;
                if (LEAF == 1) {
                    v.x = cnt[0] == 0 ? v.x : 0.0;
                    v.y = cnt[1] == 0 ? v.y : 0.0;
                } else {                                    // the tap that lands on size 0: err[0][1] for x = 0, err[1][0] for x = 1
                    v.x *= cnt[0] == 0 ? a.err[1] : (cnt[0] == 1 ? a.err[3] : 0.0);
                    v.y *= cnt[1] == 0 ? a.err[1] : (cnt[1] == 1 ? a.err[3] : 0.0);
                }
            }
            if (TRANS) {
                double* dt = o->dst + (int64_t)cur.cat * o->dst_kstride + (int64_t)(cur.col0 + c2) * ldt + 15;
                dt[0] = v.x;
                dt[ldt] = v.y;
            } else {
                double2* dst = reinterpret_cast<double2*>(o->dst + (int64_t)cur.cat * o->dst_kstride + cur.col0 + c2);
                if (MUL) {
                    const double2 old = *dst;
                    v.x *= old.x;
                    v.y *= old.y;
                }
                *dst = v;
            }
        }
        // the stores need no wait here (they drain during the next main loop)
        if (!has_next) break;
        e_cur = e_nxt;
    }
    if (a.stamps && tid == 0) {             // diagnostic build only: per-block placement + lifetime (100 MHz ticks)
        unsigned long long* o = a.stamps + 6 * (size_t)blockIdx.x;
        o[0] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));      // HW_REG_HW_ID
        o[1] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));     // HW_REG_XCC_ID
        o[2] = st0; o[3] = (ep_ticks & 0xFFFFFFFFull) | (loop_ticks << 32); o[4] = (n_done & 0xFFFFFull) | (kt_done << 20); o[5] = __builtin_amdgcn_s_memrealtime();
    }
#else
    (void)a;
#endif
}

// Row-tile height (in 16-row blocks) of a launch without extent information: the persistent grid has `slots` workgroups, a
// launch takes ceil(tiles / slots) rounds of one tile each, and a tile costs about MI (its MFMA count) -- so small launches
// are better off with lower tiles that fill their last round, while large ones want the tallest tile without row padding.
// Lower tiles are a little less efficient per flop (80-row tiles: 68.2 against 69.9 TFLOP/s on full-width launches): 0.6 %
// per step of MI.  tiles_by_mi[mi]: tiles of the whole group at that height.  Heights 2 and 3 (32 / 48 rows) exist for the
// real-data regime (mammals: M = 140, 9 K tiles, a launch is ONE round of tiles and lasts as long as one tile: 18 us at 64
// rows, of which 8 are the wave's 9 x 32 MFMAs).
int prune_gemm_pick_mi(int64_t tiles_by_mi[10], int n_cu, int kb) {
    int best = 9;
    double best_cost = 1e300;
    for (int mi = 9; mi >= 2; --mi) {
        const int slots = prune_gemm_wg_per_cu(mi, kb) * n_cu / 8 * 8;
        const int64_t rounds = (tiles_by_mi[mi] + slots - 1) / slots;
        // (32- and 48-row tiles stage a B tile for few rows: only worth it when the launch is one round anyway -- the
        // small-matrix regime, where a tile's latency is the launch's)
        const double cost = (double)rounds * mi * (1.0 + 0.006 * (9 - mi) + (mi < 4 ? 0.2 * (4 - mi) : 0.0));
        if (cost < best_cost * (1.0 - 1e-9)) { best_cost = cost; best = mi; }
    }
    return best;
}

// Persistent grid: as many workgroups per CU as are resident (the register/LDS budget admits three at up to 80-row tiles, two
// above), a multiple of 8 so that every XCD gets the same number.  XCD x (blocks x, x+8, ...) owns the (category, column
// tile) pairs x, x+8, ... of every op of the launch and its blocks share those pairs' row tiles: a small launch gets as
// many blocks per XCD as the busiest XCD (XCD 0) has tiles.
int prune_gemm_blocks(int64_t tiles_xcd0, int n_cu, int mi, int kb) {
    int blocks = std::min(prune_gemm_wg_per_cu(mi, kb) * n_cu / 8 * 8, 8 * kPlanLanes);     // (the planner deals with one lane per workgroup of an XCD)
    if (tiles_xcd0 * 8 < blocks) blocks = (int)(tiles_xcd0 * 8);
    return blocks < 8 ? 8 : blocks;
}

// ev0 / ev1 (both or neither): start / stop events attached to the dispatch itself (hipExtLaunchKernelGGL) -- the per-launch
// timing of bench.py without two extra event-record packets around every launch
#define CAFE_LAUNCH_GEMM(...)                                                                          \
    do {                                                                                               \
        if (ev0) hipExtLaunchKernelGGL((__VA_ARGS__), grid, block, 0, stream, ev0, ev1, 0, a);              \
        else hipLaunchKernelGGL((__VA_ARGS__), grid, block, 0, stream, a);                               \
    } while (0)
template <int KB, int MI>
static void launch_mi(const GemmArgs& a, GemmVariant v, dim3 grid, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1) {
    const dim3 block(256);
    const int leaf = v.leaf == 1 ? (a.err ? 3 : 1) : v.leaf;
    if (leaf == 2) {                                       // gathered sibling factor: always the launch that creates the panel
        CAFE_LAUNCH_GEMM(prune_gemm_kernel<KB, MI, false, 2>);
        return;
    }
    if (v.trans) {                                         // factor GEMM: transposed plain store
        CAFE_LAUNCH_GEMM(prune_gemm_kernel<KB, MI, false, 0, true>);
        return;
    }
    if (v.mode) {
        if (leaf == 3) CAFE_LAUNCH_GEMM(prune_gemm_kernel<KB, MI, true, 3>);
        else if (leaf == 1) CAFE_LAUNCH_GEMM(prune_gemm_kernel<KB, MI, true, 1>);
        else CAFE_LAUNCH_GEMM(prune_gemm_kernel<KB, MI, true, 0>);
    } else {
        if (leaf == 3) CAFE_LAUNCH_GEMM(prune_gemm_kernel<KB, MI, false, 3>);
        else if (leaf == 1) CAFE_LAUNCH_GEMM(prune_gemm_kernel<KB, MI, false, 1>);
        else CAFE_LAUNCH_GEMM(prune_gemm_kernel<KB, MI, false, 0>);
    }
}

hipError_t launch_prune_gemm(const GemmArgs& a, GemmVariant v, int blocks, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1) {
    if (!a.ops || a.n_ops < 1 || a.n_ops > kMaxGroupOps || !a.plan || a.plan_rounds < 1 || blocks < 8 || (blocks & 7)) return hipErrorInvalidValue;
    if (v.leaf == 2 && (v.mode || v.trans)) return hipErrorInvalidValue;       // the schedule never asks for these
    if (v.trans && (v.leaf || v.mode)) return hipErrorInvalidValue;
    dim3 grid(blocks, 1, 1);
    (void)hipGetLastError();
    if (a.kb != 8 && a.kb != 16) return hipErrorInvalidValue;
#define CAFE_MI_CASE(M) case M: if (a.kb == 8) launch_mi<8, M>(a, v, grid, stream, ev0, ev1); else launch_mi<16, M>(a, v, grid, stream, ev0, ev1); break;
    switch (a.mi) {
        CAFE_MI_CASE(2) CAFE_MI_CASE(3) CAFE_MI_CASE(4) CAFE_MI_CASE(5) CAFE_MI_CASE(6) CAFE_MI_CASE(7) CAFE_MI_CASE(8) CAFE_MI_CASE(9)
        default: return hipErrorInvalidValue;
    }
#undef CAFE_MI_CASE
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
