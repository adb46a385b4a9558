// Device-side interface of the MI355X likelihood path: argument blocks and launchers of
// the four kernels:
//   K1 bd_matrix_build  -- replaces matrix_cache::precalculate_matrices (matrix_cache.cpp:121)
//   K2 prune_gemm       -- replaces matrix::multiply for interior children (matrix_cache.cpp:28)
//   K3 leaf_gather      -- replaces the leaf branch of compute_node_probability (probability.cpp:179)
//   K4 root_reduce      -- replaces the per-family loops of base_model.cpp:89 / gamma_core.cpp:144
// gfx950 only; no other target is supported or tested.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cafe {

constexpr int kMaxCategories = 32;   // CAFE_MAX_CATEGORIES
constexpr int kMaxLeafPerOp = 4;     // leaf children folded by one gather launch / one GEMM epilogue

// GEMM tiling (fp64 MFMA 16x16x4): block tile (16*MI) x 128, K step 16, 4 waves side by side in N
constexpr int kBN = 128;
constexpr int kBK = 16;          // largest depth of a K tile (panel / matrix row counts are rounded to it); a context runs K2 with 16-deep
                                 // K tiles (small matrices: a launch is one round of tiles, fewer DMA round trips per tile) or 8-deep ones
                                 // (half the LDS per stage: four workgroups per CU) -- GemmArgs::kb
constexpr int kMaxBM = 144;          // largest row tile (MI = 9)

// Per transition matrix: a = lambda_q t_q / (1 + lambda_q t_q) of the de-quantized key
// (matrix_cache.cpp:148-149), oma2 = (1-a)^2; zero = saturated (matrix_cache.cpp:153) or
// !(coeff > 0 && coeff != 1) (probability.cpp:154): rows s >= 1 stay 0.
struct SlotParam {
    double alpha;
    double oma2;
    int32_t zero;
    int32_t pad;
};

// The matrices of a call live in two pools, slot i at base + i*stride (bd_matrix.hip):
//   row-major (kmajor = 0): P[s][c], n rows, leading dimension ld (multiple of 16), columns >= n zero;
//   k-major   (kmajor = 1): Pt[c][j] = P[j+1][c], `rows` rows (round_up(M+1, kBK); rows >= k_valid = M+1
//                           are zero), leading dimension ld >= n-1 + one row tile, columns >= n-1 zero.
struct MatrixPool {
    double* base;
    int64_t stride;
    int32_t ld;
    int32_t n;        // matrix order N = max(M,R)+1
    int32_t rows;     // k-major: rows written
    int32_t k_valid;  // k-major: M+1
    int32_t kmajor;
    // k-major only: per matrix and per block of 16 consecutive parent sizes (columns of Pt), the first and last
    // contraction index c whose entries are not all exactly 0 -- ext[(slot * ext_blocks + block) * 2 + {0, 1}], written by
    // K1 (first > last: the block is all zero).  Far from the diagonal the entries of a short branch underflow to exact
    // zeros; K2 skips the K tiles that lie outside a row tile's extent (adding exact zeros changes no bit).
    int32_t* ext;
    int32_t ext_blocks;
};

// Likelihood panels: [category][row][family], family fastest, so that a node's panel is the
// GEMM's B operand with the family axis as its N dimension.
//
// One K2 LAUNCH carries the GEMMs of a GROUP of mutually independent branches (the ready nodes of a tree level that share
// an epilogue variant): per branch one GemmOp in a device array, per launch one GemmArgs by value.  A workgroup's tile list
// (tile_plan_kernel) names, per tile, the op it belongs to; the kernel fetches that op's descriptor with scalar loads while
// the previous tile computes.  Small nodes fill the persistent grid together and a launch's tail overlaps the next node's
// head ("one launch (or one persistent-kernel phase) per height"; reference loop src/core.cpp:133-144).
struct GemmOp {
    int32_t slot[kMaxCategories];   // matrix of the child's branch per category (k-major pool)
    const double* src;              // child panel, rows = child sizes 0..M
    double* dst;                    // parent panel, or (dst_ldt > 0) the child's transposed factor panel
    int64_t src_kstride;            // doubles between the categories of src / dst / gath_src
    int64_t dst_kstride;
    int64_t gath_kstride;
    int32_t ld;                     // leading dimension of src and of a row-major dst = GEMM columns (multiple of kBN)
    int32_t rows;                   // GEMM rows = parent sizes 1..rows: M, or R when the parent is the root
    int32_t out_off;                // panel row of parent size 1: 1 (interior parent; row 0 is copied), 0 (root)
    int32_t n_row_tiles;            // ceil(rows / (16 * GemmArgs::mi)): filled per call
    int32_t n_col_tiles;
    // > 0: a factor GEMM, stored transposed: dst[column][16 - out_off + panel row], dst_ldt rows per column
    int32_t dst_ldt;
    // one leaf sibling folded into the epilogue (K3's work for a parent with one leaf and interior children)
    int32_t n_leaf;
    int32_t taxon;                  // row of `counts`
    int32_t leaf_slot[kMaxCategories];
    const int32_t* counts;          // [row][column] observed counts
    int64_t counts_ld;
    // factor panel of an interior sibling with fewer distinct columns than the parent, folded into the epilogue:
    // column f of the parent takes column gath_map[f] of the TRANSPOSED factor gath_src (gath_ld rows per column)
    const double* gath_src;
    int64_t gath_ld;
    const int32_t* gath_map;
    // per (category, column tile) of the CHILD panel: rows outside [bext[..][0], bext[..][1]] are exactly zero (extents.hip);
    // nullptr: unknown
    const int32_t* bext;
};

struct GemmArgs {
    MatrixPool pool;                // k-major pool
    MatrixPool lpool;               // row-major pool (fused leaf siblings)
    const GemmOp* ops;              // the group's descriptors (device memory)
    int32_t n_ops;
    int32_t k_valid;                // contraction extent M+1
    int32_t kb;                     // depth of a K tile: 8 or 16
    int32_t mi;                     // row tile = 16*mi, the same for every op of the launch
    int32_t n_categories;
    int32_t uniform_ld;             // > 0 (several column chunks, one column per family): columns of every panel in this chunk;
                                    // overrides GemmOp::ld / n_col_tiles
    int64_t f0;                     // first family of the chunk (offset into `counts`)
    const double* err;              // [(M+1)][3] error model of the fused leaf sibling, or nullptr
    int32_t max_family_size;
    unsigned long long* stamps;     // diagnostic: 6 words per workgroup (placement, epilogue ticks, tiles, lifetime), nullptr in production
    // Tile lists laid out by tile_plan_kernel (extents.hip): workgroup `local` of XCD `xcd` runs the tiles
    // plan[(xcd * blocks_per_xcd + local) * plan_rounds + i], i = 0, 1, ... until an entry with y == 0.
    // x: op << 24 | index of the tile in the op's list for that XCD (pair-major, row tile fastest),
    // y: first K tile << 16 | number of K tiles.
    const int2* plan;
    int32_t plan_rounds;
};
constexpr int kMaxGroupOps = 128;   // ops per K2 launch (7 bits of a plan entry)

// One K2 launch as the tile planner sees it.  K loops of unequal length (zero extents) make a fixed deal of tiles uneven:
// the planner deals each round of tiles (one per workgroup of the XCD, in the order the fixed deal would run them, so the
// row tiles of a column tile still run together) longest tile to least-loaded workgroup.  The XCD's list is the
// concatenation of the lists of the launch's ops.
struct PlanLaunch {
    const int32_t* aext;            // k-major pool extents (MatrixPool::ext), or nullptr: every K tile
    int32_t ext_blocks;
    const GemmOp* ops;              // slot[], bext, n_row_tiles, n_col_tiles of every op
    int32_t n_ops;
    int32_t uniform_ld;             // as GemmArgs
    int32_t mi, n_categories, k_valid;
    int32_t kb;                     // depth of a K tile (GemmArgs::kb)
    int32_t blocks_per_xcd;         // workgroups of the launch / 8
    int32_t rounds;                 // list length per workgroup
    int32_t fixed;                  // what an output tile costs beyond its K loop, in K tiles
    int32_t bias;                   // percent: the first-dispatched workgroup of a CU (local index < 32 of 64) runs that much faster
    int32_t bias3[4];               // three or four workgroups per CU (96 / 128 per XCD; local indices j, j + 32, ... share a CU): what a K
                                    // tile costs the first / second / ... dispatched one, in percent of the mean
    int2* plan;                     // [8][blocks_per_xcd][rounds]
};
constexpr int kPlanSlack = 2;       // spare list entries per workgroup (the planner's last rounds are dealt as one batch)
// max_rounds: the longest PlanLaunch::rounds of the launches (the planner's grid depth)
hipError_t launch_tile_plan(const PlanLaunch* d_launches, int n_launches, int max_rounds, hipStream_t stream);
// Workgroups of K2 resident on a CU.  Row tiles of up to 80 rows: four with 8-deep K tiles (26 KB of LDS each, 128 vector
// registers), three with 16-deep ones (52 KB with an unpadded B tile, <= 168 registers); the taller ones: two (<= 250
// registers).
constexpr int prune_gemm_wg_per_cu(int mi, int kb) { return mi <= 5 ? (kb <= 8 ? 4 : 3) : 2; }
constexpr int kPlanLanes = 128;     // workgroups of an XCD the tile planner can deal to (32 CUs x 3 = 96 on MI355X)
// workgroups of a K2 launch (a multiple of 8) whose busiest XCD owns `tiles_xcd0` tiles: as many as are resident, fewer when there are fewer tiles
int prune_gemm_blocks(int64_t tiles_xcd0, int n_cu, int mi, int kb);
// tiles XCD 0 owns of an op with n_pairs = categories * column tiles
inline int64_t prune_gemm_tiles_xcd0(int n_categories, int n_col_tiles, int n_row_tiles) {
    return (((int64_t)n_categories * n_col_tiles + 7) / 8) * n_row_tiles;
}

struct GatherArgs {
    int32_t n_leaf;
    int32_t taxon[kMaxLeafPerOp];
    int32_t slot[kMaxLeafPerOp][kMaxCategories];
    const int32_t* counts;          // [taxon][family] for the whole shard
    int64_t counts_ld;
    double* dst;
    int64_t panel_kstride;          // doubles between the categories of dst
    int32_t ld;
    int32_t row_off;
    int32_t rows;
    int32_t rows_store;
    int32_t mode;
    // factor panels of interior children that have fewer distinct columns than this parent (subtree-level
    // de-duplication): column f of the parent takes column map[j][f] of factor j.  Factors are stored TRANSPOSED by
    // their GEMM (prune_gemm.hip, TRANS): src[j][category][column][15 + row_off + panel row], ld_src rows per column.
    int32_t n_src;
    const double* src[2];
    int64_t ld_src[2];
    int64_t kstride_src[2];         // doubles between the categories of src[j]
    const int32_t* map[2];
    // [category][ld / 128][2] zero extent of each 128-column tile of dst (extents.hip), or nullptr: rows outside it (rounded
    // out to K2's 16-row K tiles) are neither read nor written by the assemble pass
    const int32_t* tileext;
    // TRANSPOSED copies of the leaf children's row-major matrices (leaf_transpose_kernel), laid out like a factor:
    // lt[l][category][count x][15 + row s] = P_leaf[s][x], ld_src[0] doubles per count -- so that the assemble pass reads a
    // leaf's factor as whole lines, like the interior children's, instead of one 8-byte element per matrix row.  nullptr: none
    // (the pass then gathers from the row-major matrix on its way out).
    const double* lt[kMaxLeafPerOp];
    int64_t lt_kstride;             // doubles between the categories of lt[l]
};
// One transposed copy per (leaf branch that takes part in an assemble pass, category): src slot of the row-major pool -> dst.
struct LeafTArgs {
    MatrixPool pool;                // row-major pool
    const int32_t* pairs;           // [n_pairs] branch (pair index of the row-major pool); slot = category * pool_pairs + pair
    int32_t pool_pairs;
    int32_t n_list;                 // entries of `pairs`
    int32_t n_x;                    // counts 0..n_x-1 (columns of P read)
    int32_t ld_t;                   // doubles per count in dst
    double* dst;                    // [pair][category][n_x][ld_t]
    int64_t kstride;                // doubles between categories
    int64_t pair_stride;            // doubles between pairs
};
hipError_t launch_leaf_transpose(const LeafTArgs& a, int n_pairs, int n_categories, hipStream_t stream);

// Zero extents of the likelihood panels (extents.hip).  One ExtNode per interior non-root node, static per context.
constexpr int kMaxExtChildren = 4;
struct ExtNode {
    int32_t cols;                            // padded columns of the node's panel
    int32_t n_leaf, n_inner;
    int32_t leaf_pair[kMaxExtChildren];      // the leaf child's branch: pair index in the row-major pool
    int32_t leaf_row[kMaxExtChildren];       // the leaf child's row in `cnt`
    const int32_t* cnt;                      // [leaf children][columns] observed counts
    int64_t cnt_ld;
    int32_t inner_pair[kMaxExtChildren];     // the interior child's branch: pair index in the k-major pool
    int32_t inner_cols[kMaxExtChildren];
    const int32_t* inner_map[kMaxExtChildren];      // this node's column -> the child's column (nullptr: the same)
    const int32_t* inner_colext[kMaxExtChildren];   // the child's per-column extents [category][its columns][2]
    int32_t* colext;                         // [category][cols][2]: rows outside [lo, hi] of a column are exactly zero
    int32_t* tileext;                        // [category][cols / 128][2]: hull over a 128-column tile (K2's B operand)
};
struct ExtArgs {
    const ExtNode* nodes;
    int32_t first, count;                    // the nodes of one level (independent of each other)
    const int32_t* leaf_ext;                 // row-major pool: per slot and column x, first / last row s with P[s][x] != 0
    int32_t leaf_ext_blocks, n_pairs_leaf;
    const int32_t* kext;                     // k-major pool: per slot and block of 16 parent sizes, first / last child size
    int32_t kext_blocks, n_pairs_inner;
    int32_t M;
    const double* err;                       // error model or nullptr
    int32_t n_dev;
};
hipError_t launch_node_extents(const ExtArgs& a, int max_col_tiles, int n_categories, hipStream_t stream);

struct ReduceArgs {
    const double* root;             // root panel [category][i][family]
    int64_t panel_kstride;
    int32_t ld;
    int32_t R;
    int32_t K;
    int32_t model;                  // 0 base, 1 gamma, 2 root maximum (p-value path)
    const double* prior;            // [R]  (double)float prior
    const double* log_prior;        // [R]  log((double)float prior), host libm
    const double* cat_probs;        // [K]
    int64_t f0;                     // first family of the chunk
    int64_t nf;                     // families of the chunk that are real (not padding)
    double* fam_out;                // [F] lnL_f (base) / log(lik_f) (gamma)
    double* fam_lik;                // [F] lik_f (gamma)
    double* cat_out;                // [F][K] (gamma)
    int32_t* failed;                // [F]
};

hipError_t launch_bd_matrix_build(const MatrixPool& pool, const SlotParam* d_slots, int n_slots, hipStream_t stream);
hipError_t launch_bd_matrix_build_both(const MatrixPool& pool, const MatrixPool& kpool, const SlotParam* d_slots, const SlotParam* d_kslots,
                                       int n_slots, int n_kslots, hipStream_t stream);
// One launch for a group of ops.  variant: what the epilogue of EVERY op of the group does (the host groups by it):
//   mode (0 store, 1 multiply), leaf (0 none, 1 one leaf sibling, 2 gathered sibling factor), trans (factor GEMM, transposed store).
// blocks: prune_gemm_blocks of the group; events: attached to the dispatch.
struct GemmVariant { int mode, leaf, trans; };
hipError_t launch_prune_gemm(const GemmArgs& a, GemmVariant v, int blocks, hipStream_t stream, hipEvent_t ev_start = nullptr,
                             hipEvent_t ev_stop = nullptr);
int prune_gemm_pick_mi(int64_t row_tile_pairs_by_mi[10], int n_cu, int kb);     // row-tile height (in 16-row blocks) from the group's tile counts per height
// what a K3 launch of a group of ops shares
struct GatherGroup {
    MatrixPool pool;                // row-major pool
    const GatherArgs* ops;          // device array
    int32_t n_ops;
    int32_t n_categories;
    int32_t max_family_size;        // M
    const double* err;              // [(M+1)][n_dev] error model of the call, or nullptr
    int32_t n_dev;
    int32_t uniform_ld;             // > 0 (several column chunks): columns of every panel in this chunk, overrides GatherArgs::ld
    int64_t f0;                     // first family of the chunk (offset into `counts`)
    int32_t leaf_t;                 // this call filled the transposed leaf matrices (GatherArgs::lt): no error model, copies made
};
// One launch for a group of K3 ops of one variant (same n_leaf, n_src, mode; the error model is the call's): d_ops device
// array, h_ops the same on the host (grid extents)
hipError_t launch_leaf_gather_group(const GatherGroup& g, const GatherArgs* h_ops, hipStream_t stream);
hipError_t launch_root_reduce(const ReduceArgs& a, hipStream_t stream);
// sum_f w_f * fam_out[f] and the number of failed families -> out[0], out[1] (and out_host[0..1] when not null: pinned host memory)
hipError_t launch_final_sum(const double* fam_out, const double* weights, const int32_t* failed, int64_t n,
                            double* scratch, int n_scratch, double* out, double* out_host, hipStream_t stream);
hipError_t launch_mfma_probe(double* d_out, int iters, int blocks, hipStream_t stream);
int bd_matrix_max_order();

}  // namespace cafe
