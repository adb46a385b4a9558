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

// GEMM tiling (fp64 MFMA 16x16x4, 4 waves, each a 64x64 sub-tile)
constexpr int kBM = 128;
constexpr int kBN = 128;
constexpr int kBK = 16;

// Per transition matrix: a = lambda_q t_q / (1 + lambda_q t_q) of the de-quantized key
// (matrix_cache.cpp:148-149), oma2 = (1-a)^2; zero = saturated (matrix_cache.cpp:153) or
// !(coeff > 0 && coeff != 1) (probability.cpp:154): rows s >= 1 stay 0.
struct SlotParam {
    double alpha;
    double oma2;
    int32_t zero;
    int32_t pad;
};

// All matrices of a call live in one pool: slot i at base + i*stride, row-major, leading
// dimension ld (multiple of 16, columns >= n are zero), n rows.
struct MatrixPool {
    double* base;
    int64_t stride;
    int32_t ld;
    int32_t n;
};

// Likelihood panels: [category][row][family], family fastest, so that a node's panel is the
// GEMM's B operand with the family axis as its N dimension.
struct GemmArgs {
    MatrixPool pool;
    int32_t slot[kMaxCategories];   // matrix of the child's branch per category
    const double* src;              // child panel  (chunk-relative base)
    double* dst;                    // parent panel (chunk-relative base)
    int64_t panel_kstride;          // doubles between categories of a panel
    int32_t ld;                     // panel leading dimension = chunk columns (multiple of kBN)
    int32_t kc;                     // contraction extent, round_up(M+1, kBK); src rows >= M+1 are zero
    int32_t row_off;                // 0, or 1 when the parent is the root (root index i <-> size i+1)
    int32_t rows;                   // valid output rows: M+1, or R at the root
    int32_t rows_store;             // rows written; rows..rows_store-1 are written as zero
    int32_t mode;                   // 0: dst = v, 1: dst *= v
    int32_t n_row_tiles;
    int32_t n_col_tiles;
};

struct GatherArgs {
    MatrixPool pool;
    int32_t n_leaf;
    int32_t taxon[kMaxLeafPerOp];
    int32_t slot[kMaxLeafPerOp][kMaxCategories];
    const int32_t* counts;          // [taxon][family] for the whole shard
    int64_t counts_ld;
    int64_t f0;                     // first family of the chunk
    double* dst;
    int64_t panel_kstride;
    int32_t ld;
    int32_t row_off;
    int32_t rows;
    int32_t rows_store;
    int32_t mode;
    const double* err;              // [(M+1)][n_dev] or nullptr
    int32_t n_dev;
    int32_t max_family_size;        // M
};

struct ReduceArgs {
    const double* root;             // root panel [category][i][family]
    int64_t panel_kstride;
    int32_t ld;
    int32_t R;
    int32_t K;
    int32_t model;                  // 0 base, 1 gamma
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
hipError_t launch_prune_gemm(const GemmArgs& a, int n_categories, hipStream_t stream);
hipError_t launch_leaf_gather(const GatherArgs& a, int n_categories, hipStream_t stream);
hipError_t launch_root_reduce(const ReduceArgs& a, hipStream_t stream);
// sum_f w_f * fam_out[f] and the number of failed families -> out[0], out[1]
hipError_t launch_final_sum(const double* fam_out, const double* weights, const int32_t* failed, int64_t n,
                            double* scratch, int n_scratch, double* out, hipStream_t stream);
hipError_t launch_mfma_probe(double* d_out, int iters, int blocks, hipStream_t stream);
int bd_matrix_max_order();

}  // namespace cafe
