/* cafe_mi355x.h -- C ABI of the MI355X-native birth-death likelihood path.
 *
 * Drop-in boundary for ONE path of CAFE5 (Han9527/CAFExp): what a scorer call evaluates,
 *   optimizer_scorer::calculate_score            src/optimizer_scorer.cpp:19
 *     -> model::infer_family_likelihoods         src/core.h:171
 *          base_model::infer_family_likelihoods  src/base_model.cpp:53
 *          gamma_model::infer_family_likelihoods src/gamma_core.cpp:169
 * i.e. matrix_cache::precalculate_matrices (src/matrix_cache.cpp:121), inference_prune
 * (src/core.cpp:133) for every family (and gamma category) and the per-family root reduction.
 * and, for what the reference runs once after the search (estimator::execute, src/execute.cpp:147-180):
 * the prunes of compute_pvalues (cafe_root_max), Pupko's reconstruction (cafe_reconstruct) and the Viterbi branch
 * probabilities (cafe_branch_probabilities).
 * Everything behind this header is hand-written HIP for gfx950; there is no CPU fallback:
 * every entry point fails (non-zero code / NULL + message) when no HIP device is usable.
 *
 * Conventions
 *   - plain pointers and sizes only; the caller owns every pointer it passes and nothing is
 *     retained after a call returns except what cafe_create copies;
 *   - numeric rejection is a VALUE, not an error: cafe_score returns 0 and writes +inf exactly
 *     where the reference returns -log(0) (invalid lambda base_model.cpp:56-60; !can_infer
 *     gamma_core.cpp:175-179; a zero-likelihood category gamma_core.cpp:227-236).  NaN is passed
 *     through; the scorer maps it to +inf (optimizer_scorer.cpp:30);
 *   - structural errors (bad arguments, HIP failures) return a non-zero code; the text is
 *     available from cafe_last_error.  Nothing throws across this boundary;
 *   - one call in flight per context (the reference's scorer calls are sequential).
 *
 * The reference-side binding a maintainer would add is shown in INTEGRATION.md; the C++ classes
 * that wrap this ABI behind the reference's model / optimizer_scorer interfaces live in
 * cafexp_amd/host/.
 */
#ifndef CAFE_MI355X_H
#define CAFE_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CAFE_ABI_VERSION 3
#define CAFE_MAX_CATEGORIES 32

typedef struct cafe_ctx cafe_ctx;

enum {
    CAFE_OK = 0,
    CAFE_ERR_ARGUMENT = 1,   /* malformed problem / params */
    CAFE_ERR_DEVICE = 2,     /* HIP runtime error or no usable device */
    CAFE_ERR_MEMORY = 3,     /* problem does not fit the device */
    CAFE_ERR_STATE = 4       /* call not valid in the context's state */
};

/* Static part of a scorer's state: replaces what model's constructor captures
 * (src/core.cpp:58-70: tree, families, max sizes, error-model shape) plus the clade
 * traversals the reference redoes per family (clade.cpp:255, gene_family.cpp:36). */
typedef struct cafe_problem {
    int32_t n_nodes;                 /* tree nodes; any order with children before parents; exactly one root */
    const int32_t* parent;           /* [n_nodes] parent index, -1 for the root */
    const double*  branch_length;    /* [n_nodes] clade::get_branch_length(); the root's entry is ignored */
    const int32_t* lambda_index;     /* [n_nodes] 0-based lambda of the branch above the node
                                        (clade::get_lambda_index_map, clade.cpp:154); NULL = all 0 */
    const int32_t* leaf_taxon;       /* [n_nodes] column of `counts` for a leaf, -1 for interior nodes */
    int32_t  n_taxa;
    int64_t  n_families;
    const int32_t* counts;           /* [n_families][n_taxa] gene_family::get_species_size */
    int32_t  max_family_size;        /* M, user_data.cpp:46 */
    int32_t  max_root_family_size;   /* R, user_data.cpp:45 */
    int32_t  n_lambdas;              /* lambda::count() */
    int32_t  single_lambda;          /* 1: single_lambda::is_valid (lambda > 0, lambda.h:58);
                                        0: multiple_lambda::is_valid (none < 0, lambda.cpp:59) */
    int32_t  max_categories;         /* largest K any call will use (1 for the base model) */
    int32_t  n_deviations;           /* error_model::n_deviations(), 0 when no error model */
    int32_t  device;                 /* HIP device ordinal */
    int32_t  flags;                  /* CAFE_FLAG_* */
    size_t   workspace_limit;        /* bytes of HBM the likelihood panels may take; 0 = automatic */
} cafe_problem;

#define CAFE_FLAG_NO_DEDUP 1         /* keep identical families separate (build_reference_list, base_model.cpp:27,
                                        collapses them; values are identical either way) */
#define CAFE_FLAG_NO_SUBTREE_DEDUP 2 /* one panel column per family at every node, instead of one per distinct pattern of
                                        leaf counts under the node (the same sharing as build_reference_list, applied per
                                        subtree; values are identical either way) */

enum { CAFE_MODEL_BASE = 0, CAFE_MODEL_GAMMA = 1 };

/* Per scorer call: what prepare_calculation (optimizer_scorer.cpp:54,80,123,161) mutates. */
typedef struct cafe_params {
    int32_t model;                   /* CAFE_MODEL_BASE | CAFE_MODEL_GAMMA */
    const double* lambdas;           /* [n_lambdas] */
    int32_t n_categories;            /* K (gamma); ignored for the base model */
    const double* multipliers;       /* [K] gamma_model::_lambda_multipliers */
    const double* cat_probs;         /* [K] gamma_model::_gamma_cat_probs */
    double  alpha;                   /* gamma shape, only for can_infer's alpha >= 0 test (gamma_core.cpp:128) */
    const float*  prior;             /* [R] root_equilibrium_distribution::compute(j): a FLOAT in the reference */
    const double* error_model;       /* [(M+1)][n_deviations] error_model::get_probs(x), or NULL */
} cafe_params;

/* Optional per-family outputs (host pointers, any may be NULL).
 *   base : family_lnl[f] = max_j(log L_j + log prior_j)      (results[i], base_model.cpp:105)
 *   gamma: category_likelihood[f*K+k] = max_j(L_j prior_j) p_k, family_likelihood[f] = sum_k
 *          (family_info_stash rows, gamma_core.cpp:212-216); failed[f] = 1 where a category's
 *          root vector summed to exactly 0 (gamma_core.cpp:152). */
typedef struct cafe_family_out {
    double*  family_lnl;
    double*  category_likelihood;
    double*  family_likelihood;
    int32_t* failed;
} cafe_family_out;

/* HIP-event timings and work counters of the last cafe_score (measurement). */
typedef struct cafe_stats {
    double ms_total;                 /* whole call, host wall */
    double ms_matrices;              /* K1 bd_matrix_build */
    double ms_prune;                 /* K2 prune_gemm + K3 leaf_gather over all nodes */
    double ms_gemm;                  /* K2 only */
    double ms_reduce;                /* K4 root_reduce */
    double gemm_flops;               /* sum over launches of 2*rows*(M+1)*columns, columns = the distinct subtree patterns the
                                        launch processes: every K tile of those columns.  What the tiles really ran (K2
                                        skips the K tiles outside matrix extent x panel extent): cafe_executed_flops */
    double gemm_bytes;               /* algorithmic bytes of the same launches (P + B read, C written) */
    double gemm_flops_per_family;    /* the same sum with one column per (distinct) family at every node: SURVEY 8d's
                                        per-family figure, what the launches would compute without the sharing */
    int64_t gemm_launches;
    int64_t n_matrices;              /* distinct (lambda_q, t_q) keys built */
    int64_t n_unique_families;
    int64_t n_chunks;
    int64_t matrix_bytes;
    int64_t panel_bytes;
    /* the schedule (fixed at cafe_create): */
    int64_t n_assemble_passes;       /* K3 launches that spread factor panels of de-duplicated children over a parent's columns */
    int64_t n_gather_epilogues;      /* K2 launches that fold a sibling's factor panel into their epilogue instead */
    int64_t n_leaf_passes;           /* K3 launches with leaf children only (cherries, polytomies, error models) */
    double gemm_flops_dense;         /* = gemm_flops */
} cafe_stats;

/* NULL on failure; err (optional, errlen bytes) receives the reason. */
cafe_ctx* cafe_create(const cafe_problem* problem, char* err, size_t errlen);
void      cafe_destroy(cafe_ctx* ctx);
const char* cafe_last_error(const cafe_ctx* ctx);
int       cafe_abi_version(void);

/* One model::infer_family_likelihoods call: writes -lnL (or +inf / NaN, see conventions). */
int cafe_score(cafe_ctx* ctx, const cafe_params* params, double* neg_lnl, const cafe_family_out* out);

/* The same work for a family shard, without the final host read-back: device_partial must point
 * to 2 doubles of DEVICE memory and receives {sum_f lnL_f, number of rejected/invalid families}
 * (+inf rejection that the host can decide alone is encoded as partial[1] = 1).  The kernels are
 * enqueued on `hip_stream` (a hipStream_t used as given: NULL is HIP's null stream, which is what
 * torch.cuda.current_stream() is by default) and the call returns without synchronising, so that
 * the caller can all-reduce the pair across ranks (RCCL) on the same stream: SURVEY.md 8e. */
int cafe_score_partial(cafe_ctx* ctx, const cafe_params* params, double* device_partial, void* hip_stream);
/* Turns the all-reduced pair into the scorer value: +inf if partial[1] > 0 else -partial[0]; NaN if partial[1] is NaN --
 * the mark of a shard whose call FAILED (cafe_score_partial returned an error: it then leaves {0, NaN} in its pair, best
 * effort, so that every rank of the caller's reduction learns of it). */
double cafe_finish_partial(const double host_partial[2]);

/* ---- Multi-GPU: families shard across the GPUs of a node, every GPU builds all matrices, and ONE
 * all-reduce (RCCL over xGMI) of the pair {sum lnL, rejects} closes a scorer call.  The reference has no counterpart:
 * its family loops are OpenMP (base_model.cpp:81-107, gamma_core.cpp:201-244).  Two ways to use it:
 *
 * (1) one process per GPU (torchrun, MPI, ...): every rank creates its context over ITS shard of the families (any
 *     partition gives the same -lnL; cafe_shard_plan balances the ranks), rank 0 makes an id with cafe_comm_unique_id
 *     and hands it to the others by whatever channel the launcher offers, and every rank calls cafe_comm_attach.  From
 *     then on cafe_score on every rank ends with ncclAllReduce(pair, 2 doubles, sum) on the context's stream and returns
 *     the WHOLE table's value on every rank; per-family results stay per shard.  Calls are collective: every rank must
 *     make the same sequence of cafe_score calls.  Ranks fail TOGETHER: a rank whose own call fails (HIP error,
 *     allocation, bad argument) still enters the all-reduce, with rejects = NaN, keeps its own error code, and every
 *     other rank returns CAFE_ERR_DEVICE ("another rank ... failed"); a rank that is gone altogether is caught by a
 *     deadline on the wait (environment CAFE_COMM_TIMEOUT_S at attach, default 120 s; <= 0 waits for ever): the waiting
 *     ranks abort their communicator (ncclCommAbort) and return CAFE_ERR_DEVICE.  After an abort the context has no
 *     communicator any more (attach again, or use it on its own). */
#define CAFE_COMM_ID_BYTES 128
int cafe_comm_unique_id(char id[CAFE_COMM_ID_BYTES]);
int cafe_comm_attach(cafe_ctx* ctx, const char id[CAFE_COMM_ID_BYTES], int32_t world_size, int32_t rank);
int cafe_comm_detach(cafe_ctx* ctx);

/* Balanced family partition for n_shards GPUs.  The device's unit of work is the DISTINCT pattern of leaf counts under
 * an interior node (cafe_create shares likelihood columns between families that agree on a whole subtree), so families
 * are ordered to put look-alikes next to each other (total size, then lexicographically) and cut into consecutive runs
 * of equal predicted device time.  order[n_families]: family indices in shard order; bounds[n_shards + 1]: shard r owns
 * order[bounds[r] .. bounds[r+1]).  Pure host code (no device needed); deterministic, so every rank derives the same
 * plan.  Only tree, counts and max sizes of `problem` are read. */
int cafe_shard_plan(const cafe_problem* problem, int32_t n_shards, int64_t* order, int64_t* bounds);
/* The same with a measured correction: family_scale[n_families] (table order), each in (0.1, 10).  After a few calls under
 * a first plan every rank knows how long its shard took; scale = that time / the mean over the ranks, for every family of
 * the shard, and the plan made with it moves the cuts so that the shards that ran long get less (what the prediction cannot
 * see -- how many K tiles a column's zero extent leaves at this lambda -- is in the measurement).  NULL: cafe_shard_plan. */
int cafe_shard_plan_scaled(const cafe_problem* problem, int32_t n_shards, const double* family_scale, int64_t* order, int64_t* bounds);

/* (2) one process, several GPUs: cafe_create_sharded plans the shards (cafe_shard_plan), creates one context per device
 *     of `devices` -- each driven by its own host thread and stream -- and joins them in one communicator
 *     (ncclCommInitAll).  cafe_sharded_score = model::infer_family_likelihoods over the whole table: all devices
 *     prune their shard concurrently, one all-reduce, one value.  problem->device is ignored. */
typedef struct cafe_sharded cafe_sharded;
cafe_sharded* cafe_create_sharded(const cafe_problem* problem, const int32_t* devices, int32_t n_devices, char* err, size_t errlen);
void      cafe_sharded_destroy(cafe_sharded* s);
const char* cafe_sharded_last_error(const cafe_sharded* s);
int cafe_sharded_score(cafe_sharded* s, const cafe_params* params, double* neg_lnl, const cafe_family_out* out);
/* per-family results of the last call, in the problem's family order (gathered from the shards) */
int cafe_sharded_family_results(cafe_sharded* s, const cafe_family_out* out);
int32_t cafe_sharded_size(const cafe_sharded* s);
/* shard r's context (owned by s): statistics, introspection */
cafe_ctx* cafe_sharded_context(cafe_sharded* s, int32_t r);

/* Per-family results of the last call (valid after the stream was synchronised). */
int cafe_family_results(cafe_ctx* ctx, const cafe_family_out* out);

/* P-value path.  For every family of the context: max_j L_root[j], the "observed max likelihood"
 * of compute_tree_pvalue (probability.cpp:391-399) and, on a context holding simulated families, the entries of
 * get_random_probabilities (probability.cpp:273-317, before its sort).  As in the reference the prune uses the
 * plain lambdas (one category, no multiplier), no error model and no prior: params->lambdas is the only field
 * read.  out[n_families].  cafe_family_results is not meaningful after this call. */
int cafe_root_max(cafe_ctx* ctx, const cafe_params* params, double* out);

/* compute_pvalues (probability.cpp:418-454) with the Monte-Carlo simulation on the device as well: n_simulations
 * families per root size 0..R-1 drawn from the transition-matrix rows (set_weighted_random_family_size, :320-351)
 * by a counter-based generator keyed by `seed`, pruned, sorted; pvalues[n_families] = max over root sizes of the
 * upper_bound position (:379-407).  Same distribution as the reference's procedure, a different random sample: agrees
 * with it to Monte-Carlo error, not draw for draw (the host path cafexp_amd/host/pvalues.cpp does the latter).
 * params->lambdas is the only field read; 1 <= n_simulations <= 2048. */
int cafe_pvalues(cafe_ctx* ctx, const cafe_params* params, int32_t n_simulations, uint64_t seed, double* pvalues);

/* Ancestral reconstruction.  Pupko's joint reconstruction as reconstruct_gene_family runs it
 * (gene_family_reconstructor.cpp:13-165; base_model.cpp:145, gamma_core.cpp:301): for every category k (one for
 * the base model; lambda * multipliers[k] for the gamma model) and family f, the reconstructed size of every node
 * -> states[k][f][node] (leaves carry their observed counts).  root_prior[j] = root_equilibrium_distribution::
 * compute(j) for j = 0..min(M,R) (one more entry than params->prior: the root scan of :47-62 reads compute(j) with
 * j as the SIZE, up to min(M,R) inclusive).  No error model is applied (:28-32 read the matrix column of the
 * observed count).  params: model, lambdas, n_categories, multipliers. */
int cafe_reconstruct(cafe_ctx* ctx, const cafe_params* params, const float* root_prior, int32_t* states);
/* compute_viterbi_sum (gene_family_reconstructor.cpp:361-400) for every family and node under the plain lambdas:
 * sizes[f][node] are the (reconstructed / observed) sizes, out[f][node] the branch probability, NaN where the
 * reference returns "invalid" (the root; parent size == child size). */
int cafe_branch_probabilities(cafe_ctx* ctx, const cafe_params* params, const int32_t* sizes, double* out);

/* Introspection for parity tests: the transition matrix the last call built for the branch above
 * `node` in category k (N x N row-major, N = max(M,R)+1: matrix_cache::get_matrix; for an interior
 * branch the columns c > M, which the prune never reads, are not materialised and come back 0), and the root
 * likelihood vector (R values: inference_prune's return) of family f in category k. */
int cafe_get_matrix(cafe_ctx* ctx, int32_t node, int32_t category, double* out, size_t out_len);
int cafe_get_root_likelihoods(cafe_ctx* ctx, int64_t family, int32_t category, double* out, size_t out_len);
/* The zero extents the last call worked with (parity tests: they must contain every non-zero).
 *   matrix_ext: the branch above `node` in `category` -- an interior branch: per block of 16 parent sizes s = 16b+1..16b+16
 *               the first / last child size with a non-zero entry, [ceil((N-1)/16)][2]; a leaf branch: per child size
 *               (column) x the first / last parent size with P[s][x] != 0, [N][2].  first > last: all zero.
 *   panel_ext:  interior non-root nodes, or NULL: per 128-column tile of the node's panel the first / last row (size) that
 *               may be non-zero, [columns / 128][2]; *n_tiles receives the number of tiles. */
int cafe_get_extents(cafe_ctx* ctx, int32_t node, int32_t category, int32_t* matrix_ext, size_t matrix_ext_len,
                     int32_t* panel_ext, size_t panel_ext_len, int32_t* n_tiles);
/* diagnostic: the per-COLUMN zero extents of an interior non-root node's panel in the last call, out[columns][2] (rows outside
 * [lo, hi] of a column are exactly zero; lo > hi: the whole column); *n_cols receives the panel's (padded) column count */
int cafe_debug_column_extents(cafe_ctx* ctx, int32_t node, int32_t category, int32_t* out, size_t out_len, int64_t* n_cols);
/* diagnostic: leaf branches whose matrix the context keeps a transposed copy of (leaf_reduce.hip, leaf_transpose_kernel: a
 * leaf that meets an interior sibling's factor in an assemble pass), and whether the last call used the copies (it does not
 * when it carries an error model) */
int cafe_debug_leaf_transposes(cafe_ctx* ctx, int32_t* n_branches, int32_t* used_by_last_call);
int cafe_get_stats(const cafe_ctx* ctx, cafe_stats* stats);
/* Flops the K2 launches of the last call executed: a (row tile, column tile) pair runs only the K tiles inside the
 * intersection of the matrix's non-zero extent (K1) and the panel's (extents.hip), and of those each 16-row block of the tile
 * only the ones inside its own extent -- the products left out all have an exact zero in them.  Reads the extents back and counts on the host (milliseconds): measurement only.  Needs a call that
 * was enqueued launch by launch (no graph replay). */
int cafe_executed_flops(cafe_ctx* ctx, double* flops);
/* diagnostic: the same count with every 16-row block of a tile taken over the tile's WHOLE K range (the hull of its blocks'
 * ranges) -- what the kernel issued until each block got its own range (prune_gemm.hip, block_ranges), and what rounds 2 and
 * 3a quoted their roofline fractions on; >= cafe_executed_flops */
int cafe_debug_tile_range_flops(cafe_ctx* ctx, double* flops);
/* diagnostic: the same per K2 launch of the last call, in launch order (executed[n], all_k_tiles[n] and tile_height[n] may be
 * NULL); n = cafe_stats.gemm_launches */
int cafe_debug_launch_flops(cafe_ctx* ctx, double* executed, double* all_k_tiles, int32_t* tile_height, size_t n);
int cafe_debug_launch_ms(cafe_ctx* ctx, double* ms, size_t n);      /* HIP-event duration of each (profiling on) */
/* diagnostic / test: reads back the tile lists the planner (extents.hip, tile_plan_kernel) laid out for the K2 launches of the
 * last call and checks that every tile appears exactly once with the K range its extents give; *n_planned = launches that ran
 * from a list, *worst_load = largest modelled workgroup load / mean load of its XCD.  CAFE_ERR_STATE on a mismatch. */
int cafe_debug_plan_check(cafe_ctx* ctx, int32_t* n_planned, double* worst_load);
int cafe_matrix_size(const cafe_ctx* ctx);
/* 1: bracket the phases and every K2 launch with HIP events so that cafe_stats.ms_* are measured (bench.py does).
 * 0 (default): no events. */
int cafe_set_profiling(cafe_ctx* ctx, int on);
/* 1: capture the call's fixed launch sequence once per (model, K) in a hipGraph and replay it (not while profiling);
 * 0 (default; environment CAFE_USE_GRAPH at cafe_create turns it on): enqueue launch by launch.  Same kernels, same
 * arguments, same bits; which is faster depends on the runtime. */
int cafe_set_graphs(cafe_ctx* ctx, int on);
/* diagnostic: K2's row tile is 16*mi rows, mi = 2..9, normally chosen per launch; mi forces one, 0 restores the choice */
int cafe_debug_force_tile(cafe_ctx* ctx, int mi);
/* test hook: the n-th next call of this context (n >= 1; 0 disarms) fails with CAFE_ERR_DEVICE behind its K1 launch, as a
 * HIP error in the middle of a call would -- pins the fail-together behaviour above */
int cafe_debug_fail_next(cafe_ctx* ctx, int n);
/* diagnostic (CAFE_GEMM_STAMPS=1 at cafe_create): per-block placement + timeline words of the last K2 launch */
int cafe_debug_stamps(cafe_ctx* ctx, unsigned long long* out, size_t words);

/* Stand-alone kernels exposed for unit parity tests and the roofline probe. */
/* builds `count` matrices of order n for (lambda[i], t[i]) pairs with matrix_cache_key quantization
 * applied (matrix_cache.h:42-61); out[count][n][n] is always P[s][c] row-major.  layout 0: the row-major
 * device layout leaf branches use; layout 1: the k-major layout of interior branches (built through the
 * reversibility relation, see bd_matrix.hip), converted back on the host. */
int cafe_build_matrices(int32_t device, int32_t n, int32_t count, const double* lambdas, const double* ts, int32_t layout, double* out);
/* back-to-back v_mfma_f64_16x16x4_f64 issue-rate probe: returns achieved TFLOP/s on `device`. */
int cafe_probe_fp64_mfma(int32_t device, double* tflops);

#ifdef __cplusplus
}
#endif
#endif
