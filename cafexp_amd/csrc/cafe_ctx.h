// Internal: the context behind the C ABI handle, shared by cafe_ctx.hip (scorer path) and reconstruct.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <map>
#include <string>
#include <vector>

#include "../../include/cafe_mi355x.h"
#include "cafe_kernels.h"

namespace cafe {

struct Op {
    int type;                   // 0 assemble (leaf gathers and/or factor panels of de-duplicated children), 1 gemm
    int parent;                 // node whose panel is written (for a factor GEMM: the node the factor belongs to)
    bool to_factor;             // gemm: plain store of P_child . L_child over the CHILD's distinct columns
    int dst_panel;
    int src_panel;              // gemm
    int child;                  // gemm: child node (its branch's matrix)
    int n_leaf;                 // gather
    int leaf_node[kMaxLeafPerOp];
    int mode;                   // 0 store, 1 multiply
    int n_src;                  // assemble: factor panels folded in (children with fewer distinct columns than the parent)
    int src_child[2];
    int src_panels[2];
    bool to_root;
    // gemm: the factor panel of the parent's OTHER interior child (fewer distinct columns than the parent), folded into
    // this launch's epilogue through the parent->child column map
    bool has_gath;
    int gath_child;
    int gath_panel;
    int step;                   // ops of one step are mutually independent (see Group)
    int desc;                   // index of the op's descriptor in the context's GemmOp / GatherArgs array
    bool leaf_t = false;        // assemble: every leaf child has a transposed copy of its matrix (GatherArgs::lt)
};

// A likelihood panel (or the transposed factor panel of a de-duplicated child): where it lives in the arena d_panels
struct Panel {
    int64_t cols = 0;           // columns (multiple of kBN)
    bool factor = false;        // transposed factor: [category][column][factor_ld]; else [category][rows_pad][column]
    int64_t offset = 0;         // doubles from d_panels
    int64_t kstride = 0;        // doubles between categories
    int first_step = 0, last_step = 0;   // written first / read last in these steps (arena planning)
};

// One launch: the ops of one step that share a kernel variant.  Steps come from the dependency graph of the ops (an op
// reads the panels of its children and, when it multiplies, what an earlier op left in its own): all ops of a step are
// independent, so a step is a level of ready nodes (one launch per height; reference loop core.cpp:133-144).
struct Group {
    int type = 0;               // 1: K2 (prune_gemm), 0: K3 (leaf gather / assemble)
    int step = 0;
    std::vector<int> ops;       // indices into cafe_ctx::ops, descriptors contiguous from first_desc
    int first_desc = 0;
    GemmVariant variant{0, 0, 0};
    bool to_root = false;
};

// What a call's launches need beyond the static descriptors, per (reduction, K): tile heights, tile lists.  The context
// has one for the stream path (uploaded when it changes) and every captured graph one of its own.
struct DescSet {
    GemmOp* d_gemm_ops = nullptr;
    PlanLaunch* d_plan_desc = nullptr;
    int2* d_plan = nullptr;
    std::vector<GemmOp> gemm_ops_sent;
    std::vector<PlanLaunch> plan_desc_sent;
    std::vector<int> group_mi, group_blocks, group_rounds;     // per K2 group (index = position among the K2 groups)
    std::vector<size_t> group_plan_off;
    bool plan_static_valid = false;          // no extents: the lists depend on (K, tile heights, chunk width) only
    int plan_static_K = 0;
    int64_t plan_static_cols = 0;
};

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
inline int64_t round_up64(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

}  // namespace cafe

struct cafe_ctx {
    // problem
    int n_nodes = 0, n_taxa = 0, M = 0, R = 0, N = 0, n_lambdas = 1, single_lambda = 1, Kmax = 1, n_dev = 0, device = 0;
    int root = -1;
    std::vector<int> parent, lam_idx, leaf_taxon;
    std::vector<double> blen;
    std::vector<std::vector<int>> children;
    int64_t F_all = 0, F_uniq = 0, Fp = 0;
    std::vector<int64_t> ref_of;            // family -> unique column
    std::vector<double> weights;

    // schedule
    std::vector<cafe::Op> ops;
    std::vector<cafe::Panel> panels;         // by panel id (Op::dst_panel, ...)
    std::vector<cafe::Group> groups;         // launch order
    bool grouped = false;                    // one column chunk: every panel has its own place, steps hold many ops
    int n_gemm_ops = 0, n_gather_ops = 0, n_gemm_groups = 0;
    std::vector<cafe::GemmOp> h_gemm_ops;    // static part of the K2 descriptors (n_row_tiles is filled per call)
    std::vector<cafe::GatherArgs> h_gather_ops;
    cafe::GatherArgs* d_gather_ops = nullptr;
    // transposed copies of the row-major matrices of the leaf branches that meet an interior sibling's factor in an assemble
    // pass (GatherArgs::lt): [lt_pairs.size()][Kmax][M + 1][factor_ld], filled after K1 by every call without an error model
    double* d_lt = nullptr;
    int32_t* d_lt_pairs = nullptr;
    std::vector<int32_t> lt_pairs;           // pair index in the row-major pool
    bool lt_used_last = false;
    cafe::GemmOp* h_gemm_stage = nullptr;    // pinned
    cafe::DescSet desc;                      // stream path
    bool panels_dirty = false;               // the last call returned NaN: stale NaNs may sit in padding rows; cleared before the next call
    // subtree-level de-duplication: a node's panel has one column per distinct pattern of leaf counts UNDER that node
    bool subtree_dedup = false;
    std::vector<int64_t> pat_cols;           // [n_nodes] padded distinct patterns of an interior node (0 for leaves)
    std::vector<char> edge_identity;         // [n_nodes] interior child whose columns are its parent's, in order
    std::vector<int32_t*> d_edge_map;        // [n_nodes] interior child: its column for every column of the parent
    std::vector<int32_t*> d_leaf_cnt;        // [n_nodes] interior parent: [its leaf children][its columns] observed counts
    std::vector<int> leaf_rank;              // [n_nodes] leaf: its row in the parent's d_leaf_cnt table
    int n_panels = 0, root_panel = -1;

    // device state
    bool device_ready = false;
    hipStream_t stream = nullptr;
    hipStream_t last_stream = nullptr;       // stream the last call was enqueued on
    int32_t* d_counts = nullptr;
    double* d_weights = nullptr;
    cafe::MatrixPool pool{nullptr, 0, 0, 0, 0, 0, 0};     // row-major matrices of leaf branches (K3)
    cafe::MatrixPool kpool{nullptr, 0, 0, 0, 0, 0, 1};    // k-major matrices of interior branches (K2)
    // Matrix slots are static: one per (layout, distinct quantized branch length, lambda index) PAIR and category,
    // slot = category * n_pairs[layout] + pair, so the slots of a call with K categories are a prefix of the pool and
    // slot_of never changes.  (Two pairs whose lambda * multiplier happen to quantize alike are built twice.)
    int n_pairs[2] = {0, 0};                 // [0] leaf branches (row-major pool), [1] interior branches (k-major pool)
    std::vector<int> pair_of;                // [n_nodes] pair of the branch above a node (in its layout)
    std::vector<long> pair_tq[2];            // quantized branch length of a pair (matrix_cache.h:47)
    std::vector<int> pair_lam[2];            // lambda index of a pair
    int n_distinct_pairs = 0;                // distinct (t_q, lambda index) over both layouts: matrices the reference would build per category
    int max_slots = 0, max_kslots = 0;
    // per-call parameter block, one device allocation mirrored by the pinned h_stage (one upload per call):
    // [SlotParam x max_slots][SlotParam x max_kslots][prior R][log prior R][category probabilities Kmax][error model]
    char* d_params = nullptr;
    size_t params_bytes = 0;
    cafe::SlotParam* d_slots = nullptr;                   // [max_slots] row-major, then [max_kslots] k-major
    double* d_panels = nullptr;
    int64_t panel_stride = 0;               // doubles per panel
    int64_t panel_kstride = 0;              // doubles per category inside a panel
    int rows_pad = 0, kc = 0;
    int factor_ld = 0;                       // rows per column of a transposed factor panel
    int64_t chunk_cols = 0;
    size_t workspace_limit = 0;             // cafe_problem::workspace_limit (0 = automatic)
    double *d_prior = nullptr, *d_logprior = nullptr, *d_catprobs = nullptr, *d_err = nullptr;
    double *d_fam_out = nullptr, *d_fam_lik = nullptr, *d_cat_out = nullptr;
    int32_t* d_failed = nullptr;
    double* d_scratch = nullptr;
    int n_scratch = 1024;
    double* d_result = nullptr;
    unsigned long long* d_stamps = nullptr;     // diagnostic block timeline of the LAST K2 launch (CAFE_GEMM_STAMPS=1)
    size_t stamps_words = 0;
    // pinned staging
    char* h_stage = nullptr;
    size_t stage_bytes = 0;
    double* h_result = nullptr;
    hipEvent_t ev_upload = nullptr;
    bool upload_pending = false;

    // multi-GPU: an RCCL communicator over the ranks that hold the other family shards (cafe_sharded.hip); with one
    // attached, cafe_score all-reduces {sum lnL, rejects} before the read-back
    void* comm = nullptr;                    // ncclComm_t
    bool comm_owned = false;
    int comm_world = 1, comm_rank = 0;
    // A call on a context with a communicator is collective.  A rank whose own enqueue fails still enters the all-reduce,
    // with rejects = NaN (h_poison), so that every rank returns an error instead of waiting for it; a rank that is gone
    // altogether is caught by the deadline below (the waiting ranks abort their communicator and return CAFE_ERR_DEVICE).
    double comm_timeout_s = 120.0;           // CAFE_COMM_TIMEOUT_S at cafe_comm_attach / cafe_create_sharded; <= 0: wait for ever
    double* h_poison = nullptr;              // pinned {0, NaN}
    int debug_fail_in = 0;                   // cafe_debug_fail_next: the n-th next enqueue fails behind its K1 launch

    // last call
    std::vector<int> slot_of;               // [node*Kmax + k]
    int K_last = 0, model_last = -1;
    bool last_rejected = false, have_results = false, rootmax_last = false;
    int n_slots_last = 0, n_kslots_last = 0;
    int64_t last_chunk_f0 = 0, last_chunk_nf = 0;

    // launch
    int n_cu = 0;                            // compute units of the device (K2's persistent grid)
    long stamps_launch = -1;                 // CAFE_GEMM_STAMPS_LAUNCH, read once at cafe_create
    // A call's enqueue sequence is fixed per (reduction, K): upload, K1, the schedule, K4.  It can be captured once in a
    // hipGraph and replayed (one launch call instead of ~40 to ~300).  Off by default: on ROCm 7.2 the replay measured no
    // faster than the stream enqueue at the mammals size (base 0.47 vs 0.48 ms) and slower with K = 4 (0.62 vs 0.41 ms).
    struct CallGraph {
        hipGraphExec_t exec = nullptr;
        cafe_stats stats{};                  // the work counters of the captured sequence
        cafe::DescSet desc;                  // its own descriptors and tile lists (frozen at capture)
    };
    std::map<int, CallGraph> graphs;
    int use_graph = 0;

    // zero extents of the likelihood panels (extents.hip): one descriptor per interior non-root node, grouped in levels of
    // mutually independent nodes (a node's level = 1 + its deepest interior child's); per node and category the per-column
    // and per-128-column-tile intervals outside which the panel is exactly zero
    bool panel_extents = false;
    bool no_asm_skip = false;                             // CAFE_NO_ASM_SKIP: the assemble pass writes every row (diagnostic)
    cafe::ExtNode* d_ext_nodes = nullptr;
    std::vector<int32_t*> d_colext, d_tileext;            // [n_nodes] (interior non-root nodes only)
    struct ExtLevel { int first, count, max_col_tiles; };
    std::vector<ExtLevel> ext_levels;

    // K2's row-tile height per launch is chosen from the non-zero extents of the PREVIOUS call's matrices (K1 publishes
    // them on the device; a copy lands in h_ext while the call's K2 launches run): the parameters of consecutive scorer
    // calls are close, and the choice only affects speed, never a bit of the result
    int32_t* h_ext = nullptr;                // pinned [max_kslots][ext_blocks][2]
    // tile lists of the K2 launches (tile_plan_kernel): one descriptor per launch, rebuilt per call (the tile heights may
    // change), uploaded when it differs from the last upload
    int plan_launches_last = 0;              // K2 launches of the last recorded call
    int plan_bias = 8;                       // percent by which the first-dispatched workgroup of a CU outruns the other (measured; CAFE_PLAN_BIAS)
    int plan_bias3[3] = {80, 100, 125};      // three workgroups per CU: cost of a K tile for the first / second / third dispatched (CAFE_PLAN_BIAS3=a,b,c)
    int plan_bias4[4] = {62, 88, 112, 160};  // four (CAFE_PLAN_BIAS4=a,b,c,d)
    int kb = 8;                              // depth of K2's K tiles: 8 (four workgroups per CU) or 16 (small matrices); CAFE_KB
    int plan_fixed = 8;                      // cost of an output tile beyond its K loop, in 8-deep K tiles (fitted: DESIGN.md; CAFE_PLAN_FIXED)
    size_t plan_entries = 0;
    cafe::PlanLaunch* h_plan_desc = nullptr;        // pinned
    const cafe::DescSet* desc_last = nullptr;       // descriptors of the last recorded call (diagnostics)
    bool h_ext_valid = false;
    int h_ext_K = 0;                         // categories of the call the copy belongs to

    // measurement
    struct GemmLaunch { int group; int K; int mi; int64_t cols; };   // what count_executed_flops needs (cols: the chunk's, several chunks only)
    std::vector<GemmLaunch> gemm_launches_info;
    bool stats_flops_stale = false;           // gemm_flops still holds the dense count of the last profiled call
    int profile = 0;
    int force_mi = 0;                        // diagnostic: K2 row-tile height for every launch (0 = chosen per launch)
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    std::vector<hipEvent_t> gemm_ev;
    size_t gemm_ev_used = 0;
    bool events_valid = false;
    cafe_stats stats{};

    std::string err;
};

namespace cafe {

void set_err(cafe_ctx* c, const char* fmt, ...);

#define HIP_TRY(c, expr)                                                                    \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) {                                                             \
            cafe::set_err(c, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return CAFE_ERR_DEVICE;                                                         \
        }                                                                                   \
    } while (0)

bool lambdas_valid(const cafe_ctx* c, const double* lam);
// One slot per distinct quantized (lambda * multiplier, t) and layout (matrix_cache.h:42-61), parameters uploaded on
// `s`, K1 launched: afterwards slot_of[node * Kmax + k] names the matrix of every branch and category.
int prepare_matrices(cafe_ctx* c, const double* lambdas, const double* multipliers, int K, hipStream_t s);
// Pupko reconstruction and Viterbi branch probabilities (reconstruct.hip)
int reconstruct_impl(cafe_ctx* c, const cafe_params* pr, const float* root_prior, int32_t* states);
int branch_probabilities_impl(cafe_ctx* c, const cafe_params* pr, const int32_t* sizes, double* out);
// Device-side p-values (pvalues.hip) and what it needs from cafe_ctx.hip: a context over the same tree whose family
// counts are written on the device, and the root-maximum prune of a context's families (-> d_fam_out, on stream s)
constexpr int32_t kFlagDeviceCounts = 0x40000000;        // internal cafe_problem flag
int pvalues_impl(cafe_ctx* c, const cafe_params* pr, int32_t n_simulations, uint64_t seed, double* pvalues);
cafe_ctx* create_child_for_device_counts(const cafe_ctx* parent, int64_t n_families);
void destroy_child(cafe_ctx* c);
int enqueue_rootmax(cafe_ctx* c, const double* lambdas, hipStream_t s);
// multi-GPU (cafe_sharded.hip)
int comm_allreduce_pair(cafe_ctx* c, double* d_pair, hipStream_t s);
void comm_release(cafe_ctx* c);
// waits for `s`; with a communicator attached: under the deadline comm_timeout_s and watching the communicator's
// asynchronous error state -- on either the communicator is aborted and CAFE_ERR_DEVICE returned
int comm_wait_stream(cafe_ctx* c, hipStream_t s);
void comm_abort(cafe_ctx* c);

}  // namespace cafe
