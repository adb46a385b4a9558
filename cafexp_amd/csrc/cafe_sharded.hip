// Multi-GPU scorer behind the C ABI (include/cafe_mi355x.h, "Multi-GPU").
//
// The reference parallelises one scorer call with OpenMP over families (base_model.cpp:81-107,
// gamma_core.cpp:201-244); its only cross-family operations are the final sum (base_model.cpp:107,
// gamma_core.cpp:244) and the any-failure test (gamma_core.cpp:227).  Here families shard across the GPUs of a node,
// every GPU builds all transition matrices itself, and ONE RCCL all-reduce over xGMI of {sum lnL, rejects} closes
// the call.  Host code only: the shard plan, the communicator plumbing and one host thread per device.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <numeric>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "cafe_ctx.h"

static_assert(CAFE_COMM_ID_BYTES >= sizeof(ncclUniqueId), "CAFE_COMM_ID_BYTES must hold an ncclUniqueId");

namespace cafe {

// {sum lnL, rejects} summed over the communicator's ranks, in place, on the stream the call was enqueued on
int comm_allreduce_pair(cafe_ctx* c, double* d_pair, hipStream_t s) {
    if (!c->comm) return CAFE_OK;
    const ncclResult_t r = ncclAllReduce(d_pair, d_pair, 2, ncclDouble, ncclSum, (ncclComm_t)c->comm, s);
    if (r != ncclSuccess) { set_err(c, "ncclAllReduce failed: %s", ncclGetErrorString(r)); return CAFE_ERR_DEVICE; }
    return CAFE_OK;
}

// The communicator is unusable (a peer is gone, or this rank could not even enqueue its part): abort it, so that whatever
// this rank still has in flight on it ends and cafe_destroy does not block in ncclCommDestroy.
void comm_abort(cafe_ctx* c) {
    if (c->comm && c->comm_owned) (void)ncclCommAbort((ncclComm_t)c->comm);
    c->comm = nullptr;
    c->comm_owned = false;
    c->comm_world = 1;
    c->comm_rank = 0;
}

int comm_wait_stream(cafe_ctx* c, hipStream_t s) {
    if (!c->comm || !(c->comm_timeout_s > 0)) { HIP_TRY(c, hipStreamSynchronize(s)); return CAFE_OK; }
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spin = 0;; ++spin) {
        const hipError_t e = hipStreamQuery(s);
        if (e == hipSuccess) { (void)hipGetLastError(); return CAFE_OK; }      // (the polls left hipErrorNotReady as the thread's last error)
        if (e != hipErrorNotReady) { set_err(c, "hipStreamQuery failed: %s", hipGetErrorString(e)); comm_abort(c); return CAFE_ERR_DEVICE; }
        if ((spin & 63) == 63) {
            ncclResult_t async = ncclSuccess;
            if (ncclCommGetAsyncError((ncclComm_t)c->comm, &async) == ncclSuccess && async != ncclSuccess && async != ncclInProgress) {
                set_err(c, "cafe_score: the communicator reports %s (rank %d of %d); aborted", ncclGetErrorString(async), c->comm_rank, c->comm_world);
                comm_abort(c);
                return CAFE_ERR_DEVICE;
            }
            const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (waited > c->comm_timeout_s) {
                set_err(c, "cafe_score: the all-reduce did not complete within %.0f s (CAFE_COMM_TIMEOUT_S): a rank of the communicator "
                           "is gone or never made this call (rank %d of %d); communicator aborted", c->comm_timeout_s, c->comm_rank, c->comm_world);
                comm_abort(c);
                return CAFE_ERR_DEVICE;
            }
            if (waited > 0.05) std::this_thread::sleep_for(std::chrono::microseconds(200));   // a call is milliseconds: past that, stop spinning
        }
    }
}

static double comm_timeout_from_env() {
    const char* e = std::getenv("CAFE_COMM_TIMEOUT_S");
    return e ? std::atof(e) : 120.0;
}

void comm_release(cafe_ctx* c) {
    if (c->comm && c->comm_owned) ncclCommDestroy((ncclComm_t)c->comm);
    c->comm = nullptr;
    c->comm_owned = false;
    c->comm_world = 1;
    c->comm_rank = 0;
}

namespace {

struct TreeShape {
    int n = 0, root = -1;
    std::vector<std::vector<int>> children;
    std::vector<int> leaf_taxon;
};

int read_tree(const cafe_problem* p, TreeShape& t) {
    if (!p || p->n_nodes < 3 || !p->parent || !p->leaf_taxon || !p->counts || p->n_families < 1 || p->n_taxa < 1) return CAFE_ERR_ARGUMENT;
    t.n = p->n_nodes;
    t.children.assign(t.n, {});
    t.leaf_taxon.assign(p->leaf_taxon, p->leaf_taxon + t.n);
    for (int v = 0; v < t.n; ++v) {
        const int par = p->parent[v];
        if (par < 0) { if (t.root >= 0) return CAFE_ERR_ARGUMENT; t.root = v; }
        else if (par >= t.n || par <= v) return CAFE_ERR_ARGUMENT;
        else t.children[par].push_back(v);
    }
    if (t.root < 0) return CAFE_ERR_ARGUMENT;
    for (int v = 0; v < t.n; ++v)
        if (t.children[v].empty() != (t.leaf_taxon[v] >= 0) || t.leaf_taxon[v] >= p->n_taxa) return CAFE_ERR_ARGUMENT;
    return CAFE_OK;
}

}  // namespace

constexpr double kFamilyCols = 5.0;   // what a family costs beyond its columns (K4, the passes near the root), in columns: fitted at the bench shape
// Shard plan.  prev[v][i]: for interior non-root node v, the previous position (in shard order) whose leaf counts under
// v equal those of position i, or -1: a shard [a, b) then holds #{i in [a,b) : prev[v][i] < a} distinct columns at v.
struct ShardModel {
    TreeShape tree;
    int64_t F = 0;
    std::vector<int64_t> order;                    // shard order -> family
    std::vector<int> nodes;                        // interior non-root nodes
    std::vector<std::vector<int64_t>> prev;        // [nodes.size()][F]
    std::vector<std::vector<float>> weight;        // [nodes.size()][F] cost of the position's column at that node (see shard_cost)
    std::vector<double> fam_cum;                   // [F + 1] prefix sums of the per-family term (kFamilyCols columns' worth x the family's scale)
    int rows_inner = 0, rows_root = 0, categories = 1;
};

// family_scale (or nullptr): a measured correction per family, table order -- the time a shard of an earlier plan took over
// the mean of its plan's shards, for every family of that shard (cafe_shard_plan_scaled): multiplies whatever the family
// contributes to a shard's predicted time.
int build_shard_model(const cafe_problem* p, ShardModel& m, const double* family_scale = nullptr) {
    const int rc = read_tree(p, m.tree);
    if (rc != CAFE_OK) return rc;
    const int T = p->n_taxa;
    const int64_t F = m.F = p->n_families;
    m.rows_inner = p->max_family_size + 1;
    m.rows_root = p->max_root_family_size;
    m.categories = std::max(1, p->max_categories);
    // look-alikes next to each other: total size, then the rows lexicographically (stable: equal rows keep table order)
    std::vector<int64_t> total(F, 0);
    for (int64_t f = 0; f < F; ++f)
        for (int t = 0; t < T; ++t) total[f] += p->counts[f * T + t];
    m.order.resize(F);
    std::iota(m.order.begin(), m.order.end(), (int64_t)0);
    std::stable_sort(m.order.begin(), m.order.end(), [&](int64_t a, int64_t b) {
        if (total[a] != total[b]) return total[a] < total[b];
        return std::lexicographical_compare(p->counts + a * T, p->counts + (a + 1) * T, p->counts + b * T, p->counts + (b + 1) * T);
    });
    // pattern ids per node in shard order, children first (key = the children's ids and leaf counts)
    std::vector<std::vector<int32_t>> pid(m.tree.n);
    for (int v = 0; v < m.tree.n; ++v) {
        if (m.tree.leaf_taxon[v] >= 0 || v == m.tree.root) continue;
        std::vector<int> inner, leaves;
        for (int u : m.tree.children[v]) (m.tree.leaf_taxon[u] < 0 ? inner : leaves).push_back(u);
        const size_t kw = inner.size() + leaves.size();
        std::unordered_map<std::string, std::pair<int32_t, int64_t>> seen;     // key -> (pattern id, last position)
        seen.reserve((size_t)F * 2);
        pid[v].resize(F);
        std::vector<int64_t> pv(F);
        std::vector<float> wt(F);
        std::vector<int> under;                        // taxa under v
        {
            std::vector<int> stack(1, v);
            while (!stack.empty()) {
                const int w = stack.back(); stack.pop_back();
                if (m.tree.leaf_taxon[w] >= 0) under.push_back(m.tree.leaf_taxon[w]);
                for (int u : m.tree.children[w]) stack.push_back(u);
            }
        }
        std::vector<int32_t> key(kw);
        for (int64_t i = 0; i < F; ++i) {
            size_t k = 0;
            for (int u : inner) key[k++] = pid[u][i];
            for (int u : leaves) key[k++] = p->counts[m.order[i] * T + m.tree.leaf_taxon[u]];
            int32_t big = 0;                           // largest count under v: K2 skips the all-zero rows of a column tile, and
            for (int t : under) big = std::max(big, p->counts[m.order[i] * T + t]);   // columns of large families have few
            // (fitted at M = 720: weight 1 up to a largest count of 100 = 0.14 M, +2 per M beyond)
            wt[i] = 1.0f;
            if (std::max(p->max_family_size, p->max_root_family_size) + 1 >= 256)      // (the library keeps no zero extents below that order: cafe_create)
                wt[i] += 2.0f * std::max(0.0f, (float)big - 0.14f * (float)p->max_family_size) / (float)std::max(1, p->max_family_size);
            if (family_scale) wt[i] *= (float)family_scale[m.order[i]];
            std::string ks(reinterpret_cast<const char*>(key.data()), sizeof(int32_t) * kw);
            auto it = seen.find(ks);
            if (it == seen.end()) {
                pv[i] = -1;
                pid[v][i] = (int32_t)seen.size();
                seen.emplace(std::move(ks), std::make_pair(pid[v][i], i));
            } else {
                pv[i] = it->second.second;
                pid[v][i] = it->second.first;
                it->second.second = i;
            }
        }
        m.nodes.push_back(v);
        m.prev.push_back(std::move(pv));
        m.weight.push_back(std::move(wt));
    }
    m.fam_cum.assign(F + 1, 0.0);
    for (int64_t i = 0; i < F; ++i) m.fam_cum[i + 1] = m.fam_cum[i] + kFamilyCols * (family_scale ? family_scale[m.order[i]] : 1.0);
    return CAFE_OK;
}

// Predicted device time of the shard [a, b), in columns: every interior branch costs one K2 launch and the memory passes
// of the node's panel, both linear in the node's distinct columns (the per-node constants -- padding, a launch's fixed cost --
// are the same for every shard and are left out); the two branches under the root run over one column per family, and what
// else is linear in the families (the assemble passes near the root, K4) adds 5 columns' worth per family (fitted: with 2 the
// shard of the smallest families, 11 500 of 50 000, ran 1 ms longer than the others outside K2).  A
// column's weight is 1 up to a largest count of 100 under the node and grows by 2 per M beyond: K2 runs only the K tiles
// inside matrix extent x panel extent, and the panels of large families have wide extents (measured on eight shards cut by
// plain column counts: the seven with the small families 27.0-28.9 ms, the one with the largest 32.2 ms).
// (Round 2 also tried costing a launch the way the launcher picks its row tile, whole rounds of the persistent grid x tile
// height: before K2 skipped K tiles that balanced eight shards better than column counts; tiles of unequal length no
// longer run in lockstep rounds.)
double shard_cost(const ShardModel& m, int64_t a, int64_t b) {
    double cost = 0;
    for (size_t j = 0; j < m.nodes.size(); ++j) {
        const std::vector<int64_t>& pv = m.prev[j];
        const std::vector<float>& wt = m.weight[j];
        double cols = 0;
        for (int64_t i = a; i < b; ++i) cols += pv[i] < a ? wt[i] : 0.0f;
        cost += cols;                                  // (+ half a column tile of padding and the op's fixed cost: the same for every
                                                       //  shard -- every shard runs every node -- so they drop out of the comparison)
    }
    return cost + (m.fam_cum[b] - m.fam_cum[a]);
}

int plan_shards(const ShardModel& m, int n_shards, std::vector<int64_t>& bounds) {
    const int64_t F = m.F;
    if (n_shards < 1 || n_shards > F) return CAFE_ERR_ARGUMENT;
    bounds.assign(n_shards + 1, 0);
    bounds[n_shards] = F;
    if (n_shards == 1) return CAFE_OK;
    // start: equal cumulative cost, a family's cost being its own term plus the (weighted) number of nodes at which it is
    // the first of the whole table (in shard order) to show its pattern
    std::vector<double> cum(F, 0.0);
    for (int64_t i = 0; i < F; ++i) cum[i] = m.fam_cum[i + 1] - m.fam_cum[i];
    for (size_t j = 0; j < m.nodes.size(); ++j)
        for (int64_t i = 0; i < F; ++i) cum[i] += m.prev[j][i] < 0 ? m.weight[j][i] : 0.0f;
    for (int64_t i = 1; i < F; ++i) cum[i] += cum[i - 1];
    for (int r = 1; r < n_shards; ++r)
        bounds[r] = std::lower_bound(cum.begin(), cum.end(), cum[F - 1] * r / n_shards) - cum.begin();
    auto fix = [&](std::vector<int64_t>& bd) {           // never an empty shard
        bd[0] = 0; bd[n_shards] = F;
        for (int r = 1; r < n_shards; ++r) bd[r] = std::max(bd[r], bd[r - 1] + 1);
        for (int r = n_shards - 1; r >= 1; --r) bd[r] = std::min(bd[r], bd[r + 1] - 1);
    };
    fix(bounds);
    std::vector<double> cost(n_shards);
    auto eval = [&](const std::vector<int64_t>& bd) {
        double hi = 0;
        for (int r = 0; r < n_shards; ++r) { cost[r] = shard_cost(m, bd[r], bd[r + 1]); hi = std::max(hi, cost[r]); }
        return hi;
    };
    // 1. a pattern shared across a cut is paid on both sides, which the start ignores: move every cut towards the
    //    cheaper neighbour in proportion to the imbalance, a few damped sweeps
    std::vector<int64_t> best = bounds;
    double best_hi = eval(bounds);
    for (int sweep = 0; sweep < 10; ++sweep) {
        std::vector<int64_t> nb = bounds;
        for (int r = 1; r < n_shards; ++r) {
            const double cl = cost[r - 1], cr = cost[r];
            const double per_l = cl / (double)(bounds[r] - bounds[r - 1]), per_r = cr / (double)(bounds[r + 1] - bounds[r]);
            nb[r] = bounds[r] + (int64_t)std::llround(0.35 * (cr - cl) / (per_l + per_r));
        }
        fix(nb);
        bounds = nb;
        const double hi = eval(bounds);
        if (hi < best_hi) { best_hi = hi; best = bounds; }
    }
    // 2. the cost is a step function of a shard's column counts (whole column tiles): a local search over single cuts,
    //    steps from 1/16 of a shard down to 8 families, lowers the largest of the two neighbours
    bounds = best;
    eval(bounds);
    for (int64_t step = std::max<int64_t>(8, F / n_shards / 16); step >= 8; step /= 2) {
        for (int pass = 0; pass < 2; ++pass)
            for (int r = 1; r < n_shards; ++r)
                for (int dir = -1; dir <= 1; dir += 2) {
                    const int64_t nb = bounds[r] + dir * step;
                    if (nb <= bounds[r - 1] || nb >= bounds[r + 1]) continue;
                    const double cl = shard_cost(m, bounds[r - 1], nb), cr = shard_cost(m, nb, bounds[r + 1]);
                    if (std::max(cl, cr) < std::max(cost[r - 1], cost[r]) * (1.0 - 1e-12)) { bounds[r] = nb; cost[r - 1] = cl; cost[r] = cr; }
                }
    }
    return CAFE_OK;
}

}  // namespace cafe

// ---------------------------------------------------------------------------------------------------------------
struct cafe_sharded {
    std::vector<cafe_ctx*> ctx;
    std::vector<int> devices;
    std::vector<int64_t> order, bounds;
    int64_t F = 0;
    std::string err;
    // one host thread per device
    struct Worker {
        std::thread th;
        std::mutex mu;
        std::condition_variable cv;
        std::function<void()> job;
        bool pending = false, quit = false;
    };
    std::vector<Worker*> workers;
    void run_all(const std::function<void(int)>& fn) {
        for (size_t r = 0; r < workers.size(); ++r) {
            Worker* w = workers[r];
            std::lock_guard<std::mutex> lk(w->mu);
            w->job = [fn, r]() { fn((int)r); };
            w->pending = true;
            w->cv.notify_all();
        }
        for (Worker* w : workers) {
            std::unique_lock<std::mutex> lk(w->mu);
            w->cv.wait(lk, [w] { return !w->pending; });
        }
    }
    void start_workers(int n) {
        for (int r = 0; r < n; ++r) {
            Worker* w = new Worker();
            w->th = std::thread([w]() {
                for (;;) {
                    std::unique_lock<std::mutex> lk(w->mu);
                    w->cv.wait(lk, [w] { return w->pending || w->quit; });
                    if (w->quit) return;
                    std::function<void()> job = std::move(w->job);
                    lk.unlock();
                    job();
                    lk.lock();
                    w->pending = false;
                    w->cv.notify_all();
                }
            });
            workers.push_back(w);
        }
    }
    void stop_workers() {
        for (Worker* w : workers) {
            { std::lock_guard<std::mutex> lk(w->mu); w->quit = true; w->cv.notify_all(); }
            w->th.join();
            delete w;
        }
        workers.clear();
    }
};

extern "C" {

int cafe_comm_unique_id(char id[CAFE_COMM_ID_BYTES]) {
    if (!id) return CAFE_ERR_ARGUMENT;
    ncclUniqueId u;
    if (ncclGetUniqueId(&u) != ncclSuccess) return CAFE_ERR_DEVICE;
    std::memset(id, 0, CAFE_COMM_ID_BYTES);
    std::memcpy(id, &u, sizeof u);
    return CAFE_OK;
}

int cafe_comm_attach(cafe_ctx* ctx, const char id[CAFE_COMM_ID_BYTES], int32_t world_size, int32_t rank) {
    if (!ctx) return CAFE_ERR_ARGUMENT;
    if (!id || world_size < 1 || rank < 0 || rank >= world_size) { cafe::set_err(ctx, "cafe_comm_attach: bad id / world size / rank"); return CAFE_ERR_ARGUMENT; }
    cafe::comm_release(ctx);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof u);
    ncclComm_t comm = nullptr;
    const ncclResult_t r = ncclCommInitRank(&comm, world_size, u, rank);
    if (r != ncclSuccess) { cafe::set_err(ctx, "ncclCommInitRank failed: %s", ncclGetErrorString(r)); return CAFE_ERR_DEVICE; }
    ctx->comm = comm;
    ctx->comm_owned = true;
    ctx->comm_world = world_size;
    ctx->comm_rank = rank;
    ctx->comm_timeout_s = cafe::comm_timeout_from_env();
    return CAFE_OK;
}

int cafe_comm_detach(cafe_ctx* ctx) {
    if (!ctx) return CAFE_ERR_ARGUMENT;
    if (ctx->device_ready) { (void)hipSetDevice(ctx->device); if (ctx->stream) (void)hipStreamSynchronize(ctx->stream); }
    cafe::comm_release(ctx);
    return CAFE_OK;
}

int cafe_shard_plan(const cafe_problem* problem, int32_t n_shards, int64_t* order, int64_t* bounds) {
    return cafe_shard_plan_scaled(problem, n_shards, nullptr, order, bounds);
}

int cafe_shard_plan_scaled(const cafe_problem* problem, int32_t n_shards, const double* family_scale, int64_t* order, int64_t* bounds) {
    if (!problem || !order || !bounds || n_shards < 1) return CAFE_ERR_ARGUMENT;
    if (family_scale)
        for (int64_t f = 0; f < problem->n_families; ++f)
            if (!(family_scale[f] > 0.1 && family_scale[f] < 10.0)) return CAFE_ERR_ARGUMENT;
    try {
        cafe::ShardModel m;
        int rc = cafe::build_shard_model(problem, m, family_scale);
        if (rc != CAFE_OK) return rc;
        std::vector<int64_t> b;
        rc = cafe::plan_shards(m, n_shards, b);
        if (rc != CAFE_OK) return rc;
        std::copy(m.order.begin(), m.order.end(), order);
        std::copy(b.begin(), b.end(), bounds);
        return CAFE_OK;
    } catch (const std::exception&) {
        return CAFE_ERR_MEMORY;
    }
}

cafe_sharded* cafe_create_sharded(const cafe_problem* problem, const int32_t* devices, int32_t n_devices, char* err, size_t errlen) {
    auto fail = [&](const std::string& msg) -> cafe_sharded* {
        if (err && errlen) std::snprintf(err, errlen, "%s", msg.c_str());
        return nullptr;
    };
    if (!problem || !devices || n_devices < 1) return fail("cafe_create_sharded: problem, devices and n_devices >= 1 are required");
    if (n_devices > problem->n_families) return fail("cafe_create_sharded: more devices than families");
    for (int i = 0; i < n_devices; ++i)
        for (int j = 0; j < i; ++j)
            if (devices[i] == devices[j]) return fail("cafe_create_sharded: a device is listed twice (RCCL needs one rank per device)");
    cafe_sharded* s = nullptr;
    try {
        s = new cafe_sharded();
        s->F = problem->n_families;
        s->order.resize(s->F);
        s->bounds.resize(n_devices + 1);
        if (cafe_shard_plan(problem, n_devices, s->order.data(), s->bounds.data()) != CAFE_OK) { delete s; return fail("cafe_create_sharded: malformed problem"); }
        s->devices.assign(devices, devices + n_devices);
        s->ctx.assign(n_devices, nullptr);
        s->start_workers(n_devices);
        const int T = problem->n_taxa;
        std::vector<std::string> errs(n_devices);
        s->run_all([&](int r) {                              // the contexts are built concurrently, each by the thread that will drive it
            try {                                            // (an exception must not leave a worker thread: std::terminate)
                const int64_t lo = s->bounds[r], n = s->bounds[r + 1] - lo;
                std::vector<int32_t> counts((size_t)n * T);
                for (int64_t i = 0; i < n; ++i) std::memcpy(&counts[(size_t)i * T], problem->counts + s->order[lo + i] * T, sizeof(int32_t) * T);
                cafe_problem pb = *problem;
                pb.counts = counts.data();
                pb.n_families = n;
                pb.device = s->devices[r];
                char e[512] = {0};
                s->ctx[r] = cafe_create(&pb, e, sizeof e);
                if (!s->ctx[r]) errs[r] = e;
            } catch (const std::exception& e) {
                s->ctx[r] = nullptr;
                errs[r] = e.what();
            }
        });
        for (int r = 0; r < n_devices; ++r)
            if (!s->ctx[r]) { const std::string msg = "cafe_create_sharded: shard " + std::to_string(r) + ": " + errs[r]; cafe_sharded_destroy(s); return fail(msg); }
        std::vector<ncclComm_t> comms(n_devices);
        const ncclResult_t nr = ncclCommInitAll(comms.data(), n_devices, s->devices.data());
        if (nr != ncclSuccess) { const std::string msg = std::string("cafe_create_sharded: ncclCommInitAll failed: ") + ncclGetErrorString(nr); cafe_sharded_destroy(s); return fail(msg); }
        for (int r = 0; r < n_devices; ++r) {
            s->ctx[r]->comm = comms[r]; s->ctx[r]->comm_owned = true; s->ctx[r]->comm_world = n_devices; s->ctx[r]->comm_rank = r;
            s->ctx[r]->comm_timeout_s = cafe::comm_timeout_from_env();
        }
    } catch (const std::exception& e) {
        if (s) cafe_sharded_destroy(s);
        return fail(std::string("cafe_create_sharded: ") + e.what());
    }
    if (err && errlen) err[0] = 0;
    return s;
}

void cafe_sharded_destroy(cafe_sharded* s) {
    if (!s) return;
    for (cafe_ctx* c : s->ctx) if (c) cafe_destroy(c);      // releases the communicators as well
    s->stop_workers();
    delete s;
}

const char* cafe_sharded_last_error(const cafe_sharded* s) { return s ? s->err.c_str() : "null sharded context"; }
int32_t cafe_sharded_size(const cafe_sharded* s) { return s ? (int32_t)s->ctx.size() : 0; }
cafe_ctx* cafe_sharded_context(cafe_sharded* s, int32_t r) { return (s && r >= 0 && r < (int32_t)s->ctx.size()) ? s->ctx[r] : nullptr; }

int cafe_sharded_score(cafe_sharded* s, const cafe_params* params, double* neg_lnl, const cafe_family_out* out) {
    if (!s || !neg_lnl) return CAFE_ERR_ARGUMENT;
    const int n = (int)s->ctx.size();
    std::vector<int> rc(n, CAFE_OK);
    std::vector<double> value(n, 0.0);
    // every worker: enqueue its shard, all-reduce the pair on its stream, read it back.  The all-reduce makes the call
    // collective; argument errors are detected identically on every shard before anything is enqueued.
    // A shard whose own enqueue fails still enters the all-reduce (with a poisoned pair, cafe_score), so every worker comes
    // back; the shard that failed first-hand keeps its own code and message, the others report "another rank failed".
    s->run_all([&](int r) {
        try { rc[r] = cafe_score(s->ctx[r], params, &value[r], nullptr); }
        catch (const std::exception& e) { cafe::set_err(s->ctx[r], "cafe_sharded_score: %s", e.what()); rc[r] = CAFE_ERR_MEMORY; }
    });
    int bad = -1;
    for (int r = 0; r < n; ++r)
        if (rc[r] != CAFE_OK && (bad < 0 || std::strstr(cafe_last_error(s->ctx[bad]), "another rank"))) bad = r;
    if (bad >= 0) { s->err = "shard " + std::to_string(bad) + ": " + cafe_last_error(s->ctx[bad]); return rc[bad]; }
    *neg_lnl = value[0];                                     // identical on every rank after the all-reduce
    if (out) return cafe_sharded_family_results(s, out);
    return CAFE_OK;
}

int cafe_sharded_family_results(cafe_sharded* s, const cafe_family_out* out) {
    if (!s || !out) return CAFE_ERR_ARGUMENT;
    const int n = (int)s->ctx.size();
    std::vector<int> rc(n, CAFE_OK);
    int K = 1;
    for (cafe_ctx* c : s->ctx) K = std::max(K, c->K_last);
    s->run_all([&](int r) {
        const int64_t lo = s->bounds[r], nf = s->bounds[r + 1] - lo;
        std::vector<double> lnl, cat, lik;
        std::vector<int32_t> failed;
        cafe_family_out o = {};
        if (out->family_lnl) { lnl.resize(nf); o.family_lnl = lnl.data(); }
        if (out->category_likelihood) { cat.resize((size_t)nf * K); o.category_likelihood = cat.data(); }
        if (out->family_likelihood) { lik.resize(nf); o.family_likelihood = lik.data(); }
        if (out->failed) { failed.resize(nf); o.failed = failed.data(); }
        try { rc[r] = cafe_family_results(s->ctx[r], &o); }
        catch (const std::exception&) { rc[r] = CAFE_ERR_MEMORY; }
        if (rc[r] != CAFE_OK) return;
        const bool gamma = s->ctx[r]->model_last == CAFE_MODEL_GAMMA;
        for (int64_t i = 0; i < nf; ++i) {                   // shards write disjoint families
            const int64_t f = s->order[lo + i];
            if (out->family_lnl) out->family_lnl[f] = lnl[i];
            if (gamma && out->family_likelihood) out->family_likelihood[f] = lik[i];
            if (gamma && out->category_likelihood) std::memcpy(out->category_likelihood + f * K, &cat[(size_t)i * K], sizeof(double) * K);
            if (out->failed) out->failed[f] = failed[i];
        }
    });
    for (int r = 0; r < n; ++r)
        if (rc[r] != CAFE_OK) { s->err = "shard " + std::to_string(r) + ": " + cafe_last_error(s->ctx[r]); return rc[r]; }
    return CAFE_OK;
}

}  // extern "C"
