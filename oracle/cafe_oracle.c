/* TEST INFRASTRUCTURE -- the parity oracle, NOT product code (see cafe_oracle.h).
 *
 * CPU restatement of the CAFE5 birth-death likelihood path.  All file:line
 * citations are relative to the reference root (Han9527/CAFExp).
 */
#define _GNU_SOURCE
#include "cafe_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

void orc_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ---------------------------------------------------------------------------------
 * log-binomials.  probability.cpp:58-88.  The reference caches lgamma(i) for integer
 * i < 1024 and ln C(n,r) for n,r < 100; both caches hold exactly what libm's lgamma
 * returns for the same integer argument, so calling lgamma() directly is value-identical.
 * --------------------------------------------------------------------------------- */
#define LGAMMA_TABLE 8192
static double lgamma_table[LGAMMA_TABLE];
static int lgamma_table_ready = 0;
static void lgamma_table_init(void) {             /* init_lgamma_cache, probability.cpp:66 */
#pragma omp critical(orc_lgamma_init)
    {
        if (!lgamma_table_ready) {
            for (int i = 0; i < LGAMMA_TABLE; ++i) { int sign; lgamma_table[i] = lgamma_r((double)i, &sign); }
            lgamma_table_ready = 1;
        }
    }
}
static double lgamma_int(double n) {
    if (n >= 0 && n < LGAMMA_TABLE && n == (double)(int)n) {
        if (!lgamma_table_ready) lgamma_table_init();
        return lgamma_table[(int)n];
    }
    int sign;
    return lgamma_r(n, &sign);
}

double orc_chooseln(double n, double r) {          /* probability.cpp:79 */
    if (r == 0 || (n == 0 && r == 0)) return 0;
    else if (n <= 0 || r <= 0) return log(0.0);
    return lgamma_int(n + 1) - lgamma_int(r + 1) - lgamma_int(n - r + 1);
}

/* probability.cpp:101-145 (the m < 10000 branch; the other one is unreachable at the
 * family sizes CAFE allows).  Terms are summed left to right like std::accumulate. */
double orc_bd_log_alpha(int s, int c, double log_alpha, double coeff) {
    int m = c < s ? c : s;
    int s_add_c = s + c;
    int s_add_c_sub_1 = s_add_c - 1;
    int s_sub_1 = s - 1;
    double result = 0.0;
    for (int j = 0; j <= m; j++) {
        double t = orc_chooseln(s, j) + orc_chooseln(s_add_c_sub_1 - j, s_sub_1) + (s_add_c - 2 * j) * log_alpha;
        double p = exp(t) * pow(coeff, j);
        result += p;
    }
    double lo = result < 1.0 ? result : 1.0;   /* std::max(std::min(result, 1.0), 0.0) */
    return lo > 0.0 ? lo : 0.0;
}

/* probability.cpp:147-164.  (Parent size 0 is special-cased one level up, in
 * matrix_cache::get_from_parent_fam_size_to_c, matrix_cache.cpp:70-77: see build_rows_direct.) */
double orc_bd_prob(double lambda, double t, int s, int c) {
    double alpha = lambda * t / (1 + lambda * t);
    double coeff = 1 - 2 * alpha;
    double result = 0;
    if (coeff > 0 && coeff != 1)
        result = orc_bd_log_alpha(s, c, log(alpha), coeff);
    return result;
}

/* matrix_cache.h:42-61: keys keep 9 digits of lambda and 3 of the branch length, and
 * precalculate_matrices (matrix_cache.cpp:148-149) computes from the de-quantized key. */
static void quantize_key(double lambda, double t, long* lq, long* tq) {
    *lq = (long)(lambda * 1000000000);
    *tq = (long)(t * 1000);
}
void orc_quantize(double lambda, double t, double* lambda_q, double* t_q) {
    long lq, tq;
    quantize_key(lambda, t, &lq, &tq);
    *lambda_q = (double)lq / 1000000000.0;
    *t_q = (double)tq / 1000.0;
}

int orc_is_saturated(double t, double lambda) {    /* matrix_cache.cpp:115 */
    double alpha = lambda * t / (1 + lambda * t);
    return (1 - 2 * alpha) < 0;
}

/* one matrix, reference algorithm: matrix_cache.cpp:144-164 */
static void build_rows_direct(int n, double lq, double tq, double* out, int row_lo, int row_hi) {
    int sat = orc_is_saturated(tq, lq);
    for (int s = row_lo; s < row_hi; ++s) {
        double* row = out + (size_t)s * n;
        if (s == 0) {
            for (int c = 0; c < n; ++c) row[c] = 0.0;
            row[0] = 1.0;
            continue;
        }
        if (sat) { for (int c = 0; c < n; ++c) row[c] = 0.0; continue; }
        for (int c = 0; c < n; ++c) row[c] = orc_bd_prob(lq, tq, s, c);
    }
}

void orc_build_matrix(int n, double lambda, double t, double* out) {
    double lq, tq;
    orc_quantize(lambda, t, &lq, &tq);
#pragma omp parallel for schedule(dynamic, 1)
    for (int s = 0; s < n; ++s) build_rows_direct(n, lq, tq, out, s, s + 1);
}

/* NOT the reference's algorithm: the same matrix through the s-fold convolution of the
 * single-lineage law p1(0)=a, p1(k)=(1-a)^2 a^(k-1) (SURVEY.md section 7 "Hard parts").
 * Kept in the oracle only as an independent cross-check of the O(N^2) device kernel's
 * algebra and to make large-N fixtures affordable; validated against orc_build_matrix
 * in tests/test_oracle_matrix.py. */
static void build_matrix_conv_q(int n, double lq, double tq, double* out) {
    memset(out, 0, sizeof(double) * (size_t)n * n);
    out[0] = 1.0;
    double alpha = lq * tq / (1 + lq * tq);
    double coeff = 1 - 2 * alpha;
    if (coeff < 0) return;                       /* saturated */
    if (!(coeff > 0 && coeff != 1)) return;      /* probability.cpp:154 */
    double oma2 = (1 - alpha) * (1 - alpha);
    for (int s = 1; s < n; ++s) {
        const double* prev = out + (size_t)(s - 1) * n;
        double* row = out + (size_t)s * n;
        double h = 0.0;
        for (int c = 0; c < n; ++c) {
            h = (c > 0 ? prev[c - 1] : 0.0) + alpha * h;
            double v = alpha * prev[c] + oma2 * h;
            row[c] = v;
        }
    }
    for (size_t i = (size_t)n; i < (size_t)n * n; ++i) {
        double v = out[i];
        v = v < 1.0 ? v : 1.0;
        out[i] = v > 0.0 ? v : 0.0;
    }
}
void orc_build_matrix_conv(int n, double lambda, double t, double* out) {
    double lq, tq;
    orc_quantize(lambda, t, &lq, &tq);
    build_matrix_conv_q(n, lq, tq, out);
}

/* matrix_cache.cpp:28-57 (the non-BLAS fallback): plain double loop, ascending c */
void orc_matvec(const double* mat, int n, const double* v, int s_min, int s_max, int c_min, int c_max, double* out) {
    for (int s = s_min; s <= s_max; s++) {
        double acc = 0;
        const double* row = mat + (size_t)s * n;
        for (int c = c_min; c <= c_max; c++) acc += row[c] * v[c - c_min];
        out[s - s_min] = acc;
    }
}

/* ---------------------------------------------------------------------------------
 * matrix cache keyed like matrix_cache (matrix_cache.h:42-61, matrix_cache.cpp:80-97,121-171)
 * --------------------------------------------------------------------------------- */
typedef struct { long lq, tq; double* m; } mat_entry;
typedef struct { int n; int count; int cap; mat_entry* e; } mat_cache;

static void cache_init(mat_cache* c, int n) { c->n = n; c->count = 0; c->cap = 0; c->e = NULL; }
static void cache_free(mat_cache* c) {
    for (int i = 0; i < c->count; ++i) free(c->e[i].m);
    free(c->e);
    c->e = NULL; c->count = c->cap = 0;
}
static const double* cache_get(const mat_cache* c, double t, double lambda) {
    long lq, tq;
    quantize_key(lambda, t, &lq, &tq);
    for (int i = 0; i < c->count; ++i)
        if (c->e[i].lq == lq && c->e[i].tq == tq) return c->e[i].m;
    return NULL;                                  /* reference throws (matrix_cache.cpp:90-95) */
}
/* the full {lambdas} x {branch lengths} cross product, like precalculate_matrices */
static int cache_build(mat_cache* c, const double* lambdas, int nl, const double* ts, int nt, int fast) {
    int first_new = c->count;
    for (int i = 0; i < nl; ++i) for (int j = 0; j < nt; ++j) {
        if (cache_get(c, ts[j], lambdas[i])) continue;
        if (c->count == c->cap) {
            c->cap = c->cap ? 2 * c->cap : 64;
            c->e = (mat_entry*)realloc(c->e, sizeof(mat_entry) * c->cap);
        }
        mat_entry* e = &c->e[c->count++];
        quantize_key(lambdas[i], ts[j], &e->lq, &e->tq);
        e->m = (double*)malloc(sizeof(double) * (size_t)c->n * c->n);
        if (!e->m) return -1;
    }
    int n = c->n;
    int n_new = c->count - first_new;
    if (fast) {
#pragma omp parallel for schedule(dynamic, 1)
        for (int i = 0; i < n_new; ++i) {
            mat_entry* e = &c->e[first_new + i];
            build_matrix_conv_q(n, (double)e->lq / 1000000000.0, (double)e->tq / 1000.0, e->m);
        }
    } else {
        long total = (long)n_new * n;
#pragma omp parallel for schedule(dynamic, 4)
        for (long w = 0; w < total; ++w) {           /* omp collapse(2) over (key,row), matrix_cache.cpp:144 */
            mat_entry* e = &c->e[first_new + (int)(w / n)];
            int s = (int)(w % n);
            build_rows_direct(n, (double)e->lq / 1000000000.0, (double)e->tq / 1000.0, e->m, s, s + 1);
        }
    }
    return 0;
}

double orc_time_matrices(int n, double lambda, const double* ts, int count, int fast) {
    mat_cache c;
    cache_init(&c, n);
    struct timespec a, b;
    clock_gettime(CLOCK_MONOTONIC, &a);
    cache_build(&c, &lambda, 1, ts, count, fast);
    clock_gettime(CLOCK_MONOTONIC, &b);
    cache_free(&c);
    return (b.tv_sec - a.tv_sec) + 1e-9 * (b.tv_nsec - a.tv_nsec);
}

/* ---------------------------------------------------------------------------------
 * PAML discrete gamma, gamma.cpp:15-240.  Same arithmetic, expressed with loops.
 * --------------------------------------------------------------------------------- */
double orc_point_normal(double prob) {              /* gamma.cpp:203 (AS70) */
    const double a0 = -.322232431088, a1 = -1, a2 = -.342242088547, a3 = -.0204231210245;
    const double a4 = -.453642210148e-4, b0 = .0993484626060, b1 = .588581570495;
    const double b2 = .531103462366, b3 = .103537752850, b4 = .0038560700634;
    double p = prob;
    double p1 = (p < 0.5 ? p : 1 - p);
    if (p1 < 1e-20) return -9999;
    double y = sqrt(log(1 / (p1 * p1)));
    double z = y + ((((y * a4 + a3) * y + a2) * y + a1) * y + a0) / ((((y * b4 + b3) * y + b2) * y + b1) * y + b0);
    return p < 0.5 ? -z : z;
}

double orc_incomplete_gamma(double x, double alpha, double ln_gamma_alpha) {   /* gamma.cpp:66 (AS32) */
    const double accurate = 1e-8, overflow = 1e30;
    double p = alpha, g = ln_gamma_alpha;
    if (x == 0) return 0;
    if (x < 0 || p <= 0) return -1;
    double factor = exp(p * log(x) - x - g);
    if (!(x > 1 && x >= p)) {                        /* series expansion */
        double gin = 1, term = 1, rn = p;
        do {
            rn++;
            term *= x / rn;
            gin += term;
        } while (term > accurate);
        gin *= factor / p;
        return gin;
    }
    /* continued fraction */
    double pn[6];
    double a = 1 - p, b = a + x + 1, term = 0;
    pn[0] = 1; pn[1] = x; pn[2] = x + 1; pn[3] = x * b;
    double gin = pn[2] / pn[3];
    for (;;) {
        a++; b += 2; term++;
        double an = a * term;
        for (int i = 0; i < 2; i++) pn[i + 4] = b * pn[i + 2] - an * pn[i];
        if (pn[5] != 0) {
            double rn = pn[4] / pn[5];
            double dif = fabs(gin - rn);
            if (dif <= accurate && dif <= accurate * rn)
                return 1 - factor * gin;
            gin = rn;
        }
        for (int i = 0; i < 4; i++) pn[i] = pn[i + 2];
        if (fabs(pn[4]) >= overflow)
            for (int i = 0; i < 4; i++) pn[i] /= overflow;
    }
}

double orc_point_chi2(double prob, double v) {      /* gamma.cpp:129 (AS91) */
    const double e = .5e-6, aa = .6931471805;
    double p = prob, ch, q, p1, p2, t, a, b, x;
    if (p < .000002 || p > .999998 || v <= 0) return -1;
    double g = lgamma_int(v / 2);
    double xx = v / 2;
    double c = xx - 1;
    if (v < -1.24 * log(p)) {
        ch = pow((p * xx * exp(g + xx * aa)), 1 / xx);
        if (ch - e < 0) return ch;
    } else if (v > .32) {
        x = orc_point_normal(p);
        p1 = 0.222222 / v;
        ch = v * pow((x * sqrt(p1) + 1 - p1), 3.0);
        if (ch > 2.2 * v + 6) ch = -2 * (log(1 - p) - c * log(.5 * ch) + g);
    } else {
        ch = 0.4;
        a = log(1 - p);
        do {
            q = ch;
            p1 = 1 + ch * (4.67 + ch);
            p2 = ch * (6.73 + ch * (6.66 + ch));
            t = -0.5 + (4.67 + 2 * ch) / p1 - (6.73 + ch * (13.32 + 3 * ch)) / p2;
            ch -= (1 - exp(a + g + .5 * ch + c * aa) * p2 / p1) / t;
        } while (fabs(q / ch - 1) - .01 > 0);
    }
    do {
        q = ch;
        p1 = .5 * ch;
        t = orc_incomplete_gamma(p1, xx, g);
        if (t < 0) return -1;
        p2 = p - t;
        t = p2 * exp(xx * aa + g + p1 - c * log(ch));
        b = t / ch;
        a = 0.5 * t - b * c;
        double s1 = (210 + a * (140 + a * (105 + a * (84 + a * (70 + 60 * a))))) / 420;
        double s2 = (420 + a * (735 + a * (966 + a * (1141 + 1278 * a)))) / 2520;
        double s3 = (210 + a * (462 + a * (707 + 932 * a))) / 2520;
        double s4 = (252 + a * (672 + 1182 * a) + c * (294 + a * (889 + 1740 * a))) / 5040;
        double s5 = (84 + 264 * a + c * (175 + 606 * a)) / 2520;
        double s6 = (120 + c * (346 + 127 * c)) / 5040;
        ch += t * (1 + 0.5 * t * s1 - b * c * (s1 - b * (s2 - b * (s3 - b * (s4 - b * (s5 - b * s6))))));
    } while (fabs(q / ch - 1) > e);
    return ch;
}

/* gamma.cpp:15 with median=0 and beta=alpha (get_gamma, gamma.cpp:225-240) */
void orc_discrete_gamma(int K, double alpha, double* cat_probs, double* multipliers) {
    double beta = alpha;
    double factor = alpha / beta * K;
    double lnga1 = lgamma_int(alpha + 1);
    double* freq = cat_probs;
    for (int i = 0; i < K - 1; i++)
        freq[i] = orc_point_chi2((i + 1.0) / K, 2.0 * (alpha)) / (2.0 * (beta));   /* point_gamma macro, gamma.h:6 */
    for (int i = 0; i < K - 1; i++)
        freq[i] = orc_incomplete_gamma(freq[i] * beta, alpha + 1, lnga1);
    multipliers[0] = freq[0] * factor;
    multipliers[K - 1] = (1 - freq[K - 2]) * factor;
    for (int i = 1; i < K - 1; i++) multipliers[i] = (freq[i] - freq[i - 1]) * factor;
    for (int i = 0; i < K; i++) freq[i] = 1.0 / K;
}

/* ---------------------------------------------------------------------------------
 * root priors: root_equilibrium_distribution::compute returns a FLOAT
 * (root_equilibrium_distribution.h:15).
 * --------------------------------------------------------------------------------- */
void orc_prior_uniform(int R, float* out) {          /* root_distribution.cpp:25 + root_equilibrium_distribution.cpp:26 */
    int sum = 0;
    for (int i = 0; i < R; ++i) sum += 1;
    for (int i = 0; i < R; ++i) out[i] = (float)1 / (float)sum;
}
void orc_prior_poisson(int R, double pl, float* out) {   /* poisson.cpp:19-36, root_equilibrium_distribution.h:51-57 */
    for (int i = 0; i < R; ++i) {
        double v = exp(i * log(pl) - lgamma_int(i + 1) - pl);
        out[i] = (float)v;
    }
}
/* rootdist file given: root_distribution::vectorize (root_distribution.cpp:15) expands the
 * (size,count) map into a list; compute(j) = float(list[j]) / float(sum(list)), 0 past the end */
void orc_prior_rootdist(int R, const int32_t* sizes, const int32_t* counts, int n_entries, float* out) {
    long total = 0, len = 0;
    for (int i = 0; i < n_entries; ++i) { total += (long)sizes[i] * counts[i]; len += counts[i]; }
    int* list = (int*)malloc(sizeof(int) * (size_t)(len > 0 ? len : 1));
    long k = 0;
    for (int i = 0; i < n_entries; ++i) for (int j = 0; j < counts[i]; ++j) list[k++] = sizes[i];
    for (int j = 0; j < R; ++j) out[j] = j < len ? (float)list[j] / (float)(int)total : 0.0f;
    free(list);
}

/* ---------------------------------------------------------------------------------
 * prune: core.cpp:133-144 + probability.cpp:173-242
 * --------------------------------------------------------------------------------- */
static int root_of(const orc_tree* t) {
    for (int v = 0; v < t->n_nodes; ++v) if (t->parent[v] < 0) return v;
    return -1;
}

/* L holds n_nodes vectors of stride `stride` (>= M+1, >= R) */
static int prune_with_cache(const orc_problem* pb, const orc_params* pr, const mat_cache* cache,
                            const int32_t* counts_row, double mult, double* L, int stride, double* factor, double* root_out) {
    const orc_tree* tr = &pb->tree;
    int M = pb->max_family_size, R = pb->max_root_family_size, n = cache->n;
    int nd = pb->n_deviations;
    int root = root_of(tr);
    for (int v = 0; v < tr->n_nodes; ++v) {
        double* Lv = L + (size_t)v * stride;
        int is_root = tr->parent[v] < 0;
        int len = is_root ? R : M + 1;
        if (tr->leaf_taxon[v] >= 0) {                 /* probability.cpp:179-199 */
            for (int i = 0; i < len; ++i) Lv[i] = 0.0;
            int x = counts_row[tr->leaf_taxon[v]];
            if (nd > 0 && pr->error_model) {
                const double* probs = pr->error_model + (size_t)x * nd;
                int offset = x - ((nd - 1) / 2);
                for (int i = 0; i < nd; ++i) {
                    if (offset + i < 0) continue;
                    if (offset + i > M) continue;     /* out of range in the reference (UB); never hit by valid inputs */
                    Lv[offset + i] = probs[i];
                }
            } else {
                Lv[x] = 1.0;
            }
            continue;
        }
        for (int i = 0; i < len; ++i) Lv[i] = 1;      /* probability.cpp:211-218 / 233-240 */
        int s_min = is_root ? 1 : 0, s_max = is_root ? R : M;
        for (int u = 0; u < tr->n_nodes; ++u) {       /* children in ascending index = descendant order */
            if (tr->parent[u] != v) continue;
            double lam = pr->lambdas[tr->lambda_index ? tr->lambda_index[u] : 0] * mult;   /* lambda.h:39 / :82-88 */
            const double* mat = cache_get(cache, tr->branch_length[u], lam);
            if (!mat) return -1;
            orc_matvec(mat, n, L + (size_t)u * stride, s_min, s_max, 0, M, factor);
            for (int i = 0; i < len; ++i) Lv[i] *= factor[i];
        }
    }
    memcpy(root_out, L + (size_t)root * stride, sizeof(double) * R);
    return 0;
}

static int distinct_branch_lengths(const orc_tree* tr, double* ts) {   /* clade.cpp:196-205: the set of t > 0, root included */
    int nt = 0;
    for (int v = 0; v < tr->n_nodes; ++v) {
        double t = tr->branch_length[v];
        if (!(t > 0.0)) continue;
        int dup = 0;
        for (int j = 0; j < nt; ++j) if (ts[j] == t) { dup = 1; break; }
        if (!dup) ts[nt++] = t;
    }
    return nt;
}

static double g_t_matrices = 0.0, g_t_total = 0.0;
static double now_s(void) {
    struct timespec a;
    clock_gettime(CLOCK_MONOTONIC, &a);
    return a.tv_sec + 1e-9 * a.tv_nsec;
}
void orc_last_timings(double* matrices_s, double* total_s) { *matrices_s = g_t_matrices; *total_s = g_t_total; }

static int build_all_impl(const orc_problem* pb, const orc_params* pr, const double* mults, int K, mat_cache* cache);
static int build_all(const orc_problem* pb, const orc_params* pr, const double* mults, int K, mat_cache* cache) {
    double t0 = now_s();
    int rc = build_all_impl(pb, pr, mults, K, cache);
    g_t_matrices = now_s() - t0;
    return rc;
}
static int build_all_impl(const orc_problem* pb, const orc_params* pr, const double* mults, int K, mat_cache* cache) {
    const orc_tree* tr = &pb->tree;
    int M = pb->max_family_size, R = pb->max_root_family_size;
    cache_init(cache, (M > R ? M : R) + 1);            /* base_model.cpp:77 */
    double* ts = (double*)malloc(sizeof(double) * tr->n_nodes);
    int nt = distinct_branch_lengths(tr, ts);
    double* lams = (double*)malloc(sizeof(double) * (size_t)K * pb->n_lambdas);
    int nl = 0;
    for (int k = 0; k < K; ++k)                         /* gamma_core.cpp:111-121 */
        for (int i = 0; i < pb->n_lambdas; ++i) lams[nl++] = pr->lambdas[i] * mults[k];
    int rc = cache_build(cache, lams, nl, ts, nt, pr->fast_matrices);
    free(ts); free(lams);
    return rc;
}

int orc_prune(const orc_problem* pb, const orc_params* pr, int64_t family, double mult, double* root_out) {
    mat_cache cache;
    if (build_all(pb, pr, &mult, 1, &cache)) return -1;
    int M = pb->max_family_size, R = pb->max_root_family_size;
    int stride = (M + 1 > R ? M + 1 : R);
    double* L = (double*)malloc(sizeof(double) * (size_t)stride * (pb->tree.n_nodes + 1));
    int rc = prune_with_cache(pb, pr, &cache, pb->counts + family * pb->n_taxa, mult, L, stride,
                              L + (size_t)stride * pb->tree.n_nodes, root_out);
    free(L);
    cache_free(&cache);
    return rc;
}

static int lambdas_valid(const orc_problem* pb, const orc_params* pr) {
    if (pb->single_lambda) return pr->lambdas[0] > 0;              /* lambda.h:58 */
    for (int i = 0; i < pb->n_lambdas; ++i) if (pr->lambdas[i] < 0) return 0;   /* lambda.cpp:59 */
    return 1;
}

/* base_model.cpp:53-112.  Identical families are pruned once in the reference
 * (build_reference_list, base_model.cpp:27); the value per family is the same either way. */
double orc_score_base(const orc_problem* pb, const orc_params* pr, double* family_lnl) {
    double t_start = now_s();
    if (!lambdas_valid(pb, pr)) return -log(0.0);
    mat_cache cache;
    double one = 1.0;
    if (build_all(pb, pr, &one, 1, &cache)) return NAN;
    int M = pb->max_family_size, R = pb->max_root_family_size;
    int stride = (M + 1 > R ? M + 1 : R);
    int64_t F = pb->n_families;
    double* lnl = (double*)malloc(sizeof(double) * (size_t)(F > 0 ? F : 1));
    int failed = 0;
#pragma omp parallel
    {
        double* L = (double*)malloc(sizeof(double) * (size_t)stride * (pb->tree.n_nodes + 2));
        double* factor = L + (size_t)stride * pb->tree.n_nodes;
        double* rootv = factor + stride;
#pragma omp for schedule(dynamic, 4)
        for (int64_t f = 0; f < F; ++f) {
            if (prune_with_cache(pb, pr, &cache, pb->counts + f * pb->n_taxa, 1.0, L, stride, factor, rootv)) {
#pragma omp atomic write
                failed = 1;
                continue;
            }
            double best = 0;
            for (int j = 0; j < R; ++j) {                 /* base_model.cpp:94-101 */
                double eq_freq = pr->prior[j];
                double full = log(rootv[j]) + log(eq_freq);
                if (j == 0 || full > best) best = full;   /* std::max_element keeps the first maximum */
            }
            lnl[f] = best;
        }
        free(L);
    }
    double total = 0.0;
    for (int64_t f = 0; f < F; ++f) total += lnl[f];      /* std::accumulate, base_model.cpp:107 */
    if (family_lnl) memcpy(family_lnl, lnl, sizeof(double) * (size_t)F);
    free(lnl);
    cache_free(&cache);
    g_t_total = now_s() - t_start;
    if (failed) return NAN;
    return -total;
}

/* The p-value path's likelihood of a family: max_j L_root[j] with the plain lambda, no error model, no prior
 * (get_random_probabilities probability.cpp:309-313; compute_tree_pvalue probability.cpp:397-399). */
int orc_root_max(const orc_problem* pb, const orc_params* pr, double* out) {
    if (!lambdas_valid(pb, pr)) return 1;
    orc_params q = *pr;
    q.error_model = NULL;                               /* both call sites pass a NULL error model */
    q.n_categories = 1;
    mat_cache cache;
    double one = 1.0;
    if (build_all(pb, &q, &one, 1, &cache)) return 2;
    int M = pb->max_family_size, R = pb->max_root_family_size;
    int stride = (M + 1 > R ? M + 1 : R);
    int64_t F = pb->n_families;
    int failed = 0;
#pragma omp parallel
    {
        double* L = (double*)malloc(sizeof(double) * (size_t)stride * (pb->tree.n_nodes + 2));
        double* factor = L + (size_t)stride * pb->tree.n_nodes;
        double* rootv = factor + stride;
#pragma omp for schedule(dynamic, 4)
        for (int64_t f = 0; f < F; ++f) {
            if (prune_with_cache(pb, &q, &cache, pb->counts + f * pb->n_taxa, 1.0, L, stride, factor, rootv)) {
#pragma omp atomic write
                failed = 1;
                continue;
            }
            double best = rootv[0];                     /* std::max_element */
            for (int j = 1; j < R; ++j) if (rootv[j] > best) best = rootv[j];
            out[f] = best;
        }
        free(L);
    }
    cache_free(&cache);
    return failed ? 3 : 0;
}

/* probability.cpp:379-389: position of v in the sorted conditional distribution (std::upper_bound), as a fraction */
double orc_pvalue(double v, const double* conddist, int n) {
    int lo = 0, hi = n;                                 /* first element > v */
    while (lo < hi) {
        int mid = lo + (hi - lo) / 2;
        if (!(v < conddist[mid])) lo = mid + 1; else hi = mid;
    }
    int idx = (lo != n) ? lo : n - 1;
    return idx / (double)n;
}

/* probability.cpp:401-407 for every family: the maximum over root sizes s of pvalue(observed, cond[s]);
 * cond is [R][nsim], every row sorted ascending */
void orc_tree_pvalues(const double* observed, int64_t F, const double* cond, int R, int nsim, double* out) {
    for (int64_t f = 0; f < F; ++f) {
        double best = 0.0;
        for (int s = 0; s < R; ++s) {
            double p = orc_pvalue(observed[f], cond + (size_t)s * nsim, nsim);
            if (s == 0 || p > best) best = p;
        }
        out[f] = best;
    }
}

/* ---------------------------------------------------------------------------------
 * Pupko joint reconstruction: gene_family_reconstructor.cpp:13-165 (per family), called once per family by
 * base_model.cpp:145-160 and once per family and category (lambda * multiplier) by gamma_core.cpp:301-345.
 * root_prior[j] = root_equilibrium_distribution::compute(j), j = 0..min(M,R).
 * states[k][f][node]; leaves carry the observed count.
 * --------------------------------------------------------------------------------- */
static int reconstruct_one(const orc_problem* pb, const orc_params* pr, const mat_cache* cache, const int32_t* counts_row, double mult,
                           const float* root_prior, double* L, int* C, int32_t* state) {
    const orc_tree* tr = &pb->tree;
    int M = pb->max_family_size, R = pb->max_root_family_size, n = cache->n;
    int len = M + 1;
    int root = root_of(tr);
    for (int v = 0; v < tr->n_nodes; ++v) {                         /* apply_reverse_level_order: children first */
        double* Lv = L + (size_t)v * len;
        int* Cv = C + (size_t)v * len;
        if (tr->leaf_taxon[v] >= 0) {                               /* reconstruct_leaf_node :13-33 */
            int x = counts_row[tr->leaf_taxon[v]];
            double lam = pr->lambdas[tr->lambda_index ? tr->lambda_index[v] : 0] * mult;
            const double* mat = cache_get(cache, tr->branch_length[v], lam);
            if (!mat) return -1;
            Lv[0] = 0.0;                                            /* the loop starts at i = 1 */
            for (int i = 1; i < len; ++i) Lv[i] = mat[(size_t)i * n + x];
            for (int i = 0; i < len; ++i) Cv[i] = x;
            continue;
        }
        if (v == root) {                                            /* reconstruct_root_node :35-70 */
            int lr = (M < R ? M : R) + 1;
            double max_val = -1;
            Cv[0] = 0;
            for (int j = 1; j < lr; ++j) {                          /* the same scan for every i: done once */
                double value = 1.0;
                for (int u = 0; u < tr->n_nodes; ++u) if (tr->parent[u] == v) value *= L[(size_t)u * len + j];
                double val = value * root_prior[j];
                if (val > max_val) { max_val = val; Cv[0] = j; }
            }
            continue;
        }
        double lam = pr->lambdas[tr->lambda_index ? tr->lambda_index[v] : 0] * mult;
        const double* mat = cache_get(cache, tr->branch_length[v], lam);
        if (!mat) return -1;
        for (int i = 0; i < len; ++i) {                             /* reconstruct_internal_node :72-113 */
            int max_j = 0;
            double max_val = -1;
            for (int j = 0; j < len; ++j) {
                double value = 1.0;
                for (int u = 0; u < tr->n_nodes; ++u) if (tr->parent[u] == v) value *= L[(size_t)u * len + j];
                double val = value * mat[(size_t)i * n + j];
                if (val > max_val) { max_j = j; max_val = val; }
            }
            Lv[i] = max_val;
            Cv[i] = max_j;
        }
    }
    state[root] = C[(size_t)root * len];                            /* :161-163 and the backtracker :147-155 */
    for (int v = tr->n_nodes - 1; v >= 0; --v) {                    /* parents before children */
        if (v == root) continue;
        if (tr->leaf_taxon[v] >= 0) { state[v] = counts_row[tr->leaf_taxon[v]]; continue; }
        state[v] = C[(size_t)v * len + state[tr->parent[v]]];
    }
    return 0;
}

int orc_reconstruct(const orc_problem* pb, const orc_params* pr, const float* root_prior, int32_t* states) {
    if (!lambdas_valid(pb, pr)) return 1;
    int K = pr->n_categories;
    double one = 1.0;
    const double* mults = (K > 1 || pr->multipliers) ? pr->multipliers : &one;
    mat_cache cache;
    if (build_all(pb, pr, mults, K, &cache)) return 2;
    int len = pb->max_family_size + 1, nn = pb->tree.n_nodes;
    int64_t F = pb->n_families;
    int failed = 0;
#pragma omp parallel
    {
        double* L = (double*)malloc(sizeof(double) * (size_t)len * nn);
        int* C = (int*)malloc(sizeof(int) * (size_t)len * nn);
#pragma omp for schedule(dynamic, 4) collapse(2)
        for (int k = 0; k < K; ++k)
            for (int64_t f = 0; f < F; ++f)
                if (reconstruct_one(pb, pr, &cache, pb->counts + f * pb->n_taxa, mults[k], root_prior, L, C, states + ((size_t)k * F + f) * nn)) {
#pragma omp atomic write
                    failed = 1;
                }
        free(L); free(C);
    }
    cache_free(&cache);
    return failed ? 3 : 0;
}

/* compute_viterbi_sum, gene_family_reconstructor.cpp:361-400, for every family and node under the plain lambdas;
 * sizes[f][node]; NaN where the reference returns an invalid branch_probability */
int orc_branch_probabilities(const orc_problem* pb, const orc_params* pr, const int32_t* sizes, double* out) {
    if (!lambdas_valid(pb, pr)) return 1;
    const orc_tree* tr = &pb->tree;
    double one = 1.0;
    mat_cache cache;
    if (build_all(pb, pr, &one, 1, &cache)) return 2;
    int nn = tr->n_nodes, n = cache.n, M = pb->max_family_size;
    for (int64_t f = 0; f < pb->n_families; ++f)
        for (int v = 0; v < nn; ++v) {
            double res = NAN;
            if (tr->parent[v] >= 0) {
                int ps = sizes[f * nn + tr->parent[v]], cs = sizes[f * nn + v];
                if (ps != cs) {
                    double lam = pr->lambdas[tr->lambda_index ? tr->lambda_index[v] : 0];
                    const double* mat = cache_get(&cache, tr->branch_length[v], lam);
                    if (!mat) { cache_free(&cache); return 3; }
                    double calc = mat[(size_t)ps * n + cs];
                    res = 0;
                    for (int m = 0; m < M; m++) {
                        double pm = mat[(size_t)ps * n + m];
                        if (pm == calc) res += pm / 2.0;
                        else if (pm < calc) res += pm;
                    }
                }
            }
            out[f * nn + v] = res;
        }
    cache_free(&cache);
    return 0;
}

/* gamma_core.cpp:123-246 */
double orc_score_gamma(const orc_problem* pb, const orc_params* pr, double* cat_lik, double* fam_lik) {
    double t_start = now_s();
    int K = pr->n_categories;
    const orc_tree* tr = &pb->tree;
    /* can_infer, gamma_core.cpp:123-142 (alpha >= 0 is checked by the caller that owns alpha) */
    if (!lambdas_valid(pb, pr)) return -log(0.0);
    {
        double longest = 0, largest_mult = pr->multipliers[0], largest_lambda = pr->lambdas[0];
        int first = 1;
        for (int v = 0; v < tr->n_nodes; ++v) {
            double t = tr->branch_length[v];
            if (!(t > 0.0)) continue;
            if (first || t > longest) { longest = t; first = 0; }
        }
        for (int k = 1; k < K; ++k) if (pr->multipliers[k] > largest_mult) largest_mult = pr->multipliers[k];
        for (int i = 1; i < pb->n_lambdas; ++i) if (pr->lambdas[i] > largest_lambda) largest_lambda = pr->lambdas[i];
        if (orc_is_saturated(longest, largest_mult * largest_lambda)) return -log(0.0);
    }
    mat_cache cache;
    if (build_all(pb, pr, pr->multipliers, K, &cache)) return NAN;
    int M = pb->max_family_size, R = pb->max_root_family_size;
    int stride = (M + 1 > R ? M + 1 : R);
    int64_t F = pb->n_families;
    double* loglik = (double*)malloc(sizeof(double) * (size_t)(F > 0 ? F : 1));
    int* failure = (int*)calloc((size_t)(F > 0 ? F : 1), sizeof(int));
    int broken = 0;
#pragma omp parallel
    {
        double* L = (double*)malloc(sizeof(double) * (size_t)stride * (pb->tree.n_nodes + 2));
        double* factor = L + (size_t)stride * pb->tree.n_nodes;
        double* rootv = factor + stride;
        double* cl = (double*)malloc(sizeof(double) * K);
#pragma omp for schedule(dynamic, 2)
        for (int64_t f = 0; f < F; ++f) {
            int ok = 1;
            for (int k = 0; k < K && ok; ++k) {           /* gamma_core.cpp:144-166 */
                if (prune_with_cache(pb, pr, &cache, pb->counts + f * pb->n_taxa, pr->multipliers[k], L, stride, factor, rootv)) {
#pragma omp atomic write
                    broken = 1;
                    ok = 0; break;
                }
                double sum = 0.0;
                for (int j = 0; j < R; ++j) sum += rootv[j];
                if (sum == 0.0) { ok = 0; break; }        /* "saturation" */
                double best = 0;
                for (int j = 0; j < R; ++j) {
                    double eq_freq = pr->prior[j];
                    double full = rootv[j] * eq_freq;
                    if (j == 0 || full > best) best = full;
                }
                cl[k] = best * pr->cat_probs[k];
            }
            if (!ok) { failure[f] = 1; continue; }
            double fl = 0.0;
            for (int k = 0; k < K; ++k) fl += cl[k];       /* gamma_core.cpp:207 */
            if (cat_lik) for (int k = 0; k < K; ++k) cat_lik[f * K + k] = cl[k];
            if (fam_lik) fam_lik[f] = fl;
            loglik[f] = log(fl);
        }
        free(L); free(cl);
    }
    int any_fail = 0;
    for (int64_t f = 0; f < F; ++f) any_fail |= failure[f];
    double total = 0.0;
    for (int64_t f = 0; f < F; ++f) total += loglik[f];
    free(loglik); free(failure);
    cache_free(&cache);
    g_t_total = now_s() - t_start;
    if (broken) return NAN;
    if (any_fail) return -log(0.0);                       /* gamma_core.cpp:227-236 */
    return -total;
}
