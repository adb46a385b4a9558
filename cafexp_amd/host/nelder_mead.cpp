// Nelder-Mead driver for the scorers.  Same moves and constants as the reference's
// fminsearch (src/optimizer.cpp:60-320: rho 1, chi 2, psi 0.5, sigma 0.5, delta 0.05, zero_delta
// 0.00025) and its default stop rule (NelderMeadSimilarityCutoff, optimizer.cpp:391-419: tolx/tolf
// 1e-6, or the best score moving < 1e-3 over 12 iterations).  Initial guesses are RNG-driven, so
// trajectories differ from the reference; optima are compared, not paths.
#include "cafe_host.h"

#include <algorithm>
#include <cmath>
#include <deque>

namespace cafe {

std::vector<double> optimizer::get_initial_guesses(int& calls) {
    std::vector<double> initial = _scorer->initial_guesses();
    double first = _scorer->calculate_score(initial.data());
    ++calls;
    for (int i = 0; std::isinf(first) && i < 100; ++i) {          // NUM_OPTIMIZER_INITIALIZATION_ATTEMPTS
        initial = _scorer->initial_guesses();
        first = _scorer->calculate_score(initial.data());
        ++calls;
    }
    if (std::isinf(first)) throw std::runtime_error("Failed to initialize any reasonable values");
    return initial;
}

optimizer_result optimizer::optimize() {
    const double rho = 1, chi = 2, psi = 0.5, sigma = 0.5, delta = 0.05, zero_delta = 0.00025;
    optimizer_result res;
    std::vector<double> x0 = get_initial_guesses(res.num_scorer_calls);
    const int n = (int)x0.size();
    struct vertex { std::vector<double> x; double f; };
    std::vector<vertex> simplex(n + 1);
    auto eval = [&](const std::vector<double>& x) { ++res.num_scorer_calls; return _scorer->calculate_score(x.data()); };
    auto by_score = [](const vertex& a, const vertex& b) { return a.f < b.f; };

    for (int i = 0; i <= n; ++i) {                                   // __fminsearch_min_init
        simplex[i].x = x0;
        if (i > 0) {
            const int j = i - 1;
            const bool widen = i > 1 && std::isinf(simplex[i - 1].f);
            simplex[i].x[j] = x0[j] ? (1 + (widen ? delta * 100 : delta)) * x0[j] : zero_delta;
        }
        simplex[i].f = eval(simplex[i].x);
    }
    std::sort(simplex.begin(), simplex.end(), by_score);

    std::deque<double> recent;
    std::vector<double> mean(n), xr(n), xt(n);
    int it = 0;
    for (; it < max_iterations; ++it) {
        double dx = 0, df = 0;                                       // threshold_achieved: checkV && checkF
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) dx = std::max(dx, std::fabs(simplex[i + 1].x[j] - simplex[i].x[j]));
        for (int i = 1; i <= n; ++i) df = std::max(df, std::fabs(simplex[i].f - simplex[0].f));
        if (dx <= tolx && df <= tolf) break;
        if (similarity_window > 0) {
            recent.push_back(simplex[0].f);
            if ((int)recent.size() > similarity_window) recent.pop_front();
            if ((int)recent.size() == similarity_window) {
                const auto mm = std::minmax_element(recent.begin(), recent.end());
                if (*mm.second - *mm.first < similarity_precision) break;
            }
        }
        for (int j = 0; j < n; ++j) {
            mean[j] = 0;
            for (int i = 0; i < n; ++i) mean[j] += simplex[i].x[j];
            mean[j] /= n;
        }
        vertex& worst = simplex[n];
        for (int j = 0; j < n; ++j) xr[j] = mean[j] + rho * (mean[j] - worst.x[j]);
        const double fr = eval(xr);
        bool shrink = false;
        if (fr < simplex[0].f) {
            for (int j = 0; j < n; ++j) xt[j] = mean[j] + chi * (xr[j] - mean[j]);
            const double fe = eval(xt);
            if (fe < fr) { worst.x = xt; worst.f = fe; } else { worst.x = xr; worst.f = fr; }
        } else if (fr >= worst.f) {
            if (fr > worst.f) {
                for (int j = 0; j < n; ++j) xt[j] = mean[j] + psi * (mean[j] - worst.x[j]);     // contract inside
                const double fc = eval(xt);
                if (fc < worst.f) { worst.x = xt; worst.f = fc; } else shrink = true;
            } else {
                for (int j = 0; j < n; ++j) xt[j] = mean[j] + psi * (xr[j] - mean[j]);          // contract outside
                const double fc = eval(xt);
                if (fc <= fr) { worst.x = xt; worst.f = fc; } else shrink = true;
            }
        } else {
            worst.x = xr; worst.f = fr;
        }
        if (shrink)
            for (int i = 1; i <= n; ++i) {
                for (int j = 0; j < n; ++j) simplex[i].x[j] = simplex[0].x[j] + sigma * (simplex[i].x[j] - simplex[0].x[j]);
                simplex[i].f = eval(simplex[i].x);
            }
        std::sort(simplex.begin(), simplex.end(), by_score);
    }
    res.values = simplex[0].x;
    res.score = simplex[0].f;
    res.num_iterations = it;
    return res;
}

}  // namespace cafe
