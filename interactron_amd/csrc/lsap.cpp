// Rectangular linear-sum assignment on the host: shortest augmenting path with dual updates
// (D. F. Crouse, "On implementing 2D rectangular assignment algorithms", IEEE TAES 52(4), 2016) -- the published
// algorithm behind scipy.optimize.linear_sum_assignment, which the reference calls once per image
// (models/detr_models/matcher.py:76; scipy pinned 1.8.0 in requirements.txt:8, not vendored).  Tie-breaking
// follows the same scan order (columns visited in reverse index order, a free column preferred among equal
// path costs), so assignments agree with scipy on cost matrices with exact ties (duplicated ground-truth boxes).
#include <math.h>
#include <stdint.h>

#include <algorithm>
#include <numeric>
#include <vector>

#include "common.h"

static int64_t augment(int64_t nc, const double* cost, std::vector<double>& u, std::vector<double>& v,
                       std::vector<int64_t>& path, std::vector<int64_t>& row4col, std::vector<double>& shortest,
                       int64_t i, std::vector<char>& SR, std::vector<char>& SC, std::vector<int64_t>& remaining,
                       double* p_min) {
    double min_val = 0;
    int64_t num_remaining = nc;
    for (int64_t it = 0; it < nc; ++it) remaining[it] = nc - it - 1;
    std::fill(SR.begin(), SR.end(), 0);
    std::fill(SC.begin(), SC.end(), 0);
    std::fill(shortest.begin(), shortest.end(), INFINITY);
    int64_t sink = -1;
    while (sink == -1) {
        int64_t index = -1;
        double lowest = INFINITY;
        SR[i] = 1;
        for (int64_t it = 0; it < num_remaining; ++it) {
            const int64_t j = remaining[it];
            const double r = min_val + cost[i * nc + j] - u[i] - v[j];
            if (r < shortest[j]) {
                path[j] = i;
                shortest[j] = r;
            }
            if (shortest[j] < lowest || (shortest[j] == lowest && row4col[j] == -1)) {
                lowest = shortest[j];
                index = it;
            }
        }
        min_val = lowest;
        if (min_val == INFINITY) return -1;
        const int64_t j = remaining[index];
        if (row4col[j] == -1)
            sink = j;
        else
            i = row4col[j];
        SC[j] = 1;
        remaining[index] = remaining[--num_remaining];
    }
    *p_min = min_val;
    return sink;
}

// cost: row-major [nr, nc] (float32 as produced by the cost kernel; promoted to double like scipy does).
// Writes k = min(nr, nc) pairs (row_idx[k], col_idx[k]) sorted by row.  Returns 0, -1 on bad input, -4 infeasible.
extern "C" int ix_lsap_f32(const float* cost_f32, int64_t nr, int64_t nc, int64_t* row_idx, int64_t* col_idx) {
    IX_CHECK_ARG(nr >= 0 && nc >= 0, "ix_lsap_f32: negative dims");
    if (nr == 0 || nc == 0) return IX_OK;
    IX_CHECK_ARG(cost_f32 && row_idx && col_idx, "ix_lsap_f32: null pointer");
    const bool transpose = nc < nr;
    std::vector<double> cost((size_t)(nr * nc));
    if (transpose) {
        for (int64_t i = 0; i < nr; ++i)
            for (int64_t j = 0; j < nc; ++j) cost[j * nr + i] = (double)cost_f32[i * nc + j];
        std::swap(nr, nc);
    } else {
        for (int64_t k = 0; k < nr * nc; ++k) cost[k] = (double)cost_f32[k];
    }
    for (double c : cost) {
        if (c != c || c == -INFINITY) {
            ix_set_error("ix_lsap_f32: cost matrix contains NaN or -inf");
            return IX_ERR_ARG;
        }
    }
    std::vector<double> u(nr, 0), v(nc, 0), shortest(nc);
    std::vector<int64_t> path(nc, -1), col4row(nr, -1), row4col(nc, -1), remaining(nc);
    std::vector<char> SR(nr), SC(nc);
    for (int64_t cur = 0; cur < nr; ++cur) {
        double min_val;
        const int64_t sink = augment(nc, cost.data(), u, v, path, row4col, shortest, cur, SR, SC, remaining, &min_val);
        if (sink < 0) {
            ix_set_error("ix_lsap_f32: cost matrix is infeasible");
            return -4;
        }
        u[cur] += min_val;
        for (int64_t i = 0; i < nr; ++i)
            if (SR[i] && i != cur) u[i] += min_val - shortest[col4row[i]];
        for (int64_t j = 0; j < nc; ++j)
            if (SC[j]) v[j] -= min_val - shortest[j];
        int64_t j = sink;
        while (true) {
            const int64_t i = path[j];
            row4col[j] = i;
            std::swap(col4row[i], j);
            if (i == cur) break;
        }
    }
    if (transpose) {
        std::vector<int64_t> order(nr);
        std::iota(order.begin(), order.end(), 0);
        std::sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return col4row[a] < col4row[b]; });
        for (int64_t k = 0; k < nr; ++k) {
            row_idx[k] = col4row[order[k]];
            col_idx[k] = order[k];
        }
    } else {
        for (int64_t i = 0; i < nr; ++i) {
            row_idx[i] = i;
            col_idx[i] = col4row[i];
        }
    }
    return IX_OK;
}
