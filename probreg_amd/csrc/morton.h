// Morton (Z-curve) order of a point cloud on the host: result[i] = original index of the i-th point along the curve.
// One-off work at upload (std::stable_sort over n 64-bit keys: ~10 ms per 100k points); the kernels only ever see the
// sorted arrays, so spatially close points share wavefronts / workgroups / hash-table neighbourhoods.
#pragma once
#include <math.h>
#include <stdint.h>

#include <algorithm>
#include <numeric>
#include <vector>

namespace prg {

inline uint64_t spread21(uint64_t v) {  // 21 bits -> every third bit
    v &= 0x1fffff;
    v = (v | v << 32) & 0x1f00000000ffffull;
    v = (v | v << 16) & 0x1f0000ff0000ffull;
    v = (v | v << 8) & 0x100f00f00f00f00full;
    v = (v | v << 4) & 0x10c30c30c30c30c3ull;
    v = (v | v << 2) & 0x1249249249249249ull;
    return v;
}

template <typename T>
std::vector<int> morton_order(const T* pts, int64_t n, int dim) {
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t i = 0; i < n; ++i)
        for (int k = 0; k < dim; ++k) {
            lo[k] = std::min(lo[k], (double)pts[i * dim + k]);
            hi[k] = std::max(hi[k], (double)pts[i * dim + k]);
        }
    double ext = 0.0;
    for (int k = 0; k < dim; ++k) ext = std::max(ext, hi[k] - lo[k]);
    const double scale = ext > 0.0 ? 2097151.0 / ext : 0.0;  // one isotropic 21-bit grid
    std::vector<uint64_t> key((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        uint64_t code = 0;
        for (int k = 0; k < dim; ++k) code |= spread21((uint64_t)(((double)pts[i * dim + k] - lo[k]) * scale)) << k;
        key[i] = code;
    }
    std::vector<int> perm((size_t)n);
    std::iota(perm.begin(), perm.end(), 0);
    std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) { return key[a] < key[b]; });
    return perm;
}

}  // namespace prg
