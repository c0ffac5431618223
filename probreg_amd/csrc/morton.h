// Morton (Z-curve) order of a point cloud on the host: result[i] = original index of the i-th point along the curve.
// One-off work at upload (a radix sort of n 64-bit keys: ~1 ms per 100k points); the kernels only ever see the
// sorted arrays, so spatially close points share wavefronts / workgroups / hash-table neighbourhoods.
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>
#include <numeric>
#include <string>
#include <thread>
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

// LSD radix sort of 64-bit keys, 11 bits per pass, only over the bits that are in use
inline void radix_sort_u64(std::vector<uint64_t>& key, int used_bits) {
    std::vector<uint64_t> tmp(key.size());
    uint64_t* a = key.data();
    uint64_t* b = tmp.data();
    const size_t n = key.size();
    for (int shift = 0; shift < used_bits; shift += 11) {
        size_t count[2049] = {0};
        for (size_t i = 0; i < n; ++i) ++count[((a[i] >> shift) & 2047) + 1];
        for (int d = 0; d < 2048; ++d) count[d + 1] += count[d];
        for (size_t i = 0; i < n; ++i) b[count[(a[i] >> shift) & 2047]++] = a[i];
        std::swap(a, b);
    }
    if (a != key.data()) key.swap(tmp);
}

// The Morton code and the point's index share one 64-bit key (code in the high bits, as many bits per axis as the
// index leaves room for, at most 21), so the sort moves plain integers and ties keep the input order.
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
    int idx_bits = 1;
    while (((int64_t)1 << idx_bits) < n) ++idx_bits;
    const int axis_bits = std::min(21, (64 - idx_bits) / 3);  // spread21 interleaves for three axes whatever dim is
    const double scale = ext > 0.0 ? (double)(((uint64_t)1 << axis_bits) - 1) / ext : 0.0;  // one isotropic grid
    std::vector<uint64_t> key((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        uint64_t code = 0;
        for (int k = 0; k < dim; ++k) code |= spread21((uint64_t)(((double)pts[i * dim + k] - lo[k]) * scale)) << k;
        key[i] = code << idx_bits | (uint64_t)i;
    }
    radix_sort_u64(key, idx_bits + 3 * axis_bits);
    std::vector<int> perm((size_t)n);
    const uint64_t idx_mask = ((uint64_t)1 << idx_bits) - 1;
    for (int64_t i = 0; i < n; ++i) perm[i] = (int)(key[i] & idx_mask);
    return perm;
}

// kd-tree order: result[i] = original index of the i-th point of an in-order walk of a LEFT-ALIGNED kd-tree whose leaves hold
// `leaf` points.  A node is cut across the widest axis of its bounding box; its left child takes the largest power of two
// of leaves that leaves the right child non-empty, so every aligned run of 2^k leaves - the 32-point groups, the 128-point
// blocks a wave owns, the 256-point chunks and 512-point blocks of the matrix-core sweeps - IS a subtree: one axis-aligned
// cell of the cloud, never two pieces either side of a jump of a space-filling curve.  Only the last leaf may be partial.
// What it buys over the Z-curve (CPU count on C1's clouds at EM iteration 19, 128 x 32 box tests at the 2^-48 bound): 2.92e8
// evaluated pairs instead of 3.67e8; at the noise floor 1.40e8 instead of 1.93e8 (Hilbert: 3.10e8 / 1.53e8).
// One-off host work at upload: O(n log n) (std::nth_element per node; the top subtrees on threads of their own).
template <typename T>
struct KdPoint { T c[3]; int i; };  // the coordinates travel with the index: partitioning reads memory in order, not through it

template <typename T>
void kd_order_rec(KdPoint<T>* p, int dim, int64_t n, int leaf, int spawn_depth) {
    std::vector<std::thread> helpers;  // left subtrees of the top levels, each on a thread of its own (joined below)
    while (n > leaf) {
        T lo[3], hi[3];
        for (int k = 0; k < dim; ++k) lo[k] = hi[k] = p[0].c[k];
        for (int64_t i = 1; i < n; ++i)
            for (int k = 0; k < dim; ++k) {
                lo[k] = std::min(lo[k], p[i].c[k]);
                hi[k] = std::max(hi[k], p[i].c[k]);
            }
        int ax = 0;
        for (int k = 1; k < dim; ++k)
            if (hi[k] - lo[k] > hi[ax] - lo[ax]) ax = k;
        const int64_t leaves = (n + leaf - 1) / leaf;
        int64_t left = 1;
        while (2 * left < leaves) left *= 2;  // largest power of two < leaves
        const int64_t cut = left * leaf;
        // (ties broken by the index: the order is a function of the cloud alone)
        std::nth_element(p, p + cut, p + n, [ax](const KdPoint<T>& a, const KdPoint<T>& b) {
            return a.c[ax] < b.c[ax] || (a.c[ax] == b.c[ax] && a.i < b.i);
        });
        // left: a full power-of-two subtree (recursion depth <= log2 n); the big ones near the root run concurrently with the rest
        if (spawn_depth > 0 && cut >= 4096) {
            KdPoint<T>* sub = p;
            helpers.emplace_back([sub, dim, cut, leaf, spawn_depth]() { kd_order_rec(sub, dim, cut, leaf, spawn_depth - 1); });
        } else {
            kd_order_rec(p, dim, cut, leaf, 0);
        }
        if (spawn_depth > 0) --spawn_depth;
        p += cut;  // right: iterate
        n -= cut;
    }
    for (std::thread& t : helpers) t.join();
}

template <typename T>
std::vector<int> kd_order(const T* pts, int64_t n, int dim, int leaf = 32) {
    std::vector<int> perm((size_t)n);
    if (n <= leaf) {
        std::iota(perm.begin(), perm.end(), 0);
        return perm;
    }
    std::vector<KdPoint<T>> p((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        for (int k = 0; k < 3; ++k) p[(size_t)i].c[k] = k < dim ? pts[i * dim + k] : (T)0;
        p[(size_t)i].i = (int)i;
    }
    // (part of every registration through the public API, tools/time_registration.py: the subtrees of the top four levels run on
    // threads of their own)
    kd_order_rec(p.data(), dim, n, leaf, n >= 16384 ? 4 : 0);
    for (int64_t i = 0; i < n; ++i) perm[(size_t)i] = p[(size_t)i].i;
    return perm;
}

}  // namespace prg
