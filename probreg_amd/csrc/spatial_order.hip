// The kd-tree order of a cloud (morton.h: kd_order) built ON THE DEVICE, MI355X gfx950.
//
// Every CPD plan stores its clouds in the order of a left-aligned kd-tree: a node is cut across the widest axis
// of its points' box, its left child takes the largest power of two of 32-point leaves.  On the host that is one std::nth_element
// per node - 17 ms per 100k points, part of EVERY registration through the public API (two clouds: +25 ms on a 40 ms registration).
// Here the tree is built level by level with the whole level in flight:
//   * the node sizes follow from n alone, so the host knows every level's segments without asking the device;
//   * per level: the boxes of all nodes (wave-reduced, then atomics on order-preserving integer images of the floats), one 64-bit
//     key per point - (node << 32) | image of its coordinate along the node's widest axis - and ONE stable radix sort of all
//     (key, index) pairs (rocPRIM, a plain library primitive): every node is sorted along its own axis, which is also a partition
//     at any rank; finished nodes (<= 32 points) keep their positions;
//   * ~13 levels for 100k points, ~10 launches each: about 2 ms, and the permutation never leaves the device.
// The result has the same cells as the host build (same boxes, same axes, same ranks; points with EQUAL coordinates at a cut may
// land on the other side).  prg_spatial_order returns either for tests.
#include <rocprim/device/device_radix_sort.hpp>

#include <string.h>

#include <algorithm>
#include <vector>

#include "cpd_plan.h"
#include "morton.h"

namespace {

constexpr int kBlock = 256;

// order-preserving image of a float in the unsigned integers
__device__ __forceinline__ unsigned f2o(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float o2f(unsigned o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o); }

// the segment (node) position i lies in: the last s with start[s] <= i
__device__ __forceinline__ int find_seg(const int* __restrict__ start, int nseg, int i) {
    int lo = 0, hi = nseg - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (start[mid] <= i) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__global__ __launch_bounds__(kBlock) void k_iota(int* __restrict__ perm, int n) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) perm[i] = i;
}

// bbox[seg][6] = min x, y, z, max x, y, z as order-preserving integers (initialised to ~0 / 0)
__global__ __launch_bounds__(kBlock) void k_kd_bbox(const float* __restrict__ pts, int dim, const int* __restrict__ perm, int n,
                                                    const int* __restrict__ start, int nseg, unsigned* __restrict__ bbox) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    const bool in = i < n;
    const int seg = find_seg(start, nseg, in ? i : n - 1);
    unsigned v[3] = {0u, 0u, 0u};
    if (in) {
        const int64_t j = perm[i];
        for (int k = 0; k < dim; ++k) v[k] = f2o(pts[j * dim + k]);
    }
    // a wave that lies in ONE segment (all of them near the root) reduces first, the waves of a workgroup then combine in LDS:
    // six atomics per workgroup while the segments are long (every same-address atomic is served on its own: 1563 waves x 6 on the
    // root's box cost 110 us)
    __shared__ unsigned sh[kBlock / 64][6];
    __shared__ int shseg[kBlock / 64];
    const int seg0 = __shfl(seg, 0, 64);
    const bool uniform = __all(seg == seg0 && in);
    if (uniform) {
        unsigned lo[3], hi[3];
        for (int k = 0; k < 3; ++k) lo[k] = hi[k] = v[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
            for (int k = 0; k < 3; ++k) {
                lo[k] = min(lo[k], (unsigned)__shfl_xor((int)lo[k], off, 64));
                hi[k] = max(hi[k], (unsigned)__shfl_xor((int)hi[k], off, 64));
            }
        if ((threadIdx.x & 63) == 0)
            for (int k = 0; k < 3; ++k) {
                sh[threadIdx.x >> 6][k] = lo[k];
                sh[threadIdx.x >> 6][3 + k] = hi[k];
            }
    } else if (in) {
        for (int k = 0; k < dim; ++k) {
            atomicMin(&bbox[seg * 6 + k], v[k]);
            atomicMax(&bbox[seg * 6 + 3 + k], v[k]);
        }
    }
    if ((threadIdx.x & 63) == 0) shseg[threadIdx.x >> 6] = uniform ? seg : -1;
    __syncthreads();
    if (threadIdx.x < kBlock / 64) {  // one lane per wave: merge into the first wave of the same segment, that one does the atomics
        const int w = threadIdx.x, sg = shseg[w];
        if (sg >= 0) {
            int first = w;
            for (int q = 0; q < w; ++q)
                if (shseg[q] == sg) { first = q; break; }
            if (first == w) {
                unsigned lo[3], hi[3];
                for (int k = 0; k < 3; ++k) {
                    lo[k] = sh[w][k];
                    hi[k] = sh[w][3 + k];
                }
                for (int q = w + 1; q < kBlock / 64; ++q)
                    if (shseg[q] == sg)
                        for (int k = 0; k < 3; ++k) {
                            lo[k] = min(lo[k], sh[q][k]);
                            hi[k] = max(hi[k], sh[q][3 + k]);
                        }
                for (int k = 0; k < dim; ++k) {
                    atomicMin(&bbox[sg * 6 + k], lo[k]);
                    atomicMax(&bbox[sg * 6 + 3 + k], hi[k]);
                }
            }
        }
    }
}

// key = (segment << 32) | image of the coordinate along the segment's widest axis; a finished segment keeps its order
__global__ __launch_bounds__(kBlock) void k_kd_keys(const float* __restrict__ pts, int dim, const int* __restrict__ perm, int n,
                                                    const int* __restrict__ start, const unsigned char* __restrict__ split, int nseg,
                                                    const unsigned* __restrict__ bbox, unsigned long long* __restrict__ keys) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int seg = find_seg(start, nseg, i);
    unsigned low;
    if (split[seg]) {
        int ax = 0;
        float best = o2f(bbox[seg * 6 + 3]) - o2f(bbox[seg * 6]);
        for (int k = 1; k < dim; ++k) {
            const float e = o2f(bbox[seg * 6 + 3 + k]) - o2f(bbox[seg * 6 + k]);
            if (e > best) {  // (the first widest axis, as the host build)
                best = e;
                ax = k;
            }
        }
        low = f2o(pts[(int64_t)perm[i] * dim + ax]);
    } else {
        low = (unsigned)(i - start[seg]);
    }
    keys[i] = ((unsigned long long)(unsigned)seg << 32) | low;
}

}  // namespace

namespace prg {

// pts_dev: [n][dim] floats in the caller's order (device memory); perm_dev: [n] ints, sorted position -> original index
int device_kd_order(const float* pts_dev, int64_t n64, int dim, int* perm_dev, hipStream_t st, int leaf) {
    PRG_REQUIRE(n64 > 0 && n64 < ((int64_t)1 << 31), PRG_ERR_INVALID, "spatial order: cloud of %lld points", (long long)n64);
    const int n = (int)n64;
    const unsigned grid = (unsigned)ceil_div((int64_t)n, kBlock);
    if (n <= leaf) {
        k_iota<<<grid, kBlock, 0, st>>>(perm_dev, n);
        PRG_HIP(hipGetLastError());
        return PRG_OK;
    }
    // the levels: segments (start, size) in position order; a segment of more than `leaf` points is cut where kd_order_rec cuts it
    std::vector<std::vector<int>> starts;
    std::vector<std::vector<unsigned char>> splits;
    {
        std::vector<std::pair<int, int>> segs = {{0, n}};
        for (;;) {
            std::vector<int> s;
            std::vector<unsigned char> f;
            bool any = false;
            for (const auto& sg : segs) {
                s.push_back(sg.first);
                f.push_back(sg.second > leaf ? 1 : 0);
                any |= sg.second > leaf;
            }
            if (!any) break;
            starts.push_back(s);
            splits.push_back(f);
            std::vector<std::pair<int, int>> next;
            for (const auto& sg : segs) {
                if (sg.second <= leaf) {
                    next.push_back(sg);
                    continue;
                }
                const int leaves = (sg.second + leaf - 1) / leaf;
                int left = 1;
                while (2 * left < leaves) left *= 2;
                next.push_back({sg.first, left * leaf});
                next.push_back({sg.first + left * leaf, sg.second - left * leaf});
            }
            segs.swap(next);
        }
    }
    size_t max_seg = 0;
    for (const auto& s : starts) max_seg = std::max(max_seg, s.size());
    // one scratch block: keys x 2, perm x 2, segment table, boxes, the sort's workspace
    size_t sort_bytes = 0;
    PRG_HIP(rocprim::radix_sort_pairs(nullptr, sort_bytes, (const unsigned long long*)nullptr, (unsigned long long*)nullptr,
                                      (const int*)nullptr, (int*)nullptr, (unsigned)n, 0u, 64u, st));
    const size_t off_keys = 0, off_keys2 = off_keys + round_up((int64_t)n * 8, 256), off_perm2 = off_keys2 + round_up((int64_t)n * 8, 256),
                 off_start = off_perm2 + round_up((int64_t)n * 4, 256), off_split = off_start + round_up((int64_t)max_seg * 4, 256),
                 off_bbox = off_split + round_up((int64_t)max_seg, 256), off_sort = off_bbox + round_up((int64_t)max_seg * 24, 256);
    char* scratch = nullptr;
    PRG_HIP(hipMalloc((void**)&scratch, off_sort + sort_bytes + 256));
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(scratch + off_keys);
    unsigned long long* keys2 = reinterpret_cast<unsigned long long*>(scratch + off_keys2);
    int* perm_a = perm_dev;
    int* perm_b = reinterpret_cast<int*>(scratch + off_perm2);
    int* d_start = reinterpret_cast<int*>(scratch + off_start);
    unsigned char* d_split = reinterpret_cast<unsigned char*>(scratch + off_split);
    unsigned* d_bbox = reinterpret_cast<unsigned*>(scratch + off_bbox);
    int st_code = PRG_OK;
    auto check = [&](hipError_t e) {
        if (e != hipSuccess && st_code == PRG_OK) {
            set_error("spatial order: %s", hipGetErrorString(e));
            st_code = PRG_ERR_HIP;
        }
    };
    k_iota<<<grid, kBlock, 0, st>>>(perm_a, n);
    // (every host array an asynchronous copy reads from lives until the synchronisation at the end)
    std::vector<unsigned> box_init(max_seg * 6, 0u);
    for (size_t s = 0; s < max_seg; ++s)
        for (int k = 0; k < 3; ++k) box_init[s * 6 + k] = 0xFFFFFFFFu;
    for (size_t lv = 0; lv < starts.size() && st_code == PRG_OK; ++lv) {
        const int nseg = (int)starts[lv].size();
        check(hipMemcpyAsync(d_start, starts[lv].data(), (size_t)nseg * sizeof(int), hipMemcpyHostToDevice, st));
        check(hipMemcpyAsync(d_split, splits[lv].data(), (size_t)nseg, hipMemcpyHostToDevice, st));
        check(hipMemcpyAsync(d_bbox, box_init.data(), (size_t)nseg * 6 * sizeof(unsigned), hipMemcpyHostToDevice, st));
        k_kd_bbox<<<grid, kBlock, 0, st>>>(pts_dev, dim, perm_a, n, d_start, nseg, d_bbox);
        k_kd_keys<<<grid, kBlock, 0, st>>>(pts_dev, dim, perm_a, n, d_start, d_split, nseg, d_bbox, keys);
        unsigned seg_bits = 1;
        while ((1u << seg_bits) < (unsigned)nseg) ++seg_bits;
        size_t bytes = sort_bytes;
        check(rocprim::radix_sort_pairs(scratch + off_sort, bytes, keys, keys2, perm_a, perm_b, (unsigned)n, 0u, 32u + seg_bits, st));
        std::swap(perm_a, perm_b);
    }
    if (st_code == PRG_OK && perm_a != perm_dev)
        check(hipMemcpyAsync(perm_dev, perm_a, (size_t)n * sizeof(int), hipMemcpyDeviceToDevice, st));
    check(hipGetLastError());
    check(hipStreamSynchronize(st));  // (the host vectors above were the sources of asynchronous copies; the scratch goes now)
    (void)hipFree(scratch);
    return st_code;
}

}  // namespace prg

extern "C" int prg_spatial_order(const float* points_hd, int64_t n, int dim, int on_device, int* perm_host) {
    PRG_REQUIRE(points_hd && perm_host && n > 0 && (dim == 2 || dim == 3), PRG_ERR_INVALID,
                "prg_spatial_order: need points, an output array, n > 0 and dim in {2, 3}");
    if (!on_device) {
        std::vector<float> host((size_t)n * dim);
        // (a host pointer on a box without a GPU: no HIP runtime to ask - the host build needs none)
        if (hipMemcpy(host.data(), points_hd, host.size() * sizeof(float), hipMemcpyDefault) != hipSuccess) {
            (void)hipGetLastError();
            memcpy(host.data(), points_hd, host.size() * sizeof(float));
        }
        const std::vector<int> perm = prg::kd_order(host.data(), n, dim);
        std::copy(perm.begin(), perm.end(), perm_host);
        return PRG_OK;
    }
    float* pts = nullptr;
    int* perm = nullptr;
    PRG_HIP(hipMalloc((void**)&pts, (size_t)n * dim * sizeof(float)));
    int st = PRG_OK;
    if (hipMalloc((void**)&perm, (size_t)n * sizeof(int)) != hipSuccess) {
        (void)hipFree(pts);
        prg::set_error("prg_spatial_order: out of device memory");
        return PRG_ERR_NOMEM;
    }
    if (hipMemcpy(pts, points_hd, (size_t)n * dim * sizeof(float), hipMemcpyDefault) != hipSuccess) st = PRG_ERR_HIP;
    if (st == PRG_OK) st = prg::device_kd_order(pts, n, dim, perm, nullptr);
    if (st == PRG_OK && hipMemcpy(perm_host, perm, (size_t)n * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) st = PRG_ERR_HIP;
    (void)hipFree(pts);
    (void)hipFree(perm);
    return st;
}
