// kss_octree.hip -- octree down-sampler (`Method_Octree.hpp`, named by the north star as
// a replaced subsystem; off the live registration path).
//
// Reference: PCL_octree::PCL_Octree_Simplification_WithOutNormal (Method_Octree.hpp:77-104) over
// pcl::octree::OctreePointCloudSearch (PCL 1.8.1): the occupied voxel centres of an octree of a data-derived
// resolution, in the octree's depth-first order, each replaced by its nearest cloud point.  No tree is built here:
//   * the bounding cube PCL grows point by point (adoptBoundingBoxToPoint: it starts as the first point +- resolution
//     and doubles towards every point that falls outside, in insertion order) is replayed on the host -- a few
//     compares per point, inherently sequential;
//   * oct_code_kernel turns every point into its voxel key (genOctreeKeyforPoint, double arithmetic) packed as a Morton
//     code with the x bit most significant at every level, so that ASCENDING CODE == PCL's depth-first child order;
//   * sort + unique of the codes (rocPRIM device radix sort / unique) = the occupied voxels in output order;
//   * oct_center_kernel decodes them into voxel centres (genLeafNodeCenterFromOctreeKey);
//   * the nearest cloud point of every centre is the exact NN engine of this library (kss_engine.hip).
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "kss_internal.hpp"

namespace kss {

__global__ __launch_bounds__(256) void oct_code_kernel(const float* __restrict__ pts, int n, OctBox b, unsigned long long* __restrict__ code) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned key[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) key[k] = (unsigned)(((double)pts[3 * (int64_t)i + k] - b.min[k]) / b.res);
    unsigned long long c = 0;
    for (int lev = b.depth - 1; lev >= 0; --lev)
        c = (c << 3) | (unsigned long long)((((key[0] >> lev) & 1u) << 2) | (((key[1] >> lev) & 1u) << 1) | ((key[2] >> lev) & 1u));
    code[i] = c;
}

__global__ __launch_bounds__(256) void oct_center_kernel(const unsigned long long* __restrict__ code, int m, OctBox b, float* __restrict__ cen) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= m) return;
    const unsigned long long c = code[v];
    unsigned key[3] = {0, 0, 0};
    for (int lev = 0; lev < b.depth; ++lev) {
        const unsigned tri = (unsigned)((c >> (3 * (b.depth - 1 - lev))) & 7ull);
        key[0] = (key[0] << 1) | ((tri >> 2) & 1u); key[1] = (key[1] << 1) | ((tri >> 1) & 1u); key[2] = (key[2] << 1) | (tri & 1u);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) cen[3 * (int64_t)v + k] = (float)(((double)key[k] + 0.5f) * b.res + b.min[k]);
}

// codes of all points -> sorted unique codes -> voxel centres.  `scratch(bytes)` hands out device memory owned by the
// context (grow-only); returns the number of voxels in *m_out and the centres in d_cen (capacity n).
int octree_voxels_device(hipStream_t st, const float* d_pts, int n, const OctBox& box, float* d_cen, int* m_out, std::string& err,
                         const std::function<void*(int, size_t)>& scratch) {
    unsigned long long* d_code = (unsigned long long*)scratch(0, (size_t)n * sizeof(unsigned long long));
    unsigned long long* d_sorted = (unsigned long long*)scratch(1, (size_t)n * sizeof(unsigned long long));
    unsigned* d_count = (unsigned*)scratch(2, 64);
    if (!d_code || !d_sorted || !d_count) { err = "octree: out of device memory"; return KSS_ERR_NOMEM; }
    hipLaunchKernelGGL(oct_code_kernel, dim3((n + 255) / 256), dim3(256), 0, st, d_pts, n, box, d_code);
    size_t b1 = 0, b2 = 0;
    if (rocprim::radix_sort_keys(nullptr, b1, d_code, d_sorted, (size_t)n, 0, 3 * (unsigned)box.depth, st) != hipSuccess ||
        rocprim::unique(nullptr, b2, d_sorted, d_code, d_count, (size_t)n, rocprim::equal_to<unsigned long long>(), st) != hipSuccess) {
        err = "octree: rocPRIM size query failed";
        return KSS_ERR_HIP;
    }
    void* d_tmp = scratch(3, std::max(b1, b2));
    if (!d_tmp) { err = "octree: out of device memory"; return KSS_ERR_NOMEM; }
    if (rocprim::radix_sort_keys(d_tmp, b1, d_code, d_sorted, (size_t)n, 0, 3 * (unsigned)box.depth, st) != hipSuccess) { err = "octree: sort failed"; return KSS_ERR_HIP; }
    if (rocprim::unique(d_tmp, b2, d_sorted, d_code, d_count, (size_t)n, rocprim::equal_to<unsigned long long>(), st) != hipSuccess) { err = "octree: unique failed"; return KSS_ERR_HIP; }
    unsigned m = 0;
    if (hipMemcpyAsync(&m, d_count, sizeof m, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
        err = "octree: reading the voxel count failed";
        return KSS_ERR_HIP;
    }
    hipLaunchKernelGGL(oct_center_kernel, dim3((m + 255) / 256), dim3(256), 0, st, d_code, (int)m, box, d_cen);
    if (hipGetLastError() != hipSuccess) { err = "octree: kernel launch failed"; return KSS_ERR_HIP; }
    *m_out = (int)m;
    return KSS_OK;
}

}  // namespace kss
