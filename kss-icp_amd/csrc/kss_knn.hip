// kss_knn.hip -- exact k-nearest-neighbour sweep and PCL-style surface normals (the
// "Method_Octree / pcl_kdtree NN" and "normalCompute reductions" the north star lists as replaced subsystems).
//
//   knn_sweep_kernel<K>   (K = 4 .. 64) replaces pcl::KdTreeFLANN::nearestKSearch with K > 1 (ballRegionCompute.hpp:499 K=13,
//                         Method_AIVS_SimPro.hpp:904 K=3, Method_Octree.hpp:137, pcl::NormalEstimation K=20): the same
//                         LDS-tiled source x target sweep as nn_sweep_kernel, each lane keeping a sorted top-K in
//                         registers (strict '<' insertion while targets arrive in ascending index => equal distances
//                         keep the lower index first: ascending (d2, index) order, bit-identical to the oracle).
//   normals_kernel        estimateNormal_PCL_MP_return (normalCompute.hpp:308-355): pcl::NormalEstimationOMP with
//                         k = 20 and view point (0,0,0) -- float single-pass mean/covariance over the k neighbours,
//                         pcl::eigen33 closed-form smallest eigenvector, flip towards the view point -- then the
//                         reference's renormalisation in double.
#pragma clang fp contract(off)

#include <hip/hip_runtime.h>

#include <algorithm>

#include "kss_internal.hpp"
#include "kss_device.hpp"

namespace kss {

// Work item (blockIdx.x, blockIdx.y) = 256 queries x target split blockIdx.y (tiles [y * tiles_per_split, ...)).  With one
// split the sorted top-K goes straight to idx_out / d2_out [query][k_out]; with several, every split writes its own
// top-K to part_* [split][query][K] and knn_merge_kernel merges them.  Splits exist because a few queries against a
// large cloud (the octree resolution: 1000 queries x 10^6 points) would otherwise run on four workgroups.
template <int K>
__global__ __launch_bounds__(256) void knn_sweep_kernel(const float4* __restrict__ qry, int nq, const float4* __restrict__ tgt, int nt_pad,
                                                        int tiles_per_split, int32_t* __restrict__ idx_out, float* __restrict__ d2_out, int k_out,
                                                        int32_t* __restrict__ part_idx, float* __restrict__ part_d2) {
    __shared__ float4 tile[2][NN_TILE];
    const int tid = threadIdx.x;
    const int i = blockIdx.x * 256 + tid;
    const bool valid = i < nq;
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) p = qry[i];
    float kd[K];
    int ki[K];
#pragma unroll
    for (int c = 0; c < K; ++c) { kd[c] = __builtin_inff(); ki[c] = -1; }
    const int t_begin = (int)blockIdx.y * tiles_per_split;
    const int t_end = min(nt_pad / NN_TILE, t_begin + tiles_per_split);
    float4 pre = tgt[t_begin * NN_TILE + tid];
    int buf = 0;
    for (int t = t_begin; t < t_end; ++t) {
        tile[buf][tid] = pre;
        __syncthreads();
        if (t + 1 < t_end) pre = tgt[(t + 1) * NN_TILE + tid];
        const float4* __restrict__ tl = tile[buf];
#pragma unroll 4
        for (int u = 0; u < NN_TILE; ++u) {
            const float4 q = tl[u];
            const float d = dist2<false>(p.x, p.y, p.z, q.x, q.y, q.z);
            const bool ins = d < kd[K - 1];
            // A WAVE-UNIFORM branch (ballot): the insertion below is all selects, and left to itself the compiler
            // if-converts it into ~8 K instructions executed for EVERY target (K = 32: 20x the cost of the sweep).
            // After the first few tiles almost no target enters anyone's top-K and the branch is not taken.
            if (__builtin_amdgcn_ballot_w64(ins) != 0ull) {
                // shift-insert into the ascending list: entries larger than d move up one slot, d lands behind the
                // last entry <= d (strict '<': an equal distance stays behind the earlier index).  Every new slot
                // value depends on two FIXED registers only (descending c reads kd[c - 1] before it changes).
                const int id = t * NN_TILE + u;
#pragma unroll
                for (int c = K - 1; c > 0; --c) {
                    const bool shift = ins && d < kd[c - 1];
                    const bool place = ins && !shift && d < kd[c];
                    kd[c] = shift ? kd[c - 1] : (place ? d : kd[c]);
                    ki[c] = shift ? ki[c - 1] : (place ? id : ki[c]);
                }
                const bool front = ins && d < kd[0];
                kd[0] = front ? d : kd[0];
                ki[0] = front ? id : ki[0];
            }
        }
        buf ^= 1;
    }
    if (!valid) return;
    if (gridDim.y == 1) {
#pragma unroll
        for (int c = 0; c < K; ++c)
            if (c < k_out) {
                idx_out[(int64_t)i * k_out + c] = ki[c];
                d2_out[(int64_t)i * k_out + c] = kd[c];
            }
    } else {
        const int64_t base = ((int64_t)blockIdx.y * nq + i) * K;
#pragma unroll
        for (int c = 0; c < K; ++c) { part_idx[base + c] = ki[c]; part_d2[base + c] = kd[c]; }
    }
}

// merge of the per-split top-K lists of one query: each list is ascending in (d2, index) and the splits are ascending
// index ranges, so taking the lexicographically smallest head k_out times reproduces the single-sweep order exactly.
// One thread per query; the heads' cursors live in LDS (n_split <= KNN_MAX_SPLIT bytes per thread).
constexpr int KNN_MAX_SPLIT = 64;
__global__ __launch_bounds__(64) void knn_merge_kernel(const int32_t* __restrict__ part_idx, const float* __restrict__ part_d2, int nq, int K,
                                                       int n_split, int32_t* __restrict__ idx_out, float* __restrict__ d2_out, int k_out) {
    __shared__ unsigned char cursor[KNN_MAX_SPLIT][64];
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= nq) return;
    for (int s = 0; s < n_split; ++s) cursor[s][threadIdx.x] = 0;
    for (int c = 0; c < k_out; ++c) {
        float bd = __builtin_inff();
        int bi = -1, bs = -1;
        for (int s = 0; s < n_split; ++s) {
            const int cu = cursor[s][threadIdx.x];
            if (cu >= K) continue;
            const int64_t at = ((int64_t)s * nq + i) * K + cu;
            const float d = part_d2[at];
            const int id = part_idx[at];
            if (id < 0) continue;                                   // list exhausted (fewer than K targets in the split)
            if (bs < 0 || d < bd || (d == bd && id < bi)) { bd = d; bi = id; bs = s; }
        }
        idx_out[(int64_t)i * k_out + c] = bi;
        d2_out[(int64_t)i * k_out + c] = bs >= 0 ? bd : __builtin_inff();
        if (bs >= 0) cursor[bs][threadIdx.x] = (unsigned char)(cursor[bs][threadIdx.x] + 1);
    }
}

// n_split and the scratch the caller must provide for it (0 bytes: single sweep)
int knn_plan_splits(int nq, int nt_pad, int k, size_t* scratch_bytes) {
    const int blocks = (nq + 255) / 256, tiles = nt_pad / NN_TILE;
    int n_split = 1;
    if (blocks < 256 && tiles >= 8) n_split = std::min(std::min(KNN_MAX_SPLIT, (1024 + blocks - 1) / blocks), tiles / 4);
    if (n_split < 1) n_split = 1;
    const int K = k <= 4 ? 4 : k <= 8 ? 8 : k <= 16 ? 16 : k <= 32 ? 32 : 64;
    *scratch_bytes = n_split > 1 ? (size_t)n_split * nq * K * (sizeof(int32_t) + sizeof(float)) : 0;
    return n_split;
}

void launch_knn_sweep(hipStream_t st, const float4* d_qry, int nq, const float4* d_tgt, int nt_pad, int k, int32_t* d_idx, float* d_d2,
                      int n_split, void* d_scratch) {
    const int tiles = nt_pad / NN_TILE;
    const int tps = (tiles + n_split - 1) / n_split;
    const int planned = n_split;
    n_split = (tiles + tps - 1) / tps;                                  // no empty trailing split
    const dim3 grid((nq + 255) / 256, n_split), block(256);
    const int K = k <= 4 ? 4 : k <= 8 ? 8 : k <= 16 ? 16 : k <= 32 ? 32 : 64;
    int32_t* p_idx = (int32_t*)d_scratch;
    float* p_d2 = planned > 1 ? (float*)((char*)d_scratch + (size_t)planned * nq * K * sizeof(int32_t)) : nullptr;
#define KSS_KNN_LAUNCH(KV) hipLaunchKernelGGL(knn_sweep_kernel<KV>, grid, block, 0, st, d_qry, nq, d_tgt, nt_pad, tps, d_idx, d_d2, k, p_idx, p_d2)
    if (K == 4) KSS_KNN_LAUNCH(4); else if (K == 8) KSS_KNN_LAUNCH(8); else if (K == 16) KSS_KNN_LAUNCH(16); else if (K == 32) KSS_KNN_LAUNCH(32); else KSS_KNN_LAUNCH(64);
#undef KSS_KNN_LAUNCH
    if (n_split > 1)   // (gridDim.y == 1 wrote the final lists itself)
        hipLaunchKernelGGL(knn_merge_kernel, dim3((nq + 63) / 64), dim3(64), 0, st, p_idx, p_d2, nq, K, n_split, d_idx, d_d2, k);
}

// ---- pcl::eigen33 smallest eigenpair, float, closed form (common/eigen.hpp) -----------------------------------------
__device__ __forceinline__ void pcl_roots2(float b, float c, float (&r)[3]) {
    r[0] = 0.f;
    float d = b * b - 4.0f * c;
    if (d < 0.0f) d = 0.0f;
    const float sd = sqrtf(d);
    r[2] = 0.5f * (b + sd);
    r[1] = 0.5f * (b - sd);
}

__device__ __forceinline__ void pcl_roots(const float (&m)[9], float (&r)[3]) {
    const float c0 = m[0] * m[4] * m[8] + 2.0f * m[1] * m[2] * m[5] - m[0] * m[5] * m[5] - m[4] * m[2] * m[2] - m[8] * m[1] * m[1];
    const float c1 = m[0] * m[4] - m[1] * m[1] + m[0] * m[8] - m[2] * m[2] + m[4] * m[8] - m[5] * m[5];
    const float c2 = m[0] + m[4] + m[8];
    if (fabsf(c0) < 1.1920929e-07f) { pcl_roots2(c2, c1, r); return; }
    const float s_inv3 = (float)(1.0 / 3.0), s_sqrt3 = sqrtf(3.0f);
    const float c2_over_3 = c2 * s_inv3;
    float a_over_3 = (c1 - c2 * c2_over_3) * s_inv3;
    if (a_over_3 > 0.f) a_over_3 = 0.f;
    const float half_b = 0.5f * (c0 + c2_over_3 * (2.0f * c2_over_3 * c2_over_3 - c1));
    float qq = half_b * half_b + a_over_3 * a_over_3 * a_over_3;
    if (qq > 0.f) qq = 0.f;
    const float rho = sqrtf(-a_over_3);
    const float theta = atan2f(sqrtf(-qq), half_b) * s_inv3;
    const float ct = cosf(theta), st = sinf(theta);
    r[0] = c2_over_3 + 2.0f * rho * ct;
    r[1] = c2_over_3 - rho * (ct + s_sqrt3 * st);
    r[2] = c2_over_3 - rho * (ct - s_sqrt3 * st);
    float tmp;
    if (r[0] >= r[1]) { tmp = r[0]; r[0] = r[1]; r[1] = tmp; }
    if (r[1] >= r[2]) {
        tmp = r[1]; r[1] = r[2]; r[2] = tmp;
        if (r[0] >= r[1]) { tmp = r[0]; r[0] = r[1]; r[1] = tmp; }
    }
    if (r[0] <= 0) pcl_roots2(c2, c1, r);
}

__global__ __launch_bounds__(256) void normals_kernel(const float4* __restrict__ pts, int n, const int32_t* __restrict__ knn, int k,
                                                      double* __restrict__ normals) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // computeMeanAndCovarianceMatrix (float, single pass, neighbours in kNN order)
    float a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int c = 0; c < k; ++c) {
        const float4 p = pts[knn[(int64_t)i * k + c]];
        a[0] += p.x * p.x; a[1] += p.x * p.y; a[2] += p.x * p.z;
        a[3] += p.y * p.y; a[4] += p.y * p.z; a[5] += p.z * p.z;
        a[6] += p.x; a[7] += p.y; a[8] += p.z;
    }
#pragma unroll
    for (int c = 0; c < 9; ++c) a[c] /= (float)k;
    float m[9];
    m[0] = a[0] - a[6] * a[6]; m[1] = a[1] - a[6] * a[7]; m[2] = a[2] - a[6] * a[8];
    m[4] = a[3] - a[7] * a[7]; m[5] = a[4] - a[7] * a[8]; m[8] = a[5] - a[8] * a[8];
    m[3] = m[1]; m[6] = m[2]; m[7] = m[5];
    float scale = 0.f;
#pragma unroll
    for (int c = 0; c < 9; ++c) scale = fmaxf(scale, fabsf(m[c]));
    if (scale <= 1.17549435e-38f) scale = 1.0f;
    float sm[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) sm[c] = m[c] / scale;
    float r[3];
    pcl_roots(sm, r);
    sm[0] -= r[0]; sm[4] -= r[0]; sm[8] -= r[0];
    const float v1[3] = {sm[1] * sm[5] - sm[2] * sm[4], sm[2] * sm[3] - sm[0] * sm[5], sm[0] * sm[4] - sm[1] * sm[3]};
    const float v2[3] = {sm[1] * sm[8] - sm[2] * sm[7], sm[2] * sm[6] - sm[0] * sm[8], sm[0] * sm[7] - sm[1] * sm[6]};
    const float v3[3] = {sm[4] * sm[8] - sm[5] * sm[7], sm[5] * sm[6] - sm[3] * sm[8], sm[3] * sm[7] - sm[4] * sm[6]};
    const float l1 = v1[0] * v1[0] + v1[1] * v1[1] + v1[2] * v1[2], l2 = v2[0] * v2[0] + v2[1] * v2[1] + v2[2] * v2[2],
                l3 = v3[0] * v3[0] + v3[1] * v3[1] + v3[2] * v3[2];
    float nx, ny, nz;
    if (l1 >= l2 && l1 >= l3) { const float s = sqrtf(l1); nx = v1[0] / s; ny = v1[1] / s; nz = v1[2] / s; }
    else if (l2 >= l1 && l2 >= l3) { const float s = sqrtf(l2); nx = v2[0] / s; ny = v2[1] / s; nz = v2[2] / s; }
    else { const float s = sqrtf(l3); nx = v3[0] / s; ny = v3[1] / s; nz = v3[2] / s; }
    const float4 p = pts[i];
    const float vx = 0.f - p.x, vy = 0.f - p.y, vz = 0.f - p.z;   // flipNormalTowardsViewpoint, view point (0, 0, 0)
    const float cos_theta = vx * nx + vy * ny + vz * nz;
    if (cos_theta < 0) { nx *= -1; ny *= -1; nz *= -1; }
    const double dis = sqrt((double)nx * (double)nx + (double)ny * (double)ny + (double)nz * (double)nz);   // normalCompute.hpp:342-348
    normals[3 * (int64_t)i] = (double)nx / dis;
    normals[3 * (int64_t)i + 1] = (double)ny / dis;
    normals[3 * (int64_t)i + 2] = (double)nz / dis;
}

void launch_normals(hipStream_t st, const float4* d_pts, int n, const int32_t* d_knn, int k, double* d_normals) {
    hipLaunchKernelGGL(normals_kernel, dim3((n + 255) / 256), dim3(256), 0, st, d_pts, n, d_knn, k, d_normals);
}

}  // namespace kss
