// kss_device.hpp -- device-side helpers shared by kss_kernels.hip and kss_grid.hip (gfx950).
#pragma once
#include <hip/hip_runtime.h>

#include "kss_internal.hpp"

namespace kss {

template <bool FMA>
__device__ __forceinline__ float dist2(float sx, float sy, float sz, float tx, float ty, float tz) {
    const float dx = sx - tx, dy = sy - ty, dz = sz - tz;
    if constexpr (FMA) {
        return __builtin_fmaf(dx, dx, __builtin_fmaf(dy, dy, dz * dz));
    } else {
        return (dx * dx + dy * dy) + dz * dz;   // FLANN L2_Simple: result += diff*diff in x,y,z order
    }
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// Block-wide sum of NV doubles per thread (256 threads = 4 waves). Result valid in thread c < NV.
template <int NV>
__device__ __forceinline__ double block_sum(const double (&v)[NV], double (*sh)[NV]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < NV; ++c) {
        const double r = wave_sum(v[c]);
        if (lane == 0) sh[wave][c] = r;
    }
    __syncthreads();
    double out = 0.0;
    if (threadIdx.x < NV) {
        const int nw = (blockDim.x + 63) >> 6;
        for (int w = 0; w < nw; ++w) out += sh[w][threadIdx.x];   // fixed order: reproducible
    }
    return out;
}


// Column sums of `nrows` partial rows (NSUMS doubles each) by one 256-lane workgroup, bitwise reproducible:
// lane (g, c) = (tid / NSUMS, tid % NSUMS) adds rows g, g+12, g+24, ... of column c with four independent
// accumulators (many loads in flight: the rows come from other CUs' L2 / HBM), then the 12 group totals are
// added in group order.  Returns the column total in lanes tid < NSUMS.  Needs blockDim.x == 256.
constexpr int ROWSUM_GROUPS = 12;   // 12 * 20 = 240 of the 256 lanes
__device__ __forceinline__ double rows_column_sum(const double* __restrict__ rows, int nrows, double (*shg)[NSUMS]) {
    const int g = threadIdx.x / NSUMS, c = threadIdx.x % NSUMS;
    if (g < ROWSUM_GROUPS) {
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, a4 = 0.0, a5 = 0.0, a6 = 0.0, a7 = 0.0;
        int k = g;
        constexpr int G = ROWSUM_GROUPS;
        for (; k + 7 * G < nrows; k += 8 * G) {
            a0 += rows[(int64_t)(k) * NSUMS + c];
            a1 += rows[(int64_t)(k + G) * NSUMS + c];
            a2 += rows[(int64_t)(k + 2 * G) * NSUMS + c];
            a3 += rows[(int64_t)(k + 3 * G) * NSUMS + c];
            a4 += rows[(int64_t)(k + 4 * G) * NSUMS + c];
            a5 += rows[(int64_t)(k + 5 * G) * NSUMS + c];
            a6 += rows[(int64_t)(k + 6 * G) * NSUMS + c];
            a7 += rows[(int64_t)(k + 7 * G) * NSUMS + c];
        }
        for (; k < nrows; k += G) a0 += rows[(int64_t)k * NSUMS + c];
        a0 = (a0 + a1) + (a2 + a3);
        a2 = (a4 + a5) + (a6 + a7);
        a1 = 0.0; a3 = 0.0;
        a0 = a0 + a2; a2 = 0.0;
        shg[g][c] = (a0 + a1) + (a2 + a3);
    }
    __syncthreads();
    double v = 0.0;
    if (threadIdx.x < NSUMS)
        for (int gg = 0; gg < ROWSUM_GROUPS; ++gg) v += shg[gg][threadIdx.x];
    return v;
}

// correspondence sums of one (source, matched target) pair; see KSS_NSUMS in include/kssicp.h
__device__ __forceinline__ void accumulate_corr(double (&acc)[NSUMS], float px, float py, float pz,
                                                float qx, float qy, float qz, float d2f, double max_d2) {
    const double d2 = (double)d2f;
    acc[17] += d2;
    acc[18] += sqrt(d2);
    if (!(d2 > max_d2)) {   // PCL: `if (distance[0] > max_dist_sqr) continue;`
        const double p[3] = {(double)px, (double)py, (double)pz};
        const double q[3] = {(double)qx, (double)qy, (double)qz};
        acc[0] += 1.0;
#pragma unroll
        for (int k = 0; k < 3; ++k) { acc[1 + k] += p[k]; acc[4 + k] += q[k]; }
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int l = 0; l < 3; ++l) acc[7 + 3 * k + l] += p[k] * q[l];
        acc[16] += d2;
    }
}


}  // namespace kss
