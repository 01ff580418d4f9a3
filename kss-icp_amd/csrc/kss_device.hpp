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
