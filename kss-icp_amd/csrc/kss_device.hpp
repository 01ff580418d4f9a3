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

// cell of a coordinate in a uniform grid (clamped to the grid: queries outside the bbox belong to the border cells)
__device__ __forceinline__ int cell_coord(float v, float o, float inv_h, int g) {
    int c = (int)floorf((v - o) * inv_h);
    c = c < 0 ? 0 : c;
    return c >= g ? g - 1 : c;
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

// ---- the canonical summation order of the fused cell-list pass (grid_pass_kernel) ------------------------------
// One ROW = the sums over one CHUNK of 512 consecutive (cell-sorted) sources of a pair: lane t holds source
// 512 * chunk + t.  Per f64 column the row value is a fixed binary tree over the 64 lanes of each wave -- partners at lane
// distance 32, then 16, 8, 4, 2, 1 -- followed by the 8 wave totals added in wave order.  The pair total adds the rows
// k = g, g + 25, g + 50, ... sequentially for each of 25 groups g, then the 25 group totals in group order.  Everything
// is a function of the pair's source count only, so a pair gives the same bits alone or inside a batch, whatever the
// grid size, and IEEE addition being commutative the tree can be evaluated with any exchange primitive.
constexpr int PASS_BS = 512;                  // lanes per workgroup == sources per chunk / row
constexpr int PASS_FG = PASS_BS / NSUMS;      // 25 row groups of the pair total

template <typename F>
__device__ __forceinline__ double dpp_f64(double x, F f) {
    const int lo = f(__double2loint(x)), hi = f(__double2hiint(x));
    return __hiloint2double(hi, lo);
}
// value of lane (l ^ d) for d = 8, 4, 2, 1 (inside a row of 16 lanes: DPP, no LDS)
// (every lane is written: no "old" value to keep -- with one, the compiler copies the register first, two moves per exchange)
__device__ __forceinline__ double xor8(double x) { return dpp_f64(x, [](int v) { return __builtin_amdgcn_mov_dpp(v, 0x128, 0xf, 0xf, false); }); }   // row_ror:8
__device__ __forceinline__ double xor4(double x) {
    return dpp_f64(x, [](int v) {
        int r = __builtin_amdgcn_mov_dpp(v, 0x104, 0xf, 0x5, false);         // row_shl:4 into banks 0, 2 (lanes with bit 2 clear)
        return __builtin_amdgcn_update_dpp(r, v, 0x114, 0xf, 0xa, false);    // row_shr:4 into banks 1, 3
    });
}
__device__ __forceinline__ double xor2(double x) { return dpp_f64(x, [](int v) { return __builtin_amdgcn_mov_dpp(v, 0x4e, 0xf, 0xf, false); }); }    // quad_perm [2,3,0,1]
__device__ __forceinline__ double xor1(double x) { return dpp_f64(x, [](int v) { return __builtin_amdgcn_mov_dpp(v, 0xb1, 0xf, 0xf, false); }); }    // quad_perm [1,0,3,2]

// a: lanes with the distance bit clear keep a (and receive the partner's a); b: lanes with the bit set keep b.
// Returns a_own + a_partner in the "clear" lanes and b_partner + b_own in the "set" lanes: one tree level of TWO
// columns for the price of one (v_permlane32_swap / v_permlane16_swap exchange half-waves / odd-even rows in place).
// The swaps work IN PLACE on their two registers: as inline asm with read-write operands they need no copies (the builtin
// returns a pair and the compiler copies both inputs first: two moves per swap, fifty-six per tree).  gfx950 wants two wait
// states between a VALU write of either operand and the swap that reads it; the compiler inserts them for the builtin but
// cannot see inside an asm string, so `s_nop 1` stands on both sides of the swap INSIDE the string (cdna_hip_programming.md T21: the
// documented hazard is the one before the swap; the one behind it is what the compiler's own code shows).  Without it the low
// words came back stale, every sum of every engine was off by ~1e-8 relative, and only tools/tree_check.hip (now a test)
// noticed -- because every engine was off alike.
__device__ __forceinline__ double level32(double a, double b) {
    unsigned alo = (unsigned)__double2loint(a), ahi = (unsigned)__double2hiint(a), blo = (unsigned)__double2loint(b), bhi = (unsigned)__double2hiint(b);
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(alo), "+v"(blo));
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(ahi), "+v"(bhi));
    return __hiloint2double((int)ahi, (int)alo) + __hiloint2double((int)bhi, (int)blo);
}
__device__ __forceinline__ double level16(double a, double b) {
    unsigned alo = (unsigned)__double2loint(a), ahi = (unsigned)__double2hiint(a), blo = (unsigned)__double2loint(b), bhi = (unsigned)__double2hiint(b);
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(alo), "+v"(blo));
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(ahi), "+v"(bhi));
    return __hiloint2double((int)ahi, (int)alo) + __hiloint2double((int)bhi, (int)blo);
}
__device__ __forceinline__ double tree_low4(double x) {   // levels 8, 4, 2, 1: every lane of a 16-lane row ends with the row total
    x += xor8(x); x += xor4(x); x += xor2(x); x += xor1(x);
    return x;
}

// Wave totals of 16 columns: afterwards lane 16 * q (q = 0..3) holds columns 4q .. 4q+3 in v[0..3].
__device__ __forceinline__ void wave_tree16(double (&v)[16]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = level32(v[j], v[j + 8]);     // lanes < 32: columns 0-7, lanes >= 32: columns 8-15
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = level16(v[j], v[j + 4]);     // bit 4 clear: first four of those, set: the other four
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = tree_low4(v[j]);
}
// Wave totals of 2 columns: lane 0 holds a, lane 32 holds b (same tree per column as above).
__device__ __forceinline__ double wave_tree2(double a, double b) {
    double x = level32(a, b);
    x = level16(x, x);
    return tree_low4(x);
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
