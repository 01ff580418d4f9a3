// kss_aivs.hip -- AIVS down-sampler on the GPU (the component immediately upstream of
// the registration hot path).  Replaces, with the same arithmetic and the same selection order:
//   pointPipeline_init_point_withoutUniform (pointPipeline.hpp:88-101, border :105-160),
//   BallRegion_init_withoutNormal (ballRegionCompute.hpp:114-147): AchieveXYZ :690-758, BoxInput :632-688,
//     box centres :1150-1172, neighbour boxes :975-1031 (quirks of the last x column included), scale :1194-1215,
//   AIVS_simplification (Method_AIVS_SimPro.hpp:94-154): 8-colour schedule :587-643, per-box budget :776-794,
//     per-voxel farthest-point sampling AIVS_Voroni_OpenMP_KNN :222-376, AIVS_AccurateCut_Optimization :848-957.
// Not built: the K=13 self-kNN radius estimate (ballRegionCompute.hpp:477-530); nothing on this path reads it.
//
// Mapping: the reference runs the boxes of one colour in an OpenMP loop (8 sequential colours, because a box reads
// the samples already placed in its 26 neighbours); here each colour is one launch with ONE WAVE PER BOX.  Boxes
// hold tens of points, so a wave keeps a box's running min-distances in registers/global and picks the next
// sample with a wave-wide arg-max (first maximum wins, as the serial scan does).  Distances are the reference's:
// float PointXYZ, d2 = (dx*dx + dy*dy) + dz*dz without fma, then sqrt(float) (correctly rounded on gfx950).
#pragma clang fp contract(off)

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <functional>
#include <string>
#include <vector>

#include "kss_internal.hpp"

namespace kss {

struct AivsGrid {
    int nx, ny, nz, nboxes;   // boxes are 1-based: index = x + nx*(y-1) + nx*ny*(z-1); slot 0 unused
    double minx, miny, minz, unit;
    double rate, search_radius;
};

// ---- bounding box (f64) ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void aivs_bbox_kernel(const double* __restrict__ P, int n, double* __restrict__ partial) {
    __shared__ double sh[4][6];
    double mn[3] = {__builtin_inf(), __builtin_inf(), __builtin_inf()}, mx[3] = {-__builtin_inf(), -__builtin_inf(), -__builtin_inf()};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        for (int k = 0; k < 3; ++k) {
            const double v = P[3 * (int64_t)i + k];
            mn[k] = fmin(mn[k], v);
            mx[k] = fmax(mx[k], v);
        }
    for (int off = 32; off > 0; off >>= 1)
        for (int k = 0; k < 3; ++k) {
            mn[k] = fmin(mn[k], __shfl_down(mn[k], off, 64));
            mx[k] = fmax(mx[k], __shfl_down(mx[k], off, 64));
        }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0)
        for (int k = 0; k < 3; ++k) { sh[wave][k] = mn[k]; sh[wave][3 + k] = mx[k]; }
    __syncthreads();
    if (threadIdx.x < 6) {
        double v = sh[0][threadIdx.x];
        for (int w = 1; w < 4; ++w) v = threadIdx.x < 3 ? fmin(v, sh[w][threadIdx.x]) : fmax(v, sh[w][threadIdx.x]);
        partial[blockIdx.x * 6 + threadIdx.x] = v;
    }
}

// ---- box of a point: BallRegion_BoxInput, ballRegionCompute.hpp:646-664 -------------------------------------------
__device__ __forceinline__ int aivs_box_of(const double* __restrict__ P, int i, const AivsGrid& g) {
    // 1-based box coordinates: ceil of the scaled offset, with 0 mapped to 1 (the reference's rule, in its f64 arithmetic)
    const double fx = (P[3 * (int64_t)i] - g.minx) / g.unit, fy = (P[3 * (int64_t)i + 1] - g.miny) / g.unit,
                 fz = (P[3 * (int64_t)i + 2] - g.minz) / g.unit;
    int bx = (int)fx, by = (int)fy, bz = (int)fz;
    if (bx < fx || bx == 0) bx++;
    if (by < fy || by == 0) by++;
    if (bz < fz || bz == 0) bz++;
    return bx + g.nx * (by - 1) + g.nx * g.ny * (bz - 1);
}

__global__ __launch_bounds__(256) void aivs_count_kernel(const double* __restrict__ P, int n, AivsGrid g, int32_t* __restrict__ box_of,
                                                         int32_t* __restrict__ counts, int32_t* __restrict__ bad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int b = aivs_box_of(P, i, g);
    if (b < 0 || b > g.nboxes) { atomicAdd(bad, 1); box_of[i] = 0; return; }   // the reference prints "Hello!" and indexes out of range
    box_of[i] = b;
    atomicAdd(&counts[b], 1);
}

__global__ __launch_bounds__(256) void aivs_scatter_kernel(const int32_t* __restrict__ box_of, int n, int32_t* __restrict__ cursor,
                                                           int32_t* __restrict__ members_tmp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    members_tmp[atomicAdd(&cursor[box_of[i]], 1)] = i;
}

// squareBoxes[b] holds its points in push_back (= ascending index) order: rank every member inside its box
__global__ __launch_bounds__(256) void aivs_rank_kernel(const int32_t* __restrict__ members_tmp, const int32_t* __restrict__ box_of, int n,
                                                        const int32_t* __restrict__ start, int32_t* __restrict__ members) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int me = members_tmp[j];
    const int b = box_of[me];
    const int lo = start[b], hi = start[b + 1];
    int rank = 0;
    for (int k = lo; k < hi; ++k) rank += members_tmp[k] < me ? 1 : 0;
    members[lo + rank] = me;
}

__device__ __forceinline__ void aivs_box_center(const AivsGrid& g, int box, double c[3]) {   // :1150-1172
    // inverse of aivs_box_of's linear index (1-based coordinates; a multiple of nx belongs to the last column of the row before)
    int bz = box / (g.nx * g.ny) + 1;
    const int in_layer = box % (g.nx * g.ny);
    int by = in_layer / g.nx + 1;
    int bx = in_layer % g.nx;
    if (bx == 0) { bx = g.nx; by = by - 1; }
    c[0] = (g.minx + (bx - 1) * g.unit + g.minx + bx * g.unit) / 2;
    c[1] = (g.miny + (by - 1) * g.unit + g.miny + by * g.unit) / 2;
    c[2] = (g.minz + (bz - 1) * g.unit + g.minz + bz * g.unit) / 2;
}

// per box: the member closest to the box centre (:634-686, strict '>' keeps the first minimum) and the sampling
// budget (Method_AIVS_SimPro.hpp:776-794)
__global__ __launch_bounds__(256) void aivs_box_info_kernel(const double* __restrict__ P, AivsGrid g, const int32_t* __restrict__ start,
                                                            const int32_t* __restrict__ members, int32_t* __restrict__ center_pos,
                                                            int32_t* __restrict__ sim_num) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > g.nboxes) return;
    double c[3];
    aivs_box_center(g, b, c);
    double best = 9999;
    int pos = -1;
    const int lo = start[b], hi = start[b + 1];
    for (int k = lo; k < hi; ++k) {
        const int i = members[k];
        const double dx = c[0] - P[3 * (int64_t)i], dy = c[1] - P[3 * (int64_t)i + 1], dz = c[2] - P[3 * (int64_t)i + 2];
        const double d = sqrt(dx * dx + dy * dy + dz * dz);
        if (best > d) { best = d; pos = k - lo; }
    }
    center_pos[b] = pos;
    const double simBox = (double)(hi - lo) * g.rate;
    const int t = (int)simBox;
    sim_num[b] = (simBox - t > 0.2) ? t + 1 : t;
}

__device__ __forceinline__ float aivs_dist(const double* __restrict__ P, int a, int b) {
    const float ax = (float)P[3 * (int64_t)a], ay = (float)P[3 * (int64_t)a + 1], az = (float)P[3 * (int64_t)a + 2];
    const float bx = (float)P[3 * (int64_t)b], by = (float)P[3 * (int64_t)b + 1], bz = (float)P[3 * (int64_t)b + 2];
    const float dx = ax - bx, dy = ay - by, dz = az - bz;
    return sqrtf((dx * dx + dy * dy) + dz * dz);
}

__device__ __forceinline__ int aivs_neighbor_boxes(const AivsGrid& g, int boxIndex, int out[26]) {   // :975-1031, quirks kept
    const int z_num = boxIndex / (g.nx * g.ny) + 1;
    const int leveZ = boxIndex % (g.nx * g.ny);
    const int y_num = leveZ / g.nx + 1;
    const int x_num = leveZ % g.nx;   // not wrapped for the last x column, exactly as the reference
    int xs[3], ys[3], zs[3], nxs = 0, nys = 0, nzs = 0;
    if (x_num > 1) xs[nxs++] = x_num - 1;
    xs[nxs++] = x_num;
    if (x_num < g.nx) xs[nxs++] = x_num + 1;
    if (y_num > 1) ys[nys++] = y_num - 1;
    ys[nys++] = y_num;
    if (y_num < g.ny) ys[nys++] = y_num + 1;
    if (z_num > 1) zs[nzs++] = z_num - 1;
    zs[nzs++] = z_num;
    if (z_num < g.nz) zs[nzs++] = z_num + 1;
    int m = 0;
    for (int i = 0; i < nxs; ++i)
        for (int j = 0; j < nys; ++j)
            for (int k = 0; k < nzs; ++k) {
                if (xs[i] == x_num && ys[j] == y_num && zs[k] == z_num) continue;
                const int idx = xs[i] + (ys[j] - 1) * g.nx + (zs[k] - 1) * g.nx * g.ny;
                if (idx < g.nboxes + 1 && idx >= 0) out[m++] = idx;
            }
    return m;
}


// ---- the fast form of one box's farthest-point sampling: one wave, own points in registers (four per lane), the sampled
// neighbours' coordinates in LDS.  Same float arithmetic and the same "first maximum in scan order" rule as the general
// path below.  The 26 neighbour boxes are scanned as ONE flattened list (their member ranges are fetched at once: three
// dependent memory round trips instead of 26 x 3).  Returns false -- having written nothing -- when more than `cap`
// neighbour samples are in reach; the caller then takes the general path.
__device__ __forceinline__ int aivs_dpp_xor(int v, int d) {   // lane ^ d for d = 8, 4, 2, 1 (inside a row of 16 lanes)
    if (d == 8) return __builtin_amdgcn_mov_dpp(v, 0x128, 0xf, 0xf, false);
    if (d == 4) { const int r = __builtin_amdgcn_mov_dpp(v, 0x104, 0xf, 0x5, false); return __builtin_amdgcn_update_dpp(r, v, 0x114, 0xf, 0xa, false); }
    if (d == 2) return __builtin_amdgcn_mov_dpp(v, 0x4e, 0xf, 0xf, false);
    return __builtin_amdgcn_mov_dpp(v, 0xb1, 0xf, 0xf, false);
}
__device__ __forceinline__ bool aivs_fps_fast(const double* __restrict__ P, const AivsGrid& g, int b, int lo, int m, int simNum, const double (&pc)[3], double R,
                                              const int (&nbr)[26], int nn, const int32_t* __restrict__ start, const int32_t* __restrict__ members,
                                              const int32_t* __restrict__ center_pos, uint8_t* __restrict__ labelG, int32_t* __restrict__ simiT,
                                              int32_t* __restrict__ n_samples, float* l2x, float* l2y, float* l2z, int32_t* seg, int cap) {
    const int lane = threadIdx.x;
    // own points (their loads are in flight while the neighbours are scanned): slot s of a lane is point k2 = lane + 64 s (ascending k2 per lane, as the serial scan)
    int me[4];
    float fx[4], fy[4], fz[4], md[4];
    bool live[4];                         // label 1: an own point not sampled yet
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int k2 = lane + 64 * s;
        me[s] = -1; fx[s] = fy[s] = fz[s] = 0.f; md[s] = 0.f; live[s] = false;
        if (k2 < m) {
            me[s] = members[lo + k2];
            fx[s] = (float)P[3 * (int64_t)me[s]]; fy[s] = (float)P[3 * (int64_t)me[s] + 1]; fz[s] = (float)P[3 * (int64_t)me[s] + 2];
            live[s] = true;
        }
    }
    // the neighbours' member ranges, all at once: lane q holds box q's range; seg[q] = first flattened index of box q
    int s0 = 0, cnt = 0;
    if (lane < nn) { s0 = start[nbr[lane]]; cnt = start[nbr[lane] + 1] - s0; }
    int incl = cnt;
#pragma unroll
    for (int off = 1; off < 32; off <<= 1) {
        const int v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
    }
    if (lane < nn) seg[lane] = incl - cnt;
    const int total = __shfl(incl, 31, 64);
    if (lane == 0) seg[27] = 0;
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    int n_l2 = 0;
    for (int f0 = 0; f0 < total; f0 += 64) {
        const int f = f0 + lane;
        bool hit = false;
        float hx = 0.f, hy = 0.f, hz = 0.f;
        // the box whose range holds flattened index f (ranges in box order; the search lands behind empty boxes); the shuffle
        // runs with every lane active: a permute reads nothing from a lane that is masked off
        int q = 0;
        {
            const int fc = f < total ? f : 0;
#pragma unroll
            for (int step = 16; step > 0; step >>= 1)
                if (q + step < nn && seg[q + step] <= fc) q += step;
        }
        const int bs = __shfl(s0, q, 64);
        if (f < total) {
            const int pi = members[bs + (f - seg[q])];
            const double x = P[3 * (int64_t)pi], y = P[3 * (int64_t)pi + 1], z = P[3 * (int64_t)pi + 2];
            hit = x <= pc[0] + R && x >= pc[0] - R && y <= pc[1] + R && y >= pc[1] - R && z <= pc[2] + R && z >= pc[2] - R && labelG[pi] == 0;
            hx = (float)x; hy = (float)y; hz = (float)z;
        }
        const unsigned long long mask = __ballot(hit);
        if (hit) {
            const int slot = n_l2 + __popcll(mask & ((1ull << lane) - 1ull));
            if (slot < cap) { l2x[slot] = hx; l2y[slot] = hy; l2z[slot] = hz; }
        }
        n_l2 += __popcll(mask);
    }
    if (n_l2 > cap) return false;
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    const bool any_l2 = n_l2 > 0;
    const int cpos = center_pos[b];
    const int seed_pos = (!any_l2 && cpos >= 0 && cpos < m) ? cpos : -1;   // addJ: seed the box centre only if no neighbour sample
    auto dist = [](float ax, float ay, float az, float bx, float by, float bz) {
        const float dx = ax - bx, dy = ay - by, dz = az - bz;
        return sqrtf((dx * dx + dy * dy) + dz * dz);
    };
    // the seed's coordinates (uniform): lane seed_pos & 63, slot seed_pos >> 6
    float sx = 0.f, sy = 0.f, sz = 0.f;
    int seed_pt = -1;
    if (seed_pos >= 0) {
        const int sl = seed_pos & 63, ss = seed_pos >> 6;
        const float cx = ss == 0 ? fx[0] : ss == 1 ? fx[1] : ss == 2 ? fx[2] : fx[3];
        const float cy = ss == 0 ? fy[0] : ss == 1 ? fy[1] : ss == 2 ? fy[2] : fy[3];
        const float cz = ss == 0 ? fz[0] : ss == 1 ? fz[1] : ss == 2 ? fz[2] : fz[3];
        const int cm = ss == 0 ? me[0] : ss == 1 ? me[1] : ss == 2 ? me[2] : me[3];
        sx = __shfl(cx, sl, 64); sy = __shfl(cy, sl, 64); sz = __shfl(cz, sl, 64); seed_pt = __shfl(cm, sl, 64);
    }
    // initial min-distance of every own point to the seed / the sampled neighbours (9999 with neither)
    // (one LDS read of a neighbour per FOUR evaluations; a minimum does not depend on the order of its arguments)
    {
        float best[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) best[s] = seed_pt >= 0 ? dist(fx[s], fy[s], fz[s], sx, sy, sz) : __builtin_inff();
#pragma unroll 4
        for (int l = 0; l < n_l2; ++l) {
            const float qx = l2x[l], qy = l2y[l], qz = l2z[l];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float d = dist(fx[s], fy[s], fz[s], qx, qy, qz);
                if (d < best[s]) best[s] = d;
            }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int k2 = lane + 64 * s;
            if (k2 >= m) continue;
            md[s] = k2 == seed_pos ? 0.f : (best[s] == __builtin_inff() ? 9999.0f : best[s]);
        }
    }
    int sample_index = 0;
    if (seed_pos >= 0) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
            if (lane + 64 * s == seed_pos) { simiT[lo] = seed_pt; labelG[seed_pt] = 0; live[s] = false; }
        sample_index = 1;
    }
    // farthest-point loop: first maximum of the min-distances over the still-unsampled own points (serial scan order)
    while (sample_index < simNum) {
        float bv = 0.f;
        int bk2 = 0x7fffffff;
#pragma unroll
        for (int s = 0; s < 4; ++s)
            if (live[s] && md[s] > bv) { bv = md[s]; bk2 = lane + 64 * s; }   // ascending k2 per lane: first maximum kept
#pragma unroll
        for (int off = 32; off >= 16; off >>= 1) {
            const float ov = __shfl_xor(bv, off, 64);
            const int ok = __shfl_xor(bk2, off, 64);
            if (ov > bv || (ov == bv && ok < bk2)) { bv = ov; bk2 = ok; }
        }
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) {
            const float ov = __int_as_float(aivs_dpp_xor(__float_as_int(bv), off));
            const int ok = aivs_dpp_xor(bk2, off);
            if (ov > bv || (ov == bv && ok < bk2)) { bv = ov; bk2 = ok; }
        }
        bk2 = __builtin_amdgcn_readfirstlane(bk2);
        if (bk2 == 0x7fffffff) break;   // nothing left (indexSelect == -1)
        const int sl = bk2 & 63, ss = bk2 >> 6;
        const float cx = ss == 0 ? fx[0] : ss == 1 ? fx[1] : ss == 2 ? fx[2] : fx[3];
        const float cy = ss == 0 ? fy[0] : ss == 1 ? fy[1] : ss == 2 ? fy[2] : fy[3];
        const float cz = ss == 0 ? fz[0] : ss == 1 ? fz[1] : ss == 2 ? fz[2] : fz[3];
        const int cm = ss == 0 ? me[0] : ss == 1 ? me[1] : ss == 2 ? me[2] : me[3];
        const float qx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cx), sl)), qy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cy), sl)),
                    qz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cz), sl));
        const int sel = __builtin_amdgcn_readlane(cm, sl);
        if (lane == 0) { labelG[sel] = 0; simiT[lo + sample_index] = sel; }
        ++sample_index;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int k2 = lane + 64 * s;
            if (k2 == bk2) { md[s] = 0.f; live[s] = false; }
            else if (live[s]) {
                const float d = dist(fx[s], fy[s], fz[s], qx, qy, qz);
                if (d < md[s]) md[s] = d;
            }
        }
    }
    if (lane == 0) n_samples[b] = sample_index;
    return true;
}

// ---- AIVS_Voroni_OpenMP_KNN for one box: one wave ---------------------------------------------------------------------
__device__ __forceinline__ void aivs_fps_box(const double* __restrict__ P, const AivsGrid& g, int b, const int32_t* __restrict__ start,
                                             const int32_t* __restrict__ members, const int32_t* __restrict__ center_pos,
                                             const int32_t* __restrict__ sim_num, uint8_t* __restrict__ labelG, double* __restrict__ mind,
                                             int32_t* __restrict__ simiT, int32_t* __restrict__ n_samples) {
    const int lo = start[b], m = start[b + 1] - lo;
    const int simNum = sim_num[b];
    if (m == 0 || simNum == 0) return;
    const int lane = threadIdx.x;
    double pc[3];
    aivs_box_center(g, b, pc);
    const double R = g.search_radius;
    int nbr[26];
    const int nn = aivs_neighbor_boxes(g, b, nbr);

    // already-sampled points of the neighbour boxes inside the search cube ("label 2"): the wave scans the
    // neighbours' members once, lane-strided, and compacts the hits into LDS (ballot + prefix popcount); their
    // order is irrelevant (only a minimum over them is taken).  More than L2_CAP hits fall back to re-scanning.
    constexpr int L2_CAP = 512;
    __shared__ int32_t l2[L2_CAP];
    // The usual box -- up to 256 own points, up to L2_CAP sampled neighbours in reach -- runs entirely out of registers and
    // LDS (aivs_fps_fast below: same arithmetic, same selection order); anything larger takes the general path after it.
    __shared__ float l2x[L2_CAP], l2y[L2_CAP], l2z[L2_CAP];
    __shared__ int32_t seg[28];
    if (m <= 4 * 64 && aivs_fps_fast(P, g, b, lo, m, simNum, pc, R, nbr, nn, start, members, center_pos, labelG, simiT, n_samples, l2x, l2y, l2z, seg, L2_CAP))
        return;
    int n_l2 = 0;
    for (int q = 0; q < nn; ++q) {
        const int s0 = start[nbr[q]], s1 = start[nbr[q] + 1];
        for (int l0 = s0; l0 < s1; l0 += 64) {
            const int l = l0 + lane;
            int pi = -1;
            bool hit = false;
            if (l < s1) {
                pi = members[l];
                const double x = P[3 * (int64_t)pi], y = P[3 * (int64_t)pi + 1], z = P[3 * (int64_t)pi + 2];
                hit = x <= pc[0] + R && x >= pc[0] - R && y <= pc[1] + R && y >= pc[1] - R && z <= pc[2] + R && z >= pc[2] - R && labelG[pi] == 0;
            }
            const unsigned long long mask = __ballot(hit);
            if (hit) {
                const int slot = n_l2 + __popcll(mask & ((1ull << lane) - 1ull));
                if (slot < L2_CAP) l2[slot] = pi;
            }
            n_l2 += __popcll(mask);
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    const bool any_l2 = n_l2 > 0;
    const int cpos = center_pos[b];
    const int seed_pos = (!any_l2 && cpos >= 0 && cpos < m) ? cpos : -1;   // addJ: seed the box centre only if no neighbour sample
    const int seed_pt = seed_pos >= 0 ? members[lo + seed_pos] : -1;

    // initial min-distance of every own point to the label-0/2 set
    for (int k2 = lane; k2 < m; k2 += 64) {
        const int me = members[lo + k2];
        double mt;
        if (k2 == seed_pos) {
            mt = 0;
        } else {
            float best = __builtin_inff();
            if (seed_pt >= 0) best = aivs_dist(P, me, seed_pt);
            if (any_l2 && n_l2 <= L2_CAP) {
                for (int l = 0; l < n_l2; ++l) {
                    const float d = aivs_dist(P, me, l2[l]);
                    if (d < best) best = d;
                }
            } else if (any_l2)
                for (int q = 0; q < nn; ++q) {
                    const int s0 = start[nbr[q]], s1 = start[nbr[q] + 1];
                    for (int l = s0; l < s1; ++l) {
                        const int pi = members[l];
                        const double x = P[3 * (int64_t)pi], y = P[3 * (int64_t)pi + 1], z = P[3 * (int64_t)pi + 2];
                        if (x <= pc[0] + R && x >= pc[0] - R && y <= pc[1] + R && y >= pc[1] - R && z <= pc[2] + R && z >= pc[2] - R &&
                            labelG[pi] == 0) {
                            const float d = aivs_dist(P, me, pi);
                            if (d < best) best = d;
                        }
                    }
                }
            mt = best == __builtin_inff() ? 9999.0 : (double)best;
        }
        mind[lo + k2] = mt;
    }
    // Every own point k2 is read and written ONLY by its owner lane (k2 % 64): no lane ever depends on another
    // lane's global stores; selections travel through shuffles.
    int sample_index = 0;
    if (seed_pos >= 0) {
        if (lane == (seed_pos & 63)) { simiT[lo] = seed_pt; labelG[seed_pt] = 0; }
        sample_index = 1;
    }
    // farthest-point loop: first maximum of mind over the still-unsampled own points (serial scan order)
    while (sample_index < simNum) {
        double bv = 0;
        int bk2 = 0x7fffffff;
        for (int k2 = lane; k2 < m; k2 += 64) {
            const int me = members[lo + k2];
            const double v = mind[lo + k2];
            if (labelG[me] == 1 && v > bv) { bv = v; bk2 = k2; }   // ascending k2 per lane: first maximum kept
        }
        for (int off = 32; off > 0; off >>= 1) {
            const double ov = __shfl_xor(bv, off, 64);
            const int ok = __shfl_xor(bk2, off, 64);
            if (ov > bv || (ov == bv && ok < bk2)) { bv = ov; bk2 = ok; }
        }
        if (bk2 == 0x7fffffff) break;   // nothing left (indexSelect == -1)
        const int sel = members[lo + bk2];
        if (lane == (bk2 & 63)) { mind[lo + bk2] = 0; labelG[sel] = 0; simiT[lo + sample_index] = sel; }
        ++sample_index;
        for (int k2 = lane; k2 < m; k2 += 64) {
            const int me = members[lo + k2];
            if (k2 != bk2 && labelG[me] == 1) {
                const double d = (double)aivs_dist(P, me, sel);
                if (d < mind[lo + k2]) mind[lo + k2] = d;
            }
        }
    }
    if (lane == 0) n_samples[b] = sample_index;
}

// one colour per launch.  (All eight colours in ONE launch -- a box waiting on per-box flags for its neighbours of earlier
// colours -- was built in round 3 and removed: 243 us against 8 x 21, and 4.7 against 3.0 ms at 100k points: the waiting waves
// hold the slots the boxes they wait for need, and spin on the L2.)
__global__ __launch_bounds__(64) void aivs_fps_kernel(const double* __restrict__ P, AivsGrid g, int ox, int oy, int oz,
                                                      const int32_t* __restrict__ start, const int32_t* __restrict__ members,
                                                      const int32_t* __restrict__ center_pos, const int32_t* __restrict__ sim_num,
                                                      uint8_t* __restrict__ labelG, double* __restrict__ mind,
                                                      int32_t* __restrict__ simiT, int32_t* __restrict__ n_samples) {
    // blockIdx -> the box (i, j, k) of this colour class: i = 2*bx + ox (1-based, ox in {1, 2}) etc.
    const int hx = (g.nx - ox) / 2 + 1, hy = (g.ny - oy) / 2 + 1;
    const int bi = blockIdx.x % hx, bj = (blockIdx.x / hx) % hy, bk = blockIdx.x / (hx * hy);
    const int i = 2 * bi + ox, j = 2 * bj + oy, k = 2 * bk + oz;
    if (i > g.nx || j > g.ny || k > g.nz) return;
    aivs_fps_box(P, g, i + g.nx * (j - 1) + g.nx * g.ny * (k - 1), start, members, center_pos, sim_num, labelG, mind, simiT, n_samples);
}

// ---- accurate cut: K = 3 self-kNN among the samples (brute force; ties -> lower index) -----------------------------
__device__ __forceinline__ void top3_insert(float d, int j, float (&bd)[3], int (&bi)[3]) {
    // ascending by (distance, index): equal distances keep the lower index first
    if (d < bd[2] || (d == bd[2] && j < bi[2])) {
        int pos = 2;
        while (pos > 0 && (d < bd[pos - 1] || (d == bd[pos - 1] && j < bi[pos - 1]))) { bd[pos] = bd[pos - 1]; bi[pos] = bi[pos - 1]; --pos; }
        bd[pos] = d; bi[pos] = j;
    }
}

// one wave per sample: lanes stride over the samples keeping their own top-3, lane 0 merges the 64 x 3 candidates
__global__ __launch_bounds__(64) void aivs_knn3_kernel(const double* __restrict__ P, const int32_t* __restrict__ samples, int ns,
                                                       int32_t* __restrict__ nn1, float* __restrict__ d1, float* __restrict__ d2) {
    __shared__ float sd[64][3];
    __shared__ int si[64][3];
    const int i = blockIdx.x, lane = threadIdx.x;
    const int pi = samples[i];
    const float ax = (float)P[3 * (int64_t)pi], ay = (float)P[3 * (int64_t)pi + 1], az = (float)P[3 * (int64_t)pi + 2];
    float bd[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
    int bi[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff};
    for (int j = lane; j < ns; j += 64) {
        const int pj = samples[j];
        const float dx = ax - (float)P[3 * (int64_t)pj], dy = ay - (float)P[3 * (int64_t)pj + 1], dz = az - (float)P[3 * (int64_t)pj + 2];
        top3_insert((dx * dx + dy * dy) + dz * dz, j, bd, bi);
    }
    for (int k = 0; k < 3; ++k) { sd[lane][k] = bd[k]; si[lane][k] = bi[k]; }
    __syncthreads();
    if (lane == 0) {
        float md[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
        int mi[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff};
        for (int l = 0; l < 64; ++l)
            for (int k = 0; k < 3; ++k)
                if (si[l][k] != 0x7fffffff) top3_insert(sd[l][k], si[l][k], md, mi);
        nn1[i] = mi[1];
        d1[i] = sqrtf(md[1]);
        d2[i] = sqrtf(md[2]);
    }
}

// The same from a packed copy of the samples' float coordinates (aivs_sample_coords_kernel): a step of the walk is three
// coalesced loads that depend on nothing, four steps in flight -- the kernel above goes through samples[j] and then P[]: two
// dependent round trips per step, 1.5 us each -- and the 64 x 3 candidates of a wave are merged by three rounds of a
// wave-wide minimum on (distance, index) instead of one lane inserting 192 entries.  The three nearest under a strict total
// order do not depend on the order of evaluation: same results.
__global__ __launch_bounds__(256) void aivs_sample_coords_kernel(const double* __restrict__ P, const int32_t* __restrict__ samples, int ns, float* __restrict__ co) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ns) return;
    const int pj = samples[j];
    co[j] = (float)P[3 * (int64_t)pj]; co[ns + j] = (float)P[3 * (int64_t)pj + 1]; co[2 * ns + j] = (float)P[3 * (int64_t)pj + 2];
}
__global__ __launch_bounds__(256) void aivs_knn3_packed_kernel(const float* __restrict__ co, int ns, int32_t* __restrict__ nn1, float* __restrict__ d1, float* __restrict__ d2) {
    const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= ns) return;   // uniform per wave
    const float* __restrict__ sx = co;
    const float* __restrict__ sy = co + ns;
    const float* __restrict__ sz = co + 2 * ns;
    const float ax = sx[i], ay = sy[i], az = sz[i];
    float bd[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
    int bi[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff};
    for (int j0 = lane; j0 < ns; j0 += 256) {
        float x[4], y[4], z[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = min(j0 + 64 * u, ns - 1);
            x[u] = sx[j]; y[u] = sy[j]; z[u] = sz[j];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + 64 * u;
            if (j < ns) {
                const float dx = ax - x[u], dy = ay - y[u], dz = az - z[u];
                top3_insert((dx * dx + dy * dy) + dz * dz, j, bd, bi);
            }
        }
    }
    // three rounds: the smallest head of the 64 sorted lists, popped from the list it came from
    float md[3];
    int mi[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        float v = bd[0];
        int k = bi[0];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(v, off, 64);
            const int ok = __shfl_xor(k, off, 64);
            if (ov < v || (ov == v && ok < k)) { v = ov; k = ok; }
        }
        md[r] = v; mi[r] = k;
        if (bi[0] == k && k != 0x7fffffff) { bd[0] = bd[1]; bi[0] = bi[1]; bd[1] = bd[2]; bi[1] = bi[2]; bd[2] = __builtin_inff(); bi[2] = 0x7fffffff; }
    }
    if (lane == 0) {
        nn1[i] = mi[1];
        d1[i] = sqrtf(md[1]);
        d2[i] = sqrtf(md[2]);
    }
}

// exclusive scan of small int arrays on one workgroup (boxes <= ~125k + 2): used for box starts and sample offsets
__global__ __launch_bounds__(1024) void aivs_scan_kernel(const int32_t* __restrict__ in, int n, int32_t* __restrict__ out) {
    __shared__ int sh[1024];
    const int per = (n + 1023) / 1024;
    const int lo = threadIdx.x * per, hi = min(n, lo + per);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += in[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = threadIdx.x >= off ? sh[threadIdx.x - off] : 0;
        __syncthreads();
        sh[threadIdx.x] += v;
        __syncthreads();
    }
    int run = threadIdx.x > 0 ? sh[threadIdx.x - 1] : 0;
    for (int i = lo; i < hi; ++i) { const int v = in[i]; out[i] = run; run += v; }
    if (threadIdx.x == 1023) out[n] = sh[1023];
}

__global__ __launch_bounds__(256) void aivs_gather_samples_kernel(const int32_t* __restrict__ start, const int32_t* __restrict__ n_samples,
                                                                  const int32_t* __restrict__ sample_off, const int32_t* __restrict__ simiT,
                                                                  int nboxes1, int32_t* __restrict__ samples) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nboxes1) return;
    const int k = n_samples[b], o = sample_off[b], lo = start[b];
    for (int t = 0; t < k; ++t) samples[o + t] = simiT[lo + t];
}

#define AIVS_HIP(call)                                                                                   \
    do {                                                                                                 \
        hipError_t e_ = (call);                                                                          \
        if (e_ != hipSuccess) { err = std::string(#call) + ": " + hipGetErrorString(e_); rc = KSS_ERR_HIP; goto done; } \
    } while (0)

// Host driver.  d_xyz: n packed f64 points on the device.  out_idx: indices of the selected points in the
// reference's output order.  `scratch(bytes)` returns ONE device region (the context's grow-only workspace) that is
// carved into the working arrays: no hipMalloc/hipFree on the call path.
int aivs_device(hipStream_t st, const double* d_xyz, int n, int point_num, std::vector<int32_t>& out_idx, std::string& err,
                const std::function<void*(size_t)>& scratch) {
    int rc = KSS_OK;
    out_idx.clear();
    double* d_bbox = nullptr;
    int32_t *d_box_of = nullptr, *d_counts = nullptr, *d_start = nullptr, *d_cursor = nullptr, *d_tmp = nullptr, *d_members = nullptr,
            *d_center = nullptr, *d_sim = nullptr, *d_simiT = nullptr, *d_nsamp = nullptr, *d_soff = nullptr, *d_samples = nullptr,
            *d_nn1 = nullptr, *d_bad = nullptr;
    uint8_t* d_label = nullptr;
    double* d_mind = nullptr;
    float *d_d1 = nullptr, *d_d2 = nullptr, *d_sco = nullptr;
    std::vector<double> hb(64 * 6);
    std::vector<int32_t> samples;
    AivsGrid g;
    int nb1 = 0, ns = 0;
    char* pool = nullptr;
    size_t pool_off = 0;
    auto carve = [&](size_t bytes) -> void* {
        void* p = pool + pool_off;
        pool_off += (bytes + 255) & ~(size_t)255;
        return p;
    };
    {
        // the bounding box needs 3 KB before the grid size is known: it lives at the start of a first, small region
        pool = (char*)scratch(64 * 6 * sizeof(double));
        if (!pool) { err = "aivs: out of device memory"; rc = KSS_ERR_NOMEM; goto done; }
        d_bbox = (double*)pool;
        hipLaunchKernelGGL(aivs_bbox_kernel, dim3(64), dim3(256), 0, st, d_xyz, n, d_bbox);
        AIVS_HIP(hipMemcpyAsync(hb.data(), d_bbox, hb.size() * sizeof(double), hipMemcpyDeviceToHost, st));
        AIVS_HIP(hipStreamSynchronize(st));
        double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int b = 0; b < 64; ++b)
            for (int k = 0; k < 3; ++k) { mn[k] = std::min(mn[k], hb[b * 6 + k]); mx[k] = std::max(mx[k], hb[b * 6 + 3 + k]); }
        // BallRegion_EstimateBoxScale :1194-1215 and BallRegion_AchieveXYZ :690-758
        int boxNum;
        if (n < 10000) boxNum = 10;
        else if (n < 50000) boxNum = 20;
        else if (n < 100000) boxNum = 30;
        else if (n < 500000) boxNum = 40;
        else if (n < 1000000) boxNum = 50;
        else boxNum = (int)std::pow((double)n / 8.0, 1.0 / 3.0);
        g.minx = mn[0]; g.miny = mn[1]; g.minz = mn[2];
        const double x_dis = std::fabs(mx[0] - mn[0]), y_dis = std::fabs(mx[1] - mn[1]), z_dis = std::fabs(mx[2] - mn[2]);
        double large = x_dis;
        if (large < y_dis) large = y_dis;
        if (large < z_dis) large = z_dis;
        g.unit = large / (double)boxNum;
        if (!(g.unit > 0) || !std::isfinite(g.unit)) { err = "aivs: degenerate cloud (zero extent)"; rc = KSS_ERR_ARG; goto done; }
        const double numX = x_dis / g.unit, numY = y_dis / g.unit, numZ = z_dis / g.unit;
        g.nx = (int)numX; g.ny = (int)numY; g.nz = (int)numZ;
        if (numX > (double)g.nx) g.nx++;
        if (numY > (double)g.ny) g.ny++;
        if (numZ > (double)g.nz) g.nz++;
        if (g.nx < 1 || g.ny < 1 || g.nz < 1) { err = "aivs: planar cloud (the reference divides into zero boxes)"; rc = KSS_ERR_ARG; goto done; }
        g.nboxes = g.nx * g.ny * g.nz;
        g.rate = (double)point_num / (double)n;
        g.search_radius = g.unit * 3.0 / 4.0;
        nb1 = g.nboxes + 1;
    }
    {
        const size_t N = (size_t)n, B = (size_t)nb1 + 2;
        const size_t total = 32 * 256 /* every region is rounded up to 256 bytes */ + (N * 4) * 5 + (B * 4) * 7 + N + N * 8 + 256 /*bad*/ + (N * 4) * 2 /*samples, nn1*/ + (N * 4) * 2 /*d1, d2*/ + N * 12 /*sample coordinates*/;
        pool = (char*)scratch(total);   // the earlier small region is dead: its contents were copied to the host already
        pool_off = 0;
        if (!pool) { err = "aivs: out of device memory"; rc = KSS_ERR_NOMEM; goto done; }
        d_box_of = (int32_t*)carve(N * 4); d_tmp = (int32_t*)carve(N * 4); d_members = (int32_t*)carve(N * 4);
        d_simiT = (int32_t*)carve(N * 4); d_samples = (int32_t*)carve(N * 4); d_sco = (float*)carve(N * 12);
        d_counts = (int32_t*)carve(B * 4); d_start = (int32_t*)carve(B * 4); d_cursor = (int32_t*)carve(B * 4);
        d_center = (int32_t*)carve(B * 4); d_sim = (int32_t*)carve(B * 4); d_nsamp = (int32_t*)carve(B * 4); d_soff = (int32_t*)carve(B * 4);
        d_label = (uint8_t*)carve(N); d_mind = (double*)carve(N * 8); d_bad = (int32_t*)carve(256);
        d_nn1 = (int32_t*)carve(N * 4); d_d1 = (float*)carve(N * 4); d_d2 = (float*)carve(N * 4);
    }
    AIVS_HIP(hipMemsetAsync(d_counts, 0, sizeof(int32_t) * ((size_t)nb1 + 2), st));
    AIVS_HIP(hipMemsetAsync(d_nsamp, 0, sizeof(int32_t) * ((size_t)nb1 + 2), st));
    AIVS_HIP(hipMemsetAsync(d_bad, 0, sizeof(int32_t), st));
    AIVS_HIP(hipMemsetAsync(d_label, 1, (size_t)n, st));
    {
        const dim3 gp((n + 255) / 256), bp(256);
        hipLaunchKernelGGL(aivs_count_kernel, gp, bp, 0, st, d_xyz, n, g, d_box_of, d_counts, d_bad);
        hipLaunchKernelGGL(aivs_scan_kernel, dim3(1), dim3(1024), 0, st, d_counts, nb1, d_start);
        AIVS_HIP(hipMemcpyAsync(d_cursor, d_start, sizeof(int32_t) * (size_t)nb1, hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL(aivs_scatter_kernel, gp, bp, 0, st, d_box_of, n, d_cursor, d_tmp);
        hipLaunchKernelGGL(aivs_rank_kernel, gp, bp, 0, st, d_tmp, d_box_of, n, d_start, d_members);
        hipLaunchKernelGGL(aivs_box_info_kernel, dim3((nb1 + 255) / 256), bp, 0, st, d_xyz, g, d_start, d_members, d_center, d_sim);
        // the 8 colours in the reference's order (Method_AIVS_SimPro.hpp:609-632); parity 1 -> offset 1, parity 0 -> offset 2
        const int order[8][3] = {{1, 1, 1}, {0, 1, 1}, {0, 0, 1}, {1, 0, 1}, {1, 1, 0}, {0, 1, 0}, {0, 0, 0}, {1, 0, 0}};
        for (int c = 0; c < 8; ++c) {
            const int ox = order[c][0] ? 1 : 2, oy = order[c][1] ? 1 : 2, oz = order[c][2] ? 1 : 2;
            if (ox > g.nx || oy > g.ny || oz > g.nz) continue;
            const int hx = (g.nx - ox) / 2 + 1, hy = (g.ny - oy) / 2 + 1, hz = (g.nz - oz) / 2 + 1;
            hipLaunchKernelGGL(aivs_fps_kernel, dim3(hx * hy * hz), dim3(64), 0, st, d_xyz, g, ox, oy, oz, d_start, d_members, d_center,
                               d_sim, d_label, d_mind, d_simiT, d_nsamp);
        }
        hipLaunchKernelGGL(aivs_scan_kernel, dim3(1), dim3(1024), 0, st, d_nsamp, nb1, d_soff);
        AIVS_HIP(hipGetLastError());
        int32_t h_total = 0, h_bad = 0;
        AIVS_HIP(hipMemcpyAsync(&h_total, d_soff + nb1, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        AIVS_HIP(hipMemcpyAsync(&h_bad, d_bad, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        AIVS_HIP(hipStreamSynchronize(st));
        if (h_bad) { err = "aivs: a point fell outside the box grid"; rc = KSS_ERR_ARG; goto done; }
        ns = h_total;
        if (ns <= 0) goto done;
        hipLaunchKernelGGL(aivs_gather_samples_kernel, dim3((nb1 + 255) / 256), bp, 0, st, d_start, d_nsamp, d_soff, d_simiT, nb1, d_samples);
        samples.resize((size_t)ns);
        AIVS_HIP(hipMemcpyAsync(samples.data(), d_samples, sizeof(int32_t) * (size_t)ns, hipMemcpyDeviceToHost, st));
    }
    {
        std::vector<uint8_t> alive((size_t)ns, 1);
        int64_t dTiff = (int64_t)ns - point_num;
        if (dTiff > 0 && ns >= 3) {
            // AIVS_AccurateCut_Optimization :848-957: greedy removal of one end of the closest live pair (host: O(dTiff * ns))
            hipLaunchKernelGGL(aivs_sample_coords_kernel, dim3((ns + 255) / 256), dim3(256), 0, st, d_xyz, d_samples, ns, d_sco);
            hipLaunchKernelGGL(aivs_knn3_packed_kernel, dim3((ns + 3) / 4), dim3(256), 0, st, (const float*)d_sco, ns, d_nn1, d_d1, d_d2);
            std::vector<int32_t> nn1((size_t)ns);
            std::vector<float> d1((size_t)ns), d2((size_t)ns);
            AIVS_HIP(hipMemcpyAsync(nn1.data(), d_nn1, sizeof(int32_t) * (size_t)ns, hipMemcpyDeviceToHost, st));
            AIVS_HIP(hipMemcpyAsync(d1.data(), d_d1, sizeof(float) * (size_t)ns, hipMemcpyDeviceToHost, st));
            AIVS_HIP(hipMemcpyAsync(d2.data(), d_d2, sizeof(float) * (size_t)ns, hipMemcpyDeviceToHost, st));
            AIVS_HIP(hipStreamSynchronize(st));
            while (dTiff > 0) {
                double mn = 9999;
                int64_t b1 = -1, b2 = -1;
                for (int64_t i = 0; i < ns; ++i) {
                    const int64_t b2t = nn1[(size_t)i];
                    const double dt = d1[(size_t)i];
                    if (dt < mn && alive[(size_t)i] && alive[(size_t)b2t]) { mn = dt; b1 = i; b2 = b2t; }
                }
                if (mn == 9999 || b1 == -1 || b2 == -1) break;
                int64_t del = b1;
                if ((double)d2[(size_t)b1] > (double)d2[(size_t)b2]) del = b2;
                alive[(size_t)del] = 0;
                --dTiff;
            }
        } else {
            AIVS_HIP(hipStreamSynchronize(st));
        }
        for (int i = 0; i < ns; ++i)
            if (alive[(size_t)i]) out_idx.push_back(samples[(size_t)i]);
    }
done:
    hipStreamSynchronize(st);
    return rc;
}

}  // namespace kss
