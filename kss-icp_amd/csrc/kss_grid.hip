// kss_grid.hip -- exact nearest neighbour through a uniform cell list (the HBM/L2-bound variant of (a8)).
//
// "a uniform-grid-binned NN (cell list in HBM) is the variant that becomes
// genuinely HBM-bound; keep the brute-force kernel as the exact reference and measure both".
// The result is BIT-IDENTICAL to nn_sweep_kernel (same f32 no-fma distance, ties -> lowest index):
//   build   : bbox -> per-cell counts (atomics) -> exclusive scan -> scatter into cell order.  The order
//             inside a cell is whatever the atomics give; the query's explicit (d2, idx) tie rule makes
//             the answer independent of it.
//   query   : one lane per source point.  Shells r = 0, 1, 2, ... of cells around the query's cell are
//             visited row by row (cells adjacent in x are contiguous in the sorted array, so a row of
//             2r+1 cells is ONE range).  After shell r every unvisited point is farther than the
//             distance b(r) from the query to the faces of the visited block; the search stops when
//             best_d2 < (b(r) - eps)^2 * (1 - 1e-6), a bound that is conservative w.r.t. the f32 rounding
//             of both the cell assignment and the distance, so it can only search MORE than needed.
//   warm start: inside an ICP every source remembers where its previous winner sits; that point is evaluated
//             first and its distance prunes the cells of the first block that cannot hold the winner nor a tie
//             (block_walk); the result does not depend on it.
//   fallback: a query not resolved within GridParams::rcap shells is appended to a list and resolved by the
//             brute-force sweep (nn_sweep_kernel<LIST>) -- far-away clouds never degrade below it.  (The batched
//             kernel sweeps the pair's targets inside the wave instead: gridb_nn_kernel.)
//   fusion  : the correspondence sums are accumulated by the searching lanes (the winner's coordinates are still in
//             registers) and reduced in the same launch; see grid_nn_kernel / gridb_nn_kernel.
#pragma clang fp contract(off)

#include <hip/hip_runtime.h>

#include <cstdlib>

#include "kss_internal.hpp"
#include "kss_device.hpp"

namespace kss {

// ---- bbox of the real (non-sentinel) targets ---------------------------------------------------------
__global__ __launch_bounds__(256) void grid_bbox_kernel(const float4* __restrict__ tgt, int n, float* __restrict__ partial) {
    __shared__ float sh[4][6];
    float mn[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
    float mx[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float4 p = tgt[i];
        mn[0] = fminf(mn[0], p.x); mn[1] = fminf(mn[1], p.y); mn[2] = fminf(mn[2], p.z);
        mx[0] = fmaxf(mx[0], p.x); mx[1] = fmaxf(mx[1], p.y); mx[2] = fmaxf(mx[2], p.z);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            mn[k] = fminf(mn[k], __shfl_down(mn[k], off, 64));
            mx[k] = fmaxf(mx[k], __shfl_down(mx[k], off, 64));
        }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0)
        for (int k = 0; k < 3; ++k) { sh[wave][k] = mn[k]; sh[wave][3 + k] = mx[k]; }
    __syncthreads();
    if (threadIdx.x < 6) {
        float v = sh[0][threadIdx.x];
        for (int w = 1; w < 4; ++w) v = threadIdx.x < 3 ? fminf(v, sh[w][threadIdx.x]) : fmaxf(v, sh[w][threadIdx.x]);
        partial[blockIdx.x * 6 + threadIdx.x] = v;
    }
}

void launch_grid_bbox(hipStream_t st, const float4* d_tgt, int n, float* d_partial, int n_blocks) {
    hipLaunchKernelGGL(grid_bbox_kernel, dim3(n_blocks), dim3(256), 0, st, d_tgt, n, d_partial);
}

__device__ __forceinline__ int cell_coord(float v, float o, float inv_h, int g) {
    int c = (int)floorf((v - o) * inv_h);
    c = c < 0 ? 0 : c;
    return c >= g ? g - 1 : c;
}

// ---- counting sort into cell order ---------------------------------------------------------------------
__global__ __launch_bounds__(256) void grid_count_kernel(const float4* __restrict__ tgt, int n, GridParams gp,
                                                         int32_t* __restrict__ counts) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 p = tgt[i];
    const int cx = cell_coord(p.x, gp.ox, gp.inv_h, gp.gx), cy = cell_coord(p.y, gp.oy, gp.inv_h, gp.gy),
              cz = cell_coord(p.z, gp.oz, gp.inv_h, gp.gz);
    atomicAdd(&counts[(cz * gp.gy + cy) * gp.gx + cx], 1);
}

// exclusive scan, 3 phases; each workgroup owns SCAN_CHUNK consecutive elements
constexpr int SCAN_CHUNK = 4096;   // 256 threads x 16

__global__ __launch_bounds__(256) void scan_block_sums_kernel(const int32_t* __restrict__ in, int n, int32_t* __restrict__ block_sums) {
    __shared__ int sh[4];
    const int base = blockIdx.x * SCAN_CHUNK;
    int s = 0;
    for (int k = threadIdx.x; k < SCAN_CHUNK; k += 256) {
        const int i = base + k;
        if (i < n) s += in[i];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

// single workgroup: exclusive scan of up to 1024*16 block sums, in place
__global__ __launch_bounds__(1024) void scan_of_block_sums_kernel(int32_t* __restrict__ block_sums, int nb) {
    __shared__ int sh[1024];
    const int per = (nb + 1023) / 1024;
    const int lo = threadIdx.x * per, hi = min(nb, lo + per);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += block_sums[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {   // Hillis-Steele inclusive scan over the 1024 partials
        int v = threadIdx.x >= off ? sh[threadIdx.x - off] : 0;
        __syncthreads();
        sh[threadIdx.x] += v;
        __syncthreads();
    }
    int run = threadIdx.x > 0 ? sh[threadIdx.x - 1] : 0;
    for (int i = lo; i < hi; ++i) {
        const int v = block_sums[i];
        block_sums[i] = run;
        run += v;
    }
}

// SELF = true (up to 1024 chunks): the workgroup adds up the chunk totals before its own by itself and the last element
// also writes start[n] -- two launches (scan_of_block_sums, scan_tail) less per scan, which matters when a whole cell-list
// build is ~20 launches of a few microseconds each.
template <bool SELF>
__global__ __launch_bounds__(256) void scan_apply_kernel(const int32_t* __restrict__ in, int n, const int32_t* __restrict__ block_sums,
                                                         int32_t* __restrict__ out_start, int32_t* __restrict__ cursor) {
    // each lane scans 16 consecutive elements, wave/LDS scan of the lane totals, plus the workgroup offset
    __shared__ int sh[256];
    __shared__ int sh_off[4];
    int offset = 0;
    if constexpr (SELF) {
        int part = 0;
        for (int b = threadIdx.x; b < (int)blockIdx.x; b += 256) part += block_sums[b];   // RAW chunk totals here
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
        if ((threadIdx.x & 63) == 0) sh_off[threadIdx.x >> 6] = part;
        __syncthreads();
        offset = sh_off[0] + sh_off[1] + sh_off[2] + sh_off[3];
    } else {
        offset = block_sums[blockIdx.x];   // already an exclusive scan
    }
    const int base = blockIdx.x * SCAN_CHUNK + threadIdx.x * 16;
    int v[16];
    int s = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int i = base + k;
        v[k] = i < n ? in[i] : 0;
        s += v[k];
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        int t = threadIdx.x >= off ? sh[threadIdx.x - off] : 0;
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    int run = offset + (threadIdx.x > 0 ? sh[threadIdx.x - 1] : 0);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int i = base + k;
        if (i < n) { out_start[i] = run; cursor[i] = run; }
        run += v[k];
        if (SELF && i == n - 1) out_start[n] = run;   // the total
    }
}

__global__ void scan_tail_kernel(const int32_t* __restrict__ counts, int32_t* __restrict__ start, int n) {
    // start[n] = total = start[n-1] + counts[n-1]
    if (threadIdx.x == 0 && blockIdx.x == 0) start[n] = start[n - 1] + counts[n - 1];
}

static void launch_scan(hipStream_t st, const int32_t* d_counts, int ncells, int32_t* d_start, int32_t* d_cursor, int32_t* d_block_sums) {
    const int nb = (ncells + SCAN_CHUNK - 1) / SCAN_CHUNK;
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3(nb), dim3(256), 0, st, d_counts, ncells, d_block_sums);
    if (nb <= 1024) {
        hipLaunchKernelGGL(scan_apply_kernel<true>, dim3(nb), dim3(256), 0, st, d_counts, ncells, d_block_sums, d_start, d_cursor);
    } else {
        hipLaunchKernelGGL(scan_of_block_sums_kernel, dim3(1), dim3(1024), 0, st, d_block_sums, nb);
        hipLaunchKernelGGL(scan_apply_kernel<false>, dim3(nb), dim3(256), 0, st, d_counts, ncells, d_block_sums, d_start, d_cursor);
        hipLaunchKernelGGL(scan_tail_kernel, dim3(1), dim3(64), 0, st, d_counts, d_start, ncells);
    }
}

__global__ __launch_bounds__(256) void grid_scatter_kernel(const float4* __restrict__ tgt, int n, GridParams gp,
                                                           int32_t* __restrict__ cursor, float4* __restrict__ sorted) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 p = tgt[i];
    const int cx = cell_coord(p.x, gp.ox, gp.inv_h, gp.gx), cy = cell_coord(p.y, gp.oy, gp.inv_h, gp.gy),
              cz = cell_coord(p.z, gp.oz, gp.inv_h, gp.gz);
    const int pos = atomicAdd(&cursor[(cz * gp.gy + cy) * gp.gx + cx], 1);
    p.w = __int_as_float(i);   // original index rides in .w
    sorted[pos] = p;
}

void launch_grid_build(hipStream_t st, const float4* d_tgt, int n, const GridParams& gp, int32_t* d_counts,
                       int32_t* d_start, int32_t* d_cursor, int32_t* d_block_sums, float4* d_sorted) {
    const int ncells = gp.gx * gp.gy * gp.gz;
    hipMemsetAsync(d_counts, 0, (size_t)ncells * sizeof(int32_t), st);
    hipLaunchKernelGGL(grid_count_kernel, dim3((n + 255) / 256), dim3(256), 0, st, d_tgt, n, gp, d_counts);
    launch_scan(st, d_counts, ncells, d_start, d_cursor, d_block_sums);
    hipLaunchKernelGGL(grid_scatter_kernel, dim3((n + 255) / 256), dim3(256), 0, st, d_tgt, n, gp, d_cursor, d_sorted);
}

// ---- query -----------------------------------------------------------------------------------------------
// One lane per query.  A lane keeps its best (d2, idx) packed as ONE unsigned 64-bit key (bits(d2) << 32 | idx:
// unsigned order == smaller distance first, then lower index; d2 >= +0 so its bit pattern is monotone).
// (2-16 cooperating lanes per query were measured as well: once sources are cell-sorted and the block's ranges
// are fetched up front, one lane wins at every size tried.)
constexpr int GRID_BS_DEFAULT = 512;  // workgroup size
#ifndef KSS_GRID_WALK
#define KSS_GRID_WALK 8
#endif
constexpr int GRID_WALK = KSS_GRID_WALK;   // points in flight per step of the walk (4 / 8 / 16 measured equal)

template <bool FMA>
__device__ __forceinline__ unsigned long long point_key(const float4 p, float qx, float qy, float qz) {
    const float dx = qx - p.x, dy = qy - p.y, dz = qz - p.z;
    float d;
    if constexpr (FMA) d = __builtin_fmaf(dx, dx, __builtin_fmaf(dy, dy, dz * dz));
    else d = (dx * dx + dy * dy) + dz * dz;
    return ((unsigned long long)__float_as_uint(d) << 32) | (unsigned long long)(unsigned)__float_as_int(p.w);
}

// Scan the cell-ordered points [lo, hi): four independent loads in flight per step (indices clamped to the
// last point: a duplicate cannot change a minimum), tracking where the best point sits in `sorted` and its
// coordinates (three selects per evaluation are cheaper than re-loading the winner afterwards).
template <bool FMA>
__device__ __forceinline__ void scan_range(const float4* __restrict__ sorted, int lo, int hi, float qx, float qy, float qz,
                                           unsigned long long& key, int& kpos, float4& win) {
    for (int k = lo; k < hi; k += 4) {
        const int k1 = min(k + 1, hi - 1), k2 = min(k + 2, hi - 1), k3 = min(k + 3, hi - 1);
        const float4 p0 = sorted[k], p1 = sorted[k1], p2 = sorted[k2], p3 = sorted[k3];
        const unsigned long long e0 = point_key<FMA>(p0, qx, qy, qz), e1 = point_key<FMA>(p1, qx, qy, qz),
                                 e2 = point_key<FMA>(p2, qx, qy, qz), e3 = point_key<FMA>(p3, qx, qy, qz);
        if (e0 < key) { key = e0; kpos = k; win = p0; }
        if (e1 < key) { key = e1; kpos = k1; win = p1; }
        if (e2 < key) { key = e2; kpos = k2; win = p2; }
        if (e3 < key) { key = e3; kpos = k3; win = p3; }
    }
}

// The r = 1 step of a one-lane query: the 3x3x3 block around cell (cx, cy, cz) is 9 x-rows of <= 3 cells, every
// row ONE contiguous range of `sorted`.  A row-by-row scan is a chain of ~20 dependent L2 round trips; instead ALL
// range bounds are issued at once and the points of all rows are then walked as one list.
//
// Temporal coherence: the target this source matched in the PREVIOUS iteration (position `pp` in `sorted`, -1 = none)
// is evaluated first.  Its distance d0 is an upper bound of the answer, so a row / end cell of the block whose
// distance from the query exceeds d0 cannot hold the winner NOR a tie and is not read at all: with g = the f32 gap
// to the cell's slab minus eps (eps covers the rounding of the cell assignment and of the gap), every point there
// has computed d2 >= (1 - 3 ulp) * sum(g^2) > 0.999999 * sum(g^2) > d0.  Near convergence d0 is a fraction of a
// cell edge and ~4 of the 27 cells remain.  Results do not depend on `pp` (any target is a valid bound).
//
// The surviving ranges go into a per-lane queue in LDS (column threadIdx.x: private to the lane, so no barrier)
// and are walked as ONE flattened list, GRID_WALK points in flight per step: the loads issued are the loads
// needed (the vector memory pipe bounds this kernel: 36 unconditional float4 gathers per lane used to cost more
// than the whole remaining search).  Returns the number of distance evaluations.
template <bool FMA, int BS>
__device__ __forceinline__ int block_walk(const GridParams& gp, const int32_t* __restrict__ cell_start, const float4* __restrict__ sorted,
                                          float qx, float qy, float qz, int cx, int cy, int cz, int pp, int2 (*rowq)[BS],
                                          unsigned long long& key, int& kpos, float4& win) {
    const int xl = max(cx - 1, 0), xr = min(cx + 1, gp.gx - 1);
    int s0[9], s1[9], s2[9], s3[9];
    bool ok[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int z = cz + t / 3 - 1, y = cy + t % 3 - 1;
        ok[t] = z >= 0 && z < gp.gz && y >= 0 && y < gp.gy;
        const int row = ok[t] ? (z * gp.gy + y) * gp.gx : 0;
        s0[t] = cell_start[row + xl];       // [s0, s1) = cell cx-1 (empty when it does not exist: xl == cx)
        s1[t] = cell_start[row + cx];       // [s1, s2) = cell cx
        s2[t] = cell_start[row + cx + 1];   // [s2, s3) = cell cx+1 (empty when it does not exist: xr == cx)
        s3[t] = cell_start[row + xr + 1];
    }
    float d0 = __builtin_inff();
    if (pp >= 0) {
        win = sorted[pp];
        key = point_key<FMA>(win, qx, qy, qz);
        kpos = pp;
        d0 = __uint_as_float((unsigned)(key >> 32));
    }
    // squared slack-reduced gaps from the query to the neighbouring slabs (same face expressions as the termination
    // bound of the shell loop); index 1 = the query's own slab
    const float exl = fmaxf((qx - (gp.ox + (float)cx * gp.h)) - gp.eps, 0.f), exr = fmaxf(((gp.ox + (float)(cx + 1) * gp.h) - qx) - gp.eps, 0.f);
    const float eyl = fmaxf((qy - (gp.oy + (float)cy * gp.h)) - gp.eps, 0.f), eyr = fmaxf(((gp.oy + (float)(cy + 1) * gp.h) - qy) - gp.eps, 0.f);
    const float ezl = fmaxf((qz - (gp.oz + (float)cz * gp.h)) - gp.eps, 0.f), ezr = fmaxf(((gp.oz + (float)(cz + 1) * gp.h) - qz) - gp.eps, 0.f);
    const float exl2 = exl * exl, exr2 = exr * exr;
    const float ey2[3] = {eyl * eyl, 0.f, eyr * eyr}, ez2[3] = {ezl * ezl, 0.f, ezr * ezr};
    int nrow = 0, total = 0;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const float g2 = ey2[t % 3] + ez2[t / 3];
        const bool need = ok[t] && !(d0 < g2 * 0.999999f);
        const bool left = !(d0 < (g2 + exl2) * 0.999999f), right = !(d0 < (g2 + exr2) * 0.999999f);
        const int lo = left ? s0[t] : s1[t];
        const int hi = right ? s3[t] : s2[t];
        if (need && hi > lo) {
            rowq[nrow][threadIdx.x] = make_int2(lo, hi);
            ++nrow;
            total += hi - lo;
        }
    }
    int cur = 0, end = 0, nxt = 0;
    if (nrow > 0) { const int2 v = rowq[0][threadIdx.x]; cur = v.x; end = v.y; nxt = 1; }
    constexpr int U = GRID_WALK;
    for (int e = 0; e < total; e += U) {
        int at[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            at[j] = min(cur, end - 1);   // an exhausted lane repeats its last point (cannot change a minimum)
            ++cur;
            if (cur >= end && nxt < nrow) { const int2 v = rowq[nxt][threadIdx.x]; cur = v.x; end = v.y; ++nxt; }
        }
        float4 pt[U];
#pragma unroll
        for (int j = 0; j < U; ++j) pt[j] = sorted[at[j]];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const unsigned long long kk = point_key<FMA>(pt[j], qx, qy, qz);
            if (kk < key) { key = kk; kpos = at[j]; win = pt[j]; }
        }
    }
    return total + (pp >= 0 ? 1 : 0);
}

// ---- spatial order for the SOURCES: same cell order as the target, original index in .w ------------------
// Sources are scattered into cell order with atomics (arbitrary order inside a cell) and then every element
// computes its rank among the elements of its cell by original index, which makes the final order -- and
// with it every f64 sum of the run -- independent of the atomics' arrival order.  A rigid ICP update keeps
// neighbours neighbours, so ONE sort per registration keeps the queries of a wave / workgroup / XCD in the
// same few cells (L1 / L2 hits instead of Infinity-Cache trips) for all iterations.
__global__ __launch_bounds__(256) void grid_rank_fix_kernel(const float4* __restrict__ tmp, int n, GridParams gp,
                                                            const int32_t* __restrict__ start, float4* __restrict__ out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const float4 p = tmp[j];
    const int cx = cell_coord(p.x, gp.ox, gp.inv_h, gp.gx), cy = cell_coord(p.y, gp.oy, gp.inv_h, gp.gy),
              cz = cell_coord(p.z, gp.oz, gp.inv_h, gp.gz);
    const int c = (cz * gp.gy + cy) * gp.gx + cx;
    const int lo = start[c], hi = start[c + 1];
    const int me = __float_as_int(p.w);
    int rank = 0;
    for (int k = lo; k < hi; ++k) rank += __float_as_int(tmp[k].w) < me ? 1 : 0;
    out[lo + rank] = p;
}

void launch_grid_sort_sources(hipStream_t st, const float4* d_src, int n, const GridParams& gp, int32_t* d_counts,
                              int32_t* d_start, int32_t* d_cursor, int32_t* d_block_sums, float4* d_tmp, float4* d_out) {
    launch_grid_build(st, d_src, n, gp, d_counts, d_start, d_cursor, d_block_sums, d_tmp);
    hipLaunchKernelGGL(grid_rank_fix_kernel, dim3((n + 255) / 256), dim3(256), 0, st, d_tmp, n, gp, d_start, d_out);
}

// One launch per ICP iteration: cell search + correspondence sums + the final reduction.
//   - persistent workgroups stride over the query groups and keep the 20 f64 sums in registers;
//   - each workgroup publishes one partial row, then takes a ticket on a device counter; the workgroup that
//     draws the last ticket adds the rows up IN ROW ORDER (bitwise reproducible) and writes the result --
//     normally straight into host-mapped pinned memory -- and re-arms the counter;
//   - cross-workgroup visibility follows cdna_hip_programming.md Guideline 16: storing lanes release at agent
//     scope (+ explicit s_waitcnt vmcnt(0), the ROCm 7.2 compiler hazard), one lane takes the ticket, acquires
//     at agent scope, and the workgroup barrier publishes that to the other lanes before they load.
//   Sources the search gives up on are excluded from the sums and counted in slot 19: the host then runs the
//   brute-force list pass + the stand-alone reduce (rare: only for sources far from the target).
template <bool FMA, int BS>
__global__ __launch_bounds__(BS) void grid_nn_kernel(const PairState ps_arg, const float4* __restrict__ src_in,
                                                      float4* __restrict__ src_out, int ns, GridParams gp,
                                                      const int32_t* __restrict__ cell_start, const float4* __restrict__ sorted,
                                                      unsigned long long* __restrict__ keys,
                                                      int32_t* __restrict__ list, int32_t* __restrict__ list_count,
                                                      double max_d2, double* __restrict__ partials, int32_t* __restrict__ ticket,
                                                      int32_t* __restrict__ idx_out,
                                                      float* __restrict__ d2_out, unsigned long long seq,
                                                      unsigned long long* __restrict__ pub,
                                                      unsigned long long* __restrict__ stamps, int32_t* __restrict__ pos_prev,
                                                      const PairState* __restrict__ ps_host) {
    // diagnostic stamps (100 MHz s_memrealtime; null in production): [block*16 + {0 start, 1 searched, 2 reduced,
    // 3 ticketed}], last workgroup also [4 result stored]; search phase: 6 source loaded,
    // 8 block scanned, 9 shells done; 10 = distance evaluations of the workgroup (a count)
#define KSS_STAMP(k) do { if (stamps && threadIdx.x == 0) stamps[(size_t)blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
    // search-phase stamps drain the wave's loads first, so they time the dependent round trips (diagnostic runs only)
#define KSS_STAMPW(k) do { if (stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); KSS_STAMP(k); } } while (0)
    KSS_STAMP(0);
    // The transform normally rides in the kernel arguments.  A launch that was enqueued BEFORE its transform existed
    // (behind a hipStreamWaitValue64 gate, see kss_engine.hip) fetches it from host-mapped memory instead -- uniform
    // address: scalar loads, one PCIe read per CU, overlapped with the source loads below -- and leaves at once when the
    // host cancelled it (pad[0] != 0).
    PairState ps = ps_arg;
    if (ps_host) {
        ps = *ps_host;
        if (ps.pad[0] != 0) return;   // uniform
    }
    constexpr int FG = BS / NSUMS;           // lane groups of the two column-sum stages below
    __shared__ double shf[FG][NSUMS];
    __shared__ int s_last;
    __shared__ int2 rowq[9][BS];   // block_walk's per-lane queue of point ranges
    // the 20 correspondence sums of this lane's queries (KSS_NSUMS in include/kssicp.h); ps travels as a kernel
    // argument: no per-iteration upload
    double acc[NSUMS];
#pragma unroll
    for (int j = 0; j < NSUMS; ++j) acc[j] = 0.0;

    // XCD-aware, contiguous chunks: workgroups b and b+8 share an XCD (round-robin dispatch), so the virtual
    // index below hands every XCD one contiguous eighth of the (spatially sorted) sources and its L2 then
    // holds one eighth of the cell list; a different placement only changes speed, never results.
    const int per_xcd = (int)gridDim.x / 8;                     // the launcher makes gridDim.x a multiple of 8
    const int vb = ((int)blockIdx.x % 8) * per_xcd + (int)blockIdx.x / 8;   // bijection on [0, gridDim.x)
    const int nrounds_total = (ns + BS - 1) / BS;               // one "round" = BS queries of one workgroup
    const int base = nrounds_total / (int)gridDim.x, rem = nrounds_total % (int)gridDim.x;
    const int rounds = base + (vb < rem ? 1 : 0);               // balanced: the first `rem` chunks are one longer
    const int first = vb * base + min(vb, rem);
    for (int rr = 0; rr < rounds; ++rr) {
        const int i = (first + rr) * BS + (int)threadIdx.x;
        if (i >= ns) break;
        float4 p = src_in[i];
        const int pp = pos_prev ? pos_prev[i] : -1;
        if (ps.apply) {   // pcl transformCloud with the previous iteration's Matrix4f, as nn_sweep_kernel does
            const float x = p.x, y = p.y, z = p.z;
            p.x = ((ps.m[0] * x + ps.m[1] * y) + ps.m[2] * z) + ps.m[3];
            p.y = ((ps.m[4] * x + ps.m[5] * y) + ps.m[6] * z) + ps.m[7];
            p.z = ((ps.m[8] * x + ps.m[9] * y) + ps.m[10] * z) + ps.m[11];
        }
        KSS_STAMPW(6);
        src_out[i] = p;
        const float qx = p.x, qy = p.y, qz = p.z;
        const int cx = cell_coord(qx, gp.ox, gp.inv_h, gp.gx), cy = cell_coord(qy, gp.oy, gp.inv_h, gp.gy),
                  cz = cell_coord(qz, gp.oz, gp.inv_h, gp.gz);
        unsigned long long key = ~0ull;
        int kpos = 0;
        float4 win = make_float4(0.f, 0.f, 0.f, 0.f);   // the best point so far
        bool done = false;
        // ---- r = 1: the 3x3x3 block, pruned by the previous winner's distance ----
        const int ev = block_walk<FMA, BS>(gp, cell_start, sorted, qx, qy, qz, cx, cy, cz, pp, rowq, key, kpos, win);
        if (stamps) {   // diagnostic runs: distance evaluations of the r = 1 block, summed per workgroup into slot 10
            int evs = ev;
#pragma unroll
            for (int m = 32; m > 0; m >>= 1) evs += __shfl_xor(evs, m, 64);
            if ((threadIdx.x & 63) == 0) atomicAdd(&stamps[(size_t)blockIdx.x * 16 + 10], (unsigned long long)evs);
        }
        KSS_STAMPW(8);
        for (int r = 1; r <= gp.rcap; ++r) {
            if (r > 1) {   // shell r: (2r+1)^2 rows
                const int w = 2 * r + 1;
                const int x0 = max(cx - r, 0), x1 = min(cx + r, gp.gx - 1);
                for (int t = 0; t < w * w; ++t) {
                    const int dz = t / w - r, dy = t % w - r;
                    const int z = cz + dz, y = cy + dy;
                    if (z < 0 || z >= gp.gz || y < 0 || y >= gp.gy) continue;
                    const int row = (z * gp.gy + y) * gp.gx;
                    if (dz == -r || dz == r || dy == -r || dy == r) {
                        // a row on the shell's y/z faces: the whole x extent is new
                        scan_range<FMA>(sorted, cell_start[row + x0], cell_start[row + x1 + 1], qx, qy, qz, key, kpos, win);
                    } else {
                        // interior row of shell r: only its two x end cells are new
                        if (cx - r >= 0) scan_range<FMA>(sorted, cell_start[row + cx - r], cell_start[row + cx - r + 1], qx, qy, qz, key, kpos, win);
                        if (cx + r < gp.gx) scan_range<FMA>(sorted, cell_start[row + cx + r], cell_start[row + cx + r + 1], qx, qy, qz, key, kpos, win);
                    }
                }
            }
            const float best = __uint_as_float((unsigned)(key >> 32));
            // distance from the query to the faces of the visited block; faces on the grid border are open
            float b = __builtin_inff();
            if (cx - r > 0) b = fminf(b, qx - (gp.ox + (float)(cx - r) * gp.h));
            if (cx + r < gp.gx - 1) b = fminf(b, (gp.ox + (float)(cx + r + 1) * gp.h) - qx);
            if (cy - r > 0) b = fminf(b, qy - (gp.oy + (float)(cy - r) * gp.h));
            if (cy + r < gp.gy - 1) b = fminf(b, (gp.oy + (float)(cy + r + 1) * gp.h) - qy);
            if (cz - r > 0) b = fminf(b, qz - (gp.oz + (float)(cz - r) * gp.h));
            if (cz + r < gp.gz - 1) b = fminf(b, (gp.oz + (float)(cz + r + 1) * gp.h) - qz);
            const float bs = b - gp.eps;
            if (b == __builtin_inff()) done = key != ~0ull;                 // the whole grid has been visited
            else if (bs > 0.f && best < bs * bs * 0.999999f) done = true;   // every unvisited point is strictly farther
            if (done) break;
        }
        KSS_STAMPW(9);
        if (done) {
            const float d2 = __uint_as_float((unsigned)(key >> 32));
            accumulate_corr(acc, qx, qy, qz, win.x, win.y, win.z, d2, max_d2);
            keys[i] = key;
            if (pos_prev) pos_prev[i] = kpos;
            const int oi = __float_as_int(p.w);   // original source index (sources are in cell order)
            if (idx_out) idx_out[oi] = (int)(unsigned)(key & 0xffffffffull);
            if (d2_out) d2_out[oi] = d2;
        } else {
            keys[i] = ~0ull;   // resolved by the brute-force list pass through atomicMin
            if (pos_prev) pos_prev[i] = -1;
            const int slot = atomicAdd(list_count, 1);
            list[slot] = i;
        }
    }

    KSS_STAMP(1);
    // ---- workgroup partial row, then the last workgroup finishes the job ----
    // 20 f64 per lane: a shuffle tree costs 240 ds_bpermute + dependent adds per wave (measured ~3 us of a 20 us
    // kernel).  Transpose through LDS instead: lane t stores column-major (conflict free), FG * 20 lanes each add one
    // column's rows g, g + FG, ... and the FG group totals are added in group order.
    double r = 0.0;
    {
        __shared__ double shT[NSUMS][BS + 2];
#pragma unroll
        for (int c = 0; c < NSUMS; ++c) shT[c][threadIdx.x] = acc[c];
        __syncthreads();
        const int c = threadIdx.x / FG, g = threadIdx.x % FG;
        if (c < NSUMS) {
            double a0 = 0.0, a1 = 0.0;
            int k = g;
            for (; k + FG < BS; k += 2 * FG) { a0 += shT[c][k]; a1 += shT[c][k + FG]; }
            if (k < BS) a0 += shT[c][k];
            shf[g][c] = a0 + a1;
        }
        __syncthreads();
        if (threadIdx.x < NSUMS)
            for (int gg = 0; gg < FG; ++gg) r += shf[gg][threadIdx.x];
        __syncthreads();   // shf is reused by the last workgroup below
    }
    if (threadIdx.x < NSUMS) {
        // hand-off without an L2 write-back (MI355X_MICROARCH.md, "Valid forms", table row 1): EVERY store of the
        // handed-off row is a write-through `sc1` store (relaxed agent-scope atomic store), the storing wave drains
        // them (vmcnt(0)), the workgroup barrier orders that before ONE lane's agent-scope atomic add, and the
        // workgroup whose add returns the last ticket reads every row with `sc1` loads after its own barrier.
        // A release fence here would write back all the source / key lines this XCD just dirtied (several us).
        __hip_atomic_store(&partials[(int64_t)blockIdx.x * NSUMS + threadIdx.x], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    KSS_STAMP(2);
    if (threadIdx.x == 0) {
        const int tk = atomicAdd(ticket, 1);
        s_last = tk == (int)gridDim.x - 1;
    }
    __syncthreads();
    KSS_STAMP(3);
    if (!s_last) return;
    // Column sums of the gridDim.x published rows, fixed order (bitwise reproducible): lane (g, c) adds rows g, g + FG,
    // g + 2 FG, ... of column c -- eight sc1 loads in flight per batch, i.e. ONE cross-XCD round trip for <= 8 * FG rows
    // (200 rows at C2) -- and the FG group totals are then added in group order.
    {
        const int g = threadIdx.x / NSUMS, c = threadIdx.x % NSUMS, nrows = (int)gridDim.x;
        if (g < FG) {
            double a = 0.0;
            for (int k = g; k < nrows; k += 8 * FG) {
                double t[8];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    t[j] = k + j * FG < nrows ? __hip_atomic_load(&partials[(int64_t)(k + j * FG) * NSUMS + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
#pragma unroll
                for (int j = 0; j < 8; ++j) a += t[j];
            }
            shf[g][c] = a;
        }
    }
    __syncthreads();
    double v = 0.0;
    if (threadIdx.x < NSUMS)
        for (int gg = 0; gg < FG; ++gg) v += shf[gg][threadIdx.x];
    if (threadIdx.x < NSUMS) {
        if (threadIdx.x == NSUMS - 1) v = (double)__hip_atomic_load(list_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // publish {bits(sum), seq} as ONE aligned 16-byte system-scope store per sum into host-mapped memory: the host
        // accepts a slot when its sequence number matches, so no flag has to be ordered after the data (that ordering
        // would cost a write-acknowledge round trip over PCIe) and no L2 write-back fence is needed
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        const unsigned long long vb = (unsigned long long)__double_as_longlong(v);
        u32x4 w;
        w.x = (unsigned)vb; w.y = (unsigned)(vb >> 32); w.z = (unsigned)seq; w.w = (unsigned)(seq >> 32);
        unsigned long long* dst = pub + 2 * threadIdx.x;
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(w) : "memory");
        KSS_STAMP(4);
    }
    if (threadIdx.x == 0) *ticket = 0;   // re-arm for the next launch (stream order makes it visible)
#undef KSS_STAMPW
#undef KSS_STAMP
}

// =============================================================================================
// Batched variant (configs C3 / C5: many independent pairs): one cell list PER PAIR, all pairs built and
// queried by the same launches.  Cells of pair p occupy [cell_base, cell_base + gx*gy*gz) of one global cell
// array, so ONE exclusive scan yields global positions in the pair-by-pair sorted arrays.  There is no
// brute-force fallback here: a query simply keeps adding shells until the pair's whole grid is visited.
// =============================================================================================
__device__ __forceinline__ int pair_of(int i, const GridPairDev* __restrict__ pairs, int npairs, bool by_src) {
    int lo = 0, hi = npairs - 1;   // last pair whose base <= i
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        const int base = by_src ? pairs[mid].src_base : pairs[mid].tgt_base;
        if (base <= i) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__global__ __launch_bounds__(256) void gridb_bbox_kernel(const float4* __restrict__ tgt, const GridPairDev* __restrict__ pairs,
                                                         float* __restrict__ bbox) {
    __shared__ float sh[4][6];
    const GridPairDev pr = pairs[blockIdx.x];
    float mn[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
    float mx[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    for (int i = threadIdx.x; i < pr.tgt_n; i += blockDim.x) {
        const float4 p = tgt[pr.tgt_base + i];
        mn[0] = fminf(mn[0], p.x); mn[1] = fminf(mn[1], p.y); mn[2] = fminf(mn[2], p.z);
        mx[0] = fmaxf(mx[0], p.x); mx[1] = fmaxf(mx[1], p.y); mx[2] = fmaxf(mx[2], p.z);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            mn[k] = fminf(mn[k], __shfl_down(mn[k], off, 64));
            mx[k] = fmaxf(mx[k], __shfl_down(mx[k], off, 64));
        }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0)
        for (int k = 0; k < 3; ++k) { sh[wave][k] = mn[k]; sh[wave][3 + k] = mx[k]; }
    __syncthreads();
    if (threadIdx.x < 6) {
        float v = sh[0][threadIdx.x];
        for (int w = 1; w < 4; ++w) v = threadIdx.x < 3 ? fminf(v, sh[w][threadIdx.x]) : fmaxf(v, sh[w][threadIdx.x]);
        bbox[blockIdx.x * 6 + threadIdx.x] = v;
    }
}

// counts (SCATTER = false) or scatter into cell order (SCATTER = true) of targets (BY_SRC = false: positions of
// tgt4, padding skipped, .w = pair-relative index) or sources (BY_SRC = true: .w = global source index)
template <bool SCATTER, bool BY_SRC>
__global__ __launch_bounds__(256) void gridb_bin_kernel(const float4* __restrict__ pts, int total, const GridPairDev* __restrict__ pairs,
                                                        int npairs, int32_t* __restrict__ counts_or_cursor, float4* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int pi = pair_of(i, pairs, npairs, BY_SRC);
    const GridPairDev pr = pairs[pi];
    const int local = i - (BY_SRC ? pr.src_base : pr.tgt_base);
    if (local >= (BY_SRC ? pr.src_n : pr.tgt_n)) return;   // sentinel padding of the target layout
    float4 p = pts[i];
    const GridParams& gp = pr.gp;
    const int cx = cell_coord(p.x, gp.ox, gp.inv_h, gp.gx), cy = cell_coord(p.y, gp.oy, gp.inv_h, gp.gy),
              cz = cell_coord(p.z, gp.oz, gp.inv_h, gp.gz);
    const int cell = pr.cell_base + (cz * gp.gy + cy) * gp.gx + cx;
    if constexpr (SCATTER) {
        const int pos = atomicAdd(&counts_or_cursor[cell], 1);
        p.w = __int_as_float(BY_SRC ? i : local);
        out[pos] = p;
    } else {
        atomicAdd(&counts_or_cursor[cell], 1);
    }
}

// deterministic in-cell order for the sources (see grid_rank_fix_kernel)
__global__ __launch_bounds__(256) void gridb_rank_fix_kernel(const float4* __restrict__ tmp, int total, const GridPairDev* __restrict__ pairs,
                                                             int npairs, const int32_t* __restrict__ start, float4* __restrict__ out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= total) return;
    const float4 p = tmp[j];
    const int pi = pair_of(j, pairs, npairs, true);   // sorted positions stay inside the pair's source segment
    const GridPairDev pr = pairs[pi];
    const GridParams& gp = pr.gp;
    const int cx = cell_coord(p.x, gp.ox, gp.inv_h, gp.gx), cy = cell_coord(p.y, gp.oy, gp.inv_h, gp.gy),
              cz = cell_coord(p.z, gp.oz, gp.inv_h, gp.gz);
    const int c = pr.cell_base + (cz * gp.gy + cy) * gp.gx + cx;
    const int lo = start[c], hi = start[c + 1];
    const int me = __float_as_int(p.w);
    int rank = 0;
    for (int k = lo; k < hi; ++k) rank += __float_as_int(tmp[k].w) < me ? 1 : 0;
    out[lo + rank] = p;
}

void launch_gridb_bbox(hipStream_t st, const float4* d_tgt, const GridPairDev* d_pairs, int npairs, float* d_bbox) {
    hipLaunchKernelGGL(gridb_bbox_kernel, dim3(npairs), dim3(256), 0, st, d_tgt, d_pairs, d_bbox);
}

// targets: tgt4 (padded layout, total_tgt_pad slots) -> d_sorted (pair by pair, sum of nt entries), d_start
void launch_gridb_build_targets(hipStream_t st, const float4* d_tgt, int total_tgt_pad, const GridPairDev* d_pairs, int npairs,
                                int total_cells, int32_t* d_counts, int32_t* d_start, int32_t* d_cursor, int32_t* d_block_sums,
                                float4* d_sorted) {
    hipMemsetAsync(d_counts, 0, (size_t)total_cells * sizeof(int32_t), st);
    const dim3 grid((total_tgt_pad + 255) / 256), block(256);
    hipLaunchKernelGGL((gridb_bin_kernel<false, false>), grid, block, 0, st, d_tgt, total_tgt_pad, d_pairs, npairs, d_counts, (float4*)nullptr);
    launch_scan(st, d_counts, total_cells, d_start, d_cursor, d_block_sums);
    hipLaunchKernelGGL((gridb_bin_kernel<true, false>), grid, block, 0, st, d_tgt, total_tgt_pad, d_pairs, npairs, d_cursor, d_sorted);
}

// sources: d_src (total_src float4) -> d_out in (pair, cell, original index) order; d_tmp is scratch
void launch_gridb_sort_sources(hipStream_t st, const float4* d_src, int total_src, const GridPairDev* d_pairs, int npairs,
                               int total_cells, int32_t* d_counts, int32_t* d_start, int32_t* d_cursor, int32_t* d_block_sums,
                               float4* d_tmp, float4* d_out) {
    hipMemsetAsync(d_counts, 0, (size_t)total_cells * sizeof(int32_t), st);
    const dim3 grid((total_src + 255) / 256), block(256);
    hipLaunchKernelGGL((gridb_bin_kernel<false, true>), grid, block, 0, st, d_src, total_src, d_pairs, npairs, d_counts, (float4*)nullptr);
    launch_scan(st, d_counts, total_cells, d_start, d_cursor, d_block_sums);
    hipLaunchKernelGGL((gridb_bin_kernel<true, true>), grid, block, 0, st, d_src, total_src, d_pairs, npairs, d_cursor, d_tmp);
    hipLaunchKernelGGL(gridb_rank_fix_kernel, grid, block, 0, st, d_tmp, total_src, d_pairs, npairs, d_start, d_out);
}

// query + correspondence sums: one workgroup per RedWork item (consecutive sorted sources of ONE pair, 256 per
// round), one lane per source.  Writes the transformed source, the previous-winner position and one partial row of
// the 20 sums (the matched target's coordinates are still in registers: no gather pass); finalize_sums_kernel then
// adds each pair's rows in row order.
// Fallback: a lane still unresolved after GridParams::rcap shells (a source far from its target: badly posed pairs)
// stops adding shells; its WAVE then sweeps the pair's whole target, brute force, for exactly those lanes (every lane
// reads the same target: uniform addresses, scalar-cache loads, no LDS and no barrier, so the common path stays
// barrier free).  A pass therefore never costs more than the shells plus one brute-force sweep of the pair -- without
// it the shell loop is cubic in the distance.
template <bool FMA>
__global__ __launch_bounds__(256) void gridb_nn_kernel(const RedWork* __restrict__ work, const PairState* __restrict__ state,
                                                       const GridPairDev* __restrict__ pairs, const float4* __restrict__ src_in,
                                                       float4* __restrict__ src_out, const int32_t* __restrict__ cell_start,
                                                       const float4* __restrict__ sorted, const float4* __restrict__ tgt4,
                                                       int32_t* __restrict__ pos_prev, double max_d2, double* __restrict__ partials,
                                                       int32_t* __restrict__ idx_out, float* __restrict__ d2_out) {
    __shared__ int2 rowq[9][256];   // block_walk's per-lane range queue
    __shared__ double sh[4][NSUMS];
    const RedWork w = work[blockIdx.x];
    const PairState ps = state[w.pair];
    double acc[NSUMS];
#pragma unroll
    for (int c = 0; c < NSUMS; ++c) acc[c] = 0.0;
    if (ps.active) {   // uniform: one pair per workgroup
        const GridPairDev pr = pairs[w.pair];
        const GridParams& gp = pr.gp;
        const int32_t* __restrict__ cs = cell_start + pr.cell_base;
        for (int t = threadIdx.x; t < w.src_count; t += 256) {
            const int i = w.src_begin + t;
            float4 p = src_in[i];
            if (ps.apply) {
                const float x = p.x, y = p.y, z = p.z;
                p.x = ((ps.m[0] * x + ps.m[1] * y) + ps.m[2] * z) + ps.m[3];
                p.y = ((ps.m[4] * x + ps.m[5] * y) + ps.m[6] * z) + ps.m[7];
                p.z = ((ps.m[8] * x + ps.m[9] * y) + ps.m[10] * z) + ps.m[11];
            }
            src_out[i] = p;
            const float qx = p.x, qy = p.y, qz = p.z;
            const int cx = cell_coord(qx, gp.ox, gp.inv_h, gp.gx), cy = cell_coord(qy, gp.oy, gp.inv_h, gp.gy),
                      cz = cell_coord(qz, gp.oz, gp.inv_h, gp.gz);
            unsigned long long key = ~0ull;
            int kpos = -1;
            float4 win = make_float4(0.f, 0.f, 0.f, 0.f);
            bool done = false;
            for (int r = 1; r <= gp.rcap && !done; ++r) {
                const int wd = 2 * r + 1;
                const int x0 = max(cx - r, 0), x1 = min(cx + r, gp.gx - 1);
                if (r == 1) {
                    block_walk<FMA, 256>(gp, cs, sorted, qx, qy, qz, cx, cy, cz, pos_prev ? pos_prev[i] : -1, rowq, key, kpos, win);
                } else {
                    for (int u = 0; u < wd * wd; ++u) {
                        const int dz = u / wd - r, dy = u % wd - r;
                        const int z = cz + dz, y = cy + dy;
                        if (z < 0 || z >= gp.gz || y < 0 || y >= gp.gy) continue;
                        const int row = (z * gp.gy + y) * gp.gx;
                        if (dz == -r || dz == r || dy == -r || dy == r) {
                            scan_range<FMA>(sorted, cs[row + x0], cs[row + x1 + 1], qx, qy, qz, key, kpos, win);
                        } else {
                            if (cx - r >= 0) scan_range<FMA>(sorted, cs[row + cx - r], cs[row + cx - r + 1], qx, qy, qz, key, kpos, win);
                            if (cx + r < gp.gx) scan_range<FMA>(sorted, cs[row + cx + r], cs[row + cx + r + 1], qx, qy, qz, key, kpos, win);
                        }
                    }
                }
                const float best = __uint_as_float((unsigned)(key >> 32));
                float b = __builtin_inff();
                if (cx - r > 0) b = fminf(b, qx - (gp.ox + (float)(cx - r) * gp.h));
                if (cx + r < gp.gx - 1) b = fminf(b, (gp.ox + (float)(cx + r + 1) * gp.h) - qx);
                if (cy - r > 0) b = fminf(b, qy - (gp.oy + (float)(cy - r) * gp.h));
                if (cy + r < gp.gy - 1) b = fminf(b, (gp.oy + (float)(cy + r + 1) * gp.h) - qy);
                if (cz - r > 0) b = fminf(b, qz - (gp.oz + (float)(cz - r) * gp.h));
                if (cz + r < gp.gz - 1) b = fminf(b, (gp.oz + (float)(cz + r + 1) * gp.h) - qz);
                const float bs = b - gp.eps;
                if (b == __builtin_inff()) done = key != ~0ull;                  // the pair's whole grid has been visited
                else if (bs > 0.f && best < bs * bs * 0.999999f) done = true;    // every unvisited point is strictly farther
            }
            // ---- bounded fallback: brute force over the pair's targets for the lanes the shells did not resolve ----
            if (__builtin_amdgcn_ballot_w64(!done) != 0ull) {             // wave-uniform
                const float4* __restrict__ tp = tgt4 + pr.tgt_base;      // original order; uniform addresses below
                for (int j = 0; j < pr.tgt_n; ++j) {
                    const float4 q = tp[j];
                    const float dx = qx - q.x, dy = qy - q.y, dz = qz - q.z;
                    float d;
                    if constexpr (FMA) d = __builtin_fmaf(dx, dx, __builtin_fmaf(dy, dy, dz * dz));
                    else d = (dx * dx + dy * dy) + dz * dz;
                    const unsigned long long kk = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned long long)(unsigned)j;
                    if (!done && kk < key) { key = kk; win = q; kpos = -1; }   // (no position in `sorted` known: no warm start next time)
                }
                if (!done) acc[NSUMS - 1] += 1.0;   // slot 19: lanes that needed the fallback (the host may change engine)
            }
            if (key != ~0ull) {   // (an empty target cannot happen: the plan rejects it)
                const float d2 = __uint_as_float((unsigned)(key >> 32));
                accumulate_corr(acc, qx, qy, qz, win.x, win.y, win.z, d2, max_d2);
                if (pos_prev) pos_prev[i] = kpos;   // positions are global in the pair-by-pair `sorted`
                const int oi = __float_as_int(p.w);
                if (idx_out) idx_out[oi] = (int)(unsigned)(key & 0xffffffffull);
                if (d2_out) d2_out[oi] = d2;
            }
        }
    }
    const double r = block_sum<NSUMS>(acc, sh);
    if (threadIdx.x < NSUMS) partials[(int64_t)w.partial_index * NSUMS + threadIdx.x] = r;
}

void launch_gridb_nn(hipStream_t st, bool fma, const RedWork* d_work, int n_work, const PairState* d_state,
                     const GridPairDev* d_pairs, const float4* d_src_in, float4* d_src_out, const int32_t* d_cell_start,
                     const float4* d_sorted, const float4* d_tgt4, int32_t* d_pos, double max_d2, double* d_partials,
                     int32_t* d_idx_out, float* d_d2_out) {
    if (n_work <= 0) return;
    const dim3 grid(n_work), block(256);
    if (fma)
        hipLaunchKernelGGL(gridb_nn_kernel<true>, grid, block, 0, st, d_work, d_state, d_pairs, d_src_in, d_src_out, d_cell_start,
                           d_sorted, d_tgt4, d_pos, max_d2, d_partials, d_idx_out, d_d2_out);
    else
        hipLaunchKernelGGL(gridb_nn_kernel<false>, grid, block, 0, st, d_work, d_state, d_pairs, d_src_in, d_src_out, d_cell_start,
                           d_sorted, d_tgt4, d_pos, max_d2, d_partials, d_idx_out, d_d2_out);
}

// statistics for the roofline statement: evaluations of one r = 1 pass and occupied cells (profiling only)
__global__ __launch_bounds__(256) void grid_stats_kernel(const float4* __restrict__ src, int ns, GridParams gp,
                                                         const int32_t* __restrict__ cell_start, unsigned long long* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long ev = 0, occ = 0;
    if (i < ns) {
        const float4 p = src[i];
        const int cx = cell_coord(p.x, gp.ox, gp.inv_h, gp.gx), cy = cell_coord(p.y, gp.oy, gp.inv_h, gp.gy),
                  cz = cell_coord(p.z, gp.oz, gp.inv_h, gp.gz);
        const int x0 = max(cx - 1, 0), x1 = min(cx + 1, gp.gx - 1);
        for (int t = 0; t < 9; ++t) {
            const int z = cz + t / 3 - 1, y = cy + t % 3 - 1;
            if (z < 0 || z >= gp.gz || y < 0 || y >= gp.gy) continue;
            const int row = (z * gp.gy + y) * gp.gx;
            ev += (unsigned long long)(cell_start[row + x1 + 1] - cell_start[row + x0]);
        }
    }
    const int ncells = gp.gx * gp.gy * gp.gz;
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < ncells; c += gridDim.x * blockDim.x)
        occ += cell_start[c + 1] > cell_start[c] ? 1 : 0;
    atomicAdd(&out[0], ev);
    atomicAdd(&out[1], occ);
}

void launch_grid_stats(hipStream_t st, const float4* d_src, int ns, const GridParams& gp, const int32_t* d_cell_start,
                       unsigned long long* d_out) {
    hipLaunchKernelGGL(grid_stats_kernel, dim3((ns + 255) / 256), dim3(256), 0, st, d_src, ns, gp, d_cell_start, d_out);
}

static int env_int(const char* name, int dflt) {
    if (const char* e = getenv(name)) {
        const int v = atoi(e);
        if (v > 0) return v;
    }
    return dflt;
}
// tuning hooks (defaults measured on C2, 100k x 100k; profiles/): workgroup size, workgroup cap
static int grid_bs() { const int v = env_int("KSS_GRID_BS", GRID_BS_DEFAULT); return (v == 256 || v == 512) ? v : GRID_BS_DEFAULT; }

int grid_nn_blocks(int ns) {
    const int bs = grid_bs();
    const int need = (ns + bs - 1) / bs;
    const int cap = env_int("KSS_GRID_BLOCKS", 131072 / bs);   // one 512-lane workgroup per CU
    const int nb = need < cap ? need : cap;
    return (nb + 7) / 8 * 8;   // multiple of 8: the XCD-aware block remap in the kernel is then a bijection
}

template <bool FMA, int BS>
static void grid_launch(hipStream_t st, dim3 grid, const PairState& state, const float4* d_src_in, float4* d_src_out, int ns,
                        const GridParams& gp, const int32_t* d_cell_start, const float4* d_sorted, unsigned long long* d_keys,
                        int32_t* d_list, int32_t* d_list_count, double max_d2, double* d_partials, int32_t* d_ticket,
                        int32_t* d_idx_out, float* d_d2_out, unsigned long long seq,
                        unsigned long long* d_pub, unsigned long long* d_stamps, int32_t* d_pos, const PairState* d_ps_host) {
    hipLaunchKernelGGL((grid_nn_kernel<FMA, BS>), grid, dim3(BS), 0, st, state, d_src_in, d_src_out, ns, gp, d_cell_start,
                       d_sorted, d_keys, d_list, d_list_count, max_d2, d_partials, d_ticket, d_idx_out, d_d2_out, seq,
                       d_pub, d_stamps, d_pos, d_ps_host);
}

void launch_grid_nn(hipStream_t st, bool fma, const PairState& state, const float4* d_src_in, float4* d_src_out, int ns,
                    const GridParams& gp, const int32_t* d_cell_start, const float4* d_sorted,
                    unsigned long long* d_keys, int32_t* d_list, int32_t* d_list_count, double max_d2, double* d_partials,
                    int32_t* d_ticket, int32_t* d_idx_out, float* d_d2_out, unsigned long long seq,
                    unsigned long long* d_pub, unsigned long long* d_stamps, int32_t* d_pos, const PairState* d_ps_host) {
    const dim3 grid(grid_nn_blocks(ns));
    const int bs = grid_bs();
#define KSS_GRID_ARGS st, grid, state, d_src_in, d_src_out, ns, gp, d_cell_start, d_sorted, d_keys, d_list, d_list_count, max_d2, \
                      d_partials, d_ticket, d_idx_out, d_d2_out, seq, d_pub, d_stamps, d_pos, d_ps_host
    if (fma) { if (bs == 256) grid_launch<true, 256>(KSS_GRID_ARGS); else grid_launch<true, 512>(KSS_GRID_ARGS); }
    else     { if (bs == 256) grid_launch<false, 256>(KSS_GRID_ARGS); else grid_launch<false, 512>(KSS_GRID_ARGS); }
#undef KSS_GRID_ARGS
}

}  // namespace kss
