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
//   warm start: inside an ICP every source carries its last winner {coordinates, index} and a pair {B, acc}: "every other
//             target was at distance >= B when this source was last searched; it has moved by at most acc since".  While
//             (B - acc)^2 exceeds the winner's new distance (with margins for the f32 roundings) the winner stands and the
//             source is not searched at all (grid_pass_kernel, phase A).  A source that is searched evaluates its last
//             winner first; that distance, grown by a skin, prunes the cells of the first block that cannot hold the winner
//             nor a tie (block_walk / row_walk).  Results never depend on any of this.
//   fallback: a query not resolved within GridParams::rcap shells is appended to a list and resolved by the
//             brute-force sweep (nn_sweep_kernel<LIST>) -- far-away clouds never degrade below it.  (The batched
//             kernel sweeps the pair's targets inside the wave instead.)
//   fusion  : the correspondence sums are accumulated by the searching lanes (the winner's coordinates are still in
//             registers) and reduced in the same launch: grid_pass_kernel.
#pragma clang fp contract(off)

#include <hip/hip_runtime.h>

#include <cstring>

#include <rocprim/rocprim.hpp>

#include <cstdlib>

#include "kss_internal.hpp"
#include "kss_device.hpp"

namespace kss {

// ---- counting sort into cell order: target AND source of a pair by the same launches ---------------------------
// counts / starts are ONE array of 2 * ncells entries: [0, ncells) the target's cells, [ncells, 2 ncells) the source's.
// One exclusive scan over the whole array gives the target's starts directly (and start[ncells] = nt is their end
// sentinel) and the source's starts offset by nt.  Blocks [0, nbt) handle target points, the rest source points.
__global__ __launch_bounds__(256) void grid_count2_kernel(const float4* __restrict__ tgt, int nt, int nbt, const float4* __restrict__ src, int ns,
                                                          GridParams gp, int32_t* __restrict__ counts) {
    const bool is_src = (int)blockIdx.x >= nbt;
    const int i = ((int)blockIdx.x - (is_src ? nbt : 0)) * 256 + (int)threadIdx.x;
    if (i >= (is_src ? ns : nt)) return;
    const float4 p = is_src ? src[i] : tgt[i];
    const int cx = cell_coord(p.x, gp.ox, gp.inv_h, gp.gx), cy = cell_coord(p.y, gp.oy, gp.inv_h, gp.gy),
              cz = cell_coord(p.z, gp.oz, gp.inv_h, gp.gz);
    atomicAdd(&counts[(is_src ? gp.gx * gp.gy * gp.gz : 0) + (cz * gp.gy + cy) * gp.gx + cx], 1);
}

// exclusive scan, 3 phases; each workgroup owns SCAN_CHUNK consecutive elements
constexpr int SCAN_CHUNK = 4096;   // 256 threads x 16

__global__ __launch_bounds__(256) void scan_block_sums_kernel(const int32_t* __restrict__ in, int n, int32_t* __restrict__ block_sums) {
    __shared__ int sh[4];
    typedef int32_t int4u __attribute__((ext_vector_type(4), aligned(4)));
    const int base = blockIdx.x * SCAN_CHUNK;
    int s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {   // 16-byte loads, consecutive lanes consecutive addresses (4-byte loads: 2.9 TB/s on C4's table)
        const int i = base + (k * 256 + (int)threadIdx.x) * 4;
        if (i + 3 < n) { const int4u v = *(const int4u*)(in + i); s += (v.x + v.y) + (v.z + v.w); }
        else { if (i < n) s += in[i]; if (i + 1 < n) s += in[i + 1]; if (i + 2 < n) s += in[i + 2]; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

// single workgroup: exclusive scan of the chunk totals, in place (each lane a run of consecutive totals, wave scans of the
// lane sums, the sixteen wave totals through LDS)
__global__ __launch_bounds__(1024) void scan_of_block_sums_kernel(int32_t* __restrict__ block_sums, int nb) {
    __shared__ int wave_tot[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (nb + 1023) / 1024;
    const int lo = min(tid * per, nb), hi = min(nb, lo + per);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += block_sums[i];
    int incl = s;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int run = incl - s;
    for (int w = 0; w < 16; ++w) run += w < wave ? wave_tot[w] : 0;
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
                                                         int32_t* __restrict__ out_start) {
    // the chunk as four slices of 1024 elements; lane t holds elements 4t .. 4t+3 of each slice (one 16-byte load per slice,
    // consecutive lanes consecutive addresses), wave scans of the lane totals, slice / wave totals through LDS
    // (round 3: the first form gave every lane 16 CONSECUTIVE elements -- 64 lines per load instruction -- and scanned the
    // lane totals in eight barrier-separated steps: 14.7 us per C2 build, now 6)
    typedef int32_t int4u __attribute__((ext_vector_type(4), aligned(4)));
    __shared__ int sh_tot[4][4];
    __shared__ int sh_off[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int offset = 0;
    if constexpr (SELF) {
        int part = 0;
        for (int b = tid; b < (int)blockIdx.x; b += 256) part += block_sums[b];   // RAW chunk totals here
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
        if (lane == 0) sh_off[wave] = part;
    } else {
        offset = block_sums[blockIdx.x];   // already an exclusive scan
    }
    const int base = blockIdx.x * SCAN_CHUNK;
    int4u v[4];
    int incl[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = base + (k * 256 + tid) * 4;
        if (i + 3 < n) v[k] = *(const int4u*)(in + i);
        else { v[k].x = i < n ? in[i] : 0; v[k].y = i + 1 < n ? in[i + 1] : 0; v[k].z = i + 2 < n ? in[i + 2] : 0; v[k].w = 0; }
        int x = (v[k].x + v[k].y) + (v[k].z + v[k].w);
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(x, off, 64);
            if (lane >= off) x += t;
        }
        incl[k] = x;
        if (lane == 63) sh_tot[k][wave] = x;
    }
    __syncthreads();
    if constexpr (SELF) offset = sh_off[0] + sh_off[1] + sh_off[2] + sh_off[3];
    int run = offset;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int before = 0;                 // waves in front of this one within the slice
#pragma unroll
        for (int w = 0; w < 4; ++w) before += w < wave ? sh_tot[k][w] : 0;
        const int own = (v[k].x + v[k].y) + (v[k].z + v[k].w);
        const int r0 = run + before + (incl[k] - own);
        const int i = base + (k * 256 + tid) * 4;
        int4u o;
        o.x = r0; o.y = r0 + v[k].x; o.z = o.y + v[k].y; o.w = o.z + v[k].z;
        if (i + 3 < n) *(int4u*)(out_start + i) = o;
        else { if (i < n) out_start[i] = o.x; if (i + 1 < n) out_start[i + 1] = o.y; if (i + 2 < n) out_start[i + 2] = o.z; }
        if (i <= n - 1 && n - 1 <= i + 3) out_start[n] = o.x + (n - 1 >= i ? v[k].x : 0) + (n - 1 >= i + 1 ? v[k].y : 0) + (n - 1 >= i + 2 ? v[k].z : 0) + (n - 1 >= i + 3 ? v[k].w : 0);   // the total
        run += (sh_tot[k][0] + sh_tot[k][1]) + (sh_tot[k][2] + sh_tot[k][3]);
    }
}

// Exclusive scan of the cell counts: d_start[0 .. n] (d_start[n] = total): chunk totals, their scan (inside the apply kernel
// up to 1024 chunks), apply.  On C4's 2 x 15.8M counters: 25 + 9 + 43 us = the table read twice and written once at
// 5 TB/s.  (Round 2's kernels -- every lane 16 consecutive elements, totals read four bytes a lane -- took 240 us there and
// rocPRIM's decoupled-look-back scan was used from 4M cells up: 127 us in its default configuration, a chain of look-backs
// through ~8000 tiles, 76 us configured with 64 items per lane, below.  KSS_SCAN_LIB=1 selects it for an A/B:
// tools/ab_scan.sh.)  d_scratch must hold scan_scratch_bytes(n).
using BigScanConfig = rocprim::scan_config<256, 64, rocprim::block_load_method::block_load_transpose, rocprim::block_store_method::block_store_transpose,
                                           rocprim::block_scan_algorithm::reduce_then_scan>;

size_t scan_scratch_bytes(int n) {
    size_t bytes = 0;
    if (rocprim::inclusive_scan<BigScanConfig>(nullptr, bytes, (const int32_t*)nullptr, (int32_t*)nullptr, (size_t)n, rocprim::plus<int32_t>(), (hipStream_t)0) != hipSuccess) bytes = 0;
    return std::max<size_t>(bytes, ((size_t)(n + SCAN_CHUNK - 1) / SCAN_CHUNK + 1) * sizeof(int32_t)) + 256;
}

static void launch_scan(hipStream_t st, const int32_t* d_counts, int ncells, int32_t* d_start, int32_t* d_scratch) {
    static const bool lib = getenv("KSS_SCAN_LIB") != nullptr;   // A/B: rocPRIM's look-back scan for big tables
    if (lib && ncells > 1024 * SCAN_CHUNK) {
        size_t bytes = 0;
        if (rocprim::inclusive_scan<BigScanConfig>(nullptr, bytes, d_counts, d_start + 1, (size_t)ncells, rocprim::plus<int32_t>(), st) == hipSuccess) {
            hipMemsetAsync(d_start, 0, sizeof(int32_t), st);
            if (rocprim::inclusive_scan<BigScanConfig>((void*)d_scratch, bytes, d_counts, d_start + 1, (size_t)ncells, rocprim::plus<int32_t>(), st) == hipSuccess) return;
        }
        (void)hipGetLastError();
    }
    int32_t* d_block_sums = d_scratch;
    const int nb = (ncells + SCAN_CHUNK - 1) / SCAN_CHUNK;
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3(nb), dim3(256), 0, st, d_counts, ncells, d_block_sums);
    if (nb <= 1024) {
        hipLaunchKernelGGL(scan_apply_kernel<true>, dim3(nb), dim3(256), 0, st, d_counts, ncells, d_block_sums, d_start);
    } else {
        hipLaunchKernelGGL(scan_of_block_sums_kernel, dim3(1), dim3(1024), 0, st, d_block_sums, nb);
        hipLaunchKernelGGL(scan_apply_kernel<false>, dim3(nb), dim3(256), 0, st, d_counts, ncells, d_block_sums, d_start);   // (also writes start[n])
    }
}

// scatter of both clouds: the cell's count doubles as its cursor -- counting DOWN hands out the slots start + count-1 ..
// start and leaves the array zeroed for the next build (no cursor array for the scan to write, no memset in between).
// Targets go to `sorted` (.w = original index), sources to `src_tmp` (.w = original index; grid_rank_fix_kernel orders each cell).
__global__ __launch_bounds__(256) void grid_scatter2_kernel(const float4* __restrict__ tgt, int nt, int nbt, const float4* __restrict__ src, int ns,
                                                            GridParams gp, int32_t* __restrict__ counts, const int32_t* __restrict__ start,
                                                            float4* __restrict__ sorted, float4* __restrict__ src_tmp) {
    const bool is_src = (int)blockIdx.x >= nbt;
    const int i = ((int)blockIdx.x - (is_src ? nbt : 0)) * 256 + (int)threadIdx.x;
    if (i >= (is_src ? ns : nt)) return;
    float4 p = is_src ? src[i] : tgt[i];
    const int cx = cell_coord(p.x, gp.ox, gp.inv_h, gp.gx), cy = cell_coord(p.y, gp.oy, gp.inv_h, gp.gy),
              cz = cell_coord(p.z, gp.oz, gp.inv_h, gp.gz);
    const int cell = (is_src ? gp.gx * gp.gy * gp.gz : 0) + (cz * gp.gy + cy) * gp.gx + cx;
    const int pos = start[cell] + atomicSub(&counts[cell], 1) - 1 - (is_src ? nt : 0);
    p.w = __int_as_float(i);
    (is_src ? src_tmp : sorted)[pos] = p;
}

// ---- query -----------------------------------------------------------------------------------------------
// One lane per query.  A lane keeps its best (d2, idx) packed as ONE unsigned 64-bit key (bits(d2) << 32 | idx:
// unsigned order == smaller distance first, then lower index; d2 >= +0 so its bit pattern is monotone).
// (2-16 cooperating lanes per query were measured as well: once sources are cell-sorted and the block's ranges
// are fetched up front, one lane wins at every size tried.)
template <bool FMA>
__device__ __forceinline__ unsigned long long point_key(const float4 p, float qx, float qy, float qz) {
    const float dx = qx - p.x, dy = qy - p.y, dz = qz - p.z;
    float d;
    if constexpr (FMA) d = __builtin_fmaf(dx, dx, __builtin_fmaf(dy, dy, dz * dz));
    else d = (dx * dx + dy * dy) + dz * dz;
    return ((unsigned long long)__float_as_uint(d) << 32) | (unsigned long long)(unsigned)__float_as_int(p.w);
}

// Scan the cell-ordered points [lo, hi): four independent loads in flight per step (indices clamped to the
// last point: a duplicate cannot change a minimum), tracking the best key and where that point sits in `sorted`.
// The winner's coordinates are re-read once at the end of the search (one L1-hot load) instead of being carried through
// four selects per evaluation: the batched pass is VALU-bound (93 % VALU-busy at C3) and each select is an instruction.
template <bool FMA>
__device__ __forceinline__ void scan_range(const float4* __restrict__ sorted, int lo, int hi, float qx, float qy, float qz,
                                           unsigned long long& key, int& kpos) {
    for (int k = lo; k < hi; k += 4) {
        const int k1 = min(k + 1, hi - 1), k2 = min(k + 2, hi - 1), k3 = min(k + 3, hi - 1);
        const float4 p0 = sorted[k], p1 = sorted[k1], p2 = sorted[k2], p3 = sorted[k3];
        const unsigned long long e0 = point_key<FMA>(p0, qx, qy, qz), e1 = point_key<FMA>(p1, qx, qy, qz),
                                 e2 = point_key<FMA>(p2, qx, qy, qz), e3 = point_key<FMA>(p3, qx, qy, qz);
        if (e0 < key) { key = e0; kpos = k; }
        if (e1 < key) { key = e1; kpos = k1; }
        if (e2 < key) { key = e2; kpos = k2; }
        if (e3 < key) { key = e3; kpos = k3; }
    }
}

// The r = 1 step of a one-lane query: the 3x3x3 block around cell (cx, cy, cz) is 9 x-rows of <= 3 cells, every
// row ONE contiguous range of `sorted`.  A row-by-row scan is a chain of ~20 dependent L2 round trips; instead ALL
// range bounds are issued at once and the points of all rows are then walked as one list.
//
// Pruning: the caller passes `rho`, a squared radius that is at least the distance of a target it has already
// evaluated (key / kpos on entry; +inf with none).  A row / end cell of the block whose distance from the query exceeds
// rho cannot hold the winner NOR a tie and is not read at all: with g = the f32 gap to the cell's slab minus eps (eps
// covers the rounding of the cell assignment and of the gap), every point there has computed d2 >= (1 - 3 ulp) *
// sum(g^2) > 0.999999 * sum(g^2) > rho.  Results do not depend on rho (any such radius is a valid bound).
//
// The surviving ranges go into a per-lane queue in LDS (column threadIdx.x: private to the lane, so no barrier) and are
// walked as one list, eight points in flight per step.  m2 returns the SECOND smallest computed distance among the
// (distinct) points walked -- with rho and the block's faces it bounds how far every target other than the winner is
// (the skip test of grid_pass_kernel).  Returns the number of distance evaluations.
typedef int32_t int4u __attribute__((ext_vector_type(4), aligned(4)));   // 16-byte load at 4-byte alignment (one global_load_dwordx4)

template <bool FMA, int BS, int U>
__device__ __forceinline__ int block_walk(const GridParams& gp, const int32_t* __restrict__ cell_start, const float4* __restrict__ sorted,
                                          float qx, float qy, float qz, int cx, int cy, int cz, float rho, unsigned rowmask,
                                          int2 (*rowq)[BS], unsigned long long& key, int& kpos, float& m1, float& m2) {
    // the four bounds of a row -- starts of cells cx-1, cx, cx+1, cx+2 -- are ONE unaligned 16-byte load (cell_start has a
    // readable element in front of cell 0 and two behind the last start): 9 loads per query instead of 36
    int s0[9], s1[9], s2[9], s3[9];
    bool ok[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int z = cz + t / 3 - 1, y = cy + t % 3 - 1;
        ok[t] = z >= 0 && z < gp.gz && y >= 0 && y < gp.gy && ((rowmask >> t) & 1u) != 0u;   // (rowmask: the rows this lane walks)
        const int row = ok[t] ? (z * gp.gy + y) * gp.gx : 0;
        const int4u v = *(const int4u*)(cell_start + row + cx - 1);
        s0[t] = cx > 0 ? v.x : v.y;               // [s0, s1) = cell cx-1 (empty when it does not exist)
        s1[t] = v.y;                              // [s1, s2) = cell cx
        s2[t] = v.z;                              // [s2, s3) = cell cx+1 (empty when it does not exist)
        s3[t] = cx + 1 < gp.gx ? v.w : v.z;
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
        const bool need = ok[t] && !(rho < g2 * 0.999999f);
        const bool left = !(rho < (g2 + exl2) * 0.999999f), right = !(rho < (g2 + exr2) * 0.999999f);
        const int lo = left ? s0[t] : s1[t];
        const int hi = right ? s3[t] : s2[t];
        const bool take = need && hi > lo;
        rowq[min(nrow, 8)][threadIdx.x] = make_int2(lo, hi);   // written unconditionally, kept only when taken: no branch per row
        nrow += take ? 1 : 0;
        total += take ? hi - lo : 0;
    }
    // The queued ranges are walked as ONE flattened list, 8 points in flight per step.  (Measured and rejected: predicating
    // the loads of a lane past the end of its list instead of letting it repeat its last point, +25 % search time; walking
    // two ranges per step with four points each and no per-point queue bookkeeping, fewer instructions but more dependent
    // steps: C3 9.5 -> 12.4 ms, C2 search 5.1 -> 5.8 us.)
    int cur = 0, end = 0, nxt = 0;
    if (nrow > 0) { const int2 v = rowq[0][threadIdx.x]; cur = v.x; end = v.y; nxt = 1; }
    m1 = __builtin_inff();
    m2 = __builtin_inff();
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
            if (kk < key) { key = kk; kpos = at[j]; }
            // the two smallest distances over the DISTINCT points of the list (a repeated point counts once: masked)
            const float dd = e + j < total ? __uint_as_float((unsigned)(kk >> 32)) : __builtin_inff();
            m2 = __builtin_amdgcn_fmed3f(m1, m2, dd);
            m1 = fminf(m1, dd);
        }
    }
    return total;
}

// One ROW of the 3x3x3 block per lane -- or two (phase B of the fused pass when a workgroup has few walkers: 16 lanes share
// a walker and lanes 0-8 of the group take rows 0-8, or 8 lanes share one and lane 0 also takes row 8).  Same pruning rule
// and the same m1 / m2 bookkeeping as block_walk, but no range queue and -- for the usual <= U points -- a single step: the
// serial instruction count of a search, which is what a lone wave pays for, drops to less than half.  t1 < 0: one row.
__device__ __forceinline__ void row_range(const GridParams& gp, const int32_t* __restrict__ cell_start, float qx, float qy, float qz,
                                          int cx, int cy, int cz, float rho, int t, int& lo, int& hi) {
    lo = hi = 0;
    const int tz = t / 3, ty = t - 3 * tz;
    const int z = cz + tz - 1, y = cy + ty - 1;
    if (!(t >= 0 && t < 9 && z >= 0 && z < gp.gz && y >= 0 && y < gp.gy)) return;
    const int4u v = *(const int4u*)(cell_start + (z * gp.gy + y) * gp.gx + cx - 1);
    const int s0 = cx > 0 ? v.x : v.y, s1 = v.y, s2 = v.z, s3 = cx + 1 < gp.gx ? v.w : v.z;
    const float exl = fmaxf((qx - (gp.ox + (float)cx * gp.h)) - gp.eps, 0.f), exr = fmaxf(((gp.ox + (float)(cx + 1) * gp.h) - qx) - gp.eps, 0.f);
    const float eyl = fmaxf((qy - (gp.oy + (float)cy * gp.h)) - gp.eps, 0.f), eyr = fmaxf(((gp.oy + (float)(cy + 1) * gp.h) - qy) - gp.eps, 0.f);
    const float ezl = fmaxf((qz - (gp.oz + (float)cz * gp.h)) - gp.eps, 0.f), ezr = fmaxf(((gp.oz + (float)(cz + 1) * gp.h) - qz) - gp.eps, 0.f);
    const float ey2 = ty == 0 ? eyl * eyl : ty == 2 ? eyr * eyr : 0.f, ez2 = tz == 0 ? ezl * ezl : tz == 2 ? ezr * ezr : 0.f;
    const float g2 = ey2 + ez2;
    if (rho < g2 * 0.999999f) return;
    const bool left = !(rho < (g2 + exl * exl) * 0.999999f), right = !(rho < (g2 + exr * exr) * 0.999999f);
    lo = left ? s0 : s1;
    hi = right ? s3 : s2;
    if (hi < lo) hi = lo;
}

template <bool FMA, int U, bool TWO>
__device__ __forceinline__ void row_walk(const GridParams& gp, const int32_t* __restrict__ cell_start, const float4* __restrict__ sorted,
                                         float qx, float qy, float qz, int cx, int cy, int cz, float rho, int t0, int t1,
                                         unsigned long long& key, int& kpos, float& m1, float& m2) {
    m1 = __builtin_inff();
    m2 = __builtin_inff();
    int lo0, hi0, lo1 = 0, hi1 = 0;
    row_range(gp, cell_start, qx, qy, qz, cx, cy, cz, rho, t0, lo0, hi0);   // (both bound loads are issued before either is used)
    if constexpr (TWO) row_range(gp, cell_start, qx, qy, qz, cx, cy, cz, rho, t1, lo1, hi1);
    const int n0 = hi0 - lo0, total = n0 + (hi1 - lo1);
    const int last = hi1 > lo1 ? hi1 - 1 : hi0 - 1;   // the point an exhausted slot repeats (masked below)
    for (int e = 0; e < total; e += U) {
        int at[U];
        float4 pt[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int f = e + j;
            if constexpr (TWO) at[j] = f >= total ? last : f < n0 ? lo0 + f : lo1 + (f - n0);
            else at[j] = min(lo0 + f, hi0 - 1);
        }
#pragma unroll
        for (int j = 0; j < U; ++j) pt[j] = sorted[at[j]];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const unsigned long long kk = point_key<FMA>(pt[j], qx, qy, qz);
            if (kk < key) { key = kk; kpos = at[j]; }
            const float dd = e + j < total ? __uint_as_float((unsigned)(kk >> 32)) : __builtin_inff();
            m2 = __builtin_amdgcn_fmed3f(m1, m2, dd);
            m1 = fminf(m1, dd);
        }
    }
}

// ---- spatial order for the SOURCES: same cell order as the target, original index in .w ------------------
// Sources are scattered into cell order with atomics (arbitrary order inside a cell) and then every element
// computes its rank among the elements of its cell by original index, which makes the final order -- and
// with it every f64 sum of the run -- independent of the atomics' arrival order.  A rigid ICP update keeps
// neighbours neighbours, so ONE sort per registration keeps the queries of a wave / workgroup / XCD in the
// same few cells (L1 / L2 hits instead of Infinity-Cache trips) for all iterations.
__global__ __launch_bounds__(256) void grid_rank_fix_kernel(const float4* __restrict__ tmp, int n, GridParams gp,
                                                            const int32_t* __restrict__ start /* the source half: start[c] - shift */, int shift,
                                                            float4* __restrict__ out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const float4 p = tmp[j];
    const int cx = cell_coord(p.x, gp.ox, gp.inv_h, gp.gx), cy = cell_coord(p.y, gp.oy, gp.inv_h, gp.gy),
              cz = cell_coord(p.z, gp.oz, gp.inv_h, gp.gz);
    const int c = (cz * gp.gy + cy) * gp.gx + cx;
    const int lo = start[c] - shift, hi = start[c + 1] - shift;
    const int me = __float_as_int(p.w);
    int rank = 0;
    for (int k = lo; k < hi; ++k) rank += __float_as_int(tmp[k].w) < me ? 1 : 0;
    out[lo + rank] = p;
}

// Both cell lists of a single pair: count -> scan (2 launches) -> scatter -> rank fix = 5 launches (one pair of lists used to
// be 10).  d_counts: 2 * ncells zeroed ints (zero at rest: left zeroed again); d_start: 2 * ncells + 1 ints, target starts
// in [0, ncells], source starts + nt behind; d_src is read and, after the rank fix, rewritten in cell order (d_tmp: scratch).
void launch_grid_build_pair(hipStream_t st, const float4* d_tgt, int nt, float4* d_src, int ns, const GridParams& gp, int32_t* d_counts,
                            int32_t* d_start, int32_t* d_block_sums, float4* d_sorted, float4* d_tmp) {
    const int ncells = gp.gx * gp.gy * gp.gz;
    const int nbt = (nt + 255) / 256, nbs = (ns + 255) / 256;
    hipLaunchKernelGGL(grid_count2_kernel, dim3(nbt + nbs), dim3(256), 0, st, d_tgt, nt, nbt, (const float4*)d_src, ns, gp, d_counts);
    launch_scan(st, d_counts, 2 * ncells, d_start, d_block_sums);
    hipLaunchKernelGGL(grid_scatter2_kernel, dim3(nbt + nbs), dim3(256), 0, st, d_tgt, nt, nbt, (const float4*)d_src, ns, gp, d_counts,
                       (const int32_t*)d_start, d_sorted, d_tmp);
    hipLaunchKernelGGL(grid_rank_fix_kernel, dim3(nbs), dim3(256), 0, st, (const float4*)d_tmp, ns, gp, (const int32_t*)d_start + ncells, nt,
                       d_src);
}

// ---- phase A of the fused pass, per source (shared by grid_pass_kernel and gridb_pass_kernel) -------------------------
// KEEP last pass's winner w without searching, or ask for a search.  Per source the kernels carry {B, acc}: at the position
// where the source was last searched every target other than w was at true distance >= B; acc bounds how far the source has
// moved since (sum of the per-pass displacements, rounded up).  By the triangle inequality every other target is now at true
// distance >= B - acc, so its COMPUTED squared distance (5 roundings: relative error < 6 * 2^-24) exceeds (B - acc)^2 * 0.99999.
// If the winner's computed distance is below that, w wins again, strictly -- no tie to break -- exactly the key a search would
// return.  (1e-7: a displacement component below 1.1e-19 squares to zero; over a million passes that is 2e-13 unaccounted
// for, 2e-6 of the smallest room accepted -- inside the 1e-5 margin.  A lane with less room searches.)
// (A second candidate per source -- winner and runner-up, B bounding all the others -- was built and measured in round 3: 8x
// fewer searches, and no faster: a workgroup with ANY walker pays the same chain of dependent round trips, and at two or
// three walkers per row almost every workgroup still has one; the heavier bookkeeping cost C2 20 %.  The pair-resident
// engine, whose searches cost LDS reads instead of round trips, keeps two: kss_resident.hip.)
struct SkipOut {
    unsigned long long key;     // the previous winner's key at the new position (~0: none)
    int kpos;                   // -2: "the last winner, whose coordinates are in nn_win" (-1: none)
    float rho;                  // a walker's pruning radius
    float acc;                  // a kept winner: the distance covered since the search, this pass included
    bool walker;
};
template <bool FMA>
__device__ __forceinline__ SkipOut skip_test(bool qok, float qx, float qy, float qz, float pold_x, float pold_y, float pold_z, const float4& prevp,
                                             const float2& st, bool chained_k, float skin_frac, float h) {
    SkipOut o;
    o.key = ~0ull; o.kpos = -1; o.rho = __builtin_inff(); o.acc = 0.f; o.walker = false;
    if (!qok) return o;
    o.walker = true;
    if (__float_as_uint(prevp.w) != ~0u) {   // prevp = the last winner's coordinates, .w = its index in the pair's target
        o.key = point_key<FMA>(prevp, qx, qy, qz);
        o.kpos = -2;
        const float d0 = __uint_as_float((unsigned)(o.key >> 32));
        const float mx = qx - pold_x, my = qy - pold_y, mz = qz - pold_z;
        const float moved = __builtin_amdgcn_sqrtf((mx * mx + my * my) + mz * mz);
        const float acc = (st.y + moved) * 1.00001f;
        const float room = st.x - acc;
        if (chained_k && skin_frac >= 0.f && room > 1e-7f && (room * room) * 0.99999f > d0) {
            o.walker = false;              // w again
            o.acc = acc;
        } else {
            // the walk prunes with the winner's distance grown by a skin (a fraction of the cell edge): what it then proves
            // about the other targets leaves room for the next passes.  A source that is still moving by more than a quarter
            // of the skin per pass would not profit: plain radius.
            const float skin = fmaxf(skin_frac, 0.f) * h, grown = __builtin_amdgcn_sqrtf(d0) + skin;
            o.rho = !chained_k || moved * 4.f <= skin ? fmaxf(d0, grown * grown) : d0;
        }
    }
    return o;
}

// ---- phase C of the fused pass, per source: its correspondence as 16 f64 columns (KSS_NSUMS layout), the wave's totals of
// those columns in the canonical order of kss_device.hpp (+ counts by ballot, + the two FULL columns) into its row of shw ----
template <bool FULL>
__device__ __forceinline__ void wave_row(bool have, bool kept, bool fell_back, float qx, float qy, float qz, const float4& win, double d2d, double* shw_row) {
    const int lane = threadIdx.x & 63;
    double col[16];
    {
        const double px = kept ? (double)qx : 0.0, py = kept ? (double)qy : 0.0, pz = kept ? (double)qz : 0.0;
        const double tx = kept ? (double)win.x : 0.0, ty = kept ? (double)win.y : 0.0, tz = kept ? (double)win.z : 0.0;
        col[0] = px; col[1] = py; col[2] = pz; col[3] = tx; col[4] = ty; col[5] = tz;
        col[6] = px * tx; col[7] = px * ty; col[8] = px * tz;
        col[9] = py * tx; col[10] = py * ty; col[11] = py * tz;
        col[12] = pz * tx; col[13] = pz * ty; col[14] = pz * tz;
        col[15] = kept ? d2d : 0.0;
    }
    wave_tree16(col);
    const unsigned long long mk = __builtin_amdgcn_ballot_w64(kept), mf = __builtin_amdgcn_ballot_w64(fell_back);
    double extra = 0.0;
    if constexpr (FULL) extra = wave_tree2(d2d, have ? sqrt(d2d) : 0.0);   // lane 0: sum of all d2, lane 32: sum of sqrt(d2)
    if ((lane & 15) == 0) {
        const int q = lane >> 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) shw_row[1 + 4 * q + j] = col[j];
        if (lane == 0) {
            shw_row[0] = (double)__builtin_popcountll(mk);
            shw_row[NSUMS - 1] = (double)__builtin_popcountll(mf);
            shw_row[17] = extra;
        }
        if (lane == 32) shw_row[18] = extra;
    }
}

// Phase B of the fused pass (grid_pass_kernel, gridb_pass_kernel): the first lanes of the workgroup serve the `nwalk`
// walkers whose lane numbers are in s_wl and whose requests are in their hand-over columns (s_ent), and overwrite every
// column with the answer.  NT = threads of the workgroup, WQ = columns of the range queue.
template <bool FMA, bool BATCH, int WQ, int NT, int U>
__device__ __forceinline__ void serve_walkers(const PassArgs& a, const GridPairDev& pr, const int32_t* __restrict__ cs,
                                              const float4* __restrict__ sorted, int w, int nwalk, int2 (*rowq)[WQ],
                                              unsigned (*s_ent)[PASS_BS], const unsigned short* s_wl) {
#define KSS_STAMP(k) do { if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
    enum { WK_X, WK_Y, WK_Z, WK_IDX, WK_D2, WK_R, WK_POS, WK_FL = WK_POS, WK_COLS };   // (position code in the request, flags in the answer: one column)
    const GridParams& gp = pr.gp;
    const int lane = threadIdx.x & 63;
    // The lanes that serve share the walkers: each group of L lanes walks disjoint rows of one walker's 3x3x3 block (so
    // "distinct points" still holds) and merges key / position / the two smallest distances by DPP -- the dependent chain of
    // a search drops from ~4 walk steps to ~1, on lanes the compaction has left idle anyway.
    // few walkers: one row per lane (16 lanes per walker) or one or two (8 lanes per walker, lane 0 also row 8): row_walk, no
    // range queue; more walkers: 4 lanes or one lane per walker through block_walk and its WQ queue columns
    // (the 8-lane row form only in the batch kernels: inlined into the single-pair kernels as a third walk body it cost C2 5 %
    // -- 78.1k vs 74.0k iterations/s on one box -- for a case, 33-64 walkers in a workgroup, that C2 never sees)
    const bool row16 = 16 * nwalk <= NT, row8 = BATCH && !row16 && 8 * nwalk <= NT, rows = row16 || row8;
    const int L = row16 ? 16 : row8 ? 8 : 4 * nwalk <= WQ ? 4 : 1;
    const int nserve = rows ? NT : (WQ < NT ? WQ : NT);   // lanes that search
    for (int rb = 0; rb < nwalk; rb += nserve / L)
    if ((int)threadIdx.x < nserve && (int)(threadIdx.x & ~63u) < (nwalk - rb) * L) {   // wave-uniform
        const int sub = (int)threadIdx.x & (L - 1);
        const int wj = rb + (L == 16 ? (int)threadIdx.x >> 4 : L == 8 ? (int)threadIdx.x >> 3 : L == 4 ? (int)threadIdx.x >> 2 : (int)threadIdx.x);
        const bool wk = wj < nwalk;
        // rows by lane of a quad: {0, 2, 8}, {4, 6}, {1, 7}, {3, 5} (the corner rows 0, 2, 6, 8 are the ones most often pruned)
        const unsigned rowmask = L == 1 ? 0x1ffu : sub == 0 ? 0x105u : sub == 1 ? 0x050u : sub == 2 ? 0x082u : 0x028u;
        float wx = 0.f, wy = 0.f, wz = 0.f, wrho = __builtin_inff(), bnew = 0.f;
        unsigned long long wkey = ~0ull;
        int wpos = -1, owner = 0, evl = 0, rfin = 0;   // rfin: the shell the search ended with (0: it did not)
        bool done = true, wfell = false;
        if (wk) {
            owner = (int)s_wl[wj];
            wx = __uint_as_float(s_ent[WK_X][owner]); wy = __uint_as_float(s_ent[WK_Y][owner]); wz = __uint_as_float(s_ent[WK_Z][owner]);
            wkey = ((unsigned long long)s_ent[WK_D2][owner] << 32) | (unsigned long long)s_ent[WK_IDX][owner];
            wpos = (int)s_ent[WK_POS][owner];
            wrho = __uint_as_float(s_ent[WK_R][owner]);
            done = false;
        }
        if (wk) {
            const int cx = cell_coord(wx, gp.ox, gp.inv_h, gp.gx), cy = cell_coord(wy, gp.oy, gp.inv_h, gp.gy),
                      cz = cell_coord(wz, gp.oz, gp.inv_h, gp.gz);
            // ---- r = 1: the 3x3x3 block, pruned by rho ----
            KSS_STAMP(5);
            float m1, m2;
            if (row16) row_walk<FMA, U, false>(gp, cs, sorted, wx, wy, wz, cx, cy, cz, wrho, sub < 9 ? sub : -1, -1, wkey, wpos, m1, m2);
            else if (BATCH && row8) row_walk<FMA, U, true>(gp, cs, sorted, wx, wy, wz, cx, cy, cz, wrho, sub, sub == 0 ? 8 : -1, wkey, wpos, m1, m2);
            else evl = block_walk<FMA, WQ, U>(gp, cs, sorted, wx, wy, wz, cx, cy, cz, wrho, rowmask, rowq, wkey, wpos, m1, m2);
            if (L > 1) {   // (uniform) all lanes of a group are walkers of the same source: merge by DPP
#define KSS_GROUP_MERGE(X)                                                                                                          \
do {                                                                                                                            \
    const unsigned olo = (unsigned)X((int)(unsigned)wkey), ohi = (unsigned)X((int)(unsigned)(wkey >> 32));                      \
    const int opos = X(wpos);                                                                                                   \
    const float o1 = __int_as_float(X(__float_as_int(m1))), o2 = __int_as_float(X(__float_as_int(m2)));                         \
    const unsigned long long okey = ((unsigned long long)ohi << 32) | olo;                                                      \
    if (okey < wkey) { wkey = okey; wpos = opos; }                                                                            \
    m2 = fminf(fmaxf(m1, o1), fminf(m2, o2)); /* two smallest of the union of two sorted pairs */                               \
    m1 = fminf(m1, o1);                                                                                                         \
} while (0)
                auto x1 = [](int v) { return __builtin_amdgcn_update_dpp(0, v, 0xb1, 0xf, 0xf, false); };   // quad_perm [1,0,3,2]
                auto x2 = [](int v) { return __builtin_amdgcn_update_dpp(0, v, 0x4e, 0xf, 0xf, false); };   // quad_perm [2,3,0,1]
                // lane ^ 4: row_shl:4 into the lanes with bit 2 clear, row_shr:4 into the others; lane ^ 8: row_ror:8
                auto x4 = [](int v) { const int r = __builtin_amdgcn_update_dpp(v, v, 0x104, 0xf, 0x5, false); return __builtin_amdgcn_update_dpp(r, v, 0x114, 0xf, 0xa, false); };
                auto x8 = [](int v) { return __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false); };
                KSS_GROUP_MERGE(x1);
                KSS_GROUP_MERGE(x2);
                if (L >= 8) KSS_GROUP_MERGE(x4);
                if (L == 16) KSS_GROUP_MERGE(x8);
#undef KSS_GROUP_MERGE
            }
            float face2 = __builtin_inff();    // squared distance to the faces of the block the search ended with
            for (int r = 1; r <= gp.rcap; ++r) {
                if (r > 1) {   // shell r: (2r+1)^2 rows
                    const int wd = 2 * r + 1;
                    const int x0 = max(cx - r, 0), x1 = min(cx + r, gp.gx - 1);
                    for (int t = 0; t < wd * wd; ++t) {
                        const int dz = t / wd - r, dy = t % wd - r;
                        const int z = cz + dz, y = cy + dy;
                        if (z < 0 || z >= gp.gz || y < 0 || y >= gp.gy) continue;
                        const int row = (z * gp.gy + y) * gp.gx;
                        if (dz == -r || dz == r || dy == -r || dy == r) {
                            // a row on the shell's y/z faces: the whole x extent is new
                            scan_range<FMA>(sorted, cs[row + x0], cs[row + x1 + 1], wx, wy, wz, wkey, wpos);
                        } else {
                            // interior row of shell r: only its two x end cells are new
                            if (cx - r >= 0) scan_range<FMA>(sorted, cs[row + cx - r], cs[row + cx - r + 1], wx, wy, wz, wkey, wpos);
                            if (cx + r < gp.gx) scan_range<FMA>(sorted, cs[row + cx + r], cs[row + cx + r + 1], wx, wy, wz, wkey, wpos);
                        }
                    }
                }
                const float best = __uint_as_float((unsigned)(wkey >> 32));
                // distance from the query to the faces of the visited block; faces on the grid border are open
                float b = __builtin_inff();
                if (cx - r > 0) b = fminf(b, wx - (gp.ox + (float)(cx - r) * gp.h));
                if (cx + r < gp.gx - 1) b = fminf(b, (gp.ox + (float)(cx + r + 1) * gp.h) - wx);
                if (cy - r > 0) b = fminf(b, wy - (gp.oy + (float)(cy - r) * gp.h));
                if (cy + r < gp.gy - 1) b = fminf(b, (gp.oy + (float)(cy + r + 1) * gp.h) - wy);
                if (cz - r > 0) b = fminf(b, wz - (gp.oz + (float)(cz - r) * gp.h));
                if (cz + r < gp.gz - 1) b = fminf(b, (gp.oz + (float)(cz + r + 1) * gp.h) - wz);
                const float bs = b - gp.eps;
                if (b == __builtin_inff()) done = true;                              // the whole grid has been visited
                else if (bs > 0.f && best < bs * bs * 0.999999f) { done = true; face2 = bs * bs; }   // every unvisited point is strictly farther
                if (done) { rfin = r; break; }
            }
            // What the r = 1 walk has proven about every target but the winner: walked ones are at computed distance >= m2,
            // pruned ones beyond rho, unvisited ones beyond the block's faces.  (Shells r > 1 are not tracked: B = 0.)
            if (done && rfin == 1 && wkey != ~0ull)
                bnew = fminf(__builtin_amdgcn_sqrtf(fminf(fminf(m2, wrho), face2) * 0.99999f) * 0.999999f, 1e30f);
        }
        if constexpr (BATCH) {
            // bounded fallback: brute force over the pair's targets for the lanes the shells did not resolve
            if (__builtin_amdgcn_ballot_w64(!done) != 0ull) {              // wave-uniform
                const float4* __restrict__ tp = a.tgt4 + pr.tgt_base;      // original order; uniform addresses below
                for (int j = 0; j < pr.tgt_n; ++j) {
                    const float4 q = tp[j];
                    const float dx = wx - q.x, dy = wy - q.y, dz = wz - q.z;
                    float d;
                    if constexpr (FMA) d = __builtin_fmaf(dx, dx, __builtin_fmaf(dy, dy, dz * dz));
                    else d = (dx * dx + dy * dy) + dz * dz;
                    const unsigned long long kk = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned long long)(unsigned)j;
                    if (!done && kk < wkey) { wkey = kk; wpos = -1; }   // (no position in `sorted` known: no warm start next time)
                }
                wfell = !done;
                done = true;
            }
        }
        if (wk && sub == 0) {              // (the lanes of a group agree on everything from the merge on)
            bool whave = false;
            float4 ww = make_float4(0.f, 0.f, 0.f, 0.f);
            const int io = pr.src_base + (w - pr.row_base) * PASS_BS + owner;
            if (done) {
                whave = wkey != ~0ull;
                // the winner's coordinates: carried through the r = 1 walk (the shells beyond do not carry them: read back; after
                // the in-wave sweep: read from the pair's target in original order)
                if (whave) ww = wpos >= 0 ? sorted[wpos] : wpos == -2 ? a.nn_win[io] : a.tgt4[pr.tgt_base + (int)(unsigned)(wkey & 0xffffffffull)];
            } else {   // single pair: resolved by the brute-force list pass through atomicMin, sums by the SEARCH = false launch
                a.keys[io] = ~0ull;
                const int ls = atomicAdd(a.list_count, 1);
                a.list[ls] = io;
                wfell = true;   // counted in column 19 of the row, as the batch counts its in-wave fallbacks
            }
            a.nn_win[io] = make_float4(ww.x, ww.y, ww.z, __uint_as_float(whave ? (unsigned)wkey : ~0u));
            a.nn_state[io] = make_float2(wfell ? 0.f : bnew, 0.f);
            s_ent[WK_X][owner] = __float_as_uint(ww.x); s_ent[WK_Y][owner] = __float_as_uint(ww.y); s_ent[WK_Z][owner] = __float_as_uint(ww.z);
            s_ent[WK_IDX][owner] = (unsigned)wkey; s_ent[WK_D2][owner] = (unsigned)(wkey >> 32);
            s_ent[WK_FL][owner] = (whave ? 1u : 0u) | (wfell ? 2u : 0u);
            s_ent[WK_R][owner] = __float_as_uint(wfell ? 0.f : bnew);
        }
        KSS_STAMP(8);
        if (a.stamps) {   // slot 10: evaluations, 11: evaluation slots issued (slowest lane, rounded up to 8, x 64)
            int evs = evl, mx = evl;
#pragma unroll
            for (int m = 32; m > 0; m >>= 1) { evs += __shfl_xor(evs, m, 64); mx = max(mx, __shfl_xor(mx, m, 64)); }
            if (lane == 0) {
                atomicAdd(&a.stamps[(size_t)blockIdx.x * 16 + 10], (unsigned long long)evs);
                atomicAdd(&a.stamps[(size_t)blockIdx.x * 16 + 11], (unsigned long long)(((mx + 7) / 8) * 8 * 64));
            }
        }
    }
#undef KSS_STAMP
}

// =============================================================================================
// The fused cell-list pass: ONE kernel for a single pair (C2 / C4) and for a batch of pairs (C3 / C5).
//   - one workgroup = one CHUNK of 512 consecutive cell-sorted sources of one pair, one lane per source; workgroups
//     are laid out pair by pair and handed to XCDs in contiguous runs (blockIdx % 8 remap), so the workgroups of a
//     pair -- and with them that pair's cell list -- stay in one XCD's L2;
//   - search in three phases.  A: every lane moves its source and tries to keep its last winner (the skip test above);
//     the lanes that must search ("walkers", a few per cent near convergence) queue up in LDS.  B: the first lanes of
//     the workgroup serve the walkers, 16 / 8 / 4 lanes per walker when there are few (one row or a few rows of the
//     3x3x3 block each, merged by DPP), one lane each otherwise: last winner first, the block pruned by its distance
//     (block_walk / row_walk), further shells up to GridParams::rcap, then the fallback -- BATCH: the wave sweeps the
//     pair's targets by brute force for its unresolved lanes (uniform addresses, no barrier); single pair: the source
//     goes to a device list that nn_sweep_kernel<LIST> resolves, after which a SEARCH = false launch of this kernel
//     (same rows, same order) redoes the sums from the stored winners.  C: every lane picks its answer up from LDS;
//   - sums: every lane contributes its correspondence to 16 f64 columns (+ 2 in FULL mode: sum of all d2 and of
//     sqrt(d2) for getFitnessScore / PCR_QM; counts are ballots), reduced in the canonical order of kss_device.hpp:
//     in-wave tree by permlane swaps + DPP (no LDS, no barrier), 8 wave totals through LDS, one row per workgroup;
//   - rows are handed over with write-through sc1 stores + vmcnt(0) + workgroup barrier + ONE agent-scope ticket on the
//     pair's counter (MI355X_MICROARCH.md, "Valid forms", table row 1); the workgroup drawing the pair's last ticket
//     adds the rows in the canonical order and publishes the 20 sums of the pair as {bits(sum), seq} 16-byte stores into
//     host-mapped memory, where the host spins on the sequence numbers (no completion flag, no stream sync);
//   - a pair of a single chunk skips the hand-over (0.0 + row: the same bits as the general path).
// The result of a pair is a function of its own clouds only: alone or in a batch of any size, bit for bit.
// =============================================================================================
#ifndef KSS_BATCH_WAVES
#define KSS_BATCH_WAVES 6   // waves per SIMD the batched variant is compiled for (3 workgroups per CU; 8 spills)
#endif
#ifndef KSS_SINGLE_WAVES
#define KSS_SINGLE_WAVES 1
#endif
template <bool FMA, bool FULL, bool BATCH, bool SEARCH, bool CHAIN>
__global__ __launch_bounds__(PASS_BS, BATCH ? KSS_BATCH_WAVES : KSS_SINGLE_WAVES) void grid_pass_kernel(const PassArgs a) {
    static_assert(!CHAIN || (!BATCH && SEARCH), "chained launches: single pair, search passes");
    // diagnostic stamps (100 MHz s_memrealtime): [block*16 + {0 start, 13 first loads in, 9 gate open, 14 phase A done,
    // 5 phase B entered, 8 answered, 15 phase B done, 1 searched, 2 row ready, 3 ticketed, 4 result stored (last
    // workgroup)}]; counts: 10 = distance evaluations of the r = 1 block, 11 = evaluation slots, 12 = walkers
#define KSS_STAMP(k) do { if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
    KSS_STAMP(0);
    constexpr int BS = PASS_BS;
    const int per_xcd = (int)gridDim.x / 8;                                   // the launcher makes gridDim.x a multiple of 8
    const int w = ((int)blockIdx.x % 8) * per_xcd + (int)blockIdx.x / 8;      // bijection on [0, gridDim.x): XCD-contiguous runs
    if (w >= a.total_rows) return;
    int pi = 0;
    if constexpr (BATCH) pi = a.row_pair[w];
    const GridPairDev pr = BATCH ? a.pairs[pi] : a.pair0;
    PairState ps = a.ps0;
    if constexpr (BATCH) {
        ps = a.state[pi];
        if (!ps.active) return;            // uniform: a converged pair costs one table read per workgroup
    }
    const GridParams& gp = pr.gp;
    const int32_t* __restrict__ cs = a.cell_start + pr.cell_base;
    const float4* __restrict__ sorted = a.sorted;
    constexpr int WQ = 256;                // lanes that search in phase B
    __shared__ int2 rowq[9][WQ];           // block_walk's per-lane queue of point ranges
    __shared__ double shw[BS / 64][NSUMS];
    __shared__ int s_last;
    __shared__ int sh_ps[16];
    __shared__ int s_nwalk;
    // per-lane hand-over columns.  A walker leaves its request {query, key and position of the last winner, pruning radius}
    // in its own column and its lane number in s_wl; the lanes of phase B overwrite the column with the answer {winner, key,
    // flags}.  A lane that keeps its winner writes the answer itself.  (Through LDS rather than registers: nothing but the
    // lane's identity stays live across the search: registers are what limits how many workgroups share a CU.)
    __shared__ unsigned s_ent[7][BS];
    __shared__ unsigned short s_wl[BS];
    enum { WK_X, WK_Y, WK_Z, WK_IDX, WK_D2, WK_R, WK_POS, WK_FL = WK_POS, WK_COLS };   // (position code in the request, flags in the answer: one column)
    static_assert(sizeof(PairState) == 64, "the gated launch reads the transform record as 16 dwords");
    double (*shf)[NSUMS] = reinterpret_cast<double (*)[NSUMS]>(&rowq[0][0]);   // the last workgroup's group totals: rowq is dead by then (two barriers later)
    static_assert(sizeof(double) * PASS_FG * NSUMS <= sizeof(int2) * 9 * WQ, "shf must fit inside rowq");

    const int local = (w - pr.row_base) * BS + (int)threadIdx.x;
    const bool valid = local < pr.src_n;
    const int i = pr.src_base + local;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // this lane's correspondence
    bool have = false, fell_back = false;
    float qx = 0.f, qy = 0.f, qz = 0.f, d2 = 0.f;
    float4 win = make_float4(0.f, 0.f, 0.f, 0.f);
    // what does not depend on the transform is fetched first: the source, where its last winner sits, that point, and the
    // skip state {B, acc} (see below)
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f), prevp = make_float4(0.f, 0.f, 0.f, __uint_as_float(~0u));
    float2 st = make_float2(0.f, 0.f);
    if (valid) {
        p = SEARCH ? a.src_in[i] : a.src_out[i];
        // (one round trip: nothing here depends on another load.  The sums-only pass always follows a search pass that has
        // just written every source's winner.)
        if (!SEARCH || a.use_prev) { prevp = a.nn_win[i]; st = a.nn_state[i]; }
    }
    // A CHAINED launch (a.chain_len > 1, single pair, kss_engine.hip) runs that many ICP iterations in this one kernel: every
    // pass of the loop below waits for its own transform at the gate, and the source, its winner and its skip state stay in
    // registers from one pass to the next.  What a chain saves is everything BETWEEN two launches -- the end-of-kernel
    // cache write-back, the dispatch of the next kernel (~4 us together) and its first loads -- which is more than a
    // third of an iteration at C2.  Buffers alternate (src_out <-> src_in, both work buffers then), stamps and sequence
    // numbers count up with the pass.  Everything is still written to memory each pass: whoever continues after a chain
    // (a plain launch after a fallback, the fitness pass) finds the same state a sequence of launches would have left.
    for (int step = 0;; ++step) {
    const bool last_step = !CHAIN || step + 1 >= a.chain_len;   // (CHAIN is a template parameter so that the profilers list the two forms apart)
    const unsigned long long seq_k = a.seq + (unsigned long long)step;
    const int gate_k = a.gate_seq != 0 ? a.gate_seq + step : 0;
    unsigned int* const gate_ptr = a.gate_dev + 32 * ((a.gate_slot + step) & 1);
    float4* const src_out_k = (step & 1) ? a.src_alt : a.src_out;
    // where the source was when its skip state was last brought up to date: the buffer this pass reads (chained == 1), or --
    // the fitness pass, which starts over from the original cloud -- the work buffer the pair's last pass wrote, named by the
    // host in the pair's state (chained == 2, pad[0] = 1 + buffer; 0: unknown)
    const bool last_known = a.chained == 2 && ps.pad[0] > 0;
    const bool chained_k = step > 0 || a.chained == 1 || last_known;
    const bool has_prev = __float_as_uint(prevp.w) != ~0u;   // prevp = the last winner's coordinates, .w = its index in the pair's target
    if constexpr (SEARCH) {
        __syncthreads();                   // (the shared arrays of the previous pass are dead)
        if (threadIdx.x == 0) s_nwalk = 0;
        __syncthreads();
    }
    KSS_STAMP(13);
    if constexpr (!BATCH && SEARCH) {
        if (gate_k != 0) {
            // GATED launch (kss_engine.hip): this kernel was enqueued while the previous iteration was still running, before
            // its transform existed.  It polls five self-validating 16-byte granules {3 words, stamp} in DEVICE memory (one
            // 80-byte request per poll and workgroup) until all carry this launch's stamp.  Who writes them: on a large-BAR
            // system the HOST, straight into that (fine-grained) device memory with five 16-byte stores; otherwise workgroup 0,
            // which polls a host-mapped 64-byte record (one cache line, one PCIe read per poll; the host writes the stamp
            // last and a line is read as a whole) and re-publishes it.  (All 200 workgroups polling HOST memory at once were
            // measured: 45 us per iteration.)  Both polls are bounded: a host that never answers makes the kernel leave
            // without publishing, which the host's own wait reports.  pad[0] != 0: cancelled.  (The loads above are
            // already in flight while this waits.)
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            if (a.state && w == 0 && threadIdx.x < 16) {
                // the ONE workgroup that asks the host: lanes 0-15 read the 64-byte record as one request per poll
                const int* w32 = reinterpret_cast<const int*>(a.state);
                int v = 0, n = 0;
                bool ok = true;
                for (;;) {
                    v = __hip_atomic_load(w32 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    if (__shfl(v, 15, 64) == gate_k) break;      // word 15 = pad[1], written last by the host
                    __builtin_amdgcn_s_sleep(1);
                    if (++n > a.gate_polls / 2) { ok = false; break; }
                }
                if (ok) {
                    // hand it to the other workgroups through device memory as five self-validating 16-byte granules
                    // {word 3g, 3g+1, 3g+2, stamp} (one sc1 store each: a granule is never seen torn, so no flag and no
                    // ordering between the stores is needed)
                    const int g = threadIdx.x;
                    const int w0 = __shfl(v, min(3 * g, 15), 64), w1 = __shfl(v, min(3 * g + 1, 15), 64), w2 = __shfl(v, min(3 * g + 2, 15), 64);
                    if (g < 5) {
                        u32x4 o;
                        o.x = (unsigned)w0; o.y = (unsigned)w1; o.z = (unsigned)w2; o.w = (unsigned)gate_k + kss_mix3(o.x, o.y, o.z);
                        unsigned int* dst = gate_ptr + 4 * g;
                        asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(o) : "memory");
                    }
                }
            }
            if (threadIdx.x < 8) {         // every workgroup: lanes 0-4 poll the five granules (one 80-byte request per poll)
                const unsigned int* src = gate_ptr + 4 * min((int)threadIdx.x, 4);
                u32x4 v;
                int n = 0;
                bool ok = true;
                for (;;) {
                    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(src) : "memory");   // system scope: the writer may be the host
                    if (__builtin_amdgcn_ballot_w64(v.w - kss_mix3(v.x, v.y, v.z) == (unsigned)gate_k) == 0xffull) break;   // (stamp + check of the three words: a granule seen torn fails and is read again)
                    __builtin_amdgcn_s_sleep(1);
                    if (++n > a.gate_polls) { ok = false; break; }
                }
                if (threadIdx.x < 5) { sh_ps[3 * threadIdx.x] = (int)v.x; sh_ps[3 * threadIdx.x + 1] = (int)v.y; sh_ps[3 * threadIdx.x + 2] = (int)v.z; }
                if (threadIdx.x == 0) { sh_ps[15] = gate_k; s_last = ok ? 1 : 0; }
            }
            __syncthreads();
            if (!s_last) return;
            int* d32 = reinterpret_cast<int*>(&ps);
#pragma unroll
            for (int k = 0; k < 16; ++k) d32[k] = sh_ps[k];
            if (ps.pad[0] != 0) return;
            __syncthreads();               // s_last is reused by the ticket below
            KSS_STAMP(9);
        }
    }
    // ---- the search ------------------------------------------------------------------------------------------
    // Phase A, every lane: move the source, then try to KEEP last pass's winner without searching (skip_test).  Lanes that
    // cannot ("walkers") are compacted into the first waves of the workgroup (phase B): near convergence a few per cent of
    // the sources walk, so ONE wave per workgroup pays for the walk instead of eight.
    unsigned long long key = ~0ull;
    float pold_x = p.x, pold_y = p.y, pold_z = p.z;   // where the source was in the last pass
    if (SEARCH && last_known && valid) {
        const float4 pl = (ps.pad[0] == 1 ? a.src_last0 : a.src_last1)[i];
        pold_x = pl.x; pold_y = pl.y; pold_z = pl.z;
    }
    if (valid) {
        if constexpr (SEARCH) {
            if (ps.apply) {   // pcl transformCloud with the previous iteration's Matrix4f, Eigen order, float, no fma
                const float x = p.x, y = p.y, z = p.z;
                p.x = ((ps.m[0] * x + ps.m[1] * y) + ps.m[2] * z) + ps.m[3];
                p.y = ((ps.m[4] * x + ps.m[5] * y) + ps.m[6] * z) + ps.m[7];
                p.z = ((ps.m[8] * x + ps.m[9] * y) + ps.m[10] * z) + ps.m[11];
            }
            src_out_k[i] = p;
        }
        qx = p.x; qy = p.y; qz = p.z;
        // a non-finite query matches nothing (its distances are NaN; as an integer key NaN bits would beat every real one)
        const bool qok = (qx - qx) == 0.f && (qy - qy) == 0.f && (qz - qz) == 0.f;
        if constexpr (!SEARCH) {
            // sums only: the winner is where the search left it, or what the list pass found
            if (has_prev) {
                win = prevp;
                key = point_key<FMA>(win, qx, qy, qz);
            } else if (qok) {
                key = a.keys[i];
                if (key != ~0ull) win = a.tgt4[pr.tgt_base + (int)(unsigned)(key & 0xffffffffull)];
            }
            have = key != ~0ull;
        }
    }
    if constexpr (SEARCH) {
        bool walker = false;
        float acc_keep = 0.f;
        {
            const bool qok = valid && (qx - qx) == 0.f && (qy - qy) == 0.f && (qz - qz) == 0.f;
            const SkipOut so = skip_test<FMA>(qok, qx, qy, qz, pold_x, pold_y, pold_z, prevp, st, chained_k, a.skin, gp.h);
            walker = so.walker;
            key = so.key;
            acc_keep = so.acc;
            unsigned fl = 0u;
            if (qok && !walker) {
                fl = 1u;
                a.nn_state[i] = make_float2(st.x, so.acc);
            } else if (valid && !qok) {
                a.nn_win[i] = make_float4(0.f, 0.f, 0.f, __uint_as_float(~0u));
                a.nn_state[i] = make_float2(0.f, 0.f);
            }
            // the lane's column: a walker's request, or the answer of a lane that needs no search
            const float4 c4 = walker ? make_float4(qx, qy, qz, 0.f) : prevp;
            s_ent[WK_X][threadIdx.x] = __float_as_uint(c4.x); s_ent[WK_Y][threadIdx.x] = __float_as_uint(c4.y); s_ent[WK_Z][threadIdx.x] = __float_as_uint(c4.z);
            s_ent[WK_IDX][threadIdx.x] = (unsigned)key; s_ent[WK_D2][threadIdx.x] = (unsigned)(key >> 32);
            s_ent[WK_POS][threadIdx.x] = walker ? (unsigned)so.kpos : fl;   // (a walker: where its last winner is; the others: their flags)
            s_ent[WK_R][threadIdx.x] = __float_as_uint(so.rho);
        }
        // compaction: walkers take consecutive slots (wave by wave in arrival order, lane order inside a wave)
        const unsigned long long wm = __builtin_amdgcn_ballot_w64(walker);
        if (wm != 0ull) {
            int base = 0;
            if (lane == 0) base = atomicAdd(&s_nwalk, (int)__builtin_popcountll(wm));
            base = __builtin_amdgcn_readfirstlane(base);
            if (walker) s_wl[base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(wm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)wm, 0u))] = (unsigned short)threadIdx.x;
        }
        KSS_STAMP(14);
        __syncthreads();
        const int nwalk = s_nwalk;
        if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)blockIdx.x * 16 + 12] = (unsigned long long)nwalk;
        serve_walkers<FMA, BATCH, WQ, BS, BATCH ? 4 : 8>(a, pr, cs, sorted, w, nwalk, rowq, s_ent, s_wl);
        KSS_STAMP(15);
        __syncthreads();
        // Phase C: every lane reads its column
        {
            const unsigned fl = s_ent[WK_FL][threadIdx.x];
            have = (fl & 1u) != 0u; fell_back = (fl & 2u) != 0u;
            win = make_float4(__uint_as_float(s_ent[WK_X][threadIdx.x]), __uint_as_float(s_ent[WK_Y][threadIdx.x]), __uint_as_float(s_ent[WK_Z][threadIdx.x]), 0.f);
            key = ((unsigned long long)s_ent[WK_D2][threadIdx.x] << 32) | (unsigned long long)s_ent[WK_IDX][threadIdx.x];
            if (CHAIN && !last_step) {     // chained launch: what the next pass would otherwise load
                if (walker) {
                    prevp = make_float4(win.x, win.y, win.z, __uint_as_float(have ? (unsigned)key : ~0u));
                    st = make_float2(__uint_as_float(s_ent[WK_R][threadIdx.x]), 0.f);
                } else if (have) {
                    st.y = acc_keep;       // (a kept winner: same point, same bound, more distance covered)
                } else {
                    prevp.w = __uint_as_float(~0u);
                    st = make_float2(0.f, 0.f);
                }
            }
        }
        if constexpr (BATCH) {   // (the query comes back from memory rather than staying in registers across the search)
            p = make_float4(0.f, 0.f, 0.f, 0.f);
            if (valid) p = src_out_k[i];
            qx = p.x; qy = p.y; qz = p.z;
        }
    }
    if (valid && have) {
        d2 = __uint_as_float((unsigned)(key >> 32));
        const int oi = __float_as_int(p.w);   // original source index (sources are in cell order)
        if (a.idx_out) a.idx_out[oi] = (int)(unsigned)(key & 0xffffffffull);
        if (a.d2_out) a.d2_out[oi] = d2;
    }
    KSS_STAMP(1);

    // ---- the row of this chunk, canonical order (kss_device.hpp) ----
    const double d2d = have ? (double)d2 : 0.0;
    const bool kept = have && !(d2d > a.max_d2);   // PCL: `if (distance[0] > max_dist_sqr) continue;`
    wave_row<FULL>(have, kept, fell_back, qx, qy, qz, win, d2d, shw[wave]);
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x < NSUMS)
#pragma unroll
        for (int ww = 0; ww < BS / 64; ++ww) r += shw[ww][threadIdx.x];
    KSS_STAMP(2);

    double v = 0.0 + r;                    // a single-chunk pair: exactly what the general path below computes
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    if (pr.n_rows > 1 && a.tagged_rows) {
        // Single pair whose workgroups are all resident at once (<= 256 of them): rows travel as self-validating 16-byte
        // granules {bits(sum), launch sequence number} -- one sc1 store per sum, no drain, no ticket -- and the pair's FIRST
        // workgroup, once its own row is out, polls every row until all carry this launch's number, then adds them in the
        // canonical order.  Saves the store-acknowledge wait and the ticket round trip of the general protocol below
        // (~2 us of a 11 us launch); every other workgroup is done the moment its row is stored.  The poll is bounded.
        if (threadIdx.x < NSUMS) {
            const unsigned long long rb = (unsigned long long)__double_as_longlong(r);
            u32x4 o;
            o.x = (unsigned)rb; o.y = (unsigned)(rb >> 32); o.z = (unsigned)seq_k; o.w = (unsigned)(seq_k >> 32);
            unsigned long long* dst = reinterpret_cast<unsigned long long*>(a.rows) + 2 * ((int64_t)w * NSUMS + threadIdx.x);
            asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(o) : "memory");
        }
        if (w != pr.row_base) {            // uniform
            if (last_step) return;
            continue;
        }
        KSS_STAMP(3);
        if (threadIdx.x == 0) s_last = 1;
        __syncthreads();
        {
            const int g = threadIdx.x / NSUMS, c = threadIdx.x % NSUMS;
            const unsigned long long* __restrict__ rows2 = reinterpret_cast<const unsigned long long*>(a.rows) + 2 * (int64_t)pr.row_base * NSUMS;
            if (g < PASS_FG) {
                double acc = 0.0;
                for (int k = g; k < pr.n_rows; k += 8 * PASS_FG) {
                    const unsigned long long* p[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) p[j] = rows2 + 2 * ((int64_t)(k + j * PASS_FG < pr.n_rows ? k + j * PASS_FG : k) * NSUMS + c);
                    u32x4 t0, t1, t2, t3, t4, t5, t6, t7;
                    int spins = 0;
                    for (;;) {
                        asm volatile(
                            "global_load_dwordx4 %0, %8, off sc1\n\tglobal_load_dwordx4 %1, %9, off sc1\n\t"
                            "global_load_dwordx4 %2, %10, off sc1\n\tglobal_load_dwordx4 %3, %11, off sc1\n\t"
                            "global_load_dwordx4 %4, %12, off sc1\n\tglobal_load_dwordx4 %5, %13, off sc1\n\t"
                            "global_load_dwordx4 %6, %14, off sc1\n\tglobal_load_dwordx4 %7, %15, off sc1\n\t"
                            "s_waitcnt vmcnt(0)"
                            : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)
                            : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7])
                            : "memory");
                        const unsigned lo = (unsigned)seq_k, hi = (unsigned)(seq_k >> 32);
                        const bool ok = t0.z == lo && t0.w == hi && t1.z == lo && t1.w == hi && t2.z == lo && t2.w == hi && t3.z == lo && t3.w == hi &&
                                        t4.z == lo && t4.w == hi && t5.z == lo && t5.w == hi && t6.z == lo && t6.w == hi && t7.z == lo && t7.w == hi;
                        if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;      // wave-uniform: the loads stay convergent
                        if (++spins > (1 << 20)) { s_last = 0; break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    const u32x4 tt[8] = {t0, t1, t2, t3, t4, t5, t6, t7};
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (k + j * PASS_FG < pr.n_rows) acc += __longlong_as_double((long long)(((unsigned long long)tt[j].y << 32) | tt[j].x));
                }
                shf[g][c] = acc;
            }
        }
        __syncthreads();
        if (!s_last) return;               // a row never arrived: leave without publishing, the host's wait reports it
        v = 0.0;
        if (threadIdx.x < NSUMS)
            for (int gg = 0; gg < PASS_FG; ++gg) v += shf[gg][threadIdx.x];
    } else if (pr.n_rows > 1) {
        if (threadIdx.x < NSUMS) {
            __hip_atomic_store(&a.rows[(int64_t)w * NSUMS + threadIdx.x], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if (threadIdx.x == 0) s_last = atomicAdd(&a.tickets[pi], 1) == pr.n_rows - 1;
        __syncthreads();
        KSS_STAMP(3);
        if (!s_last) return;
        // pair total: lane (g, c) adds rows g, g + 25, ... of column c -- eight sc1 loads in flight per batch -- and the 25
        // group totals are then added in group order
        {
            const int g = threadIdx.x / NSUMS, c = threadIdx.x % NSUMS;
            const double* __restrict__ rows = a.rows + (int64_t)pr.row_base * NSUMS;
            if (g < PASS_FG) {
                double acc = 0.0;
                // (rows in flight per batch: a 1M-point pair has 1954 rows, 78 per lane -- at eight per batch the pair's last
                // workgroup spent ten dependent round trips here, a fifth of the iteration; the order of the additions is the same)
                constexpr int RB = BATCH ? 8 : 32;
                for (int k = g; k < pr.n_rows; k += RB * PASS_FG) {
                    double t[RB];
#pragma unroll
                    for (int j = 0; j < RB; ++j)
                        t[j] = k + j * PASS_FG < pr.n_rows ? __hip_atomic_load(&rows[(int64_t)(k + j * PASS_FG) * NSUMS + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
#pragma unroll
                    for (int j = 0; j < RB; ++j) acc += t[j];
                }
                shf[g][c] = acc;
            }
        }
        __syncthreads();
        v = 0.0;
        if (threadIdx.x < NSUMS)
            for (int gg = 0; gg < PASS_FG; ++gg) v += shf[gg][threadIdx.x];
        if (threadIdx.x == 0) a.tickets[pi] = 0;   // re-armed for the next launch (stream order makes it visible)
    }
    if (threadIdx.x < NSUMS) {
        if (!FULL && (threadIdx.x == 17 || threadIdx.x == 18)) v = 0.0;
        // {bits(sum), seq} as ONE aligned 16-byte system-scope store per sum into host-mapped memory: the host accepts a
        // slot when its sequence number matches, so no flag has to be ordered after the data (that ordering would cost a
        // write-acknowledge round trip over PCIe) and no L2 write-back fence is needed
        const unsigned long long vb = (unsigned long long)__double_as_longlong(v);
        u32x4 o;
        o.x = (unsigned)vb; o.y = (unsigned)(vb >> 32); o.z = (unsigned)seq_k; o.w = kss_mix3(o.x, o.y, o.z);   // (check word: a slot seen torn is read again)
        if (a.test_torn && threadIdx.x == 5) {   // test hook: first a slot whose data does not fit its check word, the real one a while later
            u32x4 bad = o;
            bad.x ^= 0x00100000u;
            unsigned long long* dst0 = a.pub + 2 * ((int64_t)pi * NSUMS + threadIdx.x);
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" ::"v"(dst0), "v"(bad) : "memory");
            for (int k = 0; k < 400; ++k) __builtin_amdgcn_s_sleep(64);
        }
        unsigned long long* dst = a.pub + 2 * ((int64_t)pi * NSUMS + threadIdx.x);
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(o) : "memory");
        KSS_STAMP(4);
    }
    if (last_step) break;
    }   // passes of a chained launch
#undef KSS_STAMP
}

// =============================================================================================
// The batched form of the fused pass with TWO sources per lane: a workgroup of 256 threads (4 waves) serves the same
// row of 512 sources as a 512-thread workgroup of grid_pass_kernel<BATCH = true>, same phases, same hand-over columns, same
// sums in the same order -- lane l of wave v holds the sources that lane l of "waves" v and v + 4 hold there, and the wave
// tree runs once per half -- so every bit of the result is the same.  Why: the batched pass is RESIDENCY-bound (a
// workgroup's life is a chain of dependent memory round trips, ~12 us, while its instructions fit in ~1.5 us); with half
// the waves per row twice as many rows are in flight per CU for the same registers.  WQ = columns of the range queue:
// 256 for the first pass of a registration (every source searches), 128 afterwards (a few per cent do), which is what
// lets five workgroups share a CU's LDS.
// =============================================================================================
#ifndef KSS_BATCH2_WAVES
#define KSS_BATCH2_WAVES 5   // workgroups of 256 per CU the two-sources-per-lane form is compiled for (6: 80 VGPRs, 4 spilled: no faster)
#endif
template <bool FMA, bool FULL, int WQ>
__global__ __launch_bounds__(256, KSS_BATCH2_WAVES) void gridb_pass_kernel(const PassArgs a) {
#define KSS_STAMP(k) do { if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
    KSS_STAMP(0);
    constexpr int BS = PASS_BS, NT = 256, SPL = BS / NT;
    static_assert(SPL == 2 && WQ <= NT, "two sources per lane");
    const int per_xcd = (int)gridDim.x / 8;
    const int w = ((int)blockIdx.x % 8) * per_xcd + (int)blockIdx.x / 8;
    if (w >= a.total_rows) return;
    const int pi = a.row_pair[w];
    const GridPairDev pr = a.pairs[pi];
    const PairState ps = a.state[pi];
    if (!ps.active) return;
    const GridParams& gp = pr.gp;
    const int32_t* __restrict__ cs = a.cell_start + pr.cell_base;
    const float4* __restrict__ sorted = a.sorted;
    __shared__ int2 rowq[9][WQ];
    __shared__ double shw[BS / 64][NSUMS];
    __shared__ int s_last;
    __shared__ int s_nwalk;
    __shared__ unsigned s_ent[7][BS];
    __shared__ unsigned short s_wl[BS];
    enum { WK_X, WK_Y, WK_Z, WK_IDX, WK_D2, WK_R, WK_POS, WK_FL = WK_POS, WK_COLS };   // (position code in the request, flags in the answer: one column)
    constexpr int SHF_ROWS = (int)(sizeof(int2) * 9 * WQ / (sizeof(double) * NSUMS));
    static_assert(SHF_ROWS >= PASS_FG, "shf must fit inside rowq");
    double (*shf)[NSUMS] = reinterpret_cast<double (*)[NSUMS]>(&rowq[0][0]);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

    bool valid[SPL];
    int idx[SPL];
    float4 p[SPL], prevp[SPL];
    float2 st[SPL];
#pragma unroll
    for (int k = 0; k < SPL; ++k) {
        const int slot = k * NT + (int)threadIdx.x;
        const int local = (w - pr.row_base) * BS + slot;
        valid[k] = local < pr.src_n;
        idx[k] = pr.src_base + local;
        p[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        prevp[k] = make_float4(0.f, 0.f, 0.f, __uint_as_float(~0u));
        st[k] = make_float2(0.f, 0.f);
        if (valid[k]) {
            p[k] = a.src_in[idx[k]];
            if (a.use_prev) { prevp[k] = a.nn_win[idx[k]]; st[k] = a.nn_state[idx[k]]; }
        }
    }
    if (threadIdx.x == 0) s_nwalk = 0;
    __syncthreads();
    KSS_STAMP(13);
    const bool last_known = a.chained == 2 && ps.pad[0] > 0;
    const bool chained_k = a.chained == 1 || last_known;
    // ---- phase A (see grid_pass_kernel) ----
    bool walker[SPL];
#pragma unroll
    for (int k = 0; k < SPL; ++k) {
        const int slot = k * NT + (int)threadIdx.x;
        const int i = idx[k];
        float pold_x = p[k].x, pold_y = p[k].y, pold_z = p[k].z;
        if (last_known && valid[k]) {
            const float4 pl = (ps.pad[0] == 1 ? a.src_last0 : a.src_last1)[i];
            pold_x = pl.x; pold_y = pl.y; pold_z = pl.z;
        }
        if (valid[k]) {
            if (ps.apply) {   // pcl transformCloud with the previous iteration's Matrix4f, Eigen order, float, no fma
                const float x = p[k].x, y = p[k].y, z = p[k].z;
                p[k].x = ((ps.m[0] * x + ps.m[1] * y) + ps.m[2] * z) + ps.m[3];
                p[k].y = ((ps.m[4] * x + ps.m[5] * y) + ps.m[6] * z) + ps.m[7];
                p[k].z = ((ps.m[8] * x + ps.m[9] * y) + ps.m[10] * z) + ps.m[11];
            }
            a.src_out[i] = p[k];
        }
        const float qx = p[k].x, qy = p[k].y, qz = p[k].z;
        const bool qok = valid[k] && (qx - qx) == 0.f && (qy - qy) == 0.f && (qz - qz) == 0.f;
        const SkipOut so = skip_test<FMA>(qok, qx, qy, qz, pold_x, pold_y, pold_z, prevp[k], st[k], chained_k, a.skin, gp.h);
        walker[k] = so.walker;
        unsigned fl = 0u;
        if (qok && !so.walker) {
            fl = 1u;
            a.nn_state[i] = make_float2(st[k].x, so.acc);
        } else if (valid[k] && !qok) {
            a.nn_win[i] = make_float4(0.f, 0.f, 0.f, __uint_as_float(~0u));
            a.nn_state[i] = make_float2(0.f, 0.f);
        }
        const float4 c4 = so.walker ? make_float4(qx, qy, qz, 0.f) : prevp[k];
        s_ent[WK_X][slot] = __float_as_uint(c4.x); s_ent[WK_Y][slot] = __float_as_uint(c4.y); s_ent[WK_Z][slot] = __float_as_uint(c4.z);
        s_ent[WK_IDX][slot] = (unsigned)so.key; s_ent[WK_D2][slot] = (unsigned)(so.key >> 32);
        s_ent[WK_POS][slot] = so.walker ? (unsigned)so.kpos : fl;
        s_ent[WK_R][slot] = __float_as_uint(so.rho);
        const unsigned long long wm = __builtin_amdgcn_ballot_w64(walker[k]);
        if (wm != 0ull) {
            int base = 0;
            if (lane == 0) base = atomicAdd(&s_nwalk, (int)__builtin_popcountll(wm));
            base = __builtin_amdgcn_readfirstlane(base);
            if (walker[k]) s_wl[base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(wm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)wm, 0u))] = (unsigned short)slot;
        }
    }
    KSS_STAMP(14);
    __syncthreads();
    const int nwalk = s_nwalk;
    if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)blockIdx.x * 16 + 12] = (unsigned long long)nwalk;
    // (points in flight per walk step: 8 in the first pass, whose wide range queue leaves four workgroups per CU anyway)
    serve_walkers<FMA, true, WQ, NT, WQ == 256 ? 8 : 4>(a, pr, cs, sorted, w, nwalk, rowq, s_ent, s_wl);
    KSS_STAMP(15);
    __syncthreads();
    // ---- phase C + the row of this chunk in the canonical order: half k of the lanes plays waves 4k .. 4k + 3 ----
#pragma unroll
    for (int k = 0; k < SPL; ++k) {
        const int slot = k * NT + (int)threadIdx.x;
        const unsigned fl = s_ent[WK_FL][slot];
        const bool have = (fl & 1u) != 0u, fell_back = (fl & 2u) != 0u;
        const float4 win = make_float4(__uint_as_float(s_ent[WK_X][slot]), __uint_as_float(s_ent[WK_Y][slot]), __uint_as_float(s_ent[WK_Z][slot]), 0.f);
        const unsigned long long key = ((unsigned long long)s_ent[WK_D2][slot] << 32) | (unsigned long long)s_ent[WK_IDX][slot];
        const float qx = p[k].x, qy = p[k].y, qz = p[k].z;
        float d2 = 0.f;
        if (valid[k] && have) {
            d2 = __uint_as_float((unsigned)(key >> 32));
            const int oi = __float_as_int(p[k].w);
            if (a.idx_out) a.idx_out[oi] = (int)(unsigned)(key & 0xffffffffull);
            if (a.d2_out) a.d2_out[oi] = d2;
        }
        const double d2d = have ? (double)d2 : 0.0;
        const bool kept = have && !(d2d > a.max_d2);
        wave_row<FULL>(have, kept, fell_back, qx, qy, qz, win, d2d, shw[k * (NT / 64) + wave]);   // the wave this half plays
    }
    KSS_STAMP(1);
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x < NSUMS)
#pragma unroll
        for (int ww = 0; ww < BS / 64; ++ww) r += shw[ww][threadIdx.x];
    KSS_STAMP(2);
    double v = 0.0 + r;
    if (pr.n_rows > 1) {
        if (threadIdx.x < NSUMS) {
            __hip_atomic_store(&a.rows[(int64_t)w * NSUMS + threadIdx.x], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if (threadIdx.x == 0) s_last = atomicAdd(&a.tickets[pi], 1) == pr.n_rows - 1;
        __syncthreads();
        KSS_STAMP(3);
        if (!s_last) return;
        // pair total as in grid_pass_kernel: (group g, column c) adds rows g, g + 25, ...; the 25 group totals in order
        {
            const double* __restrict__ rows = a.rows + (int64_t)pr.row_base * NSUMS;
            for (int t = (int)threadIdx.x; t < PASS_FG * NSUMS; t += NT) {
                const int g = t / NSUMS, c = t % NSUMS;
                double acc = 0.0;
                for (int k = g; k < pr.n_rows; k += 8 * PASS_FG) {
                    double tt[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        tt[j] = k + j * PASS_FG < pr.n_rows ? __hip_atomic_load(&rows[(int64_t)(k + j * PASS_FG) * NSUMS + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc += tt[j];
                }
                shf[g][c] = acc;
            }
        }
        __syncthreads();
        v = 0.0;
        if (threadIdx.x < NSUMS)
            for (int gg = 0; gg < PASS_FG; ++gg) v += shf[gg][threadIdx.x];
        if (threadIdx.x == 0) a.tickets[pi] = 0;
    }
    if (threadIdx.x < NSUMS) {
        if (!FULL && (threadIdx.x == 17 || threadIdx.x == 18)) v = 0.0;
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        const unsigned long long vb = (unsigned long long)__double_as_longlong(v);
        u32x4 o;
        o.x = (unsigned)vb; o.y = (unsigned)(vb >> 32); o.z = (unsigned)a.seq; o.w = kss_mix3(o.x, o.y, o.z);
        unsigned long long* dst = a.pub + 2 * ((int64_t)pi * NSUMS + threadIdx.x);
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(o) : "memory");
        KSS_STAMP(4);
    }
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
// tgt4, padding skipped, .w = pair-relative index) or sources (BY_SRC = true: .w = global source index).
// The pair of a slot: ONE binary search per workgroup for its first slot (uniform: scalar loads), then a short walk per
// lane (a per-lane search over 1024 pairs was ten dependent loads per point: 0.4 of the 0.5 ms this kernel took at C3).
// The scatter counts each cell back DOWN (slots start + count-1 .. start): no cursor array, counts end zeroed.
template <bool SCATTER, bool BY_SRC>
__global__ __launch_bounds__(256) void gridb_bin_kernel(const float4* __restrict__ pts, int total, const GridPairDev* __restrict__ pairs,
                                                        int npairs, int32_t* __restrict__ counts, const int32_t* __restrict__ start,
                                                        float4* __restrict__ out) {
    const int i0 = blockIdx.x * blockDim.x;
    int pi = pair_of(i0, pairs, npairs, BY_SRC);
    const int i = i0 + threadIdx.x;
    if (i >= total) return;
    while (pi + 1 < npairs && (BY_SRC ? pairs[pi + 1].src_base : pairs[pi + 1].tgt_base) <= i) ++pi;
    const GridPairDev pr = pairs[pi];
    const int local = i - (BY_SRC ? pr.src_base : pr.tgt_base);
    if (local >= (BY_SRC ? pr.src_n : pr.tgt_n)) return;   // sentinel padding of the target layout
    float4 p = pts[i];
    const GridParams& gp = pr.gp;
    const int cx = cell_coord(p.x, gp.ox, gp.inv_h, gp.gx), cy = cell_coord(p.y, gp.oy, gp.inv_h, gp.gy),
              cz = cell_coord(p.z, gp.oz, gp.inv_h, gp.gz);
    const int cell = pr.cell_base + (cz * gp.gy + cy) * gp.gx + cx;
    if constexpr (SCATTER) {
        const int pos = start[cell] + atomicSub(&counts[cell], 1) - 1;
        p.w = __int_as_float(BY_SRC ? i : local);
        out[pos] = p;
    } else {
        atomicAdd(&counts[cell], 1);
    }
}

// deterministic in-cell order for the sources (see grid_rank_fix_kernel)
__global__ __launch_bounds__(256) void gridb_rank_fix_kernel(const float4* __restrict__ tmp, int total, const GridPairDev* __restrict__ pairs,
                                                             int npairs, const int32_t* __restrict__ start, float4* __restrict__ out) {
    int pi = pair_of((int)(blockIdx.x * blockDim.x), pairs, npairs, true);   // uniform; sorted positions stay inside the pair's source segment
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= total) return;
    while (pi + 1 < npairs && pairs[pi + 1].src_base <= j) ++pi;
    const float4 p = tmp[j];
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
                                int total_cells, int32_t* d_counts, int32_t* d_start, int32_t* d_block_sums, float4* d_sorted) {
    // d_counts: zero at rest (kss_ctx.hpp); the count-down scatters below leave it zeroed again
    const dim3 grid((total_tgt_pad + 255) / 256), block(256);
    hipLaunchKernelGGL((gridb_bin_kernel<false, false>), grid, block, 0, st, d_tgt, total_tgt_pad, d_pairs, npairs, d_counts, (const int32_t*)nullptr, (float4*)nullptr);
    launch_scan(st, d_counts, total_cells, d_start, d_block_sums);
    hipLaunchKernelGGL((gridb_bin_kernel<true, false>), grid, block, 0, st, d_tgt, total_tgt_pad, d_pairs, npairs, d_counts, (const int32_t*)d_start, d_sorted);
}

// sources: d_src (total_src float4) -> d_out in (pair, cell, original index) order; d_tmp is scratch (d_out may be d_src).
// Runs right after the target build over the same cells: the counts are zero again.
void launch_gridb_sort_sources(hipStream_t st, const float4* d_src, int total_src, const GridPairDev* d_pairs, int npairs,
                               int total_cells, int32_t* d_counts, int32_t* d_start, int32_t* d_block_sums, float4* d_tmp, float4* d_out) {
    const dim3 grid((total_src + 255) / 256), block(256);
    hipLaunchKernelGGL((gridb_bin_kernel<false, true>), grid, block, 0, st, d_src, total_src, d_pairs, npairs, d_counts, (const int32_t*)nullptr, (float4*)nullptr);
    launch_scan(st, d_counts, total_cells, d_start, d_block_sums);
    hipLaunchKernelGGL((gridb_bin_kernel<true, true>), grid, block, 0, st, d_src, total_src, d_pairs, npairs, d_counts, (const int32_t*)d_start, d_tmp);
    hipLaunchKernelGGL(gridb_rank_fix_kernel, grid, block, 0, st, d_tmp, total_src, d_pairs, npairs, d_start, d_out);
}

// ---- one workgroup builds one pair's cell lists entirely in LDS --------------------------------------------------
// The global-atomic path above spends ~0.4 ms per bin launch at C3 (10M scattered atomics over a 110 MB count array,
// four launches) plus two 330 MB scans.  A C3-size pair has ~27k cells: its counters fit the CU's LDS, so ONE
// workgroup per pair does count -> scan -> scatter for the targets, then count -> scan -> scatter -> rank fix for the
// sources, with LDS atomics only; global memory sees each point once on the way in and once on the way out.
// Same outputs as the global path: cell_start (global positions in `sorted`), `sorted` (.w = index inside the pair's
// target; order inside a cell is arbitrary, the search's tie rule does not depend on it), sources in (cell, original
// index) order with .w = global source index.
constexpr int GB_THREADS = 1024;
constexpr int GB_MAX_CELLS = 36 * 1024;   // 144 KB of counters (+ the scan's wave totals): one workgroup per CU

int gridb_lds_max_cells(bool half) { return half ? 2 * GB_MAX_CELLS : GB_MAX_CELLS; }   // (half: 16-bit counters, CellCounters)

// The per-pair counters of gridb_build_pair_kernel, in dynamic LDS sized for the batch's largest grid.  HALF: 16-bit
// counters (every pair of the batch has fewer than 65536 targets and sources, so counts and positions fit) -- half the LDS,
// two workgroups per CU at C3 instead of one.  Plain reads and writes address the halves directly (ds_read_u16 /
// ds_write_b16); the atomics go through the containing 32-bit word (no carry: a half never reaches 65536).
template <bool HALF>
struct CellCounters {
    unsigned int* w;
    __device__ __forceinline__ int get(int c) const { return HALF ? (int)reinterpret_cast<const unsigned short*>(w)[c] : (int)w[c]; }
    __device__ __forceinline__ void set(int c, int v) const {
        if (HALF) reinterpret_cast<unsigned short*>(w)[c] = (unsigned short)v; else w[c] = (unsigned)v;
    }
    __device__ __forceinline__ int add1(int c) const {   // returns the value before
        if (HALF) { const int sh = 16 * (c & 1); return (int)((atomicAdd(&w[c >> 1], 1u << sh) >> sh) & 0xffffu); }
        return (int)atomicAdd(&w[c], 1u);
    }
    __device__ __forceinline__ void zero(int ncells) const {
        const int nw = HALF ? (ncells + 1) / 2 : ncells;
        for (int c = threadIdx.x; c < nw; c += GB_THREADS) w[c] = 0u;
    }
};

// exclusive scan of the counters [0 .. n) in place by the whole workgroup; returns the total in every thread.
// store_global != nullptr: also writes base + start to store_global[c] and base + total to store_global[n].
template <bool HALF>
__device__ __forceinline__ int lds_exclusive_scan(const CellCounters<HALF>& cnt, int n, int32_t* wave_tot, int32_t* __restrict__ store_global, int base) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int chunk = ((n + GB_THREADS - 1) / GB_THREADS + 1) & ~1;   // even: a thread owns whole words of 16-bit counters
    const int lo = min(tid * chunk, n), hi = min(lo + chunk, n);
    int s = 0;
    for (int c = lo; c < hi; ++c) s += cnt.get(c);
    int incl = s;   // inclusive scan of the per-thread totals inside the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int woff = 0, total = 0;
    for (int k = 0; k < GB_THREADS / 64; ++k) {
        const int v = wave_tot[k];
        if (k < wave) woff += v;
        total += v;
    }
    int run = woff + incl - s;
    for (int c = lo; c < hi; ++c) {
        const int v = cnt.get(c);
        cnt.set(c, run);
        run += v;
    }
    __syncthreads();
    if (store_global) {
        // the starts go out in a pass of their own, consecutive lanes consecutive cells (stored from the loop above, each lane
        // wrote ITS ~28 consecutive cells: 64 partly written lines per store instruction, 3.4x the table in HBM write traffic)
        for (int c = tid; c < n; c += GB_THREADS) store_global[c] = base + cnt.get(c);
        if (tid == 0) store_global[n] = base + total;
        __syncthreads();   // (the caller turns the starts into cursors next)
    }
    return total;
}

template <bool HALF>
__global__ __launch_bounds__(GB_THREADS) void gridb_build_pair_kernel(const float4* __restrict__ tgt4, const float4* src_in, float4* src_tmp,
                                                                      float4* src_out, const GridPairDev* __restrict__ pairs,
                                                                      int32_t* __restrict__ cell_start, float4* __restrict__ sorted) {
    extern __shared__ unsigned int gb_dyn[];
    __shared__ int32_t wave_tot[GB_THREADS / 64];
    const CellCounters<HALF> cnt{gb_dyn};
    const GridPairDev pr = pairs[blockIdx.x];
    const GridParams& gp = pr.gp;
    const int ncells = gp.gx * gp.gy * gp.gz;
    const int tid = threadIdx.x;
    auto cell_of = [&](const float4& p) {
        return (cell_coord(p.z, gp.oz, gp.inv_h, gp.gz) * gp.gy + cell_coord(p.y, gp.oy, gp.inv_h, gp.gy)) * gp.gx + cell_coord(p.x, gp.ox, gp.inv_h, gp.gx);
    };
    // Every pass over the points takes them GB_U at a time per lane: the loads of a batch are issued together and the LDS
    // atomics follow (one load, one atomic per trip was a chain of ~10 memory round trips per pass, 130 us per pair).
    constexpr int GB_U = 8;
    // ---- targets ----
    cnt.zero(ncells);
    __syncthreads();
    const float4* __restrict__ tp = tgt4 + pr.tgt_base;
    for (int k0 = tid; k0 < pr.tgt_n; k0 += GB_THREADS * GB_U) {
        float4 p[GB_U];
#pragma unroll
        for (int u = 0; u < GB_U; ++u) p[u] = tp[min(k0 + u * GB_THREADS, pr.tgt_n - 1)];
#pragma unroll
        for (int u = 0; u < GB_U; ++u)
            if (k0 + u * GB_THREADS < pr.tgt_n) cnt.add1(cell_of(p[u]));
    }
    __syncthreads();
    lds_exclusive_scan<HALF>(cnt, ncells, wave_tot, cell_start + pr.cell_base, pr.sorted_base);
    float4* __restrict__ so = sorted + pr.sorted_base;
    for (int k0 = tid; k0 < pr.tgt_n; k0 += GB_THREADS * GB_U) {
        float4 p[GB_U];
#pragma unroll
        for (int u = 0; u < GB_U; ++u) p[u] = tp[min(k0 + u * GB_THREADS, pr.tgt_n - 1)];
#pragma unroll
        for (int u = 0; u < GB_U; ++u) {
            const int k = k0 + u * GB_THREADS;
            if (k < pr.tgt_n) {
                const int pos = cnt.add1(cell_of(p[u]));   // the start array doubles as the cursor
                p[u].w = __int_as_float(k);
                so[pos] = p[u];
            }
        }
    }
    __syncthreads();
    // ---- sources: same cells ----
    cnt.zero(ncells);
    __syncthreads();
    const float4* sp = src_in + pr.src_base;
    for (int k0 = tid; k0 < pr.src_n; k0 += GB_THREADS * GB_U) {
        float4 p[GB_U];
#pragma unroll
        for (int u = 0; u < GB_U; ++u) p[u] = sp[min(k0 + u * GB_THREADS, pr.src_n - 1)];
#pragma unroll
        for (int u = 0; u < GB_U; ++u)
            if (k0 + u * GB_THREADS < pr.src_n) cnt.add1(cell_of(p[u]));
    }
    __syncthreads();
    lds_exclusive_scan<HALF>(cnt, ncells, wave_tot, nullptr, 0);
    float4* tmp = src_tmp + pr.src_base;
    for (int k0 = tid; k0 < pr.src_n; k0 += GB_THREADS * GB_U) {
        float4 p[GB_U];
#pragma unroll
        for (int u = 0; u < GB_U; ++u) p[u] = sp[min(k0 + u * GB_THREADS, pr.src_n - 1)];
#pragma unroll
        for (int u = 0; u < GB_U; ++u) {
            const int k = k0 + u * GB_THREADS;
            if (k < pr.src_n) {
                const int pos = cnt.add1(cell_of(p[u]));   // afterwards the counter of cell c = END of cell c = start of cell c + 1
                p[u].w = __int_as_float(pr.src_base + k);
                tmp[pos] = p[u];
            }
        }
    }
    __syncthreads();   // (its release / acquire at workgroup scope makes the tmp stores visible to the loads below)
    // deterministic in-cell order: rank by original index (see grid_rank_fix_kernel)
    float4* out = src_out + pr.src_base;
    for (int j0 = tid; j0 < pr.src_n; j0 += GB_THREADS * GB_U) {
        float4 p[GB_U];
#pragma unroll
        for (int u = 0; u < GB_U; ++u) p[u] = tmp[min(j0 + u * GB_THREADS, pr.src_n - 1)];
#pragma unroll
        for (int u = 0; u < GB_U; ++u) {
            if (j0 + u * GB_THREADS >= pr.src_n) continue;
            const int c = cell_of(p[u]);
            const int lo = c > 0 ? cnt.get(c - 1) : 0, hi = cnt.get(c);
            const int me = __float_as_int(p[u].w);
            int rank = 0;
            for (int k = lo; k < hi; ++k) rank += __float_as_int(tmp[k].w) < me ? 1 : 0;
            out[lo + rank] = p[u];
        }
    }
}

// max_cells: the largest grid of the batch; half: no pair has 65536 or more targets or sources
bool launch_gridb_build_lds(hipStream_t st, const float4* d_tgt4, float4* d_src, float4* d_tmp, const GridPairDev* d_pairs, int npairs,
                            int32_t* d_cell_start, float4* d_sorted, int max_cells, bool half) {
    const size_t bytes = half ? (size_t)((max_cells + 1) / 2) * 4 : (size_t)max_cells * 4;
    // dynamic LDS beyond 64 KB has to be allowed per kernel AND per device: asked for on every launch (the attribute applies
    // to the device current at the call; a process-wide "done once" flag left a second device without it)
    {
        const int most = GB_MAX_CELLS * 4;
        const void* fn = half ? reinterpret_cast<const void*>(&gridb_build_pair_kernel<true>) : reinterpret_cast<const void*>(&gridb_build_pair_kernel<false>);
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, most) != hipSuccess) { (void)hipGetLastError(); return false; }
    }
    if (half)
        hipLaunchKernelGGL(gridb_build_pair_kernel<true>, dim3(npairs), dim3(GB_THREADS), bytes, st, d_tgt4, (const float4*)d_src, d_tmp, d_src, d_pairs,
                           d_cell_start, d_sorted);
    else
        hipLaunchKernelGGL(gridb_build_pair_kernel<false>, dim3(npairs), dim3(GB_THREADS), bytes, st, d_tgt4, (const float4*)d_src, d_tmp, d_src, d_pairs,
                           d_cell_start, d_sorted);
    return true;
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

int grid_pass_blocks(int total_rows) { return (total_rows + 7) / 8 * 8; }   // multiple of 8: the XCD remap is then a bijection

void launch_grid_pass(hipStream_t st, bool fma, bool full, bool batch, bool search, const PassArgs& a) {
    if (a.total_rows <= 0) return;
    const dim3 grid(grid_pass_blocks(a.total_rows)), block(PASS_BS);
#define KSS_PASS(F, U, B, S) hipLaunchKernelGGL((grid_pass_kernel<F, U, B, S, false>), grid, block, 0, st, a)
#define KSS_CHAIN(F, U) hipLaunchKernelGGL((grid_pass_kernel<F, U, false, true, true>), grid, block, 0, st, a)
    if (!search) {   // sums-only relaunch after the list pass: single pair, all 20 sums
        if (fma) KSS_PASS(true, true, false, false); else KSS_PASS(false, true, false, false);
    } else if (batch) {
        // two sources per lane (gridb_pass_kernel); the first pass of a registration (every source searches) with the wide
        // range queue.  KSS_BATCH_OLD: the one-source-per-lane form of grid_pass_kernel (A/B; same bits).
        static const bool old_form = getenv("KSS_BATCH_OLD") != nullptr;
        if (old_form) {
            if (fma) { if (full) KSS_PASS(true, true, true, true); else KSS_PASS(true, false, true, true); }
            else     { if (full) KSS_PASS(false, true, true, true); else KSS_PASS(false, false, true, true); }
        } else {
            const dim3 block2(256);
#define KSS_PASSB(F, U, Q) hipLaunchKernelGGL((gridb_pass_kernel<F, U, Q>), grid, block2, 0, st, a)
            if (!a.use_prev) {
                if (fma) { if (full) KSS_PASSB(true, true, 256); else KSS_PASSB(true, false, 256); }
                else     { if (full) KSS_PASSB(false, true, 256); else KSS_PASSB(false, false, 256); }
            } else {
                if (fma) { if (full) KSS_PASSB(true, true, 128); else KSS_PASSB(true, false, 128); }
                else     { if (full) KSS_PASSB(false, true, 128); else KSS_PASSB(false, false, 128); }
            }
#undef KSS_PASSB
        }
    } else if (a.chain_len > 1) {
        if (fma) { if (full) KSS_CHAIN(true, true); else KSS_CHAIN(true, false); }
        else     { if (full) KSS_CHAIN(false, true); else KSS_CHAIN(false, false); }
    } else {
        if (fma) { if (full) KSS_PASS(true, true, false, true); else KSS_PASS(true, false, false, true); }
        else     { if (full) KSS_PASS(false, true, false, true); else KSS_PASS(false, false, false, true); }
    }
#undef KSS_PASS
#undef KSS_CHAIN
}

}  // namespace kss
