// kss_resident.hip -- the PAIR-RESIDENT ICP kernel: one workgroup runs one whole registration (configs C3 / C5: batches
// of ModelNet40-scale pairs; the candidate batch of KSSICP_Registration, KSS_ICP.hpp:102-118).
//
// Why.  A 10k x 10k pair is 120 KB of target coordinates and 10 sources per lane of a 1024-lane workgroup: it FITS one
// CU of an MI355X -- 160 KB of LDS and a 512 KB register file -- so it never has to leave it.  The launch-per-pass
// engine (kss_grid.hip: gridb_pass_kernel) moves 64 B per source and pass through HBM plus the pair's cell list (0.9 GB
// per pass at C3) and lives through a chain of L2 round trips per row; here a registration reads its two clouds ONCE:
//   - LDS: the pair's targets in cell order as three float arrays (x, y, z), a 16-bit table of cell starts (x resolution
//     reduced by a power of two until it fits; visiting a superset of cells cannot change an exact search), the walkers'
//     request queue (aliased by the wave totals of the sums afterwards) and the pair's row sums;
//   - registers: per lane up to 10 sources -- position, where its last winner sits in the LDS arrays, and the room left
//     of the skip bound -- for all passes of the registration;
//   - per pass: (A) move every source by the transform of the previous pass and keep its winner if the skip bound still
//     proves it (the rule of kss_grid.hip, phase A, with B and acc folded into one conservative `room`); (B) the others
//     are searched in LDS -- the 3x3x3 block pruned by the winner's distance, further shells, and after GridParams::rcap
//     shells a sweep of the pair's targets shared by the 64 lanes of the wave -- a search costs LDS reads, no memory
//     round trip; (C) the 16 (+2) f64 correspondence columns are summed in the CANONICAL ORDER of kss_device.hpp (same
//     cell-sorted source order, same wave tree, same row and group order as grid_pass_kernel), so a pair gives the same
//     bits here, on the launch-per-pass engines and alone; the 20 sums are published to host-mapped memory as
//     {bits, sequence number, check} granules;
//   - the 3x3 SVD stays on the HOST (north star): the workgroup then polls ITS gate record -- five 16-byte granules
//     {3 words, stamp + check(3 words)} the host stores straight into fine-grained device memory through the BAR -- for
//     the next transform, the order to run the getFitnessScore() pass, or to stop.  Every poll is bounded: a host that
//     never answers makes the workgroup leave, which the host's own wait then reports (the caller falls back to the
//     launch-per-pass engine).  No workgroup ever waits for another workgroup, so nothing depends on co-residency: with
//     more pairs than CUs the later pairs simply start as earlier ones finish.
// Results are bit-identical to the other engines (tests/test_gpu_configs.py, tests/test_gpu_resident.py).
#pragma clang fp contract(off)

#include <hip/hip_runtime.h>

#include <type_traits>

#include "kss_internal.hpp"
#include "kss_device.hpp"

namespace kss {


typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct ResLds {
    const float* t3;                                     // targets of the pair in cell order: x, y, z of point k at [3k .. 3k + 2]
    const unsigned short* tab;                           // [rows][tw]: start of x-cell (j << xs) of the row, [tw - 1] = end of the row
    int tw, xs, nx;
};

// value of lane (l ^ 1), (l ^ 2) inside a quad (DPP, no LDS)
__device__ __forceinline__ int quad_x1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0xb1, 0xf, 0xf, false); }   // quad_perm [1,0,3,2]
__device__ __forceinline__ int quad_x2(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x4e, 0xf, 0xf, false); }   // quad_perm [2,3,0,1]
// ... (l ^ 4), (l ^ 8) inside a row of 16 lanes: row_shl:4 into the banks with bit 2 clear + row_shr:4 into the others; row_ror:8
__device__ __forceinline__ int row_x4(int v) { const int r = __builtin_amdgcn_update_dpp(v, v, 0x104, 0xf, 0x5, false); return __builtin_amdgcn_update_dpp(r, v, 0x114, 0xf, 0xa, false); }
__device__ __forceinline__ int row_x8(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false); }

constexpr unsigned RES_NOBITS = 0xffffffffu;   // "no distance": loses to every bit pattern of a d2 >= +0 (inf and NaN included)
constexpr unsigned RES_INFBITS = 0x7f800000u;

// One exact nearest-neighbour search, entirely from LDS, by LG = 1, 4 or 16 lanes (the forms of kss_grid.hip: serve_walkers).
// LG = 1: one lane per query, all nine rows of the 3x3x3 block (many queries: throughput counts, every lane has its own).
// LG = 4: the lanes of a quad take the rows {0, 2, 8}, {4, 6}, {1, 7}, {3, 5}.  LG = 16: lane r < 9 of the group walks row r
// -- with a few dozen searches per pass the workgroup waits for the longest chain of dependent LDS reads and instructions,
// not for throughput, and that chain is then ONE short row.  Groups merge by DPP.  The function is entered by whole waves
// (act: this lane has a query) because the last resort -- a sweep of all the pair's targets -- is shared by the 64 lanes.
// rho: a squared radius that is at least the squared distance of SOME target (phase A: of the source's last winner); rows and
// end cells beyond it are not read, exactly as row_range() of kss_grid.hip prunes.  rho = +inf (first pass): the lane starts
// with its own row and prunes the others with ITS best distance grown by the skin -- any radius that covers the best works.
// Returns the positions in the LDS arrays of the winner p1 (0xffff: none) and of a runner-up p2 (a point at the second
// smallest distance of the r = 1 walk; 0xffff: none), the bound B the walk proves for every target OTHER THAN THOSE TWO (0
// when it proves none) and whether the sweep was needed.  Ties in distance go to the lowest ORIGINAL target index
// (sg[pos].w), read from memory only when two computed distances are equal -- so the winner does not depend on the order of
// the points inside a cell, on which cells were visited, nor on how the rows were shared out.
template <bool FMA, int LG>
__device__ __forceinline__ void res_search(bool act, const GridParams& gp, const ResLds& L, int nt, const float4* __restrict__ sg, float skin,
                                           float qx, float qy, float qz, float rho, unsigned& out_w, float& out_b, unsigned& out_fell,
                                           unsigned long long* prof = nullptr) {
    unsigned long long tp_ = prof ? __builtin_amdgcn_s_memrealtime() : 0ull;
#define KSS_SLAP(k) do { if (prof) { const unsigned long long n_ = __builtin_amdgcn_s_memrealtime(); prof[k] += n_ - tp_; tp_ = n_; } } while (0)
    // the two best (distance bits, position) and the third smallest distance over the DISTINCT points walked
    unsigned b1 = RES_NOBITS, b2 = RES_NOBITS, m3 = RES_INFBITS;
    int p1 = -1, p2 = -1;
    bool done = true;
    float bnew = 0.f;
    unsigned fell = 0u;
    const int lane = threadIdx.x & 63, sub = lane & (LG - 1);
    auto tie_lower = [&](int k, int b) -> bool { return __float_as_uint(sg[k].w) < __float_as_uint(sg[b].w); };
    auto insert = [&](unsigned d, int k) {
        if (d < b1 || (d == b1 && (p1 < 0 || (k != p1 && tie_lower(k, p1))))) { m3 = min(b2, m3); b2 = b1; p2 = p1; b1 = d; p1 = k; }
        else if (k == p1) { }                                   // (shells r > 1 see the points of the block again)
        else if (d < b2) { m3 = min(b2, m3); b2 = d; p2 = k; }
        else m3 = min(m3, d);
    };
    // cell-ordered points [lo, hi): four per step -- their LDS reads and four distances first (slots past the end read the
    // points that follow, the arrays are padded; they are masked out), then ONE comparison of the step's smallest distance
    // with the runner-up: most steps change nothing but the third distance
    auto eval_range = [&](int lo, int hi) {
        for (int k = lo; k < hi; k += 4) {
            unsigned db[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float* t = L.t3 + 3 * (k + u);
                const unsigned d = __float_as_uint(dist2<FMA>(qx, qy, qz, t[0], t[1], t[2]));
                db[u] = k + u < hi ? d : RES_NOBITS;
            }
            const unsigned lo4 = min(min(db[0], db[1]), min(db[2], db[3]));
            if (lo4 <= b2) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (db[u] != RES_NOBITS) insert(db[u], k + u);
            } else {
                m3 = min(m3, lo4);
            }
        }
    };
    int cx = 0, cy = 0, cz = 0;
    if (act) {
        done = false;
        cx = cell_coord(qx, gp.ox, gp.inv_h, gp.gx); cy = cell_coord(qy, gp.oy, gp.inv_h, gp.gy); cz = cell_coord(qz, gp.oz, gp.inv_h, gp.gz);
        // squared slack-reduced gaps from the query to the neighbouring slabs (the expressions of kss_grid.hip: row_range)
        const float exl = fmaxf((qx - (gp.ox + (float)cx * gp.h)) - gp.eps, 0.f), exr = fmaxf(((gp.ox + (float)(cx + 1) * gp.h) - qx) - gp.eps, 0.f);
        const float eyl = fmaxf((qy - (gp.oy + (float)cy * gp.h)) - gp.eps, 0.f), eyr = fmaxf(((gp.oy + (float)(cy + 1) * gp.h) - qy) - gp.eps, 0.f);
        const float ezl = fmaxf((qz - (gp.oz + (float)cz * gp.h)) - gp.eps, 0.f), ezr = fmaxf(((gp.oz + (float)(cz + 1) * gp.h) - qz) - gp.eps, 0.f);
        const int xh1 = (1 << L.xs) - 1;
        // ---- r = 1: the 3x3x3 block, row by row, pruned by rho ----
        const float ey2[3] = {eyl * eyl, 0.f, eyr * eyr}, ez2[3] = {ezl * ezl, 0.f, ezr * ezr};
        auto walk_row = [&](int tyy, int tzz, float gy2, float gz2) {
            const int z = cz + tzz - 1, y = cy + tyy - 1;
            if (z < 0 || z >= gp.gz || y < 0 || y >= gp.gy) return;
            const float g2 = gy2 + gz2;
            if (rho < g2 * 0.999999f) return;
            const bool left = !(rho < (g2 + exl * exl) * 0.999999f), right = !(rho < (g2 + exr * exr) * 0.999999f);
            const int c_lo = left ? max(cx - 1, 0) : cx, c_hi = right ? min(cx + 2, gp.gx) : cx + 1;   // cells [c_lo, c_hi)
            const unsigned short* tr = L.tab + (z * gp.gy + y) * L.tw;
            eval_range((int)tr[c_lo >> L.xs], (int)tr[min((c_hi + xh1) >> L.xs, L.nx)]);   // (rounded outwards to the table's resolution)
        };
        // (rows in a runtime loop, ONE copy of the evaluation loop: the kernel has to stay well inside the 64 KB instruction
        // cache two CUs share -- unrolled, this function alone was 30 KB and every pass of every pair missed in it)
        const int nrow = LG == 16 ? 1 : LG == 4 ? 3 : 9;
        const unsigned order = sub == 0 ? 0xfffff820u : sub == 1 ? 0xffffff64u : sub == 2 ? 0xffffff71u : 0xffffff53u;   // LG = 4: a nibble per row, f: no more
#pragma nounroll
        for (int i = 0; i < nrow; ++i) {
            // LG = 16: lane r < 9 of the group walks row r; LG = 1: all nine, the query's own row (4) first -- without a radius
            // (first pass) it then comes from that row's best, grown by the skin
            const int t = LG == 16 ? sub : LG == 4 ? (int)((order >> (4 * i)) & 15u) : (i == 0 ? 4 : i <= 4 ? i - 1 : i);
            if (LG != 1 && t >= 9) break;
            if (LG == 1 && i == 1 && rho == __builtin_inff() && b1 != RES_NOBITS) {
                const float g = __builtin_amdgcn_sqrtf(__uint_as_float(b1)) + skin;
                rho = fmaxf(__uint_as_float(b1), g * g);
            }
            const int tzz = t / 3, tyy = t - 3 * tzz;
            walk_row(tyy, tzz, tyy == 0 ? ey2[0] : tyy == 2 ? ey2[2] : 0.f, tzz == 0 ? ez2[0] : tzz == 2 ? ez2[2] : 0.f);
        }
    }
    KSS_SLAP(2);
    if constexpr (LG > 1) {   // (wave-uniform) the lanes of a group have walked DISJOINT rows for the same source: merge by DPP
#pragma unroll
        for (int step = 0; step < (LG == 16 ? 4 : 2); ++step) {
            auto xch = [&](int v) { return step == 0 ? quad_x1(v) : step == 1 ? quad_x2(v) : step == 2 ? row_x4(v) : row_x8(v); };
            const unsigned o1 = (unsigned)xch((int)b1), o2 = (unsigned)xch((int)b2), o3 = (unsigned)xch((int)m3);
            const int q1 = xch(p1), q2 = xch(p2);
            // (both partners evaluate the same rule on the same two sorted triples: they end with the same result)
            const bool theirs = o1 < b1 || (o1 == b1 && q1 >= 0 && (p1 < 0 || tie_lower(q1, p1)));
            const unsigned n3 = min(min(m3, o3), min(max(b2, o1), max(b1, o2)));   // third smallest of the union
            if (theirs) {   // winner theirs; runner-up: my best or their second
                const bool mine2 = b1 <= o2;
                b2 = mine2 ? b1 : o2; p2 = mine2 ? p1 : q2;
                b1 = o1; p1 = q1;
            } else {        // winner mine; runner-up: their best or my second
                const bool theirs2 = o1 < b2;
                b2 = theirs2 ? o1 : b2; p2 = theirs2 ? q1 : p2;
            }
            m3 = min(n3, RES_INFBITS);
        }
    }
    if (act) {
        float face2 = __builtin_inff();
        int rfin = 0;
        // (a small target is swept by the wave in less time than one further shell takes: 64 targets per step.  The candidate
        // ICPs of a registration start from local minima of the rotation search -- a few per cent of their sources are more
        // than a cell away from every target in every pass)
        const int rcap = nt <= 4096 ? 1 : gp.rcap;
        for (int r = 1; r <= rcap; ++r) {
            if (r > 1) {   // shell r: every row of the (2r+1)^2 window over its whole x extent (points seen before are seen again: a minimum does not mind)
                const int xh1 = (1 << L.xs) - 1;
                const int c_lo = max(cx - r, 0), c_hi = min(cx + r + 1, gp.gx);
                for (int z = max(cz - r, 0); z <= min(cz + r, gp.gz - 1); ++z)
                    for (int y = max(cy - r, 0); y <= min(cy + r, gp.gy - 1); ++y) {
                        const unsigned short* tr = L.tab + (z * gp.gy + y) * L.tw;
                        eval_range((int)tr[c_lo >> L.xs], (int)tr[min((c_hi + xh1) >> L.xs, L.nx)]);
                    }
            }
            const float best = __uint_as_float(b1);   // (none: NaN -- every comparison below fails)
            // distance from the query to the faces of the visited block; faces on the grid border are open
            float b = __builtin_inff();
            if (cx - r > 0) b = fminf(b, qx - (gp.ox + (float)(cx - r) * gp.h));
            if (cx + r < gp.gx - 1) b = fminf(b, (gp.ox + (float)(cx + r + 1) * gp.h) - qx);
            if (cy - r > 0) b = fminf(b, qy - (gp.oy + (float)(cy - r) * gp.h));
            if (cy + r < gp.gy - 1) b = fminf(b, (gp.oy + (float)(cy + r + 1) * gp.h) - qy);
            if (cz - r > 0) b = fminf(b, qz - (gp.oz + (float)(cz - r) * gp.h));
            if (cz + r < gp.gz - 1) b = fminf(b, (gp.oz + (float)(cz + r + 1) * gp.h) - qz);
            const float bs = b - gp.eps;
            if (b == __builtin_inff()) done = true;                                            // the whole grid has been visited
            else if (bs > 0.f && best < bs * bs * 0.999999f) { done = true; face2 = bs * bs; }   // every unvisited point is strictly farther
            if (done) { rfin = r; break; }
        }
        // what the r = 1 walk has proven about every target but the two it returns: walked ones are at computed distance >= m3,
        // pruned ones beyond rho, unvisited ones beyond the block's faces.  (Shells r > 1 revisit points: no runner-up, B = 0.)
        if (done && rfin == 1 && p1 >= 0) bnew = fminf(__builtin_amdgcn_sqrtf(fminf(fminf(__uint_as_float(m3), rho), face2) * 0.99999f) * 0.999999f, 1e30f);
        else p2 = -1;
    }
    KSS_SLAP(3);
    // ---- last resort, shared by the wave: all the pair's targets, 64 at a time (asked for by the group's first lane) ----
    unsigned long long un = __builtin_amdgcn_ballot_w64(act && !done && sub == 0);
    if (prof) prof[5] += (unsigned long long)__builtin_popcountll(un);
    while (un != 0ull) {
        const int l = __builtin_ctzll(un);
        un &= un - 1ull;
        const float sx = __shfl(qx, l, 64), sy = __shfl(qy, l, 64), sz = __shfl(qz, l, 64);
        unsigned cb = RES_NOBITS;
        int cp = -1;
        for (int k = lane; k < nt; k += 64) {
            const float* t = L.t3 + 3 * k;
            const unsigned db = __float_as_uint(dist2<FMA>(sx, sy, sz, t[0], t[1], t[2]));
            if (db < cb) { cb = db; cp = k; }
            else if (db == cb && cp >= 0 && tie_lower(k, cp)) cp = k;
        }
        unsigned long long key = ((unsigned long long)cb << 32) | (unsigned long long)(cp >= 0 ? __float_as_uint(sg[cp].w) : 0xffffffffu);
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) {
            const unsigned long long ok = __shfl_xor(key, m, 64);
            const int op = __shfl_xor(cp, m, 64);
            if (ok < key) { key = ok; cp = op; }
        }
        if (lane == l) { p1 = cp; p2 = -1; done = true; fell = 1u; bnew = 0.f; }
    }
    KSS_SLAP(4);
#undef KSS_SLAP
    out_w = ((unsigned)(p1 >= 0 ? p1 : 0xffff) & 0xffffu) | ((unsigned)(p2 >= 0 ? p2 : 0xffff) << 16);
    out_b = bnew;
    out_fell = fell;
}

template <bool FMA>
__global__ __launch_bounds__(RES_THREADS, 1) void resident_icp_kernel(const ResArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char res_dyn[];
    __shared__ int s_ctl[4];      // [0] requests queued this round, [1] the gate was answered
    __shared__ int s_ps[16];      // the gate record (PairState layout)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = tid >> 9;
    const int pi = a.perm ? a.perm[blockIdx.x] : (int)blockIdx.x;
    const GridPairDev pr = a.pairs[pi];
    const GridParams& gp = pr.gp;
    const int nt = pr.tgt_n, ns = pr.src_n, n_rows = pr.n_rows;
    float* t3 = reinterpret_cast<float*>(res_dyn);
    unsigned short* tab = reinterpret_cast<unsigned short*>(t3 + 3 * a.ntc);
    unsigned* queue = reinterpret_cast<unsigned*>(res_dyn + (size_t)12 * a.ntc + (size_t)2 * a.tabc);
    double (*shw)[RES_THREADS / 64][NSUMS] = reinterpret_cast<double (*)[RES_THREADS / 64][NSUMS]>(queue);   // wave totals of RES_G slots: the queue is dead by then
    double (*rowv)[NSUMS] = reinterpret_cast<double (*)[NSUMS]>(queue + RES_NQ * RES_QW);
    static_assert(sizeof(double) * RES_G * (RES_THREADS / 64) * NSUMS <= sizeof(unsigned) * RES_NQ * RES_QW, "the wave totals alias the request queue");
    // diagnostic timeline (100 MHz s_memrealtime; a.stamps null in production): [pair * 16 + {0 start, 1 moved in, 15 end}];
    // sums over the passes >= 1 of {8 gate wait, 9 phase A, 10 phase B, 11 phase C, 12 pair total + publication}, 13 = phase B
    // of pass 0 (every source searches), 6 = the rest of pass 0, 14 = requests in all passes, 7 = passes; 2 - 5: res_search
    unsigned long long t_last = 0;
#define KSS_RSTAMP(k) do { if (a.stamps && tid == 0) a.stamps[(size_t)pi * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define KSS_RLAP(k) do { if (a.stamps && tid == 0) { const unsigned long long now_ = __builtin_amdgcn_s_memrealtime(); a.stamps[(size_t)pi * 16 + (k)] += now_ - t_last; t_last = now_; } } while (0)
    KSS_RSTAMP(0);

    // ---- the pair moves in: targets and cell table to LDS, sources to registers --------------------------------------
    const float4* __restrict__ sg = a.sorted + pr.sorted_base;
    for (int k = tid; k < nt + 3; k += RES_THREADS) {      // (three points of padding at +inf: an evaluation step reads four)
        float4 v = make_float4(__builtin_inff(), __builtin_inff(), __builtin_inff(), 0.f);
        if (k < nt) v = sg[k];
        t3[3 * k] = v.x; t3[3 * k + 1] = v.y; t3[3 * k + 2] = v.z;
    }
    ResLds L;
    L.t3 = t3; L.tab = tab;
    {
        const int nrows_yz = gp.gy * gp.gz;
        int xs = 0;
        while (xs < 8 && nrows_yz * (((gp.gx + (1 << xs) - 1) >> xs) + 1) > a.tabc) ++xs;   // (the host has checked that one fits)
        L.xs = xs;
        L.nx = (gp.gx + (1 << xs) - 1) >> xs;
        L.tw = L.nx + 1;
        const int32_t* __restrict__ cs = a.cell_start + pr.cell_base;
        for (int e = tid; e < nrows_yz * L.tw; e += RES_THREADS) {
            const int row = e / L.tw, j = e - row * L.tw;
            tab[e] = (unsigned short)(cs[row * gp.gx + min(j << xs, gp.gx)] - pr.sorted_base);
        }
    }
    const float skin_abs = fmaxf(a.skin, 0.f) * gp.h;
    // per slot: the position; wc = {winner, runner-up}: their positions in the LDS arrays (16 bits each, 0xffff: none) -- or,
    // while a search is under way, its queue entry (bit j of `queued`); room: what is left of the skip bound -- or, while the
    // slot waits for a search (bit j of `pend`), its pruning radius
    float px[RES_SMAX], py[RES_SMAX], pz[RES_SMAX], room[RES_SMAX];
    unsigned wc[RES_SMAX];
#pragma unroll
    for (int j = 0; j < RES_SMAX; ++j) {
        px[j] = py[j] = pz[j] = room[j] = 0.f;
        wc[j] = 0xffffffffu;
        const int s = j * RES_THREADS + tid;
        if (s < ns) {
            if (a.first_pass > 0) {   // the second launch of a split batch: where the first one left this source
                const float4 v = a.st_pos[pr.src_base + s];
                px[j] = v.x; py[j] = v.y; pz[j] = v.z; room[j] = v.w;
                wc[j] = a.st_wc[pr.src_base + s];
            } else {
                const float4 v = a.src0[pr.src_base + s];
                px[j] = v.x; py[j] = v.y; pz[j] = v.z;
            }
        }
    }
    if (tid == 0) s_ctl[2] = 0;   // searches asked for in the passes >= 1 of this launch (kept in LDS: a register for it costs this kernel 180 spills)
    __syncthreads();
    KSS_RSTAMP(1);
    if (a.stamps && tid == 0) t_last = __builtin_amdgcn_s_memrealtime();

    static_assert(RES_SMAX % 2 == 0, "the slot loops take two slots per trip");
    auto rotate2 = [&]() {                 // slot k + 2 moves to register k, slots 0 and 1 to the end
        const float x0 = px[0], y0 = py[0], z0 = pz[0], r0 = room[0], x1 = px[1], y1 = py[1], z1 = pz[1], r1 = room[1];
        const unsigned w0 = wc[0], w1 = wc[1];
#pragma unroll
        for (int k = 0; k + 2 < RES_SMAX; ++k) { px[k] = px[k + 2]; py[k] = py[k + 2]; pz[k] = pz[k + 2]; room[k] = room[k + 2]; wc[k] = wc[k + 2]; }
        px[RES_SMAX - 2] = x0; py[RES_SMAX - 2] = y0; pz[RES_SMAX - 2] = z0; room[RES_SMAX - 2] = r0; wc[RES_SMAX - 2] = w0;
        px[RES_SMAX - 1] = x1; py[RES_SMAX - 1] = y1; pz[RES_SMAX - 1] = z1; room[RES_SMAX - 1] = r1; wc[RES_SMAX - 1] = w1;
    };
    unsigned pend = 0u, queued = 0u;
    // requests of this lane's pending slots go to the queue, one LDS atomic per wave; the slots that do not fit stay pending
    auto push_pending = [&]() {
        const int mine = (int)__builtin_popcount(pend);
        int incl = mine;   // inclusive prefix over the lanes of the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int v = __shfl_up(incl, off, 64);
            if (lane >= off) incl += v;
        }
        const int wave_total = __shfl(incl, 63, 64);
        if (wave_total == 0) return;       // uniform
        int base = 0;
        if (lane == 0) base = atomicAdd(&s_ctl[0], wave_total);
        int e = __builtin_amdgcn_readfirstlane(base) + incl - mine;
#pragma unroll
        for (int j = 0; j < RES_SMAX; ++j) {
            if (((pend >> j) & 1u) != 0u) {
                if (e < RES_NQ) {
                    unsigned* q = queue + e * RES_QW;
                    q[0] = __float_as_uint(px[j]); q[1] = __float_as_uint(py[j]); q[2] = __float_as_uint(pz[j]); q[3] = __float_as_uint(room[j]);
                    wc[j] = (unsigned)e;
                    pend &= ~(1u << j);
                    queued |= 1u << j;
                }
                ++e;
            }
        }
    };

    float m[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) m[k] = 0.f;
    int apply = 0, mode = 0;      // mode 0: a regular pass; 1: the getFitnessScore() pass (the last one); 2: stop
    for (int pass = a.first_pass; pass < a.max_passes; ++pass) {
        // (the lane number goes through an opaque move once per pass: otherwise the compiler hoists every slot's addresses --
        // ten 64-bit source addresses, the LDS addresses of the sums -- out of this loop and spills them)
        int tv = tid;
        asm volatile("" : "+v"(tv));
        if (pass > 0) {
            // ---- the gate: five granules {3 words, stamp + check} the host stores through the BAR; a granule whose check
            // fails (stale, or seen torn) is simply read again ----
            if (tid < 8) {
                const unsigned expect = a.stamp0 + (unsigned)pass;
                const unsigned int* src = a.gate + (size_t)pi * 32 + 4 * min(tid, 4);
                u32x4 v;
                int n = 0;
                bool ok = true;
                for (;;) {
                    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(src) : "memory");   // system scope: the writer is the host
                    const unsigned tag = v.w - kss_mix3(v.x, v.y, v.z);
                    if (__builtin_amdgcn_ballot_w64(tag == expect) == 0xffull) break;
                    if (__builtin_amdgcn_ballot_w64(tag == a.stamp0 + RES_STAMP_ANY) == 0xffull) break;   // "whatever pass you wait for": a stop order
                    __builtin_amdgcn_s_sleep(2);
                    if (++n > a.gate_polls) { ok = false; break; }
                }
                if (tid < 5) { s_ps[3 * tid] = (int)v.x; s_ps[3 * tid + 1] = (int)v.y; s_ps[3 * tid + 2] = (int)v.z; }
                if (tid == 0) s_ctl[1] = ok ? 1 : 0;
            }
            __syncthreads();
            if (!s_ctl[1]) { if (tid == 0) res_store_exit_flag(a.exit_flags, pi, a.exit_tag, (unsigned)s_ctl[2]); return; }   // nobody answered: leave (the host's wait reports it)
#pragma unroll
            for (int k = 0; k < 12; ++k) m[k] = __int_as_float(__builtin_amdgcn_readfirstlane(s_ps[k]));   // (uniform: scalar registers)
            apply = __builtin_amdgcn_readfirstlane(s_ps[13]);
            mode = __builtin_amdgcn_readfirstlane(s_ps[14]);
            if (mode >= 2) { if (tid == 0) res_store_exit_flag(a.exit_flags, pi, a.exit_tag, (unsigned)s_ctl[2]); return; }
        }
        KSS_RLAP(pass > 0 ? 8 : 6);
        const bool fit = mode == 1;
        const bool full = fit || a.full_always != 0;
        const unsigned long long seq_k = a.seq0 + (unsigned long long)pass;
        if (tid == 0) s_ctl[0] = 0;
        __syncthreads();

        // ---- phase A: move; keep the winner while the skip bound proves it; the others ask for a search --------------------
        // Per source the lane carries its last winner w1, a runner-up w2 and `room`: when the source was last searched every
        // target OTHER THAN w1, w2 was at true distance >= B, the source has moved by at most acc since, room <= B - acc.  By the
        // triangle inequality every other target is now at true distance >= room, so its COMPUTED squared distance (five f32
        // roundings: relative error < 6 * 2^-24) exceeds room^2 * 0.99999.  If the smaller of the computed distances to w1 and
        // w2 is below that -- and they differ -- that one wins, strictly: exactly what a search would return.  (One candidate
        // is the rule of kss_grid.hip, phase A; with two, a source near the border of two targets' cells no longer searches
        // in every pass, only one near a corner of three.)  Every update of room rounds DOWN: the displacement up by 2e-5
        // relative, the difference down by 1e-6.
        unsigned fellm = 0u;               // slots resolved by the sweep
        pend = 0u; queued = 0u;
        // (ONE copy of the slot's code in a runtime loop; the slot being worked on is always register 0 of each array, and the
        // arrays are rotated by one after every trip -- ten trips bring every slot home again.  Ten unrolled copies of this
        // phase and of phase C were 40 KB of code: with the searches more than the instruction cache holds)
        auto slot_a = [&](auto kc, int j) {
            constexpr int K = decltype(kc)::value;
            const int s = j * RES_THREADS + tv;
            if (j * RES_THREADS < ns && s < ns) {
                const float ox = px[K], oy = py[K], oz = pz[K];     // where the source was in the last pass
                float x = ox, y = oy, z = oz;
                if (fit) {   // getFitnessScore(): final * ORIGINAL input
                    const float4 v = a.src0[pr.src_base + s];
                    x = v.x; y = v.y; z = v.z;
                }
                if (apply) {   // pcl transformCloud: Matrix4f x point, Eigen order, float, no fma
                    px[K] = ((m[0] * x + m[1] * y) + m[2] * z) + m[3];
                    py[K] = ((m[4] * x + m[5] * y) + m[6] * z) + m[7];
                    pz[K] = ((m[8] * x + m[9] * y) + m[10] * z) + m[11];
                } else {
                    px[K] = x; py[K] = y; pz[K] = z;
                }
                const float qx = px[K], qy = py[K], qz = pz[K];
                // a non-finite query matches nothing
                const bool qok = (qx - qx) == 0.f && (qy - qy) == 0.f && (qz - qz) == 0.f;
                if (!qok) {
                    wc[K] = 0xffffffffu;
                    room[K] = 0.f;
                } else {
                    bool walker = true;
                    float rho = __builtin_inff();
                    const unsigned w1 = wc[K] & 0xffffu, w2 = wc[K] >> 16;
                    if (w1 != 0xffffu) {
                        const float* t = t3 + 3 * w1;
                        const unsigned d1 = __float_as_uint(dist2<FMA>(qx, qy, qz, t[0], t[1], t[2]));
                        unsigned d2 = RES_NOBITS;
                        if (w2 != 0xffffu) { const float* u = t3 + 3 * w2; d2 = __float_as_uint(dist2<FMA>(qx, qy, qz, u[0], u[1], u[2])); }
                        const float d0 = __uint_as_float(min(d1, d2));
                        const float mx = qx - ox, my = qy - oy, mz = qz - oz;
                        const float moved = __builtin_amdgcn_sqrtf((mx * mx + my * my) + mz * mz);
                        const float r = (room[K] - moved * 1.00002f) * 0.999999f;
                        if (a.skin >= 0.f && r > 1e-7f && (r * r) * 0.99999f > d0 && d1 != d2) {
                            walker = false;            // the nearer of the two again
                            room[K] = r;
                            if (d2 < d1) wc[K] = w2 | (w1 << 16);
                        } else {
                            const float grown = __builtin_amdgcn_sqrtf(d0) + skin_abs;
                            rho = moved * 4.f <= skin_abs ? fmaxf(d0, grown * grown) : d0;
                        }
                    }
                    if (walker) { pend |= 1u << j; room[K] = rho; }   // (a walker's room is dead: it carries the pruning radius)
                }
            }
        };
#pragma nounroll
        for (int j = 0; j < RES_SMAX; j += 2) {
            slot_a(std::integral_constant<int, 0>(), j);
            slot_a(std::integral_constant<int, 1>(), j + 1);
            rotate2();
        }
        push_pending();
        KSS_RLAP(pass > 0 ? 9 : 6);

        // ---- phase B: the searches.  Queue entries are served sixteen lanes each (64 per round of the workgroup); what did
        // not fit is queued again -- or, when that is most of the cloud (the first pass of a registration: every source
        // searches), searched by its own lane, ten searches per lane with every lane busy ----
        for (bool first_round = true;; first_round = false) {
            __syncthreads();               // the requests are in
            if (pass > 0) KSS_RLAP(2);
            const int total = s_ctl[0];
            if (first_round && pass > 0 && tid == 0) s_ctl[2] += total;
            // most of the cloud searches (the first pass): nobody serves the queue -- the lanes whose requests went there take
            // them back (the query, the radius are still in their registers) and every lane searches its own slots
            const bool dense = total > 5 * RES_NQ;
            if (dense) { pend |= queued; queued = 0u; }
            const int nq = dense ? 0 : min(total, RES_NQ);
            if (a.stamps && tid == 0) a.stamps[(size_t)pi * 16 + 14] += (unsigned long long)nq;
            // lanes per search by how many there are: few -> sixteen lanes each (shortest chain), more -> four
            auto serve = [&](auto lgc) {
                constexpr int LGC = decltype(lgc)::value;
                for (int e0 = 0; e0 < nq; e0 += RES_THREADS / LGC) {
                    if (e0 + (64 / LGC) * wave >= nq) break;       // uniform per wave
                    const int e = e0 + tid / LGC;
                    const bool act = e < nq;
                    float qx = 0.f, qy = 0.f, qz = 0.f, rho = __builtin_inff();
                    if (act) {
                        const unsigned* q = queue + e * RES_QW;
                        qx = __uint_as_float(q[0]); qy = __uint_as_float(q[1]); qz = __uint_as_float(q[2]); rho = __uint_as_float(q[3]);
                    }
                    unsigned w, fell;
                    float bnew;
                    res_search<FMA, LGC>(act, gp, L, nt, sg, skin_abs, qx, qy, qz, rho, w, bnew, fell);
                    // (the lanes of a group have read the request before any of them overwrites it: same wave, program order)
                    if (act && (tid & (LGC - 1)) == 0) {
                        unsigned* q = queue + e * RES_QW;
                        q[0] = w; q[1] = __float_as_uint(bnew); q[2] = fell;
                    }
                }
            };
            // (measured at C3, searches per microsecond of a round: 16 lanes 13, 4 lanes 32, one lane 25 -- one lane per search only
            // pays when EVERY lane has one, which is the own-slot loop below)
            if (nq <= RES_THREADS / 16) serve(std::integral_constant<int, 16>());
            else serve(std::integral_constant<int, 4>());
            if (pass > 0) KSS_RLAP(3);
            __syncthreads();
            // the owners pick their answers up
            if (queued != 0u) {
#pragma unroll
                for (int j = 0; j < RES_SMAX; ++j) {
                    if (((queued >> j) & 1u) != 0u) {
                        const unsigned* q = queue + wc[j] * RES_QW;
                        wc[j] = q[0];
                        room[j] = __uint_as_float(q[1]);
                        fellm |= q[2] << j;
                    }
                }
                queued = 0u;
            }
            const int rem = total - nq;    // uniform
            if (rem <= 0) break;
            if (dense || rem > 4 * RES_NQ) {
                // most of the cloud searches: every lane takes its own pending slots, one per trip (no queue, no barrier: the
                // targets and the table are read-only)
                while (__builtin_amdgcn_ballot_w64(pend != 0u) != 0ull) {
                    float qx = 0.f, qy = 0.f, qz = 0.f, rho = __builtin_inff();
                    int slot = -1;
                    if (pend != 0u) {
                        slot = __builtin_ctz(pend);
#pragma unroll
                        for (int j = 0; j < RES_SMAX; ++j)
                            if (j == slot) { qx = px[j]; qy = py[j]; qz = pz[j]; rho = room[j]; }
                    }
                    unsigned w, fell;
                    float bnew;
                    res_search<FMA, 1>(slot >= 0, gp, L, nt, sg, skin_abs, qx, qy, qz, rho, w, bnew, fell);
                    if (slot >= 0) {
#pragma unroll
                        for (int j = 0; j < RES_SMAX; ++j)
                            if (j == slot) { wc[j] = w; room[j] = bnew; }
                        fellm |= fell << slot;
                        pend &= pend - 1u;
                    }
                }
                break;
            }
            __syncthreads();               // every answer of this round has been picked up
            if (tid == 0) s_ctl[0] = 0;
            __syncthreads();
            push_pending();
        }
        __syncthreads();                   // the queue becomes the wave totals
        KSS_RLAP(pass > 0 ? 10 : 13);
        if (a.stamps && tid == 0) a.stamps[(size_t)pi * 16 + 7] += 1ull;

        // ---- phase C: the rows of this pair in the canonical order (kss_device.hpp).  Slot j of lanes [512 h, 512 h + 512)
        // is row 2j + h: lane t of the row is lane t of grid_pass_kernel's workgroup ----
        {
        const int nslot = (ns + RES_THREADS - 1) / RES_THREADS;
        const bool leaving = a.split_at > 0 && pass + 1 == a.split_at;   // uniform
        auto slot_c = [&](auto kc, int j) {
            constexpr int K = decltype(kc)::value;
            const int jj = j % RES_G;
            if (j < nslot) {                               // uniform
            if (2 * j + half < n_rows) {                   // (uniform per wave)
                const int s = j * RES_THREADS + tv;
                const int lane = tv & 63, wave = tv >> 6;
                const unsigned w1 = wc[K] & 0xffffu;
                const bool have = s < ns && w1 != 0xffffu;
                const float qx = px[K], qy = py[K], qz = pz[K];
                // the last pass of the first launch of a split batch: this slot's registers rest in memory until the second one
                // (here, where the slot is at hand -- a block of its own at the end of the pass cost 60 spilled registers)
                if (leaving && s < ns) {
                    a.st_pos[pr.src_base + s] = make_float4(qx, qy, qz, room[K]);
                    a.st_wc[pr.src_base + s] = wc[K];
                }
                float wx = 0.f, wy = 0.f, wz = 0.f, d2 = 0.f;
                if (have) {
                    const float* t = t3 + 3 * w1;
                    wx = t[0]; wy = t[1]; wz = t[2];
                    d2 = dist2<FMA>(qx, qy, qz, wx, wy, wz);
                    if (a.idx_out || a.d2_out) {
                        const int oi = __float_as_int(a.src0[pr.src_base + s].w);   // original source index (sources are in cell order)
                        if (a.idx_out) a.idx_out[oi] = __float_as_int(sg[w1].w);
                        if (a.d2_out) a.d2_out[oi] = d2;
                    }
                }
                const double d2d = have ? (double)d2 : 0.0;
                const bool kept = have && !(d2d > a.max_d2);   // PCL: `if (distance[0] > max_dist_sqr) continue;`
                double col[16];
                {
                    // (the zeros of a lane that contributes nothing are selected in f32: (double)0.f is the 0.0 the other
                    // engines select, and 0.0 * 0.0 = 0.0 -- seven selects instead of thirteen 64-bit ones)
                    const double sx = (double)(kept ? qx : 0.f), sy = (double)(kept ? qy : 0.f), sz = (double)(kept ? qz : 0.f);
                    const double ux = (double)(kept ? wx : 0.f), uy = (double)(kept ? wy : 0.f), uz = (double)(kept ? wz : 0.f);
                    col[0] = sx; col[1] = sy; col[2] = sz; col[3] = ux; col[4] = uy; col[5] = uz;
                    col[6] = sx * ux; col[7] = sx * uy; col[8] = sx * uz;
                    col[9] = sy * ux; col[10] = sy * uy; col[11] = sy * uz;
                    col[12] = sz * ux; col[13] = sz * uy; col[14] = sz * uz;
                    col[15] = (double)(kept ? d2 : 0.f);
                }
                wave_tree16(col);
                const unsigned long long mk = __builtin_amdgcn_ballot_w64(kept), mf = __builtin_amdgcn_ballot_w64(((fellm >> j) & 1u) != 0u);
                double extra = 0.0;
                if (full) extra = wave_tree2(d2d, have ? sqrt(d2d) : 0.0);   // lane 0: sum of all d2, lane 32: sum of sqrt(d2)
                if ((lane & 15) == 0) {
                    const int q = lane >> 4;
#pragma unroll
                    for (int c = 0; c < 4; ++c) shw[jj][wave][1 + 4 * q + c] = col[c];
                    if (lane == 0) {
                        shw[jj][wave][0] = (double)__builtin_popcountll(mk);
                        shw[jj][wave][NSUMS - 1] = (double)__builtin_popcountll(mf);
                        shw[jj][wave][17] = extra;
                    }
                    if (lane == 32) shw[jj][wave][18] = extra;
                }
            }
            if (jj == RES_G - 1 || j == nslot - 1) {       // the group's wave totals are in: its rows
                __syncthreads();
                if (tv < RES_G * 2 * NSUMS) {
                    const int c = tv % NSUMS, hh = (tv / NSUMS) & 1, j2 = tv / (2 * NSUMS);
                    const int slot = j - jj + j2, row = 2 * slot + hh;
                    if (slot <= j && row < n_rows) {
                        double r = 0.0;
#pragma unroll
                        for (int ww = 0; ww < 8; ++ww) r += shw[j2][hh * 8 + ww][c];
                        rowv[row][c] = r;
                    }
                }
                __syncthreads();
            }
            }
        };
#pragma nounroll
        for (int j = 0; j < RES_SMAX; j += 2) {
            slot_c(std::integral_constant<int, 0>(), j);
            slot_c(std::integral_constant<int, 1>(), j + 1);
            rotate2();
        }
        }
        KSS_RLAP(pass > 0 ? 11 : 6);
        // pair total: rows k = g, g + 25, ... sequentially per group g, then the group totals in group order
        if (tid < NSUMS) {
            double v = 0.0;
            for (int gg = 0; gg < PASS_FG; ++gg) {
                double acc = 0.0;
                for (int k = gg; k < n_rows; k += PASS_FG) acc += rowv[k][tid];
                v += acc;
            }
            if (!full && (tid == 17 || tid == 18)) v = 0.0;
            // {bits(sum), sequence number, check}: ONE aligned 16-byte system-scope store per sum into host-mapped memory;
            // the host takes a slot when the number matches and the check word fits the other three
            const unsigned long long vb = (unsigned long long)__double_as_longlong(v);
            u32x4 o;
            o.x = (unsigned)vb; o.y = (unsigned)(vb >> 32); o.z = (unsigned)seq_k; o.w = kss_mix3(o.x, o.y, o.z);
            unsigned long long* dst = a.pub + 2 * ((int64_t)pi * NSUMS + tid);
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(o) : "memory");
        }
        KSS_RLAP(pass > 0 ? 12 : 6);
        if (fit || (a.split_at > 0 && pass + 1 == a.split_at)) {   // the fitness pass is the last one; so is pass split_at - 1 of a split batch's first launch
            KSS_RSTAMP(15);
            if (tid == 0) res_store_exit_flag(a.exit_flags, pi, a.exit_tag, (unsigned)s_ctl[2]);
            return;
        }
        __syncthreads();                   // (rowv and the queue are rewritten by the next pass)
    }
    if (tid == 0) res_store_exit_flag(a.exit_flags, pi, a.exit_tag, (unsigned)s_ctl[2]);
#undef KSS_RSTAMP
#undef KSS_RLAP
}

// LDS bytes of a launch whose largest pair has ntc targets (multiple of 64) and tabc table entries (multiple of 8)
size_t resident_lds_bytes(int ntc, int tabc) {
    return (size_t)12 * ntc + (size_t)2 * tabc + sizeof(unsigned) * RES_NQ * RES_QW + sizeof(double) * 2 * RES_SMAX * NSUMS;
}

int launch_resident(hipStream_t st, bool fma, int npairs, const ResArgs& a, std::string& err) {
    const size_t bytes = resident_lds_bytes(a.ntc, a.tabc);
    if (bytes > (size_t)RES_LDS_MAX) { err = "resident launch: the pair does not fit the LDS"; return KSS_ERR_ARG; }
    // dynamic LDS beyond 64 KB has to be allowed per kernel AND per device: asked for on every launch (a cheap call), checked
    const void* fn = fma ? reinterpret_cast<const void*>(&resident_icp_kernel<true>) : reinterpret_cast<const void*>(&resident_icp_kernel<false>);
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, RES_LDS_MAX) != hipSuccess) {
        (void)hipGetLastError();
        err = "resident launch: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed";
        return KSS_ERR_HIP;
    }
    if (fma) hipLaunchKernelGGL(resident_icp_kernel<true>, dim3(npairs), dim3(RES_THREADS), bytes, st, a);
    else hipLaunchKernelGGL(resident_icp_kernel<false>, dim3(npairs), dim3(RES_THREADS), bytes, st, a);
    if (hipGetLastError() != hipSuccess) { err = "resident launch failed"; return KSS_ERR_HIP; }
    return KSS_OK;
}

}  // namespace kss
