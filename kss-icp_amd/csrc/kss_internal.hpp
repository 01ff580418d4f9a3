// kss_internal.hpp -- shared declarations between the HIP kernels (kss_kernels.hip) and the
// C-ABI / host drivers (kss_ctx.hpp, kss_engine.hip, kss_api.hip).  Product code; gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <functional>
#include <string>
#include <vector>

#include "../../include/kssicp.h"
#include "kss_host_math.hpp"

namespace kss {

// ---- device-visible descriptors ---------------------------------------------------------------
// Per-pair state re-uploaded every ICP iteration (64 B): rows 0..2 of the Matrix4f to apply to
// the source on load, and whether the pair still iterates.
struct alignas(16) PairState {
    float m[12];
    int32_t active;
    int32_t apply;   // 0: copy source unchanged (identity guess, PCL copies the cloud)
    int32_t pad[2];
};

// One NN-sweep workgroup = (block of sources of one pair) x (one target split).
struct alignas(16) NNWork {
    int32_t pair;
    int32_t src_begin;   // first source point (global index into the float4 source arrays)
    int32_t src_count;   // <= 256 * S
    int32_t tgt_begin;   // first target point of this split (global index into tgt4), multiple of TILE from pair base
    int32_t tgt_count;   // targets in this split, multiple of TILE (sentinel padded)
    int32_t tgt_pair_base;  // first target of the pair (idx written relative to it)
    int32_t key_begin;   // where this block's keys start: keys[key_begin + local_source]
    int32_t write_src;   // split 0 writes the transformed sources
};

// One correspondence-reduce workgroup = 256 consecutive sources of one pair.
struct alignas(16) RedWork {
    int32_t pair;
    int32_t src_begin;
    int32_t src_count;      // <= 256
    int32_t key_begin;      // keys[key_begin + s * key_stride + t]
    int32_t key_stride;     // distance between splits (= padded source count of the pair)
    int32_t n_split;
    int32_t tgt_pair_base;
    int32_t partial_index;  // row in the partial-sums array
};

struct PairRed {            // final reduce: rows [first, first+count) of partials -> sums[pair]
    int32_t first;
    int32_t count;
};

// 32-bit check word of three data words: folded into the stamp / sequence word of every 16-byte granule that crosses
// between host and device without a flag (gate records, result slots), so that a granule seen TORN -- some words of the
// previous write, some of the new one -- fails the check and is simply polled again instead of being taken for data.
__host__ __device__ inline unsigned kss_mix3(unsigned a, unsigned b, unsigned c) {
    unsigned h = (a + 0x9E3779B1u) * 0x85EBCA77u;
    h ^= h >> 15;
    h = (h ^ (b + 0x7F4A7C15u)) * 0xC2B2AE3Du;
    h ^= h >> 13;
    h = (h ^ (c + 0x165667B1u)) * 0x27D4EB2Fu;
    h ^= h >> 16;
    return h;
}

// Uniform cell list over the target's bounding box (kss_grid.hip)
struct GridParams {
    float ox, oy, oz;   // bbox minimum
    float h, inv_h;     // cell edge
    float eps;          // absolute slack covering f32 rounding of the cell assignment
    int32_t gx, gy, gz; // cells per axis (x fastest in the linear cell index)
    int32_t rcap;       // shells searched before a query falls back to the brute-force list pass
};

// per-pair entry of the batched cell list (device table)
struct alignas(16) GridPairDev {
    GridParams gp;
    int32_t cell_base;          // first cell of this pair in the global cell arrays
    int32_t tgt_base, tgt_n;    // target segment in tgt4 (padded layout) and its real point count
    int32_t src_base, src_n;    // source segment
    int32_t tgt_pad;            // slots of the target segment in tgt4 (multiple of NN_TILE, +inf sentinels behind tgt_n)
    int32_t row_base, n_rows;   // this pair's chunks of 512 sources = workgroups of the fused pass = partial rows
    int32_t sorted_base;        // first slot of this pair's targets in the cell-ordered array (sum of the earlier pairs' target counts)
};

// one pair's target segment for the batched pack kernel
struct PackSeg {
    int64_t in_off;     // first point in the caller's packed array
    int64_t out_base;   // first slot in tgt4
    int64_t n;          // real points; slots [n, next segment) are sentinel padding
};

// arguments of the fused cell-list pass (kss_grid.hip: grid_pass_kernel), by value
struct PassArgs {
    const float4* src_in; float4* src_out;      // cell-sorted sources (.w = original index): read, transformed copy written
    const int32_t* cell_start;                  // exclusive prefix of the cell counts; [-1] and [cells + 1] are readable
    const float4* sorted;                       // targets in cell order, .w = index within the pair's target
    const float4* tgt4;                         // targets in original order (fallback sweep, SEARCH = false gathers)
    float4* nn_win;                             // per source: its last winner {x, y, z, index in the pair's target}; .w = ~0: none
    float2* nn_state;                           // per source: {B, acc} of the skip test (grid_pass_kernel, phase A)
    int32_t chained;                            // 1: src_in is what the previous search pass wrote (its queries): displacements are measurable;
                                                // 2: the pair's state names the work buffer its last pass wrote (fitness pass)
    const float4* src_last0; const float4* src_last1;   // ... the two work buffers
    float skin;                                 // pruning radius grows by skin * cell edge; < 0: never skip (A/B switch)
    unsigned long long* keys;                   // single pair: key of the sources left to the list pass
    int32_t* list; int32_t* list_count;         // single pair: those sources
    const GridPairDev* pairs;                   // BATCH: per-pair table
    const int32_t* row_pair;                    // BATCH: pair of every row (= workgroup)
    const PairState* state;                     // BATCH: per-pair transforms; single pair: host-mapped transform of a gated launch or null
    GridPairDev pair0; PairState ps0;           // single pair: by value (no upload per iteration)
    int32_t total_rows, use_prev;
    int32_t gate_polls;                         // bound of the gate's poll loop (a kernel nobody answers leaves without touching anything)
    int32_t chain_len, gate_slot;               // chained launch: passes run by this one launch (1: a plain launch); first of the two gate records
    float4* src_alt;                            // chained launch: the other work buffer (== src_in), written by every second pass
    int32_t gate_seq, tagged_rows;              // gated single-pair launch: the stamp `state->pad[1]` must carry; rows as tagged granules
    unsigned int* gate_dev;                     // ... and the device-side copy of the record (five 16-byte granules)
    double max_d2;
    double* rows; int32_t* tickets;
    unsigned long long* pub; unsigned long long seq;
    int32_t* idx_out; float* d2_out;
    unsigned long long* stamps;                 // diagnostics (null in production): see KSS_STAMP below
    int32_t test_torn;                          // test hook: one result slot is first stored with data that does not fit its check word
};

// ---- the pair-resident ICP kernel (kss_resident.hip) ---------------------------------------------------------------
constexpr int RES_THREADS = 1024;     // lanes of the workgroup that holds one pair
constexpr int RES_SMAX = 10;          // sources per lane: a pair of up to 10240 sources lives in registers
constexpr int RES_NQ = 640;           // search requests queued per round (the rest is queued again, or searched by its own lane)
constexpr int RES_QW = 4;             // words per request {query, pruning radius}; answered in place {winner | runner-up << 16, bound, fell back}
constexpr int RES_G = 4;              // source slots whose wave totals are in LDS at a time
constexpr int RES_LDS_MAX = 160 * 1024 - 256;   // dynamic LDS of a workgroup (the rest: its few static words)
// gate stamps: every launch owns 4096; stamp0 + k opens pass k (k <= 4002), stamp0 + RES_STAMP_ANY is accepted at ANY pass --
// the host's order to stop, which may overwrite a record some workgroups have not read yet
constexpr unsigned RES_STAMP_ANY = 4095u;
// "this pair's workgroups have all left": the host learns that a resident kernel is gone from these flags, not from HIP (a
// HIP call of a serving thread can wait on locks other threads hold for as long as THEIR kernels run)
__device__ __forceinline__ void res_store_exit_flag(unsigned long long* flags, int pair, unsigned tag, unsigned note = 0u) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    u32x4 o;
    o.x = tag; o.y = (unsigned)pair; o.z = note; o.w = kss_mix3(o.x, o.y, o.z);
    unsigned long long* dst = flags + 2 * (int64_t)pair;
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(o) : "memory");
}
struct ResArgs {
    const GridPairDev* pairs;                   // per-pair table (cell grid, segments, rows)
    const int32_t* cell_start;                  // exclusive prefix of the cell counts (global positions in `sorted`)
    const float4* sorted;                       // targets in cell order, .w = index within the pair's target
    const float4* src0;                         // sources in cell order, .w = global source index
    unsigned int* gate;                         // per pair 32 words of fine-grained device memory the host stores into (BAR)
    unsigned long long* pub;                    // host-mapped result slots: per pair NSUMS x {bits, seq | check << 32}
    unsigned long long seq0;                    // pass k of a pair publishes under seq0 + k
    unsigned int stamp0;                        // ... and then waits for the gate record stamped stamp0 + k + 1
    double max_d2;
    float skin;                                 // pruning radius grows by skin * cell edge; < 0: never skip (A/B switch)
    int32_t gate_polls, max_passes, full_always;
    int32_t ntc, tabc;                          // LDS capacities of this launch: targets (multiple of 64), table entries (multiple of 8)
    int32_t* idx_out; float* d2_out;            // fitness pass: per-source correspondences (null: not wanted)
    unsigned long long* exit_flags;             // host-mapped, one 16-byte granule per pair: {exit_tag, pair, searches asked for, check} stored when the pair's workgroup leaves
    unsigned int exit_tag;                      // what the flag carries (the two launches of a split batch tell theirs apart)
    // A batch of more pairs than the device runs at once is run as TWO launches (kss_engine.hip, resident_loop): every pair's
    // passes [0, split_at) first, then the rest in the order of decreasing cost, predicted from the searches the first passes
    // asked for.  Between the launches a pair's registers rest in memory: position + skip room, candidates.
    const int32_t* perm;                        // workgroup -> pair (null: identity)
    int32_t first_pass, split_at;               // this launch runs passes [first_pass, split_at) (split_at 0: to the end)
    float4* st_pos; unsigned int* st_wc;        // per source (cell order): {x, y, z, room}, {winner | runner-up << 16}
    unsigned long long* stamps;                 // diagnostics (null in production)
};
size_t resident_lds_bytes(int ntc, int tabc);
int launch_resident(hipStream_t st, bool fma, int npairs, const ResArgs& a, std::string& err);

// PCL's octree bounding cube (kss_octree.hip; replayed on the host by oct_first_point / oct_adopt)
struct OctBox {
    double min[3], max[3];
    double res;
    int32_t depth;
};

constexpr int NN_TILE = 256;      // targets staged per LDS tile (one float4 per thread)
constexpr int NN_SUB = 32;        // targets per sub-tile (arg-min bookkeeping granularity)
constexpr int NN_THREADS = 256;

// pcl::octree::OctreePointCloud (1.8.1) bounding-cube rules, restated from its published source (octree_pointcloud.hpp).
// Empty octree: cube = point +- resolution / 2, then getKeyBitSize(): depth = bits of the voxel count (>= 2 per axis),
// cube centred around the box.
inline void oct_first_point(OctBox& b, const float* p) {
    const float minValue = FLT_EPSILON;
    for (int k = 0; k < 3; ++k) { b.min[k] = p[k] - b.res / 2; b.max[k] = p[k] + b.res / 2; }
    unsigned maxv = 2;
    for (int k = 0; k < 3; ++k) maxv = std::max(maxv, (unsigned)std::ceil((b.max[k] - b.min[k] - minValue) / b.res));
    b.depth = (int)std::ceil(std::log2((double)maxv) - minValue);
    const double side = (double)(1u << b.depth) * b.res;
    for (int k = 0; k < 3; ++k) {
        const double over = (side - (b.max[k] - b.min[k])) / 2.0;
        if (over > minValue) { b.min[k] -= over; b.max[k] += over; }
    }
}
// adoptBoundingBoxToPoint: while the point is outside [min, max), add a level -- the old cube becomes the child on the
// far side of every violated UPPER bound (min moves down on the other axes).  false: deeper than 21 levels.
inline bool oct_adopt(OctBox& b, const float* p) {
    const float minValue = FLT_EPSILON;
    for (;;) {
        bool up[3], any = false;
        for (int k = 0; k < 3; ++k) { up[k] = p[k] >= b.max[k]; any = any || up[k] || p[k] < b.min[k]; }
        if (!any) return true;
        if (b.depth >= 21) return false;
        double side = (double)(1u << b.depth) * b.res;
        for (int k = 0; k < 3; ++k) if (!up[k]) b.min[k] -= side;
        ++b.depth;
        side = (double)(1u << b.depth) * b.res - minValue;
        for (int k = 0; k < 3; ++k) b.max[k] = b.min[k] + side;
    }
}

// ---- kernel launchers (kss_kernels.hip) ---------------------------------------------------------
void launch_pack_f3_to_f4(hipStream_t st, const float* d_in, int64_t n, float4* d_out, int64_t n_pad, bool sentinel);
void launch_pack_f64_to_f4(hipStream_t st, const double* d_in, int64_t n, float4* d_out, int64_t n_pad, bool sentinel);
void launch_empty(hipStream_t st);
// single pair: target (sentinel padded, + one bbox partial row of 6 floats per 256 slots) and source in ONE launch
int pack_pair_bbox_rows(int64_t nt_pad);
void launch_pack_pair(hipStream_t st, int dtype, const void* d_tgt, int64_t nt, float4* d_tgt_out, int64_t nt_pad, float* d_bbox_partial,
                      const void* d_src, int64_t ns, float4* d_src_out, unsigned int* d_box_host, unsigned box_tag);
void launch_pack_batch(hipStream_t st, const void* d_in, int dtype, const PackSeg* d_seg, int nseg, int64_t total_out, float4* d_out);

void launch_nn_sweep(hipStream_t st, int S, bool fma, const NNWork* d_work, int n_work,
                     const PairState* d_state, const float4* d_src_in, float4* d_src_out,
                     const float4* d_tgt4, unsigned long long* d_keys);

void launch_nn_sweep_list(hipStream_t st, int S, bool fma, const NNWork* d_work, int n_work,
                          const PairState* d_state, float4* d_src_cur, const float4* d_tgt4,
                          unsigned long long* d_keys, const int32_t* d_list, const int32_t* d_count);
// both cell lists of a single pair (count -> scan -> scatter -> rank fix, shared launches)
void launch_grid_build_pair(hipStream_t st, const float4* d_tgt, int nt, float4* d_src, int ns, const GridParams& gp, int32_t* d_counts,
                            int32_t* d_start, int32_t* d_block_sums, float4* d_sorted, float4* d_tmp);
size_t scan_scratch_bytes(int n);   // scratch of the cell-count scan over n cells
int grid_pass_blocks(int total_rows);
void launch_grid_pass(hipStream_t st, bool fma, bool full, bool batch, bool search, const PassArgs& a);
void launch_gridb_bbox(hipStream_t st, const float4* d_tgt, const GridPairDev* d_pairs, int npairs, float* d_bbox);
void launch_gridb_build_targets(hipStream_t st, const float4* d_tgt, int total_tgt_pad, const GridPairDev* d_pairs, int npairs,
                                int total_cells, int32_t* d_counts, int32_t* d_start, int32_t* d_block_sums, float4* d_sorted);
void launch_gridb_sort_sources(hipStream_t st, const float4* d_src, int total_src, const GridPairDev* d_pairs, int npairs,
                               int total_cells, int32_t* d_counts, int32_t* d_start, int32_t* d_block_sums, float4* d_tmp, float4* d_out);
int gridb_lds_max_cells(bool half);
// all of a batch's cell lists, one workgroup per pair, counters in LDS sized for the largest grid of the batch (max_cells <=
// gridb_lds_max_cells(half)); half: 16-bit counters, allowed when no pair has 65536 or more targets or sources
// (false: the device refused the LDS size -- the caller builds through the global-atomic path)
bool launch_gridb_build_lds(hipStream_t st, const float4* d_tgt4, float4* d_src, float4* d_tmp, const GridPairDev* d_pairs, int npairs,
                            int32_t* d_cell_start, float4* d_sorted, int max_cells, bool half);
void launch_grid_stats(hipStream_t st, const float4* d_src, int ns, const GridParams& gp, const int32_t* d_cell_start,
                       unsigned long long* d_out /* [0] evaluations of the 3x3x3 block, [1] occupied cells */);

void launch_corr_reduce(hipStream_t st, const RedWork* d_work, int n_work, const PairState* d_state,
                        const float4* d_src, const float4* d_tgt4, const unsigned long long* d_keys,
                        double max_d2, double* d_partials, int32_t* d_idx_out, float* d_d2_out, int index_in_w);
// small batches: reduce + per-pair final sum + publication in one launch (d_pair_ticket: one zeroed int per pair)
void launch_corr_reduce_publish(hipStream_t st, const RedWork* d_work, int n_work, const PairState* d_state,
                                const float4* d_src, const float4* d_tgt4, const unsigned long long* d_keys,
                                double max_d2, double* d_partials, int32_t* d_idx_out, float* d_d2_out, int index_in_w,
                                const PairRed* d_pair_red, int32_t* d_pair_ticket, unsigned long long* d_pub, unsigned long long seq);
// the candidate batch of a registration (pairs sharing one small target): sweep + sums + publication in ONE launch per pass
size_t cand_pass_lds_bytes(int nt_pad);
int cand_pass_blocks_per_pair(int64_t ns);
bool launch_cand_pass(hipStream_t st, bool fma, int npairs, const PairState* d_state, const float4* d_src_in, float4* d_src_out, const float4* d_tgt,
                      int nt_pad, int ns, double max_d2, double* d_partials, unsigned int* d_row_tag, unsigned long long* d_pub, unsigned long long seq,
                      int32_t* d_idx_out, float* d_d2_out);
// ... and the same batch with every workgroup RESIDENT for the whole registration (cand_resident_kernel): the target is staged
// once, the sources stay in registers, and between passes a candidate's workgroups wait at its gate record for the transform
// the host solves from the sums its last workgroup published -- the protocol of the pair-resident engine (ResArgs), one
// launch per batch instead of one per pass.  A workgroup owns up to CAND_TPW tiles of 32 sources of ONE candidate.
constexpr int CAND_TPW = 4;
struct CandArgs {
    const float4* src0;                         // candidate p's sources at src0 + p * ns (original order)
    const float4* tgt;                          // the shared target, nt_pad points (padding at +inf)
    int32_t nt_pad, ns;
    int32_t bpp, wpp, tpw;                      // tiles per candidate, workgroups per candidate, tiles per workgroup (<= CAND_TPW)
    double max_d2;
    double* partials;                           // one row of NSUMS per tile
    unsigned int* row_tag;                      // one word per tile: the launch-and-pass number its row belongs to (no counter to keep zero at rest)
    const unsigned int* gate;                   // per candidate 32 words the host stores into through the BAR
    unsigned long long* pub;                    // host-mapped result slots
    unsigned long long seq0;
    unsigned int stamp0;
    int32_t gate_polls, max_passes;
    int32_t* idx_out; float* d2_out;            // fitness pass: per-source correspondences (null: not wanted)
    unsigned long long* exit_flags;             // ... and the last one stores {stamp0, pair, 0, check} into the candidate's host-mapped granule
    unsigned long long* stamps;                 // diagnostics (null in production): 16 per candidate, written by its workgroup 0
};
int cand_resident_capacity(bool fma, int nt_pad);   // workgroups the device keeps resident at once (0: cannot run)
int launch_cand_resident(hipStream_t st, bool fma, int npairs, const CandArgs& a, std::string& err);
// idx-driven variant for kss_cov: d2 recomputed with the reference arithmetic
void launch_corr_reduce_idx(hipStream_t st, const float* d_src3, const float* d_tgt3, const int32_t* d_idx,
                            int64_t n, double max_d2, double* d_partials, int n_blocks);
void launch_finalize_sums(hipStream_t st, const PairRed* d_pairs, int n_pairs, const double* d_partials,
                          double* d_out /* n_pairs * NSUMS, device or host-mapped */);

// pre-shape statistics of one or two clouds: two launches, result published to host-mapped {bits, seq} slots
void launch_preshape_pair(hipStream_t st, const void* const d_xyz[2], const int64_t n[2], int dtype, double* d_partials,
                          int32_t* d_tickets, double* d_cent, unsigned long long* d_pub, unsigned long long seq);
void launch_sum_columns(hipStream_t st, const double* d_partials, int n_rows, int n_cols, double* d_out);
void launch_row_sums(hipStream_t st, const double* d_partials, int n_rows, int n_cols, double scale, double* d_out);

void launch_pose_apply(hipStream_t st, const double* d_in, int64_t n, const kss_pose& pose,
                       const double cs[6], double* d_out);
constexpr int POSE_MANY = 32;
void launch_pose_apply_many(hipStream_t st, const double* d_in, int64_t n, const kss_pose& pose, const double (*cs)[6], int count, double* d_out);
void launch_transform_apply_f64(hipStream_t st, const float T[16], const double* d_in, int64_t n, double* d_out);

void launch_transform_apply_f32(hipStream_t st, const float T[16], const float* d_in, int64_t n, float* d_out);
int octree_voxels_device(hipStream_t st, const float* d_pts, int n, const OctBox& box, float* d_cen, int* m_out, std::string& err,
                         const std::function<void*(int, size_t)>& scratch);
void launch_fps(hipStream_t st, const double* d_xyz, int n, int m, double* d_mind, int32_t* d_idx, double* d_out);

int rot_search_grain(int64_t ns, int64_t nt, int g);   // sources per workgroup = targets per tile: 256 or 128
void launch_rot_search(hipStream_t st, const double* d_src, int64_t ns, const float4* d_tgt4, int64_t nt_pad /* multiple of nth */,
                       const double* d_cs /* g*2: cos,sin */, int g, double* d_partials, int n_src_blocks /* ceil(ns / nth) */, int nth);

int knn_plan_splits(int nq, int nt_pad, int k, size_t* scratch_bytes);
void launch_knn_sweep(hipStream_t st, const float4* d_qry, int nq, const float4* d_tgt, int nt_pad, int k, int32_t* d_idx, float* d_d2,
                      int n_split, void* d_scratch);
void launch_normals(hipStream_t st, const float4* d_pts, int n, const int32_t* d_knn, int k, double* d_normals);

int preshape_blocks(int64_t n);
int stream_blocks(int64_t n);

// AIVS down-sampler (kss_aivs.hip): indices of the selected points in the reference's output order
int aivs_device(hipStream_t st, const double* d_xyz, int n, int point_num, std::vector<int32_t>& out_idx, std::string& err,
                const std::function<void*(size_t)>& scratch);

}  // namespace kss
