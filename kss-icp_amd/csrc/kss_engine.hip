// kss_engine.hip -- the registration engine behind kss_icp* / kss_nn* / kss_pcr_qm / kss_register: how pairs, source
// blocks and target splits map onto workgroups (IcpPlan), staging of the packed clouds, the cell-list setups, one NN
// pass (fused cell-list launch, batched cell lists, or sweep + reduce + finalize) and the ICP loop around it.
// Host code only orchestrates: per iteration it launches the pass, takes 20 doubles per pair back, solves the 3x3 SVD
// (kss_host_math.hpp) and evaluates the PCL convergence criteria.  There is no CPU compute fallback in this file.
#pragma clang fp contract(off)

#include <emmintrin.h>

#include "kss_ctx.hpp"

namespace kss {

static int restore_zero_at_rest(kss_ctx* c);
static bool resident_gate_available(kss_ctx* c, int npairs);

// Workgroups of ONE launch that are certainly on the chip at the same time: one per compute unit of THIS device (the fused
// single-pair kernels need at most one CU's registers and LDS each).  Tagged rows and chained launches rely on it; on a
// partitioned or smaller part the limit shrinks with the CU count instead of stalling every chain until its polls run out.
static int resident_rows_limit(kss_ctx* c) {
    if (c->cu_count <= 0) {
        hipDeviceProp_t prop;
        c->cu_count = hipGetDeviceProperties(&prop, c->device) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 1;
        (void)hipGetLastError();
    }
    return c->cu_count;
}

constexpr int PUB_PAIRS = 32;   // brute-force batches up to this many pairs are awaited by spinning on the published sums
static int ensure_pub(kss_ctx* c, int npairs = PUB_PAIRS);   // host-mapped result slots (defined with wait_seq)

struct PairGeom {
    int64_t ns, nt;
    int32_t src_base;      // first source in the float4 source arrays
    int32_t tgt_base;      // first (padded) target in tgt4
    int32_t tgt_pad;       // padded target count = n_split * chunk
    int32_t n_split, chunk;
    int32_t key_base;
    int32_t n_src_blocks;
};

struct IcpPlan {
    int npairs = 0, S = 4;
    std::vector<PairGeom> g;
    mutable std::vector<NNWork> nn;   // (single pair on the cell list: filled only if the search ever needs the list pass)
    int per_block = 0;
    size_t nn_count = 0;
    std::vector<RedWork> red;
    std::vector<PairRed> pred;
    int64_t total_src = 0, total_tgt_pad = 0, total_keys = 0;
    bool shared_target = false;
    bool grid = false;      // exact cell-list search (single pair) with brute-force list fallback
    bool gridb = false;     // batched cell lists, one per pair (in-wave brute-force fallback after rcap shells)
    bool src_in_cell_order = false;   // sources were re-ordered by a cell-list setup: .w carries the original index
    GridParams gp;
    int total_cells = 0;
    // fused cell-list pass: one row (= workgroup = chunk of PASS_BS sorted sources) table, pair by pair
    std::vector<GridPairDev> gpairs;
    std::vector<int32_t> row_pair;
    int total_rows = 0;
};
constexpr int PASS_CHUNK = 512;   // == PASS_BS of kss_device.hpp (device-only header)

// sweep work items of pair p: (block of sources) x (target split)
static void fill_nn_table(const IcpPlan& pl, int p) {
    const PairGeom& g = pl.g[p];
    const int per_block = pl.per_block;
    for (int b = 0; b < g.n_src_blocks; ++b)
        for (int s = 0; s < g.n_split; ++s) {
            NNWork w;
            w.pair = p;
            w.src_begin = g.src_base + b * per_block;
            w.src_count = (int32_t)std::min<int64_t>(per_block, g.ns - (int64_t)b * per_block);
            w.tgt_begin = g.tgt_base + s * g.chunk;
            w.tgt_count = g.chunk;
            w.tgt_pair_base = g.tgt_base;
            w.key_begin = g.key_base + (int32_t)((int64_t)s * g.ns) + b * per_block;
            w.write_src = s == 0;
            if (pl.grid) {   // LIST semantics (kss_kernels.hip): offsets into the unresolved list, key row 0
                w.src_begin = b * per_block;
                w.key_begin = g.key_base;
                w.write_src = g.src_base;
            }
            pl.nn.push_back(w);
        }
}

static int build_plan(kss_ctx* c, const int64_t* ns, const int64_t* nt, int npairs, bool shared_target,
               int S_req, int split_req, int nn_mode, IcpPlan& pl) {
    pl.npairs = npairs;
    pl.shared_target = shared_target;
    if (nn_mode == KSS_NN_AUTO) nn_mode = c->nn_mode;
    // AUTO: the fused cell-list pass (one launch + a spin per iteration) beats sweep + reduce + finalize (three launches)
    // down to a few hundred points (1.4k x 1.4k: 22 vs 36 us per iteration); below that the build (~0.1 ms) is not paid
    // back.  KSS_GRID_MIN_* : tuning hooks.  Pairs sharing one target (the candidate batch of kss_register, badly posed
    // by construction) stay on the brute-force engine.
    static const int64_t min_nt = [] { const char* e = getenv("KSS_GRID_MIN_NT"); return e ? (int64_t)atoll(e) : (int64_t)512; }();
    static const int64_t min_ns = [] { const char* e = getenv("KSS_GRID_MIN_NS"); return e ? (int64_t)atoll(e) : (int64_t)512; }();
    pl.grid = nn_mode == KSS_NN_GRID || (nn_mode == KSS_NN_AUTO && npairs == 1 && nt[0] >= min_nt && ns[0] >= min_ns);
    if (npairs != 1) pl.grid = false;
    // (Pairs sharing ONE target -- the candidate batch of kss_register -- stay on the brute-force engine: measured this round
    // on the fused batch kernel too, 14 candidates x 1406 points: the candidates start from local minima of the rotation
    // search, a few percent of their sources need the in-wave fallback EVERY pass, 123 us per pass against 25 us: 14.3 vs 4.5 ms.)
    if (npairs > 1 && !shared_target) {
        // batch: one cell list per pair.  Measured against the brute-force batch (tools/batch_small.py): brute force wins
        // at 16 pairs x 600 points, they tie at 4 x 1500, the cell lists win from 16 x 1500 and 8 x 4000 up; badly posed
        // batches move back to brute force after one pass (icp_loop).
        int64_t least_nt = nt[0], tot_ns = 0;
        for (int p = 0; p < npairs; ++p) { least_nt = std::min(least_nt, nt[p]); tot_ns += ns[p]; }
        pl.gridb = nn_mode == KSS_NN_GRID || (nn_mode == KSS_NN_AUTO && least_nt >= 2 * min_nt && tot_ns >= 32 * min_ns);
    }
    if (npairs > 1 && shared_target && nn_mode != KSS_NN_BRUTE) {
        // Pairs sharing ONE target (the candidate batch of kss_register) on the pair-resident engine -- every candidate a workgroup
        // of its own with the (small) target in LDS, no lockstep between candidates -- was built and measured in round 3: Bunny,
        // 14 candidates x 1406 points: kss_register 24.3 ms against 4.5 ms on the brute-force engine.  The candidates start from
        // local minima of the rotation search and move far in every pass: nearly every source searches in every pass, and 1406
        // cell-list searches on ONE compute unit (50-100 us) lose to the tiled sweep of the same candidate spread over the chip
        // (25 us).  KSS_RESIDENT_SHARED=1 runs it anyway (A/B; same results).
        static const bool want_res = getenv("KSS_RESIDENT_SHARED") != nullptr && atoi(getenv("KSS_RESIDENT_SHARED")) != 0;
        bool fits = want_res && nt[0] >= 64 && nt[0] <= 12000;
        for (int p = 0; p < npairs; ++p) fits = fits && ns[p] <= (int64_t)RES_SMAX * RES_THREADS;
        pl.gridb = fits && resident_gate_available(c, npairs);
    }
    const bool any_grid = pl.grid || pl.gridb;
    pl.src_in_cell_order = any_grid;
    int64_t tot = 0;
    for (int p = 0; p < npairs; ++p) {
        if (ns[p] <= 0 || nt[p] <= 0) return set_err(c, KSS_ERR_ARG, "empty cloud in ICP pair");
        if (ns[p] > 0x7fff0000ll || nt[p] > 0x7fff0000ll) return set_err(c, KSS_ERR_ARG, "problem too large for 32-bit indexing");
        tot += ns[p];
    }
    int S = S_req;
    int64_t max_tiles = 1;
    for (int p = 0; p < npairs; ++p) max_tiles = std::max<int64_t>(max_tiles, (nt[p] + NN_TILE - 1) / NN_TILE);
    auto blocks_for = [&](int s) { int64_t b = 0; for (int p = 0; p < npairs; ++p) b += (ns[p] + (int64_t)NN_THREADS * s - 1) / ((int64_t)NN_THREADS * s); return b; };
    bool small = false;
    if (S != 1 && S != 2 && S != 4 && S != 8) {
        S = tot >= 32768 ? 4 : (tot >= 8192 ? 2 : 1);
        // small problems (a few 1-2k-point pairs: the candidate batch of kss_register) cannot fill the chip even with one
        // target tile per workgroup: fewer sources per lane and single-tile splits give 4x the workgroups (measured: the
        // sweep of 14 pairs x 1.4k points 38 -> 10 us)
        while (S > 1 && blocks_for(S) * max_tiles < 2048) S /= 2;
        small = blocks_for(S) * max_tiles < 2048;
    }
    pl.S = S;
    const int per_block = NN_THREADS * S;
    const int64_t src_blocks_total = blocks_for(S);
    // enough workgroups to keep 256 CUs x 8 resident workgroups busy with >= 2 rounds
    int64_t want_split = 1;
    if (src_blocks_total < 2048) want_split = (4096 + src_blocks_total - 1) / src_blocks_total;
    if (split_req > 0) want_split = split_req;

    pl.g.resize(npairs);
    int64_t sb = 0, tb = 0, kb = 0;
    for (int p = 0; p < npairs; ++p) {
        PairGeom& g = pl.g[p];
        g.ns = ns[p]; g.nt = nt[p];
        const int64_t tiles = (nt[p] + NN_TILE - 1) / NN_TILE;
        int64_t split = std::min<int64_t>(want_split, std::max<int64_t>(1, small ? tiles : tiles / 2));
        int64_t chunk_tiles = (tiles + split - 1) / split;
        split = (tiles + chunk_tiles - 1) / chunk_tiles;
        g.n_split = (int32_t)split;
        g.chunk = (int32_t)(chunk_tiles * NN_TILE);
        g.tgt_pad = g.n_split * g.chunk;
        g.src_base = (int32_t)sb;
        g.key_base = (int32_t)kb;
        g.n_src_blocks = (int32_t)((ns[p] + per_block - 1) / per_block);
        if (shared_target && p > 0) {
            g.tgt_base = pl.g[0].tgt_base;
        } else {
            g.tgt_base = (int32_t)tb;
            tb += g.tgt_pad;
        }
        sb += ns[p];
        kb += any_grid ? ns[p] : (int64_t)g.n_split * ns[p];
        if (sb > 0x7fff0000ll || tb > 0x7fff0000ll || kb > 0x7fff0000ll)
            return set_err(c, KSS_ERR_ARG, "problem too large for 32-bit indexing");
    }
    pl.total_src = sb; pl.total_tgt_pad = tb; pl.total_keys = kb;
    pl.gpairs.clear(); pl.row_pair.clear(); pl.total_rows = 0;
    if (any_grid) {
        pl.gpairs.resize(npairs);
        for (int p = 0; p < npairs; ++p) {
            GridPairDev& gd = pl.gpairs[p];
            std::memset(&gd, 0, sizeof gd);
            gd.tgt_base = pl.g[p].tgt_base; gd.tgt_n = (int32_t)pl.g[p].nt; gd.tgt_pad = pl.g[p].tgt_pad;
            gd.src_base = pl.g[p].src_base; gd.src_n = (int32_t)pl.g[p].ns;
            gd.row_base = pl.total_rows;
            gd.n_rows = (int32_t)((pl.g[p].ns + PASS_CHUNK - 1) / PASS_CHUNK);
            for (int r = 0; r < gd.n_rows; ++r) pl.row_pair.push_back(p);
            pl.total_rows += gd.n_rows;
        }
    }

    pl.nn.clear(); pl.red.clear(); pl.pred.resize(npairs);
    pl.per_block = per_block;
    pl.nn_count = 0;
    int32_t prow = 0;
    for (int p = 0; p < npairs; ++p) {
        const PairGeom& g = pl.g[p];
        if (!pl.gridb) pl.nn_count += (size_t)g.n_src_blocks * g.n_split;
        if (!pl.gridb && !pl.grid) fill_nn_table(pl, p);   // (cell-list single pair: ~4000 entries nobody reads unless a source falls back; built then)
        pl.pred[p].first = prow;
        // reduce workgroups own 256*R consecutive sources (R = 1 up to 131k sources: the reduce is latency bound,
        // it wants many workgroups; the 240-lane final reduction handles hundreds of rows in a few microseconds)
        int64_t R = (g.ns + 256 * 512 - 1) / (256 * 512);   // <= ~512 partial rows per pair
        R = std::max<int64_t>(1, std::min<int64_t>(R, 64));
        const int64_t rchunk = 256 * R;
        const int nrb = (pl.gridb || pl.grid) ? 0 : (int)((g.ns + rchunk - 1) / rchunk);   // (the cell-list engines reduce inside the fused pass)
        for (int b = 0; b < nrb; ++b) {
            RedWork r;
            r.pair = p;
            r.src_begin = g.src_base + (int32_t)(b * rchunk);
            r.src_count = (int32_t)std::min<int64_t>(rchunk, g.ns - (int64_t)b * rchunk);
            r.key_begin = g.key_base + (int32_t)(b * rchunk);
            r.key_stride = (int32_t)g.ns;
            r.n_split = any_grid ? 1 : g.n_split;
            r.tgt_pair_base = g.tgt_base;
            r.partial_index = prow++;
            pl.red.push_back(r);
        }
        pl.pred[p].count = nrb;
    }
    return KSS_OK;
}

// Work tables of the sweep / reduce kernels (uploaded lazily on the fused cell-list path).
static int stage_tables(kss_ctx* c, const IcpPlan& pl) {
    if (c->tables_staged) return KSS_OK;
    if (pl.grid && pl.nn.empty())
        for (int p = 0; p < pl.npairs; ++p) fill_nn_table(pl, p);
    HIPCHK(c, hipMemcpyAsync(c->nn_work.p, pl.nn.data(), pl.nn.size() * sizeof(NNWork), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->red_work.p, pl.red.data(), pl.red.size() * sizeof(RedWork), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->pair_red.p, pl.pred.data(), pl.pred.size() * sizeof(PairRed), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));   // pageable sources: do not outlive this call unsynchronised
    c->tables_staged = true;
    return KSS_OK;
}

// Upload plan tables and size the workspace.
static int stage_plan(kss_ctx* c, const IcpPlan& pl) {
    KCHK(ensure(c, c->tgt4, (size_t)pl.total_tgt_pad * sizeof(float4)));
    KCHK(ensure(c, c->src0, (size_t)pl.total_src * sizeof(float4)));
    KCHK(ensure(c, c->cur[0], (size_t)pl.total_src * sizeof(float4)));
    KCHK(ensure(c, c->cur[1], (size_t)pl.total_src * sizeof(float4)));
    KCHK(ensure(c, c->keys, (size_t)pl.total_keys * sizeof(unsigned long long)));
    KCHK(ensure_zeroed(c, c->partials, std::max<size_t>(pl.red.size(), 2 * (size_t)pl.total_rows) * NSUMS * sizeof(double)));   // (x2: rows as 16-byte granules; zeroed when (re)allocated: no stale {bits, number} granule of an earlier owner of the block)
    KCHK(ensure(c, c->sums, (size_t)pl.npairs * NSUMS * sizeof(double)));
    KCHK(ensure(c, c->nn_work, pl.nn_count * sizeof(NNWork)));
    KCHK(ensure(c, c->red_work, pl.red.size() * sizeof(RedWork)));
    KCHK(ensure(c, c->pair_red, pl.pred.size() * sizeof(PairRed)));
    KCHK(ensure(c, c->state, (size_t)pl.npairs * sizeof(PairState)));
    KCHK(ensure_zeroed(c, c->pair_ticket, (size_t)std::max(pl.npairs, PUB_PAIRS) * sizeof(int32_t)));   // zero at rest (kss_ctx.hpp)
    KCHK(ensure_pinned(c, c->h_sums, c->h_sums_cap, (size_t)pl.npairs * NSUMS * sizeof(double)));
    KCHK(ensure_pinned(c, c->h_state, c->h_state_cap, (size_t)pl.npairs * sizeof(PairState)));
    KCHK(ensure_pub(c, pl.npairs));
    c->tables_staged = false;
    // (the fused cell-list path needs the tables only if a query falls back, the candidate kernels not at all: nn_pass stages
    // them the first time its sweep + reduce form runs)
    if (!pl.grid && !pl.shared_target) KCHK(stage_tables(c, pl));
    return KSS_OK;
}

// Pack the clouds of every pair into the float4 workspace (targets sentinel padded).
// dtype: KSS_F32 / KSS_F64 packed triples on the DEVICE; src_off/tgt_off in points.
static int pack_clouds(kss_ctx* c, const IcpPlan& pl, const void* d_src, const int64_t* src_off, const void* d_tgt,
                const int64_t* tgt_off, int dtype) {
    const size_t esz = dtype == KSS_F64 ? sizeof(double) : sizeof(float);
    if (pl.grid) {   // single pair on the cell list: both clouds and the target's bbox partials in one launch
        const PairGeom& g = pl.g[0];
        const int nbb = pack_pair_bbox_rows(g.tgt_pad);
        KCHK(ensure(c, c->g_bbox, (size_t)nbb * 6 * sizeof(float)));
        // up to 8192 partials (2M targets, 256 KB over PCIe) also go straight to host memory as checked granules (grid_setup)
        static const bool box_host = !(getenv("KSS_BBOX_HOST") && atoi(getenv("KSS_BBOX_HOST")) == 0);
        c->box_tag = 0; c->box_rows = 0;
        if (box_host && nbb <= 8192) {
            const size_t want = (size_t)nbb * 32;
            if (want > c->h_box_bytes) {
                if (c->h_box) { HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipHostFree(c->h_box)); c->h_box = nullptr; c->h_box_dev = nullptr; c->h_box_bytes = 0; }
                void* hp = nullptr; void* dp = nullptr;
                const size_t bytes = want + want / 2;
                if (hipHostMalloc(&hp, bytes, hipHostMallocMapped) == hipSuccess && hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess) {
                    std::memset(hp, 0, bytes);
                    c->h_box = (unsigned int*)hp; c->h_box_dev = (unsigned int*)dp; c->h_box_bytes = bytes;
                } else { if (hp) hipHostFree(hp); (void)hipGetLastError(); }
            }
            if (c->h_box) {
                c->box_tag = (unsigned)(++c->seq);
                if (c->box_tag == 0) c->box_tag = (unsigned)(++c->seq);
                c->box_rows = nbb;
            }
        }
        launch_pack_pair(c->stream, dtype, (const char*)d_tgt + (size_t)tgt_off[0] * 3 * esz, g.nt, (float4*)c->tgt4.p + g.tgt_base, g.tgt_pad,
                         (float*)c->g_bbox.p, (const char*)d_src + (size_t)src_off[0] * 3 * esz, g.ns, (float4*)c->src0.p + g.src_base,
                         c->box_tag ? c->h_box_dev : nullptr, c->box_tag);
        HIPCHK(c, hipGetLastError());
        return KSS_OK;
    }
    // sources are contiguous in both layouts
    {
        const char* base = (const char*)d_src + (size_t)src_off[0] * 3 * esz;
        if (dtype == KSS_F64) launch_pack_f64_to_f4(c->stream, (const double*)base, pl.total_src, (float4*)c->src0.p, pl.total_src, false);
        else launch_pack_f3_to_f4(c->stream, (const float*)base, pl.total_src, (float4*)c->src0.p, pl.total_src, false);
    }
    const int ntp = pl.shared_target ? 1 : pl.npairs;
    if (ntp > 8) {   // many pairs: one launch over a per-pair segment table (segments are laid out back to back in tgt4)
        std::vector<PackSeg> seg((size_t)ntp);
        for (int p = 0; p < ntp; ++p) { seg[p].in_off = tgt_off[p]; seg[p].out_base = pl.g[p].tgt_base; seg[p].n = pl.g[p].nt; }
        KCHK(ensure(c, c->pack_seg, seg.size() * sizeof(PackSeg)));
        HIPCHK(c, hipMemcpyAsync(c->pack_seg.p, seg.data(), seg.size() * sizeof(PackSeg), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));   // `seg` is pageable and about to go out of scope
        launch_pack_batch(c->stream, d_tgt, dtype, (const PackSeg*)c->pack_seg.p, ntp, pl.total_tgt_pad, (float4*)c->tgt4.p);
        HIPCHK(c, hipGetLastError());
        return KSS_OK;
    }
    for (int p = 0; p < ntp; ++p) {
        const PairGeom& g = pl.g[p];
        const char* base = (const char*)d_tgt + (size_t)tgt_off[p] * 3 * esz;
        float4* out = (float4*)c->tgt4.p + g.tgt_base;
        if (dtype == KSS_F64) launch_pack_f64_to_f4(c->stream, (const double*)base, g.nt, out, g.tgt_pad, true);
        else launch_pack_f3_to_f4(c->stream, (const float*)base, g.nt, out, g.tgt_pad, true);
    }
    HIPCHK(c, hipGetLastError());
    return KSS_OK;
}

static void choose_cells(const float mn[3], const float mx[3], int64_t nt, GridParams& gp) {
    const float ext[3] = {mx[0] - mn[0], mx[1] - mn[1], mx[2] - mn[2]};
    const float emax = std::max(ext[0], std::max(ext[1], ext[2]));
    // cell edge: a handful of points per occupied cell if the target is a surface (area ~ emax^2 * few)
    float hscale = 2.0f;
    if (const char* e = getenv("KSS_GRID_HSCALE")) { const float v = (float)atof(e); if (v > 0.05f && v < 50.f) hscale = v; }   // tuning hook
    float h = emax * hscale * std::sqrt(3.0f / (float)nt);
    h = std::max(h, emax / 255.5f);
    if (!(h > 0.f)) h = 1.f;   // all targets coincide
    gp.ox = mn[0]; gp.oy = mn[1]; gp.oz = mn[2];
    gp.h = h; gp.inv_h = 1.0f / h;
    gp.gx = std::max(1, std::min(256, (int)std::floor(ext[0] / h) + 1));
    gp.gy = std::max(1, std::min(256, (int)std::floor(ext[1] / h) + 1));
    gp.gz = std::max(1, std::min(256, (int)std::floor(ext[2] / h) + 1));
    const float mag = std::max(std::max(std::fabs(mn[0]), std::fabs(mx[0])), std::max(std::max(std::fabs(mn[1]), std::fabs(mx[1])), std::max(std::fabs(mn[2]), std::fabs(mx[2]))));
    gp.eps = 2e-6f * (mag + emax) + 1e-30f;
    gp.rcap = 4;   // shells before a query goes to the brute-force list (measured on badly initialised pairs: tools/hard_case.py)
    if (const char* e = getenv("KSS_GRID_RCAP")) { const int v = atoi(e); if (v >= 1 && v <= 64) gp.rcap = v; }   // tuning hook
}

// Build the uniform cell list over the (single) target: bbox -> cell size -> counting sort.
static int grid_setup(kss_ctx* c, IcpPlan& pl) {
    if (!pl.grid) return KSS_OK;
    const PairGeom& g = pl.g[0];
    const int nt = (int)g.nt, ns = (int)g.ns;
    const int nbb = pack_pair_bbox_rows(g.tgt_pad);   // one partial per 256 target slots, left by pack_clouds
    const float4* tgt = (const float4*)c->tgt4.p + g.tgt_base;
    ProfScope ps(c, KSS_K_GRID_BUILD);
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    bool have_box = false;
    if (c->box_tag != 0 && c->box_rows == nbb) {
        // the pack kernel's partials as checked granules in host memory: {three floats, tag + check word}, one for the minimum
        // and one for the maximum of every 256 target slots; a granule that is not there yet (or seen torn) is read again
        const unsigned long long* hq = (const unsigned long long*)c->h_box;
        const auto t0 = std::chrono::steady_clock::now();
        int g = 0;
        for (long spin = 0; g < 2 * nbb; ++spin) {
            const unsigned long long w0 = __atomic_load_n(&hq[2 * g], __ATOMIC_ACQUIRE), w1 = __atomic_load_n(&hq[2 * g + 1], __ATOMIC_ACQUIRE);
            const unsigned x = (unsigned)w0, y = (unsigned)(w0 >> 32), z = (unsigned)w1, tag = (unsigned)(w1 >> 32);
            if (tag - kss_mix3(x, y, z) == c->box_tag) {
                float f[3];
                std::memcpy(&f[0], &x, 4); std::memcpy(&f[1], &y, 4); std::memcpy(&f[2], &z, 4);
                for (int k = 0; k < 3; ++k) { if (g & 1) mx[k] = std::max(mx[k], f[k]); else mn[k] = std::min(mn[k], f[k]); }
                ++g;
                continue;
            }
            __builtin_ia32_pause();
            if ((spin & 4095) == 4095 && std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > 20.0) break;   // (a faulted launch: the copy below reports it)
        }
        have_box = g == 2 * nbb;
        if (!have_box) { for (int k = 0; k < 3; ++k) { mn[k] = INFINITY; mx[k] = -INFINITY; } }
    }
    if (!have_box) {
        std::vector<float>& hb = c->h_bbox;
        hb.resize((size_t)nbb * 6);
        HIPCHK(c, hipMemcpyAsync(hb.data(), c->g_bbox.p, hb.size() * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        for (int b = 0; b < nbb; ++b)
            for (int k = 0; k < 3; ++k) { mn[k] = std::min(mn[k], hb[(size_t)b * 6 + k]); mx[k] = std::max(mx[k], hb[(size_t)b * 6 + 3 + k]); }
    }
    if (!(std::isfinite(mn[0]) && std::isfinite(mn[1]) && std::isfinite(mn[2]) && std::isfinite(mx[0]) && std::isfinite(mx[1]) && std::isfinite(mx[2])))
        return set_err(c, KSS_ERR_ARG, "non-finite target coordinates");
    GridParams gp;
    choose_cells(mn, mx, nt, gp);
    pl.gp = gp;
    const size_t ncells = (size_t)gp.gx * gp.gy * gp.gz;
    KCHK(ensure_zeroed(c, c->g_counts, 2 * ncells * sizeof(int32_t)));   // target cells, then source cells; zero at rest (kss_ctx.hpp)
    KCHK(ensure(c, c->g_start, (2 * ncells + 8) * sizeof(int32_t)));   // [0] pad, target starts at [1 .. ncells + 1], source starts (+ nt) behind
    KCHK(ensure(c, c->g_bsums, scan_scratch_bytes((int)(2 * ncells))));
    KCHK(ensure(c, c->g_sorted, (size_t)nt * sizeof(float4)));
    KCHK(ensure(c, c->g_list, (size_t)ns * sizeof(int32_t)));
    KCHK(ensure(c, c->g_pos, (size_t)ns * sizeof(float4)));   // last winner of every source (written by the first pass before any pass reads it)
    c->nn_have = false;
    KCHK(ensure(c, c->g_nnst, (size_t)ns * sizeof(float2)));   // ... and its skip state (written by the first pass before any pass reads it)
    KCHK(ensure_zeroed(c, c->g_count, 64));   // [0] unresolved-list length: zero at rest
    if (getenv("KSS_COUNTS_CHECK")) {   // diagnostic: the zero-at-rest invariant of the cell counters, checked on the host
        HIPCHK(c, hipStreamSynchronize(c->stream));
        std::vector<int32_t> h(2 * ncells);
        HIPCHK(c, hipMemcpy(h.data(), c->g_counts.p, h.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
        size_t bad = 0, first = 0;
        for (size_t k = 0; k < h.size(); ++k)
            if (h[k] != 0) { if (!bad) first = k; ++bad; }
        if (bad) {
            std::fprintf(stderr, "[kss] counts check: %zu cells x 2 (%d x %d x %d), buffer %zu bytes, %zu non-zero counters, first at %zu = %d\n", ncells,
                         gp.gx, gp.gy, gp.gz, c->g_counts.cap, bad, first, h[first]);
            return set_err(c, KSS_ERR_HIP, "cell counters not zero at rest");
        }
    }
    pl.gpairs[0].gp = gp;
    pl.gpairs[0].cell_base = 0;
    // both cell lists by shared launches; the sources end up in src0 in the target's cell order (original index in .w);
    // cur[0] is the scatter's scratch
    launch_grid_build_pair(c->stream, tgt, nt, (float4*)c->src0.p + g.src_base, ns, gp, (int32_t*)c->g_counts.p, (int32_t*)c->g_start.p + 1,
                           (int32_t*)c->g_bsums.p, (float4*)c->g_sorted.p, (float4*)c->cur[0].p);
    HIPCHK(c, hipGetLastError());
    c->grid_stats[0] = gp.h; c->grid_stats[1] = gp.gx; c->grid_stats[2] = gp.gy; c->grid_stats[3] = gp.gz;
    if (c->stats_ns != ns || c->stats_nt != nt) { c->grid_stats[4] = 0; c->grid_stats[5] = 0; }
    c->grid_stats[6] = ns; c->grid_stats[7] = nt;
    if (c->prof && (c->stats_ns != ns || c->stats_nt != nt)) {   // one extra small kernel, once per problem size while profiling
        c->stats_ns = ns; c->stats_nt = nt;
        KCHK(ensure(c, c->scratch_c, 64));
        HIPCHK(c, hipMemsetAsync(c->scratch_c.p, 0, 16, c->stream));
        launch_grid_stats(c->stream, (const float4*)c->src0.p + g.src_base, ns, gp, (const int32_t*)c->g_start.p + 1, (unsigned long long*)c->scratch_c.p);
        unsigned long long hst[2] = {0, 0};
        HIPCHK(c, hipMemcpyAsync(hst, c->scratch_c.p, 16, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->grid_stats[5] = (double)hst[0];
        c->grid_stats[4] = (double)hst[1];
    }
    return KSS_OK;
}

// Batched cell lists: one per pair, all built by the same launches.
static int grid_setup_batch(kss_ctx* c, IcpPlan& pl) {
    if (!pl.gridb) return KSS_OK;
    const int np = pl.npairs;
    std::vector<GridPairDev>& hp = pl.gpairs;   // segments and rows filled by build_plan; cell lists chosen below
    int64_t sum_nt = 0;
    for (int p = 0; p < np; ++p) sum_nt += pl.g[p].nt;
    ProfScope ps(c, KSS_K_GRID_BUILD);
    KCHK(ensure(c, c->g_pairs, (size_t)np * sizeof(GridPairDev)));
    KCHK(ensure(c, c->g_bbox, (size_t)np * 6 * sizeof(float)));
    HIPCHK(c, hipMemcpyAsync(c->g_pairs.p, hp.data(), (size_t)np * sizeof(GridPairDev), hipMemcpyHostToDevice, c->stream));
    launch_gridb_bbox(c->stream, (const float4*)c->tgt4.p, (const GridPairDev*)c->g_pairs.p, np, (float*)c->g_bbox.p);
    std::vector<float> hb((size_t)np * 6);
    HIPCHK(c, hipMemcpyAsync(hb.data(), c->g_bbox.p, hb.size() * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    int64_t cells = 0, sorted_base = 0;
    int most_cells = 0;
    for (int p = 0; p < np; ++p) {
        const float* b = &hb[(size_t)p * 6];
        for (int k = 0; k < 6; ++k)
            if (!std::isfinite(b[k])) return set_err(c, KSS_ERR_ARG, "non-finite target coordinates");
        choose_cells(b, b + 3, pl.g[p].nt, hp[p].gp);
        hp[p].cell_base = (int32_t)cells;
        hp[p].sorted_base = (int32_t)sorted_base;
        const int64_t nc = (int64_t)hp[p].gp.gx * hp[p].gp.gy * hp[p].gp.gz;
        most_cells = (int)std::max<int64_t>(most_cells, nc);
        cells += nc;
        sorted_base += pl.g[p].nt;
        if (cells > 0x7fff0000ll) return set_err(c, KSS_ERR_ARG, "batch cell lists exceed 32-bit indexing");
    }
    pl.total_cells = (int)cells;
    HIPCHK(c, hipMemcpyAsync(c->g_pairs.p, hp.data(), (size_t)np * sizeof(GridPairDev), hipMemcpyHostToDevice, c->stream));
    KCHK(ensure(c, c->g_start, ((size_t)cells + 8) * sizeof(int32_t)));   // [0] pad, starts at [1 ..], pads behind
    KCHK(ensure(c, c->g_sorted, (size_t)sum_nt * sizeof(float4)));
    KCHK(ensure(c, c->g_pos, (size_t)pl.total_src * sizeof(float4)));   // last winners (no initialisation: the first pass does not read them)
    c->nn_have = false;
    KCHK(ensure(c, c->g_nnst, (size_t)pl.total_src * sizeof(float2)));
    KCHK(ensure(c, c->g_rowpair, pl.row_pair.size() * sizeof(int32_t)));
    HIPCHK(c, hipMemcpyAsync(c->g_rowpair.p, pl.row_pair.data(), pl.row_pair.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    static const bool no_lds = getenv("KSS_GRIDB_NOLDS") != nullptr;   // A/B switch: always the global-atomic build
    bool half = true;   // 16-bit counters: counts and in-pair positions of every pair fit
    for (int p = 0; p < np; ++p) half = half && pl.g[p].nt < 65536 && pl.g[p].ns < 65536;
    static const bool no_half = getenv("KSS_GRIDB_NOHALF") != nullptr;   // A/B switch: 32-bit counters
    if (no_half) half = false;
    // every pair's counters fit a CU's LDS: one workgroup per pair builds both of its lists (kss_grid.hip)
    const bool built_in_lds = most_cells <= gridb_lds_max_cells(half) && !no_lds &&
        launch_gridb_build_lds(c->stream, (const float4*)c->tgt4.p, (float4*)c->src0.p, (float4*)c->cur[0].p, (const GridPairDev*)c->g_pairs.p, np,
                               (int32_t*)c->g_start.p + 1, (float4*)c->g_sorted.p, (int)most_cells, half);
    if (!built_in_lds) {
        // (zero at rest, as the single-pair build expects of this buffer: a plain ensure() here once left the slack behind
        // `cells` uninitialised, and a later single pair with more cells scanned garbage counts -- see DESIGN.md, incidents)
        KCHK(ensure_zeroed(c, c->g_counts, (size_t)cells * sizeof(int32_t)));
        KCHK(ensure(c, c->g_start2, ((size_t)cells + 1) * sizeof(int32_t)));
        KCHK(ensure(c, c->g_bsums, scan_scratch_bytes((int)cells)));
        if ((cells + 4095) / 4096 > 1024 * 16) return set_err(c, KSS_ERR_ARG, "batch cell lists too large for the scan");
        launch_gridb_build_targets(c->stream, (const float4*)c->tgt4.p, (int)pl.total_tgt_pad, (const GridPairDev*)c->g_pairs.p, np,
                                   (int)cells, (int32_t*)c->g_counts.p, (int32_t*)c->g_start.p + 1,
                                   (int32_t*)c->g_bsums.p, (float4*)c->g_sorted.p);
        launch_gridb_sort_sources(c->stream, (const float4*)c->src0.p, (int)pl.total_src, (const GridPairDev*)c->g_pairs.p, np, (int)cells,
                                  (int32_t*)c->g_counts.p, (int32_t*)c->g_start2.p, (int32_t*)c->g_bsums.p,
                                  (float4*)c->cur[0].p, (float4*)c->src0.p);   // (scatter: src0 -> cur[0]; rank fix: cur[0] -> src0)
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));   // hb is about to go out of scope (pageable)
    return KSS_OK;
}

// Host-mapped result slots: PUB_PAIRS x NSUMS x {bits(sum), sequence number}.  The kernel that ends an ICP iteration
// (fused grid_nn_kernel, or finalize_sums_kernel for small batches) stores every sum TOGETHER with the launch's
// sequence number as one aligned 16-byte write: a slot whose sequence number matches holds this launch's value, so
// there is no separate completion flag and no write-acknowledge round trip between "sums stored" and "flag stored"
// on the device.
static int ensure_pub(kss_ctx* c, int npairs) {
    const size_t want = (size_t)std::max(npairs, PUB_PAIRS) * (NSUMS + 1) * 16;   // (+1: the resident engines' exit flag of the pair, behind the launch's slots)
    if (c->h_seq && want <= c->h_seq_bytes) return KSS_OK;
    if (c->h_seq) {   // grow: nothing may still be writing the old slots
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipHostFree(c->h_seq));
        c->h_seq = nullptr; c->h_seq_dev = nullptr; c->h_seq_bytes = 0;
    }
    void* p = nullptr;
    const size_t bytes = want + want / 4;
    if (hipHostMalloc(&p, bytes, hipHostMallocMapped) != hipSuccess) return set_err(c, KSS_ERR_NOMEM, "hipHostMalloc(result slots)");
    c->h_seq = (unsigned long long*)p;
    c->h_seq_bytes = bytes;
    std::memset(p, 0, bytes);
    void* d = nullptr;
    HIPCHK(c, hipHostGetDevicePointer(&d, p, 0));
    c->h_seq_dev = (unsigned long long*)d;
    return KSS_OK;
}

// Wait for the first `npairs * NSUMS` slots to carry sequence number c->seq and copy the sums to h_sums, where the rest
// of the loop expects them.  The host spins (a stream sync costs a 5-10 us wake-up per ICP iteration); after ~2 ms
// without progress it falls back to the stream sync, which also surfaces a faulted kernel instead of spinning forever.
static void gated_cancel(kss_ctx* c);

// A result slot is ONE 16-byte device store {bits of the f64, low half of the launch number, check word of those three}.  The
// host reads it as two 8-byte loads; it takes the value when the number matches AND the check word fits the bits it read -- a
// slot seen torn (new number, old bits or the reverse) fails the check and is simply read again.
static inline bool slot_value(const unsigned long long* sl, int k, unsigned long long want, unsigned long long* bits_out) {
    const unsigned long long w = __atomic_load_n(&sl[2 * k + 1], __ATOMIC_ACQUIRE);
    if ((unsigned)w != (unsigned)want) return false;
    const unsigned long long bits = __atomic_load_n(&sl[2 * k], __ATOMIC_RELAXED);
    if ((unsigned)(w >> 32) != kss_mix3((unsigned)bits, (unsigned)(bits >> 32), (unsigned)w)) return false;
    *bits_out = bits;
    return true;
}

// the 20 sums of pair p of launch `want`, if they have all landed
static inline bool collect_pair(kss_ctx* c, int p, unsigned long long want) {
    const unsigned long long* sl = c->h_seq + (size_t)2 * NSUMS * p;
    unsigned long long bits;
    if (!slot_value(sl, NSUMS - 1, want, &bits)) return false;   // the highest slot is usually the last to land
    double* out = (double*)c->h_sums + (size_t)NSUMS * p;
    for (int k = NSUMS - 1; k >= 0; --k) {
        if (!slot_value(sl, k, want, &bits)) return false;
        std::memcpy(&out[k], &bits, sizeof(double));
    }
    return true;
}

// wait for pair p's sums (spin; after ~2 ms without them: open any gate, synchronize the stream, look once more)
static int wait_pair(kss_ctx* c, int p, unsigned long long want) {
    for (long spin = 0; spin < 2000000; ++spin) {
        if (collect_pair(c, p, want)) return KSS_OK;
        __builtin_ia32_pause();
    }
    return KSS_ERR_HIP;   // the caller synchronizes (only one thread may) and retries
}

// not_published (optional): set when the stream drained without error and the slots still do not carry `want` -- the caller
// may know how to go on (a gated launch that gave up waiting for a stalled host thread has not touched anything)
static int wait_seq(kss_ctx* c, int npairs = 1, unsigned long long want = 0, const int* active = nullptr, bool* not_published = nullptr) {
    if (!want) want = c->seq;
    bool slow = false;
    for (int p = npairs - 1; p >= 0; --p) {
        if (active && !active[p]) continue;
        if (!slow && wait_pair(c, p, want) == KSS_OK) continue;
        if (!slow) {
            gated_cancel(c);   // a pre-enqueued launch behind a closed gate would make the synchronize below wait forever
            HIPCHK(c, hipStreamSynchronize(c->stream));
            slow = true;
        }
        if (!collect_pair(c, p, want)) {
            if (not_published) *not_published = true;
            return set_err(c, KSS_ERR_HIP, "kernel finished without publishing its result");
        }
    }
    return KSS_OK;
}

// the first `nslots` result slots of the launch with sequence number c->seq, as doubles (pre-shape statistics)
int ensure_pub_slots(kss_ctx* c) { return ensure_pub(c); }
int wait_slots(kss_ctx* c, int nslots, double* out) {
    const unsigned long long want = c->seq;
    auto collect = [&]() -> bool {
        for (int k = nslots - 1; k >= 0; --k) {
            unsigned long long bits;
            if (!slot_value(c->h_seq, k, want, &bits)) return false;
            std::memcpy(&out[k], &bits, sizeof(double));
        }
        return true;
    };
    for (long spin = 0; spin < 2000000; ++spin) {
        if (collect()) return KSS_OK;
        __builtin_ia32_pause();
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));   // surfaces a faulted kernel instead of spinning forever
    if (!collect()) return set_err(c, KSS_ERR_HIP, "kernel finished without publishing its result");
    return KSS_OK;
}

// ---- gated launches of the fused single-pair pass -------------------------------------------------------------
// Per ICP iteration the host has to see the sums, solve, and only then does the next transform exist: hipLaunchKernel
// (~3 us on the host), the dispatch after it and the new workgroups' first loads all sat on the critical path.  With
// gating the NEXT iteration's kernel is enqueued while the current one runs; it starts as soon as the current one ends,
// fetches its sources and previous winners, and polls a host-mapped 64-byte record for its transform (kss_grid.hip,
// grid_pass_kernel).  When the sums are in, the host solves, writes the transform into the record and stamps it with the
// launch's number (one release store): the kernel goes on from there -- no stream operation, no second PCIe round trip
// for the transform.  (Round 1 parked the kernel behind a hipStreamWaitValue64 and let it fetch the transform afterwards:
// wait-kernel -> dispatch -> fetch were three dependent steps where this has one.)  A pre-enqueued kernel the loop does not
// need (convergence, the brute-force fallback, an error) is CANCELLED: stamped with pad[0] = 1, it leaves at once.
// Nothing ever waits on a cancelled kernel, no path returns with a kernel unanswered (GatedGuard), and the kernel's poll
// is bounded, so neither side can wait forever.  A polling kernel keeps its workgroups resident (200 of ~770 slots at
// C2): other streams still run.  Exclusions: only on a stream the context owns and only while it is the only context
// of the process (several loops polling at once could crowd each other's real work out of the CUs), never with an
// all-reduce callback (its collective would queue up on this stream BEHIND the polling kernel, whose transform depends
// on it).  KSS_GATED=0 turns it off.
static bool gated_available(kss_ctx* c) {
    static const bool want = getenv("KSS_GATED") == nullptr || atoi(getenv("KSS_GATED")) != 0;   // KSS_GATED=0 switches it off
    if (!want || !c->own_stream || kss_live_contexts().load() != 1) return false;
    if (c->gated.supported < 0) {
        c->gated.supported = 0;
        // the device-side granules: on a large-BAR system in FINE-GRAINED device memory, which the host can store into directly
        // (tools/bar_probe.hip: same 2.1 us host -> kernel -> host round trip as a kernel polling host memory, but every
        // workgroup can poll it: no asking workgroup, no re-publication hop).  KSS_GATE_BAR=0: always the re-publishing form.
        static const bool want_bar = getenv("KSS_GATE_BAR") == nullptr || atoi(getenv("KSS_GATE_BAR")) != 0;
        hipDeviceProp_t prop;
        void* bar = nullptr;
        if (want_bar && hipGetDeviceProperties(&prop, c->device) == hipSuccess && prop.isLargeBar &&
            hipExtMallocWithFlags(&bar, 4096, hipDeviceMallocFinegrained) == hipSuccess && bar) {
            if (hipMemset(bar, 0, 4096) == hipSuccess && hipDeviceSynchronize() == hipSuccess) {
                c->gate_bar = (unsigned int*)bar;
                c->gated.supported = 1;
                return true;
            }
            hipFree(bar);
        }
        (void)hipGetLastError();
        void *x = nullptr, *xd = nullptr;
        if (ensure(c, c->g_gate, 256) == KSS_OK && hipMemsetAsync(c->g_gate.p, 0, 256, c->stream) == hipSuccess &&
            hipHostMalloc(&x, 2 * sizeof(PairState), hipHostMallocMapped) == hipSuccess && hipHostGetDevicePointer(&xd, x, 0) == hipSuccess) {
            std::memset(x, 0, 2 * sizeof(PairState));
            c->h_xf = (PairState*)x; c->h_xf_dev = (PairState*)xd;
            c->gated.supported = 1;
        } else {
            (void)hipGetLastError();
            if (x) hipHostFree(x);
        }
    }
    return c->gated.supported == 1;
}

static void gated_release(kss_ctx* c, const PairState& st, int skip) {   // transform (or the cancel mark), stamped
    if (c->gate_bar) {
        // five 16-byte granules {3 words, stamp} straight into device memory through the BAR; each is one aligned 16-byte
        // store (never seen torn), so their order does not matter; the fence pushes them out of the write-combining buffers
        int words[16];
        std::memcpy(words, &st, sizeof st);
        words[14] = skip;                       // pad[0]
        unsigned int* slot = c->gate_bar + 32 * c->gated.slot;   // two records, 128 bytes apart
        // (test hook: the n-th record first arrives with a granule whose words do not fit its check -- what a torn 16-byte
        // store would look like -- and only a while later as it should be; the kernel must not take the first for data)
        static const long torn_at = getenv("KSS_TEST_TORN_GATE") ? atol(getenv("KSS_TEST_TORN_GATE")) : -1;
        static long released = 0;
        const bool torn = ++released == torn_at;
        for (int pass = torn ? 0 : 1; pass < 2; ++pass) {
            for (int g = 0; g < 5; ++g) {
                // the stamp word carries the check of the granule's three data words
                const unsigned tag = (unsigned)c->gated.stamp + kss_mix3((unsigned)words[3 * g], (unsigned)words[3 * g + 1], (unsigned)words[3 * g + 2]);
                const int w1 = pass == 0 && g == 1 ? words[3 * g + 1] ^ 0x00010000 : words[3 * g + 1];
                const __m128i v = _mm_set_epi32((int)tag, words[3 * g + 2], w1, words[3 * g]);
                _mm_store_si128((__m128i*)(slot + 4 * g), v);
            }
            _mm_sfence();
            if (pass == 0) {
                const auto t0 = std::chrono::steady_clock::now();
                while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() < 200.0) __builtin_ia32_pause();
            }
        }
    } else {
        PairState* rec = &c->h_xf[c->gated.slot];
        for (int k = 0; k < 12; ++k) rec->m[k] = st.m[k];
        rec->active = st.active; rec->apply = st.apply; rec->pad[0] = skip;
        __atomic_store_n(&rec->pad[1], c->gated.stamp, __ATOMIC_RELEASE);   // the kernel accepts the record when it reads this stamp
    }
    kss_ctx::Gated& G = c->gated;
    if (!skip && --G.steps_left > 0) {
        // a chained launch: its next pass waits for the next stamp in the other record, publishes under the next sequence
        // number and reads what this pass writes
        G.slot ^= 1;
        G.stamp += 1;                       // (the launch made sure the chain's stamps do not wrap)
        G.seq += 1;
        const void* in = G.d_out; G.d_out = const_cast<void*>(G.d_in); G.d_in = in;
    } else {
        G.pending = false;
        G.steps_left = 0;
    }
}

static void gated_cancel(kss_ctx* c) {
    if (!c->gated.pending) return;
    PairState none;
    std::memset(&none, 0, sizeof none);
    gated_release(c, none, 1);
}

struct GatedGuard {   // no exit from the ICP loop leaves a gate closed
    kss_ctx* c;
    explicit GatedGuard(kss_ctx* c_) : c(c_) {}
    ~GatedGuard() { gated_cancel(c); c->gated.want_next = false; }
};

// arguments of the fused cell-list pass for this plan (single pair or batch)
static PassArgs pass_args(kss_ctx* c, const IcpPlan& pl, const float4* d_in, float4* d_out, double max_d2,
                          int32_t* d_idx_out, float* d_d2_out) {
    PassArgs a;
    std::memset(&a, 0, sizeof a);
    a.src_in = d_in; a.src_out = d_out;
    a.cell_start = (const int32_t*)c->g_start.p + 1;
    a.sorted = (const float4*)c->g_sorted.p;
    a.tgt4 = (const float4*)c->tgt4.p;
    a.nn_win = (float4*)c->g_pos.p;
    a.nn_state = (float2*)c->g_nnst.p;
    static const float skin = getenv("KSS_SKIN") ? (float)atof(getenv("KSS_SKIN")) : 0.25f;   // KSS_SKIN=-1: every source searches in every pass
    a.skin = skin;
    a.chain_len = 1;
    static const int gate_polls = getenv("KSS_GATE_POLLS") ? atoi(getenv("KSS_GATE_POLLS")) : (1 << 22);   // bound of a waiting kernel's poll (~1 us each)
    a.gate_polls = gate_polls;
    a.chained = d_in != (const float4*)c->src0.p ? 1 : c->fit_last ? 2 : 0;   // (the first pass and the fitness pass read the original cloud)
    a.src_last0 = (const float4*)c->cur[0].p; a.src_last1 = (const float4*)c->cur[1].p;
    a.keys = (unsigned long long*)c->keys.p;
    a.list = (int32_t*)c->g_list.p; a.list_count = (int32_t*)c->g_count.p;
    a.total_rows = pl.total_rows;
    static const bool noprev = getenv("KSS_GRID_NOPREV") != nullptr;   // A/B switch: the previous winner is stored but not used as a bound
    a.use_prev = !noprev && c->nn_have ? 1 : 0;   // (nn_have: a search pass of this registration has written the per-source state)
    a.max_d2 = max_d2;
    a.rows = (double*)c->partials.p; a.tickets = (int32_t*)c->pair_ticket.p;
    a.pub = c->h_seq_dev;
    a.idx_out = d_idx_out; a.d2_out = d_d2_out;
    {   // test hook: the n-th fused launch stores one of its result slots torn first (kss_grid.hip)
        static const long torn_at = getenv("KSS_TEST_TORN_SLOT") ? atol(getenv("KSS_TEST_TORN_SLOT")) : -1;
        static long launches = 0;
        a.test_torn = torn_at > 0 && ++launches == torn_at ? 1 : 0;
    }
    if (pl.gridb) {
        a.pairs = (const GridPairDev*)c->g_pairs.p;
        a.row_pair = (const int32_t*)c->g_rowpair.p;
    } else {
        a.pair0 = pl.gpairs[0];
        // rows as tagged granules + a designated reducer instead of drain + ticket.  Only the reducer waits -- for rows of
        // workgroups that nothing keeps from starting (it holds one slot of the chip) -- so the form does not need the launch
        // to be resident at once; the CHAIN built on it does (below).  At 1954 rows it takes the store-acknowledge wait and the
        // ticket round trip (2 of 9 us) off every workgroup's life and most of the last workgroup's 1954-row total off the
        // tail.  KSS_TAGGED_ROWS=0: A/B switch; KSS_TAGGED_ROWS_MAX: rows up to which the form is used (default: any).
        static const bool tagged = getenv("KSS_TAGGED_ROWS") == nullptr || atoi(getenv("KSS_TAGGED_ROWS")) != 0;
        static const int tagged_max = getenv("KSS_TAGGED_ROWS_MAX") ? atoi(getenv("KSS_TAGGED_ROWS_MAX")) : (1 << 30);
        a.tagged_rows = tagged && pl.total_rows <= std::max(tagged_max, resident_rows_limit(c)) ? 1 : 0;
    }
    return a;
}

// One NN pass (search + correspondence sums) over every active pair.  h_sums receives npairs*NSUMS.
// full: all 20 sums (the fitness / PCR_QM pass, traced runs); otherwise slots 17 and 18 stay 0 (an ICP iteration never
// reads them: one f64 sqrt per source and two reduction columns less).  active: which pairs take part (null: all).
static int nn_pass(kss_ctx* c, const IcpPlan& pl, bool fma, const float4* d_in, float4* d_out, double max_d2,
            int32_t* d_idx_out, float* d_d2_out, bool full, const int* active = nullptr) {
    PairState* hs = (PairState*)c->h_state;
    if (pl.grid) {
        // single pair: the transform rides in the kernel arguments
        PassArgs a = pass_args(c, pl, d_in, d_out, max_d2, d_idx_out, d_d2_out);
        const int nblk = grid_pass_blocks(pl.total_rows);
        // diagnostic builds of the timeline (tools/grid_stamps.py).  KSS_GRID_STAMPS=1: plain launches, buffer cleared and read
        // back every pass.  =2: the gated chain as it runs in production; two buffers alternate with the launch sequence number
        // (the launch pre-enqueued behind the last real one is cancelled but has stamped its start) and kss_debug_grid_stamps
        // fetches the one of the last launch that was waited for.
        static const int stamps_mode = getenv("KSS_GRID_STAMPS") ? atoi(getenv("KSS_GRID_STAMPS")) : 0;
        unsigned long long* stamps2 = nullptr;
        if (stamps_mode == 1) {
            KCHK(ensure(c, c->g_stamps, (size_t)nblk * 16 * sizeof(unsigned long long)));
            HIPCHK(c, hipMemsetAsync(c->g_stamps.p, 0, (size_t)nblk * 16 * sizeof(unsigned long long), c->stream));
            a.stamps = (unsigned long long*)c->g_stamps.p;
        } else if (stamps_mode == 2) {
            KCHK(ensure(c, c->g_stamps, (size_t)nblk * 32 * sizeof(unsigned long long)));
            stamps2 = (unsigned long long*)c->g_stamps.p;
            c->stamps_nblk = nblk;
        }
        const auto tl0 = c->timing ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
        unsigned long long want_seq = 0;
        bool via_gate = false;
        kss_ctx::Gated& G = c->gated;
        const bool plain_args = !a.stamps && !d_idx_out && !d_d2_out;
        if (G.pending && G.d_in == (const void*)d_in && G.d_out == (void*)d_out && G.fma == fma && G.full == full && G.max_d2 == max_d2 && plain_args) {
            // this pass was enqueued while the previous one ran: hand it its transform and open the gate
            want_seq = G.seq;
            c->seq = std::max(c->seq, G.seq);   // (sequence numbers of a chain are taken as its passes are released)
            if (G.chain && c->prof > 0) c->prof_n[KSS_K_GRID_CHAIN_PASS] += 1;
            via_gate = true;
            // (test hook: the n-th answer is never written -- a host thread that stalls for longer than the kernel's bounded
            // poll.  The waiting kernel gives up at the gate, having touched nothing, and the pass is launched again below.)
            static const long drop_at = getenv("KSS_TEST_DROP_GATE") ? atol(getenv("KSS_TEST_DROP_GATE")) : -1;
            static long releases = 0;
            if (++releases == drop_at) { G.pending = false; G.steps_left = 0; }
            else gated_release(c, hs[0], 0);
        } else {
            gated_cancel(c);   // (a pre-enqueued kernel that does not fit this pass, e.g. before the fitness pass)
            ProfScope ps(c, KSS_K_GRID_NN);
            // search + correspondence sums + final reduction in ONE launch; sums land in host-mapped memory
            a.ps0 = hs[0];
            a.seq = ++c->seq;
            if (stamps2) a.stamps = stamps2 + (size_t)(a.seq & 1) * nblk * 16;
            launch_grid_pass(c->stream, fma, full, false, true, a);
            c->nn_have = true;
            want_seq = c->seq;
        }
        HIPCHK(c, hipGetLastError());
        // HIP-event timing: a bracket behind the gate would also time the dispatch that follows the wait, so the launches
        // the profiler samples (every n-th) are plain launches; the others advance its tick here
        const bool next_sampled = c->prof == 1 || (c->prof > 1 && c->prof_tick[KSS_K_GRID_NN] % (unsigned)c->prof == 0);
        // how many iterations a pre-enqueued launch may run: all that can still follow (KSS_CHAIN=0: one, the form of every
        // system without a large BAR).  Chains need rows handed over as tagged granules (no ticket to re-arm between
        // passes) or a single row.  A single gated launch stays out of the way of the launches the profiler samples; a
        // chain is bracketed as a whole instead.
        static const bool want_chain = getenv("KSS_CHAIN") == nullptr || atoi(getenv("KSS_CHAIN")) != 0;
        int len = 1;
        if (G.want_next && !G.pending && plain_args && gated_available(c) && want_chain && c->gate_bar &&
            ((a.tagged_rows && pl.total_rows <= resident_rows_limit(c)) || pl.total_rows == 1))   // (every workgroup of a chain waits at its gates: resident at once)
            len = std::max(1, std::min(G.max_steps, 1 << 16));
        if (G.want_next && !G.pending && plain_args && gated_available(c) && (len > 1 || !next_sampled)) {
            if (c->prof > 1 && len == 1) ++c->prof_tick[KSS_K_GRID_NN];
            // enqueue the NEXT iteration behind the gate while this one runs: it reads what this pass writes (d_out) and
            // writes the other ping-pong buffer
            float4* nxt_out = d_out == (float4*)c->cur[0].p ? (float4*)c->cur[1].p : (float4*)c->cur[0].p;
            G.slot ^= 1;
            G.stamp = (G.stamp + 1) & 0x7fffffff;
            if (G.stamp == 0 || G.stamp > 0x7fffffff - len - 1) G.stamp = 1;   // (0 is what a fresh record holds; a chain's stamps do not wrap)
            PassArgs n = pass_args(c, pl, d_out, nxt_out, max_d2, nullptr, nullptr);
            n.ps0 = hs[0];
            n.gate_seq = G.stamp;
            if (c->gate_bar) {   // the host writes the granules itself
                n.state = nullptr;
                n.gate_dev = c->gate_bar;    // two records, 128 bytes apart: pass k of the launch polls record (slot + k) & 1
                n.gate_slot = G.slot;
                n.chain_len = len;
                n.src_alt = d_out;
            } else {             // workgroup 0 asks the host-mapped record and re-publishes
                n.state = c->h_xf_dev + G.slot;
                n.gate_dev = (unsigned int*)c->g_gate.p;
            }
            n.seq = c->seq + 1;          // (taken when the pass is released)
            if (stamps2) n.stamps = stamps2 + (size_t)(n.seq & 1) * nblk * 16;
            if (len > 1) {   // bracketed as a whole whenever profiling is on (KSS_K_GRID_CHAIN; per pass: KSS_K_GRID_CHAIN_PASS)
                ProfScope ps(c, KSS_K_GRID_CHAIN, true);
                launch_grid_pass(c->stream, fma, G.want_full, false, true, n);
            } else {
                launch_grid_pass(c->stream, fma, G.want_full, false, true, n);
            }
            G.chain = len > 1;
            G.pending = true; G.steps_left = len; G.seq = n.seq; G.d_in = d_out; G.d_out = nxt_out; G.fma = fma; G.full = G.want_full; G.max_d2 = max_d2;
            if (hipGetLastError() != hipSuccess) { gated_cancel(c); G.supported = 0; }   // (if it did get queued it is answered; gating is given up)
        }
        const auto tl1 = c->timing ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
        {
            bool not_published = false;
            int rcw = wait_seq(c, 1, want_seq, nullptr, &not_published);
            if (rcw != KSS_OK && via_gate && not_published) {
                // The waiting kernel left at its gate (bounded poll: this thread was stalled, or the answer never arrived) and
                // the stream has drained.  Nothing of this pass has been written: launch it as a plain pass.
                G.pending = false; G.steps_left = 0;
                c->err.clear();
                std::fprintf(stderr, "[kss] a waiting kernel was not answered in time and left; the pass is launched again\n");
                // With more rows than resident workgroups the launch may have left in part only: workgroups that started after
                // the answer finally arrived have run, written their sources and drawn tickets that nobody completed.  The
                // stream has drained (wait_seq synchronised): re-arm everything that is zero at rest before the plain launch.
                c->ws_dirty = true;
                KCHK(restore_zero_at_rest(c));
                ProfScope ps(c, KSS_K_GRID_NN);
                a.ps0 = hs[0];
                a.seq = ++c->seq;
                launch_grid_pass(c->stream, fma, full, false, true, a);
                HIPCHK(c, hipGetLastError());
                want_seq = c->seq;
                rcw = wait_seq(c, 1, want_seq);
            }
            KCHK(rcw);
        }
        if (stamps2) c->stamps_seq = want_seq;
        if (c->timing) {
            const auto tl2 = std::chrono::steady_clock::now();
            c->t_launch_us += std::chrono::duration<double, std::micro>(tl1 - tl0).count();
            c->t_wait_us += std::chrono::duration<double, std::micro>(tl2 - tl1).count();
        }
        if (a.stamps && !stamps2) {
            c->last_stamps.resize((size_t)nblk * 16);
            HIPCHK(c, hipMemcpy(c->last_stamps.data(), a.stamps, c->last_stamps.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            for (int b = 0; b < nblk; ++b) c->evals_sum += (double)c->last_stamps[(size_t)b * 16 + 10];
            c->evals_launches += 1.0;
        }
        if (((const double*)c->h_sums)[NSUMS - 1] > 0.0) {
            // sources the cell search gave up on (far from the target): brute-force sweep over the list, then the sums
            // again by a SEARCH = false launch of the fused pass -- same rows, same order, so the result does not depend
            // on WHICH sources needed the list
            gated_cancel(c);              // the list pass must not queue up behind a closed gate
            c->gated.want_next = false;   // (clouds this far apart: plain launches until the loop says otherwise)
            KCHK(stage_tables(c, pl));
            {   // the list sweep reads the pair's device-side state (active, identity): uploaded here, on the rare path only
                PairState one;
                std::memset(&one, 0, sizeof one);
                one.active = 1;
                HIPCHK(c, hipMemcpy(c->state.p, &one, sizeof one, hipMemcpyHostToDevice));
            }
            if (getenv("KSS_LIST_CHECK")) {   // diagnostic: the list the search pass has left, checked on the host before it is used
                HIPCHK(c, hipStreamSynchronize(c->stream));
                int32_t cnt = -1;
                HIPCHK(c, hipMemcpy(&cnt, c->g_count.p, sizeof cnt, hipMemcpyDeviceToHost));
                const int64_t ns0 = pl.g[0].ns;
                std::vector<int32_t> l((size_t)std::max<int64_t>(0, std::min<int64_t>(cnt, ns0)));
                if (!l.empty()) HIPCHK(c, hipMemcpy(l.data(), c->g_list.p, l.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
                int64_t bad = 0;
                for (int32_t v : l) bad += v < 0 || v >= ns0;
                std::fprintf(stderr, "[kss] list check: count %d of %lld sources (column 19 said %.0f), %lld entries out of range, nn table %zu items, S %d\n",
                             cnt, (long long)ns0, ((const double*)c->h_sums)[NSUMS - 1], (long long)bad, pl.nn.size(), pl.S);
                if (cnt < 0 || cnt > ns0 || bad) return set_err(c, KSS_ERR_HIP, "list check failed");
            }
            {
                ProfScope ps(c, KSS_K_NN_SWEEP);
                launch_nn_sweep_list(c->stream, pl.S, fma, (const NNWork*)c->nn_work.p, (int)pl.nn.size(), (const PairState*)c->state.p,
                                     d_out, (const float4*)c->tgt4.p, (unsigned long long*)c->keys.p, (const int32_t*)c->g_list.p,
                                     (const int32_t*)c->g_count.p);
            }
            HIPCHK(c, hipMemsetAsync(c->g_count.p, 0, sizeof(int32_t), c->stream));   // the list is consumed
            {
                ProfScope ps(c, KSS_K_CORR_REDUCE);
                PassArgs r = pass_args(c, pl, d_in, d_out, max_d2, d_idx_out, d_d2_out);
                r.ps0 = hs[0];
                r.seq = ++c->seq;
                launch_grid_pass(c->stream, fma, true, false, false, r);
            }
            HIPCHK(c, hipGetLastError());
            KCHK(wait_seq(c, 1, c->seq));
        }
        ((double*)c->h_sums)[NSUMS - 1] = 0.0;
        return KSS_OK;
    }
    if (pl.gridb) {
        // batch: one launch searches every active pair and publishes each pair's sums as its last workgroup finishes.
        // The per-pair states (transform to apply, active flag) are read by every workgroup: a small batch reads them
        // straight from the pinned host-mapped table (no copy operation on the stream), a large one (thousands of
        // workgroups) gets the table copied to device memory once per pass.
        PassArgs a = pass_args(c, pl, d_in, d_out, max_d2, d_idx_out, d_d2_out);
        a.state = (const PairState*)c->state.p;
        if (c->bar_state && c->bar_state_cap >= pl.npairs) {   // the host has stored the states there itself (mirror_state)
            _mm_sfence();
            a.state = c->bar_state;
        }
        static const int host_state_rows = [] { const char* e = getenv("KSS_BATCH_HOST_STATE_ROWS"); return e ? atoi(e) : 32768; }();   // tuning hook (measured at C3, 20480 workgroups: 9.71 ms from host memory vs 9.85 ms with the copy)
        if (a.state != c->bar_state && pl.total_rows <= host_state_rows) {
            void* dev = nullptr;
            if (hipHostGetDevicePointer(&dev, c->h_state, 0) == hipSuccess && dev) a.state = (const PairState*)dev;
        }
        if (a.state == (const PairState*)c->state.p)
            HIPCHK(c, hipMemcpyAsync(c->state.p, hs, (size_t)pl.npairs * sizeof(PairState), hipMemcpyHostToDevice, c->stream));
        const int nblk = grid_pass_blocks(pl.total_rows);
        if (getenv("KSS_GRID_STAMPS")) {   // diagnostic build of the timeline (tools/batch_stamps.py)
            KCHK(ensure(c, c->g_stamps, (size_t)nblk * 16 * sizeof(unsigned long long)));
            HIPCHK(c, hipMemsetAsync(c->g_stamps.p, 0, (size_t)nblk * 16 * sizeof(unsigned long long), c->stream));
            a.stamps = (unsigned long long*)c->g_stamps.p;
        }
        {
            ProfScope ps(c, KSS_K_GRID_NN);
            a.seq = ++c->seq;
            launch_grid_pass(c->stream, fma, full, true, true, a);
            c->nn_have = true;
        }
        HIPCHK(c, hipGetLastError());
        if (a.stamps) {
            HIPCHK(c, hipStreamSynchronize(c->stream));   // (the stream does not synchronise with the null stream)
            c->last_stamps.resize((size_t)nblk * 16);
            HIPCHK(c, hipMemcpy(c->last_stamps.data(), a.stamps, c->last_stamps.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        }
        if (!c->defer_wait) KCHK(wait_seq(c, pl.npairs, c->seq, active));
        return KSS_OK;
    }
    // brute-force engine.  Per-pair states as above.
    const PairState* d_state = (const PairState*)c->state.p;
    if (pl.npairs <= PUB_PAIRS) {
        void* dev = nullptr;
        if (hipHostGetDevicePointer(&dev, c->h_state, 0) == hipSuccess && dev) d_state = (const PairState*)dev;
    }
    if (d_state == (const PairState*)c->state.p)
        HIPCHK(c, hipMemcpyAsync(c->state.p, hs, (size_t)pl.npairs * sizeof(PairState), hipMemcpyHostToDevice, c->stream));
    // The candidate batch of a registration (pairs sharing ONE small target, all of one size): sweep + sums + publication in a
    // single launch per pass (kss_kernels.hip: cand_pass_kernel).  KSS_CAND_FUSED=0: sweep + reduce as before (A/B; the two
    // forms agree on every correspondence, their sums differ in the order of additions).
    static const bool cand_fused = getenv("KSS_CAND_FUSED") == nullptr || atoi(getenv("KSS_CAND_FUSED")) != 0;
    if (cand_fused && pl.shared_target && pl.npairs > 1 && !pl.src_in_cell_order && pl.g[0].tgt_pad <= 8192) {
        bool same = true;
        for (int p = 1; p < pl.npairs; ++p) same = same && pl.g[p].ns == pl.g[0].ns && pl.g[p].src_base == (int64_t)p * pl.g[0].ns;
        const int bpp = cand_pass_blocks_per_pair(pl.g[0].ns);
        if (same && ensure_zeroed(c, c->partials, (size_t)pl.npairs * bpp * NSUMS * sizeof(double)) == KSS_OK &&
            ensure_zeroed(c, c->cand_tags, (size_t)pl.npairs * bpp * sizeof(unsigned)) == KSS_OK) {
            bool launched;
            {
                ProfScope ps(c, KSS_K_NN_SWEEP);
                launched = launch_cand_pass(c->stream, fma, pl.npairs, d_state, d_in, d_out, (const float4*)c->tgt4.p + pl.g[0].tgt_base, pl.g[0].tgt_pad,
                                            (int)pl.g[0].ns, max_d2, (double*)c->partials.p, (unsigned*)c->cand_tags.p, c->h_seq_dev, c->seq + 1, d_idx_out, d_d2_out);
            }
            if (launched) {
                ++c->seq;
                HIPCHK(c, hipGetLastError());
                KCHK(wait_seq(c, pl.npairs, c->seq, active));
                return KSS_OK;
            }
        }
    }
    KCHK(stage_tables(c, pl));
    {
        ProfScope ps(c, KSS_K_NN_SWEEP);
        launch_nn_sweep(c->stream, pl.S, fma, (const NNWork*)c->nn_work.p, (int)pl.nn.size(), d_state,
                        d_in, d_out, (const float4*)c->tgt4.p, (unsigned long long*)c->keys.p);
    }
    const bool spin = pl.npairs <= PUB_PAIRS;
    {
        ProfScope ps(c, KSS_K_CORR_REDUCE);
        if (spin) {   // small batch: the reduce also adds each pair's rows up and publishes (pair tickets zeroed by stage_plan)
            launch_corr_reduce_publish(c->stream, (const RedWork*)c->red_work.p, (int)pl.red.size(), d_state,
                                       d_out, (const float4*)c->tgt4.p, (const unsigned long long*)c->keys.p, max_d2,
                                       (double*)c->partials.p, d_idx_out, d_d2_out, pl.src_in_cell_order ? 1 : 0,
                                       (const PairRed*)c->pair_red.p, (int32_t*)c->pair_ticket.p, c->h_seq_dev, ++c->seq);
        } else {
            launch_corr_reduce(c->stream, (const RedWork*)c->red_work.p, (int)pl.red.size(), d_state,
                               d_out, (const float4*)c->tgt4.p, (const unsigned long long*)c->keys.p, max_d2,
                               (double*)c->partials.p, d_idx_out, d_d2_out, pl.src_in_cell_order ? 1 : 0);
            launch_finalize_sums(c->stream, (const PairRed*)c->pair_red.p, pl.npairs, (const double*)c->partials.p, (double*)c->h_sums_dev);
        }
    }
    HIPCHK(c, hipGetLastError());
    if (spin) KCHK(wait_seq(c, pl.npairs));
    else HIPCHK(c, hipStreamSynchronize(c->stream));
    return KSS_OK;
}

// the batched pass's per-pair state table in fine-grained device memory the host can store into (null: no large BAR)
static PairState* bar_state_table(kss_ctx* c, int npairs) {
    static const bool want_bar = getenv("KSS_GATE_BAR") == nullptr || atoi(getenv("KSS_GATE_BAR")) != 0;
    if (!want_bar || c->bar_state_failed) return nullptr;
    if (c->bar_state && c->bar_state_cap >= npairs) return c->bar_state;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, c->device) != hipSuccess || !prop.isLargeBar) { c->bar_state_failed = true; return nullptr; }
    if (c->bar_state) { hipStreamSynchronize(c->stream); hipFree(c->bar_state); c->bar_state = nullptr; c->bar_state_cap = 0; }
    void* p = nullptr;
    const int cap = npairs + npairs / 4 + 64;
    if (hipExtMallocWithFlags(&p, (size_t)cap * sizeof(PairState), hipDeviceMallocFinegrained) != hipSuccess || !p) {
        (void)hipGetLastError();
        c->bar_state_failed = true;
        return nullptr;
    }
    c->bar_state = (PairState*)p; c->bar_state_cap = cap;
    return c->bar_state;
}

static void set_state(PairState& s, const float T[16], int active, int apply) {
    for (int k = 0; k < 12; ++k) s.m[k] = T[k];
    s.active = active; s.apply = apply; s.pad[0] = s.pad[1] = 0;
}
// ... and its mirror in fine-grained device memory (large-BAR systems, batched cell lists): the workgroups of the next pass
// read their pair's state from device memory, the host having stored it there directly -- no copy operation, no PCIe read
// per workgroup.  Write-only on the host side (a host READ through the BAR costs microseconds).
static inline void mirror_state(PairState* bar, int p, const PairState& s) {
    if (!bar) return;
    const __m128i* v = reinterpret_cast<const __m128i*>(&s);
    __m128i* d = reinterpret_cast<__m128i*>(bar + p);
    for (int k = 0; k < 4; ++k) _mm_store_si128(d + k, _mm_loadu_si128(v + k));
}

// ---- the pair-resident engine (kss_resident.hip), host side ------------------------------------------------------------
// One launch runs every registration of the batch, one workgroup per pair; the host's part of an ICP iteration -- 20 sums in,
// 3x3 SVD, convergence tests, transform out -- is served per PAIR as its sums land, by a few threads that each own an
// interleaved share of the pairs (workgroups start roughly in pair order, so the pairs in flight spread over all threads).
// Every pair is solved by exactly one thread with the code of the launch-per-pass loop: same results.
// Gate records live in fine-grained device memory the host stores into through the BAR (large-BAR systems only; otherwise
// the launch-per-pass engine runs).  KSS_RESIDENT=0 switches the engine off (A/B; the engines agree bit for bit).
static bool resident_capacities(const IcpPlan& pl, int* ntc_out, int* tabc_out) {
    int64_t max_nt = 0;
    for (int p = 0; p < pl.npairs; ++p) {
        if (pl.g[p].ns > (int64_t)RES_SMAX * RES_THREADS || pl.g[p].nt > 65534) return false;   // (positions are 16 bits, 0xffff: none)
        max_nt = std::max(max_nt, pl.g[p].nt);
    }
    const int ntc = (int)((max_nt + 3 + 63) / 64 * 64);   // (+3: an evaluation step reads four points)
    const int64_t avail = ((int64_t)RES_LDS_MAX - (int64_t)resident_lds_bytes(ntc, 0)) / 2 / 8 * 8;   // table entries that still fit
    if (avail < 16) return false;
    int64_t want = 16;
    for (int p = 0; p < pl.npairs; ++p) {
        const GridParams& gp = pl.gpairs[p].gp;
        const int64_t rows = (int64_t)gp.gy * gp.gz;
        int xs = 0;
        while (xs <= 8 && rows * (((gp.gx + (1 << xs) - 1) >> xs) + 1) > avail) ++xs;   // x resolution of the table halves until it fits
        if (xs > 8) return false;
        want = std::max(want, rows * (((gp.gx + (1 << xs) - 1) >> xs) + 1));
    }
    *ntc_out = ntc;
    *tabc_out = (int)((want + 7) / 8 * 8);
    return true;
}

static unsigned int* resident_gate(kss_ctx* c, int npairs) {
    static const bool want_bar = getenv("KSS_GATE_BAR") == nullptr || atoi(getenv("KSS_GATE_BAR")) != 0;
    if (!want_bar || c->res_gate_failed) return nullptr;
    if (c->res_gate && c->res_gate_cap >= npairs) return c->res_gate;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, c->device) != hipSuccess || !prop.isLargeBar) { c->res_gate_failed = true; return nullptr; }
    if (c->res_gate) { hipStreamSynchronize(c->stream); hipFree(c->res_gate); c->res_gate = nullptr; c->res_gate_cap = 0; }
    void* p = nullptr;
    const int cap = npairs + npairs / 4 + 64;
    if (hipExtMallocWithFlags(&p, (size_t)cap * 128, hipDeviceMallocFinegrained) != hipSuccess || !p || hipMemset(p, 0, (size_t)cap * 128) != hipSuccess ||
        hipDeviceSynchronize() != hipSuccess) {
        (void)hipGetLastError();
        if (p) hipFree(p);
        c->res_gate_failed = true;
        return nullptr;
    }
    c->res_gate = (unsigned int*)p; c->res_gate_cap = cap;
    return c->res_gate;
}

static bool resident_gate_available(kss_ctx* c, int npairs) { return resident_gate(c, npairs) != nullptr; }

// the 20 sums of pair p under sequence number `want`, if they have all landed
static inline bool resident_collect(const unsigned long long* h_seq, int p, unsigned long long want, double* out) {
    const unsigned long long* sl = h_seq + (size_t)2 * NSUMS * p;
    for (int k = NSUMS - 1; k >= 0; --k) {
        unsigned long long bits;
        if (!slot_value(sl, k, want, &bits)) return false;
        std::memcpy(&out[k], &bits, sizeof(double));
    }
    return true;
}

// workgroups of candidate-resident launches in flight, per device, over all contexts of the process
static std::atomic<int> g_cand_in_flight[64];
struct CandReservation {
    int dev = -1, n = 0;
    bool take(int device, int want, int cap) {
        if (device < 0 || device >= 64 || n != 0) return false;
        const int before = g_cand_in_flight[device].fetch_add(want);
        if (before + want > cap) { g_cand_in_flight[device].fetch_sub(want); return false; }
        dev = device; n = want;
        return true;
    }
    ~CandReservation() { if (n > 0) g_cand_in_flight[dev].fetch_sub(n); }
};

// *handled = false with KSS_OK: the plan does not qualify (or the engine gave up before touching any result): the caller
// runs the launch-per-pass loop
static int resident_loop(kss_ctx* c, const IcpPlan& pl, const kss_icp_params& P, kss_icp_result* results, bool* handled, bool cand = false) {
    *handled = false;
    static const bool want = getenv("KSS_RESIDENT") == nullptr || atoi(getenv("KSS_RESIDENT")) != 0;
    static const bool want_cand = getenv("KSS_CAND_RESIDENT") == nullptr || atoi(getenv("KSS_CAND_RESIDENT")) != 0;
    if (!(cand ? want_cand : want) || P.allreduce || P.max_iterations < 1 || P.max_iterations > 4000) return KSS_OK;
    if (cand ? (pl.grid || pl.gridb || !pl.shared_target || pl.src_in_cell_order || pl.g[0].tgt_pad > 8192) : !pl.gridb) return KSS_OK;
    if (std::chrono::steady_clock::now() < c->res_backoff[cand ? 1 : 0]) return KSS_OK;   // given up recently: see below
    const int np = pl.npairs;
    // what the serving loop needs to know about the launch, whichever kernel it is
    struct { unsigned long long seq0 = 0; unsigned stamp0 = 0; int32_t* idx_out = nullptr; float* d2_out = nullptr; unsigned long long* stamps = nullptr; } a;
    static const int gate_polls = getenv("KSS_GATE_POLLS") ? atoi(getenv("KSS_GATE_POLLS")) : (1 << 22);
    static const bool stamps_on = getenv("KSS_GRID_STAMPS") != nullptr;   // diagnostic timeline (tools/resident_stamps.py)
    unsigned int* gate = nullptr;
    const int max_passes = P.max_iterations + 2;
    CandReservation cand_reserve;   // (given back when this function returns: the kernel has drained by then on every path)
    auto number_launch = [&]() {
        a.seq0 = c->seq + 1;
        c->seq += (unsigned long long)max_passes + 1;
        c->res_launches = c->res_launches % 500000u + 1u;
        a.stamp0 = c->res_launches * 4096u;          // every launch has its own 4096 stamps: a record of an earlier launch never matches
    };
    auto want_correspondences = [&]() -> int {
        if (P.compute_fitness && (P.fitness_idx || P.fitness_d2)) {
            KCHK(ensure(c, c->stage_idx, (size_t)pl.total_src * sizeof(int32_t)));
            KCHK(ensure(c, c->stage_d2, (size_t)pl.total_src * sizeof(float)));
            a.idx_out = (int32_t*)c->stage_idx.p; a.d2_out = (float*)c->stage_d2.p;
        }
        return KSS_OK;
    };
    // A batch of more pairs than the device runs at once ends with a tail: 1024 uneven registrations on 256 compute units take
    // 6.5 ms for 4.8 ms of work per unit (a tenth of C3's pairs keep hundreds of searches per pass going and run twice as long;
    // whichever of them starts last is the kernel's end).  Such a batch runs as TWO launches: every pair's passes 0 .. split_at - 1
    // (default: passes 0 and 1 -- the first, where every source searches, is the same work for every pair), then the rest with
    // the pairs in the order of DECREASING cost as the first launch predicts it -- longest first, the classic rule for this
    // scheduling problem.  The predictor: the searches the pair asked for in pass 1 (its exit flag carries the count): a pair
    // that is still moving then keeps moving, and searching, for the passes to come.  (With pass 0 alone in the first launch the
    // host orders by the mean squared distance of the first correspondences instead: on C3 as good -- 6.53 vs 6.59 ms -- but by
    // accident: that distance saturates at the point spacing, the long pairs are those rotated by 2-4 degrees, one to two
    // spacings, which creep for all twenty passes, and they merely end up in the middle of that order.  The size of the first
    // ICP step predicts nothing: 7.4.)  Measured at C3 (ms per batch): one launch 7.39-7.45; split after pass 0: 6.51-6.64,
    // after pass 1: 6.59-6.66, 2: 6.83, 3: 7.05, 5: 7.15, 7: 7.63.  Between the launches a pair's registers rest in memory (20 bytes per source).
    // Same passes, same host protocol, same bits.  KSS_RESIDENT_SPLIT=0: one launch (A/B); =k: split after pass k - 1.
    static const int split_env = getenv("KSS_RESIDENT_SPLIT") ? atoi(getenv("KSS_RESIDENT_SPLIT")) : 2;
    int split_at = 0;
    ResArgs ra;
    std::memset(&ra, 0, sizeof ra);
    if (!cand) {
        int ntc = 0, tabc = 0;
        if (!resident_capacities(pl, &ntc, &tabc)) return KSS_OK;
        gate = resident_gate(c, np);
        if (!gate) return KSS_OK;
        if (split_env > 0 && np >= 2 * resident_rows_limit(c) && P.max_iterations >= split_env + 4 &&
            ensure(c, c->res_pos, (size_t)pl.total_src * sizeof(float4)) == KSS_OK && ensure(c, c->res_wc, (size_t)pl.total_src * sizeof(unsigned)) == KSS_OK &&
            ensure(c, c->res_perm, (size_t)np * sizeof(int32_t)) == KSS_OK)
            split_at = split_env;
        ra.pairs = (const GridPairDev*)c->g_pairs.p;
        ra.cell_start = (const int32_t*)c->g_start.p + 1;
        ra.sorted = (const float4*)c->g_sorted.p;
        ra.src0 = (const float4*)c->src0.p;
        ra.gate = gate;
        ra.pub = c->h_seq_dev;
        ra.exit_flags = c->h_seq_dev + (size_t)2 * NSUMS * np;
        ra.max_passes = max_passes;
        number_launch();
        ra.seq0 = a.seq0; ra.stamp0 = a.stamp0;
        ra.exit_tag = a.stamp0;
        ra.split_at = split_at;
        ra.st_pos = (float4*)c->res_pos.p; ra.st_wc = (unsigned*)c->res_wc.p;
        ra.max_d2 = P.max_corr_dist * P.max_corr_dist;
        static const float skin = getenv("KSS_SKIN") ? (float)atof(getenv("KSS_SKIN")) : 0.25f;
        ra.skin = skin;
        ra.gate_polls = gate_polls;
        ra.full_always = P.trace_sums != nullptr ? 1 : 0;
        ra.ntc = ntc; ra.tabc = tabc;
        KCHK(want_correspondences());
        ra.idx_out = a.idx_out; ra.d2_out = a.d2_out;
        if (stamps_on) {
            KCHK(ensure(c, c->g_stamps, (size_t)np * 16 * sizeof(unsigned long long)));
            HIPCHK(c, hipMemsetAsync(c->g_stamps.p, 0, (size_t)np * 16 * sizeof(unsigned long long), c->stream));
            ra.stamps = a.stamps = (unsigned long long*)c->g_stamps.p;
        }
        ProfScope ps(c, KSS_K_RESIDENT, true);
        std::string lerr;
        const int rc = launch_resident(c->stream, P.nn_fma != 0, np, ra, lerr);
        if (rc != KSS_OK) { (void)hipGetLastError(); return KSS_OK; }   // (not launched: nothing touched, the other engine runs)
    } else {
        // candidates of one registration: equal sizes, packed one after the other, original order (what cand_pass_kernel asks
        // for), and the whole launch resident at once -- up to CAND_TPW tiles of 32 sources per workgroup
        for (int p = 0; p < np; ++p)
            if (pl.g[p].ns != pl.g[0].ns || pl.g[p].src_base != (int64_t)p * pl.g[0].ns) return KSS_OK;
        const int nt_pad = (int)pl.g[0].tgt_pad;
        const bool fma = P.nn_fma != 0;
        if (c->cand_cap_pad != nt_pad || c->cand_cap_fma != (fma ? 1 : 0)) {
            c->cand_cap = cand_resident_capacity(fma, nt_pad);
            c->cand_cap_pad = nt_pad; c->cand_cap_fma = fma ? 1 : 0;
        }
        // Every workgroup of the launch has to be on the chip at once, and so do those of the launches other contexts of this
        // process (the workers of kss_register_batch) have in flight on the same device: each launch RESERVES its workgroups
        // out of the device's capacity, takes more tiles per workgroup when little is left, and goes to the launch-per-pass
        // form when that is not enough.  (KSS_CAND_CAP: a smaller capacity, to exercise exactly that.)
        static const int cap_env = getenv("KSS_CAND_CAP") ? atoi(getenv("KSS_CAND_CAP")) : 0;
        const int cap = cap_env > 0 ? std::min(c->cand_cap, cap_env) : c->cand_cap;
        const int bpp = cand_pass_blocks_per_pair(pl.g[0].ns);
        if (cap <= 0 || bpp <= 0) return KSS_OK;
        int tpw = 0;
        for (int t = 1; t <= CAND_TPW && !tpw; ++t) {
            const int want_wg = np * ((bpp + t - 1) / t);
            if (want_wg > cap) continue;
            if (cand_reserve.take(c->device, want_wg, cap)) tpw = t;
        }
        if (!tpw) return KSS_OK;
        gate = resident_gate(c, np);
        if (!gate) return KSS_OK;
        if (ensure_zeroed(c, c->partials, (size_t)np * bpp * NSUMS * sizeof(double)) != KSS_OK) return KSS_OK;
        CandArgs ca;
        std::memset(&ca, 0, sizeof ca);
        ca.src0 = (const float4*)c->src0.p;
        ca.tgt = (const float4*)c->tgt4.p + pl.g[0].tgt_base;
        ca.nt_pad = nt_pad; ca.ns = (int)pl.g[0].ns;
        ca.bpp = bpp; ca.tpw = tpw; ca.wpp = (bpp + tpw - 1) / tpw;
        ca.max_d2 = P.max_corr_dist * P.max_corr_dist;
        ca.partials = (double*)c->partials.p;
        KCHK(ensure_zeroed(c, c->cand_tags, (size_t)np * bpp * sizeof(unsigned)));   // (zeroed when (re)allocated: no tag of the block's earlier owner; pass numbers start at 1)
        ca.row_tag = (unsigned*)c->cand_tags.p;
        ca.gate = gate;
        ca.pub = c->h_seq_dev;
        ca.exit_flags = c->h_seq_dev + (size_t)2 * NSUMS * np;
        number_launch();
        ca.seq0 = a.seq0; ca.stamp0 = a.stamp0;
        ca.gate_polls = gate_polls; ca.max_passes = max_passes;
        KCHK(want_correspondences());
        ca.idx_out = a.idx_out; ca.d2_out = a.d2_out;
        if (stamps_on) {   // (tools/cand_stamps.py)
            KCHK(ensure(c, c->g_stamps, (size_t)np * 16 * sizeof(unsigned long long)));
            HIPCHK(c, hipMemsetAsync(c->g_stamps.p, 0, (size_t)np * 16 * sizeof(unsigned long long), c->stream));
            ca.stamps = a.stamps = (unsigned long long*)c->g_stamps.p;
        }
        ProfScope ps(c, KSS_K_RESIDENT, true);
        std::string lerr;
        const int rc = launch_cand_resident(c->stream, fma, np, ca, lerr);
        if (rc != KSS_OK) { (void)hipGetLastError(); return KSS_OK; }
    }

    // ---- per-pair host state (what icp_loop keeps in its vectors) ----
    enum { PH_ITER = 0, PH_FIT = 1, PH_DONE = 2 };
    struct PairHost { Convergence cv; float fin[16]; int iters = 0, converged = 0, state = 0, k = 0, phase = PH_ITER, cancelled = 0, parked = 0; double last_mse = 0.0, fitness = 0.0, step0 = 0.0; };
    std::vector<PairHost> H((size_t)np);
    for (int p = 0; p < np; ++p) {
        Convergence& cv = H[p].cv;
        cv.max_iterations = P.max_iterations;
        cv.rotation_threshold = 1.0 - P.transformation_epsilon;
        cv.translation_threshold = P.transformation_epsilon;
        cv.mse_rel = P.euclidean_fitness_epsilon;
        cv.mse_abs = P.abs_mse_epsilon;
        cv.fixed_iterations = P.fixed_iterations != 0;
        mat4_identity(H[p].fin);
    }
    if (P.trace_n) *P.trace_n = 0;
    std::atomic<int> failed{0}, kernel_done{0}, pairs_left{np}, cancel_all{0};
    int split_now = split_at;                  // > 0 while the first launch of a split batch is being served
    unsigned exit_tag_now = a.stamp0;
    std::vector<char> in_launch((size_t)np, 1);
    const int judge = cand ? c->spec_judge : -1;
    const double judge_threshold = c->spec_threshold;
    std::atomic<long long> units{0};
    std::atomic<int> query_said{0};
    const unsigned long long* h_seq = c->h_seq;
    // test hooks: the n-th record first arrives with a granule whose words do not fit its check word (what a torn 16-byte store
    // would look like), the right one 200 us later; the n-th answer is held back for 300 ms (a stalled host thread: with a small
    // KSS_GATE_POLLS the workgroup leaves, and the call starts over on the other engine)
    static const long torn_at = getenv("KSS_TEST_TORN_RES_GATE") ? atol(getenv("KSS_TEST_TORN_RES_GATE")) : -1;
    static const long stall_at = getenv("KSS_TEST_RES_STALL") ? atol(getenv("KSS_TEST_RES_STALL")) : -1;
    std::atomic<long> sent{0};
    auto send = [&](int p, const float* T, int apply, int mode, bool any_pass = false) {   // the gate record of pair p's NEXT pass (number H[p].k + 1)
        unsigned w[15];
        if (T) std::memcpy(w, T, 12 * sizeof(float)); else std::memset(w, 0, 12 * sizeof(float));
        w[12] = 1u; w[13] = (unsigned)apply; w[14] = (unsigned)mode;
        // (any_pass: an order to stop that may land while some workgroups still wait for an EARLIER record -- one record per pair)
        const unsigned stamp = a.stamp0 + (any_pass ? RES_STAMP_ANY : (unsigned)(H[p].k + 1));
        unsigned int* slot = gate + (size_t)p * 32;
        const long nth = sent.fetch_add(1) + 1;
        if (nth == stall_at) std::this_thread::sleep_for(std::chrono::milliseconds(300));
        for (int pass = nth == torn_at ? 0 : 1; pass < 2; ++pass) {
            for (int g = 0; g < 5; ++g) {
                const unsigned tag = stamp + kss_mix3(w[3 * g], w[3 * g + 1], w[3 * g + 2]);   // stamp + check of the granule's three words
                const unsigned w1 = pass == 0 && g == 2 ? w[3 * g + 1] ^ 0x00010000u : w[3 * g + 1];
                _mm_store_si128((__m128i*)(slot + 4 * g), _mm_set_epi32((int)tag, (int)w[3 * g + 2], (int)w1, (int)w[3 * g]));
            }
            _mm_sfence();
            if (pass == 0) std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
    };
    auto serve = [&](int t, int nt) {
        int remaining = 0;
        for (int p = t; p < np; p += nt)
            if (H[p].phase != PH_DONE && !H[p].parked) ++remaining;
        long idle = 0, grace = 0;
        double s[NSUMS];
        auto last_progress = std::chrono::steady_clock::now();
        long long my_units = 0;
        // (the calling thread stays until EVERY pair is done: it is the one that asks HIP whether the kernel is still there)
        while ((remaining > 0 || (t == 0 && pairs_left.load(std::memory_order_relaxed) > 0)) && !failed.load(std::memory_order_relaxed)) {
            bool progress = false;
            const bool cancelling = judge >= 0 && cancel_all.load(std::memory_order_relaxed) != 0;
            for (int p = t; p < np; p += nt) {
                PairHost& h = H[p];
                if (h.phase == PH_DONE || h.parked) continue;
                if (cancelling && p != judge) {
                    // the judge was good enough: this candidate's result will not be looked at.  Its workgroups are told to stop at
                    // whatever gate they reach next (some may be inside a pass the others will never join: rows carry the number of
                    // their launch and pass, nothing is left to clear); a candidate already on its last pass leaves by itself.
                    if (h.phase == PH_ITER) send(p, nullptr, 0, 2, true);
                    h.phase = PH_DONE; h.cancelled = 1; --remaining; pairs_left.fetch_sub(1, std::memory_order_relaxed);
                    progress = true;
                    continue;
                }
                if (!resident_collect(h_seq, p, a.seq0 + (unsigned long long)h.k, s)) continue;
                progress = true;
                ++my_units;
                if (h.phase == PH_FIT) {   // getFitnessScore(): mean d2 over ALL source points
                    h.fitness = s[17] / (double)pl.g[p].ns;
                    h.phase = PH_DONE; --remaining; pairs_left.fetch_sub(1, std::memory_order_relaxed);
                    if (p == judge && h.fitness <= judge_threshold) cancel_all.store(1, std::memory_order_relaxed);
                    continue;
                }
                bool finished = false;
                float tk[16];
                if ((int)s[0] < P.min_correspondences) {   // PCL: "Not enough correspondences found"
                    h.state = KSS_STATE_NO_CORRESPONDENCES; h.converged = 0; finished = true;
                } else {
                    rigid_from_sums(s, tk);
                    mat4_mul(tk, h.fin, h.fin);            // final = transformation_ * final
                    ++h.iters;
                    const double mse = s[16] / s[0];
                    h.last_mse = mse;
                    if (h.iters == 1) h.step0 = (3.0 - ((double)tk[0] + (double)tk[5] + (double)tk[10])) + std::sqrt((double)tk[3] * tk[3] + (double)tk[7] * tk[7] + (double)tk[11] * tk[11]);   // how far the first step went: 2 (1 - cos angle) + |t|
                    if (p == 0 && P.trace_n && *P.trace_n < P.trace_cap) {
                        if (P.trace_sums) std::memcpy(P.trace_sums + (size_t)(*P.trace_n) * NSUMS, s, NSUMS * sizeof(double));
                        if (P.trace_Tk) std::memcpy(P.trace_Tk + (size_t)(*P.trace_n) * 16, tk, 16 * sizeof(float));
                        ++*P.trace_n;
                    }
                    const bool done = h.cv.has_converged(h.iters, tk, mse);
                    h.state = h.cv.state;
                    if (done) { h.converged = 1; finished = true; }
                }
                if (!finished) {
                    send(p, tk, 1, 0);                     // the next pass applies T_k on load (transformCloud)
                    ++h.k;
                } else if (P.compute_fitness) {
                    send(p, h.fin, 1, 1);                  // final * original input, all 20 sums
                    ++h.k;
                    h.phase = PH_FIT;
                } else {
                    send(p, nullptr, 0, 2);
                    h.phase = PH_DONE; --remaining; pairs_left.fetch_sub(1, std::memory_order_relaxed);
                }
                // the first launch of a split batch: the pair's workgroup left after pass split_now - 1; the record just written
                // is what the pair's workgroup of the SECOND launch finds at its first gate
                if (split_now > 0 && h.phase != PH_DONE && h.k == split_now) {
                    h.parked = 1; --remaining; pairs_left.fetch_sub(1, std::memory_order_relaxed);
                }
            }
            if (progress) { idle = 0; grace = 0; if (t == 0) last_progress = std::chrono::steady_clock::now(); continue; }
            __builtin_ia32_pause();
            if (idle > 16384 && (idle & 1023) == 1023) std::this_thread::yield();   // (nothing for ~a millisecond and maybe more spinning threads than cores: let the others run; sooner, the call costs a large batch's answers their latency: C3 7.5 -> 8.2 ms)
            if (++idle % 4096 == 0) {
                // nothing for a while: has the kernel gone?  (Only the calling thread talks to HIP.)  A workgroup that was not
                // answered within its bounded poll has left; its pair will never publish.  After the kernel has ended every
                // result is in host memory already: a thread that still finds nothing for ~10 ms of polling gives up.
                if (t == 0 && !kernel_done.load()) {
                    // every pair's workgroups have stored their exit flag (no HIP call here: one can wait on locks that other
                    // threads of the process hold for as long as THEIR kernels run -- and this thread has pairs to answer);
                    // after two seconds of nothing, HIP is asked all the same: a faulted kernel sets no flags
                    bool gone = true;
                    const unsigned long long* fl = h_seq + (size_t)2 * NSUMS * np;
                    for (int p = 0; p < np && gone; ++p) {
                        if (!in_launch[p]) continue;
                        const unsigned long long w0 = __atomic_load_n(&fl[2 * p], __ATOMIC_ACQUIRE), w1 = __atomic_load_n(&fl[2 * p + 1], __ATOMIC_RELAXED);
                        gone = (unsigned)w0 == exit_tag_now && (unsigned)(w0 >> 32) == (unsigned)p && (unsigned)(w1 >> 32) == kss_mix3((unsigned)w0, (unsigned)(w0 >> 32), (unsigned)w1);
                    }
                    if (gone) kernel_done.store(1);
                    else if (std::chrono::duration<double>(std::chrono::steady_clock::now() - last_progress).count() > 2.0) {
                        const hipError_t q = hipStreamQuery(c->stream);
                        if (q != hipErrorNotReady) { query_said.store((int)q); kernel_done.store(1); }
                        last_progress = std::chrono::steady_clock::now();
                    }
                }
                if (kernel_done.load() && ++grace > 64) failed.store(1);
            }
        }
        units.fetch_add(my_units);
    };
    static const int res_threads = [] {
        int v = (int)std::thread::hardware_concurrency();
        v = std::max(1, std::min(v, 12));
        if (const char* e = getenv("KSS_HOST_THREADS")) { const int u = atoi(e); if (u >= 1 && u <= 64) v = u; }
        return v;
    }();
    // (eight pairs a thread: one, two, four and eight candidates per thread were measured alike on a registration's 15 -- an
    // answer is a microsecond of host work -- and several registrations at once, kss_register_batch, must not bring more
    // spinning threads than the machine has cores)
    static const int cand_div = getenv("KSS_CAND_PAIRS_PER_THREAD") ? std::max(1, atoi(getenv("KSS_CAND_PAIRS_PER_THREAD"))) : 8;
    const int nthreads = std::max(1, std::min(res_threads, cand ? (np + cand_div - 1) / cand_div : (np + 7) / 8));
    c->pool.run_threads(nthreads, serve);
    if (split_at > 0 && !failed.load()) {
        // ---- the second launch of a split batch: the parked pairs, longest first ----
        HIPCHK(c, hipStreamSynchronize(c->stream));   // every workgroup of the first launch has left (its registers are in memory)
        // cost predictor: the searches the pair asked for in passes 1 .. split_at - 1 (its exit flag carries the count); with
        // only pass 0 in the first launch, the mean squared distance of its first correspondences (KSS_RESIDENT_SPLIT_KEY=mse)
        static const bool key_mse = getenv("KSS_RESIDENT_SPLIT_KEY") != nullptr && std::string(getenv("KSS_RESIDENT_SPLIT_KEY")) == "mse";
        std::vector<std::pair<double, int>> order;
        const unsigned long long* fl = h_seq + (size_t)2 * NSUMS * np;
        for (int p = 0; p < np; ++p) {
            in_launch[p] = 0;
            if (!H[p].parked) continue;
            static const bool key_step = getenv("KSS_RESIDENT_SPLIT_KEY") != nullptr && std::string(getenv("KSS_RESIDENT_SPLIT_KEY")) == "step";
            order.emplace_back(key_step ? H[p].step0 : key_mse || split_at < 2 ? H[p].last_mse : (double)(unsigned)(fl[2 * p + 1] & 0xffffffffull), p);
        }
        std::stable_sort(order.begin(), order.end(), [](const std::pair<double, int>& x, const std::pair<double, int>& y) { return x.first > y.first; });
        const int np2 = (int)order.size();
        if (np2 > 0) {
            std::vector<int32_t> perm((size_t)np2);
            for (int i = 0; i < np2; ++i) { perm[i] = order[i].second; in_launch[order[i].second] = 1; H[order[i].second].parked = 0; }
            HIPCHK(c, hipMemcpyAsync(c->res_perm.p, perm.data(), (size_t)np2 * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));   // (pageable source)
            ra.perm = (const int32_t*)c->res_perm.p;
            ra.first_pass = split_at; ra.split_at = 0;
            ra.exit_tag = a.stamp0 + 1u;
            if (stamps_on) HIPCHK(c, hipMemsetAsync(c->g_stamps.p, 0, (size_t)np * 16 * sizeof(unsigned long long), c->stream));   // (the timeline of the SECOND launch)
            split_now = 0; exit_tag_now = ra.exit_tag;
            kernel_done.store(0); pairs_left.store(np2);
            {
                ProfScope ps(c, KSS_K_RESIDENT, true);
                std::string lerr;
                if (launch_resident(c->stream, P.nn_fma != 0, np2, ra, lerr) != KSS_OK) return set_err(c, KSS_ERR_HIP, "resident: the second launch of a split batch failed");
            }
            c->pool.run_threads(std::max(1, std::min(res_threads, (np2 + 7) / 8)), serve);
        }
    }
    if (failed.load()) {
        // let every workgroup that still waits (or has not started yet) leave, then report: the caller starts over on the
        // launch-per-pass engine (the resident kernel has written nothing but its result slots -- and, for candidates, rows
        // and row tags that carry this launch's numbers)
        for (int p = 0; p < np; ++p)
            if (H[p].phase != PH_DONE) send(p, nullptr, 0, 2, true);
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (cand) { c->ws_dirty = true; KCHK(restore_zero_at_rest(c)); }
        static const int backoff_s = getenv("KSS_RES_BACKOFF_S") ? atoi(getenv("KSS_RES_BACKOFF_S")) : 30;
        c->res_backoff[cand ? 1 : 0] = std::chrono::steady_clock::now() + std::chrono::seconds(backoff_s);   // not again for a while: each failure costs the polls' bound (~1 s)
        {
            int unfinished = 0, k_min = 1 << 30, k_max = 0;
            for (int p = 0; p < np; ++p)
                if (H[p].phase != PH_DONE) { ++unfinished; k_min = std::min(k_min, H[p].k); k_max = std::max(k_max, H[p].k); }
            std::fprintf(stderr, "[kss] the pair-resident kernel left before every pair was finished (a stalled host thread?); running the launch-per-pass engine"
                                 " (and for the next 30 s on this context)"
                                 " [%s, %d of %d pairs unfinished, waiting for passes %d..%d, %d host threads, stream query: %s]\n",
                         cand ? "candidates" : "cell lists", unfinished, np, k_min, k_max, nthreads, hipGetErrorName((hipError_t)query_said.load()));
            if (getenv("KSS_DEBUG_RES")) {   // what each unfinished pair was waiting for, and what is there
                for (int p = 0; p < np; ++p) {
                    if (H[p].phase == PH_DONE) continue;
                    const unsigned long long want = a.seq0 + (unsigned long long)H[p].k;
                    std::fprintf(stderr, "[kss]   pair %d: phase %d, waiting for the sums of pass %d (seq %llu); slots hold seq:", p, H[p].phase, H[p].k, want & 0xffffffffull);
                    const unsigned long long* sl = h_seq + (size_t)2 * NSUMS * p;
                    for (int k = 0; k < NSUMS; ++k) std::fprintf(stderr, " %llu", sl[2 * k + 1] & 0xffffffffull);
                    std::fprintf(stderr, "\n");
                }
            }
        }
        return KSS_OK;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));   // every workgroup has left (its last act was the publication just consumed)
    if (c->prof > 0) { c->prof_n[KSS_K_RESIDENT_PASS] += units.load(); }
    for (int p = 0; p < np; ++p) {
        kss_icp_result& r = results[p];
        std::memcpy(r.T, H[p].fin, 16 * sizeof(float));
        r.iterations = H[p].iters; r.converged = H[p].cancelled ? 0 : H[p].converged; r.state = H[p].cancelled ? KSS_STATE_NOT_CONVERGED : H[p].state;
        r.last_mse = H[p].last_mse; r.fitness = P.compute_fitness && !H[p].cancelled ? H[p].fitness : 0.0; r.pair_id = p;
    }
    if (judge >= 0) { c->spec_ran = true; c->spec_cancelled = cancel_all.load() != 0; }
    if (a.idx_out) {
        const size_t n0 = (size_t)pl.g[0].ns;
        if (P.fitness_idx) HIPCHK(c, hipMemcpyAsync(P.fitness_idx, a.idx_out, n0 * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
        if (P.fitness_d2) HIPCHK(c, hipMemcpyAsync(P.fitness_d2, a.d2_out, n0 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    if (stamps_on) {
        c->last_stamps.resize((size_t)np * 16);
        HIPCHK(c, hipMemcpy(c->last_stamps.data(), a.stamps, c->last_stamps.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    }
    *handled = true;
    return KSS_OK;
}

// The ICP loop over a packed workspace (src0/tgt4 already filled).
static int icp_loop(kss_ctx* c, const IcpPlan& pl_in, const kss_icp_params& P, kss_icp_result* results) {
    if (pl_in.gridb) {   // batches whose pairs fit a CU each: one launch, every pair resident for its whole registration
        bool handled = false;
        KCHK(resident_loop(c, pl_in, P, results, &handled));
        if (handled) return KSS_OK;
        if (pl_in.shared_target) {
            // candidates of one target that the resident engine declined: the brute-force engine (the packed clouds stay where
            // they are, in cell order; the engines agree on every correspondence)
            std::vector<int64_t> ns(pl_in.npairs), nt(pl_in.npairs);
            for (int p = 0; p < pl_in.npairs; ++p) { ns[p] = pl_in.g[p].ns; nt[p] = pl_in.g[p].nt; }
            IcpPlan bp;
            KCHK(build_plan(c, ns.data(), nt.data(), pl_in.npairs, true, P.nn_sources_per_thread, P.nn_target_splits, KSS_NN_BRUTE, bp));
            bp.src_in_cell_order = true;
            KCHK(stage_plan(c, bp));
            kss_icp_params Pb = P;
            Pb.nn_mode = KSS_NN_BRUTE;
            return icp_loop(c, bp, Pb, results);
        }
    }
    if (!pl_in.grid && !pl_in.gridb && pl_in.shared_target) {   // the candidate batch of a registration: one launch, every workgroup resident
        bool handled = false;
        KCHK(resident_loop(c, pl_in, P, results, &handled, true));
        if (handled) return KSS_OK;
    }
    if (c->spec_judge >= 0) return KSS_OK;   // a speculative batch (kss_register) runs on that engine only: spec_ran stays false, the caller takes the sequential route
    const IcpPlan* plan = &pl_in;   // may change to the brute-force plan below
    IcpPlan brute_plan;
    const int np = pl_in.npairs;
    std::vector<Convergence> conv(np);
    std::vector<float> fin((size_t)np * 16), Tk((size_t)np * 16);
    std::vector<int> iters(np, 0), active(np, 1), converged(np, 0), state(np, 0), solved(np, 0), was_active;
    std::vector<double> last_mse(np, 0.0);
    const bool full = P.trace_sums != nullptr;   // traced runs report all 20 sums of every pass
    PairState* hs = (PairState*)c->h_state;
    PairState* bar = pl_in.gridb ? bar_state_table(c, np) : nullptr;   // (null without a large BAR)
    float I[16];
    mat4_identity(I);
    for (int p = 0; p < np; ++p) {
        Convergence& cv = conv[p];
        cv.max_iterations = P.max_iterations;
        cv.rotation_threshold = 1.0 - P.transformation_epsilon;
        cv.translation_threshold = P.transformation_epsilon;
        cv.mse_rel = P.euclidean_fitness_epsilon;
        cv.mse_abs = P.abs_mse_epsilon;
        cv.fixed_iterations = P.fixed_iterations != 0;
        mat4_identity(&fin[(size_t)p * 16]);
        set_state(hs[p], I, 1, 0);
        mirror_state(bar, p, hs[p]);
    }
    const double max_d2 = P.max_corr_dist * P.max_corr_dist;
    const double* hsum = (const double*)c->h_sums;
    if (P.trace_n) *P.trace_n = 0;
    int n_active = P.max_iterations > 0 ? np : 0;
    if (P.max_iterations <= 0)
        for (int p = 0; p < np; ++p) { active[p] = 0; }
    int it = 0;
    GatedGuard gated_guard(c);

    // ---- per-pair solve + convergence test of the pairs [p0, p1) after a pass: pairs are independent (a large batch is
    // split over a few host threads; each pair is handled by exactly one thread with the serial code, so the results do
    // not depend on the split).  poll: the pass published per pair (batched cell lists) and the pairs are picked up as
    // their sums land, in launch order, while the rest of the launch is still running.  Returns the pairs that finished.
    auto solve_pairs = [&](int p0, int p1, unsigned long long pass_seq, bool poll_first, int* n_finished) -> int {
        std::atomic<int> finished{0}, stuck{0};
        bool poll = poll_first;
        std::fill(solved.begin() + p0, solved.begin() + p1, 0);
        auto solve = [&](int pb, int pe) {
            int fin_here = 0;
            for (int p = pb; p < pe; ++p) {
                if (!active[p] || solved[p]) continue;
                if (poll && wait_pair(c, p, pass_seq) != KSS_OK) { stuck.fetch_add(1); continue; }   // (retried below after a stream sync)
                const double* s = hsum + (size_t)p * NSUMS;
                if ((int)s[0] < P.min_correspondences) {   // PCL: "Not enough correspondences found"
                    state[p] = KSS_STATE_NO_CORRESPONDENCES; converged[p] = 0; active[p] = 0; ++fin_here; solved[p] = 1;
                    set_state(hs[p], I, 0, 0);
                    mirror_state(bar, p, hs[p]);
                    continue;
                }
                float* tk = &Tk[(size_t)p * 16];
                rigid_from_sums(s, tk);
                mat4_mul(tk, &fin[(size_t)p * 16], &fin[(size_t)p * 16]);   // final = transformation_ * final
                ++iters[p];
                const double mse = s[16] / s[0];
                last_mse[p] = mse;
                if (p == 0 && P.trace_n && *P.trace_n < P.trace_cap) {
                    if (P.trace_sums) std::memcpy(P.trace_sums + (size_t)(*P.trace_n) * NSUMS, s, NSUMS * sizeof(double));
                    if (P.trace_Tk) std::memcpy(P.trace_Tk + (size_t)(*P.trace_n) * 16, tk, 16 * sizeof(float));
                    ++*P.trace_n;
                }
                const bool done = conv[p].has_converged(iters[p], tk, mse);
                state[p] = conv[p].state;
                solved[p] = 1;
                if (done) {
                    converged[p] = 1; active[p] = 0; ++fin_here;
                    set_state(hs[p], tk, 0, 1);
                } else {
                    set_state(hs[p], tk, 1, 1);   // next sweep applies T_k on load (transformCloud)
                }
                mirror_state(bar, p, hs[p]);
            }
            finished.fetch_add(fin_here, std::memory_order_relaxed);
        };
        const int n = p1 - p0;
        if (n >= 64) {
            std::atomic<int> next{0};
            static const int kBlock = [] { const char* e = getenv("KSS_SOLVE_BLOCK"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 64; }();   // pairs claimed at a time, in launch order: the first threads start while the GPU is still on later pairs.  Tuning hook; measured at C3: blocks of 4 pairs 11.7 ms (threads share cache lines of the per-pair arrays), 16: 9.85, 64: 9.5
            c->pool.parallel_for(n, [&](int, int) {
                for (;;) {
                    const int pb = p0 + next.fetch_add(1) * kBlock;
                    if (pb >= p1) break;
                    solve(pb, std::min(p1, pb + kBlock));
                }
            });
        } else {
            solve(p0, p1);
        }
        if (stuck.load() > 0) {   // a pair did not publish in time: synchronize (surfaces a faulted kernel), then finish the stragglers
            HIPCHK(c, hipStreamSynchronize(c->stream));
            for (int p = p0; p < p1; ++p) {
                if (!active[p] || solved[p]) continue;
                if (!collect_pair(c, p, pass_seq)) return set_err(c, KSS_ERR_HIP, "kernel finished without publishing its result");
            }
            poll = false;        // their sums are in h_sums now
            solve(p0, p1);       // (pairs already solved in this pass are skipped)
        }
        *n_finished = finished.load(std::memory_order_relaxed);
        return KSS_OK;
    };

    while (n_active > 0) {
        const float4* d_in = it == 0 ? (const float4*)c->src0.p : (const float4*)c->cur[(it - 1) & 1].p;
        float4* d_out = (float4*)c->cur[it & 1].p;
        // (cancelled if this iteration converges; never with an all-reduce callback: its collective would queue up on
        // this stream BEHIND the polling kernel, whose transform depends on it)
        c->gated.want_next = plan->grid && !P.allreduce && it + 1 < P.max_iterations;
        c->gated.want_full = full;
        c->gated.max_steps = P.max_iterations - (it + 1);
        // batched cell lists: every pair's sums are published as its last workgroup finishes, and the solve loop
        // picks the pairs up in that order while the rest of the launch is still running
        c->defer_wait = plan->gridb;
        const auto tb0 = c->timing ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
        const int rc_pass = nn_pass(c, *plan, P.nn_fma != 0, d_in, d_out, max_d2, nullptr, nullptr, full, active.data());
        const bool deferred = c->defer_wait;
        c->defer_wait = false;
        const auto tb1 = c->timing ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
        KCHK(rc_pass);
        const unsigned long long pass_seq = c->seq;
        // source rows split over ranks: the sums of all ranks, identical on every rank from here on
        if (P.allreduce && P.allreduce(P.allreduce_user, (double*)c->h_sums, NSUMS) != 0)
            return set_err(c, KSS_ERR_RCCL, "icp: the allreduce callback failed");
        if (plan->gridb) was_active = active;   // the fallback statistics of this pass need the pairs that were active when it ran
        int fin_now = 0;
        KCHK(solve_pairs(0, np, pass_seq, deferred, &fin_now));
        if (c->timing && deferred) {   // batch: time to enqueue the pass vs time until every pair was solved
            const auto tb2 = std::chrono::steady_clock::now();
            c->t_launch_us += std::chrono::duration<double, std::micro>(tb1 - tb0).count();
            c->t_wait_us += std::chrono::duration<double, std::micro>(tb2 - tb1).count();
        }
        if (plan->gridb) {
            // Batched cell lists: slot 19 counts the lanes that ended in the in-wave brute-force fallback.  When more
            // than 10 % of the active sources did (badly posed pairs), the rest of this call runs on the brute-force
            // engine, whose tiled sweep is several times faster at that job; the engines agree bit for bit on every
            // correspondence, so the switch only changes speed.  The packed clouds stay where they are.
            double fallback = 0.0, act = 0.0;
            for (int p = 0; p < np; ++p)
                if (was_active[p]) { fallback += hsum[(size_t)p * NSUMS + NSUMS - 1]; act += (double)plan->g[p].ns; }
            if (fallback > 0.10 * act && !getenv("KSS_GRID_NOSWITCH")) {
                std::vector<int64_t> ns(np), nt(np);
                for (int p = 0; p < np; ++p) { ns[p] = plan->g[p].ns; nt[p] = plan->g[p].nt; }
                KCHK(build_plan(c, ns.data(), nt.data(), np, pl_in.shared_target, P.nn_sources_per_thread, P.nn_target_splits, KSS_NN_BRUTE, brute_plan));
                brute_plan.src_in_cell_order = true;
                KCHK(stage_plan(c, brute_plan));
                plan = &brute_plan;
            }
        }
        n_active -= fin_now;
        ++it;
    }
    for (int p = 0; p < np; ++p) {
        kss_icp_result& r = results[p];
        std::memcpy(r.T, &fin[(size_t)p * 16], 16 * sizeof(float));
        r.iterations = iters[p]; r.converged = converged[p]; r.state = state[p];
        r.last_mse = last_mse[p]; r.fitness = 0.0; r.pair_id = p;
    }
    if (P.compute_fitness) {
        // getFitnessScore(): NN of final * ORIGINAL input, mean d2 over all source points
        // (pad[0]: which work buffer holds the positions of the pair's last pass -- pass k writes cur[k & 1] -- so that the
        // cell-list pass can measure how far each source is from where its skip state was last brought up to date)
        const bool cell_lists = plan->grid || plan->gridb;
        for (int p = 0; p < np; ++p) {
            set_state(hs[p], &fin[(size_t)p * 16], 1, 1);
            // (a pair that ended without correspondences took part in one more pass than it counted: its last position is in
            // the other buffer -- no claim is made for it, its sources simply search)
            if (cell_lists && plan == &pl_in && iters[p] >= 1 && state[p] != KSS_STATE_NO_CORRESPONDENCES) hs[p].pad[0] = 1 + ((iters[p] - 1) & 1);
            mirror_state(bar, p, hs[p]);
        }
        c->fit_last = cell_lists && plan == &pl_in;
        struct FitGuard { kss_ctx* c; ~FitGuard() { c->fit_last = false; } } fit_guard{c};
        c->gated.want_next = false;
        int32_t* d_idx = nullptr;
        float* d_d2 = nullptr;
        if (P.fitness_idx || P.fitness_d2) {   // per-source correspondences of this pass (indexed by original source index)
            KCHK(ensure(c, c->stage_idx, (size_t)plan->total_src * sizeof(int32_t)));
            KCHK(ensure(c, c->stage_d2, (size_t)plan->total_src * sizeof(float)));
            d_idx = (int32_t*)c->stage_idx.p; d_d2 = (float*)c->stage_d2.p;
        }
        KCHK(nn_pass(c, *plan, P.nn_fma != 0, (const float4*)c->src0.p, (float4*)c->cur[0].p, max_d2, d_idx, d_d2, true));
        if (P.allreduce) {   // mean over ALL source rows of the job
            double v[2] = {hsum[17], (double)plan->g[0].ns};
            if (P.allreduce(P.allreduce_user, v, 2) != 0) return set_err(c, KSS_ERR_RCCL, "icp: the allreduce callback failed");
            results[0].fitness = v[0] / v[1];
        } else {
            for (int p = 0; p < np; ++p) results[p].fitness = hsum[(size_t)p * NSUMS + 17] / (double)plan->g[p].ns;
        }
        if (d_idx) {
            const size_t n0 = (size_t)plan->g[0].ns;
            if (P.fitness_idx) HIPCHK(c, hipMemcpyAsync(P.fitness_idx, d_idx, n0 * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
            if (P.fitness_d2) HIPCHK(c, hipMemcpyAsync(P.fitness_d2, d_d2, n0 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
        }
    }
    return KSS_OK;
}

// zero-at-rest buffers (kss_ctx.hpp): cleared here when the previous call on this context did not finish
static int restore_zero_at_rest(kss_ctx* c) {
    if (!c->ws_dirty) return KSS_OK;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (DevBuf* b : {&c->pair_ticket, &c->g_count, &c->g_counts})
        if (b->p) HIPCHK(c, hipMemsetAsync(b->p, 0, b->cap, c->stream));
    c->ws_dirty = false;
    return KSS_OK;
}
struct DirtyGuard {   // a call that returns early leaves the zero-at-rest buffers in an unknown state
    kss_ctx* c; bool ok = false;
    explicit DirtyGuard(kss_ctx* c_) : c(c_) {}
    ~DirtyGuard() { if (!ok) c->ws_dirty = true; }
};

int icp_run_dev(kss_ctx* c, const void* d_src, const int64_t* src_off, const void* d_tgt, const int64_t* tgt_off,
                int npairs, bool shared_target, int dtype, const kss_icp_params* p, kss_icp_result* results) {
    if (!c || !d_src || !d_tgt || !src_off || !tgt_off || !p || !results || npairs <= 0) return set_err(c, KSS_ERR_ARG, "icp: bad argument");
    if (p->allreduce && npairs != 1) return set_err(c, KSS_ERR_ARG, "icp: the source-row split (allreduce) is for a single pair");
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<int64_t> ns(npairs), nt(npairs);
    for (int i = 0; i < npairs; ++i) {
        ns[i] = src_off[i + 1] - src_off[i];
        nt[i] = shared_target ? tgt_off[1] - tgt_off[0] : tgt_off[i + 1] - tgt_off[i];
    }
    IcpPlan pl;
    const auto t0 = std::chrono::steady_clock::now();
    KCHK(restore_zero_at_rest(c));
    DirtyGuard guard(c);
    c->timing = getenv("KSS_TIMING") != nullptr;
    auto lap = [&](const char* what, std::chrono::steady_clock::time_point& from) {   // KSS_TIMING: host time of each setup stage
        if (!c->timing) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[kss]   %s %.1f us\n", what, std::chrono::duration<double, std::micro>(now - from).count());
        from = now;
    };
    auto ts = t0;
    KCHK(build_plan(c, ns.data(), nt.data(), npairs, shared_target, p->nn_sources_per_thread, p->nn_target_splits, p->nn_mode, pl));
    lap("plan", ts);
    KCHK(stage_plan(c, pl));
    lap("stage", ts);
    KCHK(pack_clouds(c, pl, d_src, src_off, d_tgt, tgt_off, dtype));
    lap("pack (enqueue)", ts);
    KCHK(grid_setup(c, pl));
    KCHK(grid_setup_batch(c, pl));
    lap("cell lists (bbox sync + enqueue)", ts);
    if (c->timing) HIPCHK(c, hipStreamSynchronize(c->stream));   // (only to split setup from loop in the KSS_TIMING report)
    lap("drain", ts);
    const auto t1 = std::chrono::steady_clock::now();
    c->t_launch_us = c->t_wait_us = 0;
    const int rc = icp_loop(c, pl, *p, results);
    guard.ok = rc == KSS_OK;
    const auto t2 = std::chrono::steady_clock::now();
    c->last_setup_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    c->last_loop_ms = std::chrono::duration<double, std::milli>(t2 - t1).count();
    if (c->timing)
        std::fprintf(stderr, "[kss] setup %.3f ms, loop+fitness %.3f ms (fused launches: enqueue %.1f us, wait %.1f us, rest = host math)\n",
                     c->last_setup_ms, c->last_loop_ms, c->t_launch_us, c->t_wait_us);
    return rc;
}



int nn_generic_dev(kss_ctx* c, const void* d_src, int64_t ns, const void* d_tgt, int64_t nt, int dtype,
                          int32_t* d_idx, float* d_d2, double sums_out[NSUMS]) {
    if (!c || !d_src || !d_tgt) return set_err(c, KSS_ERR_ARG, "nn: null cloud");
    if (ns <= 0 || nt <= 0) return set_err(c, KSS_ERR_ARG, "nn: empty cloud");
    HIPCHK(c, hipSetDevice(c->device));
    IcpPlan pl;
    KCHK(restore_zero_at_rest(c));
    DirtyGuard guard(c);
    KCHK(build_plan(c, &ns, &nt, 1, false, 0, 0, KSS_NN_AUTO, pl));
    KCHK(stage_plan(c, pl));
    const int64_t so[2] = {0, ns}, to[2] = {0, nt};
    KCHK(pack_clouds(c, pl, d_src, so, d_tgt, to, dtype));
    KCHK(grid_setup(c, pl));
    float I[16];
    mat4_identity(I);
    set_state(((PairState*)c->h_state)[0], I, 1, 0);
    KCHK(nn_pass(c, pl, false, (const float4*)c->src0.p, (float4*)c->cur[0].p, 1e300, d_idx, d_d2, true));
    if (sums_out) std::memcpy(sums_out, c->h_sums, NSUMS * sizeof(double));
    guard.ok = true;
    return KSS_OK;
}


}  // namespace kss
