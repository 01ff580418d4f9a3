// kss_api.hip -- the C-ABI of include/kssicp.h: context and profiling entry points, and the compute entry points over
// the engine (kss_engine.hip) and the kernel launchers (kss_kernels / kss_grid / kss_aivs / kss_knn / kss_octree .hip).
// There is no CPU compute fallback anywhere in this file.
#pragma clang fp contract(off)

#include "kss_ctx.hpp"

extern "C" {

int kss_version(void) { return KSS_VERSION; }

const char* kss_status_string(int s) {
    switch (s) {
        case KSS_OK: return "ok";
        case KSS_ERR_ARG: return "invalid argument";
        case KSS_ERR_HIP: return "HIP runtime error";
        case KSS_ERR_NOMEM: return "out of memory";
        case KSS_ERR_NODEVICE: return "no usable GPU device";
        case KSS_ERR_CAPACITY: return "output buffer too small";
        case KSS_ERR_RCCL: return "RCCL error";
        default: return "unknown status";
    }
}

const char* kss_last_error(const kss_ctx* ctx) { return ctx ? ctx->err.c_str() : ""; }

static int ctx_create_common(int device_id, void* stream, bool borrow, kss_ctx** out) {
    if (!out) return KSS_ERR_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return KSS_ERR_NODEVICE;
    if (device_id < 0 || device_id >= n) return KSS_ERR_ARG;
    if (hipSetDevice(device_id) != hipSuccess) return KSS_ERR_NODEVICE;
    kss_ctx* c = new (std::nothrow) kss_ctx();
    if (!c) return KSS_ERR_NOMEM;
    c->device = device_id;
    if (borrow) {
        c->stream = (hipStream_t)stream;
        c->own_stream = false;
    } else {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return KSS_ERR_HIP; }
        c->own_stream = true;
    }
    kss_live_contexts().fetch_add(1);
    // launch sequence numbers are unique per PROCESS, not per context: rows handed over as {bits, number} granules are taken
    // on the number alone, and a freed workspace block can come back to the next context with the old context's rows in it
    static std::atomic<unsigned long long> ctx_serial{0};
    c->seq = (ctx_serial.fetch_add(1) + 1ull) << 40;
    *out = c;
    return KSS_OK;
}

int kss_ctx_create(int device_id, kss_ctx** out) { return ctx_create_common(device_id, nullptr, false, out); }
int kss_ctx_create_on_stream(int device_id, void* hip_stream, kss_ctx** out) {
    return ctx_create_common(device_id, hip_stream, true, out);
}

int kss_ctx_destroy(kss_ctx* c) {
    if (!c) return KSS_ERR_ARG;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    prof_collect(c);
    for (auto& ep : c->ev_pool) { hipEventDestroy(ep.a); hipEventDestroy(ep.b); }
    c->ev_pool.clear();
    DevBuf* bufs[] = {&c->tgt4, &c->src0, &c->cur[0], &c->cur[1], &c->keys, &c->partials, &c->sums, &c->nn_work,
                      &c->red_work, &c->pair_red, &c->state, &c->cs, &c->scratch_a, &c->scratch_b, &c->scratch_c,
                      &c->stage_src, &c->stage_tgt, &c->stage_idx, &c->stage_d2, &c->stage_out, &c->g_counts, &c->g_start,
                      &c->g_cursor, &c->g_bsums, &c->g_sorted, &c->g_list, &c->g_count, &c->g_bbox, &c->g_partials, &c->g_start2, &c->g_pairs, &c->g_stamps, &c->g_pos, &c->g_nnst, &c->res_pos, &c->res_wc, &c->res_perm, &c->cand_tags, &c->pack_seg, &c->reg_s, &c->reg_t, &c->reg_p, &c->reg_all, &c->reg_f, &c->reg_g, &c->oct_pts, &c->oct_cen, &c->oct_a, &c->oct_b, &c->oct_tmp, &c->pair_ticket, &c->pre_partials, &c->pre_state, &c->g_rowpair, &c->g_gate};
    for (DevBuf* b : bufs)
        if (b->p) hipFree(b->p);
    if (c->h_sums) hipHostFree(c->h_sums);
    if (c->h_seq) hipHostFree(c->h_seq);
    if (c->h_box) hipHostFree(c->h_box);
    if (c->h_xf) hipHostFree(c->h_xf);
    if (c->gate_bar) hipFree(c->gate_bar);
    if (c->res_gate) hipFree(c->res_gate);
    if (c->bar_state) hipFree(c->bar_state);
    if (c->h_state) hipHostFree(c->h_state);
    for (kss_ctx* w : c->workers) kss_ctx_destroy(w);
    c->workers.clear();
    if (c->own_stream) hipStreamDestroy(c->stream);
    kss_live_contexts().fetch_sub(1);
    delete c;
    return KSS_OK;
}

int kss_ctx_synchronize(kss_ctx* c) {
    if (!c) return KSS_ERR_ARG;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return KSS_OK;
}

void* kss_ctx_stream(kss_ctx* c) { return c ? (void*)c->stream : nullptr; }

int kss_ctx_set_nn_mode(kss_ctx* c, int nn_mode) {
    if (!c || nn_mode < KSS_NN_AUTO || nn_mode > KSS_NN_GRID) return KSS_ERR_ARG;
    c->nn_mode = nn_mode;
    return KSS_OK;
}

int kss_profile_enable(kss_ctx* c, int on) {
    if (!c) return KSS_ERR_ARG;
    c->prof = on > 0 ? on : 0;
    for (int k = 0; k < KSS_K_COUNT; ++k) c->prof_tick[k] = 0;
    if (c->prof) {   // pre-create the event pairs a timed region will use
        HIPCHK(c, hipSetDevice(c->device));
        while (c->ev_pool.size() < 4096) {
            kss_ctx::EvPair ep;
            if (hipEventCreate(&ep.a) != hipSuccess) break;
            if (hipEventCreate(&ep.b) != hipSuccess) { hipEventDestroy(ep.a); break; }
            c->ev_pool.push_back(ep);
        }
    }
    return KSS_OK;
}
int kss_grid_stats(kss_ctx* c, double out[8]) {
    if (!c || !out) return KSS_ERR_ARG;
    for (int k = 0; k < 8; ++k) out[k] = c->grid_stats[k];
    return KSS_OK;
}
// diagnostic: distance evaluations of the fused grid launches since the last call (KSS_GRID_STAMPS=1): {sum, launches}
int kss_debug_grid_evals(kss_ctx* c, double out[2]) {
    if (!c || !out) return KSS_ERR_ARG;
    out[0] = c->evals_sum; out[1] = c->evals_launches;
    c->evals_sum = 0.0; c->evals_launches = 0.0;
    return KSS_OK;
}

// diagnostic: in-kernel timeline stamps of the last fused grid launch (KSS_GRID_STAMPS=1); not part of the ABI header
int kss_debug_grid_stamps(kss_ctx* c, unsigned long long* out, int64_t cap) {
    if (!c || !out) return KSS_ERR_ARG;
    if (c->stamps_nblk > 0 && c->g_stamps.p) {   // KSS_GRID_STAMPS=2: the buffer of the last launch that was waited for
        hipStreamSynchronize(c->stream);
        c->last_stamps.resize((size_t)c->stamps_nblk * 16);
        if (hipMemcpy(c->last_stamps.data(), (const unsigned long long*)c->g_stamps.p + (size_t)(c->stamps_seq & 1) * c->stamps_nblk * 16,
                      c->last_stamps.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return KSS_ERR_HIP;
    }
    const int64_t n = std::min<int64_t>(cap, (int64_t)c->last_stamps.size());
    for (int64_t i = 0; i < n; ++i) out[i] = c->last_stamps[(size_t)i];
    return (int)n;
}
int kss_profile_reset(kss_ctx* c) {
    if (!c) return KSS_ERR_ARG;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    prof_collect(c);
    for (int k = 0; k < KSS_K_COUNT; ++k) { c->prof_ms[k] = 0; c->prof_n[k] = 0; }
    return KSS_OK;
}
int kss_profile_event_overhead(kss_ctx* c, double* ms) {
    if (!c || !ms) return KSS_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    hipEvent_t a = nullptr, b = nullptr;
    HIPCHK(c, hipEventCreate(&a));
    HIPCHK(c, hipEventCreate(&b));
    for (int i = 0; i < 16; ++i) launch_empty(c->stream);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    double tot = 0.0;
    const int reps = 200;
    for (int i = 0; i < reps; ++i) {   // idle queue -> event, launch, event: the pattern of one fused ICP iteration
        hipEventRecord(a, c->stream);
        launch_empty(c->stream);
        hipEventRecord(b, c->stream);
        HIPCHK(c, hipStreamSynchronize(c->stream));
        float e = 0.f;
        HIPCHK(c, hipEventElapsedTime(&e, a, b));
        tot += e;
    }
    hipEventDestroy(a);
    hipEventDestroy(b);
    *ms = tot / reps;
    return KSS_OK;
}
int kss_profile_get(kss_ctx* c, int k, double* total_ms, int64_t* launches) {
    if (!c || k < 0 || k >= KSS_K_COUNT) return KSS_ERR_ARG;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    prof_collect(c);
    if (total_ms) *total_ms = c->prof_ms[k];
    if (launches) *launches = c->prof_n[k];
    return KSS_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// C-ABI: compute entry points
// ---------------------------------------------------------------------------------------------
extern "C" {

int kss_icp_default_params(kss_icp_params* p) {
    if (!p) return KSS_ERR_ARG;
    std::memset(p, 0, sizeof *p);
    p->max_iterations = 1000;             // Main_KSS_ICP.cpp:81
    p->max_corr_dist = 1.0;               // KSS_ICP.hpp:156
    p->transformation_epsilon = 1e-10;    // :157
    p->euclidean_fitness_epsilon = 0.001; // :158
    p->abs_mse_epsilon = 1e-12;
    p->min_correspondences = 3;
    p->compute_fitness = 1;
    return KSS_OK;
}

int kss_icp_dev(kss_ctx* c, const float* d_src, int64_t ns, const float* d_tgt, int64_t nt,
                const kss_icp_params* p, kss_icp_result* res) {
    if (!c) return KSS_ERR_ARG;
    if (ns <= 0 || nt <= 0) return set_err(c, KSS_ERR_ARG, "icp: empty cloud");
    const int64_t so[2] = {0, ns}, to[2] = {0, nt};
    return icp_run_dev(c, d_src, so, d_tgt, to, 1, false, KSS_F32, p, res);
}

int kss_icp(kss_ctx* c, const float* src, int64_t ns, const float* tgt, int64_t nt,
            const kss_icp_params* p, kss_icp_result* res) {
    if (!c || !src || !tgt) return set_err(c, KSS_ERR_ARG, "icp: null cloud");
    if (ns <= 0 || nt <= 0) return set_err(c, KSS_ERR_ARG, "icp: empty cloud");
    HIPCHK(c, hipSetDevice(c->device));
    KCHK(upload(c, c->stage_src, src, (size_t)ns * 3 * sizeof(float)));
    KCHK(upload(c, c->stage_tgt, tgt, (size_t)nt * 3 * sizeof(float)));
    return kss_icp_dev(c, (const float*)c->stage_src.p, ns, (const float*)c->stage_tgt.p, nt, p, res);
}

int kss_icp_batch_dev(kss_ctx* c, const float* d_src_all, const int64_t* src_off, const float* d_tgt_all,
                      const int64_t* tgt_off, int npairs, const kss_icp_params* p, kss_icp_result* results) {
    if (!c) return KSS_ERR_ARG;
    return icp_run_dev(c, d_src_all, src_off, d_tgt_all, tgt_off, npairs, false, KSS_F32, p, results);
}

int kss_icp_batch(kss_ctx* c, const float* src_all, const int64_t* src_off, const float* tgt_all,
                  const int64_t* tgt_off, int npairs, const kss_icp_params* p, kss_icp_result* results) {
    if (!c || !src_all || !tgt_all || !src_off || !tgt_off || npairs <= 0) return set_err(c, KSS_ERR_ARG, "icp_batch: bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    const int64_t s0 = src_off[0], s1 = src_off[npairs], t0 = tgt_off[0], t1 = tgt_off[npairs];
    if (s1 <= s0 || t1 <= t0) return set_err(c, KSS_ERR_ARG, "icp_batch: empty batch");
    KCHK(upload(c, c->stage_src, src_all + 3 * s0, (size_t)(s1 - s0) * 3 * sizeof(float)));
    KCHK(upload(c, c->stage_tgt, tgt_all + 3 * t0, (size_t)(t1 - t0) * 3 * sizeof(float)));
    std::vector<int64_t> so(npairs + 1), to(npairs + 1);
    for (int i = 0; i <= npairs; ++i) { so[i] = src_off[i] - s0; to[i] = tgt_off[i] - t0; }
    return icp_run_dev(c, c->stage_src.p, so.data(), c->stage_tgt.p, to.data(), npairs, false, KSS_F32, p, results);
}

// ---- NN -----------------------------------------------------------------------------------------
int kss_nn_dev(kss_ctx* c, const float* d_src, int64_t ns, const float* d_tgt, int64_t nt, int32_t* d_idx, float* d_d2) {
    return nn_generic_dev(c, d_src, ns, d_tgt, nt, KSS_F32, d_idx, d_d2, nullptr);
}

int kss_nn(kss_ctx* c, const float* src, int64_t ns, const float* tgt, int64_t nt, int32_t* idx, float* d2) {
    if (!c || !src || !tgt) return set_err(c, KSS_ERR_ARG, "nn: null cloud");
    if (ns <= 0 || nt <= 0) return set_err(c, KSS_ERR_ARG, "nn: empty cloud");
    HIPCHK(c, hipSetDevice(c->device));
    KCHK(upload(c, c->stage_src, src, (size_t)ns * 3 * sizeof(float)));
    KCHK(upload(c, c->stage_tgt, tgt, (size_t)nt * 3 * sizeof(float)));
    KCHK(ensure(c, c->stage_idx, (size_t)ns * sizeof(int32_t)));
    KCHK(ensure(c, c->stage_d2, (size_t)ns * sizeof(float)));
    KCHK(kss_nn_dev(c, (const float*)c->stage_src.p, ns, (const float*)c->stage_tgt.p, nt, (int32_t*)c->stage_idx.p, (float*)c->stage_d2.p));
    if (idx) HIPCHK(c, hipMemcpyAsync(idx, c->stage_idx.p, (size_t)ns * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    if (d2) HIPCHK(c, hipMemcpyAsync(d2, c->stage_d2.p, (size_t)ns * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return KSS_OK;
}

// ---- k-NN and normals -------------------------------------------------------------------------------
static int knn_generic_dev(kss_ctx* c, const void* d_q, int64_t nq, const void* d_t, int64_t nt, int dtype, int k, int32_t* d_idx, float* d_d2) {
    if (!c || !d_q || !d_t || !d_idx || !d_d2) return set_err(c, KSS_ERR_ARG, "knn: null argument");
    if (nq <= 0 || nt <= 0 || k < 1 || k > 64) return set_err(c, KSS_ERR_ARG, "knn: need nq, nt > 0 and 1 <= k <= 64");
    if (nq > 0x7fff0000ll || nt > 0x7fff0000ll) return set_err(c, KSS_ERR_ARG, "knn: cloud too large");
    HIPCHK(c, hipSetDevice(c->device));
    const int64_t nt_pad = (nt + NN_TILE - 1) / NN_TILE * NN_TILE;
    KCHK(ensure(c, c->tgt4, (size_t)nt_pad * sizeof(float4)));
    KCHK(ensure(c, c->src0, (size_t)nq * sizeof(float4)));
    if (dtype == KSS_F64) {
        launch_pack_f64_to_f4(c->stream, (const double*)d_q, nq, (float4*)c->src0.p, nq, false);
        launch_pack_f64_to_f4(c->stream, (const double*)d_t, nt, (float4*)c->tgt4.p, nt_pad, true);
    } else {
        launch_pack_f3_to_f4(c->stream, (const float*)d_q, nq, (float4*)c->src0.p, nq, false);
        launch_pack_f3_to_f4(c->stream, (const float*)d_t, nt, (float4*)c->tgt4.p, nt_pad, true);
    }
    size_t part_bytes = 0;
    const int n_split = knn_plan_splits((int)nq, (int)nt_pad, k, &part_bytes);   // few queries x many targets: split the targets
    if (part_bytes) KCHK(ensure(c, c->keys, part_bytes));
    {
        ProfScope ps(c, KSS_K_NN_SWEEP);
        launch_knn_sweep(c->stream, (const float4*)c->src0.p, (int)nq, (const float4*)c->tgt4.p, (int)nt_pad, k, d_idx, d_d2, n_split,
                         part_bytes ? c->keys.p : nullptr);
    }
    HIPCHK(c, hipGetLastError());
    return KSS_OK;
}

int kss_knn_dev(kss_ctx* c, const float* d_query, int64_t nq, const float* d_tgt, int64_t nt, int k, int32_t* d_idx, float* d_d2) {
    KCHK(knn_generic_dev(c, d_query, nq, d_tgt, nt, KSS_F32, k, d_idx, d_d2));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return KSS_OK;
}

int kss_knn(kss_ctx* c, const float* query, int64_t nq, const float* tgt, int64_t nt, int k, int32_t* idx, float* d2) {
    if (!c || !query || !tgt || !idx || !d2) return set_err(c, KSS_ERR_ARG, "knn: null argument");
    if (nq <= 0 || nt <= 0 || k < 1 || k > 64) return set_err(c, KSS_ERR_ARG, "knn: need nq, nt > 0 and 1 <= k <= 64");
    HIPCHK(c, hipSetDevice(c->device));
    KCHK(upload(c, c->stage_src, query, (size_t)nq * 3 * sizeof(float)));
    KCHK(upload(c, c->stage_tgt, tgt, (size_t)nt * 3 * sizeof(float)));
    KCHK(ensure(c, c->stage_idx, (size_t)nq * k * sizeof(int32_t)));
    KCHK(ensure(c, c->stage_d2, (size_t)nq * k * sizeof(float)));
    KCHK(knn_generic_dev(c, c->stage_src.p, nq, c->stage_tgt.p, nt, KSS_F32, k, (int32_t*)c->stage_idx.p, (float*)c->stage_d2.p));
    HIPCHK(c, hipMemcpyAsync(idx, c->stage_idx.p, (size_t)nq * k * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(d2, c->stage_d2.p, (size_t)nq * k * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return KSS_OK;
}

int kss_normals(kss_ctx* c, const double* pts, int64_t n, int k, double* normals) {
    if (!c || !pts || !normals) return set_err(c, KSS_ERR_ARG, "normals: null argument");
    if (n <= 0 || k < 1 || k > 32) return set_err(c, KSS_ERR_ARG, "normals: need n > 0 and 1 <= k <= 32");
    if (k > n) k = (int)n;
    HIPCHK(c, hipSetDevice(c->device));
    KCHK(upload(c, c->scratch_a, pts, (size_t)n * 3 * sizeof(double)));
    KCHK(ensure(c, c->stage_idx, (size_t)n * k * sizeof(int32_t)));
    KCHK(ensure(c, c->stage_d2, (size_t)n * k * sizeof(float)));
    KCHK(ensure(c, c->stage_out, (size_t)n * 3 * sizeof(double)));
    // self k-NN on the float-narrowed cloud (cloud_i.x = pointsVector[i][0]); src0 then holds the cloud as float4
    KCHK(knn_generic_dev(c, c->scratch_a.p, n, c->scratch_a.p, n, KSS_F64, k, (int32_t*)c->stage_idx.p, (float*)c->stage_d2.p));
    launch_normals(c->stream, (const float4*)c->src0.p, (int)n, (const int32_t*)c->stage_idx.p, k, (double*)c->stage_out.p);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(normals, c->stage_out.p, (size_t)n * 3 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return KSS_OK;
}

// estimateNormal_RegularNormal (normalCompute.hpp:614-742): consistent orientation by level-synchronous propagation
// over the 8-NN graph from point 0.  The 8-NN of every point is the device's work (n x 8 exact neighbours); the
// propagation itself is an O(8 n) graph walk whose levels depend on each other, hundreds of them on a surface:
// it runs on the host (one pass over the lists, a per-level stamp instead of the reference's quadratic
// "already listed" scan -- same first-occurrence order, same parents, same result).
int kss_normals_orient(kss_ctx* c, const double* pts, int64_t n, double* normals) {
    if (!c || !pts || !normals) return set_err(c, KSS_ERR_ARG, "normals_orient: null argument");
    if (n <= 0 || n > 0x7fff0000ll) return set_err(c, KSS_ERR_ARG, "normals_orient: bad size");
    HIPCHK(c, hipSetDevice(c->device));
    const int Kn = 8;                                   // :639
    const int k = n < Kn ? (int)n : Kn;
    KCHK(upload(c, c->scratch_a, pts, (size_t)n * 3 * sizeof(double)));
    KCHK(ensure(c, c->stage_idx, (size_t)n * k * sizeof(int32_t)));
    KCHK(ensure(c, c->stage_d2, (size_t)n * k * sizeof(float)));
    KCHK(knn_generic_dev(c, c->scratch_a.p, n, c->scratch_a.p, n, KSS_F64, k, (int32_t*)c->stage_idx.p, (float*)c->stage_d2.p));
    std::vector<int32_t> ki((size_t)n * k);
    std::vector<float> kd((size_t)n * k);
    HIPCHK(c, hipMemcpyAsync(ki.data(), c->stage_idx.p, ki.size() * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(kd.data(), c->stage_d2.p, kd.size() * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    std::vector<char> judge((size_t)n, 0);
    std::vector<int64_t> listed((size_t)n, -1);
    std::vector<int32_t> cur(1, 0), nxt, par;
    judge[0] = 1;                                       // start = 0 (:616, :674)
    for (int64_t level = 0; !cur.empty(); ++level) {
        nxt.clear(); par.clear();
        for (const int32_t u : cur) {
            const size_t row = (size_t)u * k;
            const int first = kd[row] == 0 ? 1 : 0;      // the query itself leads the list when its distance is 0 (:660-665)
            for (int j = first; j < first + k - 1 && j < k; ++j) {
                const int32_t v = ki[row + j];
                if (judge[(size_t)v] || listed[(size_t)v] == level) continue;
                listed[(size_t)v] = level;
                nxt.push_back(v); par.push_back(u);
            }
        }
        for (size_t a = 0; a < nxt.size(); ++a) {
            const double* np_ = normals + 3 * (size_t)par[a];
            double* ns = normals + 3 * (size_t)nxt[a];
            double a1 = np_[0] * ns[0] + np_[1] * ns[1] + np_[2] * ns[2];
            double a2 = -np_[0] * ns[0] - np_[1] * ns[1] - np_[2] * ns[2];
            a1 = a1 > 1 ? 1 : (a1 < -1 ? -1 : a1);
            a2 = a2 > 1 ? 1 : (a2 < -1 ? -1 : a2);
            if (std::acos(a1) > std::acos(a2)) { ns[0] = -ns[0]; ns[1] = -ns[1]; ns[2] = -ns[2]; }   // :716-735
            judge[(size_t)nxt[a]] = 1;
        }
        cur.swap(nxt);
    }
    return KSS_OK;
}

// ---- covariance sums -----------------------------------------------------------------------------
int kss_cov_dev(kss_ctx* c, const float* d_src, const float* d_tgt, const int32_t* d_idx, int64_t n, int64_t nt,
                double max_d2, double sums[KSS_NSUMS]) {
    if (!c || !d_src || !d_tgt || !d_idx || !sums) return set_err(c, KSS_ERR_ARG, "cov: null argument");
    if (n <= 0 || nt <= 0) return set_err(c, KSS_ERR_ARG, "cov: empty input");
    HIPCHK(c, hipSetDevice(c->device));
    const int nb = stream_blocks(n);
    KCHK(ensure(c, c->partials, (size_t)nb * NSUMS * sizeof(double)));
    KCHK(ensure(c, c->sums, NSUMS * sizeof(double)));
    KCHK(ensure_pinned(c, c->h_sums, c->h_sums_cap, NSUMS * sizeof(double)));
    {
        ProfScope ps(c, KSS_K_CORR_REDUCE);
        launch_corr_reduce_idx(c->stream, d_src, d_tgt, d_idx, n, max_d2, (double*)c->partials.p, nb);
        launch_sum_columns(c->stream, (const double*)c->partials.p, nb, NSUMS, (double*)c->sums.p);
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(c->h_sums, c->sums.p, NSUMS * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    std::memcpy(sums, c->h_sums, NSUMS * sizeof(double));
    return KSS_OK;
}

int kss_cov(kss_ctx* c, const float* src, const float* tgt, const int32_t* idx, int64_t n, int64_t nt, double max_d2,
            double sums[KSS_NSUMS]) {
    if (!c || !src || !tgt || !idx || !sums) return set_err(c, KSS_ERR_ARG, "cov: null argument");
    if (n <= 0 || nt <= 0) return set_err(c, KSS_ERR_ARG, "cov: empty input");
    for (int64_t i = 0; i < n; ++i)
        if (idx[i] < 0 || idx[i] >= nt) return set_err(c, KSS_ERR_ARG, "cov: index out of range");
    HIPCHK(c, hipSetDevice(c->device));
    KCHK(upload(c, c->stage_src, src, (size_t)n * 3 * sizeof(float)));
    KCHK(upload(c, c->stage_tgt, tgt, (size_t)nt * 3 * sizeof(float)));
    KCHK(upload(c, c->stage_idx, idx, (size_t)n * sizeof(int32_t)));
    return kss_cov_dev(c, (const float*)c->stage_src.p, (const float*)c->stage_tgt.p, (const int32_t*)c->stage_idx.p, n, nt, max_d2, sums);
}

int kss_rigid_from_sums(const double sums[KSS_NSUMS], float T[16]) {
    if (!sums || !T) return KSS_ERR_ARG;
    if (!(sums[0] >= 1.0)) return KSS_ERR_ARG;
    rigid_from_sums(sums, T);
    return KSS_OK;
}

// ---- pre-shape ------------------------------------------------------------------------------------
// Both clouds of a registration in one call: two launches (sum + centroid, radius + publication), no stream
// synchronisation -- the host spins on the {value, sequence number} slots the second launch writes.

int kss_preshape_stats_pair_dev(kss_ctx* c, const void* d_src, int64_t ns, const void* d_tgt, int64_t nt, int dtype,
                                double c_src[3], double* r_src, double c_tgt[3], double* r_tgt) {
    if (!c || !d_src || !c_src || !r_src) return set_err(c, KSS_ERR_ARG, "preshape: null argument");
    if (d_tgt && (!c_tgt || !r_tgt)) return set_err(c, KSS_ERR_ARG, "preshape: null argument");
    if (ns <= 0 || (d_tgt && nt <= 0)) return set_err(c, KSS_ERR_ARG, "preshape: empty cloud");
    if (dtype != KSS_F32 && dtype != KSS_F64) return set_err(c, KSS_ERR_ARG, "preshape: bad dtype");
    HIPCHK(c, hipSetDevice(c->device));
    const void* xyz[2] = {d_src, d_tgt};
    const int64_t n[2] = {ns, d_tgt ? nt : 0};
    const int nb = preshape_blocks(ns) + (d_tgt ? preshape_blocks(nt) : 0);
    KCHK(ensure(c, c->pre_partials, (size_t)nb * 4 * sizeof(double)));
    if (!c->pre_state.p) {
        KCHK(ensure(c, c->pre_state, 128));   // [0, 16): four tickets, [64, 128): two centroids
        HIPCHK(c, hipMemsetAsync(c->pre_state.p, 0, 128, c->stream));
    }
    KCHK(ensure_pub_slots(c));
    {
        ProfScope ps(c, KSS_K_PRESHAPE);
        launch_preshape_pair(c->stream, xyz, n, dtype, (double*)c->pre_partials.p, (int32_t*)c->pre_state.p,
                             (double*)((char*)c->pre_state.p + 64), c->h_seq_dev, ++c->seq);
    }
    HIPCHK(c, hipGetLastError());
    double h[8];
    KCHK(wait_slots(c, d_tgt ? 8 : 4, h));
    c_src[0] = h[0]; c_src[1] = h[1]; c_src[2] = h[2];
    *r_src = h[3] / (double)ns;
    if (d_tgt) {
        c_tgt[0] = h[4]; c_tgt[1] = h[5]; c_tgt[2] = h[6];
        *r_tgt = h[7] / (double)nt;
    }
    return KSS_OK;
}

int kss_preshape_stats_dev(kss_ctx* c, const void* d_xyz, int dtype, int64_t n, double centroid[3], double* mean_radius) {
    if (!c || !d_xyz || !centroid || !mean_radius) return set_err(c, KSS_ERR_ARG, "preshape: null argument");
    return kss_preshape_stats_pair_dev(c, d_xyz, n, nullptr, 0, dtype, centroid, mean_radius, nullptr, nullptr);
}

int kss_preshape_stats(kss_ctx* c, const void* xyz, int dtype, int64_t n, double centroid[3], double* mean_radius) {
    if (!c || !xyz) return set_err(c, KSS_ERR_ARG, "preshape: null argument");
    if (n <= 0) return set_err(c, KSS_ERR_ARG, "preshape: empty cloud");
    if (dtype != KSS_F32 && dtype != KSS_F64) return set_err(c, KSS_ERR_ARG, "preshape: bad dtype");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t esz = dtype == KSS_F64 ? 8 : 4;
    KCHK(upload(c, c->stage_src, xyz, (size_t)n * 3 * esz));
    return kss_preshape_stats_dev(c, c->stage_src.p, dtype, n, centroid, mean_radius);
}

// ---- pose / transform application -------------------------------------------------------------------
int kss_pose_apply_dev(kss_ctx* c, const double* d_in, int64_t n, const kss_pose* pose, double* d_out) {
    if (!c || !d_in || !d_out || !pose) return set_err(c, KSS_ERR_ARG, "pose_apply: null argument");
    if (n < 0) return set_err(c, KSS_ERR_ARG, "pose_apply: negative size");
    if (n == 0) return KSS_OK;
    HIPCHK(c, hipSetDevice(c->device));
    // cos/sin evaluated on the host so that they are the caller's libm values (reference: cos(angle) per point)
    const double cs[6] = {std::cos(pose->angle[0]), std::sin(pose->angle[0]), std::cos(pose->angle[1]),
                          std::sin(pose->angle[1]), std::cos(pose->angle[2]), std::sin(pose->angle[2])};
    {
        ProfScope ps(c, KSS_K_POSE_APPLY);
        launch_pose_apply(c->stream, d_in, n, *pose, cs, d_out);
    }
    HIPCHK(c, hipGetLastError());
    return KSS_OK;
}

int kss_pose_apply(kss_ctx* c, const double* in, int64_t n, const kss_pose* pose, double* out) {
    if (!c || !in || !out || !pose) return set_err(c, KSS_ERR_ARG, "pose_apply: null argument");
    if (n < 0) return set_err(c, KSS_ERR_ARG, "pose_apply: negative size");
    if (n == 0) return KSS_OK;
    HIPCHK(c, hipSetDevice(c->device));
    KCHK(upload(c, c->stage_src, in, (size_t)n * 3 * sizeof(double)));
    KCHK(ensure(c, c->stage_out, (size_t)n * 3 * sizeof(double)));
    KCHK(kss_pose_apply_dev(c, (const double*)c->stage_src.p, n, pose, (double*)c->stage_out.p));
    HIPCHK(c, hipMemcpyAsync(out, c->stage_out.p, (size_t)n * 3 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return KSS_OK;
}

int kss_transform_apply_dev(kss_ctx* c, const float T[16], const double* d_in, int64_t n, double* d_out) {
    if (!c || !T || !d_in || !d_out) return set_err(c, KSS_ERR_ARG, "transform_apply: null argument");
    if (n < 0) return set_err(c, KSS_ERR_ARG, "transform_apply: negative size");
    if (n == 0) return KSS_OK;
    HIPCHK(c, hipSetDevice(c->device));
    {
        ProfScope ps(c, KSS_K_POSE_APPLY);
        launch_transform_apply_f64(c->stream, T, d_in, n, d_out);
    }
    HIPCHK(c, hipGetLastError());
    return KSS_OK;
}

int kss_transform_apply(kss_ctx* c, const float T[16], const double* in, int64_t n, double* out) {
    if (!c || !T || !in || !out) return set_err(c, KSS_ERR_ARG, "transform_apply: null argument");
    if (n < 0) return set_err(c, KSS_ERR_ARG, "transform_apply: negative size");
    if (n == 0) return KSS_OK;
    HIPCHK(c, hipSetDevice(c->device));
    KCHK(upload(c, c->stage_src, in, (size_t)n * 3 * sizeof(double)));
    KCHK(ensure(c, c->stage_out, (size_t)n * 3 * sizeof(double)));
    KCHK(kss_transform_apply_dev(c, T, (const double*)c->stage_src.p, n, (double*)c->stage_out.p));
    HIPCHK(c, hipMemcpyAsync(out, c->stage_out.p, (size_t)n * 3 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return KSS_OK;
}

int kss_transform_apply_f32(kss_ctx* c, const float T[16], const float* in, int64_t n, float* out) {
    if (!c || !T || !in || !out) return set_err(c, KSS_ERR_ARG, "transform_apply_f32: null argument");
    if (n < 0) return set_err(c, KSS_ERR_ARG, "transform_apply_f32: negative size");
    if (n == 0) return KSS_OK;
    HIPCHK(c, hipSetDevice(c->device));
    KCHK(upload(c, c->stage_src, in, (size_t)n * 3 * sizeof(float)));
    KCHK(ensure(c, c->stage_out, (size_t)n * 3 * sizeof(float)));
    {
        ProfScope ps(c, KSS_K_POSE_APPLY);
        launch_transform_apply_f32(c->stream, T, (const float*)c->stage_src.p, n, (float*)c->stage_out.p);
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(out, c->stage_out.p, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return KSS_OK;
}

int kss_downsample_fps(kss_ctx* c, const double* xyz, int64_t n, int64_t m, double* out, int32_t* out_idx) {
    if (!c || !xyz || !out) return set_err(c, KSS_ERR_ARG, "downsample_fps: null argument");
    if (n <= 0 || m <= 0 || m > n || n > 0x7fff0000ll) return set_err(c, KSS_ERR_ARG, "downsample_fps: need 0 < m <= n");
    HIPCHK(c, hipSetDevice(c->device));
    KCHK(upload(c, c->scratch_a, xyz, (size_t)n * 3 * sizeof(double)));
    KCHK(ensure(c, c->scratch_b, (size_t)n * sizeof(double)));
    KCHK(ensure(c, c->stage_idx, (size_t)m * sizeof(int32_t)));
    KCHK(ensure(c, c->stage_out, (size_t)m * 3 * sizeof(double)));
    launch_fps(c->stream, (const double*)c->scratch_a.p, (int)n, (int)m, (double*)c->scratch_b.p, (int32_t*)c->stage_idx.p, (double*)c->stage_out.p);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(out, c->stage_out.p, (size_t)m * 3 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (out_idx) HIPCHK(c, hipMemcpyAsync(out_idx, c->stage_idx.p, (size_t)m * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return KSS_OK;
}

int kss_downsample_aivs(kss_ctx* c, const double* xyz, int64_t n, int64_t point_num, double* out, int64_t capacity,
                        int64_t* n_out, int32_t* out_idx) {
    if (!c || !xyz || !out || !n_out) return set_err(c, KSS_ERR_ARG, "downsample_aivs: null argument");
    if (n <= 0 || point_num <= 0 || n > 0x7fff0000ll) return set_err(c, KSS_ERR_ARG, "downsample_aivs: bad sizes");
    HIPCHK(c, hipSetDevice(c->device));
    KCHK(upload(c, c->scratch_a, xyz, (size_t)n * 3 * sizeof(double)));
    std::vector<int32_t> sel;
    std::string err;
    auto scratch = [c](size_t bytes) -> void* { return ensure(c, c->scratch_b, bytes) == KSS_OK ? c->scratch_b.p : nullptr; };
    const int rc = aivs_device(c->stream, (const double*)c->scratch_a.p, (int)n, (int)std::min<int64_t>(point_num, 0x7fffffff), sel, err, scratch);
    if (rc != KSS_OK) return set_err(c, rc, err.c_str());
    *n_out = (int64_t)sel.size();
    if ((int64_t)sel.size() > capacity) return set_err(c, KSS_ERR_CAPACITY, "downsample_aivs: output buffer too small");
    for (size_t i = 0; i < sel.size(); ++i) {
        const int32_t s = sel[i];
        out[3 * i] = xyz[3 * (size_t)s]; out[3 * i + 1] = xyz[3 * (size_t)s + 1]; out[3 * i + 2] = xyz[3 * (size_t)s + 2];
        if (out_idx) out_idx[i] = s;
    }
    return KSS_OK;
}

int kss_downsample_aivs_pair(kss_ctx* c, const double* xyz0, int64_t n0, int64_t point_num0, double* out0, int64_t capacity0, int64_t* n_out0,
                             int32_t* out_idx0, const double* xyz1, int64_t n1, int64_t point_num1, double* out1, int64_t capacity1,
                             int64_t* n_out1, int32_t* out_idx1, int rc[2]) {
    if (!c || !rc) return set_err(c, KSS_ERR_ARG, "downsample_aivs_pair: null argument");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->workers.empty()) {   // (worker contexts live as long as the parent: kss_register_batch shares them)
        kss_ctx* w = nullptr;
        const int wrc = kss_ctx_create(c->device, &w);
        if (wrc != KSS_OK) return set_err(c, wrc, "downsample_aivs_pair: cannot create a worker context");
        w->nn_mode = c->nn_mode;
        c->workers.push_back(w);
    }
    kss_ctx* w = c->workers[0];
    rc[0] = rc[1] = KSS_OK;
    // (a thread of its own: ~50 us per call; the context's pool -- woken through a condition variable -- was measured at 1.9 ms per pair of clouds against 0.65)
    std::thread other([&] { hipSetDevice(w->device); rc[1] = kss_downsample_aivs(w, xyz1, n1, point_num1, out1, capacity1, n_out1, out_idx1); });
    rc[0] = kss_downsample_aivs(c, xyz0, n0, point_num0, out0, capacity0, n_out0, out_idx0);
    other.join();
    if (rc[1] != KSS_OK && rc[0] == KSS_OK) c->err = w->err;   // (kss_last_error(ctx) then speaks of the cloud that failed)
    return KSS_OK;
}

// ---- octree down-sampler (Method_Octree.hpp:77-165) ----------------------------------------------------
int kss_downsample_octree(kss_ctx* c, const double* xyz, int64_t n, int32_t* out_idx, int64_t capacity, int64_t* n_out,
                          double* resolution_out) {
    if (!c || !xyz || !out_idx || !n_out) return set_err(c, KSS_ERR_ARG, "downsample_octree: null argument");
    if (n < 1000 || n > 0x7fff0000ll) return set_err(c, KSS_ERR_ARG, "downsample_octree: needs 1000 <= n < 2^31 points (the reference reads the first 1000 unconditionally)");
    HIPCHK(c, hipSetDevice(c->device));
    // the float cloud (cloud_i.x = pData[i][0]) on the host (box replay) and on the device
    std::vector<float> pf((size_t)n * 3);
    for (size_t i = 0; i < pf.size(); ++i) pf[i] = (float)xyz[i];
    KCHK(upload(c, c->oct_pts, pf.data(), pf.size() * sizeof(float)));
    // PCL_Octree_Resolution (:151-165) -> PCL_Octree_Estimate_Radius (:110-149): first 1000 points, kn-th neighbour
    int kn;
    if (n < 80000) kn = 2;
    else { const int md = (int)(n / 80000); kn = md >= 5 ? 35 : 7 * md; }
    KCHK(ensure(c, c->stage_idx, (size_t)1000 * kn * sizeof(int32_t)));
    KCHK(ensure(c, c->stage_d2, (size_t)1000 * kn * sizeof(float)));
    KCHK(knn_generic_dev(c, c->oct_pts.p, 1000, c->oct_pts.p, n, KSS_F32, kn, (int32_t*)c->stage_idx.p, (float*)c->stage_d2.p));
    std::vector<float> kd((size_t)1000 * kn);
    HIPCHK(c, hipMemcpyAsync(kd.data(), c->stage_d2.p, kd.size() * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    double radiusSum = 0;
    for (int i = 0; i < 1000; ++i) radiusSum = radiusSum + std::sqrt((double)kd[(size_t)i * kn + kn - 1]);
    radiusSum = radiusSum / 1000;
    const float resolution = (float)radiusSum;
    if (resolution_out) *resolution_out = (double)resolution;
    if (!(resolution > 0.f)) return set_err(c, KSS_ERR_ARG, "downsample_octree: zero resolution (coincident points)");
    // addPointsFromInputCloud: PCL's bounding cube, grown in insertion order
    OctBox box;
    std::memset(&box, 0, sizeof box);
    box.res = (double)resolution;
    oct_first_point(box, pf.data());
    for (int64_t i = 0; i < n; ++i)
        if (!oct_adopt(box, pf.data() + 3 * i)) return set_err(c, KSS_ERR_ARG, "downsample_octree: octree deeper than 21 levels");
    // occupied voxels in depth-first order -> centres
    KCHK(ensure(c, c->oct_cen, (size_t)n * 3 * sizeof(float)));
    std::string err;
    DevBuf* bufs[4] = {&c->oct_a, &c->oct_b, &c->scratch_c, &c->oct_tmp};
    auto scratch = [c, &bufs](int which, size_t bytes) -> void* { return ensure(c, *bufs[which], bytes) == KSS_OK ? bufs[which]->p : nullptr; };
    int m = 0;
    const int rc = octree_voxels_device(c->stream, (const float*)c->oct_pts.p, (int)n, box, (float*)c->oct_cen.p, &m, err, scratch);
    if (rc != KSS_OK) return set_err(c, rc, err.c_str());
    *n_out = m;
    if (m > capacity) return set_err(c, KSS_ERR_CAPACITY, "downsample_octree: output buffer too small");
    // octree.nearestKSearch(centre, 1): the exact NN engine (ties -> lowest index)
    KCHK(ensure(c, c->stage_idx, (size_t)m * sizeof(int32_t)));
    KCHK(ensure(c, c->stage_d2, (size_t)m * sizeof(float)));
    KCHK(kss_nn_dev(c, (const float*)c->oct_cen.p, m, (const float*)c->oct_pts.p, n, (int32_t*)c->stage_idx.p, (float*)c->stage_d2.p));
    HIPCHK(c, hipMemcpyAsync(out_idx, c->stage_idx.p, (size_t)m * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return KSS_OK;
}

// ---- rotation search --------------------------------------------------------------------------------
int kss_grid_angles(double step, double* angles, int capacity) {
    if (!angles || capacity <= 0 || !(step > 0)) return KSS_ERR_ARG;
    const int g = grid_angles(step, angles, capacity);
    return g < 0 ? KSS_ERR_CAPACITY : g;
}

int kss_rotation_candidates(const double* err, int g, double step, double best_angle[3], double* angle_list,
                            int list_capacity, int* n_list) {
    if (!err || g <= 0 || !best_angle || !n_list || !(step > 0)) return KSS_ERR_ARG;
    std::vector<double> ang(g);
    if (grid_angles(step, ang.data(), g) != g) return KSS_ERR_ARG;
    // global arg-min, strict '<' against errorT = 9999 in i, j, k loop order (:239, :258-265)
    double errorT = 9999;
    double bi = 0, bj = 0, bk = 0;
    for (int i = 0; i < g; ++i)
        for (int j = 0; j < g; ++j)
            for (int k = 0; k < g; ++k) {
                const double e = err[((int64_t)i * g + j) * g + k];
                if (e < errorT) { errorT = e; bi = ang[i]; bj = ang[j]; bk = ang[k]; }
            }
    best_angle[0] = bi; best_angle[1] = bj; best_angle[2] = bk;
    int nl = 0;
    for (int i = 0; i < g; ++i)
        for (int j = 0; j < g; ++j)
            for (int k = 0; k < g; ++k)
                if (is_local_min(err, g, i, j, k, 2)) {
                    if (angle_list) {
                        if (nl >= list_capacity) return KSS_ERR_CAPACITY;
                        angle_list[3 * nl + 0] = (double)i * 6.3 / step;   // :282-284
                        angle_list[3 * nl + 1] = (double)j * 6.3 / step;
                        angle_list[3 * nl + 2] = (double)k * 6.3 / step;
                    }
                    ++nl;
                }
    *n_list = nl;
    return KSS_OK;
}

int kss_rotation_search_dev(kss_ctx* c, const double* d_src, int64_t ns, const double* d_tgt, int64_t nt, double step,
                            double* err, int64_t err_capacity, int* g_out) {
    if (!c || !d_src || !d_tgt || !err || !g_out) return set_err(c, KSS_ERR_ARG, "rotation_search: null argument");
    if (ns <= 0 || nt <= 0 || !(step > 0)) return set_err(c, KSS_ERR_ARG, "rotation_search: empty cloud or bad step");
    if (ns > 0x7fff0000ll || nt > 0x7fff0000ll) return set_err(c, KSS_ERR_ARG, "rotation_search: cloud too large");
    HIPCHK(c, hipSetDevice(c->device));
    double ang[64];
    const int g = grid_angles(step, ang, 40);
    if (g <= 0) return set_err(c, KSS_ERR_ARG, "rotation_search: grid larger than 40 per axis");
    const int64_t ncand = (int64_t)g * g * g;
    if (err_capacity < ncand) return set_err(c, KSS_ERR_CAPACITY, "rotation_search: err buffer too small");
    std::vector<double> cs(2 * g);
    for (int a = 0; a < g; ++a) { cs[2 * a] = std::cos(ang[a]); cs[2 * a + 1] = std::sin(ang[a]); }
    const int64_t nt_pad = (nt + NN_TILE - 1) / NN_TILE * NN_TILE;
    const int nth = rot_search_grain(ns, nt, g);
    const int nsb = (int)((ns + nth - 1) / nth);
    const int64_t nt_sweep = (nt + nth - 1) / nth * nth;   // what the search sweeps (<= nt_pad: the rest of the padding is never read)
    KCHK(ensure(c, c->tgt4, (size_t)nt_pad * sizeof(float4)));
    KCHK(ensure(c, c->cs, cs.size() * sizeof(double)));
    KCHK(ensure(c, c->partials, (size_t)ncand * nsb * sizeof(double)));
    KCHK(ensure(c, c->scratch_c, (size_t)ncand * sizeof(double)));
    HIPCHK(c, hipMemcpyAsync(c->cs.p, cs.data(), cs.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    launch_pack_f64_to_f4(c->stream, d_tgt, nt, (float4*)c->tgt4.p, nt_pad, true);   // :232-234 narrowing to PointXYZ
    {
        ProfScope ps(c, KSS_K_ROT_SEARCH);
        launch_rot_search(c->stream, d_src, ns, (const float4*)c->tgt4.p, nt_sweep, (const double*)c->cs.p, g,
                          (double*)c->partials.p, nsb, nth);
        launch_row_sums(c->stream, (const double*)c->partials.p, (int)ncand, nsb, 1.0, (double*)c->scratch_c.p);
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(err, c->scratch_c.p, (size_t)ncand * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int64_t i = 0; i < ncand; ++i) err[i] = err[i] / (double)ns;   // :448 distanceSum / size
    *g_out = g;
    return KSS_OK;
}

int kss_rotation_search(kss_ctx* c, const double* src, int64_t ns, const double* tgt, int64_t nt, double step,
                        double* err, int64_t err_capacity, int* g_out) {
    if (!c || !src || !tgt) return set_err(c, KSS_ERR_ARG, "rotation_search: null argument");
    if (ns <= 0 || nt <= 0) return set_err(c, KSS_ERR_ARG, "rotation_search: empty cloud");
    HIPCHK(c, hipSetDevice(c->device));
    KCHK(upload(c, c->scratch_a, src, (size_t)ns * 3 * sizeof(double)));
    KCHK(upload(c, c->scratch_b, tgt, (size_t)nt * 3 * sizeof(double)));
    return kss_rotation_search_dev(c, (const double*)c->scratch_a.p, ns, (const double*)c->scratch_b.p, nt, step, err, err_capacity, g_out);
}

// ---- PCR_QM -------------------------------------------------------------------------------------------
int kss_pcr_qm(kss_ctx* c, const double* aligned, int64_t na, const double* tmpl, int64_t nt, double out[3]) {
    if (!c || !aligned || !tmpl || !out) return set_err(c, KSS_ERR_ARG, "pcr_qm: null argument");
    if (na <= 0 || nt <= 0) return set_err(c, KSS_ERR_ARG, "pcr_qm: empty cloud");
    HIPCHK(c, hipSetDevice(c->device));
    KCHK(upload(c, c->scratch_a, aligned, (size_t)na * 3 * sizeof(double)));
    KCHK(upload(c, c->scratch_b, tmpl, (size_t)nt * 3 * sizeof(double)));
    double sums[NSUMS];
    KCHK(nn_generic_dev(c, c->scratch_a.p, na, c->scratch_b.p, nt, KSS_F64, nullptr, nullptr, sums));
    const double mse = sums[17] / (double)na;      // registrationMeasure.hpp:85
    out[0] = mse;
    out[1] = std::sqrt(mse);                        // :87
    out[2] = sums[18] / (double)na;                 // :86
    return KSS_OK;
}

// ---- KSSICP_Registration on down-sampled clouds ----------------------------------------------------------
int kss_register(kss_ctx* c, const double* src_sub, int64_t nss, const double* tgt_sub, int64_t nts,
                 const double* src_full, int64_t nsf, double accurate, int iter, double* point_align,
                 kss_register_result* res) {
    if (!c || !src_sub || !tgt_sub || !res) return set_err(c, KSS_ERR_ARG, "register: null argument");
    if (nss <= 0 || nts <= 0 || nsf < 0 || (nsf > 0 && !src_full)) return set_err(c, KSS_ERR_ARG, "register: bad sizes");
    HIPCHK(c, hipSetDevice(c->device));
    std::memset(res, 0, sizeof *res);
    const bool reg_timing = getenv("KSS_TIMING") != nullptr;   // phase report on stderr (wall clock; the phases that enqueue only are charged to the next one that waits)
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!reg_timing) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[kss] register: %-34s %8.1f us\n", what, std::chrono::duration<double, std::micro>(now - t_prev).count());
        t_prev = now;
    };
    // resident copies of S', T' (f64) in the context's grow-only workspace (no hipMalloc / hipFree per registration:
    // each costs tens of microseconds and hipFree synchronises the device)
    DevBuf &dS = c->reg_s, &dT = c->reg_t, &dP = c->reg_p, &dAll = c->reg_all;
#define RCHK(expr) do { int rc_ = (expr); if (rc_ != KSS_OK) return rc_; } while (0)
    RCHK(upload(c, dS, src_sub, (size_t)nss * 3 * sizeof(double)));
    RCHK(upload(c, dT, tgt_sub, (size_t)nts * 3 * sizeof(double)));
    RCHK(ensure(c, dP, (size_t)nss * 3 * sizeof(double)));
    lap("uploads");
    // (a2) pre-shape
    double cS[3], cT[3], rS, rT;
    RCHK(kss_preshape_stats_pair_dev(c, dS.p, nss, dT.p, nts, KSS_F64, cS, &rS, cT, &rT));
    kss_pose pose;
    for (int k = 0; k < 3; ++k) { pose.shift[k] = cT[k] - cS[k]; pose.center[k] = cT[k]; pose.angle[k] = 0.0; }
    pose.scale = rT / rS;
    res->scale = pose.scale;
    for (int k = 0; k < 3; ++k) { res->c_src[k] = cS[k]; res->c_tgt[k] = cT[k]; }
    RCHK(kss_pose_apply_dev(c, (const double*)dS.p, nss, &pose, (double*)dP.p));   // S' (angle 0 -> exact identity rotation)
    lap("pre-shape statistics + pose");
    // (a4) rotation search
    std::vector<double> err(40 * 40 * 40);
    int g = 0;
    RCHK(kss_rotation_search_dev(c, (const double*)dP.p, nss, (const double*)dT.p, nts, accurate, err.data(), (int64_t)err.size(), &g));
    res->grid = g;
    lap("rotation search");
    std::vector<double> alist((size_t)3 * g * g * g);
    double best[3];
    int nl = 0;
    RCHK(kss_rotation_candidates(err.data(), g, accurate, best, alist.data(), g * g * g, &nl));
    res->n_angle_list = nl;
    lap("candidate list (host)");

    kss_icp_params ip;
    kss_icp_default_params(&ip);
    ip.max_iterations = iter;
    const int64_t to[2] = {0, nts};
    double chosen[3] = {best[0], best[1], best[2]};
    kss_icp_result r0, rfinal;
    std::vector<kss_icp_result> rr;
    bool have_judge = false, have_cands = false;
    // The judge ICP (from the best grid pose, KSS_ICP.hpp:92-93) and the candidate ICPs of the angle list (:102-118) do not
    // depend on each other -- only whether the candidates' results are USED depends on the judge's fitness (:99).  They run as
    // ONE batch sharing the target, the judge as its last pair; when the judge ends at or below the threshold the candidates
    // are told to stop where they are (the reference would not have run them; nothing of theirs is used).  Each pair is an
    // independent registration: the results equal those of the sequential calls.  KSS_REGISTER_SPEC=0, or an engine that
    // cannot cancel (no large BAR, a batch that does not fit the device at once): the sequential route below.
    static const bool want_spec = getenv("KSS_REGISTER_SPEC") == nullptr || atoi(getenv("KSS_REGISTER_SPEC")) != 0;
    auto fill_poses = [&](int n_list, bool with_judge) -> int {   // every candidate pose of S (and the judge's) in one launch
        const int count = n_list + (with_judge ? 1 : 0);
        RCHK(ensure(c, dAll, (size_t)count * nss * 3 * sizeof(double)));
        std::vector<double> cs((size_t)count * 6);
        for (int i = 0; i < count; ++i) {
            const double* ang = i < n_list ? &alist[3 * i] : best;
            // cos/sin evaluated on the host so that they are the caller's libm values (as kss_pose_apply_dev)
            for (int k = 0; k < 3; ++k) { cs[6 * i + 2 * k] = std::cos(ang[k]); cs[6 * i + 2 * k + 1] = std::sin(ang[k]); }
        }
        {
            ProfScope ps(c, KSS_K_POSE_APPLY);
            launch_pose_apply_many(c->stream, (const double*)dS.p, nss, pose, reinterpret_cast<const double (*)[6]>(cs.data()), count, (double*)dAll.p);
        }
        HIPCHK(c, hipGetLastError());
        return KSS_OK;
    };
    if (want_spec && nl > 0) {
        RCHK(fill_poses(nl, true));
        lap("candidate poses (enqueue)");
        std::vector<int64_t> so(nl + 2);
        for (int i = 0; i <= nl + 1; ++i) so[i] = (int64_t)i * nss;
        rr.resize(nl + 1);
        c->spec_judge = nl; c->spec_threshold = 0.0005; c->spec_ran = false; c->spec_cancelled = false;
        const int rc = icp_run_dev(c, dAll.p, so.data(), dT.p, to, nl + 1, true, KSS_F64, &ip, rr.data());
        c->spec_judge = -1;
        if (rc != KSS_OK) return rc;
        if (c->spec_ran) {
            r0 = rr[nl];
            have_judge = true;
            have_cands = !c->spec_cancelled;
        }
    }
    if (!have_judge) {
        // (a14) judge: ICP from the best grid pose (KSS_ICP.hpp:92-93)
        for (int k = 0; k < 3; ++k) pose.angle[k] = best[k];
        RCHK(kss_pose_apply_dev(c, (const double*)dS.p, nss, &pose, (double*)dP.p));
        const int64_t so[2] = {0, nss};
        RCHK(icp_run_dev(c, dP.p, so, dT.p, to, 1, false, KSS_F64, &ip, &r0));
    }
    res->E_d_init = r0.fitness;
    rfinal = r0;
    // the judge ICP *is* the final ICP when the threshold branch is not taken (same inputs, :93 vs :130)
    if (res->E_d_init > 0.0005 && nl > 0) {   // :99
        // (a14) all candidate ICPs as ONE batch sharing the target (:102-118)
        if (!have_cands) {
            RCHK(fill_poses(nl, false));
            std::vector<int64_t> so(nl + 1);
            for (int i = 0; i <= nl; ++i) so[i] = (int64_t)i * nss;
            rr.resize(nl);
            RCHK(icp_run_dev(c, dAll.p, so.data(), dT.p, to, nl, true, KSS_F64, &ip, rr.data()));
        }
        if (getenv("KSS_DEBUG_CANDS")) {   // every candidate's record (a diagnostic of run-to-run differences)
            char head[64]; std::snprintf(head, sizeof head, "[kss] ctx %p candidates:", (void*)c);
            std::string line = head;
            char buf[96];
            for (int i = 0; i < nl; ++i) { std::snprintf(buf, sizeof buf, " %d:(%d,%d,%.17g)", i, rr[i].iterations, rr[i].state, rr[i].fitness); line += buf; }
            std::fprintf(stderr, "%s | judge (%d,%d,%.17g) spec %d\n", line.c_str(), r0.iterations, r0.state, r0.fitness, have_cands ? 1 : 0);
        }
        double Q = 9999; int angleIndex = 0;
        for (int i = 0; i < nl; ++i) {
            const double ri = rr[i].fitness;
            if (ri < Q && ri >= 0) { Q = ri; angleIndex = i; }   // :113-116
        }
        res->used_angle_list = 1; res->angle_index = angleIndex;
        for (int k = 0; k < 3; ++k) chosen[k] = alist[3 * angleIndex + k];
        rfinal = rr[angleIndex];   // identical inputs => identical to re-running :130
    } else if (res->E_d_init > 0.0005) {
        res->used_angle_list = 1;  // empty angle list: the reference would index out of range; keep the best grid pose
    }
    lap("judge + candidate ICPs");
    for (int k = 0; k < 3; ++k) res->angle[k] = chosen[k];
    res->final_fitness = rfinal.fitness;
    res->icp_iterations = rfinal.iterations;
    res->icp_converged = rfinal.converged;
    std::memcpy(res->T_icp, rfinal.T, sizeof rfinal.T);
    // composite similarity: R = R_icp R0, t = R (c_T - s c_S) + t_icp
    double R0[9];
    euler_matrix(chosen, R0);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            res->R[3 * i + j] = (double)rfinal.T[4 * i] * R0[j] + (double)rfinal.T[4 * i + 1] * R0[3 + j] + (double)rfinal.T[4 * i + 2] * R0[6 + j];
    double v[3];
    for (int k = 0; k < 3; ++k) v[k] = cT[k] - pose.scale * cS[k];
    for (int i = 0; i < 3; ++i)
        res->t[i] = res->R[3 * i] * v[0] + res->R[3 * i + 1] * v[1] + res->R[3 * i + 2] * v[2] + (double)rfinal.T[4 * i + 3];
    // (a13) pointAlign = M * Rotation_Angle(pointSource) (:120/:124, :224-230)
    if (point_align && nsf > 0) {
        DevBuf &dF = c->reg_f, &dG = c->reg_g;
        int rc = upload(c, dF, src_full, (size_t)nsf * 3 * sizeof(double));
        if (rc == KSS_OK) rc = ensure(c, dG, (size_t)nsf * 3 * sizeof(double));
        for (int k = 0; k < 3; ++k) pose.angle[k] = chosen[k];
        if (rc == KSS_OK) rc = kss_pose_apply_dev(c, (const double*)dF.p, nsf, &pose, (double*)dG.p);
        if (rc == KSS_OK) rc = kss_transform_apply_dev(c, rfinal.T, (const double*)dG.p, nsf, (double*)dF.p);
        if (rc == KSS_OK && hipMemcpyAsync(point_align, dF.p, (size_t)nsf * 3 * sizeof(double), hipMemcpyDeviceToHost, c->stream) != hipSuccess) rc = KSS_ERR_HIP;
        if (rc == KSS_OK && hipStreamSynchronize(c->stream) != hipSuccess) rc = KSS_ERR_HIP;
        if (rc != KSS_OK) return rc;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    lap("transform of the full cloud + copy");
#undef RCHK
    return KSS_OK;
}

// ---- many full registrations, concurrently ----------------------------------------------------------------------
static int downsample_for_register(kss_ctx* w, const double* xyz, int64_t n, int64_t m, std::vector<double>& out) {
    out.resize((size_t)n * 3);
    if (m <= 0 || m >= n) { std::memcpy(out.data(), xyz, (size_t)n * 3 * sizeof(double)); return KSS_OK; }
    int64_t k = 0;
    int rc = kss_downsample_aivs(w, xyz, n, m, out.data(), n, &k, nullptr);
    if (rc == KSS_ERR_ARG) {   // degenerate extent (the reference would divide by zero): exact farthest-point sampling
        rc = kss_downsample_fps(w, xyz, n, m, out.data(), nullptr);
        k = m;
    }
    if (rc != KSS_OK) return rc;
    out.resize((size_t)k * 3);
    return KSS_OK;
}

int kss_register_batch(kss_ctx* c, const double* src_all, const int64_t* src_off, const double* tgt_all, const int64_t* tgt_off,
                       int npairs, int64_t sample_cap, double accurate, int iter, int workers, double* point_align_all,
                       kss_register_result* results) {
    if (!c || !src_all || !src_off || !tgt_all || !tgt_off || !results || npairs <= 0) return set_err(c, KSS_ERR_ARG, "register_batch: bad argument");
    for (int i = 0; i < npairs; ++i)
        if (src_off[i + 1] <= src_off[i] || tgt_off[i + 1] <= tgt_off[i]) return set_err(c, KSS_ERR_ARG, "register_batch: empty cloud");
    HIPCHK(c, hipSetDevice(c->device));
    int nw = workers > 0 ? workers : 8;
    nw = std::min(std::min(nw, npairs), 32);
    while ((int)c->workers.size() < nw) {   // worker contexts live as long as the parent
        kss_ctx* w = nullptr;
        const int rc = kss_ctx_create(c->device, &w);
        if (rc != KSS_OK) return set_err(c, rc, "register_batch: cannot create a worker context");
        w->nn_mode = c->nn_mode;
        c->workers.push_back(w);
    }
    std::atomic<int> next{0}, failed{KSS_OK};
    std::string first_error;
    std::mutex err_mutex;
    auto run = [&](kss_ctx* w) {
        hipSetDevice(w->device);
        std::vector<double> ssub, tsub;
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= npairs || failed.load() != KSS_OK) return;
            const double* S = src_all + 3 * src_off[i];
            const double* T = tgt_all + 3 * tgt_off[i];
            const int64_t ns = src_off[i + 1] - src_off[i], nt = tgt_off[i + 1] - tgt_off[i];
            int64_t pNumber = std::min(ns, nt) / 2;                      // KSS_ICP.hpp:57-63
            if (sample_cap > 0 && pNumber > sample_cap) pNumber = sample_cap;
            int rc = downsample_for_register(w, T, nt, pNumber, tsub);    // :71-75 (target first, as the reference)
            if (rc == KSS_OK) rc = downsample_for_register(w, S, ns, pNumber, ssub);
            if (rc == KSS_OK)
                rc = kss_register(w, ssub.data(), (int64_t)ssub.size() / 3, tsub.data(), (int64_t)tsub.size() / 3, S, ns, accurate, iter,
                                  point_align_all ? point_align_all + 3 * src_off[i] : nullptr, &results[i]);
            if (rc != KSS_OK) {
                std::lock_guard<std::mutex> lk(err_mutex);
                if (failed.load() == KSS_OK) { failed.store(rc); first_error = "register_batch: pair " + std::to_string(i) + ": " + w->err; }
                return;
            }
        }
    };
    std::vector<std::thread> threads;
    for (int k = 1; k < nw; ++k) threads.emplace_back(run, c->workers[(size_t)k]);
    run(c->workers[0]);
    for (std::thread& t : threads) t.join();
    if (failed.load() != KSS_OK) return set_err(c, failed.load(), first_error.c_str());
    return KSS_OK;
}

// ---- RCCL gather of result records ------------------------------------------------------------------------
// ncclAllGather is resolved at run time from librccl.so so that libkssicp.so has no link-time RCCL
// dependency (single-GPU users never load it).
int kss_rccl_allreduce_sum(void* user, double* values, int n) {
    kss_rccl_link* L = (kss_rccl_link*)user;
    if (!L || !L->ctx || !L->rccl_comm || !values || n <= 0) return KSS_ERR_ARG;
    kss_ctx* c = L->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    typedef int (*allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
    static allreduce_fn fn = nullptr;
    if (!fn) {
        void* h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return set_err(c, KSS_ERR_RCCL, "cannot dlopen librccl.so");
        fn = (allreduce_fn)dlsym(h, "ncclAllReduce");
        if (!fn) return set_err(c, KSS_ERR_RCCL, "ncclAllReduce not found in librccl.so");
    }
    const size_t bytes = (size_t)n * sizeof(double);
    KCHK(ensure(c, c->sums, std::max<size_t>(bytes, NSUMS * sizeof(double))));
    HIPCHK(c, hipMemcpyAsync(c->sums.p, values, bytes, hipMemcpyHostToDevice, c->stream));
    const int rc = fn(c->sums.p, c->sums.p, (size_t)n, /*ncclFloat64*/ 8, /*ncclSum*/ 0, L->rccl_comm, c->stream);
    if (rc != 0) return set_err(c, KSS_ERR_RCCL, "ncclAllReduce failed");
    HIPCHK(c, hipMemcpyAsync(values, c->sums.p, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return KSS_OK;
}

int kss_gather_results(kss_ctx* c, void* rccl_comm, int world_size, const kss_icp_result* local, int n_local,
                       kss_icp_result* all) {
    if (!c || !rccl_comm || !local || !all || n_local <= 0 || world_size <= 0) return set_err(c, KSS_ERR_ARG, "gather: bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    typedef int (*allgather_fn)(const void*, void*, size_t, int, void*, hipStream_t);
    static allgather_fn fn = nullptr;
    if (!fn) {
        void* h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return set_err(c, KSS_ERR_RCCL, "cannot dlopen librccl.so");
        fn = (allgather_fn)dlsym(h, "ncclAllGather");
        if (!fn) return set_err(c, KSS_ERR_RCCL, "ncclAllGather not found in librccl.so");
    }
    const size_t bytes = (size_t)n_local * sizeof(kss_icp_result);
    KCHK(ensure(c, c->scratch_a, bytes));
    KCHK(ensure(c, c->scratch_b, bytes * (size_t)world_size));
    HIPCHK(c, hipMemcpyAsync(c->scratch_a.p, local, bytes, hipMemcpyHostToDevice, c->stream));
    const int rc = fn(c->scratch_a.p, c->scratch_b.p, bytes, /*ncclInt8*/ 0, rccl_comm, c->stream);
    if (rc != 0) return set_err(c, KSS_ERR_RCCL, "ncclAllGather failed");
    HIPCHK(c, hipMemcpyAsync(all, c->scratch_b.p, bytes * (size_t)world_size, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return KSS_OK;
}

}  // extern "C"
