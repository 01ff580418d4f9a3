// kss_ctx.hpp -- the context behind the opaque kss_ctx of include/kssicp.h, and the small helpers every host-side
// translation unit of libkssicp.so uses (error reporting, grow-only device buffers, pinned staging, HIP-event timing).
// Internal to the library.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "kss_internal.hpp"
#include "kss_host_pool.hpp"

using namespace kss;

// ---------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

struct kss_ctx {
    int device = 0;
    int cu_count = 0;                   // compute units of the device (0: not asked yet)
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;
    int nn_mode = KSS_NN_AUTO;
    double grid_stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    double last_setup_ms = 0, last_loop_ms = 0;
    double t_launch_us = 0, t_wait_us = 0, t_host_us = 0;   // KSS_TIMING breakdown of the fused single-pair loop
    bool timing = false;
    bool tables_staged = false;
    int64_t stats_ns = -1, stats_nt = -1;

    // grow-only device workspace
    DevBuf tgt4, src0, cur[2], keys, partials, sums, nn_work, red_work, pair_red, state, cs, scratch_a,
        scratch_b, scratch_c, stage_src, stage_tgt, stage_idx, stage_d2, stage_out, g_counts, g_start, g_cursor,
        g_bsums, g_sorted, g_list, g_count, g_bbox, g_partials, g_start2, g_pairs, g_stamps, g_pos, g_nnst, res_pos, res_wc, res_perm, cand_tags, pack_seg, reg_s, reg_t, reg_p, reg_all, reg_f, reg_g, oct_pts, oct_cen, oct_a, oct_b, oct_tmp, pair_ticket, pre_partials, pre_state, g_rowpair, g_gate;
    HostPool pool;   // per-pair host work of batched iterations
    std::vector<kss_ctx*> workers;   // contexts of kss_register_batch's worker threads (same device, own streams)
    std::vector<unsigned long long> last_stamps;
    int stamps_nblk = 0; unsigned long long stamps_seq = 0;   // KSS_GRID_STAMPS=2
    std::vector<float> h_bbox;   // bbox partials of the last single-pair target (host copy)
    unsigned int* h_box = nullptr; unsigned int* h_box_dev = nullptr; size_t h_box_bytes = 0;   // host-mapped: the same partials as checked granules, written by the pack kernel
    unsigned box_tag = 0; int box_rows = 0;   // tag and row count of the partials the last pack launch publishes there (0: none)
    double evals_sum = 0.0, evals_launches = 0.0;   // diagnostic runs: distance evaluations of the fused grid launches
    // pinned host staging
    void* h_sums = nullptr;  size_t h_sums_cap = 0;   // host-mapped: kernels write it through h_sums_dev
    void* h_sums_dev = nullptr;
    unsigned long long* h_seq = nullptr;       // host-mapped result of the fused grid kernel: NSUMS x {bits(sum), sequence number}
    unsigned long long* h_seq_dev = nullptr;
    size_t h_seq_bytes = 0;
    // gated launches (kss_engine.hip): the next iteration's fused kernel is enqueued while the current one runs and polls
    // the host-mapped record h_xf[slot]; the host answers it with the transform and the launch's stamp
    struct Gated {
        int supported = -1;                 // -1 unknown, 0 no, 1 yes
        bool want_next = false;             // set by the ICP loop: another regular iteration may follow this pass
        bool pending = false;               // a pre-enqueued launch is waiting at its gate ...
        bool chain = false;                 // ... it is a chained launch (more than one pass) ...
        int steps_left = 0;                 // ... with this many passes still to be released (a chained launch runs several)
        int max_steps = 1;                  // set by the ICP loop: regular iterations that may still follow this pass
        int slot = 0;
        unsigned long long seq = 0;
        int32_t stamp = 0;
        const void* d_in = nullptr; void* d_out = nullptr;
        bool fma = false, full = false, want_full = false; double max_d2 = 0.0;
    } gated;
    // pair_ticket, g_count and g_counts are ZERO AT REST: every kernel that uses them leaves them zeroed (tickets are re-armed
    // by the last workgroup, the unresolved-list length is reset when the list is consumed, the cell counts are counted
    // back down by the scatter), so a registration needs no memset of its own.  A call that fails midway sets ws_dirty
    // and the next one clears them first; ensure_zeroed() clears a buffer it had to (re)allocate.
    bool ws_dirty = false;
    bool defer_wait = false;   // batched fused pass: the ICP loop polls the pairs' result slots itself
    PairState* h_xf = nullptr; PairState* h_xf_dev = nullptr;
    bool fit_last = false;            // the fitness pass: PairState::pad[0] names the buffer of each pair's last pass
    bool nn_have = false;             // the cell-list pass has written nn_win / nn_state for the current lists
    PairState* bar_state = nullptr; int bar_state_cap = 0; bool bar_state_failed = false;   // batched pass: per-pair states, same kind of memory
    unsigned int* gate_bar = nullptr;   // fine-grained device memory the host stores into through the BAR (large-BAR systems)
    unsigned int* res_gate = nullptr; int res_gate_cap = 0; bool res_gate_failed = false;   // pair-resident engine: one 128-byte gate record per pair, same kind of memory
    unsigned res_launches = 0;          // ... its launches so far (every launch has its own range of gate stamps)
    double res_passes = 0.0;            // ... (pair, pass) units its profiled launches ran
    int cand_cap = 0, cand_cap_pad = -1, cand_cap_fma = -1;   // candidate-resident kernel: workgroups resident at once for this target size
    // kss_register runs the judge ICP and the candidate ICPs as ONE speculative batch: pair spec_judge is the judge; once its
    // fitness is known and does not exceed spec_threshold the candidates are told to stop (their results are not used).
    // spec_ran: the batch ran that way (otherwise the caller takes the sequential route); spec_cancelled: candidates were stopped
    int spec_judge = -1; double spec_threshold = 0.0; bool spec_ran = false, spec_cancelled = false;
    // a resident launch that had to be given up (a workgroup nobody answered in time: another process on the GPU, a starved host)
    // keeps its engine off for this context for a while: the next calls go straight to the launch-per-pass form
    std::chrono::steady_clock::time_point res_backoff[2] = {};
    unsigned long long seq = 0;
    void* h_state = nullptr; size_t h_state_cap = 0;

    // profiling
    int prof = 0;                       // 0 = off, n = event-time every n-th launch of each kernel class
    unsigned prof_tick[KSS_K_COUNT] = {};
    struct EvPair { hipEvent_t a, b; };
    std::vector<EvPair> ev[KSS_K_COUNT];
    std::vector<EvPair> ev_pool;   // recycled event pairs: no hipEventCreate/Destroy inside timed loops
    double prof_ms[KSS_K_COUNT] = {0};
    int64_t prof_n[KSS_K_COUNT] = {0};
};

static inline int set_err(kss_ctx* c, int code, const char* what, hipError_t e = hipSuccess) {
    if (c) {
        c->err = what;
        if (e != hipSuccess) { c->err += ": "; c->err += hipGetErrorString(e); }
    }
    return code;
}

#define HIPCHK(ctx, call)                                                        \
    do {                                                                         \
        hipError_t e_ = (call);                                                  \
        if (e_ != hipSuccess) return set_err((ctx), KSS_ERR_HIP, #call, e_);     \
    } while (0)

static inline int ensure(kss_ctx* c, DevBuf& b, size_t bytes) {
    if (bytes <= b.cap) return KSS_OK;
    if (b.p) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipFree(b.p));
        b.p = nullptr; b.cap = 0;
    }
    size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) { b.p = nullptr; return set_err(c, KSS_ERR_NOMEM, "hipMalloc", e); }
    b.cap = want;
    return KSS_OK;
}

static inline int ensure_zeroed(kss_ctx* c, DevBuf& b, size_t bytes) {
    const size_t before = b.cap;   // (not the pointer: a freed block can come back at the same address with a larger size)
    const int rc = ensure(c, b, bytes);
    if (rc != KSS_OK) return rc;
    if (b.cap != before && hipMemsetAsync(b.p, 0, b.cap, c->stream) != hipSuccess) return set_err(c, KSS_ERR_HIP, "hipMemsetAsync(zero-at-rest buffer)");
    return KSS_OK;
}

static inline int ensure_pinned(kss_ctx* c, void*& p, size_t& cap, size_t bytes) {
    if (bytes <= cap) return KSS_OK;
    if (p) { HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipHostFree(p)); p = nullptr; cap = 0; }
    size_t want = bytes * 2 + 256;
    hipError_t e = hipHostMalloc(&p, want, hipHostMallocMapped);
    if (e != hipSuccess) { p = nullptr; return set_err(c, KSS_ERR_NOMEM, "hipHostMalloc", e); }
    cap = want;
    if (&p == &c->h_sums) {
        void* d = nullptr;
        HIPCHK(c, hipHostGetDevicePointer(&d, p, 0));
        c->h_sums_dev = d;
    }
    return KSS_OK;
}

#define KCHK(expr)                     \
    do {                               \
        int rc_ = (expr);              \
        if (rc_ != KSS_OK) return rc_; \
    } while (0)

struct ProfScope {   // records a start/stop event pair around a launch when profiling is on
    kss_ctx* c; int k; kss_ctx::EvPair ep; bool on;
    // every: bracket this launch whenever profiling is on (a chained launch runs tens of passes: its two events are cheap)
    ProfScope(kss_ctx* c_, int k_, bool every = false) : c(c_), k(k_), on(c_->prof > 0) {
        if (on && c->prof > 1 && !every) on = (c->prof_tick[k]++ % (unsigned)c->prof) == 0;   // sampled: the events themselves cost ~3 us
        if (!on) return;
        if (!c->ev_pool.empty()) {
            ep = c->ev_pool.back();
            c->ev_pool.pop_back();
        } else if (hipEventCreate(&ep.a) != hipSuccess || hipEventCreate(&ep.b) != hipSuccess) {
            on = false;
            return;
        }
        hipEventRecord(ep.a, c->stream);
    }
    ~ProfScope() {
        if (!on) return;
        hipEventRecord(ep.b, c->stream);
        c->ev[k].push_back(ep);
    }
};

static inline void prof_collect(kss_ctx* c) {
    for (int k = 0; k < KSS_K_COUNT; ++k) {
        for (auto& ep : c->ev[k]) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, ep.a, ep.b) == hipSuccess) {
                c->prof_ms[k] += ms; c->prof_n[k] += 1;
                if (k == KSS_K_GRID_CHAIN) c->prof_ms[KSS_K_GRID_CHAIN_PASS] += ms;   // (its count: the passes released, kss_engine.hip)
                if (k == KSS_K_RESIDENT) c->prof_ms[KSS_K_RESIDENT_PASS] += ms;       // (its count: the (pair, pass) units run)
            }
            c->ev_pool.push_back(ep);
        }
        c->ev[k].clear();
    }
}


// live contexts of this process (gated launches are used only while there is exactly one: see kss_engine.hip)
inline std::atomic<int>& kss_live_contexts() { static std::atomic<int> n{0}; return n; }

// host -> device staging of a packed cloud
static inline int upload(kss_ctx* c, DevBuf& b, const void* h, size_t bytes) {
    KCHK(ensure(c, b, bytes));
    HIPCHK(c, hipMemcpyAsync(b.p, h, bytes, hipMemcpyHostToDevice, c->stream));
    return KSS_OK;
}

// ---- the registration engine (kss_engine.hip) ---------------------------------------------------------------
namespace kss {
// packs the clouds of npairs pairs (device pointers, point offsets), builds the search structures and runs the ICP loop
int icp_run_dev(kss_ctx* c, const void* d_src, const int64_t* src_off, const void* d_tgt, const int64_t* tgt_off,
                int npairs, bool shared_target, int dtype, const kss_icp_params* p, kss_icp_result* results);
// one exact NN pass of a single pair (+ the correspondence sums when sums_out is given)
int nn_generic_dev(kss_ctx* c, const void* d_src, int64_t ns, const void* d_tgt, int64_t nt, int dtype,
                   int32_t* d_idx, float* d_d2, double sums_out[NSUMS]);
// host-mapped {value, sequence number} result slots: allocate them; wait for the first nslots of launch c->seq
int ensure_pub_slots(kss_ctx* c);
int wait_slots(kss_ctx* c, int nslots, double* out);
}  // namespace kss
