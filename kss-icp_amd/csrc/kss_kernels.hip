// kss_kernels.hip -- hand-written HIP kernels of the KSS-ICP registration core for gfx950 (CDNA4).
//
// Kernels (DESIGN.md has the roofline of each):
//   nn_sweep_kernel      (b) brute-force exact 1-NN: LDS-tiled source x target sweep, FP32 VALU bound
//   corr_reduce_kernel   (c) correspondence sums (3x3 cross-covariance etc.), f64, wave + LDS reduce
//   preshape_*           (a) KSS pre-shape centroid / mean radius, f64 block reduce, HBM streaming
//   rot_search_kernel        all g^3 Euler candidates x n(S') NN in one launch
//   pose_apply / transform_apply   streaming similarity / Matrix4f application in f64
//
// Arithmetic contract: everything that feeds a parity-checked result is evaluated WITHOUT fused
// multiply-add, in the reference's operation order (FLANN L2_Simple<float>, Eigen Matrix4f * vec4,
// initRegistration_Transfer in double).  The file is compiled with -ffp-contract=off and the pragma
// below; the optional FMA distance form uses __builtin_fmaf explicitly.
#pragma clang fp contract(off)

#include <hip/hip_runtime.h>

#include <cstdlib>

#include "kss_internal.hpp"
#include "kss_device.hpp"

namespace kss {

// ---------------------------------------------------------------------------------------------
// packing: xyz triples -> float4 (x, y, z, 0), tail filled with +inf sentinels so that padded
// targets can never win the arg-min.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void pack_to_f4_kernel(const T* __restrict__ in, int64_t n,
                                                         float4* __restrict__ out, int64_t n_pad, int sentinel) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pad) return;
    float4 v;
    if (i < n) {
        v = make_float4((float)in[3 * i], (float)in[3 * i + 1], (float)in[3 * i + 2], 0.f);
    } else {
        const float f = sentinel ? __builtin_inff() : 0.f;
        v = make_float4(f, f, f, 0.f);
    }
    out[i] = v;
}

void launch_pack_f3_to_f4(hipStream_t st, const float* d_in, int64_t n, float4* d_out, int64_t n_pad, bool sentinel) {
    if (n_pad <= 0) return;
    const int blocks = (int)((n_pad + 255) / 256);
    hipLaunchKernelGGL(pack_to_f4_kernel<float>, dim3(blocks), dim3(256), 0, st, d_in, n, d_out, n_pad, sentinel ? 1 : 0);
}
void launch_pack_f64_to_f4(hipStream_t st, const double* d_in, int64_t n, float4* d_out, int64_t n_pad, bool sentinel) {
    if (n_pad <= 0) return;
    const int blocks = (int)((n_pad + 255) / 256);
    hipLaunchKernelGGL(pack_to_f4_kernel<double>, dim3(blocks), dim3(256), 0, st, d_in, n, d_out, n_pad, sentinel ? 1 : 0);
}

// One launch for a single pair: blocks [0, nb_t) pack the target (sentinel padded) and leave one bbox partial
// {min xyz, max xyz} of the real points per block; blocks [nb_t, nb_t + nb_s) pack the source.
template <typename T>
__global__ __launch_bounds__(256) void pack_pair_kernel(const T* __restrict__ tgt, int64_t nt, float4* __restrict__ tgt_out, int64_t nt_pad,
                                                        int nb_t, float* __restrict__ bbox_partial, const T* __restrict__ src, int64_t ns,
                                                        float4* __restrict__ src_out, unsigned int* __restrict__ box_host, unsigned box_tag) {
    if ((int)blockIdx.x >= nb_t) {
        const int64_t i = (int64_t)(blockIdx.x - nb_t) * 256 + threadIdx.x;
        if (i < ns) src_out[i] = make_float4((float)src[3 * i], (float)src[3 * i + 1], (float)src[3 * i + 2], 0.f);
        return;
    }
    __shared__ float sh[4][6];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const float inf = __builtin_inff();
    float mn[3] = {inf, inf, inf}, mx[3] = {-inf, -inf, -inf};
    if (i < nt_pad) {
        float4 v = make_float4(inf, inf, inf, 0.f);   // sentinel padding: can never win an arg-min
        if (i < nt) {
            v = make_float4((float)tgt[3 * i], (float)tgt[3 * i + 1], (float)tgt[3 * i + 2], 0.f);
            mn[0] = mx[0] = v.x; mn[1] = mx[1] = v.y; mn[2] = mx[2] = v.z;
        }
        tgt_out[i] = v;
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
    if (threadIdx.x < 2) {   // lane 0: the block's minimum, lane 1: its maximum
        const int o = 3 * (int)threadIdx.x;
        float v[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            v[k] = sh[0][o + k];
            for (int w = 1; w < 4; ++w) v[k] = threadIdx.x == 0 ? fminf(v[k], sh[w][o + k]) : fmaxf(v[k], sh[w][o + k]);
            bbox_partial[blockIdx.x * 6 + o + k] = v[k];
        }
        if (box_host) {
            // ... and straight into host memory as a checked 16-byte granule {three floats, launch tag + check word}: the host
            // sizes the cell grid from the box, and a copy + stream synchronisation cost it ~10 us per registration
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            u32x4 g;
            g.x = __float_as_uint(v[0]); g.y = __float_as_uint(v[1]); g.z = __float_as_uint(v[2]); g.w = box_tag + kss_mix3(g.x, g.y, g.z);
            unsigned int* dst = box_host + ((size_t)blockIdx.x * 2 + threadIdx.x) * 4;
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(g) : "memory");
        }
    }
}

int pack_pair_bbox_rows(int64_t nt_pad) { return (int)((nt_pad + 255) / 256); }

void launch_pack_pair(hipStream_t st, int dtype, const void* d_tgt, int64_t nt, float4* d_tgt_out, int64_t nt_pad, float* d_bbox_partial,
                      const void* d_src, int64_t ns, float4* d_src_out, unsigned int* d_box_host, unsigned box_tag) {
    const int nb_t = pack_pair_bbox_rows(nt_pad), nb_s = (int)((ns + 255) / 256);
    if (dtype == KSS_F64)
        hipLaunchKernelGGL(pack_pair_kernel<double>, dim3(nb_t + nb_s), dim3(256), 0, st, (const double*)d_tgt, nt, d_tgt_out, nt_pad, nb_t,
                           d_bbox_partial, (const double*)d_src, ns, d_src_out, d_box_host, box_tag);
    else
        hipLaunchKernelGGL(pack_pair_kernel<float>, dim3(nb_t + nb_s), dim3(256), 0, st, (const float*)d_tgt, nt, d_tgt_out, nt_pad, nb_t,
                           d_bbox_partial, (const float*)d_src, ns, d_src_out, d_box_host, box_tag);
}

// empty launch: calibrates what a HIP event pair adds around a short kernel (kss_profile_event_overhead)
__global__ void empty_kernel() {}
void launch_empty(hipStream_t st) { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, st); }

// Batched form for the targets of many pairs: ONE launch instead of one per pair (a 1024-pair batch used to spend
// more time launching pack kernels than searching).  Output slot i belongs to the pair whose padded segment
// [out_base, out_base + out_pad) contains it (binary search in the per-pair table).
template <typename T>
__global__ __launch_bounds__(256) void pack_batch_kernel(const T* __restrict__ in, const PackSeg* __restrict__ seg, int nseg,
                                                         int64_t total_out, float4* __restrict__ out) {
    const int64_t i0 = (int64_t)blockIdx.x * blockDim.x;
    int lo = 0, hi = nseg - 1;             // ONE search per workgroup (uniform: scalar loads), then a short walk per lane
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (seg[mid].out_base <= i0) lo = mid; else hi = mid - 1;
    }
    const int64_t i = i0 + threadIdx.x;
    if (i >= total_out) return;
    while (lo + 1 < nseg && seg[lo + 1].out_base <= i) ++lo;
    const PackSeg sg = seg[lo];
    const int64_t k = i - sg.out_base;
    float4 v;
    if (k < sg.n) {
        const T* __restrict__ q = in + 3 * (sg.in_off + k);
        v = make_float4((float)q[0], (float)q[1], (float)q[2], 0.f);
    } else {
        const float f = __builtin_inff();   // sentinel padding: can never win an arg-min
        v = make_float4(f, f, f, 0.f);
    }
    out[i] = v;
}

void launch_pack_batch(hipStream_t st, const void* d_in, int dtype, const PackSeg* d_seg, int nseg, int64_t total_out, float4* d_out) {
    if (total_out <= 0 || nseg <= 0) return;
    const int blocks = (int)((total_out + 255) / 256);
    if (dtype == KSS_F64)
        hipLaunchKernelGGL(pack_batch_kernel<double>, dim3(blocks), dim3(256), 0, st, (const double*)d_in, d_seg, nseg, total_out, d_out);
    else
        hipLaunchKernelGGL(pack_batch_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)d_in, d_seg, nseg, total_out, d_out);
}

// ---------------------------------------------------------------------------------------------
// (b) NN sweep.  One workgroup = 256*S sources of one pair x one split of that pair's targets.
//   - sources live in registers (S per lane), optionally transformed on load by the previous
//     iteration's Matrix4f (PCL transformCloud: ((m0*x + m1*y) + m2*z) + m3, float, no fma);
//   - targets stream through a double-buffered 256-point LDS tile (one float4 per thread, one
//     barrier per tile); every lane reads the same LDS address (broadcast, conflict free);
//   - the inner loop keeps only a running MIN per 32-target sub-tile (v_min3), and remembers the
//     first sub-tile that achieved the best value; the winning sub-tile is re-scanned once at the
//     end to recover the exact lowest index.  ~8.6 VALU ops per (source, target) pair;
//   - result key = (bits(d2) << 32) | idx: unsigned-min over splits == (min d2, lowest idx).
// ---------------------------------------------------------------------------------------------
// LIST = true is the fallback pass of the grid search (kss_grid.hip): the sources are the entries of a
// device-side list (count in device memory, so the host launches the worst-case grid and surplus
// workgroups leave at once), they are already transformed, and results are merged into key row 0 with a
// 64-bit atomic min.  In that mode NNWork.src_begin is the offset into the list, NNWork.key_begin the
// pair's key base and NNWork.write_src the pair's first source index.
template <int S, bool FMA, bool LIST>
__global__ __launch_bounds__(NN_THREADS) void nn_sweep_kernel(const NNWork* __restrict__ work,
                                                              const PairState* __restrict__ state,
                                                              const float4* __restrict__ src_in,
                                                              float4* __restrict__ src_out,
                                                              const float4* __restrict__ tgt,
                                                              unsigned long long* __restrict__ keys,
                                                              const int32_t* __restrict__ list,
                                                              const int32_t* __restrict__ list_count) {
    __shared__ float4 tile[2][NN_TILE];
    const NNWork w = work[blockIdx.x];
    const PairState ps = state[w.pair];
    if (!ps.active) return;   // uniform: whole workgroup leaves
    const int tid = threadIdx.x;
    int src_count = w.src_count;
    if constexpr (LIST) {
        src_count = *list_count - w.src_begin;
        if (src_count > NN_THREADS * S) src_count = NN_THREADS * S;
        if (src_count <= 0) return;   // uniform
    }

    float sx[S], sy[S], sz[S], best[S];
    int bsub[S], gidx[S];
#pragma unroll
    for (int j = 0; j < S; ++j) {
        const int l = tid + j * NN_THREADS;
        const bool valid = l < src_count;
        float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (LIST) {
            gidx[j] = valid ? list[w.src_begin + l] : 0;
            if (valid) p = src_out[gidx[j]];
        } else {
            gidx[j] = w.src_begin + l;
            if (valid) p = src_in[gidx[j]];
            if (ps.apply) {
                const float x = p.x, y = p.y, z = p.z;
                p.x = ((ps.m[0] * x + ps.m[1] * y) + ps.m[2] * z) + ps.m[3];
                p.y = ((ps.m[4] * x + ps.m[5] * y) + ps.m[6] * z) + ps.m[7];
                p.z = ((ps.m[8] * x + ps.m[9] * y) + ps.m[10] * z) + ps.m[11];
            }
            if (valid && w.write_src) src_out[gidx[j]] = p;
        }
        sx[j] = p.x; sy[j] = p.y; sz[j] = p.z;
        best[j] = __builtin_inff();
        bsub[j] = 0;
    }

    const float4* __restrict__ tp = tgt + w.tgt_begin;
    const int ntiles = w.tgt_count / NN_TILE;
    float4 pre = tp[tid];
    int buf = 0;
    for (int t = 0; t < ntiles; ++t) {
        tile[buf][tid] = pre;
        __syncthreads();
        if (t + 1 < ntiles) pre = tp[(t + 1) * NN_TILE + tid];
        const float4* __restrict__ tl = tile[buf];
#pragma unroll 1
        for (int sub = 0; sub < NN_TILE / NN_SUB; ++sub) {
            float m[S];
#pragma unroll
            for (int j = 0; j < S; ++j) m[j] = __builtin_inff();
#pragma unroll
            for (int u = 0; u < NN_SUB; ++u) {
                const float4 q = tl[sub * NN_SUB + u];
#pragma unroll
                for (int j = 0; j < S; ++j) m[j] = fminf(m[j], dist2<FMA>(sx[j], sy[j], sz[j], q.x, q.y, q.z));
            }
#pragma unroll
            for (int j = 0; j < S; ++j) {
                if (m[j] < best[j]) {   // strict: the FIRST sub-tile holding the minimum wins
                    best[j] = m[j];
                    bsub[j] = t * (NN_TILE / NN_SUB) + sub;
                }
            }
        }
        buf ^= 1;
    }

    // exact arg-min inside the winning sub-tile (same arithmetic => same bits as `best`)
#pragma unroll
    for (int j = 0; j < S; ++j) {
        const float4* __restrict__ rp = tp + bsub[j] * NN_SUB;
        float bd = __builtin_inff();
        int bi = 0;
#pragma unroll 8
        for (int u = 0; u < NN_SUB; ++u) {
            const float4 q = rp[u];
            const float d = dist2<FMA>(sx[j], sy[j], sz[j], q.x, q.y, q.z);
            if (d < bd) { bd = d; bi = u; }
        }
        const int l = tid + j * NN_THREADS;
        if (l < src_count) {
            const unsigned idx = (unsigned)(w.tgt_begin - w.tgt_pair_base + bsub[j] * NN_SUB + bi);
            const unsigned long long key = ((unsigned long long)__float_as_uint(bd) << 32) | (unsigned long long)idx;
            if constexpr (LIST) atomicMin(&keys[w.key_begin + (gidx[j] - w.write_src)], key);
            else keys[w.key_begin + l] = key;
        }
    }
}

template <bool LIST>
static void nn_sweep_dispatch(hipStream_t st, int S, bool fma, const NNWork* d_work, int n_work, const PairState* d_state,
                              const float4* d_src_in, float4* d_src_out, const float4* d_tgt4, unsigned long long* d_keys,
                              const int32_t* d_list, const int32_t* d_count) {
    if (n_work <= 0) return;
    const dim3 grid(n_work), block(NN_THREADS);
#define KSS_NN_LAUNCH(SV, FV)                                                                                        \
    hipLaunchKernelGGL((nn_sweep_kernel<SV, FV, LIST>), grid, block, 0, st, d_work, d_state, d_src_in, d_src_out, \
                       d_tgt4, d_keys, d_list, d_count)
    if (!fma) {
        switch (S) {
            case 1: KSS_NN_LAUNCH(1, false); break;
            case 2: KSS_NN_LAUNCH(2, false); break;
            case 8: KSS_NN_LAUNCH(8, false); break;
            default: KSS_NN_LAUNCH(4, false); break;
        }
    } else {
        switch (S) {
            case 1: KSS_NN_LAUNCH(1, true); break;
            case 2: KSS_NN_LAUNCH(2, true); break;
            case 8: KSS_NN_LAUNCH(8, true); break;
            default: KSS_NN_LAUNCH(4, true); break;
        }
    }
#undef KSS_NN_LAUNCH
}

void launch_nn_sweep(hipStream_t st, int S, bool fma, const NNWork* d_work, int n_work,
                     const PairState* d_state, const float4* d_src_in, float4* d_src_out,
                     const float4* d_tgt4, unsigned long long* d_keys) {
    nn_sweep_dispatch<false>(st, S, fma, d_work, n_work, d_state, d_src_in, d_src_out, d_tgt4, d_keys, nullptr, nullptr);
}

void launch_nn_sweep_list(hipStream_t st, int S, bool fma, const NNWork* d_work, int n_work,
                          const PairState* d_state, float4* d_src_cur, const float4* d_tgt4,
                          unsigned long long* d_keys, const int32_t* d_list, const int32_t* d_count) {
    nn_sweep_dispatch<true>(st, S, fma, d_work, n_work, d_state, d_src_cur, d_src_cur, d_tgt4, d_keys, d_list, d_count);
}

// ---------------------------------------------------------------------------------------------
// (c) correspondence reduce: merge the per-split keys, gather the matched target, accumulate the
// 20 sums in f64 (wave shuffle reduce -> LDS -> one partial row per workgroup).
// ---------------------------------------------------------------------------------------------
// FUSE = true (small batches): the pair's last workgroup to finish also adds the pair's rows up and publishes the 20 sums
// as {bits, seq} pairs into host-mapped memory -- one launch less per ICP pass than corr_reduce + finalize_sums.  The rows
// are handed over as in grid_nn_kernel: write-through (sc1) stores, vmcnt(0), workgroup barrier, ONE agent-scope ticket;
// the workgroup drawing the pair's last ticket reads every row with sc1 loads, in row order (bitwise reproducible).
template <bool FUSE>
__global__ __launch_bounds__(256) void corr_reduce_kernel(const RedWork* __restrict__ work,
                                                          const PairState* __restrict__ state,
                                                          const float4* __restrict__ src,
                                                          const float4* __restrict__ tgt,
                                                          const unsigned long long* __restrict__ keys,
                                                          double max_d2, double* __restrict__ partials,
                                                          int32_t* __restrict__ idx_out,
                                                          float* __restrict__ d2_out, int index_in_w,
                                                          const PairRed* __restrict__ pair_red, int32_t* __restrict__ pair_ticket,
                                                          unsigned long long* __restrict__ pub, unsigned long long seq) {
    __shared__ double sh[4][NSUMS];
    __shared__ int s_last;
    const RedWork w = work[blockIdx.x];
    double acc[NSUMS];
#pragma unroll
    for (int c = 0; c < NSUMS; ++c) acc[c] = 0.0;
    const bool active = state[w.pair].active != 0;
    const int t0 = threadIdx.x;
    if (active)
    for (int t = t0; t < w.src_count; t += 256) {
        unsigned long long key = ~0ull;
        for (int s = 0; s < w.n_split; ++s) {
            const unsigned long long k = keys[w.key_begin + (int64_t)s * w.key_stride + t];
            key = k < key ? k : key;
        }
        if (key == ~0ull) continue;   // grid search gave up on this source: the list pass fills it, then we run again
        const float d2 = __uint_as_float((unsigned)(key >> 32));
        const int idx = (int)(unsigned)(key & 0xffffffffull);
        const float4 p = src[w.src_begin + t];
        const float4 q = tgt[w.tgt_pair_base + idx];
        accumulate_corr(acc, p.x, p.y, p.z, q.x, q.y, q.z, d2, max_d2);
        const int oi = index_in_w ? __float_as_int(p.w) : w.src_begin + t;   // grid mode keeps sources in cell order
        if (idx_out) idx_out[oi] = idx;
        if (d2_out) d2_out[oi] = d2;
    }
    const double r = block_sum<NSUMS>(acc, sh);
    if constexpr (!FUSE) {
        if (t0 < NSUMS) partials[(int64_t)w.partial_index * NSUMS + t0] = r;
    } else {
        const PairRed pr = pair_red[w.pair];
        if (t0 < NSUMS) {
            __hip_atomic_store(&partials[(int64_t)w.partial_index * NSUMS + t0], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if (t0 == 0) s_last = atomicAdd(&pair_ticket[w.pair], 1) == pr.count - 1;
        __syncthreads();
        if (!s_last) return;
        // column sums of the pair's rows, fixed order: lane (g, c) adds rows g, g + 12, ... (sc1 loads, eight in flight),
        // then the 12 group totals are added in group order -- a 100k-source pair has ~400 rows
        __shared__ double shg[ROWSUM_GROUPS][NSUMS];
        {
            const int g = t0 / NSUMS, c = t0 % NSUMS;
            if (g < ROWSUM_GROUPS) {
                double a = 0.0;
                for (int k = g; k < pr.count; k += 8 * ROWSUM_GROUPS) {
                    double tt[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        tt[j] = k + j * ROWSUM_GROUPS < pr.count
                                    ? __hip_atomic_load(&partials[(int64_t)(pr.first + k + j * ROWSUM_GROUPS) * NSUMS + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                    : 0.0;
#pragma unroll
                    for (int j = 0; j < 8; ++j) a += tt[j];
                }
                shg[g][c] = a;
            }
        }
        __syncthreads();
        if (t0 < NSUMS) {
            double v = 0.0;
            for (int gg = 0; gg < ROWSUM_GROUPS; ++gg) v += shg[gg][t0];
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            const unsigned long long vb = (unsigned long long)__double_as_longlong(v);
            u32x4 o;
            o.x = (unsigned)vb; o.y = (unsigned)(vb >> 32); o.z = (unsigned)seq; o.w = kss_mix3(o.x, o.y, o.z);
            unsigned long long* dst = pub + 2 * ((int64_t)w.pair * NSUMS + t0);
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(o) : "memory");
        }
        if (t0 == 0) pair_ticket[w.pair] = 0;   // re-arm for the next pass (stream order makes it visible)
    }
}

// ---------------------------------------------------------------------------------------------
// The candidate batch of KSSICP_Registration (KSS_ICP.hpp:102-118): up to a few dozen ICPs whose sources are poses of ONE
// <= 2000-point sample against ONE <= 2000-point target.  ONE launch per pass instead of sweep + reduce: a workgroup =
// 32 sources of one candidate x ALL targets, the (padded) target staged in LDS once, EIGHT lanes per source each sweeping an
// eighth of it (sub-tile minima, then the exact arg-min inside the winning sub-tile: the scheme of nn_sweep_kernel, same
// arithmetic, ties to the lowest index), merged by DPP; the correspondence sums of the 32 sources -> one row per workgroup;
// the candidate's last workgroup (ticket) adds its rows in a fixed order and publishes the 20 sums.  Exact brute force:
// these candidates start from local minima of the rotation search, far from convergence -- no cell list pays here.
// Eighth e of the target sits in LDS shifted by e float4 (the eighths are a multiple of 128 dwords apart: unshifted, the
// eight addresses of a wave's read would hit the same banks).
// ---------------------------------------------------------------------------------------------
constexpr int CAND_SRC = 32;      // sources per workgroup (x 8 lanes = 256 threads)
// The candidate kernels' tile: 32 sources of one candidate = four waves of eight sources; source i of a wave lives in lanes
// 8i .. 8i+7 (lane 8i keeps its sums).  For the sweep a wave regroups: SIXTEEN lanes share TWO sources (row g of the wave:
// sources 2g and 2g+1), lane e of the row sweeping slice e of the staged target for both -- one LDS read per two distance
// evaluations (one source per lane was LDS-bound at 2.4 workgroups per CU, and with the compiler's one-read-one-wait
// schedule latency-bound at one: 10.4 us per pass, measured).  Slice e = targets [e * chunk, (e + 1) * chunk), chunk a
// multiple of 8, staged at stride chunk + 1 (odd: the sixteen addresses of a wave's read fall into sixteen different bank
// groups) and padded with +inf points up to 16 * chunk.  Sub-tile minima over 8 points first, then the exact arg-min inside
// the winning sub-tile (same arithmetic => same bits), ties to the lowest index; the sixteen keys are merged by DPP.  The
// reads of the NEXT eight points are issued before the CURRENT eight are evaluated (two register sets, ping-pong).
__device__ __forceinline__ int cand_chunk(int nt_pad) { return ((nt_pad + 15) / 16 + 7) / 8 * 8; }
__device__ __forceinline__ int cand_slot(int k, int chunk) { return k + k / chunk; }
__device__ __forceinline__ void cand_stage(float4* cand_tile, const float4* __restrict__ tgt, int nt_pad, int chunk) {
    const float4 far = make_float4(__builtin_inff(), __builtin_inff(), __builtin_inff(), 0.f);
    for (int k = threadIdx.x; k < 16 * chunk; k += 256) cand_tile[cand_slot(k, chunk)] = k < nt_pad ? tgt[k] : far;
    if (threadIdx.x < 8) cand_tile[16 * chunk + 16 + threadIdx.x] = far;   // (the look-ahead of the last slice reads into these)
}

struct CandQ8 { float x[8], y[8], z[8]; };
__device__ __forceinline__ void cand_load8(CandQ8& q, const float4* __restrict__ t) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float4 v = t[j]; q.x[j] = v.x; q.y[j] = v.y; q.z[j] = v.z; }
}
template <bool FMA>
__device__ __forceinline__ void cand_min8(const CandQ8& q, float ax, float ay, float az, float bx, float by, float bz, float& ma, float& mb) {
    ma = __builtin_inff(); mb = __builtin_inff();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        ma = fminf(ma, dist2<FMA>(ax, ay, az, q.x[j], q.y[j], q.z[j]));
        mb = fminf(mb, dist2<FMA>(bx, by, bz, q.x[j], q.y[j], q.z[j]));
    }
}
template <bool FMA>
__device__ __forceinline__ unsigned long long cand_argmin8(const float4* __restrict__ t, int first, float px, float py, float pz) {
    CandQ8 q;
    cand_load8(q, t);
    float bd = __builtin_inff();
    int bi = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float d = dist2<FMA>(px, py, pz, q.x[j], q.y[j], q.z[j]);
        if (d < bd) { bd = d; bi = j; }
    }
    return ((unsigned long long)__float_as_uint(bd) << 32) | (unsigned long long)(unsigned)(first + bi);
}
__device__ __forceinline__ unsigned long long cand_key_min16(unsigned long long key) {   // minimum over the 16 lanes of a row (every lane gets it)
    auto x1 = [](int v) { return __builtin_amdgcn_mov_dpp(v, 0xb1, 0xf, 0xf, false); };
    auto x2 = [](int v) { return __builtin_amdgcn_mov_dpp(v, 0x4e, 0xf, 0xf, false); };
    auto x4 = [](int v) { const int r = __builtin_amdgcn_mov_dpp(v, 0x104, 0xf, 0x5, false); return __builtin_amdgcn_update_dpp(r, v, 0x114, 0xf, 0xa, false); };
    auto x8 = [](int v) { return __builtin_amdgcn_mov_dpp(v, 0x128, 0xf, 0xf, false); };
#define KSS_CAND_MIN(X) do { const unsigned long long o = ((unsigned long long)(unsigned)X((int)(unsigned)(key >> 32)) << 32) | (unsigned)X((int)(unsigned)key); key = o < key ? o : key; } while (0)
    KSS_CAND_MIN(x1);
    KSS_CAND_MIN(x2);
    KSS_CAND_MIN(x4);
    KSS_CAND_MIN(x8);
#undef KSS_CAND_MIN
    return key;
}
// p: this lane's source (lanes 8i .. 8i+7 hold source i of the wave).  Returns the nearest target's key of THAT source.
template <bool FMA>
__device__ __forceinline__ unsigned long long cand_sweep(const float4* cand_tile, int chunk, float px, float py, float pz) {
    const int lane = threadIdx.x & 63, e = lane & 15, row = lane & 48;
    const float ax = __shfl(px, row, 64), ay = __shfl(py, row, 64), az = __shfl(pz, row, 64);
    const float bx = __shfl(px, row + 8, 64), by = __shfl(py, row + 8, 64), bz = __shfl(pz, row + 8, 64);
    const float4* __restrict__ tl = cand_tile + e * (chunk + 1);
    float best_a = __builtin_inff(), best_b = __builtin_inff();
    int sub_a = 0, sub_b = 0;
    CandQ8 q0, q1;
    cand_load8(q0, tl);
    const int nb = chunk / 8;
    auto step = [&](const CandQ8& cur, CandQ8& nxt, int b) {
        cand_load8(nxt, tl + (b + 1) * 8);          // (past the last batch: the next slice or the padding behind the tile; never used)
        __builtin_amdgcn_sched_barrier(0);
        float ma, mb;
        cand_min8<FMA>(cur, ax, ay, az, bx, by, bz, ma, mb);
        if (ma < best_a) { best_a = ma; sub_a = b * 8; }   // strict: the FIRST sub-tile holding the minimum wins
        if (mb < best_b) { best_b = mb; sub_b = b * 8; }
        __builtin_amdgcn_sched_barrier(0);
    };
    int b = 0;
    for (; b + 1 < nb; b += 2) { step(q0, q1, b); step(q1, q0, b + 1); }
    if (b < nb) step(q0, q1, b);
    unsigned long long key_a = cand_argmin8<FMA>(tl + sub_a, e * chunk + sub_a, ax, ay, az);
    unsigned long long key_b = cand_argmin8<FMA>(tl + sub_b, e * chunk + sub_b, bx, by, bz);
    key_a = cand_key_min16(key_a);
    key_b = cand_key_min16(key_b);
    return (lane & 8) ? key_b : key_a;
}

// Workgroup total of the 20 sums when only the lanes 8i hold a source's contribution (all others +0.0): per column the tree
// of block_sum / wave_sum -- partners at lane distance 32, 16, 8, then 4, 2, 1 -- whose last three levels add +0.0 and are
// skipped (x + 0.0 == x bit for bit; a sum that starts from +0.0 is never -0.0), the first three evaluated with the in-place
// swaps of the canonical tree, two columns per exchange; then the four wave totals in wave order.  Same bits as
// block_sum<NSUMS>, a twentieth of its exchanges (each of which was two LDS permutes: 5.9 us per pass, measured).
__device__ __forceinline__ double cand_block_sum(double (&v)[NSUMS], double (*sh)[NSUMS]) {
    static_assert(NSUMS == 20, "two levels of halving: 20 -> 10 -> 5 columns");
#pragma unroll
    for (int j = 0; j < 10; ++j) v[j] = level32(v[j], v[j + 10]);   // lanes < 32: columns 0-9, lanes >= 32: columns 10-19
#pragma unroll
    for (int j = 0; j < 5; ++j) v[j] = level16(v[j], v[j + 5]);     // rows 0 / 1 / 2 / 3: columns 0-4 / 5-9 / 10-14 / 15-19
#pragma unroll
    for (int j = 0; j < 5; ++j) v[j] += xor8(v[j]);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if ((lane & 15) == 0) {
#pragma unroll
        for (int j = 0; j < 5; ++j) sh[wave][5 * (lane >> 4) + j] = v[j];
    }
    __syncthreads();
    double out = 0.0;
    if (threadIdx.x < NSUMS)
        for (int w = 0; w < 4; ++w) out += sh[w][threadIdx.x];   // fixed order: reproducible
    return out;
}

// Row hand-over of the candidate kernels WITHOUT a counter.  Every tile's row carries a tag -- the launch-and-pass number, stored
// after the row's twenty values have been acknowledged -- and the candidate's workgroup 0 waits until every tile of the
// candidate shows the tag of THIS pass before it adds the rows up.  (Round 3 handed rows over with one ticket per workgroup on
// a per-candidate counter that had to be zero at rest.  With eight registrations in flight on contexts of their own, a debug
// build counted tickets beyond the candidate's workgroups and rows of another pass in launches that ended normally: one
// registration in ten then differed from its solitary run, some grossly.  Where the extra tickets came from was not found in
// the time left; a tag cannot be left over, and a row of another pass cannot be added.)  Returns whether this workgroup adds
// the candidate's rows up (false also when a row never came: bounded; the host's wait reports it).
// stop_rec / stop_tag (resident kernel): the candidate's gate record; a stop order "at whatever pass" may have taken some of the
// candidate's workgroups away before this pass, so a waiting lane looks at the record every 32 polls and gives up when it reads one.
__device__ __forceinline__ bool cand_rows_ready(unsigned int* __restrict__ row_tag, int row0, int nrows, unsigned tag, bool reducer, int polls, int* s_flag,
                                                const unsigned int* stop_rec = nullptr, unsigned stop_tag = 0) {
    if (!reducer) return false;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x;
    if (tid == 0) *s_flag = 1;
    __syncthreads();
    for (int r = tid; r < nrows; r += 256) {
        int n = 0;
        while (__hip_atomic_load(&row_tag[row0 + r], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != tag) {
            __builtin_amdgcn_s_sleep(1);
            if (++n > polls) { *s_flag = 0; break; }
            if (stop_rec && (n & 31) == 0) {
                u32x4 v;
                asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(stop_rec) : "memory");
                if (v.w - kss_mix3(v.x, v.y, v.z) == stop_tag) { *s_flag = 0; break; }
            }
        }
    }
    __syncthreads();
    return *s_flag != 0;
}

// A candidate's last workgroup: the column sums of its rows in a fixed order (lane (g, c) adds rows g, g + 12, ..., then the
// 12 group totals in group order) and the publication of the 20 sums as checked {bits, launch number} granules.
__device__ __forceinline__ void cand_rows_publish(const double* __restrict__ partials, int row0, int nrows, double (*shg)[NSUMS],
                                                  unsigned long long* __restrict__ pub, int pair, unsigned long long seq) {
    const int tid = threadIdx.x;
    {
        const int g = tid / NSUMS, c = tid % NSUMS;
        if (g < ROWSUM_GROUPS) {
            double a = 0.0;
            for (int k = g; k < nrows; k += 8 * ROWSUM_GROUPS) {
                double tt[8];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    tt[j] = k + j * ROWSUM_GROUPS < nrows
                                ? __hip_atomic_load(&partials[(int64_t)(row0 + k + j * ROWSUM_GROUPS) * NSUMS + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                : 0.0;
#pragma unroll
                for (int j = 0; j < 8; ++j) a += tt[j];
            }
            shg[g][c] = a;
        }
    }
    __syncthreads();
    if (tid < NSUMS) {
        double v = 0.0;
        for (int gg = 0; gg < ROWSUM_GROUPS; ++gg) v += shg[gg][tid];
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        const unsigned long long vb = (unsigned long long)__double_as_longlong(v);
        u32x4 o;
        o.x = (unsigned)vb; o.y = (unsigned)(vb >> 32); o.z = (unsigned)seq; o.w = kss_mix3(o.x, o.y, o.z);
        unsigned long long* dst = pub + 2 * ((int64_t)pair * NSUMS + tid);
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(o) : "memory");
    }
}

template <bool FMA>
__global__ __launch_bounds__(256) void cand_pass_kernel(const PairState* __restrict__ state, const float4* __restrict__ src_in, float4* __restrict__ src_out,
                                                        const float4* __restrict__ tgt, int nt_pad, int ns, int blocks_per_pair, double max_d2,
                                                        double* __restrict__ partials, unsigned int* __restrict__ row_tag,
                                                        unsigned long long* __restrict__ pub, unsigned long long seq,
                                                        int32_t* __restrict__ idx_out, float* __restrict__ d2_out) {
    extern __shared__ float4 cand_tile[];
    __shared__ double sh[4][NSUMS];
    __shared__ int s_last;
    const int pair = (int)blockIdx.x / blocks_per_pair, blk = (int)blockIdx.x % blocks_per_pair;
    const PairState ps = state[pair];
    if (!ps.active) return;   // uniform: whole workgroup leaves
    const int tid = threadIdx.x, sub = tid & 7;
    const int chunk = cand_chunk(nt_pad);
    cand_stage(cand_tile, tgt, nt_pad, chunk);
    const int sl = blk * CAND_SRC + (tid >> 3);
    const bool valid = sl < ns;
    const int gi = pair * ns + sl;
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) p = src_in[gi];
    if (ps.apply) {   // pcl transformCloud with the previous iteration's Matrix4f, Eigen order, float, no fma
        const float x = p.x, y = p.y, z = p.z;
        p.x = ((ps.m[0] * x + ps.m[1] * y) + ps.m[2] * z) + ps.m[3];
        p.y = ((ps.m[4] * x + ps.m[5] * y) + ps.m[6] * z) + ps.m[7];
        p.z = ((ps.m[8] * x + ps.m[9] * y) + ps.m[10] * z) + ps.m[11];
    }
    if (valid && sub == 0) src_out[gi] = p;
    __syncthreads();
    const unsigned long long key = cand_sweep<FMA>(cand_tile, chunk, p.x, p.y, p.z);
    double acc[NSUMS];
#pragma unroll
    for (int c = 0; c < NSUMS; ++c) acc[c] = 0.0;
    const bool qok = (p.x - p.x) == 0.f && (p.y - p.y) == 0.f && (p.z - p.z) == 0.f;   // a non-finite query matches nothing
    if (valid && sub == 0 && qok && key != ~0ull) {
        const float d2 = __uint_as_float((unsigned)(key >> 32));
        const int idx = (int)(unsigned)(key & 0xffffffffull);
        const float4 q = cand_tile[cand_slot(idx, chunk)];
        accumulate_corr(acc, p.x, p.y, p.z, q.x, q.y, q.z, d2, max_d2);
        if (idx_out) idx_out[gi] = idx;
        if (d2_out) d2_out[gi] = d2;
    }
    const double r = cand_block_sum(acc, sh);
    const int row0 = pair * blocks_per_pair;
    if (tid < NSUMS) {
        __hip_atomic_store(&partials[(int64_t)(row0 + blk) * NSUMS + tid], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (tid == 0) __hip_atomic_store(&row_tag[row0 + blk], (unsigned)seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    if (!cand_rows_ready(row_tag, row0, blocks_per_pair, (unsigned)seq, blk == blocks_per_pair - 1, 1 << 22, &s_last)) return;   // (the candidate's LAST workgroup waits: every other one was dispatched before it)
    __shared__ double shg[ROWSUM_GROUPS][NSUMS];
    cand_rows_publish(partials, row0, blocks_per_pair, shg, pub, pair, seq);
}

size_t cand_pass_lds_bytes(int nt_pad) { return (size_t)(16 * (((nt_pad + 15) / 16 + 7) / 8 * 8) + 24) * sizeof(float4); }   // 16 padded slices at stride chunk + 1, eight points of look-ahead
int cand_pass_blocks_per_pair(int64_t ns) { return (int)((ns + CAND_SRC - 1) / CAND_SRC); }
// false: the device refused the LDS size (the caller runs sweep + reduce)
bool launch_cand_pass(hipStream_t st, bool fma, int npairs, const PairState* d_state, const float4* d_src_in, float4* d_src_out, const float4* d_tgt,
                      int nt_pad, int ns, double max_d2, double* d_partials, unsigned int* d_row_tag, unsigned long long* d_pub, unsigned long long seq,
                      int32_t* d_idx_out, float* d_d2_out) {
    const size_t bytes = cand_pass_lds_bytes(nt_pad);
    const void* fn = fma ? reinterpret_cast<const void*>(&cand_pass_kernel<true>) : reinterpret_cast<const void*>(&cand_pass_kernel<false>);
    if (bytes > 48 * 1024 && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) { (void)hipGetLastError(); return false; }
    const int bpp = cand_pass_blocks_per_pair(ns);
    const dim3 grid((unsigned)(npairs * bpp)), block(256);
    if (fma) hipLaunchKernelGGL(cand_pass_kernel<true>, grid, block, bytes, st, d_state, d_src_in, d_src_out, d_tgt, nt_pad, ns, bpp, max_d2, d_partials, d_row_tag, d_pub, seq, d_idx_out, d_d2_out);
    else hipLaunchKernelGGL(cand_pass_kernel<false>, grid, block, bytes, st, d_state, d_src_in, d_src_out, d_tgt, nt_pad, ns, bpp, max_d2, d_partials, d_row_tag, d_pub, seq, d_idx_out, d_d2_out);
    return true;
}

// ---------------------------------------------------------------------------------------------
// The candidate batch, every workgroup resident for the whole registration (CandArgs, kss_internal.hpp).  Same tiles, same
// sweep, same sums in the same order as cand_pass_kernel -- a candidate gives the same bits on both -- but ONE launch per
// batch: the target is staged once, a workgroup's sources stay in registers, and after a candidate's last workgroup has
// published the 20 sums of a pass all its workgroups wait at the candidate's gate record (five checked granules the host
// stores through the BAR, the protocol of resident_icp_kernel) for the next transform, the order to run the
// getFitnessScore() pass, or to stop.  Every poll is bounded: a workgroup nobody answers leaves and the host starts over on
// the launch-per-pass form.  The launch must fit the device at once (cand_resident_capacity): a workgroup that never
// starts would keep its candidate's sums from ever being complete.
// ---------------------------------------------------------------------------------------------
template <bool FMA>
__global__ __launch_bounds__(256) void cand_resident_kernel(const CandArgs a) {
    extern __shared__ float4 cand_tile[];
    __shared__ double sh[4][NSUMS];
    __shared__ double shg[ROWSUM_GROUPS][NSUMS];
    __shared__ int s_ps[16];
    __shared__ int s_ctl[2];      // [0] the gate was answered, [1] every row of the candidate has come (workgroup 0)
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const int pair = (int)blockIdx.x / a.wpp, wg = (int)blockIdx.x % a.wpp;
    const int tid = threadIdx.x, sub = tid & 7;
    const int nt_pad = a.nt_pad, ns = a.ns;
    const int chunk = cand_chunk(nt_pad);
    // diagnostic timeline (100 MHz s_memrealtime), by workgroup 0 of each candidate: [0] start, [1] target staged, [15] end; sums
    // over the passes >= 1 of {8 gate wait, 9 sweeps, 10 sums + rows, 11 tags + workgroup 0's wait for every row, 12 the candidate's total + publication}; 7 passes
    unsigned long long t_last = 0;
    const bool stamping = a.stamps != nullptr && wg == 0 && tid == 0;
#define KSS_CSTAMP(k) do { if (stamping) a.stamps[(size_t)pair * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define KSS_CLAP(k) do { if (stamping) { const unsigned long long now_ = __builtin_amdgcn_s_memrealtime(); if (pass > 0) a.stamps[(size_t)pair * 16 + (k)] += now_ - t_last; t_last = now_; } } while (0)
    KSS_CSTAMP(0);
    // leaving (every exit is uniform over the workgroup): the candidate's workgroup 0 -- the one that publishes its sums -- tells
    // the host that nothing more will come for this candidate.  (No count of the workgroups that have left: like the row
    // tickets it would have to be zero at rest, see cand_rows_ready.)
    auto leave = [&]() {
        if (wg == 0 && tid == 0) res_store_exit_flag(a.exit_flags, pair, a.stamp0);
    };
    cand_stage(cand_tile, a.tgt, nt_pad, chunk);
    // tile j of this workgroup = tile wg + j * wpp of the candidate: 32 sources, eight lanes each
    float px[CAND_TPW], py[CAND_TPW], pz[CAND_TPW];
#pragma unroll
    for (int j = 0; j < CAND_TPW; ++j) {
        px[j] = py[j] = pz[j] = 0.f;
        const int blk = wg + j * a.wpp, sl = blk * CAND_SRC + (tid >> 3);
        if (j < a.tpw && blk < a.bpp && sl < ns) {
            const float4 v = a.src0[(int64_t)pair * ns + sl];
            px[j] = v.x; py[j] = v.y; pz[j] = v.z;
        }
    }
    __syncthreads();
    KSS_CSTAMP(1);
    if (stamping) t_last = __builtin_amdgcn_s_memrealtime();
    const int row0 = pair * a.bpp;
    float m[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) m[k] = 0.f;
    int apply = 0, mode = 0;      // mode 0: a regular pass; 1: the getFitnessScore() pass (the last one); 2: stop
    for (int pass = 0; pass < a.max_passes; ++pass) {
        if (pass > 0) {
            if (tid < 8) {
                const unsigned expect = a.stamp0 + (unsigned)pass;
                const unsigned int* src = a.gate + (size_t)pair * 32 + 4 * min(tid, 4);
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
                if (tid == 0) s_ctl[0] = ok ? 1 : 0;
            }
            __syncthreads();
            if (!s_ctl[0]) { KSS_CSTAMP(15); leave(); return; }   // nobody answered: leave (the host's wait reports it)
#pragma unroll
            for (int k = 0; k < 12; ++k) m[k] = __int_as_float(__builtin_amdgcn_readfirstlane(s_ps[k]));
            apply = __builtin_amdgcn_readfirstlane(s_ps[13]);
            mode = __builtin_amdgcn_readfirstlane(s_ps[14]);
            if (mode >= 2) { KSS_CSTAMP(15); leave(); return; }
        }
        KSS_CLAP(8);
        const bool fit = mode == 1;
        // (ONE copy of the tile's code in a runtime loop: the tile being worked on is register 0 of each array, the arrays
        // rotate by one after every trip and CAND_TPW trips bring every tile home again)
#pragma unroll 1
        for (int j = 0; j < CAND_TPW; ++j) {
            const int blk = wg + j * a.wpp;
            const bool mine = j < a.tpw && blk < a.bpp;   // uniform
            if (mine) {
            const int sl = blk * CAND_SRC + (tid >> 3);
            const bool valid = sl < ns;
            const int64_t gi = (int64_t)pair * ns + sl;
            float x = px[0], y = py[0], z = pz[0];
            if (fit && valid) {   // getFitnessScore(): final * ORIGINAL input
                const float4 v = a.src0[gi];
                x = v.x; y = v.y; z = v.z;
            }
            if (apply) {   // pcl transformCloud with the previous iteration's Matrix4f, Eigen order, float, no fma
                px[0] = ((m[0] * x + m[1] * y) + m[2] * z) + m[3];
                py[0] = ((m[4] * x + m[5] * y) + m[6] * z) + m[7];
                pz[0] = ((m[8] * x + m[9] * y) + m[10] * z) + m[11];
            }
            const float4 p = make_float4(px[0], py[0], pz[0], 0.f);
            const unsigned long long key = cand_sweep<FMA>(cand_tile, chunk, p.x, p.y, p.z);
            KSS_CLAP(9);
            double acc[NSUMS];
#pragma unroll
            for (int c = 0; c < NSUMS; ++c) acc[c] = 0.0;
            const bool qok = (p.x - p.x) == 0.f && (p.y - p.y) == 0.f && (p.z - p.z) == 0.f;   // a non-finite query matches nothing
            if (valid && sub == 0 && qok && key != ~0ull) {
                const float d2 = __uint_as_float((unsigned)(key >> 32));
                const int idx = (int)(unsigned)(key & 0xffffffffull);
                const float4 q = cand_tile[cand_slot(idx, chunk)];
                accumulate_corr(acc, p.x, p.y, p.z, q.x, q.y, q.z, d2, a.max_d2);
                if (fit && a.idx_out) a.idx_out[gi] = idx;
                if (fit && a.d2_out) a.d2_out[gi] = d2;
            }
            const double r = cand_block_sum(acc, sh);
            if (tid < NSUMS) __hip_atomic_store(&a.partials[(int64_t)(row0 + blk) * NSUMS + tid], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();               // (sh is written again by the next tile)
            }
            {
                const float x0 = px[0], y0 = py[0], z0 = pz[0];
#pragma unroll
                for (int k = 0; k + 1 < CAND_TPW; ++k) { px[k] = px[k + 1]; py[k] = py[k + 1]; pz[k] = pz[k + 1]; }
                px[CAND_TPW - 1] = x0; py[CAND_TPW - 1] = y0; pz[CAND_TPW - 1] = z0;
            }
        }
        if (tid < NSUMS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        KSS_CLAP(10);
        {
            const unsigned tag = (unsigned)(a.seq0 + (unsigned long long)pass);
            if (tid < CAND_TPW) {          // this workgroup's tiles: their rows are out
                const int blk = wg + tid * a.wpp;
                if (tid < a.tpw && blk < a.bpp) __hip_atomic_store(&a.row_tag[row0 + blk], tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
            const bool sum_it = cand_rows_ready(a.row_tag, row0, a.bpp, tag, wg == 0, a.gate_polls, &s_ctl[1], a.gate + (size_t)pair * 32, a.stamp0 + RES_STAMP_ANY);
            KSS_CLAP(11);
            if (sum_it) cand_rows_publish(a.partials, row0, a.bpp, shg, a.pub, pair, a.seq0 + (unsigned long long)pass);
            else if (wg == 0) { KSS_CSTAMP(15); leave(); return; }   // stopped, or a row never came (bounded wait; the host's wait reports it)
        }
        KSS_CLAP(12);
        if (stamping) a.stamps[(size_t)pair * 16 + 7] = (unsigned long long)(pass + 1);
        if (fit) break;
    }
    KSS_CSTAMP(15);
    leave();
#undef KSS_CSTAMP
#undef KSS_CLAP
}

int cand_resident_capacity(bool fma, int nt_pad) {
    const size_t bytes = cand_pass_lds_bytes(nt_pad);
    const void* fn = fma ? reinterpret_cast<const void*>(&cand_resident_kernel<true>) : reinterpret_cast<const void*>(&cand_resident_kernel<false>);
    if (bytes > 48 * 1024 && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) { (void)hipGetLastError(); return 0; }
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, bytes) != hipSuccess || hipGetDevice(&dev) != hipSuccess ||
        hipGetDeviceProperties(&prop, dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return per_cu * prop.multiProcessorCount;
}

int launch_cand_resident(hipStream_t st, bool fma, int npairs, const CandArgs& a, std::string& err) {
    const size_t bytes = cand_pass_lds_bytes(a.nt_pad);
    const void* fn = fma ? reinterpret_cast<const void*>(&cand_resident_kernel<true>) : reinterpret_cast<const void*>(&cand_resident_kernel<false>);
    if (bytes > 48 * 1024 && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
        (void)hipGetLastError();
        err = "cand_resident: the device refused the LDS size";
        return KSS_ERR_HIP;
    }
    const dim3 grid((unsigned)(npairs * a.wpp)), block(256);
    if (fma) hipLaunchKernelGGL(cand_resident_kernel<true>, grid, block, bytes, st, a);
    else hipLaunchKernelGGL(cand_resident_kernel<false>, grid, block, bytes, st, a);
    if (hipGetLastError() != hipSuccess) { err = "cand_resident: launch failed"; return KSS_ERR_HIP; }
    return KSS_OK;
}

void launch_corr_reduce(hipStream_t st, const RedWork* d_work, int n_work, const PairState* d_state,
                        const float4* d_src, const float4* d_tgt4, const unsigned long long* d_keys,
                        double max_d2, double* d_partials, int32_t* d_idx_out, float* d_d2_out, int index_in_w) {
    if (n_work <= 0) return;
    hipLaunchKernelGGL(corr_reduce_kernel<false>, dim3(n_work), dim3(256), 0, st, d_work, d_state, d_src, d_tgt4,
                       d_keys, max_d2, d_partials, d_idx_out, d_d2_out, index_in_w, (const PairRed*)nullptr, (int32_t*)nullptr,
                       (unsigned long long*)nullptr, 0ull);
}

void launch_corr_reduce_publish(hipStream_t st, const RedWork* d_work, int n_work, const PairState* d_state,
                                const float4* d_src, const float4* d_tgt4, const unsigned long long* d_keys,
                                double max_d2, double* d_partials, int32_t* d_idx_out, float* d_d2_out, int index_in_w,
                                const PairRed* d_pair_red, int32_t* d_pair_ticket, unsigned long long* d_pub, unsigned long long seq) {
    if (n_work <= 0) return;
    hipLaunchKernelGGL(corr_reduce_kernel<true>, dim3(n_work), dim3(256), 0, st, d_work, d_state, d_src, d_tgt4,
                       d_keys, max_d2, d_partials, d_idx_out, d_d2_out, index_in_w, d_pair_red, d_pair_ticket, d_pub, seq);
}

// idx-driven variant behind kss_cov(): packed float3 clouds + an index array
__global__ __launch_bounds__(256) void corr_reduce_idx_kernel(const float* __restrict__ src,
                                                              const float* __restrict__ tgt,
                                                              const int32_t* __restrict__ idx, int64_t n,
                                                              double max_d2, double* __restrict__ partials) {
    __shared__ double sh[4][NSUMS];
    double acc[NSUMS];
#pragma unroll
    for (int c = 0; c < NSUMS; ++c) acc[c] = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t j = idx[i];
        const float px = src[3 * i], py = src[3 * i + 1], pz = src[3 * i + 2];
        const float qx = tgt[3 * j], qy = tgt[3 * j + 1], qz = tgt[3 * j + 2];
        const float d2 = dist2<false>(px, py, pz, qx, qy, qz);
        accumulate_corr(acc, px, py, pz, qx, qy, qz, d2, max_d2);
    }
    const double r = block_sum<NSUMS>(acc, sh);
    if (threadIdx.x < NSUMS) partials[(int64_t)blockIdx.x * NSUMS + threadIdx.x] = r;
}

void launch_corr_reduce_idx(hipStream_t st, const float* d_src3, const float* d_tgt3, const int32_t* d_idx,
                            int64_t n, double max_d2, double* d_partials, int n_blocks) {
    hipLaunchKernelGGL(corr_reduce_idx_kernel, dim3(n_blocks), dim3(256), 0, st, d_src3, d_tgt3, d_idx, n,
                       max_d2, d_partials);
}

// final stage of brute-force batches above PUB_PAIRS pairs: sums[pair][c] = sum over that pair's partial rows, in row
// order (deterministic), written to host-mapped memory; the host synchronizes the stream.
__global__ __launch_bounds__(256) void finalize_sums_kernel(const PairRed* __restrict__ pairs,
                                                            const double* __restrict__ partials,
                                                            double* __restrict__ out) {
    __shared__ double shg[ROWSUM_GROUPS][NSUMS];
    const PairRed pr = pairs[blockIdx.x];
    const double v = rows_column_sum(partials + (int64_t)pr.first * NSUMS, pr.count, shg);
    if (threadIdx.x < NSUMS) out[(int64_t)blockIdx.x * NSUMS + threadIdx.x] = v;
}

void launch_finalize_sums(hipStream_t st, const PairRed* d_pairs, int n_pairs, const double* d_partials, double* d_out) {
    if (n_pairs <= 0) return;
    hipLaunchKernelGGL(finalize_sums_kernel, dim3(n_pairs), dim3(256), 0, st, d_pairs, d_partials, d_out);
}

// ---------------------------------------------------------------------------------------------
// (a) KSS pre-shape statistics (initRegistration_MiddleAlign, initRegistrationKSS.hpp:144-207): centroid and mean
// distance to the centroid of ONE OR TWO clouds (source and target of a registration) in TWO launches and no stream
// synchronisation:
//   launch A  every workgroup sums its slice (f64) -> one partial row; the workgroup that draws the cloud's last
//             ticket adds the rows in row order and stores centroid = sum / n;
//   launch B  every workgroup sums sqrt(|p - centroid|^2) over the SAME slice (one IEEE f64 sqrt per point; the slice is
//             still in that XCD's L2 at <= a few million points) -> partial row; the last workgroup of the cloud adds
//             the rows and publishes {centroid, radius sum} as {bits, sequence number} 16-byte pairs into host-mapped
//             memory, where the host spins on the sequence numbers (as the ICP loop does).
// A kernel boundary (~2 us) is cheaper than a grid-wide barrier (4-7 us, MI355X_MICROARCH.md price list) and has no
// co-residency requirement.  Row hand-off as in grid_nn_kernel: sc1 stores, vmcnt(0), workgroup barrier, one ticket.
// The block decomposition of a cloud depends on its size only, so "both clouds in one call" and "one call per cloud"
// give the same bits.
// ---------------------------------------------------------------------------------------------
int stream_blocks(int64_t n) {   // streaming kernels without a hand-over: 256 CUs x 8 workgroups, grid-stride beyond
    int64_t b = (n + 255) / 256;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (int)b;
}

int preshape_blocks(int64_t n) {
    // >= 8 points per lane before another workgroup is added: every workgroup costs a row hand-over and a ticket on the
    // cloud's counter (one counter takes ~12 ns per ticket: 2048 workgroups on a 1M-point cloud spent 25 us per launch
    // queueing there).  Capped at 256 CUs x 8 workgroups, grid-stride beyond (64M points: 2048 workgroups, as before).
    // (round 3: 16 points per lane -- 245 workgroups and tickets per 1M-point cloud instead of 489: both launches of C4's two
    // clouds are held by their tails -- the serial tickets, then the last workgroup's row sum -- not by the 24 MB they stream)
    int64_t b = (n + 4095) / 4096;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (int)b;
}

struct PreCloud {
    const void* xyz;
    int64_t n;
    int32_t first_block, n_blocks;
    int32_t kind;      // 0: f64 triples, 1: f32 triples (scalar loads), 2: f32 read as a flat float4 stream
    int32_t pad;
};
struct PreArgs { PreCloud c[2]; int32_t nclouds; };

template <typename T>
__device__ __forceinline__ void pre_sum_scalar(const T* __restrict__ xyz, int64_t n, int lb, int nb, double (&acc)[3]) {
    for (int64_t i = (int64_t)lb * 256 + threadIdx.x; i < n; i += (int64_t)nb * 256) {
        acc[0] += (double)xyz[3 * i];
        acc[1] += (double)xyz[3 * i + 1];
        acc[2] += (double)xyz[3 * i + 2];
    }
}

// f32 fast path: the cloud is read as a flat array of float4.  A lane takes THREE consecutive float4 = 12 floats = exactly
// four points per trip (48 contiguous bytes per lane, 3 KB per wave, three loads in flight) -- the mapping of the radius pass
// below.  (Round 3; one float4 per lane and trip before: both clouds of C4 38 -> 34 us per call.  Built on it, measured and
// removed: both passes in ONE launch, the points kept in registers and the launches' boundary replaced by a wait for the
// centroid -- 57 us: 490 workgroups spinning on one record at three waves per SIMD cost more than the second launch.)
struct PreGroup { float4 a, b, c; };
__device__ __forceinline__ void pre_sum_group(const PreGroup& g, double (&acc)[3]) {
    acc[0] += (double)g.a.x; acc[1] += (double)g.a.y; acc[2] += (double)g.a.z;
    acc[0] += (double)g.a.w; acc[1] += (double)g.b.x; acc[2] += (double)g.b.y;
    acc[0] += (double)g.b.z; acc[1] += (double)g.b.w; acc[2] += (double)g.c.x;
    acc[0] += (double)g.c.y; acc[1] += (double)g.c.z; acc[2] += (double)g.c.w;
}
__device__ __forceinline__ void pre_sum_f32v(const float* __restrict__ xyz, int64_t n, int lb, int nb, double (&acc)[3]) {
    const float4* __restrict__ v = (const float4*)xyz;
    const int64_t ngroups = n / 4;
#pragma unroll 2
    for (int64_t q = (int64_t)lb * 256 + threadIdx.x; q < ngroups; q += (int64_t)nb * 256) {
        PreGroup g;
        g.a = v[3 * q]; g.b = v[3 * q + 1]; g.c = v[3 * q + 2];
        pre_sum_group(g, acc);
    }
    if (lb == 0 && threadIdx.x == 0)   // the last n % 4 points
        for (int64_t k = 4 * ngroups; k < n; ++k) { acc[0] += (double)xyz[3 * k]; acc[1] += (double)xyz[3 * k + 1]; acc[2] += (double)xyz[3 * k + 2]; }
}

template <typename T>
__device__ __forceinline__ double pre_radius_scalar(const T* __restrict__ xyz, int64_t n, int lb, int nb, double cx, double cy, double cz) {
    double acc = 0.0;
    for (int64_t i = (int64_t)lb * 256 + threadIdx.x; i < n; i += (int64_t)nb * 256) {
        const double xl = (double)xyz[3 * i] - cx, yl = (double)xyz[3 * i + 1] - cy, zl = (double)xyz[3 * i + 2] - cz;
        acc += sqrt((xl * xl + yl * yl) + zl * zl);
    }
    return acc;
}

// radius pass of the float4 stream: a lane takes THREE consecutive float4 = 12 floats = exactly four points (48 contiguous
// bytes per lane, 3 KB per wave: every byte is requested once -- the first version read float4 j and j+1 per lane, i.e.
// every byte twice, and ran at half the sum pass's rate).
__device__ __forceinline__ double pre_radius_f32v(const float* __restrict__ xyz, int64_t n, int lb, int nb, double cx, double cy, double cz) {
    const float4* __restrict__ v = (const float4*)xyz;
    const int64_t ngroups = n / 4;
    double acc = 0.0;
    auto add = [&](float x, float y, float z) {
        const double xl = (double)x - cx, yl = (double)y - cy, zl = (double)z - cz;
        acc += sqrt((xl * xl + yl * yl) + zl * zl);
    };
#pragma unroll 2
    for (int64_t q = (int64_t)lb * 256 + threadIdx.x; q < ngroups; q += (int64_t)nb * 256) {
        const float4 a = v[3 * q], b = v[3 * q + 1], c = v[3 * q + 2];
        add(a.x, a.y, a.z); add(a.w, b.x, b.y); add(b.z, b.w, c.x); add(c.y, c.z, c.w);
    }
    if (lb == 0 && threadIdx.x == 0)   // the last n % 4 points
        for (int64_t k = 4 * ngroups; k < n; ++k) add(xyz[3 * k], xyz[3 * k + 1], xyz[3 * k + 2]);
    return acc;
}

// hand the workgroup's row over and tell whether this workgroup drew the cloud's last ticket (then every row is visible
// to its sc1 loads)
template <int NV>
__device__ __forceinline__ bool pre_row_handoff(double r, double* __restrict__ row, int32_t* __restrict__ ticket, int nb, int* s_last) {
    if (threadIdx.x < NV) {
        __hip_atomic_store(&row[threadIdx.x], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (threadIdx.x == 0) *s_last = atomicAdd(ticket, 1) == nb - 1;
    __syncthreads();
    return *s_last != 0;
}

// launch A.  partials: 4 doubles per workgroup of the launch; tickets[0..1]; cent: 4 doubles per cloud.
__global__ __launch_bounds__(256) void preshape_sum_kernel(PreArgs a, double* __restrict__ partials, int32_t* __restrict__ tickets,
                                                           double* __restrict__ cent) {
    __shared__ double sh[4][3];
    __shared__ int s_last;
    const int ci = (a.nclouds > 1 && (int)blockIdx.x >= a.c[1].first_block) ? 1 : 0;
    const PreCloud cl = a.c[ci];
    const int lb = (int)blockIdx.x - cl.first_block, nb = cl.n_blocks;
    double acc[3] = {0.0, 0.0, 0.0};
    if (cl.kind == 0) pre_sum_scalar((const double*)cl.xyz, cl.n, lb, nb, acc);
    else if (cl.kind == 2) pre_sum_f32v((const float*)cl.xyz, cl.n, lb, nb, acc);
    else pre_sum_scalar((const float*)cl.xyz, cl.n, lb, nb, acc);
    const double r = block_sum<3>(acc, sh);
    if (!pre_row_handoff<3>(r, partials + (int64_t)blockIdx.x * 4, tickets + ci, nb, &s_last)) return;
    // centroid = (sum over the cloud's rows, lanes striding the rows, then a fixed-order block sum) / n
    double t[3] = {0.0, 0.0, 0.0};
    for (int rr = threadIdx.x; rr < nb; rr += 256)
#pragma unroll
        for (int k = 0; k < 3; ++k)
            t[k] += __hip_atomic_load(&partials[(int64_t)(cl.first_block + rr) * 4 + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double v = block_sum<3>(t, sh);
    if (threadIdx.x < 3) cent[ci * 4 + threadIdx.x] = v / (double)cl.n;
    if (threadIdx.x == 0) tickets[ci] = 0;   // re-armed for the next call (stream order)
}

// launch B.  tickets[2..3]; pub: host-mapped {bits, seq} pairs, slots ci * 4 + {0,1,2: centroid, 3: sum of radii}.
__global__ __launch_bounds__(256) void preshape_radius_kernel(PreArgs a, double* __restrict__ partials, int32_t* __restrict__ tickets,
                                                              const double* __restrict__ cent, unsigned long long* __restrict__ pub,
                                                              unsigned long long seq) {
    __shared__ double sh[4][1];
    __shared__ int s_last;
    const int ci = (a.nclouds > 1 && (int)blockIdx.x >= a.c[1].first_block) ? 1 : 0;
    const PreCloud cl = a.c[ci];
    const int lb = (int)blockIdx.x - cl.first_block, nb = cl.n_blocks;
    const double cx = cent[ci * 4], cy = cent[ci * 4 + 1], cz = cent[ci * 4 + 2];
    double acc[1];
    if (cl.kind == 0) acc[0] = pre_radius_scalar((const double*)cl.xyz, cl.n, lb, nb, cx, cy, cz);
    else if (cl.kind == 2) acc[0] = pre_radius_f32v((const float*)cl.xyz, cl.n, lb, nb, cx, cy, cz);
    else acc[0] = pre_radius_scalar((const float*)cl.xyz, cl.n, lb, nb, cx, cy, cz);
    const double r = block_sum<1>(acc, sh);
    if (!pre_row_handoff<1>(r, partials + (int64_t)blockIdx.x * 4, tickets + 2 + ci, nb, &s_last)) return;
    double t[1] = {0.0};
    for (int rr = threadIdx.x; rr < nb; rr += 256)
        t[0] += __hip_atomic_load(&partials[(int64_t)(cl.first_block + rr) * 4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double v = block_sum<1>(t, sh);
    if (threadIdx.x < 4) {
        const double o = threadIdx.x == 0 ? v : (threadIdx.x == 1 ? cx : (threadIdx.x == 2 ? cy : cz));
        const int slot = ci * 4 + (threadIdx.x == 0 ? 3 : (int)threadIdx.x - 1);
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        const unsigned long long vb = (unsigned long long)__double_as_longlong(o);
        u32x4 w;
        w.x = (unsigned)vb; w.y = (unsigned)(vb >> 32); w.z = (unsigned)seq; w.w = kss_mix3(w.x, w.y, w.z);
        unsigned long long* dst = pub + 2 * slot;
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(w) : "memory");
    }
    if (threadIdx.x == 0) tickets[2 + ci] = 0;
}

static bool f32_vector_ok(const void* p, int64_t n) { return ((uintptr_t)p & 15u) == 0 && n >= 1024; }

// d_partials: 4 doubles per workgroup (preshape_blocks(n0) + preshape_blocks(n1)); d_tickets: 4 zeroed ints;
// d_cent: 8 doubles; d_pub: 8 host-mapped {bits, seq} slots.  d_xyz[1] == nullptr: one cloud.
void launch_preshape_pair(hipStream_t st, const void* const d_xyz[2], const int64_t n[2], int dtype, double* d_partials,
                          int32_t* d_tickets, double* d_cent, unsigned long long* d_pub, unsigned long long seq) {
    PreArgs a;
    a.nclouds = d_xyz[1] ? 2 : 1;
    int first = 0;
    for (int k = 0; k < 2; ++k) {
        PreCloud& c = a.c[k];
        c.xyz = k < a.nclouds ? d_xyz[k] : nullptr;
        c.n = k < a.nclouds ? n[k] : 0;
        c.n_blocks = k < a.nclouds ? preshape_blocks(n[k]) : 0;
        c.first_block = first;
        c.kind = dtype == KSS_F64 ? 0 : (k < a.nclouds && f32_vector_ok(d_xyz[k], n[k]) ? 2 : 1);
        c.pad = 0;
        first += c.n_blocks;
    }
    hipLaunchKernelGGL(preshape_sum_kernel, dim3(first), dim3(256), 0, st, a, d_partials, d_tickets, d_cent);
    hipLaunchKernelGGL(preshape_radius_kernel, dim3(first), dim3(256), 0, st, a, d_partials, d_tickets, (const double*)d_cent, d_pub, seq);
}

// out[c] = sum over rows of partials[r][c], n_cols <= NSUMS: lanes stride over the rows (many loads in flight)
__global__ __launch_bounds__(256) void sum_columns_kernel(const double* __restrict__ partials, int n_rows, int n_cols,
                                                          double* __restrict__ out) {
    __shared__ double sh[4][NSUMS];
    double acc[NSUMS];
#pragma unroll
    for (int c = 0; c < NSUMS; ++c) acc[c] = 0.0;
    for (int r = threadIdx.x; r < n_rows; r += 256)
#pragma unroll
        for (int c = 0; c < NSUMS; ++c)
            if (c < n_cols) acc[c] += partials[(int64_t)r * n_cols + c];
    const double v = block_sum<NSUMS>(acc, sh);
    if ((int)threadIdx.x < n_cols) out[threadIdx.x] = v;
}
void launch_sum_columns(hipStream_t st, const double* d_partials, int n_rows, int n_cols, double* d_out) {
    hipLaunchKernelGGL(sum_columns_kernel, dim3(1), dim3(256), 0, st, d_partials, n_rows, n_cols, d_out);
}

// out[r] = (sum_c partials[r][c]) * scale, one thread per row, columns added in order
__global__ __launch_bounds__(256) void row_sums_kernel(const double* __restrict__ partials, int n_rows, int n_cols,
                                                       double scale, double* __restrict__ out) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    double a = 0.0;
    for (int c = 0; c < n_cols; ++c) a += partials[(int64_t)r * n_cols + c];
    out[r] = a * scale;
}
void launch_row_sums(hipStream_t st, const double* d_partials, int n_rows, int n_cols, double scale, double* d_out) {
    hipLaunchKernelGGL(row_sums_kernel, dim3((n_rows + 255) / 256), dim3(256), 0, st, d_partials, n_rows, n_cols, scale, d_out);
}

// ---------------------------------------------------------------------------------------------
// pose application (initRegistration_Rotation[_Angle]) and Matrix4f application, f64 streaming
// ---------------------------------------------------------------------------------------------
struct PoseArgs {
    double shift[3], center[3], scale;
    double cs[6];   // cos/sin of the x, y, z angles, evaluated on the host (same libm as the caller)
};

__device__ __forceinline__ void euler_rotate(double& x, double& y, double& z, double cx, double sx, double cy,
                                             double sy, double cz, double sz) {
    // initRegistration_Transfer cord 1, 2, 3 in sequence (initRegistrationKSS.hpp:365-404)
    const double y1 = y * cx - z * sx;
    const double z1 = y * sx + z * cx;
    const double x2 = z1 * sy + x * cy;
    const double z2 = z1 * cy - x * sy;
    const double x3 = x2 * cz - y1 * sz;
    const double y3 = x2 * sz + y1 * cz;
    x = x3; y = y3; z = z2;
}

__global__ __launch_bounds__(256) void pose_apply_kernel(const double* __restrict__ in, int64_t n, PoseArgs a,
                                                         double* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        double v[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double s = in[3 * i + k] + a.shift[k];            // :78-80
            v[k] = a.center[k] + (s - a.center[k]) * a.scale;       // :81-83
        }
        euler_rotate(v[0], v[1], v[2], a.cs[0], a.cs[1], a.cs[2], a.cs[3], a.cs[4], a.cs[5]);
        out[3 * i] = v[0]; out[3 * i + 1] = v[1]; out[3 * i + 2] = v[2];
    }
}

void launch_pose_apply(hipStream_t st, const double* d_in, int64_t n, const kss_pose& pose, const double cs[6], double* d_out) {
    if (n <= 0) return;
    PoseArgs a;
    for (int k = 0; k < 3; ++k) { a.shift[k] = pose.shift[k]; a.center[k] = pose.center[k]; }
    a.scale = pose.scale;
    for (int k = 0; k < 6; ++k) a.cs[k] = cs[k];
    hipLaunchKernelGGL(pose_apply_kernel, dim3(stream_blocks(n)), dim3(256), 0, st, d_in, n, a, d_out);
}

// the same pose with up to POSE_MANY different Euler angles, one output cloud per angle (the candidate poses of a
// registration: one launch instead of one per candidate); same arithmetic per point as pose_apply_kernel
struct PoseManyArgs { double shift[3], center[3], scale; double cs[POSE_MANY][6]; };
__global__ __launch_bounds__(256) void pose_apply_many_kernel(const double* __restrict__ in, int64_t n, PoseManyArgs a, double* __restrict__ out) {
    const int q = blockIdx.y;
    double* __restrict__ o = out + (int64_t)q * n * 3;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        double v[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double s = in[3 * i + k] + a.shift[k];
            v[k] = a.center[k] + (s - a.center[k]) * a.scale;
        }
        euler_rotate(v[0], v[1], v[2], a.cs[q][0], a.cs[q][1], a.cs[q][2], a.cs[q][3], a.cs[q][4], a.cs[q][5]);
        o[3 * i] = v[0]; o[3 * i + 1] = v[1]; o[3 * i + 2] = v[2];
    }
}

void launch_pose_apply_many(hipStream_t st, const double* d_in, int64_t n, const kss_pose& pose, const double (*cs)[6], int count, double* d_out) {
    if (n <= 0) return;
    for (int q0 = 0; q0 < count; q0 += POSE_MANY) {
        const int m = std::min(POSE_MANY, count - q0);
        PoseManyArgs a;
        for (int k = 0; k < 3; ++k) { a.shift[k] = pose.shift[k]; a.center[k] = pose.center[k]; }
        a.scale = pose.scale;
        for (int q = 0; q < m; ++q)
            for (int k = 0; k < 6; ++k) a.cs[q][k] = cs[q0 + q][k];
        hipLaunchKernelGGL(pose_apply_many_kernel, dim3(stream_blocks(n), m), dim3(256), 0, st, d_in, n, a, d_out + (int64_t)q0 * n * 3);
    }
}

struct M34 { float m[12]; };

__global__ __launch_bounds__(256) void transform_apply_f64_kernel(const double* __restrict__ in, int64_t n, M34 T,
                                                                  double* __restrict__ out) {
    // KSS_ICP.hpp:224-230: float coefficient x double coordinate, evaluated left to right in double
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double x = in[3 * i], y = in[3 * i + 1], z = in[3 * i + 2];
#pragma unroll
        for (int r = 0; r < 3; ++r)
            out[3 * i + r] = (((double)T.m[4 * r] * x + (double)T.m[4 * r + 1] * y) + (double)T.m[4 * r + 2] * z) + (double)T.m[4 * r + 3];
    }
}

void launch_transform_apply_f64(hipStream_t st, const float T[16], const double* d_in, int64_t n, double* d_out) {
    if (n <= 0) return;
    M34 m;
    for (int k = 0; k < 12; ++k) m.m[k] = T[k];
    hipLaunchKernelGGL(transform_apply_f64_kernel, dim3(stream_blocks(n)), dim3(256), 0, st, d_in, n, m, d_out);
}

__global__ __launch_bounds__(256) void transform_apply_f32_kernel(const float* __restrict__ in, int64_t n, M34 T,
                                                                  float* __restrict__ out) {
    // pcl transformCloud: pt_t = tr * (x, y, z, 1), Eigen order, float, no fma
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float x = in[3 * i], y = in[3 * i + 1], z = in[3 * i + 2];
#pragma unroll
        for (int r = 0; r < 3; ++r) out[3 * i + r] = ((T.m[4 * r] * x + T.m[4 * r + 1] * y) + T.m[4 * r + 2] * z) + T.m[4 * r + 3];
    }
}

void launch_transform_apply_f32(hipStream_t st, const float T[16], const float* d_in, int64_t n, float* d_out) {
    if (n <= 0) return;
    M34 m;
    for (int k = 0; k < 12; ++k) m.m[k] = T[k];
    hipLaunchKernelGGL(transform_apply_f32_kernel, dim3(stream_blocks(n)), dim3(256), 0, st, d_in, n, m, d_out);
}

// ---------------------------------------------------------------------------------------------
// farthest-point sampling (AIVS stand-in): ONE workgroup of 1024 lanes iterates m-1 times over the
// cloud, keeping min-distance-to-selected per point in global memory and an arg-max per iteration
// (wave shuffle -> LDS).  Ties -> lowest index, so the result is deterministic.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void fps_kernel(const double* __restrict__ xyz, int n, int m,
                                                   double* __restrict__ mind, int32_t* __restrict__ out_idx,
                                                   double* __restrict__ out_xyz) {
    __shared__ double s_val[16];
    __shared__ int s_idx[16];
    __shared__ int s_cur;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < n; i += 1024) mind[i] = __builtin_inf();
    int cur = 0;
    for (int k = 0; k < m; ++k) {
        if (tid == 0) {
            out_idx[k] = cur;
            out_xyz[3 * k] = xyz[3 * (int64_t)cur]; out_xyz[3 * k + 1] = xyz[3 * (int64_t)cur + 1]; out_xyz[3 * k + 2] = xyz[3 * (int64_t)cur + 2];
        }
        if (k + 1 == m) break;
        const double cx = xyz[3 * (int64_t)cur], cy = xyz[3 * (int64_t)cur + 1], cz = xyz[3 * (int64_t)cur + 2];
        double bv = -1.0;
        int bi = 0x7fffffff;
        for (int i = tid; i < n; i += 1024) {
            const double dx = xyz[3 * (int64_t)i] - cx, dy = xyz[3 * (int64_t)i + 1] - cy, dz = xyz[3 * (int64_t)i + 2] - cz;
            const double d = (dx * dx + dy * dy) + dz * dz;
            const double mnew = fmin(mind[i], d);
            mind[i] = mnew;
            if (mnew > bv) { bv = mnew; bi = i; }   // ascending i per lane: first maximum kept
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double ov = __shfl_down(bv, off, 64);
            const int oi = __shfl_down(bi, off, 64);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) { s_val[wave] = bv; s_idx[wave] = bi; }
        __syncthreads();
        if (tid == 0) {
            double v = s_val[0];
            int ix = s_idx[0];
            for (int w = 1; w < 16; ++w)
                if (s_val[w] > v || (s_val[w] == v && s_idx[w] < ix)) { v = s_val[w]; ix = s_idx[w]; }
            s_cur = ix;
        }
        __syncthreads();
        cur = s_cur;
        __syncthreads();
    }
}

void launch_fps(hipStream_t st, const double* d_xyz, int n, int m, double* d_mind, int32_t* d_idx, double* d_out) {
    hipLaunchKernelGGL(fps_kernel, dim3(1), dim3(1024), 0, st, d_xyz, n, m, d_mind, d_idx, d_out);
}

// ---------------------------------------------------------------------------------------------
// rotation search: grid = (source blocks, g^3 candidates).  Each lane rotates its pre-shaped
// source point in f64 by the candidate's Euler angles, narrows to f32 (:440-442), sweeps the
// whole target through LDS keeping only the minimum d2, and contributes sqrt((double)d2) (:444).
// ---------------------------------------------------------------------------------------------
template <int S, int NTH>
__global__ __launch_bounds__(NTH) void rot_search_kernel(const double* __restrict__ src, int ns,
                                                         const float4* __restrict__ tgt, int nt_pad,
                                                         const double* __restrict__ cs, int g,
                                                         double* __restrict__ partials) {
    // One lane = one source point under S consecutive candidates: the target tile read from LDS (one broadcast
    // ds_read_b128 per target) is shared by S distance evaluations, as the S sources per lane of nn_sweep_kernel.
    // NTH = lanes of the workgroup = sources per workgroup = targets per tile: 256, or 128 for the <= 2000-point samples of
    // KSSICP_Registration, whose sizes a 256-grain pads by up to a fifth (1406 sources: 6 x 256 = 1536, 11 x 128 = 1408;
    // same for the targets) and whose 1098 workgroups of 256 leave a fifth of the last round's CUs idle.
    __shared__ float4 tile[2][NTH];
    __shared__ double sh[4][S];
    const int g3 = g * g * g;
    const int tid = threadIdx.x;
    const int i = blockIdx.x * NTH + tid;
    const bool valid = i < ns;
    double x0 = 0.0, y0 = 0.0, z0 = 0.0;
    if (valid) { x0 = src[3 * (int64_t)i]; y0 = src[3 * (int64_t)i + 1]; z0 = src[3 * (int64_t)i + 2]; }
    float sx[S], sy[S], sz[S], best[S];
#pragma unroll
    for (int j = 0; j < S; ++j) {
        const int cand = min((int)blockIdx.y * S + j, g3 - 1);   // a surplus slot repeats the last candidate (not stored)
        const int ia = cand / (g * g), ib = (cand / g) % g, ic = cand % g;
        double x = x0, y = y0, z = z0;
        euler_rotate(x, y, z, cs[2 * ia], cs[2 * ia + 1], cs[2 * ib], cs[2 * ib + 1], cs[2 * ic], cs[2 * ic + 1]);
        sx[j] = (float)x; sy[j] = (float)y; sz[j] = (float)z;   // narrowed to float before the NN query (:440-442)
        best[j] = __builtin_inff();
    }
    const int ntiles = nt_pad / NTH;
    float4 pre = tgt[tid];
    int buf = 0;
    for (int t = 0; t < ntiles; ++t) {
        tile[buf][tid] = pre;
        __syncthreads();
        if (t + 1 < ntiles) pre = tgt[(t + 1) * NTH + tid];
        const float4* __restrict__ tl = tile[buf];
#pragma unroll 8
        for (int u = 0; u < NTH; ++u) {
            const float4 q = tl[u];
#pragma unroll
            for (int j = 0; j < S; ++j) best[j] = fminf(best[j], dist2<false>(sx[j], sy[j], sz[j], q.x, q.y, q.z));
        }
        buf ^= 1;
    }
    double acc[S];
#pragma unroll
    for (int j = 0; j < S; ++j) acc[j] = valid ? sqrt((double)best[j]) : 0.0;   // mean of sqrt(float d2) in double (:444-448)
    const double r = block_sum<S>(acc, sh);
    if (tid < S) {
        const int cand = (int)blockIdx.y * S + tid;
        if (cand < g3) partials[(int64_t)cand * gridDim.x + blockIdx.x] = r;
    }
}

// sources per workgroup (= targets per tile) of the rotation search: the grain that wastes less on padding and on the last
// round of workgroups.  KSS_ROT_NTH overrides (A/B).
int rot_search_grain(int64_t ns, int64_t nt, int g) {
    if (const char* e = getenv("KSS_ROT_NTH")) { const int v = atoi(e); if (v == 128 || v == 256) return v; }
    const double g3 = (double)g * g * g;
    double best_cost = 0.0;
    int best = 256;
    for (int nth : {256, 128}) {
        const double nsb = (double)((ns + nth - 1) / nth), ntb = (double)((nt + nth - 1) / nth);
        const double pad = (nsb * nth / (double)ns) * (ntb * nth / (double)nt);
        const double wgs = nsb * std::ceil(g3 / 4.0), per_round = 256.0 * (2048.0 / nth);   // (2048 lanes per CU at these register counts)
        const double rounds = wgs / per_round;
        const double cost = pad * std::ceil(rounds) / rounds * (nth == 128 ? 1.03 : 1.0);   // (a barrier per 128 targets instead of per 256)
        if (best_cost == 0.0 || cost < best_cost) { best_cost = cost; best = nth; }
    }
    return best;
}

void launch_rot_search(hipStream_t st, const double* d_src, int64_t ns, const float4* d_tgt4, int64_t nt_pad,
                       const double* d_cs, int g, double* d_partials, int n_src_blocks, int nth) {
    const int g3 = g * g * g;
    // candidates per lane: 4 when that still leaves >= 4 workgroups per CU, else 2 (the loop is VALU bound and wants a
    // few waves per SIMD to cover the LDS latency); KSS_ROT_S overrides for measurements
    int S = (int64_t)n_src_blocks * ((g3 + 3) / 4) >= 1024 ? 4 : 2;
    if (const char* e = getenv("KSS_ROT_S")) { const int v = atoi(e); if (v == 1 || v == 2 || v == 4) S = v; }
    const dim3 grid(n_src_blocks, (g3 + S - 1) / S);
#define KSS_ROT(SS, NN) hipLaunchKernelGGL((rot_search_kernel<SS, NN>), grid, dim3(NN), 0, st, d_src, (int)ns, d_tgt4, (int)nt_pad, d_cs, g, d_partials)
    if (nth == 128) { if (S == 4) KSS_ROT(4, 128); else if (S == 2) KSS_ROT(2, 128); else KSS_ROT(1, 128); }
    else            { if (S == 4) KSS_ROT(4, 256); else if (S == 2) KSS_ROT(2, 256); else KSS_ROT(1, 256); }
#undef KSS_ROT
}

}  // namespace kss
