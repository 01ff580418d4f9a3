/*
 * kssicp.h -- C-ABI of the MI355X-native KSS-ICP registration core (libkssicp.so).
 *
 * This is the drop-in boundary UNDER the reference's C++ class surface: the mirror classes in
 * include/KSS_ICP.hpp, include/initRegistrationKSS.hpp and include/registrationMeasure.hpp
 * (same class / method / field names as the reference) call only these entry points.  The
 * reference has no FFI of its own; each entry point below cites the
 * reference code it replaces, as path:line under PS_AIS_Simplification/.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch types.  Every call returns an int
 *     status (KSS_OK == 0, errors < 0) and never throws across the ABI.
 *   - Clouds are packed xyz triples: float[n][3] ("f32") or double[n][3] ("f64").
 *   - `*_dev` variants take DEVICE pointers (hipMalloc'd, or a torch tensor's data_ptr());
 *     the un-suffixed variants take HOST pointers and stage through the context's workspace.
 *   - One context per host thread / GPU.  Calls on one context are serialised on its HIP
 *     stream; distinct contexts are independent (matches the reference's "objects are
 *     independent" threading rule).
 *   - There is NO CPU fallback: every compute entry point fails with KSS_ERR_NODEVICE /
 *     KSS_ERR_HIP when no gfx950 device is usable.
 */
#ifndef KSSICP_H_
#define KSSICP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KSS_VERSION 100 /* 0.1.0 */

enum {
    KSS_OK = 0,
    KSS_ERR_ARG = -1,      /* null pointer, negative size, n == 0 where a cloud is required */
    KSS_ERR_HIP = -2,      /* a HIP runtime call failed; see kss_last_error() */
    KSS_ERR_NOMEM = -3,
    KSS_ERR_NODEVICE = -4, /* no usable GPU */
    KSS_ERR_CAPACITY = -5, /* caller-provided output buffer too small */
    KSS_ERR_RCCL = -6
};

enum { KSS_F32 = 0, KSS_F64 = 1 };

typedef struct kss_ctx kss_ctx;

int         kss_version(void);
const char *kss_status_string(int status);
/* human-readable detail of the last failure on this context ("" if none) */
const char *kss_last_error(const kss_ctx *ctx);

/* ---- context ---------------------------------------------------------------------------- */
int   kss_ctx_create(int device_id, kss_ctx **out);                 /* owns its HIP stream */
/* borrow an existing hipStream_t (e.g. torch.cuda.current_stream().cuda_stream) */
int   kss_ctx_create_on_stream(int device_id, void *hip_stream, kss_ctx **out);
int   kss_ctx_destroy(kss_ctx *ctx);
int   kss_ctx_synchronize(kss_ctx *ctx);
void *kss_ctx_stream(kss_ctx *ctx);

/* default NN search structure of this context (KSS_NN_AUTO / _BRUTE / _GRID, see below); used whenever an
 * entry point has no nn_mode of its own or it is KSS_NN_AUTO.  Results never depend on it. */
int   kss_ctx_set_nn_mode(kss_ctx *ctx, int nn_mode);

/* per-kernel timing with HIP events recorded on the context's stream around launches of the named kernel
 * class (used by bench.py for the roofline object).  kss_profile_enable(ctx, n): 0 = off, 1 = every launch,
 * n > 1 = every n-th launch of each class (an event pair costs a few microseconds, comparable to the fused
 * cell-list launch it brackets); kss_profile_get reports the sampled launches only. */
enum { KSS_K_NN_SWEEP = 0, KSS_K_CORR_REDUCE = 1, KSS_K_PRESHAPE = 2, KSS_K_ROT_SEARCH = 3,
       KSS_K_POSE_APPLY = 4, KSS_K_GRID_NN = 5, KSS_K_GRID_BUILD = 6,
       KSS_K_GRID_CHAIN = 7,        /* chained launches of the fused pass: one entry per launch (its whole duration) */
       KSS_K_GRID_CHAIN_PASS = 8,   /* the same time, counted per ICP pass the launches ran */
       KSS_K_RESIDENT = 9,          /* pair-resident batch kernel: one entry per launch (all passes of every pair of the batch) */
       KSS_K_RESIDENT_PASS = 10,    /* the same time, counted per (pair, pass) the launches ran */
       KSS_K_COUNT = 11 };
int kss_profile_enable(kss_ctx *ctx, int on);
/* geometry of the last cell list built on this context (KSS_NN_GRID) and, when profiling is enabled, the
 * number of (source, target) distance evaluations of one search pass at the sources' initial positions:
 * out = {cell edge h, gx, gy, gz, occupied cells, evaluations per pass, n_src, n_tgt}.  All zero if no
 * grid has been built.  Used by bench.py to state the kernel's algorithmic bytes. */
int kss_grid_stats(kss_ctx *ctx, double out[8]);
int kss_profile_reset(kss_ctx *ctx);
/* synchronises the stream; total_ms = sum of event-timed durations, launches = count */
int kss_profile_get(kss_ctx *ctx, int kernel_class, double *total_ms, int64_t *launches);
/* what a HIP event pair reads around an EMPTY kernel launched into the idle stream (ms, mean of 200): an upper
 * bound of what event timing adds to a short kernel's own duration.  bench.py reports it beside its per-launch
 * durations (context for comparing them with rocprofv3's kernel trace); it is not subtracted. */
int kss_profile_event_overhead(kss_ctx *ctx, double *ms);

/* ---- (a2) KSS pre-shape: initRegistration_MiddleAlign, initRegistrationKSS.hpp:144-207 ----
 * centroid (mean of points) and mean distance to the centroid, accumulated in f64 with a
 * wavefront/LDS block reduce.  scale = r_tgt / r_src is formed by the caller (:209). */
int kss_preshape_stats(kss_ctx *ctx, const void *xyz, int dtype, int64_t n,
                       double centroid[3], double *mean_radius);
int kss_preshape_stats_dev(kss_ctx *ctx, const void *d_xyz, int dtype, int64_t n,
                           double centroid[3], double *mean_radius);
/* both clouds of a registration (:144-207 computes S and T back to back) in ONE call: two launches for the pair, no
 * stream synchronisation; bit-identical to one call per cloud.  d_tgt may be NULL (then c_tgt / r_tgt are not written). */
int kss_preshape_stats_pair_dev(kss_ctx *ctx, const void *d_src, int64_t ns, const void *d_tgt, int64_t nt, int dtype,
                                double c_src[3], double *r_src, double c_tgt[3], double *r_tgt);

/* ---- (a3,a7) pose application: initRegistration_Rotation[_Angle], :75-109 + :365-404 ----
 * p += shift; p = center + (p - center) * scale; then Rx(angle[0]), Ry(angle[1]), Rz(angle[2])
 * about the WORLD ORIGIN, all in f64 without fused multiply-add (reference arithmetic). */
typedef struct {
    double shift[3];   /* x_middle..   = c_T - c_S   (:190-192) */
    double center[3];  /* x_middle_S.. = c_T         (:186-188) */
    double scale;      /*                             (:209)     */
    double angle[3];   /* Euler angles about x, y, z              */
} kss_pose;
int kss_pose_apply(kss_ctx *ctx, const double *in, int64_t n, const kss_pose *pose, double *out);
int kss_pose_apply_dev(kss_ctx *ctx, const double *d_in, int64_t n, const kss_pose *pose, double *d_out);

/* ---- (a8) exact 1-NN correspondence: replaces pcl::KdTreeFLANN<PointXYZ>::nearestKSearch
 *      (call sites initRegistrationKSS.hpp:236,443; registrationMeasure.hpp:63,79; PCL ICP) ----
 * Brute-force LDS-tiled source x target sweep in f32.  d2 = (dx*dx + dy*dy) + dz*dz without
 * fma (FLANN L2_Simple<float>), ties -> lowest target index.  idx/d2 may be NULL. */
int kss_nn(kss_ctx *ctx, const float *src, int64_t ns, const float *tgt, int64_t nt,
           int32_t *idx, float *d2);
int kss_nn_dev(kss_ctx *ctx, const float *d_src, int64_t ns, const float *d_tgt, int64_t nt,
               int32_t *d_idx, float *d_d2);

/* ---- exact k-NN (k <= 64): pcl::KdTreeFLANN::nearestKSearch with K > 1 (ballRegionCompute.hpp:499 K = 13,
 *      Method_AIVS_SimPro.hpp:904 K = 3, Method_Octree.hpp:137, pcl::NormalEstimation K = 20) ----
 * idx / d2: nq * k entries, per query in ascending (d2, index) order; slots beyond nt hold -1 / +inf. */
int kss_knn(kss_ctx *ctx, const float *query, int64_t nq, const float *tgt, int64_t nt, int k, int32_t *idx, float *d2);
int kss_knn_dev(kss_ctx *ctx, const float *d_query, int64_t nq, const float *d_tgt, int64_t nt, int k, int32_t *d_idx, float *d_d2);

/* ---- surface normals: estimateNormal_PCL_MP_return, normalCompute.hpp:308-355 ----
 * pcl::NormalEstimationOMP semantics (k nearest neighbours incl. the point itself, float single-pass covariance,
 * closed-form smallest eigenvector, flipped towards the view point (0,0,0)) and the reference's renormalisation in
 * double.  normals: n * 3 doubles.  The reference uses k = 20. */
int kss_normals(kss_ctx *ctx, const double *pts, int64_t n, int k, double *normals);
/* estimateNormal_RegularNormal, normalCompute.hpp:614-742 (with kss_normals: estimateNormal_PCL_MP, :358-403):
 * consistent orientation of given normals (n*3 doubles, in place) by level-synchronous propagation over the 8-NN
 * graph from point 0; a normal is negated when it points away from its parent's.  The 8-NN search runs on the
 * device, the O(8 n) propagation on the host.  Unreached points keep their normals. */
int kss_normals_orient(kss_ctx *ctx, const double *pts, int64_t n, double *normals);

/* ---- (a10) correspondence sums for TransformationEstimationSVD / umeyama (inside PCL ICP) ----
 * sums[0]=n kept (d2 <= max_d2), [1..3]=sum src, [4..6]=sum tgt[idx], [7..15]=sum src_i*tgt_j
 * (i major), [16]=sum d2 kept, [17]=sum d2 all, [18]=sum sqrt(d2) all, [19]=0.  f64, reduced
 * in a fixed order (bitwise reproducible run to run). */
#define KSS_NSUMS 20
int kss_cov(kss_ctx *ctx, const float *src, const float *tgt, const int32_t *idx, int64_t n,
            int64_t nt, double max_d2, double sums[KSS_NSUMS]);
int kss_cov_dev(kss_ctx *ctx, const float *d_src, const float *d_tgt, const int32_t *d_idx, int64_t n,
                int64_t nt, double max_d2, double sums[KSS_NSUMS]);
/* host-side: rigid transform (Umeyama without scaling, 3x3 SVD) from the sums; row-major 4x4 */
int kss_rigid_from_sums(const double sums[KSS_NSUMS], float T[16]);

/* ---- (a4,a5) rotation search: initRegistration_Rotation(), :222-296 + :430-450 ----
 * src_preshaped: S' after the similarity of kss_pose_apply with angle = 0 (f64);
 * err receives value[g][g][g] (i-major), g = trip count of for(a=0; a<6.3; a+=6.3/step). */
int kss_rotation_search(kss_ctx *ctx, const double *src_preshaped, int64_t ns,
                        const double *tgt, int64_t nt, double step,
                        double *err, int64_t err_capacity, int *g_out);
int kss_rotation_search_dev(kss_ctx *ctx, const double *d_src_preshaped, int64_t ns,
                            const double *d_tgt, int64_t nt, double step,
                            double *err /* host */, int64_t err_capacity, int *g_out);
/* host-side (a6): accumulated grid angles (:245), global arg-min (:258-265, :291-293) and the
 * 5^3 clamped local-minimum list (:276-289, :481-522).  angle_list = idx*6.3/step triples. */
int kss_grid_angles(double step, double *angles, int capacity);
int kss_rotation_candidates(const double *err, int g, double step, double best_angle[3],
                            double *angle_list, int list_capacity, int *n_list);

/* ---- (a9,a11,a12) ICP driver: pcl::IterativeClosestPoint::align as configured at
 *      KSS_ICP.hpp:155-162 (x5), PCL 1.8.1 semantics ---- */
/* In-place sum over the ranks of a job of n doubles in HOST memory; returns 0 on success.  See `allreduce` below. */
typedef int (*kss_allreduce_fn)(void *user, double *values, int n);

typedef struct {
    int    max_iterations;             /* setMaximumIterations            KSS_ICP.hpp:159 */
    double max_corr_dist;              /* setMaxCorrespondenceDistance(1) :156 */
    double transformation_epsilon;     /* setTransformationEpsilon(1e-10) :157 */
    double euclidean_fitness_epsilon;  /* setEuclideanFitnessEpsilon(1e-3):158 */
    double abs_mse_epsilon;            /* PCL default 1e-12 */
    int    min_correspondences;        /* PCL default 3 */
    int    fixed_iterations;           /* run exactly max_iterations (benchmark mode) */
    int    nn_fma;                     /* 0: reference arithmetic; 1: fused d2 (fast, not bit-parity) */
    int    compute_fitness;            /* getFitnessScore() after align (:164) */
    int    nn_sources_per_thread;      /* tuning, 0 = auto */
    int    nn_target_splits;           /* tuning, 0 = auto */
    int    nn_mode;                    /* KSS_NN_AUTO / KSS_NN_BRUTE / KSS_NN_GRID: same results bit for bit */
    /* optional per-iteration trace for parity tests (host pointers, may be NULL) */
    double *trace_sums;                /* trace_cap * KSS_NSUMS */
    float  *trace_Tk;                  /* trace_cap * 16 */
    int     trace_cap;
    int    *trace_n;
    /* optional: the correspondences getFitnessScore() sums over -- nearest target index and squared distance of every
     * source point of pair 0 under the final transform (host pointers, n_src entries each, may be NULL; filled only
     * when compute_fitness is set) */
    int32_t *fitness_idx;
    float   *fitness_d2;
    /* optional (the single-pair exchange step): ONE registration whose SOURCE ROWS are split over several
     * ranks, target replicated.  Every rank calls kss_icp[_dev] with its own rows and the same target and parameters;
     * after each NN pass the KSS_NSUMS correspondence sums are summed over the ranks through this callback, so every
     * rank solves the same 3x3 system and applies the same transform.  The result (T, iterations, fitness over ALL
     * source rows) is identical on every rank.  NULL = single-rank registration.  Single pair only. */
    kss_allreduce_fn allreduce;
    void *allreduce_user;
} kss_icp_params;

/* NN search structure.  BRUTE: the LDS-tiled source x target sweep (north star).  GRID: exact search
 * through a uniform cell list built once per target, queries unresolved within a few cell shells fall
 * back to the brute-force sweep.  AUTO picks GRID for one large pair, BRUTE otherwise. */
enum { KSS_NN_AUTO = 0, KSS_NN_BRUTE = 1, KSS_NN_GRID = 2 };

enum { KSS_STATE_NOT_CONVERGED = 0, KSS_STATE_ITERATIONS = 1, KSS_STATE_TRANSFORM = 2,
       KSS_STATE_ABS_MSE = 3, KSS_STATE_REL_MSE = 4, KSS_STATE_NO_CORRESPONDENCES = 5 };

typedef struct {
    float   T[16];       /* getFinalTransformation(), row-major Matrix4f  (:222) */
    double  fitness;     /* getFitnessScore()                             (:164) */
    double  last_mse;
    int32_t iterations;
    int32_t converged;   /* hasConverged() */
    int32_t state;
    int32_t pair_id;     /* index of the pair in a batch (global id after a gather) */
} kss_icp_result;       /* 96 bytes: the record gathered over RCCL */

int kss_icp_default_params(kss_icp_params *p);
int kss_icp(kss_ctx *ctx, const float *src, int64_t ns, const float *tgt, int64_t nt,
            const kss_icp_params *p, kss_icp_result *res);
int kss_icp_dev(kss_ctx *ctx, const float *d_src, int64_t ns, const float *d_tgt, int64_t nt,
                const kss_icp_params *p, kss_icp_result *res);
/* batch of independent registrations (KSS_ICP.hpp:102-118 candidate loop; configs C3/C5).
 * src_off/tgt_off: npairs+1 HOST offsets (in points) into the packed clouds. */
int kss_icp_batch(kss_ctx *ctx, const float *src_all, const int64_t *src_off,
                  const float *tgt_all, const int64_t *tgt_off, int npairs,
                  const kss_icp_params *p, kss_icp_result *results);
int kss_icp_batch_dev(kss_ctx *ctx, const float *d_src_all, const int64_t *src_off,
                      const float *d_tgt_all, const int64_t *tgt_off, int npairs,
                      const kss_icp_params *p, kss_icp_result *results);

/* ---- (a13) apply the ICP Matrix4f to a full-resolution f64 cloud, KSS_ICP.hpp:224-230 ---- */
int kss_transform_apply(kss_ctx *ctx, const float T[16], const double *in, int64_t n, double *out);
int kss_transform_apply_dev(kss_ctx *ctx, const float T[16], const double *d_in, int64_t n, double *d_out);

/* pcl transformCloud / the PCL `output` cloud of align(): Matrix4f x float point, float result
 * (KSS_ICP.hpp:162,170-180 read it back as pointAlign in the full-resolution overload) */
int kss_transform_apply_f32(kss_ctx *ctx, const float T[16], const float *in, int64_t n, float *out);

/* ---- plain farthest-point sampling: fallback for clouds kss_downsample_aivs rejects (zero extent) ----
 * Exact farthest-point sampling in f64 starting from point 0 (ties -> lowest index); returns the m
 * selected points in selection order.  Not an AIVS restatement. */
int kss_downsample_fps(kss_ctx *ctx, const double *xyz, int64_t n, int64_t m, double *out, int32_t *out_idx);

/* ---- AIVS down-sampler: pointPipeline_init_point_withoutUniform + BallRegion_init_withoutNormal +
 *      AIVS_Pro_init + AIVS_simplification(point_num), KSS_ICP.hpp:71-81 (Method_AIVS_SimPro.hpp:94-154,
 *      ballRegionCompute.hpp:114-147) ----
 * Voxel grid, per-voxel farthest-point sampling in the reference's 8-colour order (one launch per colour, one
 * wavefront per voxel) and the accurate cut, with the reference's arithmetic.  out must hold capacity points;
 * *n_out receives the number selected, which can differ from point_num exactly as in the reference (fewer if the
 * per-voxel budgets add up to less; more if the accurate cut runs out of live closest pairs).  out_idx (may be
 * NULL) receives the indices into xyz.  KSS_ERR_ARG for degenerate (zero-extent / planar) clouds, which divide by
 * zero in the reference; KSS_ERR_CAPACITY if capacity is too small. */
int kss_downsample_aivs(kss_ctx *ctx, const double *xyz, int64_t n, int64_t point_num, double *out, int64_t capacity,
                        int64_t *n_out, int32_t *out_idx);

/* Both clouds of a registration at once (KSS_ICP.hpp:71-81 down-samples the target, then the source; the two do not depend
 * on each other): cloud 1 runs on a worker context of ctx, on a thread of its own, while the calling thread does cloud 0.
 * AIVS on a few thousand points is a chain of ~20 small launches and four host syncs -- two of them overlap almost
 * completely.  Same selections as two kss_downsample_aivs calls.  rc[k] receives cloud k's status (the codes of
 * kss_downsample_aivs: a degenerate cloud is KSS_ERR_ARG for that cloud only); the return value is KSS_OK unless an argument
 * is bad or the worker context cannot be created. */
int kss_downsample_aivs_pair(kss_ctx *ctx, const double *xyz0, int64_t n0, int64_t point_num0, double *out0, int64_t capacity0,
                             int64_t *n_out0, int32_t *out_idx0, const double *xyz1, int64_t n1, int64_t point_num1, double *out1,
                             int64_t capacity1, int64_t *n_out1, int32_t *out_idx1, int rc[2]);

/* ---- (8f #3) octree down-sampler: PCL_octree::PCL_Octree_Simplification_WithOutNormal, Method_Octree.hpp:77-165 ----
 * resolution = mean distance of the first 1000 points to their kn-th nearest point (kn = 2 below 80000 points, else
 * 7 * (n / 80000) capped at 35); occupied voxels of a pcl::octree::OctreePointCloudSearch of that resolution (PCL
 * 1.8.1 bounding-cube rules, insertion order) in depth-first order; every voxel centre replaced by its nearest
 * cloud point (ties -> lowest index).  out_idx receives one point index per voxel (repeats possible, as in the
 * reference); *n_out the voxel count.  Needs n >= 1000.  PCL is absent from the reference tree: parity unpinned. */
int kss_downsample_octree(kss_ctx *ctx, const double *xyz, int64_t n, int32_t *out_idx, int64_t capacity,
                          int64_t *n_out, double *resolution_out /* may be NULL */);

/* ---- PCR_QM: registrationMeasure.hpp:47-98 -> out = {MSE, RMSE, MAE} ---- */
int kss_pcr_qm(kss_ctx *ctx, const double *aligned, int64_t na, const double *tmpl, int64_t nt,
               double out[3]);

/* ---- (a16) KSSICP_Registration on already down-sampled clouds, KSS_ICP.hpp:86-131 + :185-233 */
typedef struct {
    double  scale;
    double  angle[3];         /* chosen Euler angles */
    double  R[9], t[3];       /* composite similarity p' = scale*R*p + t */
    double  c_src[3], c_tgt[3]; /* pre-shape centroids of S', T' (x_middle_S.. = c_tgt, x_middle.. = c_tgt - c_src) */
    float   T_icp[16];
    double  E_d_init;         /* :93 */
    double  final_fitness;    /* :130 */
    int32_t used_angle_list;  /* :99 branch */
    int32_t angle_index;
    int32_t n_angle_list;
    int32_t icp_iterations;
    int32_t icp_converged;
    int32_t grid;             /* g */
} kss_register_result;
int kss_register(kss_ctx *ctx, const double *src_sub, int64_t nss, const double *tgt_sub, int64_t nts,
                 const double *src_full, int64_t nsf, double accurate, int iter,
                 double *point_align /* nsf*3, may be NULL */, kss_register_result *res);

/* ---- many full KSS registrations (configs C3 / C5 read as registrations, not bare ICPs): for every pair the whole
 * KSSICP_init + KSSICP_Registration sequence (KSS_ICP.hpp:53-131) -- pNumber = min(n_S, n_T) / 2 capped at sample_cap
 * (2000 in the reference, :57-63), AIVS down-sampling of both clouds (farthest-point sampling when a cloud cannot be
 * voxelised), kss_register.  A registration is a chain of small launches that leaves most of the GPU idle, so the pairs
 * are spread over `workers` host threads, each with its own context / stream on the same device (0 = pick a default);
 * every pair is handled by exactly one worker with the same code as the one-pair path, so results[i] does not depend
 * on the worker count.  src_off / tgt_off are point offsets (npairs + 1 entries) into packed double[n][3] arrays.
 * point_align_all (may be NULL) receives the aligned full-resolution sources, laid out like src_all. */
int kss_register_batch(kss_ctx *ctx, const double *src_all, const int64_t *src_off, const double *tgt_all,
                       const int64_t *tgt_off, int npairs, int64_t sample_cap, double accurate, int iter, int workers,
                       double *point_align_all, kss_register_result *results);

/* ---- (8e) RCCL-backed kss_allreduce_fn: user = &kss_rccl_link{ctx, ncclComm_t}; ncclAllReduce(sum, f64) on the
 * context's stream between a host->device and a device->host copy of the n doubles (160 B per ICP iteration:
 * latency bound, one collective per iteration) ---- */
typedef struct { kss_ctx *ctx; void *rccl_comm; } kss_rccl_link;
int kss_rccl_allreduce_sum(void *user /* kss_rccl_link* */, double *values, int n);

/* ---- (8e) gather of per-pair result records over RCCL (ncclComm_t passed as void*) ----
 * all must hold world_size * n_local records; every rank receives every record. */
int kss_gather_results(kss_ctx *ctx, void *rccl_comm, int world_size,
                       const kss_icp_result *local, int n_local, kss_icp_result *all);

#ifdef __cplusplus
}
#endif
#endif /* KSSICP_H_ */
