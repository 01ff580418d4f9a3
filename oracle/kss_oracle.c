/*
 * oracle/kss_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * Plain-C restatement of the reference registration hot path.  See kss_oracle.h for the
 * parity-pin status.  Build: oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp).
 * -ffp-contract=off is REQUIRED: the reference arithmetic (MSVC /fp:precise, FLANN
 * L2_Simple<float>) never fuses multiply-add.
 *
 * Citations: path:line under /root/reference/PS_AIS_Simplification/.
 */
#include "kss_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

void ko_free(void *p) { free(p); }

/* ------------------------------------------------------------------------------------ */
/* KSS pre-shape -- initRegistrationKSS.hpp:144-220                                      */
/* ------------------------------------------------------------------------------------ */
static void centroid_and_radius(const double *P, int64_t n, double c[3], double *r) {
    /* :146-157 serial sums then divide by size */
    double xs = 0, ys = 0, zs = 0;
    for (int64_t i = 0; i < n; i++) {
        xs = xs + P[3 * i + 0];
        ys = ys + P[3 * i + 1];
        zs = zs + P[3 * i + 2];
    }
    xs = xs / (double)n; ys = ys / (double)n; zs = zs / (double)n;
    /* :160-171 mean distance to the centroid (the max variant is commented out there) */
    double acc = 0;
    for (int64_t i = 0; i < n; i++) {
        double xl = P[3 * i + 0] - xs, yl = P[3 * i + 1] - ys, zl = P[3 * i + 2] - zs;
        acc = acc + sqrt(xl * xl + yl * yl + zl * zl);
    }
    c[0] = xs; c[1] = ys; c[2] = zs;
    *r = acc / (double)n;
}

void ko_preshape_stats(const double *S, int64_t ns, const double *T, int64_t nt, ko_preshape *o) {
    centroid_and_radius(S, ns, o->c_src, &o->r_src);
    centroid_and_radius(T, nt, o->c_tgt, &o->r_tgt);
    for (int k = 0; k < 3; k++) o->shift[k] = o->c_tgt[k] - o->c_src[k]; /* :190-192 */
    o->scale = o->r_tgt / o->r_src;                                     /* :209 */
}

void ko_similarity_apply(const double *in, int64_t n, const ko_preshape *ps, double *out) {
    /* :212-219 (and :77-84, :95-102): p += shift; p = c_T + (p - c_T) * scale */
    for (int64_t i = 0; i < n; i++)
        for (int k = 0; k < 3; k++) {
            double v = in[3 * i + k] + ps->shift[k];
            out[3 * i + k] = ps->c_tgt[k] + (v - ps->c_tgt[k]) * ps->scale;
        }
}

void ko_axis_rotate(int cord, double angle, double *p, int64_t n) {
    /* :365-404; the reference re-evaluates cos/sin per point, the value is the same */
    const double c = cos(angle), s = sin(angle);
    if (cord == 1) {
        for (int64_t i = 0; i < n; i++) {
            double y = p[3 * i + 1], z = p[3 * i + 2];
            p[3 * i + 1] = y * c - z * s;
            p[3 * i + 2] = y * s + z * c;
        }
    } else if (cord == 2) {
        for (int64_t i = 0; i < n; i++) {
            double x = p[3 * i + 0], z = p[3 * i + 2];
            p[3 * i + 0] = z * s + x * c;
            p[3 * i + 2] = z * c - x * s;
        }
    } else {
        for (int64_t i = 0; i < n; i++) {
            double x = p[3 * i + 0], y = p[3 * i + 1];
            p[3 * i + 0] = x * c - y * s;
            p[3 * i + 1] = x * s + y * c;
        }
    }
}

void ko_pose_apply(const double *in, int64_t n, const ko_preshape *ps, const double angle[3], double *out) {
    /* :75-109 */
    ko_similarity_apply(in, n, ps, out);
    ko_axis_rotate(1, angle[0], out, n);
    ko_axis_rotate(2, angle[1], out, n);
    ko_axis_rotate(3, angle[2], out, n);
}

/* ------------------------------------------------------------------------------------ */
/* exact float 1-NN                                                                      */
/* ------------------------------------------------------------------------------------ */
static inline float dist2(const float *a, const float *b, int fma) {
    float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    if (fma) return fmaf(dx, dx, fmaf(dy, dy, dz * dz));
    return (dx * dx + dy * dy) + dz * dz; /* FLANN L2_Simple: result += diff*diff, in order */
}

void ko_nn_brute(const float *q, int64_t nq, const float *t, int64_t nt, int fma, int32_t *idx, float *d2) {
#pragma omp parallel for schedule(static) if (nq >= 1024)
    for (int64_t i = 0; i < nq; i++) {
        float best = INFINITY; int32_t bi = -1;
        for (int64_t j = 0; j < nt; j++) {
            float d = dist2(q + 3 * i, t + 3 * j, fma);
            if (d < best) { best = d; bi = (int32_t)j; } /* strict <: lowest index wins ties */
        }
        idx[i] = bi; d2[i] = best;
    }
}

/* ---- kd-tree (role of FLANN KDTreeSingleIndex, leaf 15, exact search) ---- */
typedef struct { int32_t left, right; /* children, or for a leaf: left=-1, lo/hi below */
                 int32_t lo, hi; int32_t dim; float divlow, divhigh; } kd_node;
struct ko_kdtree {
    int64_t n; int leaf;
    float *pts; int32_t *orig;
    kd_node *nodes; int32_t nnodes, capnodes;
    float bbox[6];
};

static void kd_select(int32_t *perm, const float *t, int dim, int64_t lo, int64_t hi, int64_t k) {
    /* quickselect so that perm[k] holds the k-th smallest coordinate in [lo,hi) */
    hi--;
    while (lo < hi) {
        float pivot = t[3 * (int64_t)perm[(lo + hi) / 2] + dim];
        int64_t i = lo, j = hi;
        while (i <= j) {
            while (t[3 * (int64_t)perm[i] + dim] < pivot) i++;
            while (t[3 * (int64_t)perm[j] + dim] > pivot) j--;
            if (i <= j) { int32_t tmp = perm[i]; perm[i] = perm[j]; perm[j] = tmp; i++; j--; }
        }
        if (k <= j) hi = j; else if (k >= i) lo = i; else return;
    }
}

static int32_t kd_build_rec(ko_kdtree *kt, int32_t *perm, const float *t, int64_t lo, int64_t hi) {
    if (kt->nnodes == kt->capnodes) {
        kt->capnodes = kt->capnodes * 2 + 16;
        kt->nodes = (kd_node *)realloc(kt->nodes, sizeof(kd_node) * (size_t)kt->capnodes);
    }
    int32_t id = kt->nnodes++;
    kd_node nd; memset(&nd, 0, sizeof nd);
    nd.lo = (int32_t)lo; nd.hi = (int32_t)hi; nd.left = nd.right = -1;
    if (hi - lo <= kt->leaf) { kt->nodes[id] = nd; return id; }
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t i = lo; i < hi; i++)
        for (int k = 0; k < 3; k++) {
            float v = t[3 * (int64_t)perm[i] + k];
            if (v < mn[k]) mn[k] = v;
            if (v > mx[k]) mx[k] = v;
        }
    int dim = 0; float span = mx[0] - mn[0];
    for (int k = 1; k < 3; k++) if (mx[k] - mn[k] > span) { span = mx[k] - mn[k]; dim = k; }
    int64_t mid = (lo + hi) / 2;
    kd_select(perm, t, dim, lo, hi, mid);
    float dl = -INFINITY, dh = INFINITY;
    for (int64_t i = lo; i < mid; i++) { float v = t[3 * (int64_t)perm[i] + dim]; if (v > dl) dl = v; }
    for (int64_t i = mid; i < hi; i++) { float v = t[3 * (int64_t)perm[i] + dim]; if (v < dh) dh = v; }
    nd.dim = dim; nd.divlow = dl; nd.divhigh = dh;
    int32_t l = kd_build_rec(kt, perm, t, lo, mid);
    int32_t r = kd_build_rec(kt, perm, t, mid, hi);
    nd.left = l; nd.right = r;
    kt->nodes[id] = nd;
    return id;
}

ko_kdtree *ko_kdtree_build(const float *t, int64_t nt, int leaf_size) {
    ko_kdtree *kt = (ko_kdtree *)calloc(1, sizeof *kt);
    kt->n = nt; kt->leaf = leaf_size > 0 ? leaf_size : 15;
    int32_t *perm = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nt > 0 ? nt : 1));
    for (int64_t i = 0; i < nt; i++) perm[i] = (int32_t)i;
    for (int k = 0; k < 3; k++) { kt->bbox[k] = INFINITY; kt->bbox[3 + k] = -INFINITY; }
    for (int64_t i = 0; i < nt; i++)
        for (int k = 0; k < 3; k++) {
            float v = t[3 * i + k];
            if (v < kt->bbox[k]) kt->bbox[k] = v;
            if (v > kt->bbox[3 + k]) kt->bbox[3 + k] = v;
        }
    if (nt > 0) kd_build_rec(kt, perm, t, 0, nt);
    kt->pts = (float *)malloc(sizeof(float) * 3 * (size_t)(nt > 0 ? nt : 1));
    for (int64_t i = 0; i < nt; i++) memcpy(kt->pts + 3 * i, t + 3 * (int64_t)perm[i], 3 * sizeof(float));
    kt->orig = perm;
    return kt;
}

void ko_kdtree_free(ko_kdtree *kt) {
    if (!kt) return;
    free(kt->pts); free(kt->orig); free(kt->nodes); free(kt);
}

typedef struct { const ko_kdtree *kt; const float *q; int fma; float best; int32_t bi; } kd_query;

static void kd_search(kd_query *Q, int32_t node, float off[3]) {
    const kd_node *nd = &Q->kt->nodes[node];
    if (nd->left < 0) {
        const float *p = Q->kt->pts + 3 * (int64_t)nd->lo;
        for (int32_t i = nd->lo; i < nd->hi; i++, p += 3) {
            float d = dist2(Q->q, p, Q->fma);
            int32_t oi = Q->kt->orig[i];
            if (d < Q->best || (d == Q->best && oi < Q->bi)) { Q->best = d; Q->bi = oi; }
        }
        return;
    }
    int dim = nd->dim;
    float val = Q->q[dim];
    float d1 = val - nd->divlow, d2 = val - nd->divhigh;
    int32_t nearc, farc; float cut;
    if (d1 + d2 < 0) { nearc = nd->left; farc = nd->right; cut = d2 * d2; }
    else { nearc = nd->right; farc = nd->left; cut = d1 * d1; }
    kd_search(Q, nearc, off);
    float save = off[dim];
    off[dim] = cut;
    /* canonical-order lower bound: every point p of the far child has
       fl((q-p)_d^2) >= off[d] per dimension, and float addition is monotone, so
       (off0+off1)+off2 <= (dx2+dy2)+dz2.  '<=' keeps equal-distance (tie) candidates. */
    float mind = (off[0] + off[1]) + off[2];
    if (Q->fma) mind *= 0.99999f; /* fused form rounds differently: relax the bound */
    if (mind <= Q->best) kd_search(Q, farc, off);
    off[dim] = save;
}

static void kd_nn_one(const ko_kdtree *kt, const float *q, int fma, int32_t *idx, float *d2) {
    kd_query Q; Q.kt = kt; Q.q = q; Q.fma = fma; Q.best = INFINITY; Q.bi = INT32_MAX;
    float off[3];
    for (int k = 0; k < 3; k++) {
        float o = 0.f;
        if (q[k] < kt->bbox[k]) o = q[k] - kt->bbox[k];
        else if (q[k] > kt->bbox[3 + k]) o = q[k] - kt->bbox[3 + k];
        off[k] = o * o;
    }
    if (kt->n > 0) kd_search(&Q, 0, off);
    *idx = kt->n > 0 ? Q.bi : -1; *d2 = Q.best;
}

void ko_kdtree_nn(const ko_kdtree *kt, const float *q, int64_t nq, int fma, int nthreads, int32_t *idx, float *d2) {
    if (nthreads <= 1) {
        for (int64_t i = 0; i < nq; i++) kd_nn_one(kt, q + 3 * i, fma, idx + i, d2 + i);
    } else {
#pragma omp parallel for schedule(dynamic, 256) num_threads(nthreads)
        for (int64_t i = 0; i < nq; i++) kd_nn_one(kt, q + 3 * i, fma, idx + i, d2 + i);
    }
}

/* ------------------------------------------------------------------------------------ */
/* rotation search -- initRegistrationKSS.hpp:222-296, :430-450, :481-522               */
/* ------------------------------------------------------------------------------------ */
int ko_grid_angles(double step, double *angles, int cap) {
    /* :245 for (double i = 0; i < 6.3; i = i + 6.3 / step): double accumulation */
    int g = 0;
    for (double a = 0; a < 6.3; a = a + 6.3 / step) {
        if (g >= cap) return -1;
        angles[g++] = a;
    }
    return g;
}

double ko_error_ave(const double *cloud, int64_t n, const float *tf, int64_t nt, const ko_kdtree *tree) {
    /* :430-450: searchPoint.x = (float)pointS[i][0]; K=2 asked, [0] used; sqrt in double */
    double sum = 0;
    for (int64_t i = 0; i < n; i++) {
        float q[3] = {(float)cloud[3 * i], (float)cloud[3 * i + 1], (float)cloud[3 * i + 2]};
        int32_t id; float d;
        if (tree) kd_nn_one(tree, q, 0, &id, &d);
        else ko_nn_brute(q, 1, tf, nt, 0, &id, &d);
        double di = sqrt((double)d);
        sum = sum + di;
    }
    return sum / (double)n;
}

int ko_local_min(const double *value, int g, int i, int j, int k, int r) {
    /* :481-522: clamped (non-periodic) window, fails iff a strictly smaller value exists */
    double c = value[((int64_t)i * g + j) * g + k];
    int id = i - r < 0 ? 0 : i - r, iu = i + r >= g ? g - 1 : i + r;
    int jd = j - r < 0 ? 0 : j - r, ju = j + r >= g ? g - 1 : j + r;
    int kd = k - r < 0 ? 0 : k - r, ku = k + r >= g ? g - 1 : k + r;
    for (int ii = id; ii <= iu; ii++)
        for (int jj = jd; jj <= ju; jj++)
            for (int kk = kd; kk <= ku; kk++)
                if (c > value[((int64_t)ii * g + jj) * g + kk]) return 0;
    return 1;
}

int ko_rotation_search(const double *Sp, int64_t ns, const double *T, int64_t nt, double step,
                       double *value, int cap_g, double best_angle[3],
                       double *angle_list, int *n_list, int cap_list) {
    double ang[256];
    int g = ko_grid_angles(step, ang, 256);
    if (g < 0 || g > cap_g) return -1;
    /* :224-236 target narrowed to float PointXYZ, kd-tree built once */
    float *tf = (float *)malloc(sizeof(float) * 3 * (size_t)nt);
    for (int64_t i = 0; i < 3 * nt; i++) tf[i] = (float)T[i];
    ko_kdtree *tree = ko_kdtree_build(tf, nt, 15);
    double *px = (double *)malloc(sizeof(double) * 3 * (size_t)ns);
    double *pxy = (double *)malloc(sizeof(double) * 3 * (size_t)ns);
    double *pxyz = (double *)malloc(sizeof(double) * 3 * (size_t)ns);
    double errorT = 9999; /* :239 */
    double iG = 0, jG = 0, kG = 0;
    for (int a = 0; a < g; a++) {
        memcpy(px, Sp, sizeof(double) * 3 * (size_t)ns);
        ko_axis_rotate(1, ang[a], px, ns);                         /* :247 */
        for (int b = 0; b < g; b++) {
            memcpy(pxy, px, sizeof(double) * 3 * (size_t)ns);
            ko_axis_rotate(2, ang[b], pxy, ns);                    /* :250 */
            for (int c = 0; c < g; c++) {
                memcpy(pxyz, pxy, sizeof(double) * 3 * (size_t)ns);
                ko_axis_rotate(3, ang[c], pxyz, ns);               /* :253 */
                double e = ko_error_ave(pxyz, ns, tf, nt, tree);   /* :255 */
                value[((int64_t)a * g + b) * g + c] = e;
                if (e < errorT) { iG = ang[a]; jG = ang[b]; kG = ang[c]; errorT = e; } /* :258 strict */
            }
        }
    }
    int nl = 0;
    for (int a = 0; a < g; a++)
        for (int b = 0; b < g; b++)
            for (int c = 0; c < g; c++)
                if (ko_local_min(value, g, a, b, c, 2)) {           /* :279, r = 2 (:35) */
                    if (nl >= cap_list) { nl = -1; goto done; }
                    angle_list[3 * nl + 0] = (double)a * 6.3 / (double)step; /* :282-284 */
                    angle_list[3 * nl + 1] = (double)b * 6.3 / (double)step;
                    angle_list[3 * nl + 2] = (double)c * 6.3 / (double)step;
                    nl++;
                }
done:
    best_angle[0] = iG; best_angle[1] = jG; best_angle[2] = kG;     /* :291-293 */
    *n_list = nl;
    free(px); free(pxy); free(pxyz); free(tf); ko_kdtree_free(tree);
    return nl < 0 ? -2 : g;
}

/* ------------------------------------------------------------------------------------ */
/* float Matrix4f helpers (Eigen evaluation order: columns accumulated k = 0..3, no fma)  */
/* ------------------------------------------------------------------------------------ */
void ko_mat4_mul(const float A[16], const float B[16], float C[16]) {
    float R[16];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            float acc = A[4 * i + 0] * B[0 * 4 + j];
            acc = acc + A[4 * i + 1] * B[1 * 4 + j];
            acc = acc + A[4 * i + 2] * B[2 * 4 + j];
            acc = acc + A[4 * i + 3] * B[3 * 4 + j];
            R[4 * i + j] = acc;
        }
    memcpy(C, R, sizeof R);
}

void ko_transform_points_f32(const float T[16], const float *in, int64_t n, float *out) {
    /* pcl IterativeClosestPoint::transformCloud: pt_t = tr * (x,y,z,1) */
    for (int64_t i = 0; i < n; i++) {
        float x = in[3 * i], y = in[3 * i + 1], z = in[3 * i + 2];
        for (int r = 0; r < 3; r++) {
            float acc = T[4 * r + 0] * x;
            acc = acc + T[4 * r + 1] * y;
            acc = acc + T[4 * r + 2] * z;
            acc = acc + T[4 * r + 3];
            out[3 * i + r] = acc;
        }
    }
}

/* ------------------------------------------------------------------------------------ */
/* 3x3 SVD (Jacobi on A^T A) and Umeyama-without-scaling                                  */
/* ------------------------------------------------------------------------------------ */
static double det3(const double M[9]) {
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}

static void jacobi_eig3(double S[9], double V[9]) {
    /* cyclic Jacobi for a symmetric 3x3; on exit S ~ diag, V columns = eigenvectors */
    for (int i = 0; i < 9; i++) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 64; sweep++) {
        double offn = fabs(S[1]) + fabs(S[2]) + fabs(S[5]);
        double dn = fabs(S[0]) + fabs(S[4]) + fabs(S[8]);
        if (offn <= 1e-300 || offn <= 1e-18 * dn) break;
        for (int p = 0; p < 2; p++)
            for (int q = p + 1; q < 3; q++) {
                double apq = S[3 * p + q];
                if (apq == 0.0) continue;
                double app = S[3 * p + p], aqq = S[3 * q + q];
                double theta = (aqq - app) / (2.0 * apq);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; k++) { /* S <- S * J */
                    double skp = S[3 * k + p], skq = S[3 * k + q];
                    S[3 * k + p] = c * skp - s * skq;
                    S[3 * k + q] = s * skp + c * skq;
                }
                for (int k = 0; k < 3; k++) { /* S <- J^T * S */
                    double spk = S[3 * p + k], sqk = S[3 * q + k];
                    S[3 * p + k] = c * spk - s * sqk;
                    S[3 * q + k] = s * spk + c * sqk;
                }
                for (int k = 0; k < 3; k++) {
                    double vkp = V[3 * k + p], vkq = V[3 * k + q];
                    V[3 * k + p] = c * vkp - s * vkq;
                    V[3 * k + q] = s * vkp + c * vkq;
                }
            }
    }
}

void ko_svd3(const double A[9], double U[9], double s[3], double V[9]) {
    double AtA[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double acc = 0;
            for (int k = 0; k < 3; k++) acc += A[3 * k + i] * A[3 * k + j];
            AtA[3 * i + j] = acc;
        }
    double Vv[9];
    jacobi_eig3(AtA, Vv);
    double ev[3] = {AtA[0], AtA[4], AtA[8]};
    int ord[3] = {0, 1, 2};
    for (int a = 0; a < 2; a++)
        for (int b = a + 1; b < 3; b++)
            if (ev[ord[b]] > ev[ord[a]]) { int tmp = ord[a]; ord[a] = ord[b]; ord[b] = tmp; }
    for (int c = 0; c < 3; c++) {
        for (int r = 0; r < 3; r++) V[3 * r + c] = Vv[3 * r + ord[c]];
        s[c] = sqrt(ev[ord[c]] > 0 ? ev[ord[c]] : 0.0);
    }
    /* U columns = A v_c / s_c; complete rank-deficient columns orthonormally */
    double u[3][3];
    int ok[3];
    for (int c = 0; c < 3; c++) {
        double n2 = 0;
        for (int r = 0; r < 3; r++) {
            double acc = 0;
            for (int k = 0; k < 3; k++) acc += A[3 * r + k] * V[3 * k + c];
            u[c][r] = acc; n2 += acc * acc;
        }
        double nn = sqrt(n2);
        ok[c] = nn > 1e-13 * (s[0] > 0 ? s[0] : 1.0) && nn > 0;
        if (ok[c]) for (int r = 0; r < 3; r++) u[c][r] /= nn;
    }
    if (!ok[0]) { u[0][0] = 1; u[0][1] = 0; u[0][2] = 0; ok[0] = 1; }
    if (!ok[1]) {
        /* any unit vector orthogonal to u0 */
        double a[3] = {0, 0, 0};
        int m = fabs(u[0][0]) < fabs(u[0][1]) ? (fabs(u[0][0]) < fabs(u[0][2]) ? 0 : 2) : (fabs(u[0][1]) < fabs(u[0][2]) ? 1 : 2);
        a[m] = 1;
        double d = a[0] * u[0][0] + a[1] * u[0][1] + a[2] * u[0][2], n2 = 0;
        for (int r = 0; r < 3; r++) { u[1][r] = a[r] - d * u[0][r]; n2 += u[1][r] * u[1][r]; }
        n2 = sqrt(n2);
        for (int r = 0; r < 3; r++) u[1][r] /= n2;
        ok[1] = 1;
    }
    if (!ok[2]) {
        u[2][0] = u[0][1] * u[1][2] - u[0][2] * u[1][1];
        u[2][1] = u[0][2] * u[1][0] - u[0][0] * u[1][2];
        u[2][2] = u[0][0] * u[1][1] - u[0][1] * u[1][0];
    }
    for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) U[3 * r + c] = u[c][r];
}

void ko_rigid_from_sums(const double sums[20], float Tk[16]) {
    /* pcl::registration::TransformationEstimationSVD -> Eigen umeyama(src, dst, false):
       sigma = (1/n) sum (dst-mu_d)(src-mu_s)^T ; R = U S V^T ; t = mu_d - R mu_s.
       Accumulated here in double (PCL: float).  */
    double n = sums[0];
    double ms[3] = {sums[1] / n, sums[2] / n, sums[3] / n};
    double md[3] = {sums[4] / n, sums[5] / n, sums[6] / n};
    double sigma[9];
    for (int i = 0; i < 3; i++)      /* dst index (row) */
        for (int j = 0; j < 3; j++)  /* src index (col) */
            sigma[3 * i + j] = sums[7 + 3 * j + i] / n - md[i] * ms[j];
    double U[9], s[3], V[9];
    ko_svd3(sigma, U, s, V);
    double S[3] = {1, 1, 1};
    if (det3(U) * det3(V) < 0) S[2] = -1; /* Eq. (39) */
    double R[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double acc = 0;
            for (int k = 0; k < 3; k++) acc += U[3 * i + k] * S[k] * V[3 * j + k];
            R[3 * i + j] = acc;
        }
    double t[3];
    for (int i = 0; i < 3; i++) t[i] = md[i] - (R[3 * i] * ms[0] + R[3 * i + 1] * ms[1] + R[3 * i + 2] * ms[2]);
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) Tk[4 * i + j] = (float)R[3 * i + j];
        Tk[4 * i + 3] = (float)t[i];
    }
    Tk[12] = Tk[13] = Tk[14] = 0.f; Tk[15] = 1.f;
}

/* ------------------------------------------------------------------------------------ */
/* ICP -- pcl::IterativeClosestPoint::computeTransformation (PCL 1.8.1 icp.hpp),          */
/*        CorrespondenceEstimation::determineCorrespondences,                             */
/*        DefaultConvergenceCriteria::hasConverged, Registration::getFitnessScore         */
/* ------------------------------------------------------------------------------------ */
void ko_icp_default_params(ko_icp_params *p) {
    memset(p, 0, sizeof *p);
    p->max_iterations = 1000;             /* Main_KSS_ICP.cpp:81 */
    p->max_corr_dist = 1.0;               /* KSS_ICP.hpp:156 */
    p->transformation_epsilon = 1e-10;    /* :157 */
    p->euclidean_fitness_epsilon = 0.001; /* :158 */
    p->abs_mse_epsilon = 1e-12;
    p->min_correspondences = 3;
    p->use_kdtree = 1;
    p->nthreads = 1;
    p->compute_fitness = 1;
}

static void nn_pass(const ko_kdtree *tree, const float *q, int64_t nq, const float *t, int64_t nt,
                    const ko_icp_params *p, int32_t *idx, float *d2, double *acc) {
    double t0 = now_s();
    if (tree) ko_kdtree_nn(tree, q, nq, p->fma, p->nthreads, idx, d2);
    else ko_nn_brute(q, nq, t, nt, p->fma, idx, d2);
    *acc += now_s() - t0;
}

int ko_icp(const float *src, int64_t ns, const float *tgt, int64_t nt,
           const ko_icp_params *p, ko_icp_result *res, ko_icp_trace *trace) {
    memset(res, 0, sizeof *res);
    double t_start = now_s();
    float *cur = (float *)malloc(sizeof(float) * 3 * (size_t)(ns > 0 ? ns : 1));
    int32_t *idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)(ns > 0 ? ns : 1));
    float *d2 = (float *)malloc(sizeof(float) * (size_t)(ns > 0 ? ns : 1));
    memcpy(cur, src, sizeof(float) * 3 * (size_t)ns); /* *input_transformed = *input_ (identity guess) */
    ko_kdtree *tree = NULL;
    if (p->use_kdtree) { double t0 = now_s(); tree = ko_kdtree_build(tgt, nt, 15); res->build_seconds = now_s() - t0; }

    float final[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    float Tk[16];
    const double max_dist_sqr = p->max_corr_dist * p->max_corr_dist;
    const double rotation_threshold = 1.0 - p->transformation_epsilon; /* setRotationThreshold */
    const double translation_threshold = p->transformation_epsilon;
    double prev_mse = DBL_MAX; /* correspondences_prev_mse_ initial value */
    int iterations = 0, converged = 0, state = KO_STATE_NOT_CONVERGED;
    if (trace) trace->n = 0;

    do {
        nn_pass(tree, cur, ns, tgt, nt, p, idx, d2, &res->nn_seconds);
        double sums[20];
        memset(sums, 0, sizeof sums);
        for (int64_t i = 0; i < ns; i++) {
            double dd = (double)d2[i];
            sums[17] += dd;
            sums[18] += sqrt(dd);
            if (dd > max_dist_sqr) continue; /* determineCorrespondences: distance[0] > max_dist_sqr */
            const float *a = cur + 3 * i, *b = tgt + 3 * (int64_t)idx[i];
            sums[0] += 1.0;
            for (int k = 0; k < 3; k++) { sums[1 + k] += (double)a[k]; sums[4 + k] += (double)b[k]; }
            for (int k = 0; k < 3; k++)
                for (int l = 0; l < 3; l++) sums[7 + 3 * k + l] += (double)a[k] * (double)b[l];
            sums[16] += dd;
        }
        if ((int)sums[0] < p->min_correspondences) { /* "Not enough correspondences found" */
            state = KO_STATE_NO_CORRESPONDENCES; converged = 0;
            break;
        }
        ko_rigid_from_sums(sums, Tk);
        ko_transform_points_f32(Tk, cur, ns, cur);  /* transformCloud(*input_transformed, ..) */
        ko_mat4_mul(Tk, final, final);              /* final = transformation_ * final */
        ++iterations;
        double mse = sums[16] / sums[0];            /* calculateMSE(correspondences_) */
        res->last_mse = mse;
        if (trace && trace->n < trace->cap) {
            memcpy(trace->sums + 20 * (size_t)trace->n, sums, sizeof sums);
            memcpy(trace->Tk + 16 * (size_t)trace->n, Tk, sizeof Tk);
            trace->n++;
        }
        /* --- DefaultConvergenceCriteria::hasConverged --- */
        converged = 0; state = KO_STATE_NOT_CONVERGED;
        if (iterations >= p->max_iterations) { converged = 1; state = KO_STATE_ITERATIONS; }
        else if (!p->fixed_iterations) {
            float tr = Tk[0] + Tk[5] + Tk[10] - 1;                     /* float sum, as Matrix4f coeffs */
            double cos_angle = 0.5 * tr;
            float tsq = Tk[3] * Tk[3] + Tk[7] * Tk[7] + Tk[11] * Tk[11];
            double translation_sqr = tsq;
            if (cos_angle >= rotation_threshold && translation_sqr <= translation_threshold) {
                converged = 1; state = KO_STATE_TRANSFORM;  /* max_iterations_similar_transforms_ = 0 */
            } else if (fabs(mse - prev_mse) < p->abs_mse_epsilon) {
                converged = 1; state = KO_STATE_ABS_MSE;
            } else if (fabs(mse - prev_mse) / prev_mse < p->euclidean_fitness_epsilon) {
                converged = 1; state = KO_STATE_REL_MSE;
            } else {
                prev_mse = mse;
            }
        }
    } while (!converged);

    memcpy(res->T, final, sizeof final);
    res->iterations = iterations; res->converged = converged; res->state = state;
    if (p->compute_fitness) {
        /* getFitnessScore(): transform the ORIGINAL input by final, mean NN d2 over all points */
        ko_transform_points_f32(final, src, ns, cur);
        nn_pass(tree, cur, ns, tgt, nt, p, idx, d2, &res->nn_seconds);
        double f = 0;
        for (int64_t i = 0; i < ns; i++) f += (double)d2[i];
        res->fitness = ns > 0 ? f / (double)ns : DBL_MAX;
    }
    ko_kdtree_free(tree);
    free(cur); free(idx); free(d2);
    res->total_seconds = now_s() - t_start;
    return 0;
}

/* ------------------------------------------------------------------------------------ */
/* PCR_QM -- registrationMeasure.hpp:47-98                                                */
/* ------------------------------------------------------------------------------------ */
void ko_pcr_qm(const double *A, int64_t na, const double *T, int64_t nt, double out[3]) {
    float *tf = (float *)malloc(sizeof(float) * 3 * (size_t)(nt > 0 ? nt : 1));
    for (int64_t i = 0; i < 3 * nt; i++) tf[i] = (float)T[i]; /* :59-61 */
    ko_kdtree *tree = ko_kdtree_build(tf, nt, 15);
    double mse = 0, mae = 0;
    for (int64_t i = 0; i < na; i++) {
        float q[3] = {(float)A[3 * i], (float)A[3 * i + 1], (float)A[3 * i + 2]}; /* :75-78 */
        int32_t id; float d;
        kd_nn_one(tree, q, 0, &id, &d);
        double di = (double)d;    /* :80 */
        mse = mse + di;
        mae = mae + sqrt(di);     /* :81 */
    }
    mse = mse / (double)na; mae = mae / (double)na;
    out[0] = mse; out[1] = sqrt(mse); out[2] = mae; /* :85-97 */
    ko_kdtree_free(tree); free(tf);
}

/* ------------------------------------------------------------------------------------ */
/* KSSICP orchestration -- KSS_ICP.hpp:86-131 + :185-233 (down-sampled clouds given)      */
/* ------------------------------------------------------------------------------------ */
static void to_f32(const double *in, int64_t n3, float *out) { for (int64_t i = 0; i < n3; i++) out[i] = (float)in[i]; }

static void euler_matrix(const double a[3], double R[9]) {
    /* R0 = Rz(a2) * Ry(a1) * Rx(a0) with the reference's axis conventions (:365-404) */
    double cx = cos(a[0]), sx = sin(a[0]), cy = cos(a[1]), sy = sin(a[1]), cz = cos(a[2]), sz = sin(a[2]);
    double Rx[9] = {1, 0, 0, 0, cx, -sx, 0, sx, cx};
    double Ry[9] = {cy, 0, sy, 0, 1, 0, -sy, 0, cy};
    double Rz[9] = {cz, -sz, 0, sz, cz, 0, 0, 0, 1};
    double tmp[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
        double acc = 0; for (int k = 0; k < 3; k++) acc += Ry[3 * i + k] * Rx[3 * k + j]; tmp[3 * i + j] = acc; }
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
        double acc = 0; for (int k = 0; k < 3; k++) acc += Rz[3 * i + k] * tmp[3 * k + j]; R[3 * i + j] = acc; }
}

int ko_kssicp_register(const double *Ssub, int64_t nss, const double *Tsub, int64_t nts,
                       const double *Sfull, int64_t nsf, double accurate, int iter,
                       int use_kdtree, double *pointAlign, ko_kssicp_result *res) {
    memset(res, 0, sizeof *res);
    ko_preshape ps;
    ko_preshape_stats(Ssub, nss, Tsub, nts, &ps);                       /* :62 via :87 */
    double *Sp = (double *)malloc(sizeof(double) * 3 * (size_t)nss);
    ko_similarity_apply(Ssub, nss, &ps, Sp);
    int capg = 64;
    double *value = (double *)malloc(sizeof(double) * (size_t)capg * capg * capg);
    int cap_list = capg * capg * capg;
    double *alist = (double *)malloc(sizeof(double) * 3 * (size_t)cap_list);
    double best[3]; int nl = 0;
    int g = ko_rotation_search(Sp, nss, Tsub, nts, accurate, value, capg, best, alist, &nl, cap_list);
    if (g < 0) { free(Sp); free(value); free(alist); return -1; }
    res->n_angle_list = nl; res->scale = ps.scale;

    ko_icp_params ip; ko_icp_default_params(&ip);
    ip.max_iterations = iter; ip.use_kdtree = use_kdtree;
    float *tf = (float *)malloc(sizeof(float) * 3 * (size_t)nts);
    to_f32(Tsub, 3 * nts, tf);
    double *P = (double *)malloc(sizeof(double) * 3 * (size_t)nss);
    float *pf = (float *)malloc(sizeof(float) * 3 * (size_t)nss);
    ko_icp_result ir;

    ko_pose_apply(Ssub, nss, &ps, best, P);                              /* :92 */
    to_f32(P, 3 * nss, pf);
    ko_icp(pf, nss, tf, nts, &ip, &ir, NULL);                            /* :93 Judge */
    res->E_d_init = ir.fitness;
    double chosen[3] = {best[0], best[1], best[2]};
    if (res->E_d_init > 0.0005) {                                        /* :99 */
        double Q = 9999; int angleIndex = 0;                             /* :100-101 */
        for (int i = 0; i < nl; i++) {
            ko_pose_apply(Ssub, nss, &ps, alist + 3 * i, P);             /* :103 */
            to_f32(P, 3 * nss, pf);
            ko_icp(pf, nss, tf, nts, &ip, &ir, NULL);                    /* :104 */
            double ri = ir.fitness;
            if (ri < Q && ri >= 0) { Q = ri; angleIndex = i; }           /* :113-116 */
        }
        res->used_angle_list = 1; res->angle_index = angleIndex;
        if (nl > 0) for (int k = 0; k < 3; k++) chosen[k] = alist[3 * angleIndex + k]; /* :119-120 */
    }
    for (int k = 0; k < 3; k++) res->R0_angle[k] = chosen[k];
    ko_pose_apply(Ssub, nss, &ps, chosen, P);
    double *full = (double *)malloc(sizeof(double) * 3 * (size_t)(nsf > 0 ? nsf : 1));
    ko_pose_apply(Sfull, nsf, &ps, chosen, full);                        /* :120 / :124 ; :127-129 */
    to_f32(P, 3 * nss, pf);
    ko_icp(pf, nss, tf, nts, &ip, &ir, NULL);                            /* :130 -> :185-233 */
    res->final_fitness = ir.fitness;
    res->icp_iterations = ir.iterations; res->icp_converged = ir.converged;
    memcpy(res->T_icp, ir.T, sizeof ir.T);
    for (int64_t i = 0; i < nsf; i++) {                                  /* :224-230 float coeff x double coord */
        double x = full[3 * i], y = full[3 * i + 1], z = full[3 * i + 2];
        for (int r = 0; r < 3; r++)
            pointAlign[3 * i + r] = ir.T[4 * r + 0] * x + ir.T[4 * r + 1] * y + ir.T[4 * r + 2] * z + ir.T[4 * r + 3];
    }
    /* composite similarity, SURVEY.md section 3.1 */
    double R0[9]; euler_matrix(chosen, R0);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
        double acc = 0; for (int k = 0; k < 3; k++) acc += (double)ir.T[4 * i + k] * R0[3 * k + j]; res->R[3 * i + j] = acc; }
    double v[3];
    for (int k = 0; k < 3; k++) v[k] = ps.c_tgt[k] - ps.scale * ps.c_src[k];
    for (int i = 0; i < 3; i++)
        res->t[i] = res->R[3 * i] * v[0] + res->R[3 * i + 1] * v[1] + res->R[3 * i + 2] * v[2] + (double)ir.T[4 * i + 3];
    free(Sp); free(value); free(alist); free(tf); free(P); free(pf); free(full);
    return 0;
}

/* ------------------------------------------------------------------------------------ */
/* ASCII PLY vertex reader -- CPLYLoader::LoadModel parse rules, PlyLoad.cpp:10-114       */
/* ------------------------------------------------------------------------------------ */
int64_t ko_ply_load(const char *path, double **pts) {
    *pts = NULL;
    if (!strstr(path, ".ply")) return -10;           /* :13-15 extension test */
    FILE *f = fopen(path, "r");
    if (!f) return -11;                              /* :18-22 */
    char buf[1000];
    int nv = -1, nf = -1;
    if (!fgets(buf, 300, f)) { fclose(f); return -12; }
    /* :59-65 scan forward for "element vertex" */
    while (strncmp("element vertex", buf, strlen("element vertex")) != 0)
        if (!fgets(buf, 300, f)) { fclose(f); return -13; }
    sscanf(buf + strlen("element vertex"), "%i", &nv);
    /* :68-75 rewind, scan for "element face" (the reference spins forever if absent) */
    fseek(f, 0, SEEK_SET);
    while (strncmp("element face", buf, strlen("element face")) != 0)
        if (!fgets(buf, 300, f)) { fclose(f); return -14; }
    sscanf(buf + strlen("element face"), "%i", &nf);
    /* :78-82 */
    while (strncmp("end_header", buf, strlen("end_header")) != 0)
        if (!fgets(buf, 300, f)) { fclose(f); return -15; }
    if (nv < 0) { fclose(f); return -16; }
    double *P = (double *)malloc(sizeof(double) * 3 * (size_t)(nv > 0 ? nv : 1));
    for (int i = 0; i < nv; i++) {
        float x = 0, y = 0, z = 0;                   /* :93-96: parsed as float, widened (:101-103) */
        if (!fgets(buf, 300, f)) { free(P); fclose(f); return -17; }
        sscanf(buf, "%f %f %f", &x, &y, &z);
        P[3 * i] = x; P[3 * i + 1] = y; P[3 * i + 2] = z;
    }
    fclose(f);
    *pts = P;
    return nv;
}

/* ------------------------------------------------------------------------------------ */
void ko_fps(const double *xyz, int64_t n, int64_t m, int32_t *idx) {
    double *mind = (double *)malloc(sizeof(double) * (size_t)n);
    for (int64_t i = 0; i < n; i++) mind[i] = INFINITY;
    int64_t cur = 0;
    for (int64_t k = 0; k < m; k++) {
        idx[k] = (int32_t)cur;
        if (k + 1 == m) break;
        double bv = -1.0; int64_t bi = 0;
        for (int64_t i = 0; i < n; i++) {
            double dx = xyz[3 * i] - xyz[3 * cur], dy = xyz[3 * i + 1] - xyz[3 * cur + 1], dz = xyz[3 * i + 2] - xyz[3 * cur + 2];
            double d = (dx * dx + dy * dy) + dz * dz;
            if (d < mind[i]) mind[i] = d;
            if (mind[i] > bv) { bv = mind[i]; bi = i; }
        }
        cur = bi;
    }
    free(mind);
}

/* ------------------------------------------------------------------------------------ */
/* AIVS down-sampler (see kss_oracle.h for the line map)                                   */
/* ------------------------------------------------------------------------------------ */
typedef struct {
    int nx, ny, nz, nboxes;       /* boxes are 1-based: index = x + nx*(y-1) + nx*ny*(z-1) */
    double minx, miny, minz, unit;
} aivs_grid;

static int aivs_box_scale(int64_t n) {   /* ballRegionCompute.hpp:1194-1215 */
    if (n < 10000) return 10;
    if (n < 50000) return 20;
    if (n < 100000) return 30;
    if (n < 500000) return 40;
    if (n < 1000000) return 50;
    return (int)pow((double)n / 8.0, 1.0 / 3.0);
}

static void aivs_box_center(const aivs_grid *g, int boxIndex, double c[3]) {   /* :1150-1172 */
    int z_num = boxIndex / (g->nx * g->ny) + 1;
    int leveZ = boxIndex % (g->nx * g->ny);
    int y_num = leveZ / g->nx + 1;
    int x_num = leveZ % g->nx;
    if (x_num == 0) { x_num = g->nx; y_num = y_num - 1; }
    c[0] = (g->minx + (x_num - 1) * g->unit + g->minx + x_num * g->unit) / 2;
    c[1] = (g->miny + (y_num - 1) * g->unit + g->miny + y_num * g->unit) / 2;
    c[2] = (g->minz + (z_num - 1) * g->unit + g->minz + z_num * g->unit) / 2;
}

static int aivs_neighbor_boxes(const aivs_grid *g, int boxIndex, int out[26]) {   /* :975-1031, quirks kept */
    int z_num = boxIndex / (g->nx * g->ny) + 1;
    int leveZ = boxIndex % (g->nx * g->ny);
    int y_num = leveZ / g->nx + 1;
    int x_num = leveZ % g->nx;   /* NOT wrapped here (it is in aivs_box_center): last-column boxes get odd neighbours */
    int xs[3], ys[3], zs[3], nxs = 0, nys = 0, nzs = 0;
    if (x_num > 1) xs[nxs++] = x_num - 1;
    xs[nxs++] = x_num;
    if (x_num < g->nx) xs[nxs++] = x_num + 1;
    if (y_num > 1) ys[nys++] = y_num - 1;
    ys[nys++] = y_num;
    if (y_num < g->ny) ys[nys++] = y_num + 1;
    if (z_num > 1) zs[nzs++] = z_num - 1;
    zs[nzs++] = z_num;
    if (z_num < g->nz) zs[nzs++] = z_num + 1;
    int m = 0;
    for (int i = 0; i < nxs; i++)
        for (int j = 0; j < nys; j++)
            for (int k = 0; k < nzs; k++) {
                if (xs[i] == x_num && ys[j] == y_num && zs[k] == z_num) continue;
                int idx = xs[i] + (ys[j] - 1) * g->nx + (zs[k] - 1) * g->nx * g->ny;
                if (idx < g->nboxes + 1 && idx >= 0) out[m++] = idx;   /* `index_Compute < squareBoxes.size()`; a negative index would be UB there */
            }
    return m;
}

static inline float aivs_dist(const double *pts, int a, int b) {
    /* pcl::KdTreeFLANN on float PointXYZ, then sqrt(float) */
    float ax = (float)pts[3 * (int64_t)a], ay = (float)pts[3 * (int64_t)a + 1], az = (float)pts[3 * (int64_t)a + 2];
    float bx = (float)pts[3 * (int64_t)b], by = (float)pts[3 * (int64_t)b + 1], bz = (float)pts[3 * (int64_t)b + 2];
    float dx = ax - bx, dy = ay - by, dz = az - bz;
    return sqrtf((dx * dx + dy * dy) + dz * dz);
}

int64_t ko_aivs(const double *P, int64_t n, int64_t point_num, int32_t *out_idx, int64_t cap) {
    if (n <= 0 || point_num <= 0) return -1;
    /* pointPipeline_Border, pointPipeline.hpp:105-160: strict comparisons, first extreme kept */
    double maxx = P[0], minx = P[0], maxy = P[1], miny = P[1], maxz = P[2], minz = P[2];
    for (int64_t i = 0; i < n; i++) {
        double x = P[3 * i], y = P[3 * i + 1], z = P[3 * i + 2];
        if (x < minx) minx = x;
        if (x > maxx) maxx = x;
        if (y < miny) miny = y;
        if (y > maxy) maxy = y;
        if (z < minz) minz = z;
        if (z > maxz) maxz = z;
    }
    /* BallRegion_AchieveXYZ, ballRegionCompute.hpp:690-758 */
    aivs_grid g;
    int boxNum = aivs_box_scale(n);
    g.minx = minx; g.miny = miny; g.minz = minz;
    double x_dis = fabs(maxx - minx), y_dis = fabs(maxy - miny), z_dis = fabs(maxz - minz);
    double large = x_dis;
    if (large < y_dis) large = y_dis;
    if (large < z_dis) large = z_dis;
    g.unit = large / (double)boxNum;
    double numX = x_dis / g.unit, numY = y_dis / g.unit, numZ = z_dis / g.unit;
    g.nx = (int)numX; g.ny = (int)numY; g.nz = (int)numZ;
    if (numX > (double)g.nx) g.nx++;
    if (numY > (double)g.ny) g.ny++;
    if (numZ > (double)g.nz) g.nz++;
    if (g.nx < 1 || g.ny < 1 || g.nz < 1 || !(g.unit > 0)) return -2;   /* planar / degenerate clouds divide by zero in the reference */
    g.nboxes = g.nx * g.ny * g.nz;
    const int nb1 = g.nboxes + 1;
    /* BallRegion_BoxInput, :632-688 */
    int32_t *box_of = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    int32_t *count = (int32_t *)calloc((size_t)nb1 + 1, sizeof(int32_t));
    for (int64_t i = 0; i < n; i++) {
        double xNum = (P[3 * i] - g.minx) / g.unit, yNum = (P[3 * i + 1] - g.miny) / g.unit, zNum = (P[3 * i + 2] - g.minz) / g.unit;
        int xi = (int)xNum, yi = (int)yNum, zi = (int)zNum;
        if (xi < xNum || xi == 0) xi++;
        if (yi < yNum || yi == 0) yi++;
        if (zi < zNum || zi == 0) zi++;
        int idx = xi + g.nx * (yi - 1) + g.nx * g.ny * (zi - 1);
        if (idx >= nb1 || idx < 0) { free(box_of); free(count); return -3; }   /* the reference prints "Hello!" and then indexes out of range */
        box_of[i] = idx;
        count[idx]++;
    }
    int32_t *start = (int32_t *)malloc(sizeof(int32_t) * ((size_t)nb1 + 1));
    start[0] = 0;
    for (int b = 0; b < nb1; b++) start[b + 1] = start[b] + count[b];
    int32_t *members = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);   /* squareBoxes[b] in push_back (= index) order */
    int32_t *fill = (int32_t *)calloc((size_t)nb1, sizeof(int32_t));
    for (int64_t i = 0; i < n; i++) members[start[box_of[i]] + fill[box_of[i]]++] = (int32_t)i;
    /* box centre and the member closest to it (:634-686): strict '>' keeps the first minimum */
    int32_t *center_pos = (int32_t *)malloc(sizeof(int32_t) * (size_t)nb1);
    for (int b = 0; b < nb1; b++) {
        center_pos[b] = -1;
        double c[3], best = 9999;
        aivs_box_center(&g, b, c);
        for (int k = start[b]; k < start[b + 1]; k++) {
            int i = members[k];
            double d = sqrt((c[0] - P[3 * (int64_t)i]) * (c[0] - P[3 * (int64_t)i]) + (c[1] - P[3 * (int64_t)i + 1]) * (c[1] - P[3 * (int64_t)i + 1]) +
                            (c[2] - P[3 * (int64_t)i + 2]) * (c[2] - P[3 * (int64_t)i + 2]));
            if (best > d) { best = d; center_pos[b] = k - start[b]; }
        }
    }
    /* AIVS_BoxSimplification_Points, Method_AIVS_SimPro.hpp:776-794 */
    const double rate = (double)point_num / (double)n;
    int32_t *sim_num = (int32_t *)malloc(sizeof(int32_t) * (size_t)nb1);
    for (int b = 0; b < nb1; b++) {
        double simBox = (double)count[b] * rate;
        int t = (int)simBox;
        sim_num[b] = (simBox - t > 0.2) ? t + 1 : t;
    }
    /* AIVS_Voroni_OpenMP_KNN, :222-376.  labelG: 1 = not sampled, 0 = sampled */
    uint8_t *labelG = (uint8_t *)malloc((size_t)n);
    memset(labelG, 1, (size_t)n);
    int32_t *simiT = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);   /* simiT[b] lives at start[b].. (size = box size), -1 = empty */
    for (int64_t i = 0; i < n; i++) simiT[i] = -1;
    const double searchBoxRadius = g.unit * 3.0 / 4.0;
    double *mind = (double *)malloc(sizeof(double) * (size_t)n);
    for (int colour = 0; colour < 8; colour++) {
        /* AIVS_initBoxIndexNumber, :587-643: parity classes of (i, j, k); order inside a colour is irrelevant */
        for (int i = 1; i <= g.nx; i++)
            for (int j = 1; j <= g.ny; j++)
                for (int k = 1; k <= g.nz; k++) {
                    const int oi = i % 2, oj = j % 2, ok = k % 2;
                    int cls;
                    if (oi == 1 && oj == 1 && ok == 1) cls = 0;
                    else if (oi == 0 && oj == 1 && ok == 1) cls = 1;
                    else if (oi == 0 && oj == 0 && ok == 1) cls = 2;
                    else if (oi == 1 && oj == 0 && ok == 1) cls = 3;
                    else if (oi == 1 && oj == 1 && ok == 0) cls = 4;
                    else if (oi == 0 && oj == 1 && ok == 0) cls = 5;
                    else if (oi == 0 && oj == 0 && ok == 0) cls = 6;
                    else cls = 7;
                    if (cls != colour) continue;
                    const int b = i + g.nx * (j - 1) + g.nx * g.ny * (k - 1);
                    if (count[b] == 0) continue;
                    const int simNum = sim_num[b];
                    if (simNum == 0) continue;
                    double pc[3];
                    aivs_box_center(&g, b, pc);
                    const int m = count[b];
                    const int32_t *own = members + start[b];
                    /* already-sampled points of the neighbour boxes inside the search cube (label 2) */
                    int nbr[26];
                    const int nn = aivs_neighbor_boxes(&g, b, nbr);
                    int n_l2 = 0;
                    static int32_t *l2 = NULL; static int l2cap = 0;
                    for (int q = 0; q < nn; q++)
                        for (int l = start[nbr[q]]; l < start[nbr[q] + 1]; l++) {
                            const int pi = members[l];
                            const double *pn = P + 3 * (int64_t)pi;
                            if (pn[0] <= pc[0] + searchBoxRadius && pn[0] >= pc[0] - searchBoxRadius && pn[1] <= pc[1] + searchBoxRadius &&
                                pn[1] >= pc[1] - searchBoxRadius && pn[2] <= pc[2] + searchBoxRadius && pn[2] >= pc[2] - searchBoxRadius &&
                                labelG[pi] == 0) {
                                if (n_l2 == l2cap) { l2cap = l2cap * 2 + 64; l2 = (int32_t *)realloc(l2, sizeof(int32_t) * (size_t)l2cap); }
                                l2[n_l2++] = pi;
                            }
                        }
                    int sample_index = 0;
                    const int seed_pos = (n_l2 == 0 && center_pos[b] >= 0 && center_pos[b] < m) ? center_pos[b] : -1;   /* addJ */
                    /* labelTemp: own points 1 (seed 0), neighbour samples 2 */
                    for (int k2 = 0; k2 < m; k2++) {
                        if (k2 == seed_pos) {
                            mind[start[b] + k2] = 0;
                            simiT[start[b] + sample_index++] = own[k2];
                            labelG[own[k2]] = 0;
                        } else {
                            /* nearest label-0/2 point through the local kd-tree: min over them of sqrt(float d2) */
                            double mt = 9999;
                            float bestf = INFINITY;
                            if (seed_pos >= 0) bestf = aivs_dist(P, own[k2], own[seed_pos]);
                            for (int l = 0; l < n_l2; l++) { float d = aivs_dist(P, own[k2], l2[l]); if (d < bestf) bestf = d; }
                            if (bestf != INFINITY) mt = (double)bestf;
                            mind[start[b] + k2] = mt;
                        }
                    }
                    uint8_t *lab = labelG;   /* own points: labelTemp 1 <=> labelG 1 inside this box (own points start unsampled) */
                    while (sample_index < simNum) {
                        int sel = -1; double mx = 0;
                        for (int k2 = 0; k2 < m; k2++)
                            if (lab[own[k2]] == 1 && mind[start[b] + k2] > mx) { sel = k2; mx = mind[start[b] + k2]; }
                        if (sel == -1) break;
                        mind[start[b] + sel] = 0;
                        labelG[own[sel]] = 0;
                        simiT[start[b] + sample_index++] = own[sel];
                        for (int k2 = 0; k2 < m; k2++)
                            if (lab[own[k2]] == 1) {
                                double d = (double)aivs_dist(P, own[k2], own[sel]);
                                if (d < mind[start[b] + k2]) mind[start[b] + k2] = d;
                            }
                    }
                }
    }
    /* AIVS_AccurateCut_Optimization, :848-957 */
    int32_t *samples = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    int64_t ns = 0;
    for (int b = 0; b < nb1; b++)
        for (int k = start[b]; k < start[b + 1]; k++) {
            if (simiT[k] == -1) break;
            samples[ns++] = simiT[k];
        }
    int64_t dTiff = ns - point_num;
    uint8_t *alive = (uint8_t *)malloc((size_t)(ns > 0 ? ns : 1));
    memset(alive, 1, (size_t)(ns > 0 ? ns : 1));
    if (dTiff > 0 && ns >= 3) {
        /* K = 3 self-kNN among the samples: [0] self, [1] nearest, [2] second nearest (ties -> lower index) */
        int32_t *nn1 = (int32_t *)malloc(sizeof(int32_t) * (size_t)ns);
        float *d1 = (float *)malloc(sizeof(float) * (size_t)ns), *d2v = (float *)malloc(sizeof(float) * (size_t)ns);
        for (int64_t i = 0; i < ns; i++) {
            float bd[3] = {INFINITY, INFINITY, INFINITY}; int32_t bi[3] = {-1, -1, -1};
            const double *a = P + 3 * (int64_t)samples[i];
            float ax = (float)a[0], ay = (float)a[1], az = (float)a[2];
            for (int64_t j = 0; j < ns; j++) {
                const double *q = P + 3 * (int64_t)samples[j];
                float dx = ax - (float)q[0], dy = ay - (float)q[1], dz = az - (float)q[2];
                float d = (dx * dx + dy * dy) + dz * dz;
                if (d < bd[2]) {   /* insert, stable: equal distances keep the lower index first */
                    int pos = 2;
                    while (pos > 0 && d < bd[pos - 1]) { bd[pos] = bd[pos - 1]; bi[pos] = bi[pos - 1]; pos--; }
                    bd[pos] = d; bi[pos] = (int32_t)j;
                }
            }
            nn1[i] = bi[1]; d1[i] = sqrtf(bd[1]); d2v[i] = sqrtf(bd[2]);
        }
        while (dTiff > 0) {
            double mn = 9999; int64_t b1 = -1, b2 = -1;
            for (int64_t i = 0; i < ns; i++) {
                int64_t b2t = nn1[i];
                double dt = d1[i];
                if (dt < mn && alive[i] && alive[b2t]) { mn = dt; b1 = i; b2 = b2t; }
            }
            if (mn == 9999 || b1 == -1 || b2 == -1) break;
            int64_t del = b1;
            if ((double)d2v[b1] > (double)d2v[b2]) del = b2;
            alive[del] = 0;
            dTiff--;
        }
        free(nn1); free(d1); free(d2v);
    }
    int64_t nout = 0;
    for (int64_t i = 0; i < ns; i++)
        if (alive[i]) {
            if (nout >= cap) { nout = -4; break; }
            out_idx[nout++] = samples[i];
        }
    free(box_of); free(count); free(start); free(members); free(fill); free(center_pos); free(sim_num); free(labelG);
    free(simiT); free(mind); free(samples); free(alive);
    return nout;
}

/* ------------------------------------------------------------------------------------ */
/* exact k-NN and PCL-style normals                                                         */
/* ------------------------------------------------------------------------------------ */
void ko_knn_brute(const float *q, int64_t nq, const float *t, int64_t nt, int k, int32_t *idx, float *d2) {
#pragma omp parallel for schedule(static) if (nq >= 256)
    for (int64_t i = 0; i < nq; i++) {
        float *bd = d2 + i * k; int32_t *bi = idx + i * k;
        for (int c = 0; c < k; c++) { bd[c] = INFINITY; bi[c] = -1; }
        for (int64_t j = 0; j < nt; j++) {
            float d = dist2(q + 3 * i, t + 3 * j, 0);
            if (d < bd[k - 1]) {   /* ascending j: an equal distance never displaces an earlier (lower) index */
                int pos = k - 1;
                while (pos > 0 && d < bd[pos - 1]) { bd[pos] = bd[pos - 1]; bi[pos] = bi[pos - 1]; pos--; }
                bd[pos] = d; bi[pos] = (int32_t)j;
            }
        }
    }
}

static void pcl_roots2(float b, float c, float r[3]) {   /* computeRoots2 */
    r[0] = 0.f;
    float d = b * b - 4.0f * c;
    if (d < 0.0f) d = 0.0f;
    float sd = sqrtf(d);
    r[2] = 0.5f * (b + sd);
    r[1] = 0.5f * (b - sd);
}

static void pcl_roots(const float m[9], float r[3]) {   /* computeRoots, common/eigen.hpp */
    float c0 = m[0] * m[4] * m[8] + 2.0f * m[1] * m[2] * m[5] - m[0] * m[5] * m[5] - m[4] * m[2] * m[2] - m[8] * m[1] * m[1];
    float c1 = m[0] * m[4] - m[1] * m[1] + m[0] * m[8] - m[2] * m[2] + m[4] * m[8] - m[5] * m[5];
    float c2 = m[0] + m[4] + m[8];
    if (fabsf(c0) < FLT_EPSILON) { pcl_roots2(c2, c1, r); return; }
    const float s_inv3 = (float)(1.0 / 3.0), s_sqrt3 = sqrtf(3.0f);
    float c2_over_3 = c2 * s_inv3;
    float a_over_3 = (c1 - c2 * c2_over_3) * s_inv3;
    if (a_over_3 > 0.f) a_over_3 = 0.f;
    float half_b = 0.5f * (c0 + c2_over_3 * (2.0f * c2_over_3 * c2_over_3 - c1));
    float qq = half_b * half_b + a_over_3 * a_over_3 * a_over_3;
    if (qq > 0.f) qq = 0.f;
    float rho = sqrtf(-a_over_3);
    float theta = atan2f(sqrtf(-qq), half_b) * s_inv3;
    float ct = cosf(theta), st = sinf(theta);
    r[0] = c2_over_3 + 2.0f * rho * ct;
    r[1] = c2_over_3 - rho * (ct + s_sqrt3 * st);
    r[2] = c2_over_3 - rho * (ct - s_sqrt3 * st);
    float tmp;
    if (r[0] >= r[1]) { tmp = r[0]; r[0] = r[1]; r[1] = tmp; }
    if (r[1] >= r[2]) {
        tmp = r[1]; r[1] = r[2]; r[2] = tmp;
        if (r[0] >= r[1]) { tmp = r[0]; r[0] = r[1]; r[1] = tmp; }
    }
    if (r[0] <= 0) pcl_roots2(c2, c1, r);
}

void ko_normals_pcl(const double *P, int64_t n, int k, double *normals) {
    float *pf = (float *)malloc(sizeof(float) * 3 * (size_t)n);
    for (int64_t i = 0; i < 3 * n; i++) pf[i] = (float)P[i];   /* cloud_i.x = pointsVector[i][0] */
    if (k > n) k = (int)n;
    int32_t *idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)n * (size_t)k);
    float *d2 = (float *)malloc(sizeof(float) * (size_t)n * (size_t)k);
    ko_knn_brute(pf, n, pf, n, k, idx, d2);
    for (int64_t i = 0; i < n; i++) {
        /* computeMeanAndCovarianceMatrix (float, single pass) */
        float a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int c = 0; c < k; c++) {
            const float *p = pf + 3 * (int64_t)idx[i * k + c];
            a[0] += p[0] * p[0]; a[1] += p[0] * p[1]; a[2] += p[0] * p[2];
            a[3] += p[1] * p[1]; a[4] += p[1] * p[2]; a[5] += p[2] * p[2];
            a[6] += p[0]; a[7] += p[1]; a[8] += p[2];
        }
        for (int c = 0; c < 9; c++) a[c] /= (float)k;
        float m[9];
        m[0] = a[0] - a[6] * a[6]; m[1] = a[1] - a[6] * a[7]; m[2] = a[2] - a[6] * a[8];
        m[4] = a[3] - a[7] * a[7]; m[5] = a[4] - a[7] * a[8]; m[8] = a[5] - a[8] * a[8];
        m[3] = m[1]; m[6] = m[2]; m[7] = m[5];
        /* solvePlaneParameters -> eigen33 (smallest eigenvalue + its eigenvector) */
        float scale = 0.f;
        for (int c = 0; c < 9; c++) if (fabsf(m[c]) > scale) scale = fabsf(m[c]);
        if (scale <= FLT_MIN) scale = 1.0f;
        float sm[9];
        for (int c = 0; c < 9; c++) sm[c] = m[c] / scale;
        float r[3];
        pcl_roots(sm, r);
        sm[0] -= r[0]; sm[4] -= r[0]; sm[8] -= r[0];
        float v1[3] = {sm[1] * sm[5] - sm[2] * sm[4], sm[2] * sm[3] - sm[0] * sm[5], sm[0] * sm[4] - sm[1] * sm[3]};      /* row0 x row1 */
        float v2[3] = {sm[1] * sm[8] - sm[2] * sm[7], sm[2] * sm[6] - sm[0] * sm[8], sm[0] * sm[7] - sm[1] * sm[6]};      /* row0 x row2 */
        float v3[3] = {sm[4] * sm[8] - sm[5] * sm[7], sm[5] * sm[6] - sm[3] * sm[8], sm[3] * sm[7] - sm[4] * sm[6]};      /* row1 x row2 */
        float l1 = v1[0] * v1[0] + v1[1] * v1[1] + v1[2] * v1[2], l2 = v2[0] * v2[0] + v2[1] * v2[1] + v2[2] * v2[2],
              l3 = v3[0] * v3[0] + v3[1] * v3[1] + v3[2] * v3[2];
        float nx, ny, nz;
        if (l1 >= l2 && l1 >= l3) { float s = sqrtf(l1); nx = v1[0] / s; ny = v1[1] / s; nz = v1[2] / s; }
        else if (l2 >= l1 && l2 >= l3) { float s = sqrtf(l2); nx = v2[0] / s; ny = v2[1] / s; nz = v2[2] / s; }
        else { float s = sqrtf(l3); nx = v3[0] / s; ny = v3[1] / s; nz = v3[2] / s; }
        /* flipNormalTowardsViewpoint, view point (0, 0, 0) */
        const float *p = pf + 3 * i;
        float vx = 0.f - p[0], vy = 0.f - p[1], vz = 0.f - p[2];
        float cos_theta = vx * nx + vy * ny + vz * nz;
        if (cos_theta < 0) { nx *= -1; ny *= -1; nz *= -1; }
        /* normalCompute.hpp:342-348: renormalise in double */
        double dis = sqrt((double)nx * (double)nx + (double)ny * (double)ny + (double)nz * (double)nz);
        normals[3 * i] = nx / dis; normals[3 * i + 1] = ny / dis; normals[3 * i + 2] = nz / dis;
    }
    free(pf); free(idx); free(d2);
}

/* ---- estimateNormal_RegularNormal (normalCompute.hpp:614-742) ---------------------------------------------- */
void ko_normals_regular(const double *pts, int64_t n, double *normals) {
    if (n <= 0) return;
    const int Kn = 8;
    float *pf = (float *)malloc((size_t)n * 3 * sizeof(float));
    for (int64_t i = 0; i < 3 * n; i++) pf[i] = (float)pts[i];
    const int k = n < Kn ? (int)n : Kn;
    int32_t *ki = (int32_t *)malloc((size_t)n * Kn * sizeof(int32_t));
    float *kd = (float *)malloc((size_t)n * Kn * sizeof(float));
    ko_knn_brute(pf, n, pf, n, k, ki, kd);
    /* neighbour lists: drop the query itself (:660-665) */
    int32_t *nb = (int32_t *)malloc((size_t)n * (Kn - 1) * sizeof(int32_t));
    int *nnb = (int *)malloc((size_t)n * sizeof(int));
    for (int64_t i = 0; i < n; i++) {
        const int first = kd[(size_t)i * k] == 0 ? 1 : 0;
        int c = 0;
        for (int j = first; j < first + k - 1 && j < k; j++) nb[(size_t)i * (Kn - 1) + c++] = ki[(size_t)i * k + j];
        nnb[i] = c;
    }
    char *judge = (char *)calloc((size_t)n, 1);
    int64_t *level_of = (int64_t *)malloc((size_t)n * sizeof(int64_t));   /* "already listed in this level" stamp = the :694-699 scan */
    for (int64_t i = 0; i < n; i++) level_of[i] = -1;
    int32_t *cur = (int32_t *)malloc((size_t)n * sizeof(int32_t)), *nxt = (int32_t *)malloc((size_t)n * sizeof(int32_t));
    int32_t *par = (int32_t *)malloc((size_t)n * sizeof(int32_t));
    int64_t ncur = 1, level = 0;
    cur[0] = 0; judge[0] = 1;
    while (ncur > 0) {
        int64_t nn = 0;
        for (int64_t a = 0; a < ncur; a++) {
            const int32_t u = cur[a];
            for (int j = 0; j < nnb[u]; j++) {
                const int32_t v = nb[(size_t)u * (Kn - 1) + j];
                if (judge[v] || level_of[v] == level) continue;
                level_of[v] = level; nxt[nn] = v; par[nn] = u; nn++;
            }
        }
        for (int64_t a = 0; a < nn; a++) {
            const double *np_ = normals + 3 * (size_t)par[a];
            double *ns = normals + 3 * (size_t)nxt[a];
            double a1 = np_[0] * ns[0] + np_[1] * ns[1] + np_[2] * ns[2];
            double a2 = -np_[0] * ns[0] - np_[1] * ns[1] - np_[2] * ns[2];
            if (a1 > 1) a1 = 1;
            if (a1 < -1) a1 = -1;
            if (a2 > 1) a2 = 1;
            if (a2 < -1) a2 = -1;
            if (acos(a1) > acos(a2)) { ns[0] = -ns[0]; ns[1] = -ns[1]; ns[2] = -ns[2]; }
            judge[nxt[a]] = 1;
        }
        int32_t *t = cur; cur = nxt; nxt = t;
        ncur = nn; level++;
    }
    free(pf); free(ki); free(kd); free(nb); free(nnb); free(judge); free(level_of); free(cur); free(nxt); free(par);
}

/* ---- octree down-sampler (Method_Octree.hpp:77-165 over PCL 1.8.1's OctreePointCloudSearch) ---------------- */
typedef struct { double min[3], max[3], res; int depth; int defined; } ko_octbox;

static void ko_oct_first_point(ko_octbox *b, const float *p) {
    /* adoptBoundingBoxToPoint, empty octree: box = point +- resolution / 2, then getKeyBitSize() */
    const float minValue = FLT_EPSILON;
    for (int k = 0; k < 3; k++) { b->min[k] = p[k] - b->res / 2; b->max[k] = p[k] + b->res / 2; }
    unsigned int maxv = 2;
    for (int k = 0; k < 3; k++) {
        unsigned int mk = (unsigned int)ceil((b->max[k] - b->min[k] - minValue) / b->res);
        if (mk > maxv) maxv = mk;
    }
    b->depth = (int)ceil(log2((double)maxv) - minValue);
    const double side = (double)(1u << b->depth) * b->res;
    for (int k = 0; k < 3; k++) {   /* leaf_count_ == 0: centre the box inside the cube */
        const double over = (side - (b->max[k] - b->min[k])) / 2.0;
        if (over > minValue) { b->min[k] -= over; b->max[k] += over; }
    }
    b->defined = 1;
}

static int ko_oct_adopt(ko_octbox *b, const float *p) {
    const float minValue = FLT_EPSILON;
    for (;;) {
        if (!b->defined) { ko_oct_first_point(b, p); continue; }
        int lo[3], up[3], any = 0;
        for (int k = 0; k < 3; k++) { lo[k] = p[k] < b->min[k]; up[k] = p[k] >= b->max[k]; any |= lo[k] | up[k]; }
        if (!any) return 0;
        if (b->depth >= 21) return -1;   /* 63-bit Morton codes below; PCL itself shifts an int by the depth */
        /* one more level: the old cube becomes the child on the far side of every violated upper bound */
        double side = (double)(1u << b->depth) * b->res;
        for (int k = 0; k < 3; k++) if (!up[k]) b->min[k] -= side;
        b->depth++;
        side = (double)(1u << b->depth) * b->res - minValue;
        for (int k = 0; k < 3; k++) b->max[k] = b->min[k] + side;
    }
}

static int ko_cmp_u64(const void *a, const void *b) {
    const uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}

int64_t ko_octree_downsample(const double *pts, int64_t n, int32_t *out_idx, int64_t cap, double *resolution_out) {
    if (!pts || n < 1000 || !out_idx) return -1;
    float *pf = (float *)malloc((size_t)n * 3 * sizeof(float));
    for (int64_t i = 0; i < 3 * n; i++) pf[i] = (float)pts[i];                 /* cloud_i.x = pData[i][0]: narrowed */
    /* PCL_Octree_Resolution */
    int kn;
    if (n < 80000) kn = 2;
    else { const int md = (int)(double)(n / 80000); kn = md >= 5 ? 35 : 7 * md; }
    int32_t *ki = (int32_t *)malloc((size_t)1000 * kn * sizeof(int32_t));
    float *kd = (float *)malloc((size_t)1000 * kn * sizeof(float));
    ko_knn_brute(pf, 1000, pf, n, kn, ki, kd);
    double radiusSum = 0;
    for (int i = 0; i < 1000; i++) radiusSum = radiusSum + sqrt((double)kd[(size_t)i * kn + kn - 1]);
    radiusSum = radiusSum / 1000;
    const float resolution = (float)radiusSum;
    free(ki); free(kd);
    if (resolution_out) *resolution_out = (double)resolution;
    if (!(resolution > 0.f)) { free(pf); return -2; }                          /* coincident points: PCL asserts */
    /* addPointsFromInputCloud: grow the box in insertion order */
    ko_octbox b; memset(&b, 0, sizeof b); b.res = (double)resolution;
    for (int64_t i = 0; i < n; i++)
        if (ko_oct_adopt(&b, pf + 3 * i) != 0) { free(pf); return -3; }
    /* genOctreeKeyforPoint with the final box, packed so that ascending code == depth-first child order */
    uint64_t *code = (uint64_t *)malloc((size_t)n * sizeof(uint64_t));
    for (int64_t i = 0; i < n; i++) {
        unsigned int key[3];
        for (int k = 0; k < 3; k++) key[k] = (unsigned int)(((double)pf[3 * i + k] - b.min[k]) / b.res);
        uint64_t c = 0;
        for (int lev = b.depth - 1; lev >= 0; lev--)
            c = (c << 3) | (uint64_t)((((key[0] >> lev) & 1u) << 2) | (((key[1] >> lev) & 1u) << 1) | ((key[2] >> lev) & 1u));
        code[i] = c;
    }
    qsort(code, (size_t)n, sizeof(uint64_t), ko_cmp_u64);
    int64_t m = 0;
    for (int64_t i = 0; i < n; i++) if (i == 0 || code[i] != code[i - 1]) code[m++] = code[i];
    /* voxel centres (genLeafNodeCenterFromOctreeKey) and their nearest cloud point */
    float *cen = (float *)malloc((size_t)m * 3 * sizeof(float));
    for (int64_t v = 0; v < m; v++) {
        unsigned int key[3] = {0, 0, 0};
        for (int lev = 0; lev < b.depth; lev++) {
            const unsigned int tri = (unsigned int)((code[v] >> (3 * (b.depth - 1 - lev))) & 7u);
            key[0] = (key[0] << 1) | ((tri >> 2) & 1u); key[1] = (key[1] << 1) | ((tri >> 1) & 1u); key[2] = (key[2] << 1) | (tri & 1u);
        }
        for (int k = 0; k < 3; k++) cen[3 * v + k] = (float)(((double)key[k] + 0.5f) * b.res + b.min[k]);
    }
    int32_t *nn = (int32_t *)malloc((size_t)m * sizeof(int32_t));
    float *nd = (float *)malloc((size_t)m * sizeof(float));
    ko_kdtree *t = ko_kdtree_build(pf, n, 15);
    ko_kdtree_nn(t, cen, m, 0, 8, nn, nd);
    ko_kdtree_free(t);
    for (int64_t v = 0; v < m && v < cap; v++) out_idx[v] = nn[v];
    free(nn); free(nd); free(cen); free(code); free(pf);
    return m;
}

/* ------------------------------------------------------------------------------------ */
uint64_t ko_splitmix64(uint64_t seed, uint64_t counter) {
    uint64_t z = seed + (counter + 1) * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
