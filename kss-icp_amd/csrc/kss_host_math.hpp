// kss_host_math.hpp -- host-side small math of the registration core (product code).
//
// The north star solves the 3x3 SVD / rotation on the host: everything here is O(1) per ICP
// iteration.  Float Matrix4f helpers follow Eigen's evaluation order without fused
// multiply-add, as pcl::IterativeClosestPoint (Scalar=float) evaluates them.
// References: path:line under PS_AIS_Simplification/ and PCL 1.8.1.
#pragma once
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace kss {

constexpr int NSUMS = 20;

// ---- 3x3 SVD by one-sided (Hestenes) Jacobi: A = U diag(s) V^T, row-major ----------------
inline double det3(const double M[9]) {
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) +
           M[2] * (M[3] * M[7] - M[4] * M[6]);
}

inline void svd3(const double A[9], double U[9], double s[3], double V[9]) {
    double B[9], W[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    std::memcpy(B, A, sizeof B);
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool rotated = false;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double alpha = 0, beta = 0, gamma = 0;
                for (int k = 0; k < 3; ++k) {
                    alpha += B[3 * k + p] * B[3 * k + p];
                    beta += B[3 * k + q] * B[3 * k + q];
                    gamma += B[3 * k + p] * B[3 * k + q];
                }
                if (gamma == 0.0 || std::fabs(gamma) <= 1e-15 * std::sqrt(alpha * beta)) continue;
                rotated = true;
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / std::sqrt(1.0 + t * t), sn = c * t;
                for (int k = 0; k < 3; ++k) {
                    const double bp = B[3 * k + p], bq = B[3 * k + q];
                    B[3 * k + p] = c * bp - sn * bq;
                    B[3 * k + q] = sn * bp + c * bq;
                    const double wp = W[3 * k + p], wq = W[3 * k + q];
                    W[3 * k + p] = c * wp - sn * wq;
                    W[3 * k + q] = sn * wp + c * wq;
                }
            }
        if (!rotated) break;
    }
    double nrm[3];
    int ord[3] = {0, 1, 2};
    for (int j = 0; j < 3; ++j)
        nrm[j] = std::sqrt(B[j] * B[j] + B[3 + j] * B[3 + j] + B[6 + j] * B[6 + j]);
    for (int a = 0; a < 2; ++a)
        for (int b = a + 1; b < 3; ++b)
            if (nrm[ord[b]] > nrm[ord[a]]) { int tmp = ord[a]; ord[a] = ord[b]; ord[b] = tmp; }
    double u[3][3];
    bool ok[3];
    for (int c = 0; c < 3; ++c) {
        const int j = ord[c];
        s[c] = nrm[j];
        ok[c] = nrm[j] > 1e-13 * (nrm[ord[0]] > 0 ? nrm[ord[0]] : 1.0) && nrm[j] > 0;
        for (int r = 0; r < 3; ++r) {
            V[3 * r + c] = W[3 * r + j];
            u[c][r] = ok[c] ? B[3 * r + j] / nrm[j] : 0.0;
        }
    }
    // complete a rank-deficient U orthonormally (planar / collinear correspondences)
    if (!ok[0]) { u[0][0] = 1; u[0][1] = 0; u[0][2] = 0; }
    if (!ok[1]) {
        int m = 0;
        if (std::fabs(u[0][1]) < std::fabs(u[0][m])) m = 1;
        if (std::fabs(u[0][2]) < std::fabs(u[0][m])) m = 2;
        double a[3] = {0, 0, 0};
        a[m] = 1;
        const double d = u[0][m];
        double n2 = 0;
        for (int r = 0; r < 3; ++r) { u[1][r] = a[r] - d * u[0][r]; n2 += u[1][r] * u[1][r]; }
        n2 = std::sqrt(n2);
        for (int r = 0; r < 3; ++r) u[1][r] /= n2;
    }
    if (!ok[2]) {
        u[2][0] = u[0][1] * u[1][2] - u[0][2] * u[1][1];
        u[2][1] = u[0][2] * u[1][0] - u[0][0] * u[1][2];
        u[2][2] = u[0][0] * u[1][1] - u[0][1] * u[1][0];
    }
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) U[3 * r + c] = u[c][r];
}

// ---- TransformationEstimationSVD -> Eigen umeyama(src, dst, with_scaling=false) -----------
// sigma = (1/n) sum (dst - mu_d)(src - mu_s)^T, R = U diag(1,1,+-1) V^T, t = mu_d - R mu_s.
inline void rigid_from_sums(const double sums[NSUMS], float T[16]) {
    const double n = sums[0];
    const double mu_s[3] = {sums[1] / n, sums[2] / n, sums[3] / n};
    const double mu_d[3] = {sums[4] / n, sums[5] / n, sums[6] / n};
    double sigma[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) sigma[3 * i + j] = sums[7 + 3 * j + i] / n - mu_d[i] * mu_s[j];
    double U[9], sv[3], V[9];
    svd3(sigma, U, sv, V);
    const double sgn = (det3(U) * det3(V) < 0) ? -1.0 : 1.0;
    double R[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            R[3 * i + j] = U[3 * i] * V[3 * j] + U[3 * i + 1] * V[3 * j + 1] + sgn * U[3 * i + 2] * V[3 * j + 2];
    for (int i = 0; i < 3; ++i) {
        const double ti = mu_d[i] - (R[3 * i] * mu_s[0] + R[3 * i + 1] * mu_s[1] + R[3 * i + 2] * mu_s[2]);
        T[4 * i + 0] = (float)R[3 * i + 0];
        T[4 * i + 1] = (float)R[3 * i + 1];
        T[4 * i + 2] = (float)R[3 * i + 2];
        T[4 * i + 3] = (float)ti;
    }
    T[12] = T[13] = T[14] = 0.f;
    T[15] = 1.f;
}

// ---- Matrix4f product, Eigen order: ((a0*b0 + a1*b1) + a2*b2) + a3*b3, no fma -------------
inline void mat4_mul(const float A[16], const float B[16], float C[16]) {
    float R[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            volatile float acc = A[4 * i] * B[j];   // volatile: forbid contraction by any host compiler
            acc = acc + A[4 * i + 1] * B[4 + j];
            acc = acc + A[4 * i + 2] * B[8 + j];
            acc = acc + A[4 * i + 3] * B[12 + j];
            R[4 * i + j] = acc;
        }
    std::memcpy(C, R, sizeof R);
}

inline void mat4_identity(float T[16]) {
    for (int i = 0; i < 16; ++i) T[i] = (i % 5 == 0) ? 1.f : 0.f;
}

// ---- pcl::registration::DefaultConvergenceCriteria<float>::hasConverged (PCL 1.8.1) -------
struct Convergence {
    int max_iterations = 1000;
    double rotation_threshold = 1.0 - 1e-10;   // setRotationThreshold(1 - transformation_epsilon)
    double translation_threshold = 1e-10;      // setTranslationThreshold(transformation_epsilon)
    double mse_rel = 1e-3;                     // setRelativeMSE(euclidean_fitness_epsilon)
    double mse_abs = 1e-12;
    bool fixed_iterations = false;
    double prev_mse = DBL_MAX;                 // correspondences_prev_mse_
    int state = 0;

    // iterations = nr_iterations_ AFTER the increment; Tk = this iteration's transformation_
    bool has_converged(int iterations, const float Tk[16], double cur_mse) {
        state = 0;
        if (iterations >= max_iterations) { state = 1; return true; }
        if (fixed_iterations) return false;
        volatile float tr = Tk[0] + Tk[5];
        tr = tr + Tk[10];
        tr = tr - 1.f;
        const double cos_angle = 0.5 * (double)tr;
        volatile float tsq = Tk[3] * Tk[3];
        tsq = tsq + Tk[7] * Tk[7];
        tsq = tsq + Tk[11] * Tk[11];
        const double translation_sqr = (double)tsq;
        if (cos_angle >= rotation_threshold && translation_sqr <= translation_threshold) { state = 2; return true; }
        if (std::fabs(cur_mse - prev_mse) < mse_abs) { state = 3; return true; }
        if (std::fabs(cur_mse - prev_mse) / prev_mse < mse_rel) { state = 4; return true; }
        prev_mse = cur_mse;
        return false;
    }
};

// ---- rotation-search bookkeeping: initRegistrationKSS.hpp:245, :258-265, :276-293, :481-522 --
inline int grid_angles(double step, double* out, int cap) {
    int g = 0;
    for (double a = 0; a < 6.3; a = a + 6.3 / step) {   // double accumulation, as the reference
        if (g >= cap) return -1;
        out[g++] = a;
    }
    return g;
}

inline bool is_local_min(const double* v, int g, int i, int j, int k, int r) {
    const double c = v[((int64_t)i * g + j) * g + k];
    const int i0 = i - r < 0 ? 0 : i - r, i1 = i + r >= g ? g - 1 : i + r;
    const int j0 = j - r < 0 ? 0 : j - r, j1 = j + r >= g ? g - 1 : j + r;
    const int k0 = k - r < 0 ? 0 : k - r, k1 = k + r >= g ? g - 1 : k + r;
    for (int a = i0; a <= i1; ++a)
        for (int b = j0; b <= j1; ++b)
            for (int d = k0; d <= k1; ++d)
                if (c > v[((int64_t)a * g + b) * g + d]) return false;   // non-strict: plateaus all pass
    return true;
}

// R0 = Rz(a2) * Ry(a1) * Rx(a0), reference axis conventions (:365-404)
inline void euler_matrix(const double a[3], double R[9]) {
    const double cx = std::cos(a[0]), sx = std::sin(a[0]);
    const double cy = std::cos(a[1]), sy = std::sin(a[1]);
    const double cz = std::cos(a[2]), sz = std::sin(a[2]);
    // Ry*Rx
    const double M[9] = {cy, sy * sx, sy * cx, 0, cx, -sx, -sy, cy * sx, cy * cx};
    for (int j = 0; j < 3; ++j) {
        R[j] = cz * M[j] - sz * M[3 + j];
        R[3 + j] = sz * M[j] + cz * M[3 + j];
        R[6 + j] = M[6 + j];
    }
}

}  // namespace kss
