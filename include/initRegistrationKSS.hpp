// include/initRegistrationKSS.hpp -- MI355X mirror of the reference class `initRegistration_KSS`
// (PS_AIS_Simplification/initRegistrationKSS.hpp:28-524).  Same public names, argument meaning and
// side effects (stdout lines included); the arithmetic runs on the GPU through the C-ABI (kssicp.h):
//   initRegistration_MiddleAlign (:144-220)  -> kss_preshape_stats x2 + kss_pose_apply (angle 0)
//   initRegistration_Rotation()  (:222-296)  -> kss_rotation_search + kss_rotation_candidates
//   initRegistration_Rotation[_Angle](cloud) (:75-109) -> kss_pose_apply
// Dropped member: `pcl::KdTreeFLANN<pcl::PointXYZ> kdtree` (:49) -- the brute-force device sweep needs
// no index structure.  Written from scratch against the reference's behaviour; no reference code.
#pragma once
#include <chrono>
#include <cmath>
#include <iostream>
#include <vector>

#include "kss_runtime.hpp"

class initRegistration_KSS {
private:
    double step = 8;
    std::vector<double> value;   // g*g*g error volume, i-major (reference: vector<vector<vector<double>>>)
    int irange = 0, jrange = 0, krange = 0;
    int r = 2;   // kernel radius (:35)

public:
    double x_middle_S = 0, y_middle_S = 0, z_middle_S = 0;   // target middle point (c_T)
    double x_middle = 0, y_middle = 0, z_middle = 0;         // source-to-target middle vector (c_T - c_S)
    double scale = 1;                                        // source transfer scale
    std::vector<double> angle;                               // best grid angles (accumulated doubles)
    std::vector<std::vector<double>> angleList;              // 5^3 local minima, idx*6.3/step
    std::vector<std::vector<double>> pointSource;
    std::vector<std::vector<double>> pointTarget;
    std::vector<double> rotationRecord;                      // unused in the reference as well

public:
    void initRegistration_init(std::vector<std::vector<double>> pointinput,
                               std::vector<std::vector<double>> pointinput2, double accurate) {
        step = accurate;
        std::cout << "initRegistration start." << std::endl;
        pointSource = pointinput;
        pointTarget = pointinput2;
        std::cout << "initRegistration middle align." << std::endl;
        initRegistration_MiddleAlign();
        std::cout << "initRegistration rotation." << std::endl;
        const auto t0 = std::chrono::steady_clock::now();
        initRegistration_Rotation();
        const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::cout << "alignment time cost:" << sec << "s" << std::endl;
    }

    std::vector<std::vector<double>> initRegistration_Rotation(std::vector<std::vector<double>> sourceOri) {
        return apply_pose(sourceOri, angle.size() >= 3 ? angle.data() : zero3());
    }

    std::vector<std::vector<double>> initRegistration_Rotation_Angle(std::vector<std::vector<double>> sourceOri,
                                                                     std::vector<double> angle_T) {
        return apply_pose(sourceOri, angle_T.data());
    }

    // :111-140.  The reference subtracts/adds x_middle_S on ALL three axes (a bug it never exercises:
    // the function has no caller); reproduced as written.
    std::vector<std::vector<double>> initRegistration_Rotation_Axis(std::vector<std::vector<double>> sourceOri,
                                                                    int axis, double angleV) {
        if (axis < 1 || axis > 3) {
            std::cout << "error! illegal rotation" << std::endl;
            for (auto& p : sourceOri) { p[0] -= x_middle_S; p[1] -= x_middle_S; p[2] -= x_middle_S; }
            return sourceOri;
        }
        kss_pose a;
        for (int k = 0; k < 3; ++k) { a.shift[k] = -x_middle_S; a.center[k] = 0; a.angle[k] = 0; }
        a.scale = 1;
        a.angle[axis - 1] = angleV;
        std::vector<double> in = kss_host::pack(sourceOri), mid(in.size()), out(in.size());
        kss_host::Runtime::check(kss_pose_apply(kss_host::Runtime::ctx(), in.data(), (int64_t)sourceOri.size(), &a, mid.data()), "kss_pose_apply");
        kss_pose b;
        for (int k = 0; k < 3; ++k) { b.shift[k] = x_middle_S; b.center[k] = 0; b.angle[k] = 0; }
        b.scale = 1;
        kss_host::Runtime::check(kss_pose_apply(kss_host::Runtime::ctx(), mid.data(), (int64_t)sourceOri.size(), &b, out.data()), "kss_pose_apply");
        return kss_host::unpack(out);
    }

    // accessors the reference exposes only through `#define private public` style probing
    const std::vector<double>& errorVolume() const { return value; }
    int gridSize() const { return irange; }

private:
    static const double* zero3() { static const double z[3] = {0, 0, 0}; return z; }

    kss_pose pose_of(const double* ang) const {
        kss_pose p;
        p.shift[0] = x_middle; p.shift[1] = y_middle; p.shift[2] = z_middle;
        p.center[0] = x_middle_S; p.center[1] = y_middle_S; p.center[2] = z_middle_S;
        p.scale = scale;
        p.angle[0] = ang[0]; p.angle[1] = ang[1]; p.angle[2] = ang[2];
        return p;
    }

    std::vector<std::vector<double>> apply_pose(const std::vector<std::vector<double>>& cloud, const double* ang) {
        if (cloud.empty()) return cloud;
        const kss_pose p = pose_of(ang);
        std::vector<double> in = kss_host::pack(cloud), out(in.size());
        kss_host::Runtime::check(kss_pose_apply(kss_host::Runtime::ctx(), in.data(), (int64_t)cloud.size(), &p, out.data()), "kss_pose_apply");
        return kss_host::unpack(out);
    }

    void initRegistration_MiddleAlign() {
        kss_ctx* c = kss_host::Runtime::ctx();
        std::vector<double> s = kss_host::pack(pointSource), t = kss_host::pack(pointTarget);
        double cS[3], cT[3], rS = 0, rT = 0;
        kss_host::Runtime::check(kss_preshape_stats(c, s.data(), KSS_F64, (int64_t)pointSource.size(), cS, &rS), "kss_preshape_stats(source)");
        kss_host::Runtime::check(kss_preshape_stats(c, t.data(), KSS_F64, (int64_t)pointTarget.size(), cT, &rT), "kss_preshape_stats(target)");
        x_middle_S = cT[0]; y_middle_S = cT[1]; z_middle_S = cT[2];
        x_middle = cT[0] - cS[0]; y_middle = cT[1] - cS[1]; z_middle = cT[2] - cS[2];
        scale = rT / rS;
        pointSource = apply_pose(pointSource, zero3());   // :212-219 "scale uniform"
    }

    void initRegistration_Rotation() {
        kss_ctx* c = kss_host::Runtime::ctx();
        std::vector<double> s = kss_host::pack(pointSource), t = kss_host::pack(pointTarget);
        std::vector<double> err(40 * 40 * 40);
        int g = 0;
        kss_host::Runtime::check(kss_rotation_search(c, s.data(), (int64_t)pointSource.size(), t.data(), (int64_t)pointTarget.size(),
                                                     step, err.data(), (int64_t)err.size(), &g), "kss_rotation_search");
        value.assign(err.begin(), err.begin() + (size_t)g * g * g);
        irange = jrange = krange = g;
        std::vector<double> list((size_t)3 * g * g * g);
        double best[3];
        int nl = 0;
        kss_host::Runtime::check(kss_rotation_candidates(value.data(), g, step, best, list.data(), g * g * g, &nl), "kss_rotation_candidates");
        for (int i = 0; i < nl; ++i) angleList.push_back({list[3 * i], list[3 * i + 1], list[3 * i + 2]});   // appended, as :285
        angle.push_back(best[0]); angle.push_back(best[1]); angle.push_back(best[2]);                        // :291-293
        std::cout << "i:" << best[0] << "j:" << best[1] << "k:" << best[2] << std::endl;
        std::cout << std::endl;
    }
};
