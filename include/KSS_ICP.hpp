// include/KSS_ICP.hpp -- MI355X mirror of the reference class `KSSICP`
// (PS_AIS_Simplification/KSS_ICP.hpp:38-393).  Same public methods, argument meaning, stdout lines and
// the public field `pointAlign`, so the front-end call sequence
//     KSSICP ki; ki.KSSICP_init(S, T, 8); ki.KSSICP_Registration(1000); A = ki.pointAlign;
// (Main_KSS_ICP.cpp:79-82) compiles unchanged.  Every pcl::IterativeClosestPoint block of the reference
// (:155-162 and four copies) becomes one kss_icp call; KSSICP_Registration's pose search + candidate ICP
// batch + final ICP (:86-131) is one kss_register call.  Written from scratch; no reference code.
//
// Down-sampling (:71-81): AIVS on the device (kss_downsample_aivs: the reference's voxel grid, 8-colour per-voxel
// farthest-point sampling and accurate cut).  Clouds the reference cannot voxelise (zero extent along an axis:
// it divides by zero) fall back to kss_downsample_fps, exact farthest-point sampling.
#pragma once
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "initRegistrationKSS.hpp"
#include "kss_runtime.hpp"

class KSSICP {
private:
    std::vector<std::vector<double>> pointSource;
    std::vector<std::vector<double>> pointTarget;
    int pNumber = 0;
    double accurateG = 8;

public:
    std::vector<std::vector<double>> pointAlign;
    kss_register_result lastRegistration;   // extra: (R, t, s) of the last KSSICP_Registration (reference emits none)

public:
    void KSSICP_init(std::vector<std::vector<double>> ps, std::vector<std::vector<double>> pt, double accurate) {
        accurateG = accurate;
        pointSource = ps;
        pointTarget = pt;
        pNumber = (int)(pointSource.size() > pointTarget.size() ? pointTarget.size() : pointSource.size());
        pNumber = pNumber / 2;             // :63 integer division
        if (pNumber > 2000) pNumber = 2000;   // :64-66
    }

    void KSSICP_Registration(int iter) {
        std::vector<std::vector<double>> pointCloudT, pointCloudS;                               // :71-75 (target), :77-81 (source)
        downsample_both(pointTarget, pointSource, pNumber, pointCloudT, pointCloudS);
        std::cout << "initRegistration start." << std::endl;
        std::vector<double> s = kss_host::pack(pointCloudS), t = kss_host::pack(pointCloudT), f = kss_host::pack(pointSource);
        std::vector<double> align(f.size());
        kss_host::Runtime::check(kss_register(kss_host::Runtime::ctx(), s.data(), (int64_t)pointCloudS.size(), t.data(),
                                              (int64_t)pointCloudT.size(), f.data(), (int64_t)pointSource.size(), accurateG,
                                              iter, align.data(), &lastRegistration), "kss_register");
        std::cout << "i:" << lastRegistration.angle[0] << "j:" << lastRegistration.angle[1] << "k:" << lastRegistration.angle[2] << std::endl;
        std::cout << "has converged: " << lastRegistration.icp_converged << std::endl;
        std::cout << "score: " << lastRegistration.final_fitness << std::endl;
        print_matrix(lastRegistration.T_icp);
        // :127-130: pointSource becomes the pose-aligned full-resolution source, pointAlign = M * pointSource
        kss_pose p = pose_of(lastRegistration);
        std::vector<double> posed(f.size());
        kss_host::Runtime::check(kss_pose_apply(kss_host::Runtime::ctx(), f.data(), (int64_t)pointSource.size(), &p, posed.data()), "kss_pose_apply");
        pointSource = kss_host::unpack(posed);
        pointAlign = kss_host::unpack(align);
    }

    // full-resolution ICP on the members (:133-183): pointAlign = PCL's float output cloud
    double shapeRegistration_ICP(int iter) {
        std::vector<float> s = kss_host::pack_f32(pointSource), t = kss_host::pack_f32(pointTarget);
        kss_icp_result r = run_icp(iter, s, t, true);
        std::vector<float> out(s.size());
        kss_host::Runtime::check(kss_transform_apply_f32(kss_host::Runtime::ctx(), r.T, s.data(), (int64_t)pointSource.size(), out.data()), "kss_transform_apply_f32");
        pointAlign = kss_host::unpack_f32(out);
        return r.fitness;
    }

    // ICP on (ps, pt), then pointAlign = Matrix4f * member pointSource in double (:185-233)
    double shapeRegistration_ICP(int iter, std::vector<std::vector<double>> ps, std::vector<std::vector<double>> pt) {
        std::vector<float> s = kss_host::pack_f32(ps), t = kss_host::pack_f32(pt);
        kss_icp_result r = run_icp(iter, s, t, true);
        std::vector<double> f = kss_host::pack(pointSource), out(f.size());
        kss_host::Runtime::check(kss_transform_apply(kss_host::Runtime::ctx(), r.T, f.data(), (int64_t)pointSource.size(), out.data()), "kss_transform_apply");
        pointAlign = kss_host::unpack(out);
        return r.fitness;
    }

    // fitness only; Q is unused in the reference too (:236-274)
    double shapeRegistration_ICP_AngleList(int iter, double Q, std::vector<std::vector<double>> ps, std::vector<std::vector<double>> pt) {
        (void)Q;
        std::vector<float> s = kss_host::pack_f32(ps), t = kss_host::pack_f32(pt);
        return run_icp(iter, s, t, false).fitness;
    }

    // the registered cloud of (ps -> pt) (:276-321)
    std::vector<std::vector<double>> shapeRegistration_ICP_AngleListV(int iter, double Q, std::vector<std::vector<double>> ps,
                                                                      std::vector<std::vector<double>> pt) {
        (void)Q;
        std::vector<float> s = kss_host::pack_f32(ps), t = kss_host::pack_f32(pt);
        kss_icp_result r = run_icp(iter, s, t, false);
        std::vector<float> out(s.size());
        kss_host::Runtime::check(kss_transform_apply_f32(kss_host::Runtime::ctx(), r.T, s.data(), (int64_t)ps.size(), out.data()), "kss_transform_apply_f32");
        return kss_host::unpack_f32(out);
    }

    double shapeRegistration_ICP_Judge(int iter, std::vector<std::vector<double>> ps, std::vector<std::vector<double>> pt) {   // :323-356
        std::vector<float> s = kss_host::pack_f32(ps), t = kss_host::pack_f32(pt);
        return run_icp(iter, s, t, false).fitness;
    }

    std::vector<std::vector<double>> IntrinsicICP_pointSource() { return pointSource; }
    std::vector<std::vector<double>> IntrinsicICP_pointTarget() { return pointTarget; }

private:
    // the similarity + Euler pose chosen by kss_register (shift = c_T - c_S, centre = c_T, :186-192)
    static kss_pose pose_of(const kss_register_result& r) {
        kss_pose p;
        for (int k = 0; k < 3; ++k) { p.shift[k] = r.c_tgt[k] - r.c_src[k]; p.center[k] = r.c_tgt[k]; p.angle[k] = r.angle[k]; }
        p.scale = r.scale;
        return p;
    }

    static std::vector<std::vector<double>> downsample(const std::vector<std::vector<double>>& cloud, int m) {
        if (m <= 0 || cloud.empty()) return cloud;
        std::vector<double> in = kss_host::pack(cloud), out(in.size());
        int64_t k = 0;
        const int rc = kss_downsample_aivs(kss_host::Runtime::ctx(), in.data(), (int64_t)cloud.size(), m, out.data(), (int64_t)cloud.size(), &k, nullptr);
        if (rc == KSS_ERR_ARG && (size_t)m < cloud.size()) {   // degenerate extent: the reference would divide by zero
            std::cout << "AIVS: degenerate cloud, farthest-point sampling instead" << std::endl;
            kss_host::Runtime::check(kss_downsample_fps(kss_host::Runtime::ctx(), in.data(), (int64_t)cloud.size(), m, out.data(), nullptr), "kss_downsample_fps");
            k = m;
        } else {
            kss_host::Runtime::check(rc, "kss_downsample_aivs");
        }
        out.resize((size_t)k * 3);
        return kss_host::unpack(out);
    }

    // the two down-samplings of :71-81 do not depend on each other: one call, both clouds at once (kss_downsample_aivs_pair)
    static void downsample_both(const std::vector<std::vector<double>>& a, const std::vector<std::vector<double>>& b, int m,
                                std::vector<std::vector<double>>& out_a, std::vector<std::vector<double>>& out_b) {
        if (m <= 0 || a.empty() || b.empty()) { out_a = downsample(a, m); out_b = downsample(b, m); return; }
        std::vector<double> ia = kss_host::pack(a), ib = kss_host::pack(b), oa(ia.size()), ob(ib.size());
        int64_t ka = 0, kb = 0;
        int rc[2] = {KSS_OK, KSS_OK};
        kss_host::Runtime::check(kss_downsample_aivs_pair(kss_host::Runtime::ctx(), ia.data(), (int64_t)a.size(), m, oa.data(), (int64_t)a.size(), &ka, nullptr,
                                                          ib.data(), (int64_t)b.size(), m, ob.data(), (int64_t)b.size(), &kb, nullptr, rc), "kss_downsample_aivs_pair");
        if (rc[0] != KSS_OK) out_a = downsample(a, m);     // (the one-cloud path handles a degenerate cloud and reports anything else)
        else { oa.resize((size_t)ka * 3); out_a = kss_host::unpack(oa); }
        if (rc[1] != KSS_OK) out_b = downsample(b, m);
        else { ob.resize((size_t)kb * 3); out_b = kss_host::unpack(ob); }
    }

    static void print_matrix(const float* T) {
        for (int r = 0; r < 4; ++r) std::cout << T[4 * r] << " " << T[4 * r + 1] << " " << T[4 * r + 2] << " " << T[4 * r + 3] << std::endl;
    }

    kss_icp_result run_icp(int iter, const std::vector<float>& s, const std::vector<float>& t, bool verbose) {
        kss_icp_params p;
        kss_icp_default_params(&p);   // MaxCorrespondenceDistance 1, TransformationEpsilon 1e-10, EuclideanFitnessEpsilon 1e-3
        p.max_iterations = iter;
        kss_icp_result r;
        kss_host::Runtime::check(kss_icp(kss_host::Runtime::ctx(), s.data(), (int64_t)(s.size() / 3), t.data(), (int64_t)(t.size() / 3), &p, &r), "kss_icp");
        if (verbose) {   // :165-167
            std::cout << "has converged: " << r.converged << std::endl;
            std::cout << "score: " << r.fitness << std::endl;
            print_matrix(r.T);
        }
        return r;
    }

    void save_PointCloud(std::vector<std::vector<double>> pointCloud, std::string Path) {   // :381-391 (appends)
        std::ofstream fout(Path, std::ios::app);
        fout << pointCloud.size() << std::endl;
        for (size_t i = 0; i < pointCloud.size(); i++) fout << pointCloud[i][0] << " " << pointCloud[i][1] << " " << pointCloud[i][2] << std::endl;
        fout << std::endl;
        fout.close();
    }
};
