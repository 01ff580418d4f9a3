// kss_icp -- command-line front-end of the MI355X KSS-ICP core.
//
// Drives the mirror classes exactly the way the reference's only live entry point does
// (PS_AIS_Simplification/Main_KSS_ICP.cpp:73-91: load two PLYs, KSSICP_init(S, T, 8),
// KSSICP_Registration(1000), PCR_QM, append the aligned cloud to an .xyz file) and prints the same progress
// lines, but takes its paths from argv like the reference's shipped executables (EXE/Readme.txt:8-15) instead
// of the hard-coded E:// paths (:67-71).  It also prints the (s, R, t) the reference never emits.
//
//   usage: kss_icp source.ply target.ply [result.xyz]
#include <cstdio>
#include <exception>
#include <iostream>
#include <string>
#include <vector>

#include "KSS_ICP.hpp"
#include "PlyLoad.h"
#include "registrationMeasure.hpp"

namespace {

typedef std::vector<std::vector<double>> Cloud;

Cloud read_ply_vertices(const std::string& path) {
    std::vector<char> name(path.begin(), path.end());
    name.push_back('\0');
    CPLYLoader loader;              // same loader class name / LoadModel(char*) contract as the reference
    loader.LoadModel(name.data());
    return loader.points;
}

// ".xyz": first line N, then "x y z" per line, then an empty line; opened in APPEND mode like the reference
// (Main_KSS_ICP.cpp:49-59), so running twice into the same file stacks two clouds.
bool append_xyz(const Cloud& cloud, const std::string& path) {
    std::FILE* f = std::fopen(path.c_str(), "a");
    if (!f) return false;
    std::fprintf(f, "%zu\n", cloud.size());
    for (const std::vector<double>& p : cloud) std::fprintf(f, "%g %g %g\n", p[0], p[1], p[2]);
    std::fprintf(f, "\n");
    std::fclose(f);
    return true;
}

int run(const std::string& src_path, const std::string& tgt_path, const std::string& out_path) {
    std::cout << "start!" << std::endl << "load ply:" << std::endl;
    const Cloud source = read_ply_vertices(src_path);
    const Cloud target = read_ply_vertices(tgt_path);
    std::cout << "load ply finished." << std::endl;
    if (source.empty() || target.empty()) {
        std::cout << "empty point cloud" << std::endl;
        return 1;
    }
    std::cout << "registration runing." << std::endl;   // (sic) the reference's spelling

    KSSICP registration;
    registration.KSSICP_init(source, target, 8);
    registration.KSSICP_Registration(1000);
    const Cloud aligned = registration.pointAlign;
    std::cout << "registration finished." << std::endl;

    const kss_register_result& r = registration.lastRegistration;
    std::printf("scale: %.9g\n", r.scale);
    std::printf("R: %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g\n", r.R[0], r.R[1], r.R[2], r.R[3], r.R[4], r.R[5], r.R[6], r.R[7], r.R[8]);
    std::printf("t: %.9g %.9g %.9g\n", r.t[0], r.t[1], r.t[2]);
    std::fflush(stdout);

    std::cout << "Measurement:" << std::endl;
    PCR_QM quality;
    quality.PCR_QM_init(aligned, target);
    const std::vector<double> m = quality.PCR_QM_ReturnResult();
    std::cout << "Registration Measure" << ":" << "MSE: " << m[0] << " RMSE: " << m[1] << " MAE: " << m[2] << std::endl;
    if (!append_xyz(aligned, out_path)) {
        std::cout << "cannot write " << out_path << std::endl;
        return 1;
    }
    return 0;
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 3) {
        std::cout << "usage: " << argv[0] << " source.ply target.ply [result.xyz]" << std::endl;
        return 2;
    }
    try {
        return run(argv[1], argv[2], argc > 3 ? argv[3] : "Registration.xyz");
    } catch (const std::exception& e) {
        std::cout << "failed: " << e.what() << std::endl;   // no GPU / HIP error: there is no CPU fallback
        return 1;
    }
}
