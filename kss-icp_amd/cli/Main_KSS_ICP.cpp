// kss_icp -- command-line front-end of the MI355X KSS-ICP core.
//
// Drives the mirror classes exactly the way the reference's only live entry point does
// (PS_AIS_Simplification/Main_KSS_ICP.cpp:73-91: load two PLYs, KSSICP_init(S, T, 8),
// KSSICP_Registration(1000), PCR_QM, append the aligned cloud to an .xyz file) and prints the same progress
// lines, but takes its paths from argv like the reference's shipped executables (EXE/Readme.txt:8-15) instead
// of the hard-coded E:// paths (:67-71).  It also prints the (s, R, t) the reference never emits.
//
//   usage: kss_icp source target [result.xyz]
// Inputs: ASCII or binary-little-endian .ply, or the reference's text clouds (count, then "x y z" rows: .xyz / .gird /
// .wlop); see cloud_io.hpp.
#include <cstdio>
#include <exception>
#include <iostream>
#include <string>
#include <vector>

#include "KSS_ICP.hpp"
#include "cloud_io.hpp"
#include "registrationMeasure.hpp"

namespace {

using kss_cli::Cloud;
using kss_cli::append_xyz;

int run(const std::string& src_path, const std::string& tgt_path, const std::string& out_path) {
    std::cout << "start!" << std::endl << "load ply:" << std::endl;
    const Cloud source = kss_cli::read_cloud(src_path);
    const Cloud target = kss_cli::read_cloud(tgt_path);
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
        std::cout << "usage: " << argv[0] << " source target [result.xyz]   (.ply ascii/binary, or count + xyz rows)" << std::endl;
        return 2;
    }
    try {
        return run(argv[1], argv[2], argc > 3 ? argv[3] : "Registration.xyz");
    } catch (const std::exception& e) {
        std::cout << "failed: " << e.what() << std::endl;   // no GPU / HIP error: there is no CPU fallback
        return 1;
    }
}
