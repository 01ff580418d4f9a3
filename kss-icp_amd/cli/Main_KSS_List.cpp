// kss_icp_list -- batch front-end: registers a list of objects, one KSSICP per object, and prints the per-object
// time and quality tables.
//
// Follows the reference's batch main (PS_AIS_Simplification/Main_KSS_List.cpp:127-181, shipped commented out): for
// each name: load <name>source / <name>target, KSSICP_init(S, T, 8), KSSICP_Registration(1000), save the aligned
// and the target cloud as <name>Align.xyz / <name>Target.xyz, PCR_QM; then "name:time: t" and
// "name:MSE: .. RMSE: .. MAE: .." tables.  The reference's object list and E:// directories are hard-coded (its list
// is empty); here they come from argv:
//
//   usage: kss_icp_list <input dir> <output dir> <name> [<name> ...]
// For every name the first existing of  <name>source.ply, <name>source.xyz, <name>.gird  is the source and
// <name>target.ply, <name>target.xyz, <name>.wlop  the target (the last pair is how data/registration/ names them).
#include <chrono>
#include <cstdio>
#include <exception>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "KSS_ICP.hpp"
#include "cloud_io.hpp"
#include "registrationMeasure.hpp"

namespace {

using kss_cli::Cloud;

std::string first_existing(const std::string& dir, const std::string& name, const char* const suffixes[3]) {
    for (int k = 0; k < 3; ++k) {
        const std::string p = dir + "/" + name + suffixes[k];
        if (std::ifstream(p).good()) return p;
    }
    return std::string();
}

struct Row { std::string name; double seconds; std::vector<double> measure; };

int run(const std::string& in_dir, const std::string& out_dir, const std::vector<std::string>& names) {
    static const char* const src_suffix[3] = {"source.ply", "source.xyz", ".gird"};
    static const char* const tgt_suffix[3] = {"target.ply", "target.xyz", ".wlop"};
    std::vector<Row> rows;
    for (const std::string& name : names) {
        const auto t1 = std::chrono::steady_clock::now();
        const std::string fs = first_existing(in_dir, name, src_suffix), ft = first_existing(in_dir, name, tgt_suffix);
        if (fs.empty() || ft.empty()) {
            std::cout << name << ": no source/target file in " << in_dir << std::endl;
            continue;
        }
        const Cloud source = kss_cli::read_cloud(fs), target = kss_cli::read_cloud(ft);
        std::cout << "load ply finished." << std::endl;
        if (source.empty() || target.empty()) {
            std::cout << name << ": empty point cloud" << std::endl;
            continue;
        }
        std::cout << "registration runing." << std::endl;   // (sic)
        KSSICP ki;
        ki.KSSICP_init(source, target, 8);
        ki.KSSICP_Registration(1000);
        const Cloud aligned = ki.pointAlign;
        const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();   // the reference stops its clock here
        std::cout << "registration finished." << std::endl << "Measurement:" << std::endl;
        kss_cli::append_xyz(aligned, out_dir + "/" + name + "Align.xyz");
        kss_cli::append_xyz(target, out_dir + "/" + name + "Target.xyz");
        PCR_QM pq;
        pq.PCR_QM_init(aligned, target);
        const std::vector<double> m = pq.PCR_QM_ReturnResult();
        std::cout << "Registration Measure" << ":" << "MSE: " << m[0] << " RMSE: " << m[1] << " MAE: " << m[2] << std::endl;
        rows.push_back({name, seconds, m});
    }
    for (const Row& r : rows) std::cout << r.name << ":" << "time: " << r.seconds << std::endl;
    for (const Row& r : rows)
        std::cout << r.name << ":" << "MSE: " << r.measure[0] << " RMSE: " << r.measure[1] << " MAE: " << r.measure[2] << std::endl;
    return rows.size() == names.size() ? 0 : 1;
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 4) {
        std::cout << "usage: " << argv[0] << " <input dir> <output dir> <name> [<name> ...]" << std::endl;
        return 2;
    }
    try {
        return run(argv[1], argv[2], std::vector<std::string>(argv + 3, argv + argc));
    } catch (const std::exception& e) {
        std::cout << "failed: " << e.what() << std::endl;   // no GPU / HIP error: there is no CPU fallback
        return 1;
    }
}
