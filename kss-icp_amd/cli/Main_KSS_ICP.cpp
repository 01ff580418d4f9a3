// Main_KSS_ICP.cpp -- command-line front-end with the reference's call sequence
// (PS_AIS_Simplification/Main_KSS_ICP.cpp:61-93): load two PLYs, KSSICP_init(S, T, 8),
// KSSICP_Registration(1000), PCR_QM, save .xyz.  The reference hard-codes E:// paths (:67-71); its shipped
// EXEs take `source.ply target.ply` (EXE/Readme.txt:8-15), which is what this takes, plus an optional
// output path.   usage: kss_icp source.ply target.ply [result.xyz]
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "KSS_ICP.hpp"
#include "PlyLoad.h"
#include "registrationMeasure.hpp"

using namespace std;

vector<vector<double>> Load_PLY(string FileName) {
    CPLYLoader plyLoader;
    vector<char> p(FileName.begin(), FileName.end());
    p.push_back('\0');
    plyLoader.LoadModel(p.data());
    return plyLoader.points;
}

void save_PointCloud(vector<vector<double>> pointCloud, string Path) {   // ios::app, as the reference (:49-59)
    ofstream fout(Path, ios::app);
    fout << pointCloud.size() << endl;
    for (size_t i = 0; i < pointCloud.size(); i++) fout << pointCloud[i][0] << " " << pointCloud[i][1] << " " << pointCloud[i][2] << endl;
    fout << endl;
    fout.close();
}

int main(int argc, char* argv[]) {
    if (argc < 3) {
        cout << "usage: " << argv[0] << " source.ply target.ply [result.xyz]" << endl;
        return 2;
    }
    std::cout << "start!" << endl;
    std::cout << "load ply:" << endl;
    string fileSource = argv[1], fileTarget = argv[2];
    string fileSaveSource = argc > 3 ? argv[3] : "Registration.xyz";
    vector<vector<double>> pointSource = Load_PLY(fileSource);
    vector<vector<double>> pointTarget = Load_PLY(fileTarget);
    vector<vector<double>> pointAlign;
    std::cout << "load ply finished." << endl;
    if (pointSource.empty() || pointTarget.empty()) {
        cout << "empty point cloud" << endl;
        return 1;
    }
    std::cout << "registration runing." << endl;
    try {
        KSSICP ki;
        ki.KSSICP_init(pointSource, pointTarget, 8);
        ki.KSSICP_Registration(1000);
        pointAlign = ki.pointAlign;
        std::cout << "registration finished." << endl;
        const kss_register_result& r = ki.lastRegistration;
        std::cout << "scale: " << r.scale << endl;
        std::cout << "R: " << r.R[0] << " " << r.R[1] << " " << r.R[2] << " " << r.R[3] << " " << r.R[4] << " " << r.R[5] << " "
                  << r.R[6] << " " << r.R[7] << " " << r.R[8] << endl;
        std::cout << "t: " << r.t[0] << " " << r.t[1] << " " << r.t[2] << endl;
        std::cout << "Measurement:" << endl;
        PCR_QM pq;
        pq.PCR_QM_init(pointAlign, pointTarget);
        vector<double> measure_i = pq.PCR_QM_ReturnResult();
        std::cout << "Registration Measure" << ":" << "MSE: " << measure_i[0] << " RMSE: " << measure_i[1] << " MAE: " << measure_i[2] << endl;
        save_PointCloud(pointAlign, fileSaveSource);
    } catch (const std::exception& e) {
        cout << "failed: " << e.what() << endl;
        return 1;
    }
    return 0;
}
