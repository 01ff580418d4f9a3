// include/normalCompute.hpp -- MI355X mirror of the point-cloud half of the reference class `NormalEstimation`
// (PS_AIS_Simplification/normalCompute.hpp:17-742): PCL-style normals (k = 20, view point at the origin) and their
// consistent re-orientation.  The mesh-based estimators (:34-306, GLMmodel) belong to the OpenGL viewer and are not
// mirrored.  Same method names, by-value arguments, public members and normal-file format (count, then rows;
// append mode) as the reference; the k-NN / covariance / eigenvector work runs on the device (kss_normals,
// kss_normals_orient).
#pragma once
#include <cstdio>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "kss_runtime.hpp"

class NormalEstimation {
public:
    std::vector<std::vector<double>> normalVector;
    std::string fileNormal;

    void estimateNormal_init(std::string fileNormal_input) { fileNormal = fileNormal_input; }

    // :308-355 -- pcl::NormalEstimationOMP (k = 20), renormalised in double; no re-orientation, nothing stored
    std::vector<std::vector<double>> estimateNormal_PCL_MP_return(std::vector<std::vector<double>> pointsVector) {
        return compute(pointsVector);
    }

    // :358-403 -- the same normals, re-oriented (estimateNormal_RegularNormal), kept in normalVector and appended to
    // the normal file
    void estimateNormal_PCL_MP(std::vector<std::vector<double>> pointsVector) {
        normalVector = compute(pointsVector);
        std::vector<std::vector<double>> nR = estimateNormal_RegularNormal(pointsVector, normalVector);
        normalVector.clear();
        normalVector = nR;
        normalSave(normalVector);
    }

    // :405-437 (the reference re-opens stdin on the file; a stream does the same job)
    bool normalLoad() {
        std::ifstream fin(fileNormal);
        if (!fin) return false;
        if (normalVector.size() > 0) normalVector.clear();
        int numSum = 0;
        fin >> numSum;
        for (int i = 0; i < numSum; i++) {
            double x_i, y_i, z_i;
            if (!(fin >> x_i >> y_i >> z_i)) break;
            normalVector.push_back({x_i, y_i, z_i});
        }
        return true;
    }

    // :614-742
    std::vector<std::vector<double>> estimateNormal_RegularNormal(std::vector<std::vector<double>> pointCloudData,
                                                                  std::vector<std::vector<double>> pointNormal) {
        std::cout << "RegularNormal start:" << std::endl;
        std::cout << "Init kdtree" << std::endl;
        const std::vector<double> p = kss_host::pack(pointCloudData);
        std::vector<double> nrm = kss_host::pack(pointNormal);
        kss_host::Runtime::check(kss_normals_orient(kss_host::Runtime::ctx(), p.data(), (int64_t)pointCloudData.size(), nrm.data()), "kss_normals_orient");
        std::cout << std::endl;
        std::cout << "RegularNormal end:" << std::endl;
        return kss_host::unpack(nrm);
    }

private:
    std::vector<std::vector<double>> compute(const std::vector<std::vector<double>>& pointsVector) {
        const std::vector<double> p = kss_host::pack(pointsVector);
        std::vector<double> nrm(p.size());
        kss_host::Runtime::check(kss_normals(kss_host::Runtime::ctx(), p.data(), (int64_t)pointsVector.size(), 20, nrm.data()), "kss_normals");
        return kss_host::unpack(nrm);
    }

    // :597-612
    void normalSave(std::vector<std::vector<double>> n) {
        if (fileNormal.size() == 2) {
            std::cout << "normal file name is empty!" << std::endl;
        } else {
            std::ofstream fout(fileNormal, std::ios::app);
            fout << n.size() << std::endl;
            for (size_t i = 0; i < n.size(); i++) fout << n[i][0] << " " << n[i][1] << " " << n[i][2] << std::endl;
            fout << std::endl;
            fout.close();
        }
    }
};
