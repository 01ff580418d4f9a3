// include/Method_Octree.hpp -- MI355X mirror of the reference class `PCL_octree`
// (PS_AIS_Simplification/Method_Octree.hpp:12-165): octree down-sampling of a point cloud, with or without normals.
// The PCL octree + kNN radius estimate behind it run on the device (kss_downsample_octree); the class keeps the
// reference's method names, by-value arguments, result nesting and progress lines.
#pragma once
#include <chrono>
#include <iostream>
#include <vector>

#include "kss_runtime.hpp"

class PCL_octree {
public:
    // result[0] = down-sampled points, result[1] = their normals (:19-75)
    std::vector<std::vector<std::vector<double>>> PCL_Octree_Simplification(std::vector<std::vector<double>> pData,
                                                                            std::vector<std::vector<double>> nData) {
        std::cout << "PCL down-sampling start:" << std::endl;
        const auto t1 = std::chrono::steady_clock::now();
        const std::vector<int32_t> sel = select(pData);
        std::vector<std::vector<double>> resultP, resultN;
        resultP.reserve(sel.size());
        resultN.reserve(sel.size());
        for (int32_t s : sel) {
            resultP.push_back(pData[(size_t)s]);
            resultN.push_back(nData[(size_t)s]);
        }
        const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
        std::cout << "Octree down-sampling:" << seconds << "s" << std::endl;
        std::cout << "Original Point:" << pData.size() << std::endl;
        std::cout << "Down-sampling Point:" << resultP.size() << std::endl;
        std::vector<std::vector<std::vector<double>>> result;
        result.push_back(resultP);
        result.push_back(resultN);
        return result;
    }

    // (:77-104)
    std::vector<std::vector<double>> PCL_Octree_Simplification_WithOutNormal(std::vector<std::vector<double>> pData) {
        const std::vector<int32_t> sel = select(pData);
        std::vector<std::vector<double>> resultP;
        resultP.reserve(sel.size());
        for (int32_t s : sel) resultP.push_back(pData[(size_t)s]);
        return resultP;
    }

    double lastResolution = 0.0;   // the resolution PCL_Octree_Resolution chose (:151-165); not exposed by the reference

private:
    std::vector<int32_t> select(const std::vector<std::vector<double>>& pData) {
        const std::vector<double> p = kss_host::pack(pData);
        std::vector<int32_t> idx(pData.size());
        int64_t m = 0;
        kss_host::Runtime::check(kss_downsample_octree(kss_host::Runtime::ctx(), p.data(), (int64_t)pData.size(), idx.data(),
                                                       (int64_t)idx.size(), &m, &lastResolution), "kss_downsample_octree");
        idx.resize((size_t)m);
        return idx;
    }
};
