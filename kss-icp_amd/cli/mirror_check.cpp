// mirror_check.cpp -- exercises the C++ mirror classes (include/*.hpp) end to end on the GPU and prints
// machine-readable lines that tests/test_gpu_cli.py compares with the oracle.
// usage: mirror_check src.xyzn tgt.xyzn   (files: first line N, then N lines "x y z")
#include <cstdio>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "KSS_ICP.hpp"
#include "Method_Octree.hpp"
#include "normalCompute.hpp"
#include "initRegistrationKSS.hpp"
#include "registrationMeasure.hpp"

static std::vector<std::vector<double>> load(const char* path) {
    std::ifstream fin(path);
    std::vector<std::vector<double>> out;
    size_t n = 0;
    fin >> n;
    for (size_t i = 0; i < n; ++i) {
        double x, y, z;
        fin >> x >> y >> z;
        out.push_back({x, y, z});
    }
    return out;
}

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    auto S = load(argv[1]), T = load(argv[2]);
    std::printf("LOADED %zu %zu\n", S.size(), T.size());
    try {
        initRegistration_KSS ir;
        ir.initRegistration_init(S, T, 6);
        std::printf("PRESHAPE %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", ir.x_middle_S, ir.y_middle_S, ir.z_middle_S, ir.x_middle,
                    ir.y_middle, ir.z_middle, ir.scale);
        std::printf("ANGLE %.17g %.17g %.17g NLIST %zu GRID %d\n", ir.angle[0], ir.angle[1], ir.angle[2], ir.angleList.size(), ir.gridSize());
        auto P = ir.initRegistration_Rotation(S);
        std::printf("POSE0 %.17g %.17g %.17g\n", P[0][0], P[0][1], P[0][2]);
        KSSICP ki;
        ki.KSSICP_init(S, T, 6);
        const double f = ki.shapeRegistration_ICP_Judge(1000, P, T);
        std::printf("JUDGE %.17g\n", f);
        const double f2 = ki.shapeRegistration_ICP(1000, P, T);
        std::printf("ICP2 %.17g ALIGN0 %.17g %.17g %.17g\n", f2, ki.pointAlign[0][0], ki.pointAlign[0][1], ki.pointAlign[0][2]);
        // the members no front-end path reaches: the full-resolution overload (KSS_ICP.hpp:133-183: ICP on the MEMBER
        // clouds, pointAlign = PCL's float output cloud), the registered-cloud variant (:276-321) and the two pose helpers
        // of initRegistration_KSS (:95-140)
        KSSICP k3;
        k3.KSSICP_init(P, T, 6);
        const double f3 = k3.shapeRegistration_ICP(1000);
        const size_t last = k3.pointAlign.size() - 1;
        std::printf("ICPFULL %.17g %zu %.17g %.17g %.17g %.17g %.17g %.17g\n", f3, k3.pointAlign.size(), k3.pointAlign[0][0], k3.pointAlign[0][1],
                    k3.pointAlign[0][2], k3.pointAlign[last][0], k3.pointAlign[last][1], k3.pointAlign[last][2]);
        auto V = ki.shapeRegistration_ICP_AngleListV(1000, 0.0, P, T);
        std::printf("ANGLV %zu %.17g %.17g %.17g %.17g %.17g %.17g\n", V.size(), V[0][0], V[0][1], V[0][2], V[7][0], V[7][1], V[7][2]);
        std::printf("ANGL %.17g\n", ki.shapeRegistration_ICP_AngleList(1000, 0.0, P, T));
        auto RA = ir.initRegistration_Rotation_Angle(S, std::vector<double>{0.7875, 5.5125, 3.15});
        std::printf("ROTANG %.17g %.17g %.17g %.17g %.17g %.17g\n", RA[0][0], RA[0][1], RA[0][2], RA[11][0], RA[11][1], RA[11][2]);
        auto RX = ir.initRegistration_Rotation_Axis(S, 2, 0.6);
        std::printf("ROTAXIS %.17g %.17g %.17g %.17g %.17g %.17g\n", RX[0][0], RX[0][1], RX[0][2], RX[11][0], RX[11][1], RX[11][2]);
        auto RB = ir.initRegistration_Rotation_Axis(S, 7, 0.6);   // illegal axis: the shifted cloud comes back (:131-133)
        std::printf("ROTAXISBAD %.17g %.17g %.17g\n", RB[0][0], RB[0][1], RB[0][2]);
        PCR_QM pq;
        pq.PCR_QM_init(P, T);
        auto m = pq.PCR_QM_ReturnResult();
        std::printf("QM %.17g %.17g %.17g\n", m[0], m[1], m[2]);
        KSSICP k2;
        k2.KSSICP_init(S, T, 6);
        k2.KSSICP_Registration(1000);
        std::printf("REG scale %.17g fitness %.17g n %zu\n", k2.lastRegistration.scale, k2.lastRegistration.final_fitness, k2.pointAlign.size());
        NormalEstimation ne;
        ne.estimateNormal_init(std::string(argv[2]) + ".normals.tmp");
        std::remove(ne.fileNormal.c_str());
        auto N0 = ne.estimateNormal_PCL_MP_return(T);
        ne.estimateNormal_PCL_MP(T);
        NormalEstimation ne2;
        ne2.estimateNormal_init(ne.fileNormal);
        const bool loaded = ne2.normalLoad();
        std::printf("NORMALS %zu %zu %d %zu first %.17g %.17g %.17g oriented %.17g %.17g %.17g\n", N0.size(), ne.normalVector.size(), loaded ? 1 : 0,
                    ne2.normalVector.size(), N0[0][0], N0[0][1], N0[0][2], ne.normalVector[5][0], ne.normalVector[5][1], ne.normalVector[5][2]);
        std::remove(ne.fileNormal.c_str());
        PCL_octree oc;
        auto D = oc.PCL_Octree_Simplification_WithOutNormal(T);
        auto DN = oc.PCL_Octree_Simplification(T, T);     // (normals stand-in: any per-point rows)
        std::printf("OCTREE %zu %.17g %zu %zu first %.17g %.17g %.17g\n", D.size(), oc.lastResolution, DN[0].size(), DN[1].size(), D[0][0], D[0][1], D[0][2]);
    } catch (const std::exception& e) {
        std::printf("FAILED %s\n", e.what());
        return 1;
    }
    return 0;
}
