// include/registrationMeasure.hpp -- MI355X mirror of the reference class `PCR_QM`
// (PS_AIS_Simplification/registrationMeasure.hpp:21-99): MSE / RMSE / MAE of aligned -> template nearest
// neighbour distances.  One device NN sweep + f64 reduction (kss_pcr_qm) replaces the FLANN K=1 loop (:63-83).
#pragma once
#include <iostream>
#include <vector>

#include "kss_runtime.hpp"

class PCR_QM {
private:
    std::vector<std::vector<double>> a;
    std::vector<std::vector<double>> t;
    std::vector<double> MSERA;

public:
    void PCR_QM_init(std::vector<std::vector<double>> alignV, std::vector<std::vector<double>> templateV) {
        a = alignV;
        t = templateV;
        PCR_QM_Start();
    }

    std::vector<double> PCR_QM_ReturnResult() { return MSERA; }

private:
    void PCR_QM_Start() {
        std::vector<double> pa = kss_host::pack(a), pt = kss_host::pack(t);
        double out[3] = {0, 0, 0};
        kss_host::Runtime::check(kss_pcr_qm(kss_host::Runtime::ctx(), pa.data(), (int64_t)a.size(), pt.data(), (int64_t)t.size(), out), "kss_pcr_qm");
        std::cout << "Result:" << std::endl;
        std::cout << "MSE:  " << out[0] << std::endl;
        std::cout << "RMSE: " << out[1] << std::endl;
        std::cout << "MAE:  " << out[2] << std::endl;
        MSERA.clear();
        MSERA.push_back(out[0]);
        MSERA.push_back(out[1]);
        MSERA.push_back(out[2]);
    }
};
