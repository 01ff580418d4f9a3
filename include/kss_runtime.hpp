// kss_runtime.hpp -- tiny RAII holder of the per-thread kss_ctx used by the C++ mirror classes
// (include/KSS_ICP.hpp, include/initRegistrationKSS.hpp, include/registrationMeasure.hpp).
// One context per host thread (the C-ABI's threading rule); device chosen by KSS_DEVICE (default 0).
#pragma once
#include <cstdlib>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "kssicp.h"

namespace kss_host {

typedef std::vector<std::vector<double>> vvd;

class Runtime {
public:
    static kss_ctx* ctx() {
        thread_local Runtime rt;
        return rt.c_;
    }
    // The reference reports problems on stdout and carries on; a missing GPU is not something the
    // port can "carry on" from (there is no CPU fallback), so it is a hard error.
    static void check(int rc, const char* where) {
        if (rc == KSS_OK) return;
        std::string msg = std::string(where) + ": " + kss_status_string(rc);
        const char* d = kss_last_error(ctx_or_null());
        if (d && *d) msg += std::string(" (") + d + ")";
        std::cout << "KSS-ICP error: " << msg << std::endl;
        throw std::runtime_error(msg);
    }

private:
    kss_ctx* c_ = nullptr;
    static kss_ctx*& slot() { thread_local kss_ctx* s = nullptr; return s; }
    static kss_ctx* ctx_or_null() { return slot(); }
    Runtime() {
        const char* dev = std::getenv("KSS_DEVICE");
        const int rc = kss_ctx_create(dev ? std::atoi(dev) : 0, &c_);
        if (rc != KSS_OK) {
            std::cout << "KSS-ICP error: cannot create GPU context: " << kss_status_string(rc) << std::endl;
            throw std::runtime_error(std::string("kss_ctx_create: ") + kss_status_string(rc));
        }
        slot() = c_;
    }
    ~Runtime() {
        if (c_) kss_ctx_destroy(c_);
        slot() = nullptr;
    }
};

// vector<vector<double>> <-> packed xyz
inline std::vector<double> pack(const vvd& p) {
    std::vector<double> out(p.size() * 3);
    for (size_t i = 0; i < p.size(); ++i) { out[3 * i] = p[i][0]; out[3 * i + 1] = p[i][1]; out[3 * i + 2] = p[i][2]; }
    return out;
}
inline std::vector<float> pack_f32(const vvd& p) {   // cloud_i.x = ps[i][0]: double -> float narrowing (PointXYZ)
    std::vector<float> out(p.size() * 3);
    for (size_t i = 0; i < p.size(); ++i) { out[3 * i] = (float)p[i][0]; out[3 * i + 1] = (float)p[i][1]; out[3 * i + 2] = (float)p[i][2]; }
    return out;
}
inline vvd unpack(const std::vector<double>& a) {
    vvd out(a.size() / 3, std::vector<double>(3));
    for (size_t i = 0; i < out.size(); ++i) { out[i][0] = a[3 * i]; out[i][1] = a[3 * i + 1]; out[i][2] = a[3 * i + 2]; }
    return out;
}
inline vvd unpack_f32(const std::vector<float>& a) {
    vvd out(a.size() / 3, std::vector<double>(3));
    for (size_t i = 0; i < out.size(); ++i) { out[i][0] = a[3 * i]; out[i][1] = a[3 * i + 1]; out[i][2] = a[3 * i + 2]; }
    return out;
}

}  // namespace kss_host
