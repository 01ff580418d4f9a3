// tools/tree_check.hip -- the canonical wave trees of kss_device.hpp against a plain __shfl_xor restatement of the same tree
// (levels 32, 16, 8, 4, 2, 1), bit for bit, on random inputs.  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I kss-icp_amd/csrc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "kss_device.hpp"
using namespace kss;
__global__ void k(const double* in, double* out16, double* out2, double* ref16, double* ref2) {
    const int l = threadIdx.x;
    double v[16], r[16];
    for (int j = 0; j < 16; ++j) { v[j] = in[j * 64 + l]; r[j] = v[j]; }
    wave_tree16(v);
    for (int j = 0; j < 4; ++j) out16[j * 64 + l] = v[j];
    for (int j = 0; j < 16; ++j) for (int m = 32; m > 0; m >>= 1) r[j] += __shfl_xor(r[j], m, 64);
    for (int j = 0; j < 16; ++j) ref16[j * 64 + l] = r[j];
    double a = in[16 * 64 + l], b = in[17 * 64 + l];
    out2[l] = wave_tree2(a, b);
    for (int m = 32; m > 0; m >>= 1) { a += __shfl_xor(a, m, 64); b += __shfl_xor(b, m, 64); }
    ref2[l] = a; ref2[64 + l] = b;
}
int main() {
    double h[18 * 64];
    srand(7);
    for (auto& x : h) x = (double)rand() / RAND_MAX * ((rand() & 1) ? 1.0 : 1e-3);
    double *d, *o16, *o2, *r16, *r2;
    hipMalloc(&d, sizeof h); hipMalloc(&o16, 4 * 64 * 8); hipMalloc(&o2, 64 * 8); hipMalloc(&r16, 16 * 64 * 8); hipMalloc(&r2, 128 * 8);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o16, o2, r16, r2);
    double a16[4 * 64], a2[64], b16[16 * 64], b2[128];
    hipMemcpy(a16, o16, sizeof a16, hipMemcpyDeviceToHost); hipMemcpy(a2, o2, sizeof a2, hipMemcpyDeviceToHost);
    hipMemcpy(b16, r16, sizeof b16, hipMemcpyDeviceToHost); hipMemcpy(b2, r2, sizeof b2, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int q = 0; q < 4; ++q) for (int j = 0; j < 4; ++j) if (memcmp(&a16[j * 64 + 16 * q], &b16[(4 * q + j) * 64], 8)) { ++bad; printf("tree16 col %d: %.17g vs %.17g\n", 4 * q + j, a16[j * 64 + 16 * q], b16[(4 * q + j) * 64]); }
    if (memcmp(&a2[0], &b2[0], 8)) { ++bad; printf("tree2 a: %.17g vs %.17g\n", a2[0], b2[0]); }
    if (memcmp(&a2[32], &b2[64], 8)) { ++bad; printf("tree2 b: %.17g vs %.17g\n", a2[32], b2[64]); }
    printf("tree check: %d mismatches\n", bad);
    return bad != 0;
}
