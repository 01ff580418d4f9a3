// waitvalue_probe.hip -- how long does a host -> GPU -> host hand-off take on this machine?
//   (A) launch:     host calls hipLaunchKernel for a one-wave kernel that stores a sequence number into host-mapped
//                   memory; time from just before the launch call to the host seeing the number;
//   (B) pre-queued: hipStreamWaitValue64 + the same kernel are enqueued FIRST, then the host releases them by writing
//                   the awaited value; time from that host store to the host seeing the kernel's number.
// (B) - (A) is what a pre-enqueued ICP iteration could save per iteration.  Diagnostic only; build with
//   hipcc --offload-arch=gfx950 -O2 tools/waitvalue_probe.hip -o gpurun_out/waitvalue_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void publish(unsigned long long* out, const unsigned long long* in_host, unsigned long long seq) {
    unsigned long long v = seq;
    if (in_host) v += __hip_atomic_load(in_host, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) & 0;   // one PCIe read, as a real kernel would fetch its transform
    if (threadIdx.x == 0) __hip_atomic_store(out, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

static double us(std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
    return std::chrono::duration<double, std::micro>(b - a).count();
}

static bool spin(volatile unsigned long long* p, unsigned long long want) {
    for (long i = 0; i < 200000000; ++i) {
        if (*p == want) return true;
        __builtin_ia32_pause();
    }
    return false;
}

int main() {
    hipStream_t st;
    CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    unsigned long long *h_out = nullptr, *d_out = nullptr, *h_arg = nullptr, *d_arg = nullptr;
    CHK(hipHostMalloc((void**)&h_out, 64, hipHostMallocMapped));
    CHK(hipHostGetDevicePointer((void**)&d_out, h_out, 0));
    CHK(hipHostMalloc((void**)&h_arg, 64, hipHostMallocMapped));
    CHK(hipHostGetDevicePointer((void**)&d_arg, h_arg, 0));
    *h_out = 0; *h_arg = 0;
    int can = 0;
    hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
    std::printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    const int N = 300;
    std::vector<double> ta, tb, tc;
    unsigned long long seq = 0;
    for (int i = 0; i < 20; ++i) { hipLaunchKernelGGL(publish, dim3(1), dim3(64), 0, st, d_out, (const unsigned long long*)nullptr, ++seq); if (!spin(h_out, seq)) return 2; }
    for (int i = 0; i < N; ++i) {   // (A)
        const auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(publish, dim3(1), dim3(64), 0, st, d_out, (const unsigned long long*)nullptr, ++seq);
        if (!spin(h_out, seq)) { std::printf("A: timeout\n"); return 2; }
        ta.push_back(us(t0, std::chrono::steady_clock::now()));
    }
    if (can) {
        // signal memory for the wait; fall back to host-mapped memory if the allocation flag is refused
        unsigned long long* sig = nullptr;
        bool sig_is_host = false;
        if (getenv("PROBE_HOST_WORD") || hipExtMallocWithFlags((void**)&sig, 8, hipMallocSignalMemory) != hipSuccess) {
            (void)hipGetLastError();
            CHK(hipHostMalloc((void**)&sig, 64, hipHostMallocMapped));
            sig_is_host = true;
        }
        std::printf("wait word: %s\n", sig_is_host ? "host-mapped" : "hipMallocSignalMemory");
        unsigned long long gate = 0;
        // the gate word is written by the host: directly if host memory, else through a second stream's write-value op
        hipStream_t st2;
        CHK(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
        if (sig_is_host) *sig = 0; else { CHK(hipStreamWriteValue64(st2, sig, 0, 0)); CHK(hipStreamSynchronize(st2)); }
        for (int pass = 0; pass < 2; ++pass) {
            std::vector<double>& tt = pass == 0 ? tb : tc;
            for (int i = 0; i < N; ++i) {   // (B): pass 1 also makes the kernel read its argument from host memory
                ++gate; ++seq;
                hipError_t e = hipStreamWaitValue64(st, sig, gate, hipStreamWaitValueGte, 0xffffffffffffffffull);
                if (e != hipSuccess) { std::printf("hipStreamWaitValue64: %s\n", hipGetErrorString(e)); return 3; }
                hipLaunchKernelGGL(publish, dim3(1), dim3(64), 0, st, d_out, pass == 1 ? (const unsigned long long*)d_arg : nullptr, seq);
                // give the command processor time to reach the wait packet, as a running ICP kernel would
                const auto tw = std::chrono::steady_clock::now();
                while (us(tw, std::chrono::steady_clock::now()) < 30.0) {}
                const auto t0 = std::chrono::steady_clock::now();
                if (sig_is_host) __atomic_store_n(sig, gate, __ATOMIC_RELEASE);
                else if (hipStreamWriteValue64(st2, sig, gate, 0) != hipSuccess) { std::printf("write value failed\n"); return 3; }
                if (!spin(h_out, seq)) {
                    std::printf("B: timeout (releasing)\n");
                    if (sig_is_host) *sig = ~0ull;
                    return 2;
                }
                tt.push_back(us(t0, std::chrono::steady_clock::now()));
            }
        }
    }
    auto med = [](std::vector<double> v) { if (v.empty()) return -1.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    auto p10 = [](std::vector<double> v) { if (v.empty()) return -1.0; std::sort(v.begin(), v.end()); return v[v.size() / 10]; };
    std::printf("(A) launch -> visible:                 median %.2f us, p10 %.2f us\n", med(ta), p10(ta));
    std::printf("(B) gate store -> visible:             median %.2f us, p10 %.2f us\n", med(tb), p10(tb));
    std::printf("(B') same, kernel reads host argument: median %.2f us, p10 %.2f us\n", med(tc), p10(tc));
    hipStreamSynchronize(st);
    return 0;
}
