// bar_probe.hip -- can the HOST store straight into fine-grained DEVICE memory (large BAR), and how long does a kernel
// polling that word take to see it?  Compared with the same kernel polling a host-mapped word (PCIe read per poll).
// Diagnostic only (a system without a CPU-visible BAR makes the direct store fault: run it as its own process).
//   hipcc --offload-arch=gfx950 -O2 tools/bar_probe.hip -o tools/bar_probe && tools/bar_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

__global__ void wait_word(const volatile unsigned* word, unsigned want, unsigned long long* out_host) {
    unsigned n = 0;
    while (__hip_atomic_load((const unsigned*)word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != want) {
        __builtin_amdgcn_s_sleep(1);
        if (++n > (1u << 24)) break;
    }
    // tell the host (host-mapped word) that we saw it
    __hip_atomic_store(out_host, (unsigned long long)want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

static double run(unsigned* word_dev, volatile unsigned* word_host, unsigned long long* ack, unsigned long long* ack_dev, int reps) {
    double tot = 0;
    for (int r = 1; r <= reps; ++r) {
        hipLaunchKernelGGL(wait_word, dim3(1), dim3(64), 0, 0, (const volatile unsigned*)word_dev, (unsigned)r, ack_dev);
        // let the kernel start polling
        const auto t_spin = std::chrono::steady_clock::now();
        while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_spin).count() < 200.0) {}
        const auto t0 = std::chrono::steady_clock::now();
        *word_host = (unsigned)r;
        __atomic_thread_fence(__ATOMIC_SEQ_CST);
        while (__atomic_load_n(ack, __ATOMIC_ACQUIRE) != (unsigned long long)r) {}
        tot += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        hipDeviceSynchronize();
    }
    return tot / reps;
}

int main() {
    unsigned long long *ack = nullptr, *ack_dev = nullptr;
    hipHostMalloc((void**)&ack, 64, hipHostMallocMapped);
    hipHostGetDevicePointer((void**)&ack_dev, ack, 0);
    *ack = 0;
    // (1) host-mapped word: the kernel polls over PCIe
    unsigned *hw = nullptr, *hw_dev = nullptr;
    hipHostMalloc((void**)&hw, 64, hipHostMallocMapped);
    hipHostGetDevicePointer((void**)&hw_dev, hw, 0);
    *hw = 0;
    std::printf("host-mapped word, kernel polls over PCIe: host store -> kernel saw it -> host saw the ack: %.2f us\n", run(hw_dev, hw, ack, ack_dev, 200));
    // (2) fine-grained device memory written by the host through the BAR
    unsigned* dw = nullptr;
    if (hipExtMallocWithFlags((void**)&dw, 4096, hipDeviceMallocFinegrained) != hipSuccess) { std::printf("hipExtMallocWithFlags(fine-grained) failed\n"); return 0; }
    hipMemset(dw, 0, 4096);
    hipDeviceSynchronize();
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, dw) == hipSuccess) std::printf("fine-grained device allocation: type %d, hostPointer %p, devicePointer %p\n", (int)at.type, at.hostPointer, at.devicePointer);
    std::fflush(stdout);
    *ack = 0;
    std::printf("fine-grained device word written by the host through the BAR: %.2f us\n", run(dw, (volatile unsigned*)dw, ack, ack_dev, 200));
    return 0;
}
