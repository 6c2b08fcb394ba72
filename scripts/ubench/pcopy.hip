// Micro-benchmark (round 6, VERDICT r5 #1c): what a kernel that fetches a call's parameter blob from pinned host
// memory costs, by variant.  Each variant is launched 200 times back to back on one stream between two of its own
// dispatch time stamps (hipExtLaunchKernelGGL start / stop events on every launch): kernel duration as the queue sees it.
// Build: hipcc --offload-arch=gfx950 -O3 -o pcopy pcopy.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void k_empty() {}

// round-5 form: 256 threads, load -> store loop
__global__ void __launch_bounds__(256) k_loop(uint4* dst, const uint4* src, int n16, unsigned long long* flag, unsigned long long seq) {
    for (int i = threadIdx.x; i < n16; i += 256) dst[i] = src[i];
    if (flag != nullptr) {
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence_system();
            __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// MODE 0: all loads first, fence + release flag; 1: no flag at all; 2: relaxed flag store without the fence;
// 3: loads only into registers, stores, no flag (same as 1, 1024 threads); 4: flag only, no copy
template <int MODE>
__global__ void __launch_bounds__(1024) k_once(uint4* dst, const uint4* src, int n16, unsigned long long* flag, unsigned long long seq) {
    if (MODE != 4) {
        uint4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = q * 1024 + (int)threadIdx.x;
            if (i < n16) v[q] = src[i];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = q * 1024 + (int)threadIdx.x;
            if (i < n16) dst[i] = v[q];
        }
    }
    if (MODE == 0 || MODE == 4) {
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence_system();
            __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    } else if (MODE == 2) {
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// a consumer that reads its parameters straight from the pinned blob (what a fused head kernel would do):
// 100 workgroups, each reads 48 bytes of "its task" and writes one value
__global__ void __launch_bounds__(256) k_direct(double* out, const double* src) {
    const double a = src[(blockIdx.x % 1000) * 6 + (threadIdx.x % 6)];
    out[blockIdx.x * 256 + threadIdx.x] = a * 2.0;
}
__global__ void __launch_bounds__(256) k_direct_dev(double* out, const double* src) {
    const double a = src[(blockIdx.x % 1000) * 6 + (threadIdx.x % 6)];
    out[blockIdx.x * 256 + threadIdx.x] = a * 2.0;
}

int main() {
    const int bytes_list[3] = {7424, 20480, 65536};
    hipStream_t s;
    CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    void *h = nullptr, *d = nullptr, *hw = nullptr;
    unsigned long long* flag = nullptr;
    double* out = nullptr;
    CHK(hipHostMalloc(&h, 65536, hipHostMallocDefault));
    CHK(hipHostMalloc(&hw, 65536, hipHostMallocWriteCombined));
    CHK(hipHostMalloc((void**)&flag, 64, hipHostMallocDefault));
    CHK(hipMalloc(&d, 65536));
    CHK(hipMalloc((void**)&out, 100 * 256 * 8));
    const int NREP = 200;
    std::vector<hipEvent_t> ea(NREP), eb(NREP);
    for (int i = 0; i < NREP; ++i) { CHK(hipEventCreate(&ea[i])); CHK(hipEventCreate(&eb[i])); }
    auto report = [&](const char* name, int bytes) -> int {
        CHK(hipStreamSynchronize(s));
        std::vector<float> ms(NREP);
        for (int i = 0; i < NREP; ++i) CHK(hipEventElapsedTime(&ms[i], ea[i], eb[i]));
        std::sort(ms.begin(), ms.end());
        printf("%-44s bytes %6d  median %6.2f us  p10 %6.2f  p90 %6.2f\n", name, bytes, ms[NREP / 2] * 1e3, ms[NREP / 10] * 1e3, ms[NREP * 9 / 10] * 1e3);
        return 0;
    };
    for (int i = 0; i < NREP; ++i) hipExtLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, ea[i], eb[i], 0);
    if (report("empty kernel (1 wave)", 0)) return 1;
    for (int b = 0; b < 3; ++b) {
        const int bytes = bytes_list[b], n16 = bytes / 16;
        for (int i = 0; i < NREP; ++i) hipExtLaunchKernelGGL(k_loop, dim3(1), dim3(256), 0, s, ea[i], eb[i], 0, (uint4*)d, (const uint4*)h, n16, flag, (unsigned long long)i);
        if (report("r5: 256 threads, load->store loop, fence+flag", bytes)) return 1;
        if (n16 <= 4096) {
            for (int i = 0; i < NREP; ++i) hipExtLaunchKernelGGL(k_once<0>, dim3(1), dim3(1024), 0, s, ea[i], eb[i], 0, (uint4*)d, (const uint4*)h, n16, flag, (unsigned long long)i);
            if (report("r6: 1024 threads, loads first, fence+flag", bytes)) return 1;
            for (int i = 0; i < NREP; ++i) hipExtLaunchKernelGGL(k_once<1>, dim3(1), dim3(1024), 0, s, ea[i], eb[i], 0, (uint4*)d, (const uint4*)h, n16, flag, (unsigned long long)i);
            if (report("    the same, no flag", bytes)) return 1;
            for (int i = 0; i < NREP; ++i) hipExtLaunchKernelGGL(k_once<2>, dim3(1), dim3(1024), 0, s, ea[i], eb[i], 0, (uint4*)d, (const uint4*)h, n16, flag, (unsigned long long)i);
            if (report("    the same, relaxed flag, no fence", bytes)) return 1;
            for (int i = 0; i < NREP; ++i) hipExtLaunchKernelGGL(k_once<0>, dim3(1), dim3(1024), 0, s, ea[i], eb[i], 0, (uint4*)d, (const uint4*)hw, n16, flag, (unsigned long long)i);
            if (report("    r6 form from write-combined host memory", bytes)) return 1;
        }
    }
    for (int i = 0; i < NREP; ++i) hipExtLaunchKernelGGL(k_once<4>, dim3(1), dim3(64), 0, s, ea[i], eb[i], 0, (uint4*)d, (const uint4*)h, 0, flag, (unsigned long long)i);
    if (report("flag only (fence + release store), no copy", 0)) return 1;
    for (int i = 0; i < NREP; ++i) hipExtLaunchKernelGGL(k_direct, dim3(100), dim3(256), 0, s, ea[i], eb[i], 0, out, (const double*)h);
    if (report("100 workgroups reading 48 B each from HOST", 4800)) return 1;
    {   // how many PCIe reads fit beside a launch: W workgroups whose four waves each read one 48-byte record
        double* big = nullptr;
        CHK(hipMalloc((void**)&big, 4096 * 256 * 8));
        for (int W : {100, 250, 500, 1000, 2500}) {
            for (int i = 0; i < NREP; ++i) hipExtLaunchKernelGGL(k_direct, dim3(W), dim3(256), 0, s, ea[i], eb[i], 0, big, (const double*)h);
            char nm[64];
            snprintf(nm, sizeof nm, "%d workgroups x 4 waves reading from HOST", W);
            if (report(nm, W * 48)) return 1;
            for (int i = 0; i < NREP; ++i) hipExtLaunchKernelGGL(k_direct_dev, dim3(W), dim3(256), 0, s, ea[i], eb[i], 0, big, (const double*)d);
            snprintf(nm, sizeof nm, "%d workgroups x 4 waves reading from DEVICE", W);
            if (report(nm, W * 48)) return 1;
        }
    }
    for (int i = 0; i < NREP; ++i) hipExtLaunchKernelGGL(k_direct_dev, dim3(100), dim3(256), 0, s, ea[i], eb[i], 0, out, (const double*)d);
    if (report("100 workgroups reading 48 B each from DEVICE", 4800)) return 1;
    // hipMemcpyAsync of the same sizes, for reference: time of 200 copies back to back / 200
    for (int b = 0; b < 3; ++b) {
        hipEvent_t a0, a1;
        CHK(hipEventCreate(&a0)); CHK(hipEventCreate(&a1));
        CHK(hipEventRecord(a0, s));
        for (int i = 0; i < NREP; ++i) CHK(hipMemcpyAsync(d, h, bytes_list[b], hipMemcpyHostToDevice, s));
        CHK(hipEventRecord(a1, s));
        CHK(hipStreamSynchronize(s));
        float ms = 0;
        CHK(hipEventElapsedTime(&ms, a0, a1));
        printf("%-44s bytes %6d  mean   %6.2f us (back to back)\n", "hipMemcpyAsync H2D", bytes_list[b], ms * 1e3 / NREP);
    }
    return 0;
}
