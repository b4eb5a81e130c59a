// Where the HIP runtime's start-up time goes in a fresh process (run on the GPU box: build/hip_startup [k|m|p]).
//   k: first kernel launch before the first copy;  m: first pageable copy first;  p: first copy from pinned memory first
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
__global__ void touch(int* p) { p[threadIdx.x] = threadIdx.x; }
static std::chrono::steady_clock::time_point t_last;
static void lap(const char* what) {
    const auto now = std::chrono::steady_clock::now();
    std::printf("%-34s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
    t_last = now;
}
int main(int argc, char** argv) {
    const char mode = argc > 1 ? argv[1][0] : 'm';
    t_last = std::chrono::steady_clock::now();
    int n = 0;
    (void)hipGetDeviceCount(&n); lap("hipGetDeviceCount");
    (void)hipSetDevice(0); lap("hipSetDevice");
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0); lap("hipGetDeviceProperties");
    void* d = nullptr; (void)hipMalloc(&d, 64 << 20); lap("hipMalloc 64 MB");
    std::vector<char> host(4 << 20, 1); lap("host vector");
    void* pinned = nullptr;
    auto kernel = [&] { touch<<<1, 64>>>((int*)d); (void)hipDeviceSynchronize(); };
    auto copy = [&] { (void)hipMemcpy(d, host.data(), host.size(), hipMemcpyHostToDevice); };
    auto pcopy = [&] { (void)hipHostMalloc(&pinned, 4 << 20); lap("  hipHostMalloc 4 MB"); std::memset(pinned, 1, 4 << 20); (void)hipMemcpy(d, pinned, 4 << 20, hipMemcpyHostToDevice); };
    if (mode == 'k') { kernel(); lap("first kernel + sync"); copy(); lap("first pageable copy 4 MB"); }
    else if (mode == 'p') { pcopy(); lap("first pinned copy 4 MB"); kernel(); lap("first kernel + sync"); copy(); lap("first pageable copy 4 MB"); }
    else { copy(); lap("first pageable copy 4 MB"); kernel(); lap("first kernel + sync"); }
    copy(); lap("second pageable copy 4 MB");
    kernel(); lap("second kernel + sync");
    (void)hipMemset(d, 0, 4 << 20); lap("hipMemset 4 MB");
    hipStream_t st; (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking); lap("hipStreamCreate");
    return 0;
}
