// micro-benchmark: 30 dependent tiny kernels on one stream, launched one by one vs replayed as a captured hipGraph
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_tiny(int *p, int r) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += r; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main() {
    int *d; CK(hipMalloc(&d, 4096)); CK(hipMemset(d, 0, 4096));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int K = 30, REP = 200;
    for (int w = 0; w < 3; w++) { for (int k = 0; k < K; k++) k_tiny<<<64, 256, 0, s>>>(d, k); CK(hipStreamSynchronize(s)); }
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < REP; r++) { for (int k = 0; k < K; k++) k_tiny<<<64, 256, 0, s>>>(d, k); CK(hipStreamSynchronize(s)); }
    double direct = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / REP;
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int k = 0; k < K; k++) { k_tiny<<<64, 256, 0, s>>>(d, k); if (k % 10 == 0) CK(hipMemsetAsync(d + 16, 0, 64, s)); }
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int w = 0; w < 3; w++) { CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s)); }
    t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < REP; r++) { CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s)); }
    double graph = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / REP;
    printf("30 tiny kernels + sync: direct %.1f us, graph replay %.1f us\n", direct, graph);
    return 0;
}
