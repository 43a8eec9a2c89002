// Microbenchmark behind DESIGN.md's note on k_sapx: how fast can K workgroups of 256 threads read
// the SAME sequence of rows (each workgroup its own 4 KiB segment of a 64 KiB row) when the rows
// are scattered over a 1 GiB matrix, against consecutive rows?   hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
template <int G>
__global__ __launch_bounds__(256) void k_rows(const uint4 *m, const int *rows, int nrows, size_t pitch16, unsigned *sink)
{
    const size_t seg = (size_t)blockIdx.x * 256 + threadIdx.x;
    unsigned acc = 0;
    for (int r0 = 0; r0 < nrows; r0 += G) {
        uint4 v[G];
#pragma unroll
        for (int g = 0; g < G; g++) v[g] = m[(size_t)rows[r0 + g] * pitch16 + seg];
#pragma unroll
        for (int g = 0; g < G; g++) acc ^= v[g].x + v[g].y + v[g].z + v[g].w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
int main()
{
    const int n = 16384;
    const size_t pitch16 = n * 4 / 16;   // uint4 per row
    uint4 *m;
    hipMalloc(&m, (size_t)n * n * 4);
    hipMemset(m, 1, (size_t)n * n * 4);
    const int R = 8192;
    std::vector<int> seq(R), rnd(R);
    srand(1);
    for (int i = 0; i < R; i++) {
        seq[i] = i;
        rnd[i] = rand() % n;
    }
    int *d_rows;
    unsigned *sink;
    hipMalloc(&d_rows, R * 4);
    hipMalloc(&sink, 4);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int mode = 0; mode < 2; mode++) {
        hipMemcpy(d_rows, mode ? rnd.data() : seq.data(), R * 4, hipMemcpyHostToDevice);
        for (int K : {1, 4, 16}) {
            for (int G : {8, 32}) {
                float best = 1e9;
                for (int rep = 0; rep < 3; rep++) {
                    hipEventRecord(a);
                    if (G == 8) k_rows<8><<<K, 256>>>(m, d_rows, R, pitch16, sink);
                    else k_rows<32><<<K, 256>>>(m, d_rows, R, pitch16, sink);
                    hipEventRecord(b);
                    hipEventSynchronize(b);
                    float ms;
                    hipEventElapsedTime(&ms, a, b);
                    best = ms < best ? ms : best;
                }
                printf("%s rows, K=%2d workgroups, %2d rows in flight: %7.3f us per row, %6.1f GB/s per workgroup\n",
                       mode ? "scattered  " : "consecutive", K, G, 1e3 * best / R, 4096.0 * R / (best * 1e-3) / 1e9);
            }
        }
    }
    return 0;
}
