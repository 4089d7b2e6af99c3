// Probe: HBM read bandwidth of the GEMM A-operand access pattern.  A [M][C] fp32 row-major; a workgroup owns 128-row tiles and
// walks the row in k-chunks of KC floats (one float4 per thread, 256 threads): pattern "tile" = what mlp_gemm does;
// pattern "blocked" = same bytes from a [M/128][C/KC][128][KC] layout (each stage one contiguous block).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int KC, bool BLOCKED>
__global__ __launch_bounds__(256) void rd(const float *a, int64_t M, int C, float *out)
{
    const int tid = threadIdx.x;
    constexpr int TPR = KC / 4;           // threads per row piece
    constexpr int RPI = 256 / TPR;        // rows per pass
    const int64_t ntiles = M / 128;
    float4 acc = make_float4(0, 0, 0, 0);
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        for (int kc = 0; kc < C / KC; ++kc) {
#pragma unroll
            for (int i = 0; i < 128 / RPI; ++i) {
                const int r = tid / TPR + RPI * i, q = (tid % TPR) * 4;
                const float *p = BLOCKED ? a + ((t * (C / KC) + kc) * 128 + r) * KC + q : a + (t * 128 + r) * C + kc * KC + q;
                const float4 v = *reinterpret_cast<const float4 *>(p);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = 1.f;
}
template <int KC, bool BLOCKED>
static void run(const float *a, int64_t M, int C, float *out, const char *name, int grid)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((rd<KC, BLOCKED>), dim3(grid), dim3(256), 0, 0, a, M, C, out);
    hipEventRecord(e0);
    const int n = 10;
    for (int w = 0; w < n; ++w) hipLaunchKernelGGL((rd<KC, BLOCKED>), dim3(grid), dim3(256), 0, 0, a, M, C, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s C=%4d grid=%4d  %.1f us  %.0f GB/s\n", name, C, grid, ms / n * 1e3, (double)M * C * 4 / (ms / n * 1e-3) / 1e9);
}
int main()
{
    const int64_t M = 262144;
    float *a, *out; hipMalloc(&a, M * 1024 * 4); hipMalloc(&out, 4); hipMemset(a, 0, M * 1024 * 4);
    for (int C : {128, 256}) for (int grid : {512, 2048}) {
        run<16, false>(a, M, C, out, "row-major, 64 B pieces", grid);
        run<32, false>(a, M, C, out, "row-major, 128 B pieces", grid);
        run<64, false>(a, M, C, out, "row-major, 256 B pieces", grid);
        run<16, true>(a, M, C, out, "blocked [128][16]", grid);
        run<32, true>(a, M, C, out, "blocked [128][32]", grid);
    }
    return 0;
}
