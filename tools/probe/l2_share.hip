// Two halves of a 1024-thread workgroup stream the SAME rows (as a dX role and a dW role of one layer would): does the pair cost one pass over
// HBM or two?  Variants: free-running; a barrier per tile (lockstep); role 1 trailing role 0 by LAG tiles behind a per-tile barrier.
//   hipcc --offload-arch=gfx950 -O3 -w tools/probe/l2_share.hip -o tools/probe/l2_share.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
constexpr int U = 4;    // 512-thread blocks of float4 per tile and array: 32 KB + 32 KB per tile
template <int ROLES, int BAR, int LAG>
__global__ __launch_bounds__(512 * ROLES) void reader(const float4 *__restrict__ y, const float4 *__restrict__ g, long n4, float *out)
{
    const int role = threadIdx.x >> 9, t = threadIdx.x & 511;
    float s = 0.f;
    const long ntile = n4 / (512 * U);
    const long steps = (ntile - blockIdx.x + gridDim.x - 1) / gridDim.x;
    for (long j = 0; j < steps + (ROLES > 1 ? LAG : 0); ++j) {
        const long jj = j - (role ? LAG : 0);
        if (jj >= 0 && jj < steps) {
            const long base = (blockIdx.x + jj * gridDim.x) * (512 * U) + t;
            float4 a[U], c[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { a[u] = y[base + 512 * u]; c[u] = g[base + 512 * u]; }
#pragma unroll
            for (int u = 0; u < U; ++u) s += a[u].x + c[u].x + a[u].w + c[u].w;
        }
        if (BAR) __syncthreads();
    }
    if (s == 12345.678f) out[0] = s;
}
template <int ROLES, int BAR, int LAG> float run(const float4 *y, const float4 *g, long n4, float *out, int grid)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> ts;
    for (int rep = 0; rep < 9; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((reader<ROLES, BAR, LAG>), dim3(grid), dim3(512 * ROLES), 0, 0, y, g, n4, out);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    return ts[4] * 1e3f;
}
int main()
{
    const long n4 = 134l * 1000 * 1000 / 16 / (512 * U * 256) * (512 * U * 256);
    float4 *y, *g; float *out;
    hipMalloc(&y, n4 * 16); hipMalloc(&g, n4 * 16); hipMalloc(&out, 4);
    hipMemset(y, 0, n4 * 16); hipMemset(g, 0, n4 * 16);
    printf("2 x %.0f MB\n", n4 * 16 / 1e6);
    printf("one role, 256 workgroups                      %6.1f us\n", run<1, 0, 0>(y, g, n4, out, 256));
    printf("one role, 512 workgroups                      %6.1f us\n", run<1, 0, 0>(y, g, n4, out, 512));
    printf("two roles free-running                        %6.1f us\n", run<2, 0, 0>(y, g, n4, out, 256));
    printf("two roles, barrier per tile, lag 0            %6.1f us\n", run<2, 1, 0>(y, g, n4, out, 256));
    printf("two roles, barrier per tile, lag 1            %6.1f us\n", run<2, 1, 1>(y, g, n4, out, 256));
    printf("two roles, barrier per tile, lag 2            %6.1f us\n", run<2, 1, 2>(y, g, n4, out, 256));
    printf("two roles, barrier per tile, lag 4            %6.1f us\n", run<2, 1, 4>(y, g, n4, out, 256));
    printf("two roles, no barrier, lag 2 at the start     %6.1f us\n", run<2, 0, 2>(y, g, n4, out, 256));
    return 0;
}
