// Does the 256 MiB Infinity Cache serve a consumer that re-reads 268 MB a producer has just swept, and does the consumer's order matter?
// producer: reads y (134 MB), writes g (134 MB), all workgroups sweeping front to back together (the row-streaming kernels' tile order).
// consumer: reads y and g -- (a) same sweep front to back, (b) the sweep back to front, (c) one contiguous chunk per workgroup (dw_rows_kernel).
//   hipcc --offload-arch=gfx950 -O3 tools/probe/mall_order.hip -o /tmp/mall_order && /tmp/mall_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int T = 512;
__global__ __launch_bounds__(T) void producer(const float4 *__restrict__ y, float4 *__restrict__ g, long n4)
{
    for (long i = (long)blockIdx.x * T + threadIdx.x; i < n4; i += (long)gridDim.x * T) {
        float4 v = y[i]; v.x += 1.f; g[i] = v;
    }
}
template <int MODE>
__global__ __launch_bounds__(T) void consumer(const float4 *__restrict__ y, const float4 *__restrict__ g, long n4, float *out)
{
    float s = 0.f;
    if (MODE == 2) {
        const long per = (n4 + gridDim.x - 1) / gridDim.x, b = blockIdx.x * per, e = b + per < n4 ? b + per : n4;
        for (long i = b + threadIdx.x; i < e; i += T) { const float4 a = y[i], c = g[i]; s += a.x + c.x + a.w + c.w; }
    } else {
        const long nblk = n4 / T;
        for (long q = blockIdx.x; q < nblk; q += gridDim.x) {
            const long blk = MODE == 1 ? nblk - 1 - q : q;
            const long i = blk * T + threadIdx.x;
            const float4 a = y[i], c = g[i]; s += a.x + c.x + a.w + c.w;
        }
    }
    if (s == 12345.678f) out[0] = s;
}
int main()
{
    for (long mb : {134l, 100l, 64l}) {
        const long n4 = mb * 1000 * 1000 / 16 / T * T;
        float4 *y, *g; float *out;
        CK(hipMalloc(&y, n4 * 16)); CK(hipMalloc(&g, n4 * 16)); CK(hipMalloc(&out, 4));
        CK(hipMemset(y, 0, n4 * 16));
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const char *names[3] = {"same sweep, front to back", "sweep back to front", "one contiguous chunk per workgroup"};
        for (int mode = 0; mode < 3; ++mode) {
            std::vector<float> ts;
            for (int rep = 0; rep < 9; ++rep) {
                hipLaunchKernelGGL(producer, dim3(512), dim3(T), 0, 0, y, g, n4);
                hipEventRecord(e0, 0);
                if (mode == 0) hipLaunchKernelGGL(consumer<0>, dim3(512), dim3(T), 0, 0, y, g, n4, out);
                if (mode == 1) hipLaunchKernelGGL(consumer<1>, dim3(512), dim3(T), 0, 0, y, g, n4, out);
                if (mode == 2) hipLaunchKernelGGL(consumer<2>, dim3(512), dim3(T), 0, 0, y, g, n4, out);
                hipEventRecord(e1, 0);
                CK(hipEventSynchronize(e1));
                float ms; hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms);
            }
            std::sort(ts.begin(), ts.end());
            printf("2 x %ld MB  %-36s %7.1f us  %5.2f TB/s\n", mb, names[mode], ts[4] * 1e3, 2.0 * n4 * 16 / (ts[4] * 1e-3) / 1e12);
        }
        // back-to-back consumers alternating direction (what dW then dX would do)
        {
            std::vector<float> ts;
            for (int rep = 0; rep < 9; ++rep) {
                hipLaunchKernelGGL(producer, dim3(512), dim3(T), 0, 0, y, g, n4);
                hipLaunchKernelGGL(consumer<1>, dim3(512), dim3(T), 0, 0, y, g, n4, out);
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL(consumer<0>, dim3(512), dim3(T), 0, 0, y, g, n4, out);
                hipEventRecord(e1, 0);
                CK(hipEventSynchronize(e1));
                float ms; hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms);
            }
            std::sort(ts.begin(), ts.end());
            printf("2 x %ld MB  %-36s %7.1f us  %5.2f TB/s\n", mb, "front to back AFTER a back-to-front", ts[4] * 1e3, 2.0 * n4 * 16 / (ts[4] * 1e-3) / 1e12);
        }
        hipFree(y); hipFree(g); hipFree(out);
    }
    return 0;
}
