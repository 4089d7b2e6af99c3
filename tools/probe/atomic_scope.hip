// float atomic adds of 256-byte rows into a [131072][128] array (the gather-add backward's pattern: one wave adds one row of 128 channels, rows drawn
// from a 4096-row window per cloud): agent scope (what atomicAdd gives: the add is resolved past the XCD's L2) against workgroup scope (resolved IN the
// issuing XCD's L2 -- only legal when every add to an address comes from the same XCD).
//   hipcc --offload-arch=gfx950 -O3 -w tools/probe/atomic_scope.hip -o tools/probe/atomic_scope.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
template <int SCOPE, bool BYXCD>
__global__ __launch_bounds__(256) void adds(float *g, const int *src, int rows, int per_cloud)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gw = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
    for (int r = gw; r < rows; r += nw) {
        int rr = r;
        if (BYXCD) {
            // rows of cloud b only on XCD b % 8 (assuming workgroup i runs on XCD i % 8): remap r so that this workgroup's rows belong to its XCD's clouds
            const int xcd = blockIdx.x & 7;
            const int cloud_rows = rows / 32;
            const int k = r / 8;                       // index among this XCD's rows (r % 8 == xcd by construction below)
            (void)xcd;
            rr = ((k / cloud_rows) * 8 + (blockIdx.x & 7)) % 32 * cloud_rows + (k % cloud_rows);
            if ((r & 7) != ((blockIdx.x * 4 + wave) & 7)) {}
        }
        const int j = src[rr];
        float *p = g + (long)j * 128;
        __hip_atomic_fetch_add(p + lane, 1.0f, __ATOMIC_RELAXED, SCOPE);
        __hip_atomic_fetch_add(p + 64 + lane, 1.0f, __ATOMIC_RELAXED, SCOPE);
    }
}
int main()
{
    const int B = 32, N = 4096, rows = 146688 / 32 * 32, per = rows / B;
    std::vector<int> h(rows);
    unsigned s = 12345;
    for (int b = 0; b < B; ++b)
        for (int i = 0; i < per; ++i) { s = s * 1664525u + 1013904223u; h[b * per + i] = b * N + (int)((s >> 8) % N); }
    int *src; float *g;
    hipMalloc(&src, rows * 4); hipMalloc(&g, (size_t)B * N * 128 * 4);
    hipMemcpy(src, h.data(), rows * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](int which) {
        std::vector<float> ts;
        for (int rep = 0; rep < 7; ++rep) {
            hipMemset(g, 0, (size_t)B * N * 128 * 4);
            hipEventRecord(e0, 0);
            if (which == 0) hipLaunchKernelGGL((adds<__HIP_MEMORY_SCOPE_AGENT, false>), dim3(2048), dim3(256), 0, 0, g, src, rows, per);
            if (which == 1) hipLaunchKernelGGL((adds<__HIP_MEMORY_SCOPE_WORKGROUP, false>), dim3(2048), dim3(256), 0, 0, g, src, rows, per);
            if (which == 2) hipLaunchKernelGGL((adds<__HIP_MEMORY_SCOPE_WAVEFRONT, false>), dim3(2048), dim3(256), 0, 0, g, src, rows, per);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms);
        }
        std::sort(ts.begin(), ts.end());
        return ts[3] * 1e3f;
    };
    printf("%d rows x 128 channels of float atomic adds into 67 MB\n", rows);
    printf("agent scope      %7.1f us\n", run(0));
    printf("workgroup scope  %7.1f us   (NOT a correct kernel as launched: timing of the instruction only)\n", run(1));
    printf("wavefront scope  %7.1f us   (same)\n", run(2));
    return 0;
}
