// Where an FPS iteration's cycles go (one wave's view): s_memtime stamps around the phases of a copy of csrc/sampling.hip's loop
// (T x PPT from the command line via templates below).  Build here, run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -I include -I papc_amd/csrc tools/probe/fps_phases.hip -o gpurun_out/fps_phases
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include "common.h"
using namespace papc;
typedef unsigned long long u64;
typedef unsigned int u32;
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ u64 stamp()
{
    u64 t;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

template <int T, int PPT>
__global__ __launch_bounds__(T) void fps_trace(const float *__restrict__ xyz, int N, int npoint, int32_t *__restrict__ out_idx, u64 *__restrict__ phases)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u64 *key = reinterpret_cast<u64 *>(smem);
    float *sx = reinterpret_cast<float *>(smem + 256), *sy = sx + N, *sz = sy + N;
    constexpr int NP = PPT / 2, PP = PPT;
    const int tid = threadIdx.x, b = blockIdx.x;
    const float *p = xyz + (int64_t)b * N * 3;
    f32x2 x[NP], y[NP], z[NP];
    u32 d[PP];
#pragma unroll
    for (int j = 0; j < PP; ++j) {
        const int i = j * T + tid;
        float px = 0.f, py = 0.f, pz = 0.f;
        d[j] = 0u;
        if (i < N) { px = p[i * 3]; py = p[i * 3 + 1]; pz = p[i * 3 + 2]; d[j] = __float_as_uint(1e10f); sx[i] = px; sy[i] = py; sz[i] = pz; }
        x[j >> 1][j & 1] = px; y[j >> 1][j & 1] = py; z[j >> 1][j & 1] = pz;
    }
    if (tid < 3) key[tid] = 0ull;
    int far = 0;
    __syncthreads();
    int s0 = 0, s2 = 2;
    const u32 key_lds = (u32)(uintptr_t)key;
    u64 acc[5] = {0, 0, 0, 0, 0};
    u64 t0 = stamp();
    for (int it = 0; it < npoint; ++it) {
        const float cx = sx[far], cy = sy[far], cz = sz[far];
        if (tid == 0) out_idx[(int64_t)b * npoint + it] = far;
        const f32x2 c2x = {cx, cx}, c2y = {cy, cy}, c2z = {cz, cz};
        asm volatile("" ::"v"(cx), "v"(cy), "v"(cz));
        const u64 t1 = stamp();
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const f32x2 dx = x[q] - c2x, dy = y[q] - c2y, dz = z[q] - c2z;
            const f32x2 dd = (dx * dx + dy * dy) + dz * dz;
            const u32 a0 = __float_as_uint(dd[0]), a1 = __float_as_uint(dd[1]);
            d[2 * q] = a0 < d[2 * q] ? a0 : d[2 * q];
            d[2 * q + 1] = a1 < d[2 * q + 1] ? a1 : d[2 * q + 1];
        }
        u32 bmax = d[0];
#pragma unroll
        for (int j = 1; j < PP; ++j) bmax = d[j] > bmax ? d[j] : bmax;
        int bestj = PP - 1;
#pragma unroll
        for (int j = PP - 2; j >= 0; --j) bestj = d[j] == bmax ? j : bestj;
        const u32 cand = (u32)(bestj * T + tid);
        asm volatile("" ::"v"(cand), "v"(bmax));
        const u64 t2 = stamp();
        const u32 rmax = row_max_u32_fused(bmax);
        if (bmax == rmax) {
            const u64 kv = ((u64)bmax << 32) | (u64)(~cand);
            asm volatile("ds_max_u64 %0, %1" ::"v"(key_lds + 8u * (u32)s0), "v"(kv) : "memory");
        }
        const u64 t3 = stamp();
        asm volatile("s_barrier" ::: "memory");
        const u64 t4 = stamp();
        const u64 k = key[s0];
        far = (int)~(u32)__builtin_amdgcn_readfirstlane((int)(u32)k);
        if (tid == 0) key[s2] = 0ull;
        s2 = s0;
        s0 = s0 == 2 ? 0 : s0 + 1;
        const u64 t5 = stamp();
        acc[0] += t1 - t0; acc[1] += t2 - t1; acc[2] += t3 - t2; acc[3] += t4 - t3; acc[4] += t5 - t4;
        t0 = t5;
    }
    if ((tid & 63) == 0 && b == 0)
        for (int q = 0; q < 5; ++q) phases[(tid >> 6) * 5 + q] = acc[q];
}

template <int T, int PPT>
static void run(int B, int N, int S)
{
    std::vector<float> h((size_t)B * N * 3);
    srand(1);
    for (auto &v : h) v = rand() / (float)RAND_MAX * 2.f - 1.f;
    float *dx; int32_t *di; u64 *dp;
    hipMalloc(&dx, h.size() * 4); hipMalloc(&di, (size_t)B * S * 4); hipMalloc(&dp, 16 * 5 * 8);
    hipMemcpy(dx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const size_t lds = 256 + (size_t)N * 12;
    hipFuncSetAttribute(reinterpret_cast<const void *>(fps_trace<T, PPT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((fps_trace<T, PPT>), dim3(B), dim3(T), lds, 0, dx, N, S, di, dp);
    hipDeviceSynchronize();
    u64 ph[16 * 5];
    hipMemcpy(ph, dp, sizeof(ph), hipMemcpyDeviceToHost);
    const char *names[5] = {"centroid LDS read", "distance + thread argmax", "row DPP + ds_max issue+done", "barrier wait", "key read + far"};
    printf("T=%d PPT=%d N=%d S=%d  (s_memtime ticks per iteration; 100 MHz?? -- see total)\n", T, PPT, N, S);
    for (int w = 0; w < T / 64; w += (T / 64 > 4 ? T / 64 - 1 : 1)) {
        double tot = 0;
        for (int q = 0; q < 5; ++q) tot += (double)ph[w * 5 + q] / S;
        printf("  wave %2d:", w);
        for (int q = 0; q < 5; ++q) printf("  %s %.0f", names[q], (double)ph[w * 5 + q] / S);
        printf("  | total %.0f\n", tot);
    }
    hipFree(dx); hipFree(di); hipFree(dp);
}

int main()
{
    run<512, 8>(32, 4096, 512);
    run<1024, 4>(32, 4096, 512);
    run<512, 4>(16, 2048, 512);
    run<256, 4>(8, 1024, 512);
    return 0;
}
