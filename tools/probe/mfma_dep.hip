// Issue rate of v_mfma_f32_32x32x16_bf16 on gfx950 as a function of the number of accumulators it rotates over
// (1 = every MFMA depends on the previous one, 2, 4, 8): does a 2-accumulator wave (WN = 2 stream flavours) stall?
//   hipcc --offload-arch=gfx950 -O2 tools/probe/mfma_dep.hip -o /tmp/md && /tmp/md
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
template <int NA>
__global__ __launch_bounds__(256) void k(int iters, float *out)
{
    floatx16 acc[NA];
    for (int i = 0; i < NA; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(1.0f + i * 0.01f); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j % NA] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j % NA], 0, 0, 0);
    }
    float r = 0.f;
    for (int i = 0; i < NA; ++i) r += acc[i][0] + acc[i][9];
    if (r == 123.456f) out[0] = r;
}
template <int NA>
static void run(float *d)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    hipLaunchKernelGGL(k<NA>, dim3(256), dim3(256), 0, 0, 100, d);   // one wave per SIMD
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NA>, dim3(256), dim3(256), 0, 0, iters, d);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%d accumulator(s): %.2f ns per MFMA (one wave per SIMD)\n", NA, 1e6 * ms / iters / 8);
}
int main()
{
    float *d;
    hipMalloc(&d, 4);
    run<1>(d); run<2>(d); run<4>(d); run<8>(d);
    return 0;
}
