// Does the 6-bit vmcnt counter saturate (instead of stalling issue) when a wave has more than 63 VMEM operations outstanding?
// Each wave: NST dword stores (never waited for), then one 16-byte load A from a cold line, then 8 more loads B;
// s_waitcnt vmcnt(8) must then guarantee A (loads return in order: if A is outstanding so are the 8 younger B).  A's register is
// copied right after the wait and compared with its value after vmcnt(0).
//   hipcc --offload-arch=gfx950 -O2 tools/probe/vmcnt_sat.hip -o /tmp/vmcnt_sat && /tmp/vmcnt_sat
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const float *src, float *dst, int nst, size_t stride_f, int *bad)
{
    const int lane = threadIdx.x;
    const size_t wid = blockIdx.x;
    float *d = dst + (wid * 128 * 64 + lane);               // store target: one dword per lane, 128 distinct rows
    const float *a = src + wid * stride_f + lane * 4;          // cold 16-byte loads
    f32x4 A = {-1.f, -1.f, -1.f, -1.f}, B0 = A, B1 = A, B2 = A, B3 = A, B4 = A, B5 = A, B6 = A, B7 = A;
    float one = 1.f;
    for (int i = 0; i < nst; ++i)
        asm volatile("global_store_dword %0, %1, off" ::"v"(d + (size_t)i * 64), "v"(one) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(A) : "v"(a));
    asm volatile("global_load_dwordx4 %0, %1, off offset:1024" : "+v"(B0) : "v"(a));
    asm volatile("global_load_dwordx4 %0, %1, off offset:2048" : "+v"(B1) : "v"(a));
    asm volatile("global_load_dwordx4 %0, %1, off offset:3072" : "+v"(B2) : "v"(a));
    asm volatile("global_load_dwordx4 %0, %1, off offset:-1024" : "+v"(B3) : "v"(a + 2048));
    asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(B4) : "v"(a + 2048));
    asm volatile("global_load_dwordx4 %0, %1, off offset:1024" : "+v"(B5) : "v"(a + 2048));
    asm volatile("global_load_dwordx4 %0, %1, off offset:2048" : "+v"(B6) : "v"(a + 2048));
    asm volatile("global_load_dwordx4 %0, %1, off offset:3072" : "+v"(B7) : "v"(a + 2048));
    f32x4 early;
    asm volatile("s_waitcnt vmcnt(8)\n\tv_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7"
                 : "=&v"(early.x), "=&v"(early.y), "=&v"(early.z), "=&v"(early.w), "+v"(A.x), "+v"(A.y), "+v"(A.z), "+v"(A.w));
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(A), "+v"(B0), "+v"(B1), "+v"(B2), "+v"(B3), "+v"(B4), "+v"(B5), "+v"(B6), "+v"(B7));
    const bool mism = early.x != A.x || early.y != A.y || early.z != A.z || early.w != A.w;
    if (mism) atomicAdd(bad, 1);
    if (B0.x + B1.x + B2.x + B3.x + B4.x + B5.x + B6.x + B7.x == 12345.f) atomicAdd(bad, 1000000);
}

int main()
{
    const int NW = 4096;
    const size_t stride_f = 8192;   // 32 KB per wave
    float *src, *dst;
    int *bad;
    hipMalloc(&src, NW * stride_f * 4);
    hipMalloc(&dst, (size_t)NW * 128 * 64 * 4);
    hipMalloc(&bad, 4);
    std::vector<float> h(NW * stride_f);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)(i % 1000) + 1.f;
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int nst : {0, 16, 32, 48, 54, 55, 56, 60, 64, 72, 96, 128}) {
        int tot = 0;
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(bad, 0, 4);
            // evict: touch a big buffer?  (src is 128 MB: mostly cold in L2 between reps)
            hipLaunchKernelGGL(probe, dim3(NW), dim3(64), 0, 0, src, dst, nst, stride_f, bad);
            int hb = 0;
            hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
            tot += hb;
        }
        printf("stores outstanding before the loads: %3d  (+9 loads)  -> lanes with a stale read after vmcnt(8): %d\n", nst, tot);
    }
    return 0;
}
