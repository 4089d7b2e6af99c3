// Do vector loads of different shapes return (decrement vmcnt) in issue order?  Each wave issues 8 cold 16-byte loads in the
// SGPR-base form (group A, HBM misses), then 16 dword loads in the 64-bit-VGPR-address form from a hot line (group B), then
// waits vmcnt(16): in-order return means all of A has landed.  A's last register is copied right after the wait and compared
// with its value after vmcnt(0).  Second experiment: the same with 24 dword STORES between A and the wait instead of loads.
//   hipcc --offload-arch=gfx950 -O2 tools/probe/vmcnt_order.hip -o tools/probe/vmcnt_order
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void probe(const float *src, const float *hot, float *dst, size_t stride_f, int *bad)
{
    const int lane = threadIdx.x;
    const size_t wid = blockIdx.x;
    const char *sbase = reinterpret_cast<const char *>(src + wid * stride_f);   // wave-uniform
    const unsigned voff = lane * 512;                                              // every lane its own line
    f32x4 A[8];
    for (int i = 0; i < 8; ++i) A[i] = f32x4{-1.f, -1.f, -1.f, -1.f};
    float B[16];
    for (int i = 0; i < 16; ++i) B[i] = -2.f;
    float *d = dst + wid * 64 * 32 + lane;
    float one = 1.f;
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:0" : "+v"(A[0]) : "v"(voff), "s"(sbase));
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:16" : "+v"(A[1]) : "v"(voff), "s"(sbase));
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:32" : "+v"(A[2]) : "v"(voff), "s"(sbase));
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:128" : "+v"(A[3]) : "v"(voff), "s"(sbase));
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:144" : "+v"(A[4]) : "v"(voff), "s"(sbase));
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:256" : "+v"(A[5]) : "v"(voff), "s"(sbase));
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:272" : "+v"(A[6]) : "v"(voff), "s"(sbase));
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:384" : "+v"(A[7]) : "v"(voff), "s"(sbase));
    if (MODE == 0) {
        const float *h = hot + lane;
#define LD(i) asm volatile("global_load_dword %0, %1, off offset:" #i : "+v"(B[i / 256]) : "v"(h))
        asm volatile("global_load_dword %0, %1, off offset:0" : "+v"(B[0]) : "v"(h));
        asm volatile("global_load_dword %0, %1, off offset:256" : "+v"(B[1]) : "v"(h));
        asm volatile("global_load_dword %0, %1, off offset:512" : "+v"(B[2]) : "v"(h));
        asm volatile("global_load_dword %0, %1, off offset:768" : "+v"(B[3]) : "v"(h));
        asm volatile("global_load_dword %0, %1, off offset:1024" : "+v"(B[4]) : "v"(h));
        asm volatile("global_load_dword %0, %1, off offset:1280" : "+v"(B[5]) : "v"(h));
        asm volatile("global_load_dword %0, %1, off offset:1536" : "+v"(B[6]) : "v"(h));
        asm volatile("global_load_dword %0, %1, off offset:1792" : "+v"(B[7]) : "v"(h));
        asm volatile("global_load_dword %0, %1, off offset:2048" : "+v"(B[8]) : "v"(h));
        asm volatile("global_load_dword %0, %1, off offset:2304" : "+v"(B[9]) : "v"(h));
        asm volatile("global_load_dword %0, %1, off offset:2560" : "+v"(B[10]) : "v"(h));
        asm volatile("global_load_dword %0, %1, off offset:2816" : "+v"(B[11]) : "v"(h));
        asm volatile("global_load_dword %0, %1, off offset:3072" : "+v"(B[12]) : "v"(h));
        asm volatile("global_load_dword %0, %1, off offset:3328" : "+v"(B[13]) : "v"(h));
        asm volatile("global_load_dword %0, %1, off offset:3584" : "+v"(B[14]) : "v"(h));
        asm volatile("global_load_dword %0, %1, off offset:3840" : "+v"(B[15]) : "v"(h));
    } else {
        for (int i = 0; i < 16; ++i) asm volatile("global_store_dword %0, %1, off" ::"v"(d + (size_t)i * 64), "v"(one) : "memory");
    }
    f32x4 early;
    asm volatile("s_waitcnt vmcnt(16)\n\tv_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7"
                 : "=&v"(early.x), "=&v"(early.y), "=&v"(early.z), "=&v"(early.w), "+v"(A[7].x), "+v"(A[7].y), "+v"(A[7].z), "+v"(A[7].w));
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(A[0]), "+v"(A[1]), "+v"(A[2]), "+v"(A[3]), "+v"(A[4]), "+v"(A[5]), "+v"(A[6]), "+v"(A[7]));
    for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(B[i]));
    const bool mism = early.x != A[7].x || early.y != A[7].y || early.z != A[7].z || early.w != A[7].w;
    if (mism) atomicAdd(bad, 1);
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += B[i];
    for (int i = 0; i < 7; ++i) s += A[i].x;
    if (s == 12345.678f) atomicAdd(bad, 1000000);
}

int main()
{
    const int NW = 8192;
    const size_t stride_f = 64 * 128;   // 32 KB per wave
    float *src, *dst, *hot;
    int *bad;
    if (hipMalloc(&src, NW * stride_f * 4) != hipSuccess) return 1;
    if (hipMalloc(&dst, (size_t)NW * 64 * 32 * 4) != hipSuccess) return 1;
    if (hipMalloc(&hot, 64 * 1024) != hipSuccess) return 1;
    if (hipMalloc(&bad, 4) != hipSuccess) return 1;
    std::vector<float> h(NW * stride_f);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)(i % 1000) + 1.f;
    (void)hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemset(hot, 0, 64 * 1024);
    for (int mode = 0; mode < 2; ++mode) {
        int tot = 0;
        for (int rep = 0; rep < 5; ++rep) {
            (void)hipMemset(bad, 0, 4);
            if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(NW), dim3(64), 0, 0, src, hot, dst, stride_f, bad);
            else hipLaunchKernelGGL(probe<1>, dim3(NW), dim3(64), 0, 0, src, hot, dst, stride_f, bad);
            int hb = 0;
            (void)hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
            tot += hb;
        }
        printf("%s between the cold 16-byte loads and vmcnt(16): lanes whose 8th cold load had NOT landed: %d of %d\n",
               mode == 0 ? "16 hot dword LOADS " : "16 dword STORES    ", tot, 5 * NW * 64);
    }
    return 0;
}
