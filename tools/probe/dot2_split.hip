// Is x - bf16(x) via v_dot2c_f32_bf16 bit-identical to the expand-and-subtract form used by split3()?   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float floatx2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector((floatx2_t){a, b}, bf16x2_t)); }
__global__ void k(const float *x, float *ra, float *rb, int n, unsigned clo, unsigned chi)
{
    const int i = (blockIdx.x * 256 + threadIdx.x) * 2;
    if (i + 1 >= n) return;
    const float a = x[i], b = x[i + 1];
    const unsigned p = pack_bf16x2(a, b);
    ra[i] = a - __uint_as_float(p << 16);
    ra[i + 1] = b - __uint_as_float(p & 0xffff0000u);
    const bf16x2_t pv = __builtin_bit_cast(bf16x2_t, p);
    const bf16x2_t mlo = __builtin_bit_cast(bf16x2_t, clo), mhi = __builtin_bit_cast(bf16x2_t, chi);
    rb[i] = __builtin_amdgcn_fdot2_f32_bf16(pv, mlo, a, false);
    rb[i + 1] = __builtin_amdgcn_fdot2_f32_bf16(pv, mhi, b, false);
}
int main()
{
    const int n = 1 << 22;
    float *h = (float *)malloc(n * 4), *ha = (float *)malloc(n * 4), *hb = (float *)malloc(n * 4);
    srand(1);
    for (int i = 0; i < n; ++i) {
        unsigned u = ((unsigned)rand() << 16) ^ (unsigned)rand();
        if (i % 3 == 0) u = (u & 0x807fffffu) | ((100u + (unsigned)(rand() % 60)) << 23);   // moderate exponents
        memcpy(&h[i], &u, 4);
        if (h[i] != h[i] || h[i] - h[i] != 0.f) h[i] = 1.0f / (float)(i + 1);
    }
    float *d, *da, *db;
    hipMalloc(&d, n * 4); hipMalloc(&da, n * 4); hipMalloc(&db, n * 4);
    hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 512), dim3(256), 0, 0, d, da, db, n, 0x0000BF80u, 0xBF800000u);
    hipMemcpy(ha, da, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hb, db, n * 4, hipMemcpyDeviceToHost);
    long bad = 0, badnorm = 0;
    for (int i = 0; i < n; ++i) if (memcmp(&ha[i], &hb[i], 4)) { ++bad; if (fabsf(h[i]) > 1e-30f && fabsf(h[i]) < 1e30f) { if (badnorm < 5) printf("x=%a sub=%a dot2=%a\n", h[i], ha[i], hb[i]); ++badnorm; } }
    printf("mismatches %ld of %d (in the normal range: %ld)\n", bad, n, badnorm);
    return 0;
}
