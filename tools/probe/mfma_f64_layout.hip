// Operand / accumulator layout of v_mfma_f64_16x16x4_f64 on gfx950 (probe for pfn.hip's Gram pass).
//   hipcc --offload-arch=gfx950 -O2 tools/probe/mfma_f64_layout.hip -o /tmp/mfma_f64_layout && /tmp/mfma_f64_layout
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ void k(double *out)
{
    const int l = threadIdx.x;
    // hypothesis: A lane l holds A[i = l & 15][k = l >> 4], B lane l holds B[k = l >> 4][j = l & 15]
    const int i = l & 15, kk = l >> 4;
    const double a = (double)(i + 1) * (kk == 0 ? 1.0 : (kk == 1 ? 0.001 : 0.0));      // A[i][0] = i+1, A[i][1] = (i+1)/1000
    const double b = (kk == 0 ? (double)((l & 15) + 1) * 100.0 : (kk == 1 ? 7.0 : 0.0));  // B[0][j] = 100 (j+1), B[1][j] = 7
    double4_t c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];   // expect D[i][j] = 100 (i+1)(j+1) + 0.007 (i+1)
}
int main()
{
    double *d, h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int ok_a = 1, ok_b = 1;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            const double v = h[l * 4 + r];
            const int j = l & 15;
            const int ia = 4 * (l >> 4) + r, ib = (l >> 4) + 4 * r;
            if (v != 100.0 * (ia + 1) * (j + 1) + 0.007 * (ia + 1)) ok_a = 0;
            if (v != 100.0 * (ib + 1) * (j + 1) + 0.007 * (ib + 1)) ok_b = 0;
        }
    printf("layout i=4*(l>>4)+r: %d   layout i=(l>>4)+4*r: %d\n", ok_a, ok_b);
    for (int l = 0; l < 64; l += 13) printf("lane %2d: %.3f %.3f %.3f %.3f\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    return 0;
}
