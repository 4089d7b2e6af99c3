// Probe: operand layout of v_mfma_f32_32x32x16_bf16 and exactness of the 3-way bf16 split (build: hipcc --offload-arch=gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx2 __attribute__((ext_vector_type(2)));

__device__ inline unsigned pk_bf16(float a, float b) {
    bf16x2 v = __builtin_convertvector((floatx2){a, b}, bf16x2);
    return __builtin_bit_cast(unsigned, v);
}

__global__ void probe(const float *A, const float *B, float *C, float *split_err)
{
    const int lane = threadIdx.x, l31 = lane & 31, hi = lane >> 5;
    // 3-way split of A and B rows: plane p of element (row, k)
    unsigned a[3][4], b[3][4];
    float maxerr = 0.f;
    for (int i = 0; i < 4; ++i) {
        float xa0 = A[l31 * 16 + 8 * hi + 2 * i], xa1 = A[l31 * 16 + 8 * hi + 2 * i + 1];
        float xb0 = B[l31 * 16 + 8 * hi + 2 * i], xb1 = B[l31 * 16 + 8 * hi + 2 * i + 1];
        float r0 = xa0, r1 = xa1, s0 = xb0, s1 = xb1;
        float suma0 = 0.f;
        for (int p = 0; p < 3; ++p) {
            unsigned pa = pk_bf16(r0, r1), pb = pk_bf16(s0, s1);
            a[p][i] = pa; b[p][i] = pb;
            float h0 = __uint_as_float(pa << 16), h1 = __uint_as_float(pa & 0xffff0000u);
            suma0 += h0;
            r0 -= h0; r1 -= h1;
            s0 -= __uint_as_float(pb << 16); s1 -= __uint_as_float(pb & 0xffff0000u);
        }
        maxerr = fmaxf(maxerr, fabsf(r0) / fmaxf(fabsf(xa0), 1e-30f));
    }
    split_err[lane] = maxerr;
    floatx16 acc = {0};
    const int pa_[6] = {2, 1, 0, 1, 0, 0}, pb_[6] = {0, 1, 2, 0, 1, 0};  // small terms first
    for (int t = 0; t < 6; ++t) {
        bf16x8 va, vb;
        unsigned ua[4] = {a[pa_[t]][0], a[pa_[t]][1], a[pa_[t]][2], a[pa_[t]][3]};
        unsigned ub[4] = {b[pb_[t]][0], b[pb_[t]][1], b[pb_[t]][2], b[pb_[t]][3]};
        __builtin_memcpy(&va, ua, 16); __builtin_memcpy(&vb, ub, 16);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, vb, acc, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * hi) * 32 + l31] = acc[r];
}

int main()
{
    float hA[32 * 16], hB[32 * 16], hC[32 * 32], hE[64];
    srand(1);
    for (int i = 0; i < 512; ++i) { hA[i] = (float)rand() / RAND_MAX * 2 - 1; hB[i] = ((float)rand() / RAND_MAX * 2 - 1) * (1 + i % 7); }
    float *dA, *dB, *dC, *dE;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, sizeof hC); hipMalloc(&dE, sizeof hE);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dC, dE);
    hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost); hipMemcpy(hE, dE, sizeof hE, hipMemcpyDeviceToHost);
    double maxrel = 0, maxabs = 0, f32rel = 0;
    for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) {
        double ref = 0, mag = 0; float f = 0.f;
        for (int k = 0; k < 16; ++k) { ref += (double)hA[m * 16 + k] * hB[n * 16 + k]; mag += fabs((double)hA[m * 16 + k] * hB[n * 16 + k]); f = fmaf(hA[m * 16 + k], hB[n * 16 + k], f); }
        maxrel = fmax(maxrel, fabs(hC[m * 32 + n] - ref) / mag); maxabs = fmax(maxabs, fabs(hC[m * 32 + n] - ref));
        f32rel = fmax(f32rel, fabs(f - ref) / mag);
    }
    float se = 0; for (int i = 0; i < 64; ++i) se = fmaxf(se, hE[i]);
    printf("C[m][n] = sum_k A[m][k]*B[n][k]: max |err|/sum|ab| = %.3e (fp32 fma chain: %.3e), max abs %.3e, split residual rel %.3e\n", maxrel, f32rel, maxabs, se);
    return maxrel < 1e-6 ? 0 : 1;
}
