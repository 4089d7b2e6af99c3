// interp.hip -- the interpolation half of PointNetFeaturePropagation
// (/root/reference/PAPC/models/layers/pointnet2_basic_layers.py:284-335) for gfx950.
//
//   three_nn        : per query point the three smallest squared distances to the S support points (:315-318,
//                     canonical square_distance arithmetic, ascending, lowest index on ties), the inverse-distance
//                     weights (:320-322) and the true neighbour indices
//   interpolate     : out[b,n,:] = sum_j points2[b, idx[b,n,j], :] * w[b,n,j]   (:323) and its gradient
//
// Index-exact / bit-exact contract like sampling.hip: compiled with -ffp-contract=off, explicit fmaf only where the
// reference's matmul has one.
#include "common.h"

namespace papc {

constexpr int NN_T = 256;
constexpr int NN_CHUNK = 4096;  // support points staged per pass (64 KiB of float4)

__global__ __launch_bounds__(NN_T) void three_nn_kernel(const float *__restrict__ xyz1, int64_t sb1, int64_t sn1, int64_t sc1,
                                                        const float *__restrict__ xyz2, int64_t sb2, int64_t sn2, int64_t sc2,
                                                        int N, int S, float *__restrict__ dist3, int32_t *__restrict__ idx3,
                                                        float *__restrict__ w3)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4 *pts = reinterpret_cast<float4 *>(smem);
    const int b = blockIdx.y;
    const int n = blockIdx.x * NN_T + threadIdx.x;
    const bool ok = n < N;
    const float *q = xyz1 + (int64_t)b * sb1 + (int64_t)(ok ? n : 0) * sn1;
    const float a0 = q[0], a1 = q[sc1], a2 = q[2 * sc1];
    const float aa = (a0 * a0 + a1 * a1) + a2 * a2;  // sum(src ** 2, -1)  (:37)
    float d0 = INFINITY, d1 = INFINITY, d2 = INFINITY;
    int i0 = 0, i1 = 0, i2 = 0;
    const float *p2 = xyz2 + (int64_t)b * sb2;
    for (int c0 = 0; c0 < S; c0 += NN_CHUNK) {
        const int n_in = min(NN_CHUNK, S - c0);
        if (c0 > 0) __syncthreads();
        for (int i = threadIdx.x; i < n_in; i += NN_T) {
            const int64_t g = (int64_t)(c0 + i) * sn2;
            const float x = p2[g], y = p2[g + sc2], z = p2[g + 2 * sc2];
            pts[i] = make_float4(x, y, z, (x * x + y * y) + z * z);
        }
        __syncthreads();
        for (int j = 0; j < n_in; ++j) {
            const float4 pt = pts[j];  // broadcast read
            const float dot = fmaf(a2, pt.z, fmaf(a1, pt.y, a0 * pt.x));
            float d = -2.0f * dot;
            d = d + aa;
            d = d + pt.w;
            // stable insertion (strict <): equal distances keep ascending index order, like a stable sort of the row
            if (d < d2) {
                const int jj = c0 + j;
                if (d < d1) {
                    d2 = d1; i2 = i1;
                    if (d < d0) { d1 = d0; i1 = i0; d0 = d; i0 = jj; }
                    else { d1 = d; i1 = jj; }
                } else { d2 = d; i2 = jj; }
            }
        }
    }
    if (ok) {
        const int64_t o = ((int64_t)b * N + n) * 3;
        dist3[o] = d0; dist3[o + 1] = d1; dist3[o + 2] = d2;
        idx3[o] = i0; idx3[o + 1] = i1; idx3[o + 2] = i2;
        const float r0 = 1.0f / (d0 + 1e-8f), r1 = 1.0f / (d1 + 1e-8f), r2 = 1.0f / (d2 + 1e-8f);  // :320
        const float norm = (r0 + r1) + r2;                                                          // :321
        w3[o] = r0 / norm; w3[o + 1] = r1 / norm; w3[o + 2] = r2 / norm;                            // :322
    }
}

template <int V>
__global__ __launch_bounds__(256) void interpolate_kernel(const float *__restrict__ points2, const int32_t *__restrict__ idx3,
                                                          const float *__restrict__ w3, int S, int D, int64_t rows, int N,
                                                          float *__restrict__ out)
{
    const int DV = D / V;
    const int64_t total = rows * DV;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = e / DV;
        const int c = (int)(e - row * DV) * V;
        const int64_t b = row / N;
        const int j0 = idx3[row * 3], j1 = idx3[row * 3 + 1], j2 = idx3[row * 3 + 2];
        const float w0 = w3[row * 3], w1 = w3[row * 3 + 1], w2 = w3[row * 3 + 2];
        const float *pa = points2 + (b * S + j0) * (int64_t)D + c;
        const float *pb = points2 + (b * S + j1) * (int64_t)D + c;
        const float *pc = points2 + (b * S + j2) * (int64_t)D + c;
        float *o = out + row * (int64_t)D + c;
#pragma unroll
        for (int i = 0; i < V; ++i) o[i] = (pa[i] * w0 + pb[i] * w1) + pc[i] * w2;  // sum(index_points(.)*weight, axis=2)  (:323)
    }
}

__global__ __launch_bounds__(256) void interpolate_bwd_kernel(const float *__restrict__ gout, const int32_t *__restrict__ idx3,
                                                              const float *__restrict__ w3, int S, int D, int64_t rows, int N,
                                                              float *__restrict__ gpoints2)
{
    const int64_t total = rows * D;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = e / D;
        const int c = (int)(e - row * D);
        const int64_t b = row / N;
        const float g = gout[e];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int s = idx3[row * 3 + j];
            unsafeAtomicAdd(&gpoints2[(b * S + s) * (int64_t)D + c], g * w3[row * 3 + j]);
        }
    }
}

// The same gradient when every query's neighbours are support points 0, 1, 2 -- what the reference's sort-then-argsort computes
// (pointnet2_basic_layers.py:316-317, layers.PointNetFeaturePropagation neighbours="reference"): gpoints2[b, j, :] = sum_n w3[b, n, j] g[b, n, :]
// for j < 3 and zero for every other support point.  A column reduction per cloud instead of 3 N D atomics on three rows: one workgroup
// per (cloud, 64-channel slice), 16 row lanes, fixed-order LDS fold (deterministic); the workgroup also writes its slice's zeros, so the
// output needs no fill.
__global__ __launch_bounds__(1024) void interpolate_bwd_first3_kernel(const float *__restrict__ gout, const float *__restrict__ w3, int N, int S, int D,
                                                                      float *__restrict__ gpoints2)
{
    __shared__ float red[3][16][64];
    const int b = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    const bool cok = c < D;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    const float *g = gout + (int64_t)b * N * D + (cok ? c : 0);
    const float *w = w3 + (int64_t)b * N * 3;
    for (int n0 = rl; n0 < N; n0 += 16 * 8) {       // eight rows' loads in flight per lane (one at a time: N / 16 dependent round trips, 44 us at N = 2048)
        float v[8], w0[8], w1[8], w2[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int n = n0 + 16 * j < N ? n0 + 16 * j : n0;
            v[j] = cok ? g[(int64_t)n * D] : 0.f;
            w0[j] = w[n * 3 + 0]; w1[j] = w[n * 3 + 1]; w2[j] = w[n * 3 + 2];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (n0 + 16 * j < N) { a0 = fmaf(v[j], w0[j], a0); a1 = fmaf(v[j], w1[j], a1); a2 = fmaf(v[j], w2[j], a2); }
    }
    red[0][rl][threadIdx.x & 63] = a0; red[1][rl][threadIdx.x & 63] = a1; red[2][rl][threadIdx.x & 63] = a2;
    __syncthreads();
    float *o = gpoints2 + (int64_t)b * S * D;
    if (threadIdx.x < 192) {
        const int j = threadIdx.x >> 6, cc = blockIdx.x * 64 + (threadIdx.x & 63);
        float t = 0.f;
#pragma unroll
        for (int l = 0; l < 16; ++l) t += red[j][l][threadIdx.x & 63];
        if (cc < D && j < S) o[(int64_t)j * D + cc] = t;
    }
    for (int s = 3 + rl; s < S; s += 16)
        if (cok) o[(int64_t)s * D + c] = 0.f;
}

static inline unsigned ew_grid(int64_t total) { return (unsigned)std::max<int64_t>(1, std::min<int64_t>(cdiv(total, 256), 256 * 16)); }

}  // namespace papc

using namespace papc;

extern "C" {

int papc_three_nn_f32(const float *xyz1, int64_t sb1, int64_t sn1, int64_t sc1, const float *xyz2, int64_t sb2, int64_t sn2,
                      int64_t sc2, int B, int N, int S, float *dist3, int32_t *idx3, float *weight3, papc_stream_t stream)
{
    PAPC_REQUIRE(xyz1 && xyz2 && dist3 && idx3 && weight3, PAPC_E_INVALID, "papc_three_nn_f32: null pointer");
    PAPC_REQUIRE(B >= 1 && N >= 1 && S >= 3 && B <= 65535, PAPC_E_INVALID, "papc_three_nn_f32: B=%d N=%d S=%d (S must be >= 3)", B, N, S);
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_GROUP, st);
    const size_t lds = (size_t)std::min(S, NN_CHUNK) * 16;
    if (lds > 48 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(three_nn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return check_launch("papc_three_nn_f32: hipFuncSetAttribute");
    }
    hipLaunchKernelGGL(three_nn_kernel, dim3((unsigned)cdiv(N, NN_T), (unsigned)B), dim3(NN_T), lds, st, xyz1, sb1, sn1, sc1, xyz2, sb2,
                       sn2, sc2, N, S, dist3, idx3, weight3);
    return check_launch("papc_three_nn_f32");
}

int papc_three_interpolate_f32(const float *points2, const int32_t *idx3, const float *weight3, int B, int N, int S, int D,
                               float *out, papc_stream_t stream)
{
    PAPC_REQUIRE(points2 && idx3 && weight3 && out, PAPC_E_INVALID, "papc_three_interpolate_f32: null pointer");
    PAPC_REQUIRE(B >= 1 && N >= 1 && S >= 1 && D >= 1, PAPC_E_INVALID, "papc_three_interpolate_f32: bad sizes");
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_GROUP, st);
    const int64_t rows = (int64_t)B * N;
    if (D % 4 == 0) hipLaunchKernelGGL(interpolate_kernel<4>, dim3(ew_grid(rows * (D / 4))), dim3(256), 0, st, points2, idx3, weight3, S, D, rows, N, out);
    else hipLaunchKernelGGL(interpolate_kernel<1>, dim3(ew_grid(rows * D)), dim3(256), 0, st, points2, idx3, weight3, S, D, rows, N, out);
    return check_launch("papc_three_interpolate_f32");
}

int papc_three_interpolate_bwd_f32(const float *grad_out, const int32_t *idx3, const float *weight3, int B, int N, int S, int D,
                                   float *grad_points2, papc_stream_t stream)
{
    PAPC_REQUIRE(grad_out && idx3 && weight3 && grad_points2, PAPC_E_INVALID, "papc_three_interpolate_bwd_f32: null pointer");
    PAPC_REQUIRE(B >= 1 && N >= 1 && S >= 1 && D >= 1, PAPC_E_INVALID, "papc_three_interpolate_bwd_f32: bad sizes");
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_GROUP, st);
    const int64_t rows = (int64_t)B * N;
    hipLaunchKernelGGL(interpolate_bwd_kernel, dim3(ew_grid(rows * D)), dim3(256), 0, st, grad_out, idx3, weight3, S, D, rows, N, grad_points2);
    return check_launch("papc_three_interpolate_bwd_f32");
}

int papc_three_interpolate_bwd_first3_f32(const float *grad_out, const float *weight3, int B, int N, int S, int D, float *grad_points2,
                                          papc_stream_t stream)
{
    PAPC_REQUIRE(grad_out && weight3 && grad_points2, PAPC_E_INVALID, "papc_three_interpolate_bwd_first3_f32: null pointer");
    PAPC_REQUIRE(B >= 1 && B <= 65535 && N >= 1 && S >= 3 && D >= 1, PAPC_E_INVALID, "papc_three_interpolate_bwd_first3_f32: B=%d N=%d S=%d D=%d (S >= 3)", B, N, S, D);
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_GROUP, st);
    hipLaunchKernelGGL(interpolate_bwd_first3_kernel, dim3((unsigned)cdiv(D, 64), (unsigned)B), dim3(1024), 0, st, grad_out, weight3, N, S, D, grad_points2);
    return check_launch("papc_three_interpolate_bwd_first3_f32");
}

}  // extern "C"
