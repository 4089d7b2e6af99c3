// xyz1.hip -- the first layer of a set-abstraction stack that is fed by coordinates only (gfx950).
//
// SA1 of the classifiers groups xyz alone (points = None: /root/reference/PAPC/models/layers/pointnet2_basic_layers.py:152-153,
// classify/pointnet2/pointnet2.py:11,33): its first conv is y[m, c] = W[c] . x[m] + b[c] with x[m] = xyz[idx[m]] - new_xyz[m / K], THREE
// input channels, on M = B*S*K = 524 288 rows.  As a stored tensor y is 134 MB that four later kernels re-read; as a function it is 12
// bytes per row.  Everything train-mode BatchNorm and the backward need from the dense [M, C] activations of this layer is a function
// of the inputs' second moments (the PFN layer's Gram-matrix argument, pfn.hip, with K = 3):
//
//   forward   sum_m y_c = W_c . Sx + M b_c,   sum_m y_c^2 = W_c Sxx W_c^T + 2 b_c W_c . Sx + M b_c^2      (Sx = sum x, Sxx = sum x x^T)
//             -> mean, invstd, scale, shift with NO pass over y; the layer folds into the next one's operand,
//                a[m, c] = relu(scale_c y + shift_c) = relu(wf_c . x[m] + t_c),   wf_c = scale_c W_c,  t_c = scale_c b_c + shift_c
//             (mlp_stream.hip A_XYZ computes it in registers: 3 FMAs per element instead of a 4-byte load)
//   backward  with p = dz * [a > 0]:  S_c = sum_m p,  T_c = sum_m p x  (one pass over dz, this file) give
//             dbeta = S,  dgamma = invstd (W_c . T_c + (b_c - mean_c) S_c),  c1 = S / M,  c2 = dgamma / M,
//             dW_c = scale_c (T_c - c1 Sx - c2 invstd (W_c Sxx + (b_c - mean_c) Sx))
//             -- no BN-backward sums in the dX epilogue of the layer above, no second read of dz, no y.
//
// Sx / Sxx are accumulated in float64 (products of two fp32 values are exact there): the variance is a difference of second moments.
// The grouped, centred coordinates themselves are written once per step as float4 rows xc [M, 4] (8 MB) that the consumers stream.
#include "mlp_loaders.h"

namespace papc {

constexpr int XYZ_T = 256;           // threads per workgroup
constexpr int XYZ_MAX_PARTS = 512;   // workgroups of the grouping pass = rows of its partial-moment buffer

// ---- group + centre (sample_and_group :146-147 for a coordinates-only layer) and the 10 moments ---------------------------------
// partial[b][0..9] = sum x, y, z, xx, xy, xz, yy, yz, zz, count over the workgroup's rows (float64)
__global__ __launch_bounds__(XYZ_T) void xyz_group_kernel(GroupSrc g, int64_t M, float4 *__restrict__ xc, double *__restrict__ partial)
{
    __shared__ double red[XYZ_T / 64][10];
    double s[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) s[i] = 0.0;
    const int64_t per = (M + gridDim.x - 1) / gridDim.x;
    const int64_t m0 = (int64_t)blockIdx.x * per, m1 = min(M, m0 + per);
    for (int64_t m = m0 + threadIdx.x; m < m1; m += XYZ_T) {
        const uint32_t grp = fdiv((uint32_t)m, g.divK);
        const uint32_t b = fdiv(grp, g.divS);
        int j = g.idx ? g.idx[m] : (int)(m - (int64_t)b * g.S * g.K);
        const bool ok = j >= 0 && j < g.N;                       // (no-hit sentinel N: a zero row, as the row kernels' GROUP loader)
        j = ok ? j : 0;
        const float *pp = g.xyz + (int64_t)b * g.sb + (int64_t)j * g.sn;
        const float *qq = g.new_xyz + (int64_t)grp * 3;
        float x = pp[0] - qq[0], y = pp[g.sc] - qq[1], z = pp[2 * g.sc] - qq[2];
        if (!ok) { x = 0.f; y = 0.f; z = 0.f; }
        xc[m] = make_float4(x, y, z, 0.f);
        const double dx = x, dy = y, dz = z;
        s[0] += dx; s[1] += dy; s[2] += dz;
        s[3] += dx * dx; s[4] += dx * dy; s[5] += dx * dz; s[6] += dy * dy; s[7] += dy * dz; s[8] += dz * dz;
        s[9] += 1.0;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        double v = s[i];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) red[wave][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < 10) {
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < XYZ_T / 64; ++w) v += red[w][threadIdx.x];
        partial[(int64_t)blockIdx.x * 16 + threadIdx.x] = v;
    }
}

// moments [parts][16] -> [16], fixed order (weight-independent: runs with the grouping pass, off the critical path)
__global__ __launch_bounds__(256) void xyz_gram_fold_kernel(const double *__restrict__ partial, int parts, double *__restrict__ gram)
{
    __shared__ double red[16][16];
    const int e = threadIdx.x & 15, sl = threadIdx.x >> 4;
    double s = 0.0;
    if (e < 10) for (int t = sl; t < parts; t += 16) s += partial[(int64_t)t * 16 + e];
    red[sl][e] = s;
    __syncthreads();
    if (threadIdx.x < 16) {
        double v = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) v += red[q][threadIdx.x];
        gram[threadIdx.x] = threadIdx.x < 10 ? v : 0.0;
    }
}

// moments [parts][16] -> gram[16] (fixed order), then per channel the BatchNorm constants and the folded first layer
__global__ __launch_bounds__(256) void xyz_l1_finalize_kernel(const double *__restrict__ partial, int parts, double M, const float *__restrict__ w, int ldw,
                                                              int xcol0, const float *__restrict__ bias, const float *__restrict__ gamma,
                                                              const float *__restrict__ beta, float eps, float momentum, int C, float *mean,
                                                              float *invstd, float *scale, float *shift, float *rmean, float *rvar, float *wf,
                                                              double *gram)
{
    __shared__ double red[16][16];
    __shared__ double G[10];
    const int e = threadIdx.x & 15, sl = threadIdx.x >> 4;      // 16 entries (10 used) x 16 slices
    double s = 0.0;
    if (e < 10) for (int t = sl; t < parts; t += 16) s += partial[(int64_t)t * 16 + e];
    red[sl][e] = s;
    __syncthreads();
    if (threadIdx.x < 10) {
        double v = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) v += red[q][threadIdx.x];
        G[threadIdx.x] = v;
        gram[threadIdx.x] = v;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        const double w0 = w[(int64_t)c * ldw + xcol0], w1 = w[(int64_t)c * ldw + xcol0 + 1], w2 = w[(int64_t)c * ldw + xcol0 + 2];
        const double b = bias ? (double)bias[c] : 0.0;
        const double mx = G[0] / M, my = G[1] / M, mz = G[2] / M;
        // covariance of the inputs (biased), then the variance of y = W x + b: W Cov W^T
        const double cxx = G[3] / M - mx * mx, cxy = G[4] / M - mx * my, cxz = G[5] / M - mx * mz;
        const double cyy = G[6] / M - my * my, cyz = G[7] / M - my * mz, czz = G[8] / M - mz * mz;
        const double mu = w0 * mx + w1 * my + w2 * mz + b;
        double var = w0 * (w0 * cxx + w1 * cxy + w2 * cxz) + w1 * (w0 * cxy + w1 * cyy + w2 * cyz) + w2 * (w0 * cxz + w1 * cyz + w2 * czz);
        if (var < 0.0) var = 0.0;
        const double is = 1.0 / sqrt(var + (double)eps);
        const double sc = (gamma ? (double)gamma[c] : 1.0) * is;
        const double sh = (beta ? (double)beta[c] : 0.0) - mu * sc;
        mean[c] = (float)mu; invstd[c] = (float)is; scale[c] = (float)sc; shift[c] = (float)sh;
        if (rmean) rmean[c] = momentum * rmean[c] + (1.f - momentum) * (float)mu;     // paddle: momentum weighs the running value
        if (rvar) rvar[c] = momentum * rvar[c] + (1.f - momentum) * (float)var;
        // a = relu(scale (W x + b) + shift) = relu(wf . x + t)
        wf[c * 4 + 0] = (float)(sc * w0); wf[c * 4 + 1] = (float)(sc * w1); wf[c * 4 + 2] = (float)(sc * w2);
        wf[c * 4 + 3] = (float)(sc * b + sh);
    }
}

// ---- backward: one pass over dz.  A lane owns 4 channels; per row p = dz [wf . x + t > 0]; sums of p and p x -------------------------
__global__ __launch_bounds__(XYZ_T) void xyz_l1_bwd_kernel(const float *__restrict__ dz, const float4 *__restrict__ xc, const float *__restrict__ wf,
                                                           int64_t M, int C, int64_t rows_per_chunk, float *__restrict__ partial)
{
    __shared__ float red[XYZ_T * 16];
    const int tid = threadIdx.x;
    const int CQ = C >> 2;                   // channel quads per row (<= 64)
    const int RSL = XYZ_T / CQ;              // row slots of the workgroup
    const int cq = tid % CQ, slot = tid / CQ;
    const bool act = slot < RSL;
    const int c = cq * 4;
    float4 k[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) k[i] = ld4(wf + (c + i) * 4);
    float a[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i][0] = a[i][1] = a[i][2] = a[i][3] = 0.f;
    const int64_t mbeg = (int64_t)blockIdx.x * rows_per_chunk, mend = min(M, mbeg + rows_per_chunk);
    constexpr int U = 4;                     // rows in flight per lane
    if (act) {
        for (int64_t m0 = mbeg + slot; m0 < mend; m0 += (int64_t)U * RSL) {
            float4 vz[U], vx[U];
            bool ok[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t m = m0 + (int64_t)u * RSL;
                ok[u] = m < mend;
                const int64_t mc = ok[u] ? m : mbeg;
                vz[u] = ld4(dz + mc * C + c);
                vx[u] = xc[mc];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (!ok[u]) continue;
                const float g[4] = {vz[u].x, vz[u].y, vz[u].z, vz[u].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float z = fmaf(k[i].z, vx[u].z, fmaf(k[i].y, vx[u].y, fmaf(k[i].x, vx[u].x, k[i].w)));
                    const float p = z > 0.f ? g[i] : 0.f;
                    a[i][0] = fmaf(p, vx[u].x, a[i][0]); a[i][1] = fmaf(p, vx[u].y, a[i][1]); a[i][2] = fmaf(p, vx[u].z, a[i][2]);
                    a[i][3] += p;
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) red[tid * 16 + i * 4 + j] = act ? a[i][j] : 0.f;
    __syncthreads();
    // thread t < C * 4: (channel, slot j); row slots summed in slot order (deterministic)
    float *out = partial + (int64_t)blockIdx.x * C * 4;
    for (int t = tid; t < C * 4; t += XYZ_T) {
        const int ch = t >> 2, j = t & 3;
        const int q = ch >> 2, i = ch & 3;
        float sacc = 0.f;
        for (int sl = 0; sl < RSL; ++sl) sacc += red[(sl * CQ + q) * 16 + i * 4 + j];
        out[t] = sacc;
    }
}

// partial [parts][C][4] (T0, T1, T2, S) + the input moments -> dgamma, dbeta, dW [C][3]  (closed form, float64).
// A workgroup owns 4 channels: 16 entries x 64 slices of the partial rows, summed in fixed order.  The kernel is one latency chain (it is the
// LAST node of the backward): with 64 slices a thread's share of <= 768 partial rows is 12 loads, all in flight at once -- one memory round
// trip instead of six (17.6 -> ~7 us in the replayed step).
__global__ __launch_bounds__(1024) void xyz_l1_bwd_finalize_kernel(const float *__restrict__ partial, int parts, double M, int C, const double *__restrict__ gram,
                                                                   const float *__restrict__ w, int ldw, int xcol0, const float *__restrict__ bias,
                                                                   const float *__restrict__ mean, const float *__restrict__ invstd,
                                                                   const float *__restrict__ scale, float *dgamma, float *dbeta, float *dw, int accumulate)
{
    __shared__ double red[64][17], red2[8][17];
    const int e = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int e0 = blockIdx.x * 16;                  // first entry (channel * 4 + slot) of this workgroup
    const int c = blockIdx.x * 4 + threadIdx.x;
    const bool fin = threadIdx.x < 4 && c < C;       // (the finalizing threads fetch their constants with the partial rows, not behind the barriers)
    const int cc = fin ? c : 0;
    const double w0 = w[(int64_t)cc * ldw + xcol0], w1 = w[(int64_t)cc * ldw + xcol0 + 1], w2 = w[(int64_t)cc * ldw + xcol0 + 2];
    const double b = bias ? (double)bias[cc] : 0.0;
    const double mu = mean[cc], is = invstd[cc], sc = scale[cc];
    double gq[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) gq[i] = gram[i];
    double s = 0.0;
    if (e0 + e < C * 4) {
        for (int t0 = sl; t0 < parts; t0 += 64 * 12) {   // 12 loads in flight per thread
            float v[12];
#pragma unroll
            for (int q = 0; q < 12; ++q) v[q] = (t0 + 64 * q < parts) ? partial[(int64_t)(t0 + 64 * q) * C * 4 + e0 + e] : 0.f;
#pragma unroll
            for (int q = 0; q < 12; ++q) s += (double)v[q];
        }
    }
    red[sl][e] = s;
    __syncthreads();
    if (sl < 8) {                                    // 64 slices -> 8 -> 1, fixed order
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q) t += red[sl * 8 + q][e];
        red2[sl][e] = t;
    }
    __syncthreads();
    if (sl == 0) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q) t += red2[q][e];
        red[0][e] = t;
    }
    __syncthreads();
    if (fin) {
        const double T0 = red[0][threadIdx.x * 4 + 0], T1 = red[0][threadIdx.x * 4 + 1], T2 = red[0][threadIdx.x * 4 + 2], S = red[0][threadIdx.x * 4 + 3];
        const double dg = is * (w0 * T0 + w1 * T1 + w2 * T2 + (b - mu) * S);     // sum p xhat
        dbeta[c] = accumulate ? dbeta[c] + (float)S : (float)S;
        dgamma[c] = accumulate ? dgamma[c] + (float)dg : (float)dg;
        const double c1 = S / M, c2 = dg / M;
        const double Sx[3] = {gq[0], gq[1], gq[2]};
        const double Sxx[3][3] = {{gq[3], gq[4], gq[5]}, {gq[4], gq[6], gq[7]}, {gq[5], gq[7], gq[8]}};
        const double T[3] = {T0, T1, T2};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double yx = w0 * Sxx[0][k] + w1 * Sxx[1][k] + w2 * Sxx[2][k] + (b - mu) * Sx[k];      // sum_m (y - mean) x_k
            const float gk_ = (float)(sc * (T[k] - c1 * Sx[k] - c2 * is * yx));
            float *o = dw + (int64_t)c * ldw + xcol0 + k;
            *o = accumulate ? *o + gk_ : gk_;
        }
    }
}

}  // namespace papc

using namespace papc;

extern "C" {

int papc_xyz_parts(int64_t M) { return M >= 1 ? (int)std::min<int64_t>(cdiv(M, 1024), XYZ_MAX_PARTS) : 0; }
int papc_xyz_bwd_parts(int64_t M) { return M >= 1 ? (int)cdiv(M, 1024) : 0; }

int papc_xyz_group_f32(const papc_group_src *grp, int B, float *xc, double *gram_partial, papc_stream_t stream)
{
    PAPC_REQUIRE(grp && grp->xyz && grp->new_xyz && xc && gram_partial, PAPC_E_INVALID, "papc_xyz_group_f32: null pointer");
    PAPC_REQUIRE(B >= 1 && grp->N >= 1 && grp->S >= 1 && grp->K >= 1, PAPC_E_INVALID, "papc_xyz_group_f32: bad sizes");
    PAPC_REQUIRE(aligned16(xc), PAPC_E_INVALID, "papc_xyz_group_f32: xc must be 16-byte aligned");
    const int64_t M = (int64_t)B * grp->S * grp->K;
    PAPC_REQUIRE(M < (1ll << 31), PAPC_E_UNSUPPORTED, "papc_xyz_group_f32: M=%lld >= 2^31 rows", (long long)M);
    GroupSrc g;
    memset(&g, 0, sizeof(g));
    g.xyz = grp->xyz; g.sb = grp->sb; g.sn = grp->sn; g.sc = grp->sc; g.new_xyz = grp->new_xyz; g.idx = grp->idx;
    g.N = grp->N; g.S = grp->S; g.K = grp->K; g.D = 0; g.xyz_first = 1;
    g.divK = make_fastdiv((uint32_t)grp->K); g.divS = make_fastdiv((uint32_t)grp->S);
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_GROUP, st);
    hipLaunchKernelGGL(xyz_group_kernel, dim3((unsigned)papc_xyz_parts(M)), dim3(XYZ_T), 0, st, g, M, reinterpret_cast<float4 *>(xc), gram_partial);
    return check_launch("papc_xyz_group_f32");
}

int papc_xyz_gram_fold_f32(const double *gram_partial, int parts, double *gram, papc_stream_t stream)
{
    PAPC_REQUIRE(gram_partial && gram && parts >= 1, PAPC_E_INVALID, "papc_xyz_gram_fold_f32: bad arguments");
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_GROUP, st);
    hipLaunchKernelGGL(xyz_gram_fold_kernel, dim3(1), dim3(256), 0, st, gram_partial, parts, gram);
    return check_launch("papc_xyz_gram_fold_f32");
}

int papc_xyz_l1_finalize_f32(const double *gram_partial, int parts, int64_t M, const float *w, int ldw, int xcol0, const float *bias,
                             const float *gamma, const float *beta, float eps, float momentum, int C, float *mean, float *invstd, float *scale,
                             float *shift, float *running_mean, float *running_var, float *wf, double *gram, papc_stream_t stream)
{
    PAPC_REQUIRE(gram_partial && w && mean && invstd && scale && shift && wf && gram, PAPC_E_INVALID, "papc_xyz_l1_finalize_f32: null pointer");
    PAPC_REQUIRE(parts >= 1 && M >= 1 && C >= 1 && ldw >= xcol0 + 3 && xcol0 >= 0, PAPC_E_INVALID, "papc_xyz_l1_finalize_f32: bad sizes");
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_MISC, st);
    hipLaunchKernelGGL(xyz_l1_finalize_kernel, dim3(1), dim3(256), 0, st, gram_partial, parts, (double)M, w, ldw, xcol0, bias, gamma, beta, eps,
                       momentum, C, mean, invstd, scale, shift, running_mean, running_var, wf, gram);
    return check_launch("papc_xyz_l1_finalize_f32");
}

int papc_xyz_l1_bwd_f32(const float *dz, const float *xc, const float *wf, int64_t M, int C, float *partial, papc_stream_t stream)
{
    PAPC_REQUIRE(dz && xc && wf && partial, PAPC_E_INVALID, "papc_xyz_l1_bwd_f32: null pointer");
    PAPC_REQUIRE(M >= 1 && C >= 4 && C <= 256 && C % 4 == 0, PAPC_E_UNSUPPORTED, "papc_xyz_l1_bwd_f32: C=%d must be a multiple of 4 in [4, 256]", C);
    PAPC_REQUIRE(aligned16(dz) && aligned16(xc) && aligned16(wf), PAPC_E_INVALID, "papc_xyz_l1_bwd_f32: 16-byte alignment");
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_BWD_DW, st);
    hipLaunchKernelGGL(xyz_l1_bwd_kernel, dim3((unsigned)papc_xyz_bwd_parts(M)), dim3(XYZ_T), 0, st, dz, reinterpret_cast<const float4 *>(xc), wf, M, C,
                       (int64_t)1024, partial);
    return check_launch("papc_xyz_l1_bwd_f32");
}

int papc_xyz_l1_bwd_finalize_f32(const float *partial, int parts, int64_t M, int C, const double *gram, const float *w, int ldw, int xcol0,
                                 const float *bias, const float *mean, const float *invstd, const float *scale, float *dgamma, float *dbeta,
                                 float *dw, int accumulate, papc_stream_t stream)
{
    PAPC_REQUIRE(partial && gram && w && mean && invstd && scale && dgamma && dbeta && dw, PAPC_E_INVALID, "papc_xyz_l1_bwd_finalize_f32: null pointer");
    PAPC_REQUIRE(parts >= 1 && M >= 1 && C >= 1 && C <= 256 && ldw >= xcol0 + 3 && xcol0 >= 0, PAPC_E_INVALID, "papc_xyz_l1_bwd_finalize_f32: bad sizes");
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_BWD_DW, st);
    hipLaunchKernelGGL(xyz_l1_bwd_finalize_kernel, dim3((unsigned)cdiv(C, 4)), dim3(1024), 0, st, partial, parts, (double)M, C, gram, w, ldw, xcol0, bias, mean, invstd,
                       scale, dgamma, dbeta, dw, accumulate);
    return check_launch("papc_xyz_l1_bwd_finalize_f32");
}

}  // extern "C"
