// bn_ops.hip -- the HBM-bound passes around the MFMA kernels (gfx950): train-mode BatchNorm statistics
// finalisation, BN+ReLU+max over nsample, the reductions of the BN/ReLU/max backward, fixed-order partial
// sums, and the harness's Adam step.
// Reference ops: nn.BatchNorm2D (train) + F.relu + paddle.max(axis=2),
// /root/reference/PAPC/models/layers/pointnet2_basic_layers.py:190,217,219.
#include "common.h"

namespace papc {

template <int V> struct Vec;
template <> struct Vec<4> {
    float4 v;
    __device__ static Vec load(const float *p) { Vec r; r.v = *reinterpret_cast<const float4 *>(p); return r; }
    __device__ void store(float *p) const { *reinterpret_cast<float4 *>(p) = v; }
    __device__ float &operator[](int i) { return (&v.x)[i]; }
    __device__ float operator[](int i) const { return (&v.x)[i]; }
};
template <> struct Vec<1> {
    float v;
    __device__ static Vec load(const float *p) { Vec r; r.v = *p; return r; }
    __device__ void store(float *p) const { *p = v; }
    __device__ float &operator[](int) { return v; }
    __device__ float operator[](int) const { return v; }
};

// this lane's share (rows tl, tl + 64, ...) of the [n_tiles][2][C] partial rows of channel c, summed in row order; the loads of 8
// rows are issued before the first add (the rows are a dependent chain of ~0.5 us loads otherwise)
template <int D>
__device__ __forceinline__ void sum_partial_rows_d(const float *__restrict__ part, int n_tiles, int C, int c, int tl, double &s1, double &s2)
{
    for (int t0 = tl; t0 < n_tiles; t0 += 64 * D) {
        float a[D], b[D];
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const int t = t0 + 64 * j;
            const int64_t o = (int64_t)(t < n_tiles ? t : t0) * 2 * C + c;
            a[j] = part[o];
            b[j] = part[o + C];
        }
#pragma unroll
        for (int j = 0; j < D; ++j) {
            if (t0 + 64 * j < n_tiles) { s1 += (double)a[j]; s2 += (double)b[j]; }
        }
    }
}
// (the same sums in the same order at either depth; beyond 512 rows -- the gather-add layer's 2048 -- sixteen rows in flight halve the round trips)
__device__ __forceinline__ void sum_partial_rows(const float *__restrict__ part, int n_tiles, int C, int c, int tl, double &s1, double &s2)
{
    if (n_tiles > 512) sum_partial_rows_d<16>(part, n_tiles, C, c, tl, s1, s2);
    else sum_partial_rows_d<8>(part, n_tiles, C, c, tl, s1, s2);
}

// lane l of a wave holds the double of part-lane l: the sum over the 64 part-lanes in a FIXED shape, left in lane 0 -- groups of four
// part-lanes as a two-level tree, the sixteen groups added in order (the shape the 16-wave form of the finalizes used: same bits)
__device__ __forceinline__ double lane_get(double v, int l)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void fold_part_lanes(double &s1, double &s2)
{
    s1 += __shfl_xor(s1, 1); s2 += __shfl_xor(s2, 1);
    s1 += __shfl_xor(s1, 2); s2 += __shfl_xor(s2, 2);
    double t1 = 0.0, t2 = 0.0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { t1 += lane_get(s1, 4 * w); t2 += lane_get(s2, 4 * w); }
    s1 = t1; s2 = t2;
}

// ---------------------------------------------------------------------------------------------------
// stats partials [n_tiles][2][C] -> mean, invstd, scale = gamma*invstd, shift = beta - mean*scale
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void bn_finalize_kernel(const float *__restrict__ part, int n_tiles, int64_t M, int C,
                                                        const float *__restrict__ gamma, const float *__restrict__ beta,
                                                        float eps, float momentum, float *mean, float *invstd, float *scale,
                                                        float *shift, float *rmean, float *rvar)
{
    // ONE WAVE per channel, no LDS, <= 64 registers: a launch-sized kernel has to find room on a CU that another stream's persistent
    // GEMM has filled to the last KB of LDS (config 3 runs its MSG branches on three streams: the 1024-thread / 16 KB form of this
    // kernel waited 13-21 us on average, up to 0.3 ms, for a whole CU to drain).  Lane = part-lane.
    const int c = blockIdx.x, tl = threadIdx.x;
    // (the channel's affine parameters and running statistics are requested ahead of the partial rows: one round trip, not two)
    const float gam = gamma ? gamma[c] : 1.f, bet = beta ? beta[c] : 0.f;
    const float rm0 = rmean ? rmean[c] : 0.f, rv0 = rvar ? rvar[c] : 0.f;
    double s1 = 0.0, s2 = 0.0;
    sum_partial_rows(part, n_tiles, C, c, tl, s1, s2);
    fold_part_lanes(s1, s2);
    if (tl == 0) {
        const double mu = s1 / (double)M;
        double var = s2 / (double)M - mu * mu;  // biased variance (paddle BatchNorm training)
        if (var < 0.0) var = 0.0;
        const double is = 1.0 / sqrt(var + (double)eps);
        const double sc = (double)gam * is;
        mean[c] = (float)mu;
        invstd[c] = (float)is;
        scale[c] = (float)sc;
        shift[c] = (float)((double)bet - mu * sc);
        if (rmean) rmean[c] = momentum * rm0 + (1.f - momentum) * (float)mu;   // paddle: momentum weighs the running value
        if (rvar) rvar[c] = momentum * rv0 + (1.f - momentum) * (float)var;
    }
}

// ---------------------------------------------------------------------------------------------------
// out[g,c] = max_k relu(scale*y[g*K+k,c]+shift), argmax = first k attaining it
// ---------------------------------------------------------------------------------------------------
template <int V>
__global__ __launch_bounds__(256) void bn_relu_max_kernel(const float *__restrict__ y, const float *__restrict__ scale,
                                                          const float *__restrict__ shift, int64_t G, int K, int C,
                                                          float *__restrict__ out, int32_t *__restrict__ argmax)
{
    const int CV = C / V;
    const int64_t total = G * CV;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t g = e / CV;
        const int c = (int)(e - g * CV) * V;
        const Vec<V> sc = Vec<V>::load(scale + c), sh = Vec<V>::load(shift + c);
        Vec<V> best;
        int bi[V];
#pragma unroll
        for (int i = 0; i < V; ++i) { best[i] = -1.0f; bi[i] = 0; }
        const float *yp = y + (g * K) * (int64_t)C + c;
        for (int k = 0; k < K; ++k) {
            const Vec<V> v = Vec<V>::load(yp + (int64_t)k * C);
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const float z = relu_np(fmaf(sc[i], v[i], sh[i]));
                if (z > best[i] || z != z) { best[i] = z; bi[i] = k; }      // (a NaN wins and stays: torch.max / paddle.max propagate it)
            }
        }
        best.store(out + g * C + c);
        if (argmax) {
#pragma unroll
            for (int i = 0; i < V; ++i) argmax[g * C + c + i] = bi[i];
        }
    }
}

__global__ __launch_bounds__(256) void bn_select_max_kernel(float *__restrict__ gmax, const float *__restrict__ gmin,
                                                            const int32_t *__restrict__ amax, const int32_t *__restrict__ amin,
                                                            const float *__restrict__ scale, const float *__restrict__ shift,
                                                            int64_t total, int C, float *__restrict__ out, int32_t *__restrict__ argmax)
{
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        const float sc = scale[c], sh = shift[c];
        const bool up = sc >= 0.f;   // relu(sc*y+sh) is non-decreasing in y for sc >= 0, non-increasing otherwise
        const float sel = up ? gmax[e] : gmin[e];
        out[e] = relu_np(fmaf(sc, sel, sh));
        gmax[e] = sel;               // the raw pre-BN value behind out[e]: the backward reductions read it instead of gathering y
        if (argmax) argmax[e] = up ? amax[e] : amin[e];
    }
}

template <int V>
__global__ __launch_bounds__(256) void bn_relu_kernel(const float *__restrict__ y, const float *__restrict__ scale,
                                                      const float *__restrict__ shift, int64_t M, int C, float *__restrict__ z)
{
    const int CV = C / V;
    const int64_t total = M * CV;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % CV) * V;
        const Vec<V> sc = Vec<V>::load(scale + c), sh = Vec<V>::load(shift + c);
        Vec<V> v = Vec<V>::load(y + e * V);
#pragma unroll
        for (int i = 0; i < V; ++i) v[i] = fmaxf(fmaf(sc[i], v[i], sh[i]), 0.f);
        v.store(z + e * V);
    }
}

// ---------------------------------------------------------------------------------------------------
// backward reductions: p = dz*[scale*y+shift > 0]; partial[blk] = (sum p, sum p*xhat) per channel
// threads: (cg = channel group, rl = row lane); block covers a contiguous range of rows / groups
// ---------------------------------------------------------------------------------------------------
template <int V, bool MAXMODE>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float *__restrict__ dz, const float *__restrict__ gout,
                                                            const int32_t *__restrict__ argmax, int K,
                                                            const float *__restrict__ y, const float *__restrict__ mean,
                                                            const float *__restrict__ invstd, const float *__restrict__ scale,
                                                            const float *__restrict__ shift, int64_t M, int C,
                                                            float *__restrict__ part, float *__restrict__ psel_out = nullptr)
{
    __shared__ float red[2][256 * 4];
    const int CV = C / V;                       // channel groups
    const int CG = CV < 256 ? CV : 256;         // channel groups per pass
    const int RL = 256 / CG;                    // row lanes
    const int cgi = threadIdx.x % CG, rl = threadIdx.x / CG;
    const int64_t units = MAXMODE ? M / K : M;  // groups or rows
    const int64_t per = (units + gridDim.x - 1) / gridDim.x;
    const int64_t u0 = (int64_t)blockIdx.x * per, u1 = min(units, u0 + per);

    for (int cb = 0; cb < CV; cb += CG) {
        const int cg = cb + cgi;
        Vec<V> a1, a2;
#pragma unroll
        for (int i = 0; i < V; ++i) { a1[i] = 0.f; a2[i] = 0.f; }
        if (cg < CV && rl < RL) {
            const int c = cg * V;
            const Vec<V> sc = Vec<V>::load(scale + c), sh = Vec<V>::load(shift + c);
            const Vec<V> mu = Vec<V>::load(mean + c), is = Vec<V>::load(invstd + c);
            for (int64_t u = u0 + rl; u < u1; u += RL) {
                if (MAXMODE) {
                    Vec<V> g = Vec<V>::load(gout + u * C + c);
                    Vec<V> ys;
                    if (dz) {        // MAX mode: dz carries ysel [M/K, C], the raw y at the argmax (papc_bn_select_max_f32)
                        ys = Vec<V>::load(dz + u * C + c);
                    } else {
#pragma unroll
                        for (int i = 0; i < V; ++i) ys[i] = y[(u * K + argmax[u * C + c + i]) * (int64_t)C + c + i];
                    }
#pragma unroll
                    for (int i = 0; i < V; ++i) {
                        const float yv = ys[i];
                        const float z = fmaf(sc[i], yv, sh[i]);
                        const float p = z > 0.f ? g[i] : 0.f;
                        a1[i] += p;
                        a2[i] = fmaf(p, (yv - mu[i]) * is[i], a2[i]);
                        g[i] = sc[i] * p;
                    }
                    if (psel_out) g.store(psel_out + u * C + c);   // scale * p: the sparse operand of papc_mlp_bwd_dx_max_f32 / _dw_max_f32
                } else {
                    const Vec<V> d = Vec<V>::load(dz + u * C + c);
                    const Vec<V> yv = Vec<V>::load(y + u * C + c);
#pragma unroll
                    for (int i = 0; i < V; ++i) {
                        const float z = fmaf(sc[i], yv[i], sh[i]);
                        const float p = z > 0.f ? d[i] : 0.f;
                        a1[i] += p;
                        a2[i] = fmaf(p, (yv[i] - mu[i]) * is[i], a2[i]);
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < V; ++i) { red[0][threadIdx.x * V + i] = a1[i]; red[1][threadIdx.x * V + i] = a2[i]; }
        __syncthreads();
        // thread t < CG*V sums over row lanes for channel (cb*V + t)
        for (int t = threadIdx.x; t < CG * V; t += 256) {
            const int c = cb * V + t;
            if (c < C) {
                const int g = t / V, i = t - g * V;
                float s1 = 0.f, s2 = 0.f;
                for (int r = 0; r < RL; ++r) { s1 += red[0][(r * CG + g) * V + i]; s2 += red[1][(r * CG + g) * V + i]; }
                part[((int64_t)blockIdx.x * 2 + 0) * C + c] = s1;
                part[((int64_t)blockIdx.x * 2 + 1) * C + c] = s2;
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(64) void bn_bwd_finalize_kernel(const float *__restrict__ part, int n_tiles, int64_t M, int C,
                                                            float *dgamma, float *dbeta, float *c1, float *c2, int accumulate)
{
    // one wave per channel, no LDS (see bn_finalize_kernel)
    const int c = blockIdx.x, tl = threadIdx.x;
    double s1 = 0.0, s2 = 0.0;
    sum_partial_rows(part, n_tiles, C, c, tl, s1, s2);
    fold_part_lanes(s1, s2);
    if (tl == 0) {
        if (dbeta) dbeta[c] = (accumulate & 1) ? dbeta[c] + (float)s1 : (float)s1;
        if (dgamma) dgamma[c] = (accumulate & 1) ? dgamma[c] + (float)s2 : (float)s2;
        // bit 1 of `accumulate`: the BatchNorm ran on its RUNNING statistics (eval mode): the batch-mean terms of its backward vanish
        const bool eval_bn = (accumulate & 2) != 0;
        c1[c] = eval_bn ? 0.f : (float)(s1 / (double)M);
        c2[c] = eval_bn ? 0.f : (float)(s2 / (double)M);
    }
}

// eval-mode BatchNorm constants from the running statistics: the same four vectors bn_finalize_kernel derives from a batch
__global__ void bn_eval_consts_kernel(const float *__restrict__ rmean, const float *__restrict__ rvar, const float *__restrict__ gamma,
                                      const float *__restrict__ beta, float eps, int C, float *mean, float *invstd, float *scale, float *shift)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double mu = (double)rmean[c];
    const double is = 1.0 / sqrt((double)rvar[c] + (double)eps);
    const double sc = (gamma ? (double)gamma[c] : 1.0) * is;
    mean[c] = (float)mu;
    invstd[c] = (float)is;
    scale[c] = (float)sc;
    shift[c] = (float)((beta ? (double)beta[c] : 0.0) - mu * sc);
}

// backward of out[g,c] = max_k x[g*K+k, c] (first maximum wins): dx[g*K+k, c] = (k == argmax[g,c]) ? gout[g,c] : 0, written densely
__global__ __launch_bounds__(256) void group_max_bwd_kernel(const float *__restrict__ gout, const int32_t *__restrict__ argmax, int64_t G,
                                                            int K, int C, float *__restrict__ dx)
{
    const int64_t total = G * K * (int64_t)C;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        const int64_t m = e / C;
        const int64_t g = m / K;
        const int k = (int)(m - g * K);
        dx[e] = (argmax[g * C + c] == k) ? gout[g * C + c] : 0.f;
    }
}

__global__ __launch_bounds__(1024) void reduce_partials_kernel(const float *__restrict__ part, int n_chunks, int64_t n,
                                                               int64_t ld, float *__restrict__ out, int64_t n1,
                                                               float *__restrict__ out2, int accumulate)
{
    __shared__ float red[16][64];
    const int el = threadIdx.x & 63, cl = threadIdx.x >> 6;  // lane = element (coalesced), wave = chunk lane
    const int64_t i = (int64_t)blockIdx.x * 64 + el;
    float s = 0.f;
    if (i < n) {
        for (int t0 = cl; t0 < n_chunks; t0 += 16 * 8) {     // 8 loads in flight per lane; summed in chunk order
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int t = t0 + 16 * j;
                v[j] = part[(int64_t)(t < n_chunks ? t : t0) * ld + i];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) s += (t0 + 16 * j < n_chunks) ? v[j] : 0.f;
        }
    }
    red[cl][el] = s;
    __syncthreads();
    if (cl == 0 && i < n) {
#pragma unroll
        for (int g = 1; g < 16; ++g) s += red[g][el];
        float *o = i < n1 ? out + i : out2 + (i - n1);  // elements [0,n1) -> out, [n1,n) -> out2
        *o = accumulate ? *o + s : s;
    }
}

// few chunks of many elements (the group_all layers): a lane owns 4 consecutive elements, the 4 waves of a workgroup split the
// chunks; n1, ld multiples of 4 and 16-byte aligned pointers
__global__ __launch_bounds__(256) void reduce_partials_wide_kernel(const float *__restrict__ part, int n_chunks, int64_t n, int64_t ld,
                                                                   float *__restrict__ out, int64_t n1, float *__restrict__ out2,
                                                                   int accumulate)
{
    __shared__ float4 red[4][64];
    const int el = threadIdx.x & 63, cl = threadIdx.x >> 6;
    const int64_t i = ((int64_t)blockIdx.x * 64 + el) * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n) {
        for (int t0 = cl; t0 < n_chunks; t0 += 4 * 4) {
            float4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int t = t0 + 4 * j;
                v[j] = *reinterpret_cast<const float4 *>(part + (int64_t)(t < n_chunks ? t : t0) * ld + i);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (t0 + 4 * j < n_chunks) { s.x += v[j].x; s.y += v[j].y; s.z += v[j].z; s.w += v[j].w; }
            }
        }
    }
    red[cl][el] = s;
    __syncthreads();
    if (cl == 0 && i < n) {
#pragma unroll
        for (int g = 1; g < 4; ++g) { const float4 r = red[g][el]; s.x += r.x; s.y += r.y; s.z += r.z; s.w += r.w; }
        float4 *o = reinterpret_cast<float4 *>(i < n1 ? out + i : out2 + (i - n1));
        if (accumulate) { const float4 p = *o; s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w; }
        *o = s;
    }
}

template <bool ZERO>
__global__ __launch_bounds__(256) void adam_kernel(float *__restrict__ p, float *__restrict__ g, float *__restrict__ m,
                                                   float *__restrict__ v, int64_t n, float lr, float b1, float b2, float omb1, float omb2,
                                                   float eps, float wd, float bc1, float bc2, float gscale)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float gi = g[i] * gscale + wd * p[i];  // L2 regularisation folded into the gradient (paddle weight_decay=float)
        const float mi = b1 * m[i] + omb1 * gi;       // omb = 1 - beta formed in double on the host: 1.f - 0.999f is off by 1.3e-5
        const float vi = b2 * v[i] + omb2 * gi * gi;
        m[i] = mi; v[i] = vi;
        p[i] -= lr * (mi / bc1) / (sqrtf(vi / bc2) + eps);
        if (ZERO) g[i] = 0.f;
    }
}

// Adam with the step count in DEVICE memory (papc_adam_step_dev_f32): the launch takes no host scalar that changes from step to step, so it
// can live inside a captured hipGraph (an eager launch behind a graph replay starts 8-20 us after the graph's last kernel).  step_dev[0] is
// advanced by papc_adam_tick (one thread, anywhere earlier in the step -- e.g. on the sampling branch), never by this kernel: every block
// reads the same value.  The bias corrections are formed in double from it, like the host form.
// TICK: the kernel advances the count itself -- it applies step step_dev[0] + 1 and the block that FINISHES last (a ticket in step_dev[1],
// which returns to zero) stores that number: every block has read the old value by then.  One atomic per block; for a step without a branch
// to hide the tick launch on.
template <bool ZERO, bool TICK, int V>
__global__ __launch_bounds__(256) void adam_dev_kernel(float *__restrict__ p, float *__restrict__ g, float *__restrict__ m,
                                                       float *__restrict__ v, int64_t n, float lr, double beta1, double beta2,
                                                       float eps, float wd, int64_t *__restrict__ step_dev, float gscale)
{
    __shared__ float s_bc[2];
    __shared__ int64_t s_t;
    // the first elements' loads go out BEFORE the bias corrections (two double-precision pow of a device-side step count, a microsecond of one
    // thread's time with the other 255 behind a barrier): V = 4 elements per thread where the bucket allows
    const int64_t nv = n / V, stride = (int64_t)gridDim.x * blockDim.x, i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    Vec<V> gq, pq, mq, vq;
    if (i0 < nv) { gq = Vec<V>::load(g + i0 * V); pq = Vec<V>::load(p + i0 * V); mq = Vec<V>::load(m + i0 * V); vq = Vec<V>::load(v + i0 * V); }
    if (threadIdx.x == 0) {
        const int64_t ti = step_dev[0] + (TICK ? 1 : 0);
        const double t = (double)ti;
        s_t = ti;
        s_bc[0] = (float)(1.0 - pow(beta1, t));
        s_bc[1] = (float)(1.0 - pow(beta2, t));
    }
    __syncthreads();
    const float bc1 = s_bc[0], bc2 = s_bc[1];
    const float b1 = (float)beta1, b2 = (float)beta2, omb1 = (float)(1.0 - beta1), omb2 = (float)(1.0 - beta2);
    for (int64_t i = i0; i < nv; i += stride) {
        if (i != i0) { gq = Vec<V>::load(g + i * V); pq = Vec<V>::load(p + i * V); mq = Vec<V>::load(m + i * V); vq = Vec<V>::load(v + i * V); }
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const float gi = gq[e] * gscale + wd * pq[e];
            const float mi = b1 * mq[e] + omb1 * gi;
            const float vi = b2 * vq[e] + omb2 * gi * gi;
            mq[e] = mi; vq[e] = vi;
            pq[e] -= lr * (mi / bc1) / (sqrtf(vi / bc2) + eps);
            gq[e] = 0.f;
        }
        mq.store(m + i * V); vq.store(v + i * V); pq.store(p + i * V);
        if (ZERO) gq.store(g + i * V);
    }
    if (TICK) {
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned long long done = atomicAdd(reinterpret_cast<unsigned long long *>(step_dev + 1), 1ull);
            if (done == (unsigned long long)gridDim.x - 1ull) {
                step_dev[0] = s_t;
                __hip_atomic_store(reinterpret_cast<unsigned long long *>(step_dev + 1), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}
__global__ void adam_tick_kernel(int64_t *step_dev) { step_dev[0] += 1; }

// ---- a device-side gate between two streams (papc_flag_set / papc_flag_wait) -----------------------------------------------------------------
// A forked branch inside a replayed hipGraph costs the MAIN chain ~60 us per step on MI355X whatever the branch holds (bench.py, round 5: an
// empty branch = fork + join).  Two graphs on two streams with no edge between them cost nothing -- but the second one must not start before the
// first has reached a given point.  The gate is FOUR 32-bit words: [0] the number of openings so far (papc_flag_set adds `value`, release, agent
// scope), [1] the number of openings waited for so far (owned by the waiting stream), [2] a STICKY count of waits that gave up, [3] reserved.
// A wait expects opening number [1] + 1 and spins (one lane, bounded by `max_spins` sleeps so that a mis-ordered launch sequence cannot hang the
// queue) until [0] has reached it; it never writes [0], so a wait that gave up and the late opening behind it leave the two counts aligned --
// the next wait expects the NEXT opening (rounds 4-5 used one word that the wait returned to zero: a late opening then left a stale 1 and every
// later wait passed one opening early, silently).  A give-up is counted in [2] for the host to read and abort on (papc_flag_timeouts' contract:
// correctness of whatever the gate orders must not rest on it once [2] != 0).
__global__ void flag_set_kernel(unsigned *flag, unsigned value, int64_t *counter)
{
    if (counter) counter[0] += 1;        // (a step counter riding on the same launch: FlatAdam's device step count)
    __hip_atomic_fetch_add(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void flag_wait_kernel(unsigned *flag, long long max_spins, int widx)
{
    if (threadIdx.x != 0) return;
    const unsigned want = flag[widx] + 1u;        // (widx: this waiter's own count -- word 1, or word 3 for a second waiter on the same gate)
    long long n = 0;
    while ((int)(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
        if (n >= max_spins) { __hip_atomic_fetch_add(flag + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        __builtin_amdgcn_s_sleep(32);
        ++n;
    }
    flag[widx] = want;
}

__global__ __launch_bounds__(256) void fill_kernel(float *__restrict__ p, int64_t n, float v)
{
    const int64_t n4 = n >> 2;
    float4 *p4 = reinterpret_cast<float4 *>(p);
    const float4 v4 = make_float4(v, v, v, v);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) p4[i] = v4;
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) p[(n4 << 2) + threadIdx.x] = v;
}

// dst[r, c] = src[r, c] (or dst[c, r] with transpose) for a [rows, cols] block of matrices with arbitrary row strides
__global__ __launch_bounds__(256) void copy2d_kernel(const float *__restrict__ src, int64_t src_ld, float *__restrict__ dst, int64_t dst_ld,
                                                     int rows, int cols, int transpose)
{
    const int64_t total = (int64_t)rows * cols;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        // consecutive threads write consecutive destination elements (the small operand being re-laid-out is read through L2)
        int r, c;
        if (transpose) { c = (int)(e / rows); r = (int)(e - (int64_t)c * rows); dst[(int64_t)c * dst_ld + r] = src[(int64_t)r * src_ld + c]; }
        else { r = (int)(e / cols); c = (int)(e - (int64_t)r * cols); dst[(int64_t)r * dst_ld + c] = src[(int64_t)r * src_ld + c]; }
    }
}

// Up to 8 strided 3-D copies in one launch: dst[b*db + r*dr + c*dc] = src[b*sb + r*sr + c*sc] over a [B, R, C] index space each -- the
// concatenations and transposed copies around the layers (torch.cat of the MSG branches, pointnet2_basic_layers.py:280; of points1 and the
// interpolated features, :326-327; points.transpose(1, 2) :205) and their backward splits.  32 x 32 (r, c) tiles go through LDS so that both
// the reads and the writes run along whichever index is contiguous on their side.
struct CopyJobs {
    const float *src[8];
    float *dst[8];
    int B[8], R[8], C[8];
    int64_t sb[8], sr[8], sc[8], db[8], dr[8], dc[8];
};

__global__ __launch_bounds__(256) void copy_strided_batch_kernel(CopyJobs j)
{
    __shared__ float tile[32][33];
    const int q = blockIdx.y;
    const int R = j.R[q], C = j.C[q];
    const int tr = (R + 31) >> 5, tc = (C + 31) >> 5;
    const int64_t ntiles = (int64_t)j.B[q] * tr * tc;
    const float *src = j.src[q];
    float *dst = j.dst[q];
    const int64_t sb = j.sb[q], sr = j.sr[q], sc = j.sc[q], db = j.db[q], dr = j.dr[q], dc = j.dc[q];
    const bool src_along_c = sc == 1 || sr != 1, dst_along_c = dc == 1 || dr != 1;     // the index consecutive threads walk, per side
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int b = (int)(t / (tr * tc));
        const int rem = (int)(t - (int64_t)b * tr * tc);
        const int r0 = (rem / tc) * 32, c0 = (rem % tc) * 32;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int a = ty + 8 * i;                                    // slow index of the pass
            const int r = r0 + (src_along_c ? a : tx), c = c0 + (src_along_c ? tx : a);
            if (r < R && c < C) tile[r - r0][c - c0] = src[b * sb + r * sr + c * sc];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int a = ty + 8 * i;
            const int r = r0 + (dst_along_c ? a : tx), c = c0 + (dst_along_c ? tx : a);
            if (r < R && c < C) dst[b * db + r * dr + c * dc] = tile[r - r0][c - c0];
        }
        __syncthreads();
    }
}

// out[r * out_ld + c] (+)= sum_t part[t * ld + r * cols + c]   (fixed summation tree: deterministic)
// A workgroup = 16 consecutive elements x 64 chunk slices: slice s sums chunks s, s+64, ... in order, then the 64 slice sums are folded
// through LDS in a fixed binary tree.  (One thread per element walking all chunks took 80+ us for the 2048 x 192 gather-add partials.)
__global__ __launch_bounds__(1024) void reduce_partials_strided_kernel(const float *__restrict__ part, int n_chunks, int64_t ld, int rows, int cols,
                                                                       float *__restrict__ out, int64_t out_ld, int accumulate)
{
    __shared__ float red[64][17];
    const int el = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int64_t n = (int64_t)rows * cols;
    const int64_t e = (int64_t)blockIdx.x * 16 + el;
    float s = 0.f;
    if (e < n) {
        for (int t0 = sl; t0 < n_chunks; t0 += 64 * 4) {
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (t0 + 64 * j < n_chunks) ? part[(int64_t)(t0 + 64 * j) * ld + e] : 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) s += v[j];
        }
    }
    red[sl][el] = s;
    __syncthreads();
    for (int h = 32; h >= 1; h >>= 1) {
        if (sl < h) red[sl][el] += red[sl + h][el];
        __syncthreads();
    }
    if (sl == 0 && e < n) {
        const int r = (int)(e / cols), c = (int)(e - (int64_t)r * cols);
        float *o = out + (int64_t)r * out_ld + c;
        *o = accumulate ? *o + red[0][el] : red[0][el];
    }
}

__global__ __launch_bounds__(256) void scale_by_kernel(const float *__restrict__ x, const float *__restrict__ scalar, int64_t n, float *__restrict__ out)
{
    const float g = scalar[0];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = x[i] * g;
}

static inline unsigned ew_grid(int64_t total) { return (unsigned)std::max<int64_t>(1, std::min<int64_t>(cdiv(total, 256), 256 * 16)); }

// The same reduction for FEW, LONG groups (PointNet-Basic: 8 clouds x 1024 points x 1024 channels): one thread per (group, channel quad)
// walking K rows is 2048 threads on the whole chip (300 us for 33 MB).  Here a workgroup owns (group, 256 channels) and its 16 row slices
// walk K / 16 consecutive rows each; the slices are folded through LDS in row order with a strict >, i.e. the first row attaining the max
// wins, as in the serial scan.
static __device__ __forceinline__ float4 ld4f(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__global__ __launch_bounds__(1024) void bn_relu_max_split_kernel(const float *__restrict__ y, const float *__restrict__ scale,
                                                                const float *__restrict__ shift, int64_t G, int K, int C,
                                                                float *__restrict__ out, int32_t *__restrict__ argmax)
{
    __shared__ float sv[16][64][4];
    __shared__ int si[16][64][4];
    const int cq = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int64_t g = blockIdx.x;
    const int c = (blockIdx.y * 64 + cq) * 4;
    const bool cok = c < C;       // C % 4 == 0
    const int per = (K + 15) / 16;
    const int k0 = sl * per, k1 = min(K, k0 + per);
    float4 best = make_float4(-1.f, -1.f, -1.f, -1.f);
    int4 bi = make_int4(0, 0, 0, 0);
    if (cok) {
        const float4 sc = ld4f(scale + c), sh = ld4f(shift + c);
        const float *yp = y + (g * K) * (int64_t)C + c;
        for (int k = k0; k < k1; ++k) {
            const float4 v = ld4f(yp + (int64_t)k * C);
            float z;
            z = relu_np(fmaf(sc.x, v.x, sh.x)); if (z > best.x || z != z) { best.x = z; bi.x = k; }
            z = relu_np(fmaf(sc.y, v.y, sh.y)); if (z > best.y || z != z) { best.y = z; bi.y = k; }
            z = relu_np(fmaf(sc.z, v.z, sh.z)); if (z > best.z || z != z) { best.z = z; bi.z = k; }
            z = relu_np(fmaf(sc.w, v.w, sh.w)); if (z > best.w || z != z) { best.w = z; bi.w = k; }
        }
    }
    sv[sl][cq][0] = best.x; sv[sl][cq][1] = best.y; sv[sl][cq][2] = best.z; sv[sl][cq][3] = best.w;
    si[sl][cq][0] = bi.x; si[sl][cq][1] = bi.y; si[sl][cq][2] = bi.z; si[sl][cq][3] = bi.w;
    __syncthreads();
    if (threadIdx.x < 256) {
        const int q = threadIdx.x >> 2, i = threadIdx.x & 3;
        const int cc = (blockIdx.y * 64 + q) * 4 + i;
        float b = -1.f;
        int bk = 0;
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) {       // slices in row order, strict >: the first row attaining the max wins
            const float v = sv[s2][q][i];
            if (v > b || v != v) { b = v; bk = si[s2][q][i]; }
        }
        if (cc < C) {
            out[g * C + cc] = b;
            if (argmax) argmax[g * C + cc] = bk;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Max-pooled last layer, backward without its dense output y (papc_mlp_bwd_dx_max_f32):
//   dy = s*p - e*y + f            with s = scale, e = s*c2*invstd, f = e*mean - s*c1   (the BN+ReLU backward, expanded)
//   y  = A W^T + b                A = relu(bn(y_prev)) [M, Ci], W [Co, Ci]
//   dX = dy W = (s*p) W - A (W^T E W) + (f - e*b) W
// p is non-zero only at the argmax row of each (group, channel): psel[g, c] = s_c * [s_c*ysel + shift_c > 0] * gout[g, c].
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_max_psel_kernel(const float *__restrict__ gout, const float *__restrict__ ysel,
                                                          const float *__restrict__ scale, const float *__restrict__ shift, int64_t total,
                                                          int C, float *__restrict__ psel)
{
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        const float sc = scale[c];
        psel[e] = fmaf(sc, ysel[e], shift[c]) > 0.f ? sc * gout[e] : 0.f;
    }
}

// block n (< Ci): row n of wcat [Ci, Co + Ci] = [ W^T | -W^T E W ], hbias[n] = sum_c q_c W[c, n], q = f - e*b; block 0 also writes e, q
__global__ __launch_bounds__(256) void bn_max_wcat_kernel(const float *__restrict__ w, const float *__restrict__ bias,
                                                          const float *__restrict__ scale, const float *__restrict__ mean,
                                                          const float *__restrict__ invstd, const float *__restrict__ c1,
                                                          const float *__restrict__ c2, int Co, int Ci, float *__restrict__ wcat,
                                                          float *__restrict__ hbias, float *__restrict__ e_out, float *__restrict__ q_out)
{
    extern __shared__ float sm[];          // [Co] e_c * W[c, n]   then [256] reduction scratch (doubles)
    float *ewn = sm;
    double *red = reinterpret_cast<double *>(sm + ((Co + 1) & ~1));
    const int n = blockIdx.x, tid = threadIdx.x;
    double hacc = 0.0;
    for (int c = tid; c < Co; c += 256) {
        const float s = scale[c];
        const float e = s * c2[c] * invstd[c];
        const float q = e * (mean[c] - (bias ? bias[c] : 0.f)) - s * c1[c];
        const float wv = w[(int64_t)c * Ci + n];
        ewn[c] = e * wv;
        hacc += (double)q * (double)wv;
        wcat[(int64_t)n * (Co + Ci) + c] = wv;
        if (n == 0) { e_out[c] = e; q_out[c] = q; }
    }
    red[tid] = hacc;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (tid < st) red[tid] += red[tid + st];
        __syncthreads();
    }
    if (tid == 0) hbias[n] = (float)red[0];
    for (int j = tid; j < Ci; j += 256) {
        double g = 0.0;
        for (int c = 0; c < Co; ++c) g += (double)ewn[c] * (double)w[(int64_t)c * Ci + j];
        wcat[(int64_t)n * (Co + Ci) + Co + j] = (float)(-g);
    }
}

// up to 8 small row-major matrices transposed in one launch (the W^T operands of a stack's dX GEMMs)
struct TransposeBatch {
    const float *src[8];
    float *dst[8];
    int rows[8], cols[8];
    int ld[8], copy[8];      // source row stride; copy != 0: dst [rows, cols] = the block itself (a column slice made contiguous), not its transpose
};

__global__ __launch_bounds__(256) void transpose_batch_kernel(TransposeBatch t)
{
    __shared__ float tile[32][33];
    const int m = blockIdx.y;
    const int rows = t.rows[m], cols = t.cols[m];
    const int tiles_c = (cols + 31) >> 5, tiles = ((rows + 31) >> 5) * tiles_c;
    const float *__restrict__ src = t.src[m];
    float *__restrict__ dst = t.dst[m];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int tl = blockIdx.x; tl < tiles; tl += gridDim.x) {
        const int r0 = (tl / tiles_c) * 32, c0 = (tl % tiles_c) * 32;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = r0 + ty + 8 * i, c = c0 + tx;
            const float v = (r < rows && c < cols) ? src[(int64_t)r * t.ld[m] + c] : 0.f;
            tile[ty + 8 * i][tx] = v;
            if (t.copy[m] && r < rows && c < cols) dst[(int64_t)r * cols + c] = v;
        }
        if (t.copy[m]) continue;          // (uniform per job)
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = c0 + ty + 8 * i, r = r0 + tx;
            if (r < rows && c < cols) dst[(int64_t)c * rows + r] = tile[tx][ty + 8 * i];
        }
        __syncthreads();
    }
}

// up to 8 partial reductions in one launch (blockIdx.y = job): the dW partials of all layers of a stack are folded together at the end
// of its backward instead of one launch-latency-sized kernel per layer
struct ReduceBatch {
    const float *part[8]; float *out1[8], *out2[8];
    int64_t ld[8], n1[8], n[8];
    int n_chunks[8], accumulate[8];
};
__global__ __launch_bounds__(1024) void reduce_partials_batch_kernel(ReduceBatch b)
{
    __shared__ float red[16][64];
    const int job = blockIdx.y;
    const float *__restrict__ part = b.part[job];
    const int64_t n = b.n[job], ld = b.ld[job];
    const int n_chunks = b.n_chunks[job];
    if ((int64_t)blockIdx.x * 64 >= n) return;               // (uniform per workgroup: this job is narrower than the widest one)
    const int el = threadIdx.x & 63, cl = threadIdx.x >> 6;  // lane = element (coalesced), wave = chunk lane
    const int64_t i = (int64_t)blockIdx.x * 64 + el;
    float s = 0.f;
    if (i < n) {
        for (int t0 = cl; t0 < n_chunks; t0 += 16 * 8) {     // 8 loads in flight per lane; summed in chunk order
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int t = t0 + 16 * j;
                v[j] = part[(int64_t)(t < n_chunks ? t : t0) * ld + i];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) s += (t0 + 16 * j < n_chunks) ? v[j] : 0.f;
        }
    }
    red[cl][el] = s;
    __syncthreads();
    if (cl == 0 && i < n) {
#pragma unroll
        for (int g = 1; g < 16; ++g) s += red[g][el];
        float *o = i < b.n1[job] ? b.out1[job] + i : b.out2[job] + (i - b.n1[job]);
        *o = b.accumulate[job] ? *o + s : s;
    }
}

// ---- deferred folds: every partial reduction of a training step's backward in ONE launch (papc_fold_jobs_f32) ---------------------------
// Flat grid: workgroup b belongs to the job whose [wg0, wg0 + nwg) range holds it.  Two shapes, uniform per workgroup:
//   many chunks (kind 0): 64 elements x 16 chunk lanes, 8 loads in flight per lane, chunk lanes folded in lane order -- the summation
//                         order of reduce_partials_batch_kernel, so a deferred fold equals the stack's own launch bit for bit;
//   few chunks  (kind 1): a thread owns a float4 of 4096 consecutive elements per workgroup and walks the chunks in order -- the order of
//                         pg_fold_kernel (smallm.hip): the split-K partials of the planes path (<= 8 chunks of up to 512 K elements).
struct FoldBatch {
    const float *part[PAPC_FOLD_MAX]; float *out[PAPC_FOLD_MAX];
    int64_t ld[PAPC_FOLD_MAX], out_ld[PAPC_FOLD_MAX];
    int n_chunks[PAPC_FOLD_MAX], rows[PAPC_FOLD_MAX], cols[PAPC_FOLD_MAX], wg0[PAPC_FOLD_MAX + 1];
    unsigned acc_mask, wide_mask, vec_mask;
    int count;
};
// FW = chunk lanes (waves) per workgroup (PAPC_FOLD_WAVES).  Round 6 tried 8 instead of 16 -- the step's 654 workgroups of 1024 threads are two
// residency rounds (512 fit the chip), the second a quarter full; at 512 threads all are resident at once -- and measured it 7 us SLOWER per step
// (1.461 against 1.454 ms, same box, fixed plan): 16 stays the default.
template <int FW>
__global__ __launch_bounds__(64 * FW) void fold_jobs_kernel(FoldBatch b)
{
    __shared__ float red[FW][64];
    int job = 0;
    while (job + 1 < b.count && (int)blockIdx.x >= b.wg0[job + 1]) ++job;       // (<= 24 scalar compares)
    const int wg = (int)blockIdx.x - b.wg0[job];
    const float *__restrict__ part = b.part[job];
    float *__restrict__ out = b.out[job];
    const int64_t ld = b.ld[job], out_ld = b.out_ld[job];
    const int n_chunks = b.n_chunks[job], cols = b.cols[job];
    const int64_t n = (int64_t)b.rows[job] * cols;
    const bool acc = (b.acc_mask >> job) & 1u;
    if ((b.wide_mask >> job) & 1u) {        // few chunks, contiguous output (out_ld == cols), n % 4 == 0, 16-byte aligned
        const int64_t e = ((int64_t)wg * (64 * FW) + threadIdx.x) * 4;
        if (e >= n) return;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int t0 = 0; t0 < n_chunks; t0 += 8) {          // 8 loads in flight, summed in chunk order
            float4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const float4 *>(part + (int64_t)(t0 + j < n_chunks ? t0 + j : t0) * ld + e);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (t0 + j < n_chunks) { if (t0 + j == 0) s = v[j]; else { s.x += v[j].x; s.y += v[j].y; s.z += v[j].z; s.w += v[j].w; } }
        }
        float4 *o = reinterpret_cast<float4 *>(out + e);
        if (acc) { const float4 a = *o; s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w; }
        *o = s;
        return;
    }
    const int el = threadIdx.x & 63, cl = threadIdx.x >> 6;  // lane = element (coalesced), wave = chunk lane
    if ((b.vec_mask >> job) & 1u) {         // many chunks, float4 lanes (n % 4 == 0, contiguous output): the same order per element, 1 KB per wave and chunk
        __shared__ float4 red4[FW][64];
        const int64_t i4 = ((int64_t)wg * 64 + el) * 4;
        float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i4 < n) {
            for (int t0 = cl; t0 < n_chunks; t0 += FW * 8) {
                float4 v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int t = t0 + FW * j;
                    v[j] = *reinterpret_cast<const float4 *>(part + (int64_t)(t < n_chunks ? t : t0) * ld + i4);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (t0 + FW * j < n_chunks) { s4.x += v[j].x; s4.y += v[j].y; s4.z += v[j].z; s4.w += v[j].w; }
            }
        }
        red4[cl][el] = s4;
        __syncthreads();
        if (cl == 0 && i4 < n) {
#pragma unroll
            for (int g = 1; g < FW; ++g) { const float4 r = red4[g][el]; s4.x += r.x; s4.y += r.y; s4.z += r.z; s4.w += r.w; }
            float4 *o = reinterpret_cast<float4 *>(out + i4);
            if (acc) { const float4 a = *o; s4.x += a.x; s4.y += a.y; s4.z += a.z; s4.w += a.w; }
            *o = s4;
        }
        return;
    }
    const int64_t i = (int64_t)wg * 64 + el;
    float s = 0.f;
    if (i < n) {
        for (int t0 = cl; t0 < n_chunks; t0 += FW * 8) {     // 8 loads in flight per lane; summed in chunk order
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int t = t0 + FW * j;
                v[j] = part[(int64_t)(t < n_chunks ? t : t0) * ld + i];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) s += (t0 + FW * j < n_chunks) ? v[j] : 0.f;
        }
    }
    red[cl][el] = s;
    __syncthreads();
    if (cl == 0 && i < n) {
#pragma unroll
        for (int g = 1; g < FW; ++g) s += red[g][el];
        const int r = (int)(i / cols), c = (int)(i - (int64_t)r * cols);
        float *o = out + (int64_t)r * out_ld + c;
        *o = acc ? *o + s : s;
    }
}

}  // namespace papc


using namespace papc;

extern "C" {

int papc_bn_finalize_f32(const float *stats_partial, int n_tiles, int64_t M, int C, const float *gamma,
                         const float *beta, float eps, float momentum, float *mean, float *invstd,
                         float *scale, float *shift, float *running_mean, float *running_var,
                         papc_stream_t stream)
{
    PAPC_REQUIRE(stats_partial && mean && invstd && scale && shift, PAPC_E_INVALID, "papc_bn_finalize_f32: null pointer");
    PAPC_REQUIRE(n_tiles >= 1 && M >= 1 && C >= 1, PAPC_E_INVALID, "papc_bn_finalize_f32: bad sizes");
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_MISC, st);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)C), dim3(64), 0, st, stats_partial, n_tiles, M, C, gamma,
                       beta, eps, momentum, mean, invstd, scale, shift, running_mean, running_var);
    return check_launch("papc_bn_finalize_f32");
}

int papc_bn_relu_max_f32(const float *y, const float *scale, const float *shift, int64_t G, int K, int C,
                         float *out, int32_t *argmax, papc_stream_t stream)
{
    PAPC_REQUIRE(y && scale && shift && out, PAPC_E_INVALID, "papc_bn_relu_max_f32: null pointer");
    PAPC_REQUIRE(G >= 1 && K >= 1 && C >= 1, PAPC_E_INVALID, "papc_bn_relu_max_f32: bad sizes");
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_BN_RELU_MAX, st);
    const bool v4 = (C % 4 == 0) && aligned16(y) && aligned16(scale) && aligned16(shift) && aligned16(out);
    if (v4 && K >= 256 && G * (C / 4) < 65536 && G <= 65535) {   // few, long groups: split the rows over the workgroup
        hipLaunchKernelGGL(bn_relu_max_split_kernel, dim3((unsigned)G, (unsigned)cdiv(C, 256)), dim3(1024), 0, st, y, scale, shift, G, K, C, out, argmax);
        return check_launch("papc_bn_relu_max_f32");
    }
    if (v4) hipLaunchKernelGGL(bn_relu_max_kernel<4>, dim3(ew_grid(G * (C / 4))), dim3(256), 0, st, y, scale, shift, G, K, C, out, argmax);
    else hipLaunchKernelGGL(bn_relu_max_kernel<1>, dim3(ew_grid(G * C)), dim3(256), 0, st, y, scale, shift, G, K, C, out, argmax);
    return check_launch("papc_bn_relu_max_f32");
}

int papc_bn_select_max_f32(float *gmax, const float *gmin, const int32_t *amax, const int32_t *amin,
                           const float *scale, const float *shift, int64_t G, int C, float *out, int32_t *argmax,
                           papc_stream_t stream)
{
    PAPC_REQUIRE(gmax && gmin && amax && amin && scale && shift && out, PAPC_E_INVALID, "papc_bn_select_max_f32: null pointer");
    PAPC_REQUIRE(G >= 1 && C >= 1, PAPC_E_INVALID, "papc_bn_select_max_f32: bad sizes");
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_BN_RELU_MAX, st);
    hipLaunchKernelGGL(bn_select_max_kernel, dim3(ew_grid(G * C)), dim3(256), 0, st, gmax, gmin, amax, amin, scale, shift, G * C, C, out, argmax);
    return check_launch("papc_bn_select_max_f32");
}

int papc_bn_relu_f32(const float *y, const float *scale, const float *shift, int64_t M, int C, float *z, papc_stream_t stream)
{
    PAPC_REQUIRE(y && scale && shift && z, PAPC_E_INVALID, "papc_bn_relu_f32: null pointer");
    PAPC_REQUIRE(M >= 1 && C >= 1, PAPC_E_INVALID, "papc_bn_relu_f32: bad sizes");
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_MISC, st);
    const bool v4 = (C % 4 == 0) && aligned16(y) && aligned16(scale) && aligned16(shift) && aligned16(z);
    if (v4) hipLaunchKernelGGL(bn_relu_kernel<4>, dim3(ew_grid(M * (C / 4))), dim3(256), 0, st, y, scale, shift, M, C, z);
    else hipLaunchKernelGGL(bn_relu_kernel<1>, dim3(ew_grid(M * C)), dim3(256), 0, st, y, scale, shift, M, C, z);
    return check_launch("papc_bn_relu_f32");
}

int papc_bn_bwd_reduce_f32(int dz_mode, const float *dz, const float *gout, const int32_t *argmax, int K,
                           const float *y, const float *mean, const float *invstd, const float *scale,
                           const float *shift, int64_t M, int C, int n_parts, float *red_partial, papc_stream_t stream)
{
    PAPC_REQUIRE(mean && invstd && scale && shift && red_partial, PAPC_E_INVALID, "papc_bn_bwd_reduce_f32: null pointer");
    PAPC_REQUIRE(y || (dz_mode != PAPC_DZ_DENSE && dz), PAPC_E_INVALID, "papc_bn_bwd_reduce_f32: y may be NULL only in MAX mode with dz = ysel");
    PAPC_REQUIRE(M >= 1 && C >= 1 && n_parts >= 1, PAPC_E_INVALID, "papc_bn_bwd_reduce_f32: bad sizes");
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_BWD_REDUCE, st);
    if (dz_mode == PAPC_DZ_DENSE) {
        PAPC_REQUIRE(dz, PAPC_E_INVALID, "papc_bn_bwd_reduce_f32: DENSE needs dz");
        const bool v4 = (C % 4 == 0) && aligned16(dz) && aligned16(y) && aligned16(mean) && aligned16(invstd) && aligned16(scale) && aligned16(shift);
        if (v4) hipLaunchKernelGGL((bn_bwd_reduce_kernel<4, false>), dim3(n_parts), dim3(256), 0, st, dz, gout, argmax, K, y, mean, invstd, scale, shift, M, C, red_partial);
        else hipLaunchKernelGGL((bn_bwd_reduce_kernel<1, false>), dim3(n_parts), dim3(256), 0, st, dz, gout, argmax, K, y, mean, invstd, scale, shift, M, C, red_partial);
    } else {
        PAPC_REQUIRE(gout && argmax && K >= 1 && M % K == 0, PAPC_E_INVALID, "papc_bn_bwd_reduce_f32: MAX needs gout/argmax and K | M");
        const bool v4 = (C % 4 == 0) && aligned16(gout) && (!dz || aligned16(dz)) && aligned16(mean) && aligned16(invstd) && aligned16(scale) && aligned16(shift);
        if (v4) hipLaunchKernelGGL((bn_bwd_reduce_kernel<4, true>), dim3(n_parts), dim3(256), 0, st, dz, gout, argmax, K, y, mean, invstd, scale, shift, M, C, red_partial);
        else hipLaunchKernelGGL((bn_bwd_reduce_kernel<1, true>), dim3(n_parts), dim3(256), 0, st, dz, gout, argmax, K, y, mean, invstd, scale, shift, M, C, red_partial);
    }
    return check_launch("papc_bn_bwd_reduce_f32");
}

int papc_bn_bwd_reduce_max_f32(const float *ysel, const float *gout, int K, const float *mean, const float *invstd, const float *scale,
                               const float *shift, int64_t M, int C, int n_parts, float *red_partial, float *psel, papc_stream_t stream)
{
    PAPC_REQUIRE(ysel && gout && mean && invstd && scale && shift && red_partial, PAPC_E_INVALID, "papc_bn_bwd_reduce_max_f32: null pointer");
    PAPC_REQUIRE(M >= 1 && C >= 1 && n_parts >= 1 && K >= 1 && M % K == 0, PAPC_E_INVALID, "papc_bn_bwd_reduce_max_f32: bad sizes");
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_BWD_REDUCE, st);
    const bool v4 = (C % 4 == 0) && aligned16(gout) && aligned16(ysel) && (!psel || aligned16(psel)) && aligned16(mean) && aligned16(invstd) && aligned16(scale) && aligned16(shift);
    if (v4) hipLaunchKernelGGL((bn_bwd_reduce_kernel<4, true>), dim3(n_parts), dim3(256), 0, st, ysel, gout, (const int32_t *)nullptr, K, (const float *)nullptr, mean, invstd, scale, shift, M, C, red_partial, psel);
    else hipLaunchKernelGGL((bn_bwd_reduce_kernel<1, true>), dim3(n_parts), dim3(256), 0, st, ysel, gout, (const int32_t *)nullptr, K, (const float *)nullptr, mean, invstd, scale, shift, M, C, red_partial, psel);
    return check_launch("papc_bn_bwd_reduce_max_f32");
}

int papc_bn_bwd_finalize_f32(const float *red_partial, int n_tiles, int64_t M, int C, float *dgamma,
                             float *dbeta, float *c1, float *c2, int accumulate, papc_stream_t stream)
{
    PAPC_REQUIRE(red_partial && c1 && c2, PAPC_E_INVALID, "papc_bn_bwd_finalize_f32: null pointer");
    PAPC_REQUIRE(n_tiles >= 1 && M >= 1 && C >= 1, PAPC_E_INVALID, "papc_bn_bwd_finalize_f32: bad sizes");
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_MISC, st);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((unsigned)C), dim3(64), 0, st, red_partial, n_tiles, M, C, dgamma, dbeta, c1, c2, accumulate);
    return check_launch("papc_bn_bwd_finalize_f32");
}

int papc_bn_eval_consts_f32(const float *running_mean, const float *running_var, const float *gamma, const float *beta, float eps, int C,
                            float *mean, float *invstd, float *scale, float *shift, papc_stream_t stream)
{
    PAPC_REQUIRE(running_mean && running_var && mean && invstd && scale && shift, PAPC_E_INVALID, "papc_bn_eval_consts_f32: null pointer");
    PAPC_REQUIRE(C >= 1, PAPC_E_INVALID, "papc_bn_eval_consts_f32: C=%d", C);
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_MISC, st);
    hipLaunchKernelGGL(bn_eval_consts_kernel, dim3((unsigned)cdiv(C, 256)), dim3(256), 0, st, running_mean, running_var, gamma, beta, eps, C,
                       mean, invstd, scale, shift);
    return check_launch("papc_bn_eval_consts_f32");
}

int papc_group_max_bwd_f32(const float *gout, const int32_t *argmax, int64_t G, int K, int C, float *dx, papc_stream_t stream)
{
    PAPC_REQUIRE(gout && argmax && dx, PAPC_E_INVALID, "papc_group_max_bwd_f32: null pointer");
    PAPC_REQUIRE(G >= 1 && K >= 1 && C >= 1, PAPC_E_INVALID, "papc_group_max_bwd_f32: bad sizes");
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_MISC, st);
    const int64_t total = G * K * (int64_t)C;
    hipLaunchKernelGGL(group_max_bwd_kernel, dim3((unsigned)std::min<int64_t>(cdiv(total, 256), 8192)), dim3(256), 0, st, gout, argmax, G, K, C, dx);
    return check_launch("papc_group_max_bwd_f32");
}

int papc_reduce_partials2_f32(const float *partial, int n_chunks, int64_t ld, int64_t n1, float *out1, int64_t n2,
                              float *out2, int accumulate, papc_stream_t stream)
{
    PAPC_REQUIRE(partial && out1 && (n2 == 0 || out2), PAPC_E_INVALID, "papc_reduce_partials2_f32: null pointer");
    PAPC_REQUIRE(n_chunks >= 1 && n1 >= 1 && n2 >= 0 && ld >= n1 + n2, PAPC_E_INVALID, "papc_reduce_partials2_f32: bad sizes");
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_MISC, st);
    const int64_t n = n1 + n2;
    const bool wide = n_chunks <= 64 && n >= 16384 && n1 % 4 == 0 && n2 % 4 == 0 && ld % 4 == 0 && aligned16(partial) && aligned16(out1) &&
                      (n2 == 0 || aligned16(out2));
    if (wide)
        hipLaunchKernelGGL(reduce_partials_wide_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, partial, n_chunks, n, ld, out1, n1, out2, accumulate);
    else
        hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)cdiv(n, 64)), dim3(1024), 0, st, partial, n_chunks, n, ld, out1, n1, out2, accumulate);
    return check_launch("papc_reduce_partials2_f32");
}

int papc_reduce_partials_batch_f32(const papc_reduce_job *jobs, int count, papc_stream_t stream)
{
    PAPC_REQUIRE(jobs, PAPC_E_INVALID, "papc_reduce_partials_batch_f32: null jobs");
    PAPC_REQUIRE(count >= 1 && count <= 8, PAPC_E_INVALID, "papc_reduce_partials_batch_f32: count=%d not in [1, 8]", count);
    ReduceBatch b;
    memset(&b, 0, sizeof(b));
    int64_t nmax = 0;
    for (int i = 0; i < count; ++i) {
        const papc_reduce_job &j = jobs[i];
        PAPC_REQUIRE(j.partial && j.out1 && (j.n2 == 0 || j.out2), PAPC_E_INVALID, "papc_reduce_partials_batch_f32: null pointer in job %d", i);
        PAPC_REQUIRE(j.n_chunks >= 1 && j.n1 >= 1 && j.n2 >= 0 && j.ld >= j.n1 + j.n2, PAPC_E_INVALID, "papc_reduce_partials_batch_f32: bad sizes in job %d", i);
        b.part[i] = j.partial; b.out1[i] = j.out1; b.out2[i] = j.out2; b.ld[i] = j.ld; b.n1[i] = j.n1; b.n[i] = j.n1 + j.n2;
        b.n_chunks[i] = j.n_chunks; b.accumulate[i] = j.accumulate;
        nmax = std::max(nmax, j.n1 + j.n2);
    }
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_MISC, st);
    hipLaunchKernelGGL(reduce_partials_batch_kernel, dim3((unsigned)cdiv(nmax, 64), (unsigned)count), dim3(1024), 0, st, b);
    return check_launch("papc_reduce_partials_batch_f32");
}

int papc_fold_jobs_f32(const papc_fold_job *jobs, int count, papc_stream_t stream)
{
    PAPC_REQUIRE(jobs, PAPC_E_INVALID, "papc_fold_jobs_f32: null jobs");
    PAPC_REQUIRE(count >= 1, PAPC_E_INVALID, "papc_fold_jobs_f32: count=%d", count);
    hipStream_t st = as_stream(stream);
    const int FW = knob(KNOB_FOLD_WAVES) == 16 ? 16 : 8;
    for (int j0 = 0; j0 < count; j0 += PAPC_FOLD_MAX) {
        const int nj = std::min(PAPC_FOLD_MAX, count - j0);
        FoldBatch b;
        memset(&b, 0, sizeof(b));
        int64_t wgs = 0;
        for (int i = 0; i < nj; ++i) {
            const papc_fold_job &j = jobs[j0 + i];
            PAPC_REQUIRE(j.partial && j.out, PAPC_E_INVALID, "papc_fold_jobs_f32: null pointer in job %d", j0 + i);
            PAPC_REQUIRE(j.n_chunks >= 1 && j.rows >= 1 && j.cols >= 1 && j.ld >= (int64_t)j.rows * j.cols && j.out_ld >= j.cols, PAPC_E_INVALID,
                         "papc_fold_jobs_f32: bad sizes in job %d", j0 + i);
            const int64_t n = (int64_t)j.rows * j.cols;
            const bool wide = j.n_chunks <= 16 && n >= 16384 && n % 4 == 0 && j.ld % 4 == 0 && (j.rows == 1 || j.out_ld == j.cols) && aligned16(j.partial) && aligned16(j.out);
            b.part[i] = j.partial; b.out[i] = j.out; b.ld[i] = j.ld; b.out_ld[i] = j.out_ld; b.n_chunks[i] = j.n_chunks; b.rows[i] = j.rows; b.cols[i] = j.cols;
            if (j.accumulate) b.acc_mask |= 1u << i;
            const bool vec = !wide && n >= 1024 && n % 4 == 0 && j.ld % 4 == 0 && (j.rows == 1 || j.out_ld == j.cols) && aligned16(j.partial) && aligned16(j.out);
            if (wide) b.wide_mask |= 1u << i;
            if (vec) b.vec_mask |= 1u << i;
            b.wg0[i] = (int)wgs;
            wgs += wide ? cdiv(n, 4 * 64 * FW) : (vec ? cdiv(n, 256) : cdiv(n, 64));
            PAPC_REQUIRE(wgs < (1ll << 30), PAPC_E_UNSUPPORTED, "papc_fold_jobs_f32: too many elements");
        }
        b.wg0[nj] = (int)wgs;
        b.count = nj;
        ProfScope prof(PAPC_K_BWD_DW, st);
        if (FW == 8) hipLaunchKernelGGL(fold_jobs_kernel<8>, dim3((unsigned)wgs), dim3(512), 0, st, b);
        else hipLaunchKernelGGL(fold_jobs_kernel<16>, dim3((unsigned)wgs), dim3(1024), 0, st, b);
        const int rc = check_launch("papc_fold_jobs_f32");
        if (rc != PAPC_OK) return rc;
    }
    return PAPC_OK;
}

int papc_reduce_partials_f32(const float *partial, int n_chunks, int64_t n, float *out, int accumulate, papc_stream_t stream)
{
    PAPC_REQUIRE(partial && out, PAPC_E_INVALID, "papc_reduce_partials_f32: null pointer");
    PAPC_REQUIRE(n_chunks >= 1 && n >= 1, PAPC_E_INVALID, "papc_reduce_partials_f32: bad sizes");
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_MISC, st);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)cdiv(n, 64)), dim3(1024), 0, st, partial, n_chunks, n, n, out, n, (float *)nullptr, accumulate);
    return check_launch("papc_reduce_partials_f32");
}

int papc_fill_f32(float *p, int64_t n, float value, papc_stream_t stream)
{
    PAPC_REQUIRE(p && n >= 1, PAPC_E_INVALID, "papc_fill_f32: null pointer or n < 1");
    PAPC_REQUIRE(aligned16(p), PAPC_E_INVALID, "papc_fill_f32: p must be 16-byte aligned");
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_MISC, st);
    hipLaunchKernelGGL(fill_kernel, dim3(ew_grid(n / 4 + 1)), dim3(256), 0, st, p, n, value);
    return check_launch("papc_fill_f32");
}

int papc_copy2d_f32(const float *src, int64_t src_ld, float *dst, int64_t dst_ld, int rows, int cols, int transpose, papc_stream_t stream)
{
    PAPC_REQUIRE(src && dst && rows >= 1 && cols >= 1, PAPC_E_INVALID, "papc_copy2d_f32: null pointer or empty block");
    PAPC_REQUIRE(src_ld >= cols && dst_ld >= (transpose ? rows : cols), PAPC_E_INVALID, "papc_copy2d_f32: row stride shorter than a row");
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_MISC, st);
    hipLaunchKernelGGL(copy2d_kernel, dim3(ew_grid((int64_t)rows * cols)), dim3(256), 0, st, src, src_ld, dst, dst_ld, rows, cols, transpose);
    return check_launch("papc_copy2d_f32");
}

int papc_copy_strided_batch_f32(const papc_copy_job *jobs, int count, papc_stream_t stream)
{
    PAPC_REQUIRE(jobs, PAPC_E_INVALID, "papc_copy_strided_batch_f32: null jobs");
    PAPC_REQUIRE(count >= 1 && count <= 8, PAPC_E_INVALID, "papc_copy_strided_batch_f32: count=%d not in [1, 8]", count);
    CopyJobs j;
    memset(&j, 0, sizeof(j));
    int64_t tmax = 0;
    for (int i = 0; i < count; ++i) {
        const papc_copy_job &c = jobs[i];
        PAPC_REQUIRE(c.src && c.dst && c.B >= 1 && c.R >= 1 && c.C >= 1, PAPC_E_INVALID, "papc_copy_strided_batch_f32: null pointer or empty job %d", i);
        j.src[i] = c.src; j.dst[i] = c.dst; j.B[i] = c.B; j.R[i] = c.R; j.C[i] = c.C;
        j.sb[i] = c.sb; j.sr[i] = c.sr; j.sc[i] = c.sc; j.db[i] = c.db; j.dr[i] = c.dr; j.dc[i] = c.dc;
        tmax = std::max(tmax, (int64_t)c.B * cdiv(c.R, 32) * cdiv(c.C, 32));
    }
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_MISC, st);
    hipLaunchKernelGGL(copy_strided_batch_kernel, dim3((unsigned)std::min<int64_t>(tmax, 4096), (unsigned)count), dim3(256), 0, st, j);
    return check_launch("papc_copy_strided_batch_f32");
}

int papc_reduce_partials_strided_f32(const float *partial, int n_chunks, int64_t ld, int rows, int cols, float *out, int64_t out_ld,
                                     int accumulate, papc_stream_t stream)
{
    PAPC_REQUIRE(partial && out, PAPC_E_INVALID, "papc_reduce_partials_strided_f32: null pointer");
    PAPC_REQUIRE(n_chunks >= 1 && rows >= 1 && cols >= 1 && ld >= (int64_t)rows * cols && out_ld >= cols, PAPC_E_INVALID,
                 "papc_reduce_partials_strided_f32: bad sizes");
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_MISC, st);
    hipLaunchKernelGGL(reduce_partials_strided_kernel, dim3((unsigned)cdiv((int64_t)rows * cols, 16)), dim3(1024), 0, st, partial, n_chunks, ld,
                       rows, cols, out, out_ld, accumulate);
    return check_launch("papc_reduce_partials_strided_f32");
}

int papc_scale_by_f32(const float *x, const float *scalar, int64_t n, float *out, papc_stream_t stream)
{
    PAPC_REQUIRE(x && scalar && out && n >= 1, PAPC_E_INVALID, "papc_scale_by_f32: null pointer or n < 1");
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_MISC, st);
    hipLaunchKernelGGL(scale_by_kernel, dim3(ew_grid(n)), dim3(256), 0, st, x, scalar, n, out);
    return check_launch("papc_scale_by_f32");
}

int papc_bn_max_prep_f32(const float *gout, const float *ysel, const float *scale, const float *shift, const float *mean,
                         const float *invstd, const float *c1, const float *c2, const float *w, const float *bias, int64_t G, int Co,
                         int Ci, float *psel, float *wcat, float *hbias, float *e_out, float *q_out, papc_stream_t stream)
{
    PAPC_REQUIRE(scale && shift && mean && invstd && c1 && c2 && w && wcat && hbias && e_out && q_out && (!psel || (gout && ysel)), PAPC_E_INVALID,
                 "papc_bn_max_prep_f32: null pointer");
    PAPC_REQUIRE(G >= 1 && Co >= 1 && Ci >= 1 && Co <= 8192, PAPC_E_INVALID, "papc_bn_max_prep_f32: bad sizes");
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_BWD_REDUCE, st);
    if (psel) hipLaunchKernelGGL(bn_max_psel_kernel, dim3(ew_grid(G * Co)), dim3(256), 0, st, gout, ysel, scale, shift, G * Co, Co, psel);   // (NULL: papc_bn_bwd_reduce_max_f32 wrote it)
    const size_t lds = (size_t)((Co + 1) & ~1) * sizeof(float) + 256 * sizeof(double);
    hipLaunchKernelGGL(bn_max_wcat_kernel, dim3((unsigned)Ci), dim3(256), lds, st, w, bias, scale, mean, invstd, c1, c2, Co, Ci, wcat, hbias, e_out, q_out);
    return check_launch("papc_bn_max_prep_f32");
}

int papc_transpose_batch_f32(const float *const *src, float *const *dst, const int *rows, const int *cols, int count, papc_stream_t stream)
{
    return papc_transpose_batch_ld_f32(src, nullptr, dst, rows, cols, nullptr, count, stream);
}

int papc_transpose_batch_ld_f32(const float *const *src, const int *src_ld, float *const *dst, const int *rows, const int *cols, const int *copy, int count,
                                papc_stream_t stream)
{
    PAPC_REQUIRE(src && dst && rows && cols, PAPC_E_INVALID, "papc_transpose_batch_f32: null pointer");
    PAPC_REQUIRE(count >= 1 && count <= 8, PAPC_E_INVALID, "papc_transpose_batch_f32: count=%d not in [1, 8]", count);
    TransposeBatch t{};
    int max_tiles = 1;
    for (int i = 0; i < count; ++i) {
        PAPC_REQUIRE(src[i] && dst[i] && rows[i] >= 1 && cols[i] >= 1, PAPC_E_INVALID, "papc_transpose_batch_f32: bad entry %d", i);
        t.src[i] = src[i]; t.dst[i] = dst[i]; t.rows[i] = rows[i]; t.cols[i] = cols[i];
        t.ld[i] = src_ld ? src_ld[i] : cols[i]; t.copy[i] = copy ? copy[i] : 0;
        PAPC_REQUIRE(t.ld[i] >= cols[i], PAPC_E_INVALID, "papc_transpose_batch_f32: entry %d: row stride shorter than a row", i);
        max_tiles = std::max(max_tiles, (int)(cdiv(rows[i], 32) * cdiv(cols[i], 32)));
    }
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_MISC, st);
    hipLaunchKernelGGL(transpose_batch_kernel, dim3((unsigned)std::min(max_tiles, 512), (unsigned)count), dim3(256), 0, st, t);
    return check_launch("papc_transpose_batch_f32");
}

static int adam_step(float *param, float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, float lr, double beta1, double beta2, float eps,
                     float weight_decay, int step, float grad_scale, bool zero, papc_stream_t stream, const char *who)
{
    PAPC_REQUIRE(param && grad && exp_avg && exp_avg_sq, PAPC_E_INVALID, "%s: null pointer", who);
    PAPC_REQUIRE(n >= 1 && step >= 1, PAPC_E_INVALID, "%s: bad n/step", who);
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_MISC, st);
    // everything that depends on the betas only is formed in double here (the optimiser's hyper-parameters are doubles on the host)
    const float bc1 = (float)(1.0 - pow(beta1, (double)step)), bc2 = (float)(1.0 - pow(beta2, (double)step));
    if (zero)
        hipLaunchKernelGGL(adam_kernel<true>, dim3(ew_grid(n)), dim3(256), 0, st, param, grad, exp_avg, exp_avg_sq, n, lr, (float)beta1, (float)beta2,
                           (float)(1.0 - beta1), (float)(1.0 - beta2), eps, weight_decay, bc1, bc2, grad_scale);
    else
        hipLaunchKernelGGL(adam_kernel<false>, dim3(ew_grid(n)), dim3(256), 0, st, param, grad, exp_avg, exp_avg_sq, n, lr, (float)beta1, (float)beta2,
                           (float)(1.0 - beta1), (float)(1.0 - beta2), eps, weight_decay, bc1, bc2, grad_scale);
    return check_launch(who);
}

int papc_adam_step_f32(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n,
                       float lr, double beta1, double beta2, float eps, float weight_decay, int step,
                       float grad_scale, papc_stream_t stream)
{
    return adam_step(param, const_cast<float *>(grad), exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, false, stream, "papc_adam_step_f32");
}

int papc_adam_tick(int64_t *step_dev, papc_stream_t stream)
{
    PAPC_REQUIRE(step_dev, PAPC_E_INVALID, "papc_adam_tick: null pointer");
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_MISC, st);
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, st, step_dev);
    return check_launch("papc_adam_tick");
}

int papc_flag_set(uint32_t *flag, uint32_t value, int64_t *counter, papc_stream_t stream)
{
    PAPC_REQUIRE(flag, PAPC_E_INVALID, "papc_flag_set: null pointer");
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(flag_set_kernel, dim3(1), dim3(1), 0, st, flag, value, counter);
    return check_launch("papc_flag_set");
}

int papc_flag_wait(uint32_t *flag, int64_t max_spins, papc_stream_t stream)
{
    PAPC_REQUIRE(flag && max_spins >= 0, PAPC_E_INVALID, "papc_flag_wait: null pointer / negative max_spins");
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(flag_wait_kernel, dim3(1), dim3(64), 0, st, flag, (long long)max_spins, 1);
    return check_launch("papc_flag_wait");
}

int papc_flag_wait_slot(uint32_t *flag, int slot, int64_t max_spins, papc_stream_t stream)
{
    PAPC_REQUIRE(flag && max_spins >= 0 && (slot == 0 || slot == 1), PAPC_E_INVALID, "papc_flag_wait_slot: null pointer / negative max_spins / slot not in {0, 1}");
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(flag_wait_kernel, dim3(1), dim3(64), 0, st, flag, (long long)max_spins, slot == 0 ? 1 : 3);
    return check_launch("papc_flag_wait_slot");
}

int papc_adam_step_dev_f32(float *param, float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, float lr, double beta1, double beta2,
                           float eps, float weight_decay, const int64_t *step_dev, float grad_scale, int zero_grad, papc_stream_t stream)
{
    PAPC_REQUIRE(param && grad && exp_avg && exp_avg_sq && step_dev, PAPC_E_INVALID, "papc_adam_step_dev_f32: null pointer");
    PAPC_REQUIRE(n >= 1, PAPC_E_INVALID, "papc_adam_step_dev_f32: n=%lld", (long long)n);
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_MISC, st);
    int64_t *sd = const_cast<int64_t *>(step_dev);
    const bool zero = (zero_grad & 1) != 0, tick = (zero_grad & 2) != 0;
    const bool v4 = n % 4 == 0 && aligned16(param) && aligned16(grad) && aligned16(exp_avg) && aligned16(exp_avg_sq);
#define ADAM_DEV(Z, T) do { if (v4) hipLaunchKernelGGL((adam_dev_kernel<Z, T, 4>), dim3(ew_grid(n / 4)), dim3(256), 0, st, param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, sd, grad_scale); \
                            else hipLaunchKernelGGL((adam_dev_kernel<Z, T, 1>), dim3(ew_grid(n)), dim3(256), 0, st, param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, sd, grad_scale); } while (0)
    if (zero && tick) ADAM_DEV(true, true);
    else if (zero) ADAM_DEV(true, false);
    else if (tick) ADAM_DEV(false, true);
    else ADAM_DEV(false, false);
#undef ADAM_DEV
    return check_launch("papc_adam_step_dev_f32");
}

int papc_adam_step_zero_f32(float *param, float *grad, float *exp_avg, float *exp_avg_sq, int64_t n,
                            float lr, double beta1, double beta2, float eps, float weight_decay, int step,
                            float grad_scale, papc_stream_t stream)
{
    return adam_step(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, true, stream, "papc_adam_step_zero_f32");
}

}  // extern "C"
