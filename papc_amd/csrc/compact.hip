// compact.hip -- the compacted form of a grouped stack: distinct neighbours only (gfx950).
//
// query_ball_point (PAPC/models/layers/pointnet2_basic_layers.py:98-126) pads every neighbourhood to nsample slots with copies of its first
// hit (:118-124).  Copies are IDENTICAL rows through every layer of the stack that consumes them (:214-217): they change nothing in the max
// over the neighbourhood (:219) and enter the train-mode BatchNorm statistics and every backward sum only through their multiplicity.  On
// the benchmark's clouds 49 % of SA2's rows (nsample = 64, radius 0.4, 512 points) are such copies.  The compacted stack computes the same
// function with the same gradients on the distinct rows:
//
//   * layout: group g owns the rows [start[g], start[g + 1]) -- its distinct neighbours in the ball query's (ascending) order, then copies
//     of the first one up to a multiple of 8 rows (the MFMA kernels hand a lane 8 consecutive rows: a lane never straddles two groups).
//     The last group also takes the rows that round the total up to a multiple of 128 (whole tiles everywhere).  The row count lives in
//     DEVICE memory (rows[0]); kernels size their loops from it, so a captured step replays with whatever the next batch's clouds give.
//   * multiplicity: a physical row counts once; what is left of a group's nsample slots -- coef[g] = nsample - (rows of the group) -- is
//     carried by its first row.  Forward: sum_rows y + sum_g coef[g] y[start[g]] (and the same for y^2) are the statistics of the padded
//     tensor (bn_stats_corr_kernel appends the second sum as extra partial rows).  Backward: the gradient of a distinct row is the sum
//     over its copies, dy = s (p - w (c1 + xhat c2)) with w = wrow[row] = 1 + coef[g] on a group's first row and 1 elsewhere (p, the
//     max / ReLU gradient, reaches a copy set once: the first-maximum rule picks its first row); the BN-backward sums over the rows of
//     the summed gradient are the padded tensor's sums.
//   * the neighbourhood max runs over a group's physical rows (seg_max_kernel), argmax = absolute row.
//
// Everything here is integer / streaming work on [G] and [rows, C] arrays: HBM- and latency-bound, no matrix cores.
#include "common.h"

namespace papc {

// ---- plan: ball-query lists -> compact layout ------------------------------------------------------------------------------------------
// rows of group g, rounded up to 8: 1 + #(k >= 1 : idx[g, k] != idx[g, 0]).  The ball query emits ascending distinct indices and then
// pads with the first one, so an entry that repeats the first IS padding.  One wave per group.
__global__ __launch_bounds__(256) void compact_count_kernel(const int32_t *__restrict__ idx, int G, int K, int32_t *__restrict__ cnt8)
{
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (g >= G) return;
    const int32_t *row = idx + (int64_t)g * K;
    const int32_t first = row[0];
    int n = 0;
    for (int k = lane; k < K; k += 64) n += (k > 0 && row[k] != first) ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o);
    if (lane == 0) cnt8[g] = (n + 1 + 7) & ~7;
}

// start[g] = exclusive prefix sum of cnt8; rows[0] = total rounded up to 128, rows[1] = total.  One workgroup (G is a few thousand).
__global__ __launch_bounds__(1024) void compact_scan_kernel(const int32_t *__restrict__ cnt8, int G, int32_t *__restrict__ start,
                                                            int32_t *__restrict__ rows)
{
    __shared__ int part[1024];
    const int tid = threadIdx.x;
    const int per = (G + 1023) / 1024;
    const int g0 = tid * per, g1 = min(G, g0 + per);
    int s = 0;
    for (int g = g0; g < g1; ++g) s += cnt8[g];
    part[tid] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {          // Hillis-Steele inclusive scan of the per-thread sums
        const int v = tid >= o ? part[tid - o] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = tid ? part[tid - 1] : 0;
    for (int g = g0; g < g1; ++g) { start[g] = run; run += cnt8[g]; }
    if (tid == 1023) {
        const int total = part[1023];
        start[G] = total;
        rows[0] = (total + 127) & ~127;
        rows[1] = total;
    }
}

// per-row arrays.  One wave per group; the last group also writes the tail rows (copies of its first neighbour).
__global__ __launch_bounds__(256) void compact_fill_kernel(const int32_t *__restrict__ idx, int G, int K, const int32_t *__restrict__ start,
                                                           const int32_t *__restrict__ rows, int32_t *__restrict__ cidx,
                                                           int32_t *__restrict__ seg_grp, float *__restrict__ wrow, float *__restrict__ coef)
{
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (g >= G) return;
    const int s = start[g];
    int n = start[g + 1] - s;                      // <= K: the padded list already holds copies of the first hit behind the distinct ones
    const int32_t *row = idx + (int64_t)g * K;
    const int32_t first = row[0];
    const int extra = (g == G - 1) ? rows[0] - rows[1] : 0;
    const float cf = (float)(K - n - extra);
    for (int k = lane; k < n + extra; k += 64) {
        cidx[s + k] = k < n ? row[k] : first;
        wrow[s + k] = k == 0 ? 1.f + cf : 1.f;
        if ((k & 7) == 0) seg_grp[(s + k) >> 3] = g;
    }
    if (lane == 0) coef[g] = cf;
}

// ---- forward: what the copies add to the BatchNorm statistics ----------------------------------------------------------------------------
// out[r][0][c] = sum_{g in block r} coef[g] y[start[g], c], out[r][1][c] = ... y^2: extra rows of a layer's statistics partials.
// Thread = (row lane, channel quad): the few groups a thread owns are independent loads, all in flight at once; the row lanes of a block
// are folded through LDS in fixed order (deterministic).
__global__ __launch_bounds__(256) void bn_stats_corr_kernel(const float *__restrict__ y, int C, const int32_t *__restrict__ start,
                                                            const float *__restrict__ coef, int G, float *__restrict__ out)
{
    __shared__ float4 red[2][256];
    const int r = blockIdx.x, R = gridDim.x;
    const int CQ = C >> 2;                              // host-checked: C % 4 == 0, C <= 1024
    const int RL = 256 / CQ > 0 ? 256 / CQ : 1;         // row lanes per pass
    for (int c0 = 0; c0 < CQ; c0 += 256) {              // (C > 1024 channels would take several passes; one in practice)
        const int cq = c0 + (int)threadIdx.x % (CQ < 256 ? CQ : 256), rl = (int)threadIdx.x / (CQ < 256 ? CQ : 256);
        float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
        if (cq < CQ && rl < RL) {
            constexpr int U = 4;
            for (int g0 = r + R * rl; g0 < G; g0 += R * RL * U) {
                float4 v[U];
                float cf[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int g = g0 + u * R * RL;
                    const int gg = g < G ? g : g0;
                    cf[u] = g < G ? coef[gg] : 0.f;
                    v[u] = *reinterpret_cast<const float4 *>(y + (int64_t)start[gg] * C + 4 * cq);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    s1.x = fmaf(cf[u], v[u].x, s1.x); s1.y = fmaf(cf[u], v[u].y, s1.y); s1.z = fmaf(cf[u], v[u].z, s1.z); s1.w = fmaf(cf[u], v[u].w, s1.w);
                    s2.x = fmaf(cf[u] * v[u].x, v[u].x, s2.x); s2.y = fmaf(cf[u] * v[u].y, v[u].y, s2.y);
                    s2.z = fmaf(cf[u] * v[u].z, v[u].z, s2.z); s2.w = fmaf(cf[u] * v[u].w, v[u].w, s2.w);
                }
            }
        }
        red[0][threadIdx.x] = s1; red[1][threadIdx.x] = s2;
        __syncthreads();
        const int cqn = CQ < 256 ? CQ : 256;
        if ((int)threadIdx.x < cqn && c0 + (int)threadIdx.x < CQ) {
            float4 t1 = make_float4(0.f, 0.f, 0.f, 0.f), t2 = t1;
            for (int l = 0; l < RL; ++l) {
                const float4 a = red[0][l * cqn + threadIdx.x], b = red[1][l * cqn + threadIdx.x];
                t1.x += a.x; t1.y += a.y; t1.z += a.z; t1.w += a.w;
                t2.x += b.x; t2.y += b.y; t2.z += b.z; t2.w += b.w;
            }
            *reinterpret_cast<float4 *>(out + ((int64_t)r * 2 + 0) * C + 4 * (c0 + threadIdx.x)) = t1;
            *reinterpret_cast<float4 *>(out + ((int64_t)r * 2 + 1) * C + 4 * (c0 + threadIdx.x)) = t2;
        }
        __syncthreads();
    }
}

// ---- forward: neighbourhood max over ragged groups ------------------------------------------------------------------------------------------
// out[g, c] = max_rows relu(scale y + shift) = relu(scale (scale >= 0 ? max y : min y) + shift); argmax = the first row attaining it (absolute
// row index), ysel = the raw y there (the backward reductions read it instead of gathering y).  Thread = (group, channel quad).
__global__ __launch_bounds__(256) void seg_max_kernel(const float *__restrict__ y, int C, const int32_t *__restrict__ start,
                                                      const float *__restrict__ scale, const float *__restrict__ shift, int G,
                                                      float *__restrict__ out, int32_t *__restrict__ argmax, float *__restrict__ ysel)
{
    const int CQ = C >> 2;
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)G * CQ) return;
    const int g = (int)(e / CQ), c = (int)(e - (int64_t)g * CQ) * 4;
    const int r0 = start[g], r1 = start[g + 1];
    const float4 sc = *reinterpret_cast<const float4 *>(scale + c), sh = *reinterpret_cast<const float4 *>(shift + c);
    // one extremum per channel: the sign of scale is known here, so the minimum (scale < 0) is followed as the maximum of the value with its sign
    // bit flipped -- strict comparisons either way, the first row wins ties
    const float s4[4] = {sc.x, sc.y, sc.z, sc.w}, h4[4] = {sh.x, sh.y, sh.z, sh.w};
    int flip[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) flip[i] = s4[i] >= 0.f ? 0 : (int)0x80000000u;
    float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int ix[4] = {r0, r0, r0, r0};
    const float *p = y + (int64_t)r0 * C + c;
    int r = r0;
    for (; r + 4 <= r1; r += 4) {                   // (groups are multiples of 8 rows: four loads in flight per step; eight measured no faster)
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4 *>(p + (int64_t)u * C);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float a[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float k = __int_as_float(__float_as_int(a[i]) ^ flip[i]);
                if (k > mx[i]) { mx[i] = k; ix[i] = r + u; }
            }
        }
        p += (int64_t)4 * C;
    }
    for (; r < r1; ++r) {
        const float4 v = *reinterpret_cast<const float4 *>(p);
        const float a[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float k = __int_as_float(__float_as_int(a[i]) ^ flip[i]);
            if (k > mx[i]) { mx[i] = k; ix[i] = r; }
        }
        p += C;
    }
    float o[4], ys[4];
    int am[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ys[i] = __int_as_float(__float_as_int(mx[i]) ^ flip[i]);
        am[i] = ix[i];
        o[i] = relu_np(fmaf(s4[i], ys[i], h4[i]));      // (NaN statistics -> NaN out, like the padded form: bn_select_max)
    }
    const int64_t q = (int64_t)g * C + c;
    *reinterpret_cast<float4 *>(out + q) = make_float4(o[0], o[1], o[2], o[3]);
    *reinterpret_cast<int4 *>(argmax + q) = make_int4(am[0], am[1], am[2], am[3]);
    *reinterpret_cast<float4 *>(ysel + q) = make_float4(ys[0], ys[1], ys[2], ys[3]);
}

}  // namespace papc

using namespace papc;

extern "C" {

int papc_compact_plan_f32(const int32_t *idx, int G, int K, int32_t *cnt8, int32_t *start, int32_t *rows, int32_t *cidx, int32_t *seg_grp,
                          float *wrow, float *coef, papc_stream_t stream)
{
    PAPC_REQUIRE(idx && cnt8 && start && rows && cidx && seg_grp && wrow && coef, PAPC_E_INVALID, "papc_compact_plan_f32: null pointer");
    PAPC_REQUIRE(G >= 1 && K >= 8 && K % 8 == 0, PAPC_E_UNSUPPORTED, "papc_compact_plan_f32: G=%d, nsample=%d must be a multiple of 8", G, K);
    PAPC_REQUIRE((int64_t)G * K < (1ll << 31), PAPC_E_UNSUPPORTED, "papc_compact_plan_f32: >= 2^31 rows");
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_GROUP, st);
    const unsigned nb = (unsigned)cdiv(G, 4);
    hipLaunchKernelGGL(compact_count_kernel, dim3(nb), dim3(256), 0, st, idx, G, K, cnt8);
    hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(1024), 0, st, cnt8, G, start, rows);
    hipLaunchKernelGGL(compact_fill_kernel, dim3(nb), dim3(256), 0, st, idx, G, K, start, rows, cidx, seg_grp, wrow, coef);
    return check_launch("papc_compact_plan_f32");
}

int papc_compact_corr_parts(void) { return 64; }

int papc_bn_stats_corr_f32(const float *y, int C, const int32_t *start, const float *coef, int G, float *stats_rows, papc_stream_t stream)
{
    PAPC_REQUIRE(y && start && coef && stats_rows, PAPC_E_INVALID, "papc_bn_stats_corr_f32: null pointer");
    PAPC_REQUIRE(G >= 1 && C >= 4 && C % 4 == 0 && aligned16(y) && aligned16(stats_rows), PAPC_E_INVALID, "papc_bn_stats_corr_f32: C %% 4 == 0 and 16-byte aligned rows");
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_MISC, st);
    hipLaunchKernelGGL(bn_stats_corr_kernel, dim3((unsigned)papc_compact_corr_parts()), dim3(256), 0, st, y, C, start, coef, G, stats_rows);
    return check_launch("papc_bn_stats_corr_f32");
}

int papc_bn_relu_max_seg_f32(const float *y, int C, const int32_t *start, const float *scale, const float *shift, int G, float *out,
                             int32_t *argmax, float *ysel, papc_stream_t stream)
{
    PAPC_REQUIRE(y && start && scale && shift && out && argmax && ysel, PAPC_E_INVALID, "papc_bn_relu_max_seg_f32: null pointer");
    PAPC_REQUIRE(G >= 1 && C >= 4 && C % 4 == 0, PAPC_E_UNSUPPORTED, "papc_bn_relu_max_seg_f32: C=%d must be a multiple of 4", C);
    PAPC_REQUIRE(aligned16(y) && aligned16(scale) && aligned16(shift) && aligned16(out) && aligned16(argmax) && aligned16(ysel), PAPC_E_INVALID,
                 "papc_bn_relu_max_seg_f32: 16-byte alignment");
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_BN_RELU_MAX, st);
    hipLaunchKernelGGL(seg_max_kernel, dim3((unsigned)cdiv((int64_t)G * (C / 4), 256)), dim3(256), 0, st, y, C, start, scale, shift, G, out, argmax, ysel);
    return check_launch("papc_bn_relu_max_seg_f32");
}

}  // extern "C"
