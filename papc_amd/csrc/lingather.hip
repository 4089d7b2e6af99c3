// lingather.hip -- first layer of a grouped stack with the linear map taken BEFORE the gather, for gfx950.
//
// Reference: PAPC/models/layers/pointnet2_basic_layers.py:146-153 (grouped_xyz_norm / index_points(points, idx) / concat) feeding
// conv1 of the stack (:215-217).  A row of that layer's input is [ xyz_j - centre (3) | feats_j (D) ] for neighbour j of a group, so
//     y[m, :] = W_x (xyz_j - centre) + W_f feats_j + b  =  P[j, :] + W_x (xyz_j - centre) + b,     P = feats W_f^T  [B*N, C].
// W_f feats_j depends on the SOURCE POINT only: every point sits in S*K/N (16 for SA2 of the SSG classifier) neighbourhoods, so the
// D-wide product is computed once per point (a B*N-row library GEMM) instead of once per (group, neighbour) row, and the layer
// itself becomes a streaming gather-add: one P row (C floats, L2-resident table) + a rank-3 update per output row.
// Backward, with G[j, :] = sum over the rows m that gathered point j of dY[m, :]:
//     grad_feats = G W_f,   dW_f = G^T feats   (B*N-row GEMMs),   dW_x[c, t] = sum_m dY[m, c] (xyz_j - centre)[t]  (streamed here).
// Both kernels are HBM-bound on the [M, C] activation stream (write y | read dz, y).
#include "mlp_loaders.h"

namespace papc {

struct LinGatherArgs {
    const float *P;             // [B*N, C]
    const float *xyz; int64_t sb, sn, sc;
    const float *new_xyz;       // [B*S, 3]
    const int32_t *idx;         // [B*S*K]
    const float *w; int ldw, xcol0;   // layer weight [C, ldw]; xyz columns xcol0 .. xcol0+2
    const float *bias;          // [C] or null
    int B, N, S, K, C;
    int64_t M;
    float *y;                   // fwd: [M, C]
    float *stats;               // fwd: [gridDim.x][2][C] column sums / sums of squares of y
    // compacted stack (compact.hip; all NULL for a padded one): point index per physical row, group of every 8-row segment, the physical
    // row count in device memory (M then is the capacity); the backward's per-row multiplicity weight is d.wrow
    const int32_t *cidx, *seg_grp, *rows_dev;
    const float *wstat;         // fwd, compacted: the rows' multiplicity weights -> statistics of the padded tensor (sum w y, sum w y^2)
    DySrc d;                    // bwd: dz, y, BN constants (DENSE)
    float *G;                   // bwd: [B*N, C], pre-zeroed, atomically accumulated
    float *dwx;                 // bwd: [gridDim.x][C][3]
    int rows_per_wg;
};

constexpr int LG_T = 256;

struct LgRow {
    int j;            // source point (or -1: no-hit sentinel -> zero input row)
    int b;
    float dx, dy, dz; // xyz_j - centre
};

__device__ __forceinline__ LgRow lg_row(const LinGatherArgs &a, int64_t m)
{
    LgRow r;
    const int g = a.cidx ? a.seg_grp[m >> 3] : (int)(m / a.K);
    r.b = g / a.S;
    int j = a.cidx ? a.cidx[m] : a.idx[m];
    if (j < 0 || j >= a.N) { r.j = -1; r.dx = r.dy = r.dz = 0.f; return r; }
    r.j = j;
    const float *p = a.xyz + (int64_t)r.b * a.sb + (int64_t)j * a.sn;
    const float *c = a.new_xyz + (int64_t)g * 3;
    r.dx = p[0] - c[0]; r.dy = p[a.sc] - c[1]; r.dz = p[2 * a.sc] - c[2];      // grouped_xyz - new_xyz (:147)
    return r;
}

constexpr int LG_ROWS = 256;                    // rows a workgroup stages metadata for at a time

// y = P[j] + W_x d + b, per-workgroup column statistics; thread = (row slot, channel quad).  The per-row metadata (neighbour index, centred
// coordinates) is fetched ONCE per row by one thread and staged in LDS: the C/4 threads of a row would otherwise each repeat the same
// seven loads (the kernel was bound by memory instructions, as the backward one was)
__global__ __launch_bounds__(LG_T) void lingather_fwd_kernel(LinGatherArgs a)
{
    __shared__ float red[LG_T * 8];
    __shared__ int s_p[LG_ROWS];                 // b * N + j, or -1 for a no-hit row
    __shared__ float s_dx[LG_ROWS], s_dy[LG_ROWS], s_dz[LG_ROWS];
    __shared__ float s_wq[LG_ROWS];              // wstat: the row's multiplicity weight
    const bool wq_on = a.wstat != nullptr;
    const int tid = threadIdx.x;
    const int CQ = a.C >> 2, RSL = LG_T / CQ;
    const int cq = tid % CQ, slot = tid / CQ;
    const bool act = slot < RSL;
    const int c = cq * 4;
    float4 w0, w1, w2, bv = make_float4(0.f, 0.f, 0.f, 0.f);
    {
        const float *wp = a.w + (int64_t)c * a.ldw + a.xcol0;
        w0 = make_float4(wp[0], wp[a.ldw], wp[2 * a.ldw], wp[3 * a.ldw]);
        w1 = make_float4(wp[1], wp[a.ldw + 1], wp[2 * a.ldw + 1], wp[3 * a.ldw + 1]);
        w2 = make_float4(wp[2], wp[a.ldw + 2], wp[2 * a.ldw + 2], wp[3 * a.ldw + 2]);
        if (a.bias) bv = ld4(a.bias + c);
    }
    int64_t rows_all = a.M, rpw = a.rows_per_wg;
    if (a.rows_dev) { rows_all = *a.rows_dev; rpw = (rows_all + gridDim.x - 1) / gridDim.x; }
    const int64_t mbeg = (int64_t)blockIdx.x * rpw, mend = min(rows_all, mbeg + rpw);
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    constexpr int U = 4;
    for (int64_t base = mbeg; base < mend; base += LG_ROWS) {
        const int nrows = (int)min((int64_t)LG_ROWS, mend - base);
        __syncthreads();
        for (int rr = tid; rr < nrows; rr += LG_T) {
            const LgRow r = lg_row(a, base + rr);
            s_p[rr] = r.j < 0 ? -1 : r.b * a.N + r.j;
            s_dx[rr] = r.dx; s_dy[rr] = r.dy; s_dz[rr] = r.dz;
            if (wq_on) s_wq[rr] = a.wstat[base + rr];
        }
        __syncthreads();
        if (!act) continue;
        for (int r0 = slot; r0 < nrows; r0 += U * RSL) {
            float4 pv[U];
            int pj[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int rr = min(r0 + u * RSL, nrows - 1);
                pj[u] = s_p[rr];
                pv[u] = ld4(a.P + (int64_t)(pj[u] < 0 ? 0 : pj[u]) * a.C + c);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int rr = r0 + u * RSL;
                if (rr >= nrows) continue;
                const float dx = s_dx[rr], dy = s_dy[rr], dz = s_dz[rr];
                float4 v = pj[u] < 0 ? make_float4(0.f, 0.f, 0.f, 0.f) : pv[u];
                v.x = fmaf(dz, w2.x, fmaf(dy, w1.x, fmaf(dx, w0.x, v.x + bv.x)));
                v.y = fmaf(dz, w2.y, fmaf(dy, w1.y, fmaf(dx, w0.y, v.y + bv.y)));
                v.z = fmaf(dz, w2.z, fmaf(dy, w1.z, fmaf(dx, w0.z, v.z + bv.z)));
                v.w = fmaf(dz, w2.w, fmaf(dy, w1.w, fmaf(dx, w0.w, v.w + bv.w)));
                *reinterpret_cast<float4 *>(a.y + (base + rr) * a.C + c) = v;
                if (wq_on) {          // (compacted stack: a group's first row stands for its padding copies too)
                    const float wq = s_wq[rr];
                    s1.x = fmaf(wq, v.x, s1.x); s1.y = fmaf(wq, v.y, s1.y); s1.z = fmaf(wq, v.z, s1.z); s1.w = fmaf(wq, v.w, s1.w);
                    s2.x = fmaf(wq * v.x, v.x, s2.x); s2.y = fmaf(wq * v.y, v.y, s2.y); s2.z = fmaf(wq * v.z, v.z, s2.z); s2.w = fmaf(wq * v.w, v.w, s2.w);
                    continue;
                }
                s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
                s2.x = fmaf(v.x, v.x, s2.x); s2.y = fmaf(v.y, v.y, s2.y); s2.z = fmaf(v.z, v.z, s2.z); s2.w = fmaf(v.w, v.w, s2.w);
            }
        }
    }
    __syncthreads();
    float *rd = red + tid * 8;
    rd[0] = act ? s1.x : 0.f; rd[1] = act ? s1.y : 0.f; rd[2] = act ? s1.z : 0.f; rd[3] = act ? s1.w : 0.f;
    rd[4] = act ? s2.x : 0.f; rd[5] = act ? s2.y : 0.f; rd[6] = act ? s2.z : 0.f; rd[7] = act ? s2.w : 0.f;
    __syncthreads();
    for (int t = tid; t < 2 * a.C; t += LG_T) {       // (which, channel): sum over the row slots in slot order
        const int which = t / a.C, ch = t - which * a.C;
        float sacc = 0.f;
        for (int sl = 0; sl < RSL; ++sl) sacc += red[(sl * CQ + (ch >> 2)) * 8 + which * 4 + (ch & 3)];
        a.stats[((int64_t)blockIdx.x * 2 + which) * a.C + ch] = sacc;
    }
}

// dY rows -> G[j] (atomic), dW_x partial sums.  Lane = channel, so one atomic instruction of a wave covers 64 consecutive floats of a
// G row (the quad-per-lane mapping of the forward kernel would issue four quarter-dense atomics instead).  Ball-query padding
// repeats each group's first neighbour, often for half of the nsample slots: those rows are summed in a register per group and
// flushed with one atomic.
__global__ __launch_bounds__(LG_T) void lingather_bwd_kernel(LinGatherArgs a)
{
    __shared__ float red[LG_T * 3];
    // per-row metadata, fetched ONCE per row by one thread (the 2 x C/64 waves that work on a row would otherwise each repeat the
    // same eight loads: the kernel was bound by memory instructions, not by bytes or by the atomics)
    __shared__ int s_j[LG_ROWS], s_b[LG_ROWS], s_grp[LG_ROWS];
    __shared__ float s_dx[LG_ROWS], s_dy[LG_ROWS], s_dz[LG_ROWS];
    __shared__ unsigned char s_dup[LG_ROWS];
    __shared__ float s_w[LG_ROWS];               // compacted stack: the row's multiplicity weight (1 otherwise)
    const int tid = threadIdx.x;
    const int RSL = LG_T / a.C;                   // C <= 256 (host-checked)
    const int ch = tid % a.C, slot = tid / a.C;
    const bool act = slot < RSL;
    const DySrc &d = a.d;
    const float ksc = d.scale[ch], ksh = d.shift[ch], kmu = d.mean[ch];
    const float kA = ksc * d.c1[ch], kB = ksc * d.c2[ch] * d.invstd[ch];
    int64_t rows_all = a.M, rpw = a.rows_per_wg;
    if (a.rows_dev) { rows_all = *a.rows_dev; rpw = (rows_all + gridDim.x - 1) / gridDim.x; }
    const int64_t mbeg = (int64_t)blockIdx.x * rpw, mend = min(rows_all, mbeg + rpw);
    const bool cp = a.cidx != nullptr;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f;
    float dsum = 0.f;                             // pending sum of padding duplicates ...
    int dgrp = -1, djf = -1, dbat = 0;            // ... of group dgrp (first neighbour djf, cloud dbat)
    for (int64_t base = mbeg; base < mend; base += LG_ROWS) {
        const int nrows = (int)min((int64_t)LG_ROWS, mend - base);
        __syncthreads();
        for (int rr = tid; rr < nrows; rr += LG_T) {
            const int64_t m = base + rr;
            const LgRow r = lg_row(a, m);
            s_j[rr] = r.j; s_b[rr] = r.b; s_dx[rr] = r.dx; s_dy[rr] = r.dy; s_dz[rr] = r.dz;
            if (cp) {             // (a compacted group holds at most 7 copies of its first neighbour: no pre-summing)
                s_grp[rr] = 0; s_dup[rr] = 0; s_w[rr] = a.d.wrow[m];
            } else {
                const int grp = (int)(m / a.K);
                const int jf = a.idx[(int64_t)grp * a.K];
                s_grp[rr] = grp; s_w[rr] = 1.f;
                s_dup[rr] = (r.j == jf && m != (int64_t)grp * a.K) ? 1 : 0;
            }
        }
        __syncthreads();
        if (!act) continue;
        constexpr int U = 8;
        for (int r0 = slot; r0 < nrows; r0 += U * RSL) {
            float vy[U], vz[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int rr = min(r0 + u * RSL, nrows - 1);
                vy[u] = d.y[(base + rr) * a.C + ch];
                vz[u] = d.dz[(base + rr) * a.C + ch];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int rr = r0 + u * RSL;
                if (rr >= nrows) continue;
                const int j = s_j[rr];
                if (j < 0) continue;
                const float z = fmaf(ksc, vy[u], ksh);
                const float pp = z > 0.f ? vz[u] : 0.f;
                const float g = cp ? fmaf(-s_w[rr], fmaf(kB, vy[u] - kmu, kA), ksc * pp) : fmaf(ksc, pp, -fmaf(kB, vy[u] - kmu, kA));
                acc0 = fmaf(g, s_dx[rr], acc0); acc1 = fmaf(g, s_dy[rr], acc1); acc2 = fmaf(g, s_dz[rr], acc2);
                if (s_dup[rr]) {                  // a padding duplicate of the group's first neighbour
                    const int grp = s_grp[rr];
                    if (grp != dgrp) {
                        if (dgrp >= 0 && djf >= 0 && djf < a.N) unsafeAtomicAdd(a.G + ((int64_t)dbat * a.N + djf) * a.C + ch, dsum);
                        dsum = 0.f; dgrp = grp; djf = j; dbat = s_b[rr];
                    }
                    dsum += g;
                } else {
                    unsafeAtomicAdd(a.G + ((int64_t)s_b[rr] * a.N + j) * a.C + ch, g);
                }
            }
        }
    }
    if (act && dgrp >= 0 && djf >= 0 && djf < a.N) unsafeAtomicAdd(a.G + ((int64_t)dbat * a.N + djf) * a.C + ch, dsum);
    __syncthreads();
    red[tid * 3 + 0] = act ? acc0 : 0.f; red[tid * 3 + 1] = act ? acc1 : 0.f; red[tid * 3 + 2] = act ? acc2 : 0.f;
    __syncthreads();
    float *out = a.dwx + (int64_t)blockIdx.x * a.C * 3;
    for (int t = tid; t < a.C * 3; t += LG_T) {
        const int c = t / 3, k = t - c * 3;
        float sacc = 0.f;
        for (int sl = 0; sl < RSL; ++sl) sacc += red[(sl * a.C + c) * 3 + k];
        out[t] = sacc;
    }
}

// ---- point lists: the inverse of the grouping (papc_point_lists_f32) ------------------------------------------------------------------------
// One workgroup per cloud.  (1) counts per source point in LDS, (2) exclusive scan -> prange, (3) the cloud's groups in order, 64 rows a step: a row
// takes the next slot of its point's list, so every list is ascending in the row index -- a FIXED summation order for the backward below.
// Within a step the ball query's structure makes the slots unique except for the padding copies of the group's first neighbour (one ballot ranks
// them); any other repeated index inside a step (a caller's own lists) is detected after the LDS atomics and ranked by a 64-step lane sweep.
// PADDED layout: a group's padding copies (rows k > 0 that repeat its first neighbour) are rows with identical input, hence identical
// activations and identical gradients -- the first-maximum rule hands the max-pool's gradient to row 0, never to a copy -- so they enter the
// first neighbour's list as ONE entry: the first copy's row with weight = their number (pmeta.w; 1 for every other entry).  Without that a point
// that is the first neighbour of a few groups owns a list of hundreds of rows, walked by one wave.
struct PlArgs {
    const float *xyz; int64_t sb, sn, sc;
    const float *new_xyz;
    const int32_t *idx, *cidx, *start, *rows_dev;
    const float *wrow;
    int N, S, K, G;
    int32_t *prange, *prow;
    float4 *pmeta;
};

// W waves per cloud, each owning a contiguous chunk of the cloud's groups and its own cursor row cw[w][.] in LDS: the serial walk (the LDS cursor
// chain of step (3)) is S / W groups long instead of S, and a point's list is the concatenation of the waves' pieces in wave order -- still
// ascending in the row index.
constexpr int PL_MAXW = 16;

__global__ __launch_bounds__(64 * PL_MAXW) void point_lists_kernel(PlArgs a, int W)
{
    extern __shared__ int pl_lds[];
    int *cw = pl_lds;                              // [W][N]: counts, then cursors
    __shared__ int scan[64 * PL_MAXW];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NT = 64 * W;
    const bool cp = a.cidx != nullptr;
    const int g0 = b * a.S, g1 = g0 + a.S;
    const int32_t *src = cp ? a.cidx : a.idx;
    const int gper = (a.S + W - 1) / W;
    const int wg0 = min(g1, g0 + wave * gper), wg1 = min(g1, wg0 + gper);      // this wave's groups
    auto row_of = [&](int g) { return cp ? (g == a.G ? a.rows_dev[0] : a.start[g]) : g * a.K; };
    const int r0 = row_of(g0);
    int *gflag = cw + W * a.N;                     // [S] padded layout: the group's padding copies have been counted
    for (int e = tid; e < W * a.N + (cp ? 0 : a.S); e += NT) cw[e] = 0;
    __syncthreads();
    if (wave < W) {
        int *cnt = cw + wave * a.N;
        const int m0 = row_of(wg0), m1 = row_of(wg1);
        for (int m = m0 + lane; m < m1; m += 64) {
            const int j = src[m];
            if (j < 0 || j >= a.N) continue;
            bool copy = false;
            int gl = 0;
            if (!cp) {                             // PADDED layout: a group's copies of its first neighbour (identical rows, identical gradients) are ONE entry
                gl = m / a.K - g0;
                const int k = m - (g0 + gl) * a.K;
                copy = k > 0 && j == src[m - k];
            }
            if (!copy || atomicExch(&gflag[gl], 1) == 0) atomicAdd(&cnt[j], 1);
        }
    }
    __syncthreads();
    {   // a thread owns a contiguous block of points: totals over the waves, block scan across threads, then every wave's first slot per point
        const int per = (a.N + NT - 1) / NT;
        const int p0 = min(a.N, tid * per), p1 = min(a.N, p0 + per);
        int s = 0;
        for (int p = p0; p < p1; ++p)
            for (int w = 0; w < W; ++w) s += cw[w * a.N + p];
        scan[tid] = s;
        __syncthreads();
        for (int o = 1; o < NT; o <<= 1) {          // Hillis-Steele inclusive scan of the threads' sums
            const int v = tid >= o ? scan[tid - o] : 0;
            __syncthreads();
            scan[tid] += v;
            __syncthreads();
        }
        int run = r0 + scan[tid] - s;
        for (int p = p0; p < p1; ++p) {
            a.prange[2 * ((int64_t)b * a.N + p)] = run;
            for (int w = 0; w < W; ++w) {
                const int c = cw[w * a.N + p];
                cw[w * a.N + p] = run;
                run += c;
            }
            a.prange[2 * ((int64_t)b * a.N + p) + 1] = run;
        }
    }
    __syncthreads();
    if (wave >= W) return;
    int *cur = cw + wave * a.N;
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    for (int g = wg0; g < wg1; ++g) {
        const int s = cp ? a.start[g] : g * a.K;
        const int n = cp ? ((g == a.G - 1 ? a.rows_dev[0] : a.start[g + 1]) - s) : a.K;
        const int first = src[s];
        const float cx = a.new_xyz[(int64_t)g * 3], cy = a.new_xyz[(int64_t)g * 3 + 1], cz = a.new_xyz[(int64_t)g * 3 + 2];
        int ncopy = 0, mcopy = 0;                  // padded layout: the group's padding copies, the first of them
        for (int k0 = 0; k0 < n; k0 += 64) {
            const int k = k0 + lane;
            const bool in = k < n;
            const int m = s + (in ? k : 0);
            const int p = in ? src[m] : -1;
            const bool valid = in && p >= 0 && p < a.N;
            const bool copy = valid && k > 0 && p == first;
            const bool uniq = valid && !copy;
            int pos = 0;
            if (uniq) pos = atomicAdd(&cur[p], 1);
            // (one wave per cursor row: the LDS atomics of one instruction have all retired before the wave's next LDS read)
            const bool clash = uniq && cur[p] != pos + 1;
            if (__ballot(clash)) {                 // repeated indices that are not padding copies: rank them by lane
                int base = pos, rank = 0;
                for (int l = 0; l < 64; ++l) {
                    const int pl = __builtin_amdgcn_readlane(p, l), ol = __builtin_amdgcn_readlane(pos, l);
                    const int ul = __builtin_amdgcn_readlane((int)uniq, l);
                    if (ul && pl == p) { base = min(base, ol); rank += l < lane ? 1 : 0; }
                }
                if (uniq) pos = base + rank;
            }
            const unsigned long long cm = __ballot(copy);
            if (cm && cp) {
                const int leader = __ffsll((long long)cm) - 1;
                int base = 0;
                if (lane == leader) base = atomicAdd(&cur[first], __popcll(cm));
                base = __builtin_amdgcn_readlane(base, leader);
                if (copy) pos = base + __popcll(cm & lt);
            } else if (cm) {                       // (padded: counted here, ONE weighted entry behind the group's rows)
                if (!ncopy) mcopy = s + k0 + (__ffsll((long long)cm) - 1);
                ncopy += __popcll(cm);
            }
            if (valid && (cp || !copy)) {
                const float *q = a.xyz + (int64_t)b * a.sb + (int64_t)p * a.sn;
                a.prow[pos] = m;
                a.pmeta[pos] = make_float4(q[0] - cx, q[a.sc] - cy, q[2 * a.sc] - cz, a.wrow ? a.wrow[m] : 1.f);
            }
        }
        if (ncopy && lane == 0) {
            // the copies are rows with the first neighbour's input: identical activations, identical gradients (the first-maximum rule sends
            // the max-pool's gradient to row 0, never to a copy) -- the entry stands for all of them with weight = their number
            const int pos = atomicAdd(&cur[first], 1);
            const float *q = a.xyz + (int64_t)b * a.sb + (int64_t)first * a.sn;
            a.prow[pos] = mcopy;
            a.pmeta[pos] = make_float4(q[0] - cx, q[a.sc] - cy, q[2 * a.sc] - cz, (float)ncopy);
        }
    }
}

// ---- backward over the point lists: G[p] = sum over p's rows (ascending) of dY[m], no atomics; the xyz columns of dW as before -------------------
// A WAVE owns a point at a time: its two half-waves take the list's even and odd entries (lane & 31 = channel quad for C = 128: one 512-byte row
// per half-wave and load), U entries each per pass, so both halves always work on the same list (no divergence between them) and an average list
// (9 rows) is two dependent round trips: the entries, then their rows.  The halves' sums are added in fixed order (even + odd).  Eight points per
// workgroup keep ~2000 workgroups in flight: the kernel is a latency-bound gather of 4C-byte rows, it wants every wave slot of the chip.
constexpr int LGL_PPW = 8;                      // points per workgroup (two per wave)

template <bool CP, int U = 4, int PPW = LGL_PPW>
__global__ __launch_bounds__(LG_T) void lingather_bwd_lists_kernel(LinGatherArgs a, const int32_t *__restrict__ prange, const int32_t *__restrict__ prow,
                                                                   const float4 *__restrict__ pmeta, int64_t BN)
{
    __shared__ float4 red[LG_T * 3];
    const int tid = threadIdx.x;
    const int CQ = a.C >> 2;                      // host-checked: 64 % CQ == 0 (C in {16, 32, 64, 128, 256})
    const int lane = tid & 63, wave = tid >> 6;
    const int SUB = 64 / CQ;                      // list entries a wave takes per load (2 for C = 128)
    const int cq = lane % CQ, sub = lane / CQ, c = cq * 4;
    const DySrc &d = a.d;
    const float4 ksc = ld4(d.scale + c), ksh = ld4(d.shift + c), kmu = ld4(d.mean + c);
    float4 kA, kB;
    {
        const float4 c1 = ld4(d.c1 + c), c2 = ld4(d.c2 + c), is = ld4(d.invstd + c);
        kA = make_float4(ksc.x * c1.x, ksc.y * c1.y, ksc.z * c1.z, ksc.w * c1.w);
        kB = make_float4(ksc.x * c2.x * is.x, ksc.y * c2.y * is.y, ksc.z * c2.z * is.z, ksc.w * c2.w * is.w);
    }
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0;
    const int64_t p0 = (int64_t)blockIdx.x * PPW;
    auto gval = [&](float y, float z, float sc, float sh, float mu, float A, float Bc, float w) {
        const float pre = fmaf(sc, y, sh);
        const float pp = pre > 0.f ? z : 0.f;
        return CP ? fmaf(-w, fmaf(Bc, y - mu, A), sc * pp) : fmaf(sc, pp, -fmaf(Bc, y - mu, A));
    };
    for (int64_t pt = p0 + wave; pt < min(BN, p0 + PPW); pt += LG_T / 64) {
        const int rs = prange[2 * pt], re = prange[2 * pt + 1];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = rs + sub; k < re + sub; k += U * SUB) {      // (uniform trip count across the wave: the halves differ by one entry at most)
            int m[U];
            float4 mt[U], vy[U], vz[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int kk = min(k + u * SUB, re - 1);
                m[u] = prow[kk];
                mt[u] = pmeta[kk];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                vy[u] = ld4(d.y + (int64_t)m[u] * a.C + c);
                vz[u] = ld4(d.dz + (int64_t)m[u] * a.C + c);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (k + u * SUB >= re) continue;
                float4 g;
                g.x = gval(vy[u].x, vz[u].x, ksc.x, ksh.x, kmu.x, kA.x, kB.x, mt[u].w);
                g.y = gval(vy[u].y, vz[u].y, ksc.y, ksh.y, kmu.y, kA.y, kB.y, mt[u].w);
                g.z = gval(vy[u].z, vz[u].z, ksc.z, ksh.z, kmu.z, kA.z, kB.z, mt[u].w);
                g.w = gval(vy[u].w, vz[u].w, ksc.w, ksh.w, kmu.w, kA.w, kB.w, mt[u].w);
                if (!CP) { g.x *= mt[u].w; g.y *= mt[u].w; g.z *= mt[u].w; g.w *= mt[u].w; }      // (padded: an entry may stand for a group's copies; 1 otherwise)
                acc.x += g.x; acc.y += g.y; acc.z += g.z; acc.w += g.w;
                a0.x = fmaf(g.x, mt[u].x, a0.x); a0.y = fmaf(g.y, mt[u].x, a0.y); a0.z = fmaf(g.z, mt[u].x, a0.z); a0.w = fmaf(g.w, mt[u].x, a0.w);
                a1.x = fmaf(g.x, mt[u].y, a1.x); a1.y = fmaf(g.y, mt[u].y, a1.y); a1.z = fmaf(g.z, mt[u].y, a1.z); a1.w = fmaf(g.w, mt[u].y, a1.w);
                a2.x = fmaf(g.x, mt[u].z, a2.x); a2.y = fmaf(g.y, mt[u].z, a2.y); a2.z = fmaf(g.z, mt[u].z, a2.z); a2.w = fmaf(g.w, mt[u].z, a2.w);
            }
        }
        // the wave's SUB partial sums, added in sub order (fixed): sub 0 collects
        for (int o = CQ; o < 64; o <<= 1) {
            acc.x += __shfl_down(acc.x, o); acc.y += __shfl_down(acc.y, o); acc.z += __shfl_down(acc.z, o); acc.w += __shfl_down(acc.w, o);
        }
        if (sub == 0) *reinterpret_cast<float4 *>(a.G + pt * a.C + c) = acc;
    }
    red[tid * 3 + 0] = a0; red[tid * 3 + 1] = a1; red[tid * 3 + 2] = a2;
    __syncthreads();
    float *out = a.dwx + (int64_t)blockIdx.x * a.C * 3;
    const int SL = LG_T / CQ;                     // (wave, sub) slots per channel quad
    for (int t = tid; t < a.C * 3; t += LG_T) {       // (channel, coordinate): the slots summed in slot order
        const int ch = t / 3, k = t - ch * 3;
        float sacc = 0.f;
        for (int sl = 0; sl < SL; ++sl) {
            const float4 v = red[(sl * CQ + (ch >> 2)) * 3 + k];
            sacc += (ch & 3) == 0 ? v.x : (ch & 3) == 1 ? v.y : (ch & 3) == 2 ? v.z : v.w;
        }
        out[t] = sacc;
    }
}


// ---- per-point moments of the lists (papc_point_lists.pmom) -------------------------------------------------------------------------------------
// Eight lanes per source point take its list's entries round robin and add up in a fixed tree: w_j = sum w, D_j = sum w d, M2_j = sum w d d^T over
// the list's entries (w, d) = pmeta.  Weight-independent like the lists: built with them (papc_point_lists_f32), read by the backward below.
__global__ __launch_bounds__(256) void point_moments_kernel(const int32_t *__restrict__ prange, const float4 *__restrict__ pmeta, float4 *__restrict__ pmom,
                                                            int64_t BN)
{
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t pt = min(gid >> 3, BN - 1);      // (whole 8-lane groups stay converged for the shuffles; the surplus groups repeat the last point)
    const int l = (int)(gid & 7);
    const int rs = prange[2 * pt], re = prange[2 * pt + 1];
    float v[10] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // w | d0 d1 d2 | m00 m01 m02 m11 m12 m22
    for (int k = rs + l; k < re; k += 8) {
        const float4 e = pmeta[k];
        const float wx = e.w * e.x, wy = e.w * e.y, wz = e.w * e.z;
        v[0] += e.w; v[1] += wx; v[2] += wy; v[3] += wz;
        v[4] = fmaf(wx, e.x, v[4]); v[5] = fmaf(wx, e.y, v[5]); v[6] = fmaf(wx, e.z, v[6]);
        v[7] = fmaf(wy, e.y, v[7]); v[8] = fmaf(wy, e.z, v[8]); v[9] = fmaf(wz, e.z, v[9]);
    }
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        v[i] += __shfl_xor(v[i], 4); v[i] += __shfl_xor(v[i], 2); v[i] += __shfl_xor(v[i], 1);
    }
    if (l == 0 && (gid >> 3) < BN) {
        pmom[3 * pt] = make_float4(v[0], v[1], v[2], v[3]);
        pmom[3 * pt + 1] = make_float4(v[4], v[5], v[6], v[7]);
        pmom[3 * pt + 2] = make_float4(v[8], v[9], 0.f, 0.f);
    }
}

// ---- the list backward WITHOUT re-reading y (papc_lingather_bwd_pp_f32) -------------------------------------------------------------------------
// dz arrives MASKED (p = dz where the layer's ReLU is open: papc_bwd_red.store_masked on the dX launch that wrote it), so the only per-row
// quantity left is p itself; everything the BatchNorm backward takes from y[m] = P[j] + W_x d_m + b is linear in the per-point moments of the
// list entries' (w, d) -- pmom, made with the lists -- and in P[j]:
//   G[j]      = sc S1 - kB (w_j (P[j] + b - mu) + W_x D_j) - kA w_j
//   dW_x[c,t] = sum_j [ sc sum_{m in j} p d_t - kB ((P[j] + b - mu)_c D_j[t] + sum_s W_x[c,s] M2_j[s,t]) - kA D_j[t] ]
// One [rows, C] stream gathered instead of two.  Measured by elimination (tools/probe/r06_lg_pp_variants.sh, round 6) the kernel is NOT its
// bytes: rows in list order instead of gathered made 50 -> 46 us, and of the first version's 50 us the entry loads cost 19, the row loads 16, the
// empty skeleton (launch, range, constants, tail) 15.  Hence the geometry: ONE point per wave, the whole wave on one row at a time (lane = C / 64
// channels), so range, row numbers, d and the moments are WAVE-UNIFORM; a list's entries arrive in one coalesced vector load (lane u = entry u)
// and are handed round by readlane (scalar loads per entry -- the scalar cache -- were the 19 us); U row loads in flight for U x C/64 vector
// registers; no barrier ahead of the gathers (the per-channel constants sit in the lane's own registers); 8 waves per SIMD.  Eight waves =
// eight points per workgroup, their finished dW_x shares added in wave order.
struct LgPpExtra { const float *P, *w, *bias; const float4 *pmom; int ldw, xcol0; };
constexpr int LGP_T = 64 * LGL_PPW;

template <int VEC, int U, bool WEIGHTED>
__global__ __launch_bounds__(LGP_T) void lingather_bwd_pp_kernel(LinGatherArgs a, LgPpExtra e, const int32_t *__restrict__ prange,
                                                                 const int32_t *__restrict__ prow, const float4 *__restrict__ pmeta, int64_t BN)
{
    __shared__ float red[LGL_PPW][3][64 * VEC];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int c = lane * VEC;
    const DySrc &d = a.d;
    float sc[VEC], kA[VEC], kB[VEC], boff[VEC], w0[VEC], w1[VEC], w2[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        const float s_ = d.scale[c + v], e2 = d.c2[c + v] * d.invstd[c + v];
        sc[v] = s_; kA[v] = s_ * d.c1[c + v]; kB[v] = s_ * e2;
        boff[v] = (e.bias ? e.bias[c + v] : 0.f) - d.mean[c + v];
        const float *wp = e.w + (int64_t)(c + v) * e.ldw + e.xcol0;
        w0[v] = wp[0]; w1[v] = wp[1]; w2[v] = wp[2];
    }
    float at[3][VEC];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int v = 0; v < VEC; ++v) at[t][v] = 0.f;
    const int64_t pt = (int64_t)blockIdx.x * LGL_PPW + wave;      // (wave-uniform)
    if (pt < BN) {
        const int rs = prange[2 * pt], re = prange[2 * pt + 1];
        const float4 mo = e.pmom[3 * pt], m1 = e.pmom[3 * pt + 1], m2 = e.pmom[3 * pt + 2];      // w_j, D_j | M2 00 01 02 11 | 12 22
        float Pv[VEC], acc[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) { Pv[v] = e.P[pt * a.C + c + v]; acc[v] = 0.f; }
        for (int base = rs; base < re; base += 64) {
            const int nn = min(64, re - base);
            int rowv = 0;
            float ex = 0.f, ey = 0.f, ez = 0.f, ew = 1.f;      // (WEIGHTED -- the padded layout: an entry may stand for a group's padding copies)
            if (lane < nn) {
                rowv = prow[base + lane];
                const float4 t4 = pmeta[base + lane];
                ex = t4.x; ey = t4.y; ez = t4.z;
                if (WEIGHTED) ew = t4.w;
            }
            for (int u0 = 0; u0 < nn; u0 += U) {
                int m[U];
                float dx[U], dy[U], dzc[U], wu[U];
                float vz[U][VEC];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int uu = min(u0 + u, nn - 1);
                    m[u] = __builtin_amdgcn_readlane(rowv, uu);
                    dx[u] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ex), uu));
                    dy[u] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ey), uu));
                    dzc[u] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ez), uu));
                    wu[u] = WEIGHTED ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ew), uu)) : 1.f;
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const float *src = d.dz + (int64_t)m[u] * a.C + c;
                    if constexpr (VEC == 4) { const float4 t4 = ld4(src); vz[u][0] = t4.x; vz[u][1] = t4.y; vz[u][2] = t4.z; vz[u][3] = t4.w; }
                    else if constexpr (VEC == 2) { const float2 t2 = *reinterpret_cast<const float2 *>(src); vz[u][0] = t2.x; vz[u][1] = t2.y; }
                    else vz[u][0] = src[0];
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (u0 + u >= nn) continue;
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        const float g = WEIGHTED ? vz[u][v] * wu[u] : vz[u][v];
                        acc[v] += g;
                        at[0][v] = fmaf(g, dx[u], at[0][v]); at[1][v] = fmaf(g, dy[u], at[1][v]); at[2][v] = fmaf(g, dzc[u], at[2][v]);
                    }
                }
            }
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const float pb = Pv[v] + boff[v];                                                             // P[j] + b - mu
            const float yv = fmaf(w2[v], mo.w, fmaf(w1[v], mo.z, fmaf(w0[v], mo.y, mo.x * pb)));           // sum_m w (y[m] - mu)
            a.G[pt * a.C + c + v] = fmaf(sc[v], acc[v], -fmaf(kB[v], yv, kA[v] * mo.x));
            // the point's finished share of dW_x: column t of W_x M2_j is (w0 M2[0][t] + w1 M2[1][t] + w2 M2[2][t])
            const float q0 = fmaf(w2[v], m1.z, fmaf(w1[v], m1.y, fmaf(w0[v], m1.x, pb * mo.y)));
            const float q1 = fmaf(w2[v], m2.x, fmaf(w1[v], m1.w, fmaf(w0[v], m1.y, pb * mo.z)));
            const float q2 = fmaf(w2[v], m2.y, fmaf(w1[v], m2.x, fmaf(w0[v], m1.z, pb * mo.w)));
            at[0][v] = fmaf(sc[v], at[0][v], -fmaf(kB[v], q0, kA[v] * mo.y));
            at[1][v] = fmaf(sc[v], at[1][v], -fmaf(kB[v], q1, kA[v] * mo.z));
            at[2][v] = fmaf(sc[v], at[2][v], -fmaf(kB[v], q2, kA[v] * mo.w));
        }
    }
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int v = 0; v < VEC; ++v) red[wave][t][lane * VEC + v] = at[t][v];
    __syncthreads();
    float *out = a.dwx + (int64_t)blockIdx.x * a.C * 3;
    for (int t = threadIdx.x; t < a.C * 3; t += LGP_T) {       // (channel, coordinate): the waves' shares in wave order
        const int ch = t / 3, k = t - ch * 3;
        float sA = 0.f;
#pragma unroll
        for (int w = 0; w < LGL_PPW; ++w) sA += red[w][k][ch];
        out[t] = sA;
    }
}

}  // namespace papc

using namespace papc;

static int lg_check(const papc_group_src *g, int B, int C, const char *who)
{
    PAPC_REQUIRE(g && g->xyz && g->new_xyz && g->idx, PAPC_E_INVALID, "%s: group source needs xyz, new_xyz and idx", who);
    PAPC_REQUIRE(B >= 1 && g->N >= 1 && g->S >= 1 && g->K >= 1, PAPC_E_INVALID, "%s: bad B/N/S/K", who);
    PAPC_REQUIRE(C >= 4 && C % 4 == 0 && C <= 1024, PAPC_E_UNSUPPORTED, "%s: C=%d must be a multiple of 4 in [4, 1024]", who, C);
    PAPC_REQUIRE((int64_t)B * g->S * g->K < (1ll << 31), PAPC_E_UNSUPPORTED, "%s: >= 2^31 rows", who);
    PAPC_REQUIRE(!g->cidx || (g->seg_grp && g->rows_dev), PAPC_E_INVALID, "%s: a compacted group source needs cidx, seg_grp and rows_dev", who);
    return 0;
}

static void lg_fill(LinGatherArgs &a, const papc_group_src *g, int B, int C)
{
    a.xyz = g->xyz; a.sb = g->sb; a.sn = g->sn; a.sc = g->sc; a.new_xyz = g->new_xyz; a.idx = g->idx;
    a.B = B; a.N = g->N; a.S = g->S; a.K = g->K; a.C = C; a.M = (int64_t)B * g->S * g->K;
    a.cidx = g->cidx; a.seg_grp = g->seg_grp; a.rows_dev = g->rows_dev;
}

extern "C" {

// workgroups (= rows of the partial buffers): the kernels are chains of dependent loads (neighbour index -> point), so they want
// many resident waves -- 8 workgroups of 256 threads per CU
int papc_lingather_parts(int64_t M)
{
    const int cap = knob(KNOB_LG_PARTS);
    return (int)std::min<int64_t>(cap, std::max<int64_t>(1, (M + 127) / 128));
}

int papc_lingather_fwd_f32(const float *P, const papc_group_src *grp, int B, const float *w, int ldw, int xcol0, const float *bias, int C,
                           float *y, float *stats_partial, papc_stream_t stream)
{
    int rc = lg_check(grp, B, C, "papc_lingather_fwd_f32");
    if (rc) return rc;
    PAPC_REQUIRE(P && w && y && stats_partial, PAPC_E_INVALID, "papc_lingather_fwd_f32: null pointer");
    PAPC_REQUIRE(aligned16(P) && aligned16(y) && (!bias || aligned16(bias)) && ldw >= xcol0 + 3 && xcol0 >= 0, PAPC_E_INVALID,
                 "papc_lingather_fwd_f32: alignment / weight columns");
    LinGatherArgs a;
    memset(&a, 0, sizeof(a));
    lg_fill(a, grp, B, C);
    a.P = P; a.w = w; a.ldw = ldw; a.xcol0 = xcol0; a.bias = bias; a.y = y; a.stats = stats_partial;
    a.wstat = grp->cidx ? grp->wstat : nullptr;
    const int parts = papc_lingather_parts(a.M);
    a.rows_per_wg = (int)((a.M + parts - 1) / parts);
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_MLP_GEMM, st);
    hipLaunchKernelGGL(lingather_fwd_kernel, dim3((unsigned)parts), dim3(LG_T), 0, st, a);
    return check_launch("papc_lingather_fwd_f32");
}

int papc_lingather_list_parts(int64_t BN) { return (int)((BN + LGL_PPW - 1) / LGL_PPW); }

static bool lg_lists_usable(const papc_group_src *g, int C);
// the backward that gathers dz alone: the lists with their per-point moments, a first-layer width the kernel is built for (C / 64 channels per lane)
static bool lg_pp_usable(const papc_group_src *g, int /*B: no size limit of its own*/, int C)
{
    return lg_lists_usable(g, C) && g->plists->pmom && (C == 64 || C == 128 || C == 256) && knob(KNOB_LG_PP) != 0;
}

static bool lg_lists_usable(const papc_group_src *g, int C)
{
    return g->plists && g->plists->prange && g->plists->prow && g->plists->pmeta && (g->plists->compact != 0) == (g->cidx != nullptr) && C % 4 == 0 &&
           C / 4 <= 64 && 64 % (C / 4) == 0 && knob(KNOB_LG_LISTS) != 0;
}

int papc_lingather_bwd_lists_ok(const papc_group_src *grp, int C) { return grp && lg_lists_usable(grp, C) ? 1 : 0; }
int papc_lingather_bwd_pp_ok(const papc_group_src *grp, int B, int C) { return grp && lg_pp_usable(grp, B, C) ? 1 : 0; }

int papc_lingather_bwd_parts(const papc_group_src *grp, int B, int C)
{
    if (!grp) return 0;
    const int64_t M = (int64_t)B * grp->S * grp->K;
    return lg_lists_usable(grp, C) ? papc_lingather_list_parts((int64_t)B * grp->N) : papc_lingather_parts(M);
}

int papc_point_lists_f32(const papc_group_src *grp, int B, const int32_t *start, int32_t *prange, int32_t *prow, float *pmeta, float *pmom,
                         papc_stream_t stream)
{
    int rc = lg_check(grp, B, 4, "papc_point_lists_f32");
    if (rc) return rc;
    PAPC_REQUIRE(prange && prow && pmeta, PAPC_E_INVALID, "papc_point_lists_f32: null pointer");
    PAPC_REQUIRE(!grp->cidx == !start, PAPC_E_INVALID, "papc_point_lists_f32: a compacted grouping needs start, a padded one must not pass it");
    PAPC_REQUIRE(!grp->cidx || grp->wstat, PAPC_E_INVALID, "papc_point_lists_f32: a compacted grouping needs its row weights (grp->wstat = wrow)");
    PAPC_REQUIRE(grp->N <= 8192, PAPC_E_UNSUPPORTED, "papc_point_lists_f32: N=%d > 8192 source points per cloud", grp->N);
    PAPC_REQUIRE(aligned16(pmeta) && (!pmom || aligned16(pmom)), PAPC_E_INVALID, "papc_point_lists_f32: pmeta / pmom must be 16-byte aligned");
    PlArgs a;
    memset(&a, 0, sizeof(a));
    a.xyz = grp->xyz; a.sb = grp->sb; a.sn = grp->sn; a.sc = grp->sc; a.new_xyz = grp->new_xyz; a.idx = grp->idx; a.cidx = grp->cidx; a.start = start;
    a.rows_dev = grp->rows_dev; a.wrow = grp->cidx ? grp->wstat : nullptr; a.N = grp->N; a.S = grp->S; a.K = grp->K; a.G = B * grp->S;
    a.prange = prange; a.prow = prow; a.pmeta = reinterpret_cast<float4 *>(pmeta);
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_GROUP, st);
    const int W = std::max(1, std::min(PL_MAXW, 16384 / grp->N));          // waves per cloud: W cursor rows of N ints in LDS (<= 64 KB)
    const size_t lds = ((size_t)W * grp->N + (grp->cidx ? 0 : grp->S)) * sizeof(int);      // cursor rows + (padded layout) one flag per group
    if (lds > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void *>(point_lists_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return check_launch("papc_point_lists_f32: hipFuncSetAttribute");
    PAPC_REQUIRE(lds <= 150 * 1024, PAPC_E_UNSUPPORTED, "papc_point_lists_f32: N=%d S=%d do not fit the builder's LDS", grp->N, grp->S);
    hipLaunchKernelGGL(point_lists_kernel, dim3((unsigned)B), dim3(64 * W), lds, st, a, W);
    rc = check_launch("papc_point_lists_f32");
    if (rc || !pmom) return rc;
    const int64_t BN = (int64_t)B * grp->N;
    hipLaunchKernelGGL(point_moments_kernel, dim3((unsigned)((BN * 8 + 255) / 256)), dim3(256), 0, st, prange, reinterpret_cast<const float4 *>(pmeta),
                       reinterpret_cast<float4 *>(pmom), BN);
    return check_launch("papc_point_lists_f32 (moments)");
}

int papc_lingather_bwd_pp_f32(const papc_bwd_dy *dy, const papc_group_src *grp, int B, int C, const float *P, const float *w, int ldw, int xcol0,
                              const float *bias, float *G, float *dwx_partial, papc_stream_t stream)
{
    int rc = lg_check(grp, B, C, "papc_lingather_bwd_pp_f32");
    if (rc) return rc;
    PAPC_REQUIRE(dy && dy->dz_mode == PAPC_DZ_DENSE && dy->dz && dy->mean && dy->invstd && dy->scale && dy->c1 && dy->c2, PAPC_E_INVALID,
                 "papc_lingather_bwd_pp_f32: needs a dense (masked) dZ source with its BatchNorm-backward constants");
    PAPC_REQUIRE(P && w && G && dwx_partial && ldw >= xcol0 + 3 && xcol0 >= 0, PAPC_E_INVALID, "papc_lingather_bwd_pp_f32: null pointer / weight columns");
    PAPC_REQUIRE(papc_lingather_bwd_pp_ok(grp, B, C), PAPC_E_UNSUPPORTED, "papc_lingather_bwd_pp_f32: needs the grouping's point lists WITH their moments "
                 "(papc_point_lists_f32, pmom) in the layout the stack runs and C in {64, 128, 256} (papc_lingather_bwd_pp_ok)");
    PAPC_REQUIRE(aligned16(G) && aligned16(dy->dz) && aligned16(P) && (!bias || aligned16(bias)), PAPC_E_INVALID, "papc_lingather_bwd_pp_f32: 16-byte alignment");
    LinGatherArgs a;
    memset(&a, 0, sizeof(a));
    lg_fill(a, grp, B, C);
    a.d.dz = dy->dz; a.d.mean = dy->mean; a.d.invstd = dy->invstd; a.d.scale = dy->scale; a.d.c1 = dy->c1; a.d.c2 = dy->c2; a.d.C = C;
    a.G = G; a.dwx = dwx_partial;
    const papc_point_lists &pl = *grp->plists;
    LgPpExtra e{P, w, bias, reinterpret_cast<const float4 *>(pl.pmom), ldw, xcol0};
    const int64_t BN = (int64_t)B * grp->N;
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_BWD_DW, st);
    const dim3 grid((unsigned)papc_lingather_list_parts(BN));
    const float4 *pm = reinterpret_cast<const float4 *>(pl.pmeta);
    // (U = 8 rows in flight per wave; 6 / 12 / 16 and two points per wave measured the same +- 1 us on SA2 of the SSG classifier, round 6)
#define PAPC_LGP_GO(VEC) do { if (a.cidx) hipLaunchKernelGGL((lingather_bwd_pp_kernel<VEC, 8, false>), grid, dim3(LGP_T), 0, st, a, e, pl.prange, pl.prow, pm, BN); \
                              else hipLaunchKernelGGL((lingather_bwd_pp_kernel<VEC, 8, true>), grid, dim3(LGP_T), 0, st, a, e, pl.prange, pl.prow, pm, BN); } while (0)
    if (C == 64) PAPC_LGP_GO(1);
    else if (C == 128) PAPC_LGP_GO(2);
    else PAPC_LGP_GO(4);
#undef PAPC_LGP_GO
    return check_launch("papc_lingather_bwd_pp_f32");
}

int papc_lingather_bwd_f32(const papc_bwd_dy *dy, const papc_group_src *grp, int B, int C, float *G, float *dwx_partial, papc_stream_t stream)
{
    int rc = lg_check(grp, B, C, "papc_lingather_bwd_f32");
    if (rc) return rc;
    PAPC_REQUIRE(dy && dy->dz_mode == PAPC_DZ_DENSE && dy->dz && dy->y && dy->mean && dy->invstd && dy->scale && dy->shift && dy->c1 && dy->c2,
                 PAPC_E_INVALID, "papc_lingather_bwd_f32: needs a dense dY source");
    PAPC_REQUIRE(G && dwx_partial, PAPC_E_INVALID, "papc_lingather_bwd_f32: null pointer");
    PAPC_REQUIRE(C <= LG_T, PAPC_E_UNSUPPORTED, "papc_lingather_bwd_f32: C=%d > %d", C, LG_T);
    LinGatherArgs a;
    memset(&a, 0, sizeof(a));
    lg_fill(a, grp, B, C);
    a.d.dz = dy->dz; a.d.y = dy->y; a.d.mean = dy->mean; a.d.invstd = dy->invstd; a.d.scale = dy->scale; a.d.shift = dy->shift;
    a.d.c1 = dy->c1; a.d.c2 = dy->c2; a.d.C = C; a.d.wrow = dy->wrow;
    PAPC_REQUIRE(!a.cidx == !dy->wrow, PAPC_E_INVALID, "papc_lingather_bwd_f32: compacted group source and compacted dY source go together");
    a.G = G; a.dwx = dwx_partial;
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_BWD_DW, st);
    if (lg_lists_usable(grp, C)) {
        // the grouping's point lists: every row of G is written (a point in no group gets zeros), each as the sum of its rows in ascending order
        PAPC_REQUIRE(aligned16(G) && aligned16(dy->y) && aligned16(dy->dz), PAPC_E_INVALID, "papc_lingather_bwd_f32: G / y / dz must be 16-byte aligned");
        const int64_t BN = (int64_t)B * grp->N;
        const unsigned nwg = (unsigned)papc_lingather_list_parts(BN);
        const papc_point_lists &pl = *grp->plists;
        const int var = knob(KNOB_LGL_VARIANT);
        const float4 *pm = reinterpret_cast<const float4 *>(pl.pmeta);
        if (a.cidx && var == 1) hipLaunchKernelGGL((lingather_bwd_lists_kernel<true, 2, 8>), dim3(nwg), dim3(LG_T), 0, st, a, pl.prange, pl.prow, pm, BN);
        else if (a.cidx && var == 3) hipLaunchKernelGGL((lingather_bwd_lists_kernel<true, 8, 8>), dim3(nwg), dim3(LG_T), 0, st, a, pl.prange, pl.prow, pm, BN);
        else if (a.cidx)
            hipLaunchKernelGGL(lingather_bwd_lists_kernel<true>, dim3(nwg), dim3(LG_T), 0, st, a, pl.prange, pl.prow, reinterpret_cast<const float4 *>(pl.pmeta), BN);
        else
            hipLaunchKernelGGL(lingather_bwd_lists_kernel<false>, dim3(nwg), dim3(LG_T), 0, st, a, pl.prange, pl.prow, reinterpret_cast<const float4 *>(pl.pmeta), BN);
        return check_launch("papc_lingather_bwd_f32");
    }
    const int parts = papc_lingather_parts(a.M);
    a.rows_per_wg = (int)((a.M + parts - 1) / parts);
    hipLaunchKernelGGL(lingather_bwd_kernel, dim3((unsigned)parts), dim3(LG_T), 0, st, a);
    return check_launch("papc_lingather_bwd_f32");
}

}  // extern "C"
