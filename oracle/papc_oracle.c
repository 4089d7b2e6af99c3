/*
 * papc_oracle.c -- CPU restatement of the AgentMaker/PAPC hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This file is the checker, never the product: only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it.  The shipped path is the HIP library in
 * papc_amd/csrc and fails loudly when that library is missing.
 *
 * PARITY UNPINNED: the reference (PAPC, pure Python on PaddlePaddle) has no tests, golden
 * vectors or fixtures for this path, and PaddlePaddle is not installable here, so nothing in
 * this file could be checked against an execution of the reference itself.  What it follows
 * is the reference *source*, function by function (citations below are relative to
 * /root/reference/PAPC/models/layers/pointnet2_basic_layers.py unless stated), with the one
 * thing the source leaves to the tensor runtime -- fp32 rounding order -- fixed to the
 * canonical arithmetic written out in SURVEY.md section 8a:
 *
 *   dot(a,b)  = fmaf(a2,b2, fmaf(a1,b1, a0*b0))          (k-ordered FMA chain, as sgemm K=3)
 *   |a|^2     = (a0*a0 + a1*a1) + a2*a2                    (rounded squares, left-to-right)
 *   sqdist    = ((-2*dot) + |a|^2) + |b|^2                 (square_distance :36-38)
 *   fps dist  = (dx*dx + dy*dy) + dz*dz                    (farthest_point_sample :86)
 *
 * Compile with -ffp-contract=off so the compiler adds no FMA of its own.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- square_distance(src, dst)  :26-40 -------------------------------------------- */
static inline float orc_norm2(const float *a) { return (a[0] * a[0] + a[1] * a[1]) + a[2] * a[2]; }

static inline float orc_sqdist(const float *a, float aa, const float *b, float bb)
{
    float dot = fmaf(a[2], b[2], fmaf(a[1], b[1], a[0] * b[0])); /* paddle.matmul :36 */
    float d = -2.0f * dot;                                         /* -2 * matmul   :36 */
    d = d + aa;                                                    /* += sum(src**2) :37 */
    d = d + bb;                                                    /* += sum(dst**2) :38 */
    return d;
}

/* src [B,N,3], dst [B,M,3] -> out [B,N,M] */
void orc_square_distance(const float *src, const float *dst, int B, int N, int M, float *out)
{
    for (int b = 0; b < B; ++b)
        for (int i = 0; i < N; ++i) {
            const float *a = src + ((size_t)b * N + i) * 3;
            float aa = orc_norm2(a);
            for (int j = 0; j < M; ++j) {
                const float *p = dst + ((size_t)b * M + j) * 3;
                out[((size_t)b * N + i) * M + j] = orc_sqdist(a, aa, p, orc_norm2(p));
            }
        }
}

/* ---- farthest_point_sample(xyz, npoint)  :65-95 ---------------------------------------
 * xyz [B,N,3]; start[b] replaces paddle.randint (:76); init_dist is 1.0 in the reference (:75).
 * distance = where(dist < distance, dist, distance)  (:87-92, strict <, NaN never replaces)
 * farthest = argmax(distance)  (:93; lowest index on ties)                              */
void orc_fps(const float *xyz, int B, int N, int npoint, const int64_t *start, float init_dist,
             int32_t *out)
{
    float *dist = (float *)malloc(sizeof(float) * (size_t)N);
    for (int b = 0; b < B; ++b) {
        const float *p = xyz + (size_t)b * N * 3;
        for (int i = 0; i < N; ++i) dist[i] = init_dist;
        int64_t far = start[b];
        for (int it = 0; it < npoint; ++it) {
            out[(size_t)b * npoint + it] = (int32_t)far; /* centroids[:, i] = farthest :80 */
            float cx = p[far * 3 + 0], cy = p[far * 3 + 1], cz = p[far * 3 + 2];
            float best = -INFINITY;
            int64_t besti = 0;
            for (int i = 0; i < N; ++i) {
                float dx = p[i * 3 + 0] - cx, dy = p[i * 3 + 1] - cy, dz = p[i * 3 + 2] - cz;
                float d = (dx * dx + dy * dy) + dz * dz; /* sum((xyz-centroid)**2,-1) :86 */
                if (d < dist[i]) dist[i] = d;            /* :87-92 */
                if (dist[i] > best) { best = dist[i]; besti = i; } /* argmax, first max :93 */
            }
            far = besti;
        }
    }
    free(dist);
}

/* ---- query_ball_point(radius, nsample, xyz, new_xyz)  :98-126 -------------------------
 * Net semantics of tile/mask/sort/pad (:110-124): the first `nsample` indices j, ascending,
 * with !(sqdist(new_xyz_s, xyz_j) > thr); remaining slots take the first hit; with no hit
 * every slot is N.  thr = (float)((double)radius*(double)radius)  (python scalar radius**2
 * promoted to the tensor dtype, :112).  xyz [B,N,3], new_xyz [B,S,3] -> out [B,S,nsample].   */
void orc_ball_query(const float *xyz, const float *new_xyz, int B, int N, int S, float thr,
                    int nsample, int64_t *out)
{
    float *bb = (float *)malloc(sizeof(float) * (size_t)N);
    for (int b = 0; b < B; ++b) {
        const float *p = xyz + (size_t)b * N * 3;
        for (int j = 0; j < N; ++j) bb[j] = orc_norm2(p + j * 3);
        for (int s = 0; s < S; ++s) {
            const float *q = new_xyz + ((size_t)b * S + s) * 3;
            float aa = orc_norm2(q);
            int64_t *o = out + ((size_t)b * S + s) * nsample;
            int cnt = 0;
            for (int j = 0; j < N && cnt < nsample; ++j) {
                float d = orc_sqdist(q, aa, p + j * 3, bb[j]);
                if (!(d > thr)) o[cnt++] = j; /* mask = sqrdists > radius**2 :112 */
            }
            int64_t first = cnt ? o[0] : (int64_t)N;
            for (int k = cnt; k < nsample; ++k) o[k] = first; /* :118-124 */
        }
    }
    free(bb);
}

/* ---- index_points(points, idx)  :43-62 -------------------------------------------------
 * points [B,N,C], idx [B,S] (flattened trailing dims) -> out [B,S,C]                       */
void orc_index_points(const float *points, const int64_t *idx, int B, int N, int C, int S, float *out)
{
    for (int b = 0; b < B; ++b)
        for (int s = 0; s < S; ++s) {
            int64_t j = idx[(size_t)b * S + s];
            memcpy(out + ((size_t)b * S + s) * C, points + ((size_t)b * N + j) * C, sizeof(float) * C);
        }
}

/* ---- 1x1 conv on rows: y[m,o] = (sum_k x[m,k]*w[o,k]) + bias[o]  (nn.Conv2D(cin,cout,1) :189,217)
 * canonical k-ordered fmaf chain from 0 (== v_mfma_f32_32x32x2_f32 accumulation), bias added last.
 * x [M,Cin], w [Cout,Cin], bias [Cout] or NULL -> y [M,Cout]                                  */
void orc_conv1x1(const float *x, const float *w, const float *bias, int64_t M, int Cin, int Cout, float *y)
{
    for (int64_t m = 0; m < M; ++m) {
        const float *xr = x + m * Cin;
        for (int o = 0; o < Cout; ++o) {
            const float *wr = w + (size_t)o * Cin;
            float acc = 0.0f;
            for (int k = 0; k < Cin; ++k) acc = fmaf(xr[k], wr[k], acc);
            y[m * Cout + o] = bias ? acc + bias[o] : acc;
        }
    }
}

/* per-channel batch statistics in double: mean[c], biased var[c] over M rows (BatchNorm2D train, :190) */
void orc_bn_stats(const float *y, int64_t M, int C, double *mean, double *var)
{
    for (int c = 0; c < C; ++c) { mean[c] = 0.0; var[c] = 0.0; }
    for (int64_t m = 0; m < M; ++m)
        for (int c = 0; c < C; ++c) mean[c] += (double)y[m * C + c];
    for (int c = 0; c < C; ++c) mean[c] /= (double)M;
    for (int64_t m = 0; m < M; ++m)
        for (int c = 0; c < C; ++c) { double d = (double)y[m * C + c] - mean[c]; var[c] += d * d; }
    for (int c = 0; c < C; ++c) var[c] /= (double)M;
}

/* batched a @ b^T with the canonical k-ordered chain: a [B,N,C], b [B,M,C] -> out [B,N,M]
 * (stands in for paddle.matmul(src, dst.transpose([0,2,1])) at :36)                        */
void orc_matmul_nt(const float *a, const float *b, int B, int N, int M, int C, float *out)
{
    for (int bi = 0; bi < B; ++bi)
        for (int i = 0; i < N; ++i) {
            const float *ar = a + ((size_t)bi * N + i) * C;
            for (int j = 0; j < M; ++j) {
                const float *br = b + ((size_t)bi * M + j) * C;
                float acc = ar[0] * br[0];
                for (int k = 1; k < C; ++k) acc = fmaf(ar[k], br[k], acc);
                out[((size_t)bi * N + i) * M + j] = acc;
            }
        }
}
