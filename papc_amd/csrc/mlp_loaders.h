// mlp_loaders.h -- on-the-fly A-operand producers shared by the forward / dX / dW MFMA kernels.
//
// Nothing between the layers of a relu(bn(conv1x1(.))) stack is materialised except the pre-BN conv
// outputs y_l: the grouped input rows (pointnet2_basic_layers.py:146-153), the BN+ReLU of the previous
// layer (:217) and the gradient dY of the current layer are all recomputed inside the operand load.
//
// Internal channel order of a GROUP layer is always [feats(D), xyz(3)] so the feature part is float4
// aligned; `xyz_first` (SSG order, :151) only changes the weight-column mapping gk().
#pragma once
#include "common.h"

namespace papc {

enum { A_PLAIN = PAPC_A_PLAIN, A_BNRELU = PAPC_A_BNRELU, A_GROUP = PAPC_A_GROUP, A_DY_DENSE = 3, A_DY_MAX = 4 };

struct GroupSrc {
    const float *xyz; int64_t sb, sn, sc;
    const float *new_xyz; const float *feats; const int32_t *idx;
    int N, S, K, D, xyz_first;
};

struct DySrc {
    const float *dz; const float *gout; const int32_t *argmax; int K;
    const float *y; const float *mean, *invstd, *scale, *shift, *c1, *c2;
};

struct ASrc {
    const float *x; int64_t ldx;          // PLAIN / BNRELU
    const float *sc, *sh;                 // BNRELU
    GroupSrc g;                           // GROUP
    DySrc d;                              // DY_*
    int vec;                              // float4 loads allowed (alignment + Kin % 4 == 0)
};

// internal input-channel index -> column in the caller's weight matrix
__device__ __forceinline__ int gk(const GroupSrc &g, int k) { return g.xyz_first ? (k < g.D ? k + 3 : k - g.D) : k; }

// per-row context, computed once per (thread, row tile)
struct RowCtx {
    int64_t m;     // global row
    bool valid;
    int j;         // GROUP: source point index
    int64_t b;     // GROUP: cloud
    int64_t grp;   // GROUP / DY_MAX: m / K
    int kin;       // DY_MAX: m % K
};

template <int AMODE>
__device__ __forceinline__ RowCtx make_row(const ASrc &a, int64_t m, int64_t M)
{
    RowCtx r;
    r.m = m; r.valid = m < M; r.j = 0; r.b = 0; r.grp = 0; r.kin = 0;
    if (!r.valid) return r;
    if (AMODE == A_GROUP) {
        r.grp = m / a.g.K;
        r.b = r.grp / a.g.S;
        r.j = a.g.idx ? a.g.idx[m] : (int)(m - r.b * (int64_t)a.g.S * a.g.K);
        if (r.j < 0 || r.j >= a.g.N) r.valid = false;  // no-hit sentinel N (reference raises) -> zero row
    } else if (AMODE == A_DY_MAX) {
        r.grp = m / a.d.K;
        r.kin = (int)(m - r.grp * a.d.K);
    }
    return r;
}

// per-(thread, k-chunk) constants: the thread's 4 consecutive channels k..k+3
struct KConst {
    float4 c0, c1, c2, c3, c4, c5;
};

__device__ __forceinline__ float4 ld4_or_zero(const float *p, int k, int K)
{
    float4 v;
    if (k + 3 < K) { v = *reinterpret_cast<const float4 *>(p + k); }
    else {
        v.x = k < K ? p[k] : 0.f; v.y = k + 1 < K ? p[k + 1] : 0.f; v.z = k + 2 < K ? p[k + 2] : 0.f; v.w = 0.f;
    }
    return v;
}
__device__ __forceinline__ float4 ld4s_or_zero(const float *p, int k, int K)  // scalar loads (unaligned base)
{
    float4 v;
    v.x = k < K ? p[k] : 0.f; v.y = k + 1 < K ? p[k + 1] : 0.f; v.z = k + 2 < K ? p[k + 2] : 0.f; v.w = k + 3 < K ? p[k + 3] : 0.f;
    return v;
}

template <int AMODE>
__device__ __forceinline__ KConst make_kconst(const ASrc &a, int k, int Kin)
{
    KConst c;
    c.c0 = c.c1 = c.c2 = c.c3 = c.c4 = c.c5 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k >= Kin) return c;
    if (AMODE == A_BNRELU) {
        c.c0 = ld4s_or_zero(a.sc, k, Kin); c.c1 = ld4s_or_zero(a.sh, k, Kin);
    } else if (AMODE == A_DY_DENSE || AMODE == A_DY_MAX) {
        c.c0 = ld4s_or_zero(a.d.scale, k, Kin); c.c1 = ld4s_or_zero(a.d.shift, k, Kin);
        c.c2 = ld4s_or_zero(a.d.mean, k, Kin);  c.c3 = ld4s_or_zero(a.d.invstd, k, Kin);
        c.c4 = ld4s_or_zero(a.d.c1, k, Kin);    c.c5 = ld4s_or_zero(a.d.c2, k, Kin);
    }
    return c;
}

__device__ __forceinline__ float dy_elem(float dz, float y, float sc, float sh, float mean, float invstd, float c1, float c2)
{
    // backward of relu(sc*y+sh) with train-mode BN: p = dz*[z>0]; dy = sc*(p - mean(p) - xhat*mean(p*xhat))
    const float z = fmaf(sc, y, sh);
    const float p = z > 0.f ? dz : 0.f;
    const float xhat = (y - mean) * invstd;
    return sc * ((p - c1) - xhat * c2);
}

// the thread's 4 consecutive A elements (row r, internal channels k..k+3); zero outside [0,M)x[0,Kin)
template <int AMODE>
__device__ __forceinline__ float4 load_a4(const ASrc &a, const RowCtx &r, int k, int Kin, const KConst &kc)
{
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!r.valid || k >= Kin) return v;
    if (AMODE == A_PLAIN || AMODE == A_BNRELU) {
        const float *row = a.x + r.m * a.ldx;
        v = a.vec ? ld4_or_zero(row, k, Kin) : ld4s_or_zero(row, k, Kin);
        if (AMODE == A_BNRELU) {  // relu(bn(.)) of the previous layer, folded (:217)
            v.x = fmaxf(fmaf(kc.c0.x, v.x, kc.c1.x), 0.f); v.y = fmaxf(fmaf(kc.c0.y, v.y, kc.c1.y), 0.f);
            v.z = fmaxf(fmaf(kc.c0.z, v.z, kc.c1.z), 0.f); v.w = fmaxf(fmaf(kc.c0.w, v.w, kc.c1.w), 0.f);
            if (k + 1 >= Kin) v.y = 0.f;
            if (k + 2 >= Kin) v.z = 0.f;
            if (k + 3 >= Kin) v.w = 0.f;
        }
    } else if (AMODE == A_GROUP) {
        const GroupSrc &g = a.g;
        if (k + 3 < g.D && a.vec) {
            v = *reinterpret_cast<const float4 *>(g.feats + (r.b * g.N + r.j) * (int64_t)g.D + k);
        } else {
            float e[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int kk = k + i;
                if (kk < g.D) e[i] = g.feats[(r.b * g.N + r.j) * (int64_t)g.D + kk];
                else if (kk < g.D + 3) {
                    const int c = kk - g.D;  // grouped_xyz - new_xyz (:147); group_all passes new_xyz = 0 (:170)
                    e[i] = g.xyz[r.b * g.sb + (int64_t)r.j * g.sn + c * g.sc] - g.new_xyz[r.grp * 3 + c];
                } else e[i] = 0.f;
            }
            v = make_float4(e[0], e[1], e[2], e[3]);
        }
    } else {  // A_DY_DENSE / A_DY_MAX
        const DySrc &d = a.d;
        const float4 y = a.vec ? ld4_or_zero(d.y + r.m * (int64_t)Kin, k, Kin) : ld4s_or_zero(d.y + r.m * (int64_t)Kin, k, Kin);
        float4 dz;
        if (AMODE == A_DY_DENSE) {
            dz = a.vec ? ld4_or_zero(d.dz + r.m * (int64_t)Kin, k, Kin) : ld4s_or_zero(d.dz + r.m * (int64_t)Kin, k, Kin);
        } else {
            const float *gp = d.gout + r.grp * (int64_t)Kin;
            const int32_t *ap = d.argmax + r.grp * (int64_t)Kin;
            const float4 g = a.vec ? ld4_or_zero(gp, k, Kin) : ld4s_or_zero(gp, k, Kin);
            dz.x = (ap[k] == r.kin) ? g.x : 0.f;
            dz.y = (k + 1 < Kin && ap[k + 1] == r.kin) ? g.y : 0.f;
            dz.z = (k + 2 < Kin && ap[k + 2] == r.kin) ? g.z : 0.f;
            dz.w = (k + 3 < Kin && ap[k + 3] == r.kin) ? g.w : 0.f;
        }
        v.x = dy_elem(dz.x, y.x, kc.c0.x, kc.c1.x, kc.c2.x, kc.c3.x, kc.c4.x, kc.c5.x);
        v.y = k + 1 < Kin ? dy_elem(dz.y, y.y, kc.c0.y, kc.c1.y, kc.c2.y, kc.c3.y, kc.c4.y, kc.c5.y) : 0.f;
        v.z = k + 2 < Kin ? dy_elem(dz.z, y.z, kc.c0.z, kc.c1.z, kc.c2.z, kc.c3.z, kc.c4.z, kc.c5.z) : 0.f;
        v.w = k + 3 < Kin ? dy_elem(dz.w, y.w, kc.c0.w, kc.c1.w, kc.c2.w, kc.c3.w, kc.c4.w, kc.c5.w) : 0.f;
    }
    return v;
}

}  // namespace papc
