// mlp_loaders.h -- on-the-fly A-operand producers shared by the forward / dX / dW MFMA kernels.
//
// Nothing between the layers of a relu(bn(conv1x1(.))) stack is materialised except the pre-BN conv
// outputs y_l: the grouped input rows (pointnet2_basic_layers.py:146-153), the BN+ReLU of the previous
// layer (:217) and the gradient dY of the current layer are all recomputed inside the operand load.
//
// Internal channel order of a GROUP layer is always [feats(D), xyz(3)] so the feature part is float4
// aligned; `xyz_first` (SSG order, :151) only changes the weight-column mapping gk().
//
// Two flavours per producer, chosen at compile time:
//   VEC = true : every access is an UNCONDITIONAL float4 (or int4) load from a clamped in-range address and
//                out-of-range lanes are zeroed by a select afterwards -- no branch sits around a load, so the
//                compiler batches the loads of a stage and keeps them in flight across the MFMA phase.
//                Needs 16-byte aligned bases and channel counts that are multiples of 4 (D % 4 == 0 for GROUP).
//   VEC = false: element-wise predicated loads for ragged shapes (correct for anything, slow).
// fetch_a4 only issues loads (raw registers); finish_a4 does the arithmetic once they have landed.
#pragma once
#include "common.h"

namespace papc {

enum { A_PLAIN = PAPC_A_PLAIN, A_BNRELU = PAPC_A_BNRELU, A_GROUP = PAPC_A_GROUP, A_DY_DENSE = 3, A_DY_MAX = 4, A_MAXCAT = 5, A_XYZ = PAPC_A_XYZ };
// A_XYZ (xyz1.hip): the operand is the activation of a coordinates-only first layer, recomputed from the grouped centred coordinates:
// a[m, k] = relu(wf[k][0] x + wf[k][1] y + wf[k][2] z + wf[k][3]) with (x, y, z) = xc[m] (float4 rows, a.x / a.ldx = 4) and wf = a.sc [Kin][4].
// A_MAXCAT (dX of a max-pooled last layer without reading its output y, see papc_mlp_bwd_dx_max_f32): the operand row is the
// concatenation [ P (d.C columns) | relu(bn(x)) (Kin - d.C columns) ] where P[m, c] = (m % K == argmax[m / K, c]) ? psel[m / K, c] : 0
// is the sparse max-backward gradient (psel = d.gout, already masked and scaled) and x / sc / sh are the layer's BN+ReLU input.
// d.C is a multiple of 16, so a 16-wide k stage lies entirely in one of the two regions.

// unsigned 32-bit division by an invariant (Granlund-Montgomery round-up form)
struct FastDiv {
    uint32_t mul, sh1, sh2;
};
static inline FastDiv make_fastdiv(uint32_t d)
{
    FastDiv f;
    uint32_t l = 0;
    while ((1ull << l) < d) ++l;
    f.mul = (uint32_t)((((1ull << 32) * ((1ull << l) - d)) / d) + 1);
    f.sh1 = l < 1 ? l : 1;
    f.sh2 = l - f.sh1;
    return f;
}
__device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv &f)
{
    const uint32_t t = __umulhi(n, f.mul);
    return (t + ((n - t) >> f.sh1)) >> f.sh2;
}

struct GroupSrc {
    const float *xyz; int64_t sb, sn, sc;
    const float *new_xyz; const float *feats; const int32_t *idx;
    int N, S, K, D, xyz_first;
    FastDiv divK, divS;
};

struct DySrc {
    const float *dz; const float *gout; const int32_t *argmax; int K;
    const float *y; const float *mean, *invstd, *scale, *shift, *c1, *c2;
    FastDiv divK;
    int C;  // channels of the layer (row stride of y / dz / gout / argmax)
    // compacted stack (compact.hip): per-row multiplicity weight of the BatchNorm-backward term, group of every 8-row segment (ragged
    // groups; argmax then holds absolute rows), physical row count in device memory.  All NULL for a padded stack.
    const float *wrow; const int32_t *seg_grp; const int32_t *rows_dev;
    const float *psel;    // optional (DY_MAX): scale * relu-masked gout per (group, channel) -- papc_bwd_dy::psel
};

struct ASrc {
    const float *x; int64_t ldx;          // PLAIN / BNRELU
    const float *sc, *sh;                 // BNRELU
    GroupSrc g;                           // GROUP
    DySrc d;                              // DY_*
    int vec;                              // host-side: the VEC = true flavour is legal for this operand
};

// internal input-channel index -> column in the caller's weight matrix
__device__ __forceinline__ int gk(const GroupSrc &g, int k) { return g.xyz_first ? (k < g.D ? k + 3 : k - g.D) : k; }

// per-row context (32-bit: M < 2^31 is enforced by the host entry points)
struct RowCtx {
    int m;       // global row (clamped to 0 when invalid)
    int j;       // GROUP: source point index
    int b;       // GROUP: cloud
    int grp;     // GROUP / DY_MAX: m / K
    int kin;     // DY_MAX: m % K
    bool valid;
    // row base pointers for the VEC fetch (computed once per row; a k-chunk fetch only adds its channel offset):
    //   PLAIN/BNRELU: x row;  GROUP: feats row, xyz point, centroid;  DY_*: y row, dz row | gout row, argmax row
    const float *p0, *p1;
    const void *p2;
};

// jpre >= -1: neighbour index already loaded by the caller (prefetched); jpre == -2: load it here
template <int AMODE>
__device__ __forceinline__ RowCtx make_row(const ASrc &a, int64_t m64, int64_t M, int jpre = -2)
{
    RowCtx r;
    r.valid = m64 < M;
    r.m = r.valid ? (int)m64 : 0;
    r.j = 0; r.b = 0; r.grp = 0; r.kin = 0;
    r.p0 = nullptr; r.p1 = nullptr; r.p2 = nullptr;
    if (AMODE == A_PLAIN || AMODE == A_BNRELU) {
        r.p0 = a.x + (int64_t)r.m * a.ldx;
    } else if (AMODE == A_DY_DENSE) {
        r.p0 = a.d.y + (int64_t)r.m * a.d.C;
        r.p1 = a.d.dz + (int64_t)r.m * a.d.C;
    }
    if (AMODE == A_GROUP) {
        r.grp = (int)fdiv((uint32_t)r.m, a.g.divK);
        r.b = (int)fdiv((uint32_t)r.grp, a.g.divS);
        int j;
        if (a.g.idx) j = (jpre == -2) ? a.g.idx[r.m] : jpre;
        else j = r.m - r.b * a.g.S * a.g.K;
        if (j < 0 || j >= a.g.N) { r.valid = false; j = 0; }  // no-hit sentinel N (the reference raises) -> zero row
        r.j = j;
        r.p0 = a.g.feats + ((int64_t)r.b * a.g.N + j) * a.g.D;
        r.p1 = a.g.xyz + (int64_t)r.b * a.g.sb + (int64_t)j * a.g.sn;
        r.p2 = a.g.new_xyz + (int64_t)r.grp * 3;
    } else if (AMODE == A_DY_MAX) {
        r.grp = (int)fdiv((uint32_t)r.m, a.d.divK);
        r.kin = r.m - r.grp * a.d.K;
        r.p0 = a.d.y + (int64_t)r.m * a.d.C;
        r.p1 = a.d.gout + (int64_t)r.grp * a.d.C;
        r.p2 = a.d.argmax + (int64_t)r.grp * a.d.C;
    } else if (AMODE == A_MAXCAT) {
        r.grp = (int)fdiv((uint32_t)r.m, a.d.divK);
        r.kin = r.m - r.grp * a.d.K;
        r.p0 = a.x + (int64_t)r.m * a.ldx - a.d.C;      // indexed with the concatenated k (>= d.C in the dense region)
        r.p1 = a.d.gout + (int64_t)r.grp * a.d.C;
        r.p2 = a.d.argmax + (int64_t)r.grp * a.d.C;
    }
    return r;
}

// per-(thread, channel group) constants: the thread's 4 consecutive channels k..k+3
struct KConst {
    float4 c0, c1, c2, c3, c4, c5;
};

__device__ __forceinline__ float4 ld4s_or_zero(const float *p, int k, int K)  // element-wise predicated (slow path)
{
    float4 v;
    v.x = k < K ? p[k] : 0.f; v.y = k + 1 < K ? p[k + 1] : 0.f; v.z = k + 2 < K ? p[k + 2] : 0.f; v.w = k + 3 < K ? p[k + 3] : 0.f;
    return v;
}
__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }

template <int AMODE, bool VEC>
__device__ __forceinline__ KConst make_kconst(const ASrc &a, int k, int Kin)
{
    KConst c;
    c.c0 = c.c1 = c.c2 = c.c3 = c.c4 = c.c5 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (AMODE == A_BNRELU) {
        if (VEC) { const int kk = k < Kin ? k : 0; c.c0 = ld4(a.sc + kk); c.c1 = ld4(a.sh + kk); }
        else { c.c0 = ld4s_or_zero(a.sc, k, Kin); c.c1 = ld4s_or_zero(a.sh, k, Kin); }
    } else if (AMODE == A_MAXCAT) {   // VEC only: BN+ReLU constants of the dense region (unused in the sparse one)
        const int kk = (k >= a.d.C && k < Kin) ? k - a.d.C : 0;
        c.c0 = ld4(a.sc + kk); c.c1 = ld4(a.sh + kk);
    } else if (AMODE == A_DY_DENSE || AMODE == A_DY_MAX) {
        if (VEC) {
            const int kk = k < Kin ? k : 0;
            c.c0 = ld4(a.d.scale + kk); c.c1 = ld4(a.d.shift + kk); c.c2 = ld4(a.d.mean + kk);
            c.c3 = ld4(a.d.invstd + kk); c.c4 = ld4(a.d.c1 + kk);   c.c5 = ld4(a.d.c2 + kk);
        } else {
            c.c0 = ld4s_or_zero(a.d.scale, k, Kin); c.c1 = ld4s_or_zero(a.d.shift, k, Kin);
            c.c2 = ld4s_or_zero(a.d.mean, k, Kin);  c.c3 = ld4s_or_zero(a.d.invstd, k, Kin);
            c.c4 = ld4s_or_zero(a.d.c1, k, Kin);    c.c5 = ld4s_or_zero(a.d.c2, k, Kin);
        }
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

struct Raw3 {
    float4 p;  // x / feats|xyz / y
    float4 q;  // new_xyz (GROUP xyz columns) / dz / gout
    int4 r;    // argmax (DY_MAX)
};

template <int AMODE, bool VEC>
__device__ __forceinline__ Raw3 fetch_a4(const ASrc &a, const RowCtx &r, int k, int Kin)
{
    Raw3 w;
    w.p = make_float4(0.f, 0.f, 0.f, 0.f);
    w.q = make_float4(0.f, 0.f, 0.f, 0.f);
    w.r = make_int4(-1, -1, -1, -1);
    if (VEC) {
        // unconditional loads from clamped addresses (r.m / r.j / r.grp are already clamped to 0 when invalid)
        const int kk = k < Kin ? k : 0;
        if (AMODE == A_PLAIN || AMODE == A_BNRELU) {
            w.p = ld4(r.p0 + kk);
        } else if (AMODE == A_MAXCAT) {   // one value load (psel | x) and one int4 (argmax | a valid dummy), both unconditional
            const bool sp = kk < a.d.C;
            w.p = ld4((sp ? r.p1 : r.p0) + kk);
            w.r = *reinterpret_cast<const int4 *>(reinterpret_cast<const int32_t *>(r.p2) + (sp ? kk : 0));
        } else if (AMODE == A_GROUP) {
            const GroupSrc &g = a.g;
            if (kk < g.D) {
                w.p = ld4(r.p0 + kk);
            } else {  // the xyz slot (k == D): three coordinates and the centroid they are centred on
                const float *pp = r.p1;
                const float *qq = reinterpret_cast<const float *>(r.p2);
                w.p = make_float4(pp[0], pp[g.sc], pp[2 * g.sc], 0.f);
                w.q = make_float4(qq[0], qq[1], qq[2], 0.f);
            }
        } else {
            w.p = ld4(r.p0 + kk);
            w.q = ld4(r.p1 + kk);
            if (AMODE == A_DY_MAX) w.r = *reinterpret_cast<const int4 *>(reinterpret_cast<const int32_t *>(r.p2) + kk);
        }
        return w;
    }
    // ---- slow path: element-wise predicated
    if (!r.valid || k >= Kin) return w;
    if (AMODE == A_PLAIN || AMODE == A_BNRELU) {
        w.p = ld4s_or_zero(a.x + (int64_t)r.m * a.ldx, k, Kin);
    } else if (AMODE == A_GROUP) {
        const GroupSrc &g = a.g;
        float e[4], f[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int kk = k + i;
            e[i] = 0.f; f[i] = 0.f;
            if (kk < g.D) e[i] = g.feats[((int64_t)r.b * g.N + r.j) * g.D + kk];
            else if (kk < g.D + 3) {
                const int c = kk - g.D;
                e[i] = g.xyz[(int64_t)r.b * g.sb + (int64_t)r.j * g.sn + c * g.sc];
                f[i] = g.new_xyz[(int64_t)r.grp * 3 + c];
            }
        }
        w.p = make_float4(e[0], e[1], e[2], e[3]);
        w.q = make_float4(f[0], f[1], f[2], f[3]);
    } else {
        const DySrc &d = a.d;
        w.p = ld4s_or_zero(d.y + (int64_t)r.m * Kin, k, Kin);
        if (AMODE == A_DY_DENSE) {
            w.q = ld4s_or_zero(d.dz + (int64_t)r.m * Kin, k, Kin);
        } else {
            const int32_t *ap = d.argmax + (int64_t)r.grp * Kin;
            w.q = ld4s_or_zero(d.gout + (int64_t)r.grp * Kin, k, Kin);
            w.r.x = ap[k];
            w.r.y = k + 1 < Kin ? ap[k + 1] : -1;
            w.r.z = k + 2 < Kin ? ap[k + 2] : -1;
            w.r.w = k + 3 < Kin ? ap[k + 3] : -1;
        }
    }
    return w;
}

template <int AMODE, bool VEC>
__device__ __forceinline__ float4 finish_a4(const ASrc &a, const RowCtx &r, int k, int Kin, const KConst &kc, const Raw3 &w)
{
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool ok = r.valid && k < Kin;
    // element masks inside the float4 (only the slow path can straddle Kin; VEC shapes have Kin % 4 == 0,
    // except the GROUP xyz slot whose 4th element is a structural zero)
    const bool ky = VEC || k + 1 < Kin, kz = VEC || k + 2 < Kin, kw = VEC || k + 3 < Kin;
    if (AMODE == A_PLAIN) {
        v = w.p;
    } else if (AMODE == A_BNRELU) {  // relu(bn(.)) of the previous layer, folded (:217)
        v.x = fmaxf(fmaf(kc.c0.x, w.p.x, kc.c1.x), 0.f);
        v.y = ky ? fmaxf(fmaf(kc.c0.y, w.p.y, kc.c1.y), 0.f) : 0.f;
        v.z = kz ? fmaxf(fmaf(kc.c0.z, w.p.z, kc.c1.z), 0.f) : 0.f;
        v.w = kw ? fmaxf(fmaf(kc.c0.w, w.p.w, kc.c1.w), 0.f) : 0.f;
    } else if (AMODE == A_MAXCAT) {
        if (k < a.d.C) {              // uniform over the stage
            v.x = (w.r.x == r.kin) ? w.p.x : 0.f; v.y = (w.r.y == r.kin) ? w.p.y : 0.f;
            v.z = (w.r.z == r.kin) ? w.p.z : 0.f; v.w = (w.r.w == r.kin) ? w.p.w : 0.f;
        } else {
            v.x = fmaxf(fmaf(kc.c0.x, w.p.x, kc.c1.x), 0.f); v.y = fmaxf(fmaf(kc.c0.y, w.p.y, kc.c1.y), 0.f);
            v.z = fmaxf(fmaf(kc.c0.z, w.p.z, kc.c1.z), 0.f); v.w = fmaxf(fmaf(kc.c0.w, w.p.w, kc.c1.w), 0.f);
        }
    } else if (AMODE == A_GROUP) {   // feats pass through (q = 0: p - 0 is exact); xyz columns: grouped_xyz - new_xyz (:147)
        v.x = w.p.x - w.q.x; v.y = w.p.y - w.q.y; v.z = w.p.z - w.q.z; v.w = w.p.w - w.q.w;
    } else {
        float4 dz = w.q;
        if (AMODE == A_DY_MAX) {
            dz.x = (w.r.x == r.kin) ? w.q.x : 0.f; dz.y = (w.r.y == r.kin) ? w.q.y : 0.f;
            dz.z = (w.r.z == r.kin) ? w.q.z : 0.f; dz.w = (w.r.w == r.kin) ? w.q.w : 0.f;
        }
        v.x = dy_elem(dz.x, w.p.x, kc.c0.x, kc.c1.x, kc.c2.x, kc.c3.x, kc.c4.x, kc.c5.x);
        v.y = ky ? dy_elem(dz.y, w.p.y, kc.c0.y, kc.c1.y, kc.c2.y, kc.c3.y, kc.c4.y, kc.c5.y) : 0.f;
        v.z = kz ? dy_elem(dz.z, w.p.z, kc.c0.z, kc.c1.z, kc.c2.z, kc.c3.z, kc.c4.z, kc.c5.z) : 0.f;
        v.w = kw ? dy_elem(dz.w, w.p.w, kc.c0.w, kc.c1.w, kc.c2.w, kc.c3.w, kc.c4.w, kc.c5.w) : 0.f;
    }
    if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
    return v;
}

// ---- fp32 on the bf16 matrix pipe: exact 3-way split.
// x = x1 + x2 + x3 with x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2) (round-to-nearest-even each; the
// subtractions are exact in fp32).  Three 8-bit significands cover the 24-bit fp32 significand, so the split is exact
// (up to underflow of the tails), and a product a*b is recovered as the six bf16 products
//   a1*b1 + (a1*b2 + a2*b1) + (a1*b3 + a2*b2 + a3*b1)
// accumulated in fp32 by v_mfma_f32_32x32x16_bf16; the dropped terms (a2*b3, a3*b2, a3*b3) are below 2^-26 |a*b|,
// i.e. under the rounding error of a single fp32 multiply-add.  6 MFMAs of 32 cycles per 32x32x16 block against
// 8 x 64 cycles of v_mfma_f32_32x32x2_f32 for the same block: 2.7x the fp32 matrix rate at fp32 accuracy
// (tools/probe/mfma_bf16_layout.hip measures 6.5e-8 max |err| / sum|a_k b_k| vs 1.5e-7 for the fp32 fma chain).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float floatx2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack_bf16x2(float a, float b)  // -> v_cvt_pk_bf16_f32 (a in the low half)
{
    return __builtin_bit_cast(unsigned, __builtin_convertvector((floatx2_t){a, b}, bf16x2_t));
}
__device__ __forceinline__ float bf16_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

// planes p0 (leading), p1, p2 of four consecutive-k values, each as 4 packed bf16 (8 bytes)
#ifndef PAPC_SPLIT_PK
#define PAPC_SPLIT_PK 0   // 1: the two exact subtractions of a pair as one v_pk_add_f32 -- measured SLOWER in the LDS-staged kernels (dW family 0.85 -> 0.90 ms/step: the producers pay moves to form aligned register pairs); the row-streaming kernel, whose pairs are natural, uses its own split3_pair
#endif
__device__ __forceinline__ void split3(float4 v, uint2 &p0, uint2 &p1, uint2 &p2)
{
#if PAPC_SPLIT_PK
    floatx2_t a = {v.x, v.y}, b = {v.z, v.w};
    p0.x = pack_bf16x2(a.x, a.y); p0.y = pack_bf16x2(b.x, b.y);
    a = a - floatx2_t{bf16_lo(p0.x), bf16_hi(p0.x)}; b = b - floatx2_t{bf16_lo(p0.y), bf16_hi(p0.y)};
    p1.x = pack_bf16x2(a.x, a.y); p1.y = pack_bf16x2(b.x, b.y);
    a = a - floatx2_t{bf16_lo(p1.x), bf16_hi(p1.x)}; b = b - floatx2_t{bf16_lo(p1.y), bf16_hi(p1.y)};
    p2.x = pack_bf16x2(a.x, a.y); p2.y = pack_bf16x2(b.x, b.y);
#else
    p0.x = pack_bf16x2(v.x, v.y); p0.y = pack_bf16x2(v.z, v.w);
    v.x -= bf16_lo(p0.x); v.y -= bf16_hi(p0.x); v.z -= bf16_lo(p0.y); v.w -= bf16_hi(p0.y);
    p1.x = pack_bf16x2(v.x, v.y); p1.y = pack_bf16x2(v.z, v.w);
    v.x -= bf16_lo(p1.x); v.y -= bf16_hi(p1.x); v.z -= bf16_lo(p1.y); v.w -= bf16_hi(p1.y);
    p2.x = pack_bf16x2(v.x, v.y); p2.y = pack_bf16x2(v.z, v.w);
#endif
}

}  // namespace papc
