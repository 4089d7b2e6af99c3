// pfn.hip -- PointPillars PillarFeatureNet with one (last) PFNLayer, forward and backward, for gfx950.
//
// Reference: /root/reference/PAPC/models/detect/pointpillars/models/bones/pillars.py
//   PillarFeatureNet.forward :79-108 (cluster mean :82, f_cluster :83, f_center :86-88, concat :91-95, padding
//   mask :99-102) -> PFNLayer.forward :29-37 (Linear no bias :30, BatchNorm1D train eps=1e-3 :31, ReLU :32, max
//   over the T points :34, last layer returns the max :36-37).
//
// One 64-lane wave per pillar.  Phase A: lanes = points (coalesced float4 loads of [T,4]); the decorated, masked
// 9-channel rows go to LDS.  Phase B: lanes = output channels (C <= 64): each lane keeps its weight row w[c,0:9]
// in VGPRs and walks the T rows with broadcast ds_read_b128 -- K = 9 is too short for MFMA to pay (it would pad to
// 128 rows x 10), the kernel is VALU/HBM balanced.  Train-mode BN needs grid-wide statistics, so forward is two
// passes (stats, then apply+ReLU+max) that both recompute the 9->C linear layer instead of storing [P,T,C].
#include "common.h"

namespace papc {

constexpr int PFN_WAVES = 4;
constexpr int PFN_TMAX = 128;  // points per pillar supported by the LDS staging
constexpr int PFN_LD = 12;     // LDS row stride (9 channels padded to 3 x float4)

struct PfnArgs {
    const float *feat; const int32_t *nvox; const int32_t *coors; int P, T;
    float vx, vy, xo, yo;
    const float *w; int C;
    // forward
    float *stats; const float *scale, *shift; float *out; int32_t *argmax;
    // backward
    const float *gout; const int32_t *amax; const float *mean, *invstd, *c1, *c2;
    float *red; float *dwp;
    int with_dist;   // PFN_DECORATE only: also write ||xyz|| as a 10th channel (pillars.py:92-94)
};

enum { PFN_STATS = 0, PFN_APPLY = 1, PFN_BWD_RED = 2, PFN_BWD_DW = 3, PFN_DECORATE = 4 };

// Phase A for one pillar: decorate + mask, rows -> LDS (this wave's slab)
__device__ __forceinline__ void pfn_stage(const PfnArgs &a, int p, float *rows, int lane)
{
    const int T = a.T;
    const int nv = a.nvox[p];
    float4 f[2];
    float sx = 0.f, sy = 0.f, sz = 0.f;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int t = lane + 64 * h;
        f[h] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t < T) f[h] = *reinterpret_cast<const float4 *>(a.feat + ((int64_t)p * T + t) * 4);
        sx += f[h].x; sy += f[h].y; sz += f[h].z;
    }
    // features[:, :, :3].sum(axis=1) over ALL T rows (zero padding included) / num_voxels   (:82)
    sx = readlane63_f32(wave_sum_f32_to_lane63(sx));
    sy = readlane63_f32(wave_sum_f32_to_lane63(sy));
    sz = readlane63_f32(wave_sum_f32_to_lane63(sz));
    const float fn = (float)nv;
    const float mx = sx / fn, my = sy / fn, mz = sz / fn;
    const float pcx = (float)a.coors[(int64_t)p * 4 + 3] * a.vx + a.xo;  // coors[:,3]*vx + x_offset  (:87)
    const float pcy = (float)a.coors[(int64_t)p * 4 + 2] * a.vy + a.yo;  // coors[:,2]*vy + y_offset  (:88)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int t = lane + 64 * h;
        if (t < T) {
            const float mk = t < nv ? 1.f : 0.f;  // get_paddings_indicator: actual_num > t  (libs/tools/__init__.py:26-35)
            float *r = rows + t * PFN_LD;
            const float4 v = f[h];
            *reinterpret_cast<float4 *>(r) = make_float4(v.x * mk, v.y * mk, v.z * mk, v.w * mk);
            *reinterpret_cast<float4 *>(r + 4) = make_float4((v.x - mx) * mk, (v.y - my) * mk, (v.z - mz) * mk, (v.x - pcx) * mk);
            // (slot 9: points_dist = paddle.norm(features[:, :, :3], 2, 2) of the with_distance variant, :92-94)
            *reinterpret_cast<float4 *>(r + 8) = make_float4((v.y - pcy) * mk, sqrtf((v.x * v.x + v.y * v.y) + v.z * v.z) * mk, 0.f, 0.f);
        }
    }
}

__device__ __forceinline__ float pfn_dot(const float *r, const float (&w)[9])
{
    const float4 a = *reinterpret_cast<const float4 *>(r);
    const float4 b = *reinterpret_cast<const float4 *>(r + 4);
    const float c = r[8];
    float y = a.x * w[0];
    y = fmaf(a.y, w[1], y); y = fmaf(a.z, w[2], y); y = fmaf(a.w, w[3], y);
    y = fmaf(b.x, w[4], y); y = fmaf(b.y, w[5], y); y = fmaf(b.z, w[6], y); y = fmaf(b.w, w[7], y);
    y = fmaf(c, w[8], y);
    return y;
}

template <int MODE>
__global__ __launch_bounds__(64 * PFN_WAVES) void pfn_kernel(PfnArgs a)
{
    __shared__ __attribute__((aligned(16))) float smem[PFN_WAVES * PFN_TMAX * PFN_LD + PFN_WAVES * 64 * 10];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *rows = smem + wave * PFN_TMAX * PFN_LD;
    float *xred = smem + PFN_WAVES * PFN_TMAX * PFN_LD;  // [wave][10][64]
    const int c = lane;
    const bool cok = c < a.C;

    float w[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) w[k] = (cok && MODE != PFN_DECORATE) ? a.w[c * 9 + k] : 0.f;
    float sc = 0.f, sh = 0.f, mu = 0.f, is = 0.f, c1 = 0.f, c2 = 0.f;
    if (MODE != PFN_STATS && MODE != PFN_DECORATE && cok) { sc = a.scale[c]; sh = a.shift[c]; }
    if ((MODE == PFN_BWD_RED || MODE == PFN_BWD_DW) && cok) { mu = a.mean[c]; is = a.invstd[c]; }
    if (MODE == PFN_BWD_DW && cok) { c1 = a.c1[c]; c2 = a.c2[c]; }

    float acc[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) acc[i] = 0.f;

    for (int p = blockIdx.x * PFN_WAVES + wave; p < a.P; p += gridDim.x * PFN_WAVES) {
        pfn_stage(a, p, rows, lane);
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (MODE == PFN_DECORATE) {
            const int nc = a.with_dist ? 10 : 9;
            for (int e = lane; e < a.T * nc; e += 64) {
                const int t = e / nc, k = e - t * nc;
                a.out[((int64_t)p * a.T + t) * nc + k] = rows[t * PFN_LD + k];
            }
        } else if (MODE == PFN_STATS) {
            for (int t = 0; t < a.T; ++t) {
                const float y = pfn_dot(rows + t * PFN_LD, w);
                acc[0] += y;
                acc[1] = fmaf(y, y, acc[1]);
            }
        } else if (MODE == PFN_APPLY) {
            float best = -1.f;
            int bi = 0;
            for (int t = 0; t < a.T; ++t) {
                const float y = pfn_dot(rows + t * PFN_LD, w);
                const float z = fmaxf(fmaf(sc, y, sh), 0.f);
                if (z > best) { best = z; bi = t; }
            }
            if (cok) {
                a.out[(int64_t)p * a.C + c] = best;
                if (a.argmax) a.argmax[(int64_t)p * a.C + c] = bi;
            }
        } else if (MODE == PFN_BWD_RED) {
            if (cok) {
                const int am = a.amax[(int64_t)p * a.C + c];
                const float y = pfn_dot(rows + am * PFN_LD, w);
                const float z = fmaf(sc, y, sh);
                const float g = z > 0.f ? a.gout[(int64_t)p * a.C + c] : 0.f;
                acc[0] += g;
                acc[1] = fmaf(g, (y - mu) * is, acc[1]);
            }
        } else {  // PFN_BWD_DW
            const int am = cok ? a.amax[(int64_t)p * a.C + c] : -1;
            const float g = cok ? a.gout[(int64_t)p * a.C + c] : 0.f;
            for (int t = 0; t < a.T; ++t) {
                const float *r = rows + t * PFN_LD;
                const float y = pfn_dot(r, w);
                const float z = fmaf(sc, y, sh);
                const float pp = (t == am && z > 0.f) ? g : 0.f;
                const float dy = sc * ((pp - c1) - ((y - mu) * is) * c2);
                const float4 x0 = *reinterpret_cast<const float4 *>(r);
                const float4 x1 = *reinterpret_cast<const float4 *>(r + 4);
                acc[0] = fmaf(dy, x0.x, acc[0]); acc[1] = fmaf(dy, x0.y, acc[1]); acc[2] = fmaf(dy, x0.z, acc[2]);
                acc[3] = fmaf(dy, x0.w, acc[3]); acc[4] = fmaf(dy, x1.x, acc[4]); acc[5] = fmaf(dy, x1.y, acc[5]);
                acc[6] = fmaf(dy, x1.z, acc[6]); acc[7] = fmaf(dy, x1.w, acc[7]); acc[8] = fmaf(dy, r[8], acc[8]);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }

    if (MODE == PFN_APPLY || MODE == PFN_DECORATE) return;
    constexpr int NA = (MODE == PFN_BWD_DW) ? 9 : 2;
#pragma unroll
    for (int i = 0; i < NA; ++i) xred[(wave * 10 + i) * 64 + lane] = acc[i];
    __syncthreads();
    if (wave == 0 && cok) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            float s = 0.f;
#pragma unroll
            for (int g = 0; g < PFN_WAVES; ++g) s += xred[(g * 10 + i) * 64 + lane];
            if (MODE == PFN_BWD_DW) a.dwp[((int64_t)blockIdx.x * a.C + c) * 9 + i] = s;
            else if (MODE == PFN_STATS) a.stats[((int64_t)blockIdx.x * 2 + i) * a.C + c] = s;
            else a.red[((int64_t)blockIdx.x * 2 + i) * a.C + c] = s;
        }
    }
}

static int pfn_blocks(int P) { return (int)std::min<int64_t>(cdiv(P, PFN_WAVES), 1024); }

static int pfn_check(const char *who, const float *features, const int32_t *nv, const int32_t *coors, int P, int T,
                     const float *w, int C)
{
    PAPC_REQUIRE(features && nv && coors && w, PAPC_E_INVALID, "%s: null pointer", who);
    PAPC_REQUIRE(P >= 1 && T >= 1, PAPC_E_INVALID, "%s: P=%d T=%d must be >= 1", who, P, T);
    PAPC_REQUIRE(T <= PFN_TMAX, PAPC_E_UNSUPPORTED, "%s: T=%d > %d points per pillar", who, T, PFN_TMAX);
    PAPC_REQUIRE(C >= 1 && C <= 64, PAPC_E_UNSUPPORTED, "%s: C=%d not in [1,64]", who, C);
    PAPC_REQUIRE(aligned16(features), PAPC_E_INVALID, "%s: features must be 16-byte aligned", who);
    return PAPC_OK;
}

template <int MODE>
static int launch_pfn(const PfnArgs &a, hipStream_t st, const char *who)
{
    ProfScope prof(PAPC_K_PFN, st);
    hipLaunchKernelGGL(pfn_kernel<MODE>, dim3(pfn_blocks(a.P)), dim3(64 * PFN_WAVES), 0, st, a);
    return check_launch(who);
}

}  // namespace papc

using namespace papc;

extern "C" {

int papc_pfn_num_blocks(int P) { return P >= 1 ? pfn_blocks(P) : 0; }

int papc_pfn_decorate_f32(const float *features, const int32_t *num_voxels, const int32_t *coors, int P, int T,
                          float vx, float vy, float x_offset, float y_offset, int with_distance, float *out, papc_stream_t stream)
{
    const float dummy = 0.f;
    int rc = pfn_check("papc_pfn_decorate_f32", features, num_voxels, coors, P, T, &dummy, 1);
    if (rc) return rc;
    PAPC_REQUIRE(out, PAPC_E_INVALID, "papc_pfn_decorate_f32: null out");
    PfnArgs a;
    memset(&a, 0, sizeof(a));
    a.feat = features; a.nvox = num_voxels; a.coors = coors; a.P = P; a.T = T; a.vx = vx; a.vy = vy; a.xo = x_offset; a.yo = y_offset;
    a.with_dist = with_distance ? 1 : 0;
    a.C = 1; a.out = out;
    return launch_pfn<PFN_DECORATE>(a, as_stream(stream), "papc_pfn_decorate_f32");
}

int papc_pfn_stats_f32(const float *features, const int32_t *num_voxels, const int32_t *coors, int P, int T,
                       float vx, float vy, float x_offset, float y_offset, const float *w, int C,
                       float *stats_partial, int *n_blocks_out, papc_stream_t stream)
{
    int rc = pfn_check("papc_pfn_stats_f32", features, num_voxels, coors, P, T, w, C);
    if (rc) return rc;
    PAPC_REQUIRE(stats_partial, PAPC_E_INVALID, "papc_pfn_stats_f32: null stats_partial");
    PfnArgs a;
    memset(&a, 0, sizeof(a));
    a.feat = features; a.nvox = num_voxels; a.coors = coors; a.P = P; a.T = T; a.vx = vx; a.vy = vy; a.xo = x_offset; a.yo = y_offset;
    a.w = w; a.C = C; a.stats = stats_partial;
    if (n_blocks_out) *n_blocks_out = pfn_blocks(P);
    return launch_pfn<PFN_STATS>(a, as_stream(stream), "papc_pfn_stats_f32");
}

int papc_pfn_apply_f32(const float *features, const int32_t *num_voxels, const int32_t *coors, int P, int T,
                       float vx, float vy, float x_offset, float y_offset, const float *w, int C,
                       const float *scale, const float *shift, float *out, int32_t *argmax, papc_stream_t stream)
{
    int rc = pfn_check("papc_pfn_apply_f32", features, num_voxels, coors, P, T, w, C);
    if (rc) return rc;
    PAPC_REQUIRE(scale && shift && out, PAPC_E_INVALID, "papc_pfn_apply_f32: null pointer");
    PfnArgs a;
    memset(&a, 0, sizeof(a));
    a.feat = features; a.nvox = num_voxels; a.coors = coors; a.P = P; a.T = T; a.vx = vx; a.vy = vy; a.xo = x_offset; a.yo = y_offset;
    a.w = w; a.C = C; a.scale = scale; a.shift = shift; a.out = out; a.argmax = argmax;
    return launch_pfn<PFN_APPLY>(a, as_stream(stream), "papc_pfn_apply_f32");
}

int papc_pfn_bwd_reduce_f32(const float *features, const int32_t *num_voxels, const int32_t *coors, int P,
                            int T, float vx, float vy, float x_offset, float y_offset, const float *w, int C,
                            const float *gout, const int32_t *argmax, const float *mean, const float *invstd,
                            const float *scale, const float *shift, float *red_partial, papc_stream_t stream)
{
    int rc = pfn_check("papc_pfn_bwd_reduce_f32", features, num_voxels, coors, P, T, w, C);
    if (rc) return rc;
    PAPC_REQUIRE(gout && argmax && mean && invstd && scale && shift && red_partial, PAPC_E_INVALID, "papc_pfn_bwd_reduce_f32: null pointer");
    PfnArgs a;
    memset(&a, 0, sizeof(a));
    a.feat = features; a.nvox = num_voxels; a.coors = coors; a.P = P; a.T = T; a.vx = vx; a.vy = vy; a.xo = x_offset; a.yo = y_offset;
    a.w = w; a.C = C; a.gout = gout; a.amax = argmax; a.mean = mean; a.invstd = invstd; a.scale = scale; a.shift = shift; a.red = red_partial;
    return launch_pfn<PFN_BWD_RED>(a, as_stream(stream), "papc_pfn_bwd_reduce_f32");
}

int papc_pfn_bwd_dw_f32(const float *features, const int32_t *num_voxels, const int32_t *coors, int P, int T,
                        float vx, float vy, float x_offset, float y_offset, const float *w, int C,
                        const float *gout, const int32_t *argmax, const float *mean, const float *invstd,
                        const float *scale, const float *shift, const float *c1, const float *c2,
                        float *dw_partial, papc_stream_t stream)
{
    int rc = pfn_check("papc_pfn_bwd_dw_f32", features, num_voxels, coors, P, T, w, C);
    if (rc) return rc;
    PAPC_REQUIRE(gout && argmax && mean && invstd && scale && shift && c1 && c2 && dw_partial, PAPC_E_INVALID, "papc_pfn_bwd_dw_f32: null pointer");
    PfnArgs a;
    memset(&a, 0, sizeof(a));
    a.feat = features; a.nvox = num_voxels; a.coors = coors; a.P = P; a.T = T; a.vx = vx; a.vy = vy; a.xo = x_offset; a.yo = y_offset;
    a.w = w; a.C = C; a.gout = gout; a.amax = argmax; a.mean = mean; a.invstd = invstd; a.scale = scale; a.shift = shift;
    a.c1 = c1; a.c2 = c2; a.dwp = dw_partial;
    return launch_pfn<PFN_BWD_DW>(a, as_stream(stream), "papc_pfn_bwd_dw_f32");
}

}  // extern "C"
