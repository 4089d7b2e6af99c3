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
//
// The Gram path (papc_pfn_gram_f32 & co, the one PillarFeatureNet uses): with K = 9 input channels everything the train-mode
// BatchNorm and the weight gradient need from the DENSE [P*T, C] activations is a function of the inputs' 10x10 Gram matrix
//   G = sum_rows [x | 1]^T [x | 1]        (x = the decorated, masked 9-channel row)
// because y = x W^T:  sum y_c = W_c . colsum,  sum y_c^2 = W_c G W_c^T,  sum y_c x_k = (W G)_ck.  So
//   forward statistics:  mean_c = W_c.colsum / M,  var_c = W_c G W_c^T / M - mean_c^2              (no 64-channel pass at all)
//   backward:            dW_ck = sc_c ( T_ck - c1_c colsum_k - c2_c invstd_c ((W G)_ck - mean_c colsum_k) )
// with T_ck = sum over (pillar, channel) of the max-pooled gradient times the argmax row's x_k -- sparse: one row per (pillar, c).
// G is accumulated in float64 on the matrix pipe (v_mfma_f64_16x16x4_f64: products of fp32 values are exact in f64), since the
// quadratic forms cancel: raw coordinates are O(70 m), a channel's spread O(1).  The only dense passes left are the Gram pass
// (reads the 19 MB input once) and the apply pass.
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
    double *gram;    // PFN_GRAM: [gridDim.x][256] partial Gram matrices
    int with_dist;   // PFN_DECORATE only: also write ||xyz|| as a 10th channel (pillars.py:92-94)
    int zero_padded; // the caller states that rows t >= num_voxels of `features` are zero (the reference's voxeliser zero-initialises its buffers,
                     // libs/ops/point_cloud/point_cloud_ops.py:148): only the real rows are loaded
    // PFN_GRAM with ticket != NULL: the workgroup that finishes last folds the partials and writes the BatchNorm constants (pfn_gram_finalize_body)
    unsigned *ticket; const float *gamma, *beta; float eps, momentum; double Mrows; int Cfin;
    float *o_mean, *o_invstd, *o_scale, *o_shift, *rmean, *rvar; double *gram_out;
};

// ---- last-arriving workgroup ------------------------------------------------------------------------------------------
// The Gram pass and the backward fold end in a one-workgroup tail (BatchNorm constants / dW from folded sums).  As launches of their own
// these tails are 5-9 us of launch + dependency latency around ~3 us of work; here the workgroup that finishes LAST runs them: every
// workgroup publishes its partial row, takes a ticket, and the holder of the last ticket does the tail.  No device-scope fence: a release
// fence writes back EVERYTHING dirty in the XCD's L2 (the previous kernels' outputs -- measured here: +60 us per frame), so the partial rows
// are written and read with agent-scope accesses instead (pfn_st / pfn_ld: they bypass the per-XCD L2), ordered against the ticket by
// waiting for the stores' acknowledgements (workgroup-scope release = s_waitcnt vmcnt(0)).  The ticket word returns to zero for the next
// launch.
template <class T>
__device__ __forceinline__ void pfn_st(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class T>
__device__ __forceinline__ T pfn_ld(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool pfn_last_block(unsigned *ticket)
{
    __shared__ unsigned s_last;
    // every agent-scope store of this thread has been acknowledged before the workgroup takes its ticket.  A workgroup-scope release fence
    // does NOT emit this wait on gfx950 (round-4 advisor: the ISA went store -> s_barrier -> atomic), so it is spelled out; the build's ISA
    // audit (papc_amd/_isa_audit.py::audit_ticket) checks that every ticket atomic of this file sits behind one.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (t == gridDim.x - 1) ? 1u : 0u;
        if (s_last) pfn_st(ticket, 0u);
    }
    __syncthreads();
    return s_last != 0u;
}
template <bool COH>
__device__ void pfn_gram_finalize_body(const double *part, int n_blocks, double M, const float *w, int C, const float *gamma, const float *beta,
                                       float eps, float momentum, float *mean, float *invstd, float *scale, float *shift, float *rmean, float *rvar,
                                       double *gram);

enum { PFN_STATS = 0, PFN_APPLY = 1, PFN_BWD_RED = 2, PFN_BWD_DW = 3, PFN_DECORATE = 4, PFN_GRAM = 5, PFN_BWD_SPARSE = 6 };
typedef double double4_t __attribute__((ext_vector_type(4)));
constexpr int PFN_NACC = 11;   // accumulator slots per channel (BWD_SPARSE: sum p, sum p*xhat, 9 x sum p*x_k)

// Phase A for one pillar: decorate + mask, rows -> LDS (this wave's slab).  Split into the loads (pfn_fetch) and the arithmetic
// (pfn_stage_from) so that a mode whose per-pillar work is short can put the next pillar's loads in flight first.
struct PfnRaw { float4 f[2]; int nv, cx, cy; };

// `nv_known` >= 0: the pillar's point count, fetched an iteration earlier (zero_padded: the row loads are sized by it)
__device__ __forceinline__ PfnRaw pfn_fetch(const PfnArgs &a, int p, int lane, int nv_known = -1)
{
    PfnRaw r;
    r.nv = nv_known >= 0 ? nv_known : a.nvox[p];
    const int lim = a.zero_padded ? (r.nv < a.T ? r.nv : a.T) : a.T;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int t = lane + 64 * h;
        r.f[h] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t < lim) r.f[h] = *reinterpret_cast<const float4 *>(a.feat + ((int64_t)p * a.T + t) * 4);
    }
    r.cx = a.coors[(int64_t)p * 4 + 3];
    r.cy = a.coors[(int64_t)p * 4 + 2];
    return r;
}

template <bool DIST = true>
__device__ __forceinline__ void pfn_stage_from(const PfnArgs &a, const PfnRaw &raw, float *rows, int lane)
{
    const int T = a.T;
    const int nv = raw.nv;
    float sx = 0.f, sy = 0.f, sz = 0.f;
#pragma unroll
    for (int h = 0; h < 2; ++h) { sx += raw.f[h].x; sy += raw.f[h].y; sz += raw.f[h].z; }
    // features[:, :, :3].sum(axis=1) over ALL T rows (zero padding included) / num_voxels   (:82)
    sx = readlane63_f32(wave_sum_f32_to_lane63(sx));
    sy = readlane63_f32(wave_sum_f32_to_lane63(sy));
    sz = readlane63_f32(wave_sum_f32_to_lane63(sz));
    const float fn = (float)nv;
    const float mx = sx / fn, my = sy / fn, mz = sz / fn;
    const float pcx = (float)raw.cx * a.vx + a.xo;  // coors[:,3]*vx + x_offset  (:87)
    const float pcy = (float)raw.cy * a.vy + a.yo;  // coors[:,2]*vy + y_offset  (:88)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int t = lane + 64 * h;
        if (t < T) {
            const float mk = t < nv ? 1.f : 0.f;  // get_paddings_indicator: actual_num > t  (libs/tools/__init__.py:26-35)
            float *r = rows + t * PFN_LD;
            const float4 v = raw.f[h];
            *reinterpret_cast<float4 *>(r) = make_float4(v.x * mk, v.y * mk, v.z * mk, v.w * mk);
            *reinterpret_cast<float4 *>(r + 4) = make_float4((v.x - mx) * mk, (v.y - my) * mk, (v.z - mz) * mk, (v.x - pcx) * mk);
            // (slot 9: points_dist = paddle.norm(features[:, :, :3], 2, 2) of the with_distance variant, :92-94)
            // (DIST = false: the passes of the 9-channel layer never read the slot; a correctly rounded sqrt is ~30 instructions)
            *reinterpret_cast<float4 *>(r + 8) = make_float4((v.y - pcy) * mk, DIST ? sqrtf((v.x * v.x + v.y * v.y) + v.z * v.z) * mk : 0.f, 0.f, 0.f);
        }
    }
}

template <bool DIST = true>
__device__ __forceinline__ void pfn_stage(const PfnArgs &a, int p, float *rows, int lane)
{
    pfn_stage_from<DIST>(a, pfn_fetch(a, p, lane), rows, lane);
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

template <int MODE, int WAVES = PFN_WAVES>
__global__ __launch_bounds__(64 * WAVES) void pfn_kernel(PfnArgs a)
{
    constexpr int PFN_WAVES = WAVES;   // (shadows the default: the Gram pass runs 16-wave workgroups, fewer partial rows to fold)
    __shared__ __attribute__((aligned(16))) float smem[PFN_WAVES * PFN_TMAX * PFN_LD + PFN_WAVES * 64 * PFN_NACC];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *rows = smem + wave * PFN_TMAX * PFN_LD;
    float *xred = smem + PFN_WAVES * PFN_TMAX * PFN_LD;  // [wave][PFN_NACC][64]
    const int c = lane;
    const bool cok = c < a.C;

    float w[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) w[k] = (cok && MODE != PFN_DECORATE && MODE != PFN_GRAM) ? a.w[c * 9 + k] : 0.f;
    float sc = 0.f, sh = 0.f, mu = 0.f, is = 0.f, c1 = 0.f, c2 = 0.f;
    if (MODE != PFN_STATS && MODE != PFN_DECORATE && MODE != PFN_GRAM && cok) { sc = a.scale[c]; sh = a.shift[c]; }
    if ((MODE == PFN_BWD_RED || MODE == PFN_BWD_DW || MODE == PFN_BWD_SPARSE) && cok) { mu = a.mean[c]; is = a.invstd[c]; }
    if (MODE == PFN_BWD_DW && cok) { c1 = a.c1[c]; c2 = a.c2[c]; }

    float acc[PFN_NACC];
#pragma unroll
    for (int i = 0; i < PFN_NACC; ++i) acc[i] = 0.f;
    double4_t gacc[4];                        // PFN_GRAM: this lane's 4 entries of the 16x16 f64 accumulator, x4 independent chains
#pragma unroll
    for (int q = 0; q < 4; ++q) gacc[q] = double4_t{0.0, 0.0, 0.0, 0.0};

    constexpr bool PF = MODE == PFN_GRAM || MODE == PFN_BWD_SPARSE;      // modes whose per-pillar work is short: software-pipelined loads
    PfnRaw nxt = {};
    int nv2 = 0;              // point count of the pillar AFTER the prefetched one (so that a count-sized fetch never waits for its count)
    if (PF && (int)(blockIdx.x * PFN_WAVES + wave) < a.P) {
        const int p0 = blockIdx.x * PFN_WAVES + wave, p1 = p0 + gridDim.x * PFN_WAVES;
        nxt = pfn_fetch(a, p0, lane);
        if (p1 < a.P) nv2 = a.nvox[p1];
    }
    for (int p = blockIdx.x * PFN_WAVES + wave; p < a.P; p += gridDim.x * PFN_WAVES) {
        PfnRaw cur = {};
        if (PF) {   // the next pillar's loads fly under this pillar's work
            cur = nxt;
            const int pn = p + gridDim.x * PFN_WAVES, pnn = pn + gridDim.x * PFN_WAVES;
            if (pn < a.P) nxt = pfn_fetch(a, pn, lane, nv2);
            if (pnn < a.P) nv2 = a.nvox[pnn];
            pfn_stage_from<false>(a, cur, rows, lane);
        } else {
            pfn_stage<MODE == PFN_DECORATE>(a, p, rows, lane);
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (MODE == PFN_DECORATE) {
            const int nc = a.with_dist ? 10 : 9;
            for (int e = lane; e < a.T * nc; e += 64) {
                const int t = e / nc, k = e - t * nc;
                a.out[((int64_t)p * a.T + t) * nc + k] = rows[t * PFN_LD + k];
            }
        } else if (MODE == PFN_GRAM) {
            // G += X^T X over this pillar's T rows, 4 rows per v_mfma_f64_16x16x4_f64.  Lane (ch = lane & 15, kk = lane >> 4) holds
            // X[4m + kk][ch] for BOTH operands (A[i][k] = X[k][i], B[k][j] = X[k][j]).  Channels 0..8 = the decorated row, 9 = the
            // distance slot, 10 = the constant 1 (column sums), 11..15 = 0.
            // Rows t >= num_voxels are all-zero after the mask (:99-102): they add nothing to G except their count to G[10][10], so the
            // MFMAs run over the real rows only (a KITTI pillar holds ~12 of its 100 slots) and the padded rows are counted once.
            const int ch = lane & 15, kk = lane >> 4;
            const int nvc = cur.nv < 0 ? 0 : (cur.nv < a.T ? cur.nv : a.T);
            const int nm = (nvc + 3) >> 2;
            const int chc = ch < PFN_LD ? ch : 0;
            if (lane == 42) gacc[0][2] += (double)(a.T - nvc);     // G[10][10]: accumulator entry i = (lane >> 4) + 4 r, j = lane & 15
            for (int m0 = 0; m0 < nm; m0 += 4) {   // 4 row quads per trip on 4 accumulators (a dependent f64 MFMA chain does not pipeline)
                float x[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int t = 4 * (m0 + q) + kk;
                    x[q] = rows[(t < nvc ? t : 0) * PFN_LD + chc];
                    x[q] = (t < nvc && ch < PFN_LD) ? (ch == 10 ? 1.f : x[q]) : 0.f;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const double xd = (double)x[q];
                    gacc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(xd, xd, gacc[q], 0, 0, 0);
                }
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
        } else if (MODE == PFN_BWD_RED || MODE == PFN_BWD_SPARSE) {
            if (cok) {
                const int am = a.amax[(int64_t)p * a.C + c];
                const float *r = rows + am * PFN_LD;
                const float y = pfn_dot(r, w);
                const float z = fmaf(sc, y, sh);
                const float g = z > 0.f ? a.gout[(int64_t)p * a.C + c] : 0.f;
                acc[0] += g;
                acc[1] = fmaf(g, (y - mu) * is, acc[1]);
                if (MODE == PFN_BWD_SPARSE) {   // T[c][k] += p * x_k of the argmax row
                    const float4 x0 = *reinterpret_cast<const float4 *>(r);
                    const float4 x1 = *reinterpret_cast<const float4 *>(r + 4);
                    acc[2] = fmaf(g, x0.x, acc[2]); acc[3] = fmaf(g, x0.y, acc[3]); acc[4] = fmaf(g, x0.z, acc[4]);
                    acc[5] = fmaf(g, x0.w, acc[5]); acc[6] = fmaf(g, x1.x, acc[6]); acc[7] = fmaf(g, x1.y, acc[7]);
                    acc[8] = fmaf(g, x1.z, acc[8]); acc[9] = fmaf(g, x1.w, acc[9]); acc[10] = fmaf(g, r[8], acc[10]);
                }
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
    if (MODE == PFN_GRAM) {
        // fold the 4 waves (fixed order) and store this workgroup's partial G as [i][j]: j = lane & 15, i = (lane >> 4) + 4 r
        // (the accumulator layout of v_mfma_f64_16x16x4_f64, tools/probe/mfma_f64_layout.hip)
        double *dred = reinterpret_cast<double *>(xred);   // [wave][4][64] doubles (2 KB of each wave's 2.75 KB slab)
#pragma unroll
        for (int r = 0; r < 4; ++r) dred[(wave * 4 + r) * 64 + lane] = (gacc[0][r] + gacc[1][r]) + (gacc[2][r] + gacc[3][r]);
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double sum = 0.0;
#pragma unroll
                for (int g = 0; g < PFN_WAVES; ++g) sum += dred[(g * 4 + r) * 64 + lane];
                double *dst = a.gram + (int64_t)blockIdx.x * 256 + ((lane >> 4) + 4 * r) * 16 + (lane & 15);
                if (a.ticket) pfn_st(dst, sum);
                else *dst = sum;
            }
        }
        if constexpr (WAVES == 16) {      // (the fold wants 1024 threads)
            if (a.ticket && pfn_last_block(a.ticket))
                pfn_gram_finalize_body<true>(a.gram, (int)gridDim.x, a.Mrows, a.w, a.Cfin, a.gamma, a.beta, a.eps, a.momentum, a.o_mean, a.o_invstd, a.o_scale,
                                       a.o_shift, a.rmean, a.rvar, a.gram_out);
        }
        return;
    }
    constexpr int NA = (MODE == PFN_BWD_DW) ? 9 : (MODE == PFN_BWD_SPARSE ? PFN_NACC : 2);
#pragma unroll
    for (int i = 0; i < NA; ++i) xred[(wave * PFN_NACC + i) * 64 + lane] = acc[i];
    __syncthreads();
    if (wave == 0 && cok) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            float s = 0.f;
#pragma unroll
            for (int g = 0; g < PFN_WAVES; ++g) s += xred[(g * PFN_NACC + i) * 64 + lane];
            if (MODE == PFN_BWD_DW) a.dwp[((int64_t)blockIdx.x * a.C + c) * 9 + i] = s;
            else if (MODE == PFN_BWD_SPARSE) a.red[((int64_t)blockIdx.x * PFN_NACC + i) * a.C + c] = s;
            else if (MODE == PFN_STATS) a.stats[((int64_t)blockIdx.x * 2 + i) * a.C + c] = s;
            else a.red[((int64_t)blockIdx.x * 2 + i) * a.C + c] = s;
        }
    }
}

// ---- apply pass on the bf16 matrix pipe -------------------------------------------------------------------------------
// y [T, C] = X [T, 9] W^T for one pillar is 4 row tiles x 2 column tiles of v_mfma_f32_32x32x16_bf16 with the fp32 operands as
// exact 3-way bf16 splits (6 products, fp32 accumulate: mlp_loaders.h).  K = 9 pads to ONE 16-wide k block: lane (row = lane & 31,
// half = lane >> 5) feeds channels 0..7 / 8..15 of its row straight from the decorated LDS slab; W^T lives in registers for the
// whole kernel.  The accumulator layout (lane = output channel, 16 rows per lane) is exactly what BN + ReLU + max over the T rows
// want: z = fma(scale, y, shift) and a running (max, first row) per lane, one cross-half merge per pillar.  The lanes-are-channels
// VALU form of this pass (pfn_kernel<PFN_APPLY>) spends 19 instructions per row and channel-wave; this one ~6.
typedef __bf16 pfn_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 pfn_bf16x2 __attribute__((ext_vector_type(2)));
typedef float pfn_f32x2 __attribute__((ext_vector_type(2)));
typedef float pfn_f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ unsigned pfn_pack2(float a, float b)
{
    return __builtin_bit_cast(unsigned, __builtin_convertvector((pfn_f32x2){a, b}, pfn_bf16x2));
}
// three bf16 planes of 8 consecutive-k values: v = pl[0] + pl[1] + pl[2] exactly
__device__ __forceinline__ void pfn_split8(const float (&vin)[8], pfn_bf16x8 (&pl)[3])
{
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = vin[i];
    unsigned q[3][4];
#pragma unroll
    for (int lvl = 0; lvl < 3; ++lvl) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned u = pfn_pack2(v[2 * j], v[2 * j + 1]);
            q[lvl][j] = u;
            v[2 * j] -= __uint_as_float(u << 16);
            v[2 * j + 1] -= __uint_as_float(u & 0xffff0000u);
        }
    }
#pragma unroll
    for (int lvl = 0; lvl < 3; ++lvl) pl[lvl] = __builtin_bit_cast(pfn_bf16x8, make_uint4(q[lvl][0], q[lvl][1], q[lvl][2], q[lvl][3]));
}

__global__ __launch_bounds__(64 * PFN_WAVES) void pfn_apply_mfma_kernel(PfnArgs a)
{
    __shared__ __attribute__((aligned(16))) float smem[PFN_WAVES * PFN_TMAX * PFN_LD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    float *rows = smem + wave * PFN_TMAX * PFN_LD;

    // B operand: lane (n = l31, half) holds W[32 wn + n][8 half .. 8 half + 7]  (k >= 9 and channels >= C are zero)
    pfn_bf16x8 bq[2][3];
    float sc[2], sh[2];
#pragma unroll
    for (int wn = 0; wn < 2; ++wn) {
        const int c = wn * 32 + l31;
        float wv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 8 * hi + j;
            wv[j] = (c < a.C && k < 9) ? a.w[c * 9 + k] : 0.f;
        }
        pfn_split8(wv, bq[wn]);
        sc[wn] = c < a.C ? a.scale[c] : 0.f;
        sh[wn] = c < a.C ? a.shift[c] : 0.f;
    }
    constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};   // smallest terms first

    PfnRaw nxt = {};
    int nv2 = 0;
    if ((int)(blockIdx.x * PFN_WAVES + wave) < a.P) {
        const int p0 = blockIdx.x * PFN_WAVES + wave, p1 = p0 + gridDim.x * PFN_WAVES;
        nxt = pfn_fetch(a, p0, lane);
        if (p1 < a.P) nv2 = a.nvox[p1];
    }
    for (int p = blockIdx.x * PFN_WAVES + wave; p < a.P; p += gridDim.x * PFN_WAVES) {
        const PfnRaw cur = nxt;
        const int pn = p + gridDim.x * PFN_WAVES, pnn = pn + gridDim.x * PFN_WAVES;
        if (pn < a.P) nxt = pfn_fetch(a, pn, lane, nv2);      // the next pillar's points fly under this pillar's tiles
        if (pnn < a.P) nv2 = a.nvox[pnn];
        pfn_stage_from<false>(a, cur, rows, lane);
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        float best[2] = {0.f, 0.f};   // max_t relu(z_t): starts at relu's floor, row 0 (what a strict `>` scan from -1 over relu(z) ends with)
        int bi[2] = {0, 0};
        // Rows t >= num_voxels are zero after the mask (:99-102) and the Linear has no bias (:23): y = 0, z = shift for every one of
        // them.  Only the tiles that hold real rows go through the matrix pipe; the padded rows enter the max as ONE candidate,
        // z = shift at row num_voxels (the first of them: the first-maximum rule), below.
        const int nvc = cur.nv < 0 ? 0 : (cur.nv < a.T ? cur.nv : a.T);
        const int ntr = (nvc + 31) >> 5;
        for (int t = 0; t < ntr; ++t) {
            // A operand of row 32 t + l31: channels 8 half .. 8 half + 7 (slab row = 12 floats: the upper half's second quad is past the row)
            const float *r = rows + (32 * t + l31) * PFN_LD + 8 * hi;
            const float4 x0 = *reinterpret_cast<const float4 *>(r);
            float4 x1 = *reinterpret_cast<const float4 *>(rows + (32 * t + l31) * PFN_LD + 4);
            if (hi) x1 = make_float4(0.f, 0.f, 0.f, 0.f);
            const float xv[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
            pfn_bf16x8 af[3];
            pfn_split8(xv, af);
            const bool full = 32 * (t + 1) <= a.T;
#pragma unroll
            for (int wn = 0; wn < 2; ++wn) {
                pfn_f32x16 acc;
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
                for (int m = 0; m < 6; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PA[m]], bq[wn][PB[m]], acc, 0, 0, 0);
                // C/D layout: column = l31, row = (q & 3) + 8 (q >> 2) + 4 half -- ascending in q
                if (full) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const int row = 32 * t + (q & 3) + 8 * (q >> 2) + 4 * hi;
                        const float z = fmaf(sc[wn], acc[q], sh[wn]);
                        const bool up = z > best[wn];
                        best[wn] = up ? z : best[wn];
                        bi[wn] = up ? row : bi[wn];
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const int row = 32 * t + (q & 3) + 8 * (q >> 2) + 4 * hi;
                        const float z = fmaf(sc[wn], acc[q], sh[wn]);
                        const bool up = row < a.T && z > best[wn];
                        best[wn] = up ? z : best[wn];
                        bi[wn] = up ? row : bi[wn];
                    }
                }
            }
        }
        // the two halves of a lane pair hold disjoint rows of the same column: larger z wins, the earlier row on a tie
#pragma unroll
        for (int wn = 0; wn < 2; ++wn) {
            const float ob = __shfl_xor(best[wn], 32);
            const int oi = __shfl_xor(bi[wn], 32);
            const bool take = ob > best[wn] || (ob == best[wn] && oi < bi[wn]);
            float fb = take ? ob : best[wn];
            int fi = take ? oi : bi[wn];
            // the padded rows (all after the real ones: only a strictly larger value moves the argmax; the zero rows inside the last
            // processed tile have already offered the same value at the same first row)
            if (nvc < a.T && sh[wn] > fb) { fb = sh[wn]; fi = nvc; }
            const int c = wn * 32 + l31;
            if (hi == 0 && c < a.C) {
                a.out[(int64_t)p * a.C + c] = fb;
                if (a.argmax) a.argmax[(int64_t)p * a.C + c] = fi;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- Gram path, small kernels --------------------------------------------------------------------------------------
constexpr int PFN_GRAM_WAVES = 16;
constexpr int PFN_GRAM_BLOCKS = 256;   // workgroups (16 waves each) of the Gram pass = rows of its partial buffer
constexpr int PFN_GN = 11;             // channels of G that are used: 9 decorated + distance slot + the constant 1

// gram_partial [n_blocks][256] -> G [16][16] (f64, fixed summation tree), then the train-mode BatchNorm constants of the
// 9 -> C linear layer straight from G (see the file header).  One workgroup of 1024 threads: 128 entry lanes x 8 block slices.
template <bool COH>     // COH: the partial rows were written by other workgroups of THIS launch (agent-scope loads)
__device__ void pfn_gram_finalize_body(const double *part, int n_blocks, double M, const float *w, int C, const float *gamma, const float *beta,
                                       float eps, float momentum, float *mean, float *invstd, float *scale, float *shift, float *rmean, float *rvar,
                                       double *gram)
{
    __shared__ double red[8][128];
    __shared__ double G[PFN_GN][PFN_GN];
    const int e = threadIdx.x & 127, sl = threadIdx.x >> 7;
    const int gi = e / PFN_GN, gj = e - gi * PFN_GN;
    double s = 0.0;
    if (e < PFN_GN * PFN_GN) {
        // NF loads in flight per thread: the kernel is a latency chain (the last-arriving workgroup's loads go past its L2: one round of 32;
        // the summation order -- blocks sl, sl + 8, ... -- does not depend on NF)
        constexpr int NF = COH ? 32 : 16;
        for (int b = sl; b < n_blocks; b += 8 * NF) {
            double v[NF];
#pragma unroll
            for (int q = 0; q < NF; ++q) {
                const double *src = part + (int64_t)(b + 8 * q) * 256 + gi * 16 + gj;
                v[q] = (b + 8 * q < n_blocks) ? (COH ? pfn_ld(src) : *src) : 0.0;
            }
#pragma unroll
            for (int q = 0; q < NF; ++q) s += v[q];
        }
    }
    red[sl][e] = s;
    __syncthreads();
    for (int h = 4; h >= 1; h >>= 1) {
        if (sl < h) red[sl][e] += red[sl + h][e];
        __syncthreads();
    }
    if (sl == 0 && e < PFN_GN * PFN_GN) {
        G[gi][gj] = red[0][e];
        gram[gi * 16 + gj] = red[0][e];
    }
    __syncthreads();
    const int c = threadIdx.x;
    if (c < C && mean) {
        double wc[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) wc[k] = (double)w[c * 9 + k];
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            s1 += wc[k] * G[k][10];
            double t = 0.0;
#pragma unroll
            for (int l = 0; l < 9; ++l) t += wc[l] * G[k][l];
            s2 += wc[k] * t;
        }
        const double mu = s1 / M;
        double var = s2 / M - mu * mu;   // biased variance (paddle BatchNorm training), as bn_finalize_kernel
        if (var < 0.0) var = 0.0;
        const double is = 1.0 / sqrt(var + (double)eps);
        const double sc = (gamma ? (double)gamma[c] : 1.0) * is;
        mean[c] = (float)mu;
        invstd[c] = (float)is;
        scale[c] = (float)sc;
        shift[c] = (float)((beta ? (double)beta[c] : 0.0) - mu * sc);
        if (rmean) rmean[c] = momentum * rmean[c] + (1.f - momentum) * (float)mu;
        if (rvar) rvar[c] = momentum * rvar[c] + (1.f - momentum) * (float)var;
    }
}

__global__ __launch_bounds__(1024) void pfn_gram_finalize_kernel(const double *__restrict__ part, int n_blocks, double M, const float *__restrict__ w,
                                                                int C, const float *__restrict__ gamma, const float *__restrict__ beta,
                                                                float eps, float momentum, float *mean, float *invstd, float *scale,
                                                                float *shift, float *rmean, float *rvar, double *gram)
{
    pfn_gram_finalize_body<false>(part, n_blocks, M, w, C, gamma, beta, eps, momentum, mean, invstd, scale, shift, rmean, rvar, gram);
}

// sums [11][C] (sum p, sum p*xhat, T[c][0..8]; reduced over the workgroups by papc_reduce_partials_f32) + G -> dgamma, dbeta, dW [C][9]
template <bool COH>
__device__ __forceinline__ void pfn_bwd_finalize_body(const float *sums, double M, const float *w, int C, const double *gram, const float *mean,
                                                      const float *invstd, const float *scale, float *dgamma, float *dbeta, float *dw, int flags)
{
    const int eval_bn = flags & 1, accumulate = flags & 2;   // bit 1: add into dgamma / dbeta / dw (a parameter's .grad)
    const int c = threadIdx.x;
    if (c >= C) return;
    auto ld = [&](int i) -> float { return COH ? pfn_ld(sums + i) : sums[i]; };
    const double s1 = (double)ld(0 * C + c), s2 = (double)ld(1 * C + c);
    dbeta[c] = accumulate ? dbeta[c] + (float)s1 : (float)s1;
    dgamma[c] = accumulate ? dgamma[c] + (float)s2 : (float)s2;
    const double sc = (double)scale[c];
    // eval-mode BatchNorm (running statistics): dy = sc * p, the batch-mean terms vanish
    const double c1 = eval_bn ? 0.0 : s1 / M, c2 = eval_bn ? 0.0 : s2 / M;
    const double mu = (double)mean[c], is = (double)invstd[c];
    double wc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wc[k] = (double)w[c * 9 + k];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        double wg = 0.0;
#pragma unroll
        for (int l = 0; l < 9; ++l) wg += wc[l] * gram[l * 16 + k];
        const double colsum = gram[k * 16 + 10];
        const double t = (double)ld((2 + k) * C + c);
        const float g = (float)(sc * (t - c1 * colsum - c2 * is * (wg - mu * colsum)));
        dw[c * 9 + k] = accumulate ? dw[c * 9 + k] + g : g;
    }
}

__global__ __launch_bounds__(64) void pfn_bwd_finalize_kernel(const float *__restrict__ sums, double M, const float *__restrict__ w, int C,
                                                             const double *__restrict__ gram, const float *__restrict__ mean,
                                                             const float *__restrict__ invstd, const float *__restrict__ scale,
                                                             float *dgamma, float *dbeta, float *dw, int flags)
{
    pfn_bwd_finalize_body<false>(sums, M, w, C, gram, mean, invstd, scale, dgamma, dbeta, dw, flags);
}

// The fold of the sparse pass's partial rows (the summation order of reduce_partials_kernel, bn_ops.hip: 64 elements x 16 chunk lanes, 8 loads
// in flight) with the finalize above as the last-arriving workgroup's tail: one launch instead of two.
__global__ __launch_bounds__(1024) void pfn_bwd_fold_finalize_kernel(const float *__restrict__ part, int n_chunks, int64_t n, float *sums, unsigned *ticket,
                                                                    double M, const float *w, int C, const double *gram, const float *mean,
                                                                    const float *invstd, const float *scale, float *dgamma, float *dbeta, float *dw, int flags)
{
    __shared__ float red[16][64];
    const int el = threadIdx.x & 63, cl = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + el;
    float s = 0.f;
    if (i < n) {
        for (int t0 = cl; t0 < n_chunks; t0 += 16 * 32) {     // (32 loads in flight; the same summation order as with 8)
            float v[32];
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                const int t = t0 + 16 * j;
                v[j] = part[(int64_t)(t < n_chunks ? t : t0) * n + i];
            }
#pragma unroll
            for (int j = 0; j < 32; ++j) s += (t0 + 16 * j < n_chunks) ? v[j] : 0.f;
        }
    }
    red[cl][el] = s;
    __syncthreads();
    if (cl == 0 && i < n) {
#pragma unroll
        for (int g = 1; g < 16; ++g) s += red[g][el];
        pfn_st(sums + i, s);
    }
    if (pfn_last_block(ticket)) pfn_bwd_finalize_body<true>(sums, M, w, C, gram, mean, invstd, scale, dgamma, dbeta, dw, flags);
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

// The decoration for ANY point width F >= 3 and any T (pillars.py:79-102 is written on features[:, :, :3] and features[:, :, :2]; the
// extra columns ride along): one wave per pillar, lanes = points.  Row = [F raw | xyz - cluster mean | x, y - pillar centre | (norm)].
// The cluster sums add this lane's points in t order, then across the wave -- for T <= 128 the same order as the F = 4 kernels above.
__global__ __launch_bounds__(256) void pfn_decorate_nf_kernel(const float *__restrict__ feat, const int32_t *__restrict__ nvox,
                                                               const int32_t *__restrict__ coors, int P, int T, int F, float vx, float vy,
                                                               float xo, float yo, int with_dist, float *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= P) return;
    const float *f = feat + (int64_t)p * T * F;
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int t = lane; t < T; t += 64) { sx += f[(int64_t)t * F]; sy += f[(int64_t)t * F + 1]; sz += f[(int64_t)t * F + 2]; }
    sx = readlane63_f32(wave_sum_f32_to_lane63(sx));
    sy = readlane63_f32(wave_sum_f32_to_lane63(sy));
    sz = readlane63_f32(wave_sum_f32_to_lane63(sz));
    const int nv = nvox[p];
    const float fn = (float)nv;
    const float mx = sx / fn, my = sy / fn, mz = sz / fn;                                  // (:82)
    const float pcx = (float)coors[(int64_t)p * 4 + 3] * vx + xo;                          // (:87)
    const float pcy = (float)coors[(int64_t)p * 4 + 2] * vy + yo;                          // (:88)
    const int nc = F + 5 + (with_dist ? 1 : 0);
    for (int t = lane; t < T; t += 64) {
        const float mk = t < nv ? 1.f : 0.f;
        const float *r = f + (int64_t)t * F;
        float *o = out + ((int64_t)p * T + t) * nc;
        const float x = r[0], y = r[1], z = r[2];
        for (int c = 0; c < F; ++c) o[c] = r[c] * mk;
        o[F] = (x - mx) * mk; o[F + 1] = (y - my) * mk; o[F + 2] = (z - mz) * mk;
        o[F + 3] = (x - pcx) * mk; o[F + 4] = (y - pcy) * mk;
        if (with_dist) o[F + 5] = sqrtf((x * x + y * y) + z * z) * mk;                      // (:92-94)
    }
}

// ---- the fused launches papc_pfn_fwd / papc_pfn_bwd (sa_mlp.hip) use -------------------------------------------------------------
// papc_pfn_gram_f32 + papc_pfn_gram_finalize_f32 in one launch
int pfn_gram_stats(const float *features, const int32_t *num_voxels, const int32_t *coors, int P, int T, float vx, float vy, float x_offset, float y_offset, int zero_padded,
                   double *gram_partial, const float *w, int C, const float *gamma, const float *beta, float eps, float momentum, float *mean, float *invstd,
                   float *scale, float *shift, float *running_mean, float *running_var, double *gram, unsigned *ticket, hipStream_t st)
{
    const float dummy = 0.f;
    int rc = pfn_check("papc_pfn_fwd (gram)", features, num_voxels, coors, P, T, &dummy, 1);
    if (rc) return rc;
    PAPC_REQUIRE(gram_partial && gram && ticket && w && mean && invstd && scale && shift, PAPC_E_INVALID, "papc_pfn_fwd (gram): null pointer");
    PAPC_REQUIRE(C >= 1 && C <= 64, PAPC_E_UNSUPPORTED, "papc_pfn_fwd (gram): C=%d not in [1,64]", C);
    PfnArgs a;
    memset(&a, 0, sizeof(a));
    a.feat = features; a.nvox = num_voxels; a.coors = coors; a.P = P; a.T = T; a.vx = vx; a.vy = vy; a.xo = x_offset; a.yo = y_offset;
    a.C = 1; a.gram = gram_partial; a.zero_padded = zero_padded;
    a.ticket = ticket; a.w = w; a.Cfin = C; a.gamma = gamma; a.beta = beta; a.eps = eps; a.momentum = momentum; a.Mrows = (double)((int64_t)P * T);
    a.o_mean = mean; a.o_invstd = invstd; a.o_scale = scale; a.o_shift = shift; a.rmean = running_mean; a.rvar = running_var; a.gram_out = gram;
    ProfScope prof(PAPC_K_PFN, st);
    hipLaunchKernelGGL((pfn_kernel<PFN_GRAM, PFN_GRAM_WAVES>), dim3(papc_pfn_gram_blocks(P)), dim3(64 * PFN_GRAM_WAVES), 0, st, a);
    return check_launch("papc_pfn_fwd (gram + statistics)");
}

// papc_reduce_partials_f32 + papc_pfn_bwd_finalize_f32 in one launch
int pfn_bwd_fold_finalize(const float *partial, int n_chunks, float *sums, int64_t M, const float *w, int C, const double *gram, const float *mean,
                          const float *invstd, const float *scale, float *dgamma, float *dbeta, float *dw, int flags, unsigned *ticket, hipStream_t st)
{
    PAPC_REQUIRE(partial && sums && w && gram && mean && invstd && scale && dgamma && dbeta && dw && ticket, PAPC_E_INVALID, "papc_pfn_bwd (fold): null pointer");
    PAPC_REQUIRE(C >= 1 && C <= 64 && M >= 1 && n_chunks >= 1, PAPC_E_UNSUPPORTED, "papc_pfn_bwd (fold): C=%d not in [1,64]", C);
    const int64_t n = (int64_t)PFN_NACC * C;
    ProfScope prof(PAPC_K_PFN, st);
    hipLaunchKernelGGL(pfn_bwd_fold_finalize_kernel, dim3((unsigned)cdiv(n, 64)), dim3(1024), 0, st, partial, n_chunks, n, sums, ticket, (double)M, w, C, gram, mean,
                       invstd, scale, dgamma, dbeta, dw, flags);
    return check_launch("papc_pfn_bwd (fold + finalize)");
}

extern "C" int papc_pfn_gram_blocks(int P);

int pfn_apply_impl(const float *features, const int32_t *num_voxels, const int32_t *coors, int P, int T,
                       float vx, float vy, float x_offset, float y_offset, const float *w, int C,
                       const float *scale, const float *shift, float *out, int32_t *argmax, int zero_padded, papc_stream_t stream)
{
    int rc = pfn_check("papc_pfn_apply_f32", features, num_voxels, coors, P, T, w, C);
    if (rc) return rc;
    PAPC_REQUIRE(scale && shift && out, PAPC_E_INVALID, "papc_pfn_apply_f32: null pointer");
    PfnArgs a;
    memset(&a, 0, sizeof(a));
    a.feat = features; a.nvox = num_voxels; a.coors = coors; a.P = P; a.T = T; a.vx = vx; a.vy = vy; a.xo = x_offset; a.yo = y_offset; a.zero_padded = zero_padded;
    a.w = w; a.C = C; a.scale = scale; a.shift = shift; a.out = out; a.argmax = argmax;
    if (knob(KNOB_PFN_MFMA)) {   // default: the matrix-pipe flavour (PAPC_PFN_MFMA=0: lanes-are-channels VALU flavour)
        hipStream_t st = as_stream(stream);
        ProfScope prof(PAPC_K_PFN, st);
        hipLaunchKernelGGL(pfn_apply_mfma_kernel, dim3(pfn_blocks(P)), dim3(64 * PFN_WAVES), 0, st, a);   // (512 .. 3000 workgroups: same time)
        return check_launch("papc_pfn_apply_f32");
    }
    return launch_pfn<PFN_APPLY>(a, as_stream(stream), "papc_pfn_apply_f32");
}

int pfn_gram_impl(const float *features, const int32_t *num_voxels, const int32_t *coors, int P, int T,
                      float vx, float vy, float x_offset, float y_offset, double *gram_partial, int zero_padded, papc_stream_t stream)
{
    const float dummy = 0.f;
    int rc = pfn_check("papc_pfn_gram_f32", features, num_voxels, coors, P, T, &dummy, 1);
    if (rc) return rc;
    PAPC_REQUIRE(gram_partial, PAPC_E_INVALID, "papc_pfn_gram_f32: null gram_partial");
    PfnArgs a;
    memset(&a, 0, sizeof(a));
    a.feat = features; a.nvox = num_voxels; a.coors = coors; a.P = P; a.T = T; a.vx = vx; a.vy = vy; a.xo = x_offset; a.yo = y_offset; a.zero_padded = zero_padded;
    a.C = 1; a.gram = gram_partial;
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_PFN, st);
    hipLaunchKernelGGL((pfn_kernel<PFN_GRAM, PFN_GRAM_WAVES>), dim3(papc_pfn_gram_blocks(P)), dim3(64 * PFN_GRAM_WAVES), 0, st, a);
    return check_launch("papc_pfn_gram_f32");
}

int pfn_bwd_sparse_impl(const float *features, const int32_t *num_voxels, const int32_t *coors, int P,
                            int T, float vx, float vy, float x_offset, float y_offset, const float *w, int C,
                            const float *gout, const int32_t *argmax, const float *mean, const float *invstd,
                            const float *scale, const float *shift, float *partial, int zero_padded, papc_stream_t stream)
{
    int rc = pfn_check("papc_pfn_bwd_sparse_f32", features, num_voxels, coors, P, T, w, C);
    if (rc) return rc;
    PAPC_REQUIRE(gout && argmax && mean && invstd && scale && shift && partial, PAPC_E_INVALID, "papc_pfn_bwd_sparse_f32: null pointer");
    PfnArgs a;
    memset(&a, 0, sizeof(a));
    a.feat = features; a.nvox = num_voxels; a.coors = coors; a.P = P; a.T = T; a.vx = vx; a.vy = vy; a.xo = x_offset; a.yo = y_offset; a.zero_padded = zero_padded;
    a.w = w; a.C = C; a.gout = gout; a.amax = argmax; a.mean = mean; a.invstd = invstd; a.scale = scale; a.shift = shift; a.red = partial;
    return launch_pfn<PFN_BWD_SPARSE>(a, as_stream(stream), "papc_pfn_bwd_sparse_f32");
}

}  // namespace papc

using namespace papc;

extern "C" {

int papc_pfn_decorate_nf_f32(const float *features, int F, const int32_t *num_voxels, const int32_t *coors, int P, int T,
                             float vx, float vy, float x_offset, float y_offset, int with_distance, float *out, papc_stream_t stream)
{
    PAPC_REQUIRE(features && num_voxels && coors && out, PAPC_E_INVALID, "papc_pfn_decorate_nf_f32: null pointer");
    PAPC_REQUIRE(P >= 1 && T >= 1 && F >= 3, PAPC_E_INVALID, "papc_pfn_decorate_nf_f32: P=%d T=%d F=%d (F >= 3: x, y, z lead every point)", P, T, F);
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_PFN, st);
    hipLaunchKernelGGL(pfn_decorate_nf_kernel, dim3((unsigned)cdiv(P, 4)), dim3(256), 0, st, features, num_voxels, coors, P, T, F, vx, vy, x_offset,
                       y_offset, with_distance ? 1 : 0, out);
    return check_launch("papc_pfn_decorate_nf_f32");
}

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
    return papc::pfn_apply_impl(features, num_voxels, coors, P, T, vx, vy, x_offset, y_offset, w, C, scale, shift, out, argmax, 0, stream);
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

/* ---- Gram path (see the file header) ---- */
int papc_pfn_gram_blocks(int P) { return P >= 1 ? (int)std::min<int64_t>(cdiv(P, PFN_GRAM_WAVES), PFN_GRAM_BLOCKS) : 0; }

int papc_pfn_gram_f32(const float *features, const int32_t *num_voxels, const int32_t *coors, int P, int T,
                      float vx, float vy, float x_offset, float y_offset, double *gram_partial, papc_stream_t stream)
{
    return papc::pfn_gram_impl(features, num_voxels, coors, P, T, vx, vy, x_offset, y_offset, gram_partial, 0, stream);
}

int papc_pfn_gram_finalize_f32(const double *gram_partial, int n_blocks, int64_t M, const float *w, int C, const float *gamma,
                               const float *beta, float eps, float momentum, float *mean, float *invstd, float *scale, float *shift,
                               float *running_mean, float *running_var, double *gram, papc_stream_t stream)
{
    PAPC_REQUIRE(gram_partial && gram, PAPC_E_INVALID, "papc_pfn_gram_finalize_f32: null pointer");
    PAPC_REQUIRE(n_blocks >= 1 && M >= 1, PAPC_E_INVALID, "papc_pfn_gram_finalize_f32: bad sizes");
    if (mean) {
        PAPC_REQUIRE(w && invstd && scale && shift, PAPC_E_INVALID, "papc_pfn_gram_finalize_f32: null pointer");
        PAPC_REQUIRE(C >= 1 && C <= 64, PAPC_E_UNSUPPORTED, "papc_pfn_gram_finalize_f32: C=%d not in [1,64]", C);
    }
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_PFN, st);
    hipLaunchKernelGGL(pfn_gram_finalize_kernel, dim3(1), dim3(1024), 0, st, gram_partial, n_blocks, (double)M, w, C, gamma, beta, eps, momentum,
                       mean, invstd, scale, shift, running_mean, running_var, gram);
    return check_launch("papc_pfn_gram_finalize_f32");
}

int papc_pfn_bwd_sparse_f32(const float *features, const int32_t *num_voxels, const int32_t *coors, int P,
                            int T, float vx, float vy, float x_offset, float y_offset, const float *w, int C,
                            const float *gout, const int32_t *argmax, const float *mean, const float *invstd,
                            const float *scale, const float *shift, float *partial, papc_stream_t stream)
{
    return papc::pfn_bwd_sparse_impl(features, num_voxels, coors, P, T, vx, vy, x_offset, y_offset, w, C, gout, argmax, mean, invstd, scale, shift, partial, 0, stream);
}

int papc_pfn_bwd_finalize_f32(const float *sums, int64_t M, const float *w, int C, const double *gram, const float *mean,
                              const float *invstd, const float *scale, float *dgamma, float *dbeta, float *dw, int eval_bn,
                              papc_stream_t stream)
{
    PAPC_REQUIRE(sums && w && gram && mean && invstd && scale && dgamma && dbeta && dw, PAPC_E_INVALID, "papc_pfn_bwd_finalize_f32: null pointer");
    PAPC_REQUIRE(C >= 1 && C <= 64 && M >= 1, PAPC_E_UNSUPPORTED, "papc_pfn_bwd_finalize_f32: C=%d not in [1,64]", C);
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_PFN, st);
    hipLaunchKernelGGL(pfn_bwd_finalize_kernel, dim3(1), dim3(64), 0, st, sums, (double)M, w, C, gram, mean, invstd, scale, dgamma, dbeta, dw, eval_bn);
    return check_launch("papc_pfn_bwd_finalize_f32");
}

}  // extern "C"
