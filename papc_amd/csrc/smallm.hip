// smallm.hip -- the "planes" path for shared-MLP stacks with FEW rows (gfx950).
//
// The group_all set-abstraction layer (sample_and_group_all + Conv2D/BatchNorm2D/ReLU x3 + max over the points,
// /root/reference/PAPC/models/layers/pointnet2_basic_layers.py:160-176, :215-219; PointNet2_SSG_Clas.sa3,
// /root/reference/PAPC/models/classify/pointnet2/pointnet2.py:16) runs its three layers on M = B*128 = 4096 rows with 259 -> 256 ->
// 512 -> 1024 channels: 17.7 GFLOP forward + backward on 29 MB of activations that never leave L2 / the Infinity Cache.  The row
// GEMM kernels of mlp_gemm.hip are built for M in the hundreds of thousands: on 4096 rows they run one tile per workgroup, one wave
// per SIMD, and every 16-wide k stage pays its whole load -> transform -> split -> LDS -> barrier -> MFMA latency chain
// (~1.8 us, 5x the matrix time).  Here the operand transform is taken OUT of the GEMM:
//
//   * prep kernels apply the elementwise part once per element (concat of sample_and_group_all; BN+ReLU of the previous layer with
//     the batch statistics folded from the producer's per-tile partials in the prologue; the BN/ReLU/max backward dY with its two
//     per-channel constants folded the same way), split the result into the three bf16 planes of the exact fp32 product
//     (mlp_loaders.h: x = p0 + p1 + p2) and store them in MFMA FRAGMENT ORDER: a (32-row block, 16-wide k block, plane) fragment is
//     1 KiB, lane l owning the 16 bytes of row (l & 31), k = 8 (l >> 5) .. + 7 -- exactly what v_mfma_f32_32x32x16_bf16 wants
//     from lane l.  Both orientations are written (rows x channels for the forward / dX contraction over channels, channels x rows
//     for the dW contraction over rows), so every GEMM of the stack is the same "NT" product of two plane sets.
//   * pg_gemm_kernel is then a pure matrix kernel: 1 KiB fragments go global -> LDS by LDS-DMA (global_load_lds_dwordx4, no VGPRs,
//     no VALU), three k32 stages in a ring with a counted s_waitcnt vmcnt so two stages stay in flight across the single barrier
//     of a stage, contiguous conflict-free ds_read_b128 of whole fragments, 48 (or 24) MFMAs per wave and stage.  Epilogues from
//     the accumulator layout: bias + store + BN statistics partials (+ the max / min / first-argmax over a 128-row group: the
//     neighbourhood max of :219 when the group is one row tile), store + BN-backward sums of the layer below (dX), or split-K
//     partial store (dW).
//
// Six MFMA products per fp32 product, fp32 accumulate, smallest terms first: same arithmetic as the row kernels (<= 1e-5 parity).
#include "mlp_loaders.h"

namespace papc {

typedef float floatx16 __attribute__((ext_vector_type(16)));

enum { PG_EPI_STORE = 0, PG_EPI_FWD = 1, PG_EPI_FWD_GMAX = 2, PG_EPI_RED = 3 };
enum { PG_PREP_PLAIN = 0, PG_PREP_CONCAT = 1, PG_PREP_BNRELU = 2, PG_PREP_DY_DENSE = 3, PG_PREP_DY_MAX = 4 };

__host__ __device__ static inline int64_t ru64(int64_t a, int64_t b) { return (a + b - 1) / b * b; }

// ---- fragment-ordered planes -------------------------------------------------------------------------------------------------
// planes of an [R x K] matrix (K = contraction): rows padded to 128, K to 32; fragment (rb, kb, plane) at byte
// ((rb * KB + kb) * 3 + plane) * 1024 with KB = Kpad / 16; inside it lane slot l = (r & 31) + 32 * ((k & 15) >> 3), 16 bytes = 8 bf16
// of consecutive k.
__device__ __forceinline__ void store_planes8(char *frag0, int slot, const float (&v)[8])
{
    uint2 a0, a1, a2, b0, b1, b2;
    split3(make_float4(v[0], v[1], v[2], v[3]), a0, a1, a2);
    split3(make_float4(v[4], v[5], v[6], v[7]), b0, b1, b2);
    *reinterpret_cast<uint4 *>(frag0 + slot * 16) = make_uint4(a0.x, a0.y, b0.x, b0.y);
    *reinterpret_cast<uint4 *>(frag0 + 1024 + slot * 16) = make_uint4(a1.x, a1.y, b1.x, b1.y);
    *reinterpret_cast<uint4 *>(frag0 + 2048 + slot * 16) = make_uint4(a2.x, a2.y, b2.x, b2.y);
}

// ---- weights (any strided fp32 matrix) -> planes -----------------------------------------------------------------------------
struct WJob {
    const float *src; int64_t sr, sc;   // element (r, k) = src[r * sr + k * sc]
    int R, K;                           // valid rows / contraction length (the rest of the padded planes is written as zeros)
    char *dst;
};
struct WJobs {
    WJob j[8];
};

__global__ __launch_bounds__(256) void pg_prep_w_kernel(WJobs jobs)
{
    const WJob jb = jobs.j[blockIdx.y];
    const int KB = (int)(ru64(jb.K, 32) / 16), RB = (int)(ru64(jb.R, 32) / 32);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r_in = lane & 31, h = lane >> 5;
    for (int f = blockIdx.x * 4 + wv; f < RB * KB; f += gridDim.x * 4) {
        const int rb = f / KB, kb = f - rb * KB;
        const int r = rb * 32 + r_in, k0 = kb * 16 + h * 8;
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (r < jb.R && k0 + i < jb.K) ? jb.src[(int64_t)r * jb.sr + (int64_t)(k0 + i) * jb.sc] : 0.f;
        store_planes8(jb.dst + ((int64_t)rb * KB + kb) * 3072, lane, v);
    }
}

// ---- rows -> planes, with the elementwise part of the layer -------------------------------------------------------------------
struct PrepArgs {
    int64_t M; int C;                    // source rows x channels
    const float *x; int64_t ldx;         // PLAIN: input; BNRELU: y of the previous layer; DY_*: y of this layer
    const float *xyz; int64_t sb, sn, sc; const float *feats; int N, D, xyz_first;   // CONCAT: rows [xyz | feats] of sample_and_group_all
    const float *dz;                     // DY_DENSE [M, C]
    const float *gout, *ysel; const int32_t *argmax; int K;   // DY_MAX: [M/K, C] each
    const float *stats; int parts;       // BNRELU: [parts][2][C] sums / sums of squares of x over row ranges
    const float *gamma, *beta; float eps, momentum; float *rmean, *rvar;
    float *mean, *invstd, *scale, *shift;   // BNRELU: written (by the first row block); DY_*: read
    const float *red; int red_parts;     // DY_*: [red_parts][2][C] sums of p and p * xhat (DY_MAX: may be null -> taken from gout / ysel)
    float *dgamma, *dbeta; int accumulate;
    char *P; int KBp;                    // planes [M x C] (contraction over channels), KB = ru(C, 32) / 16; may be null
    char *PT; int KBt;                   // planes [C x M] (contraction over rows),     KB = ru(M, 32) / 16; may be null
};

// one 32-channel block's constants in LDS: [0] scale [1] shift [2] mean [3] invstd [4] c1 [5] c2
template <int MODE>
__device__ __forceinline__ void prep_consts(const PrepArgs &p, int cb, float (*cs)[32], double (*rd)[8][32])
{
    const int cl = threadIdx.x & 31, pl = threadIdx.x >> 5;   // 32 channels x 8 part lanes
    const int c = cb * 32 + cl;
    const bool cok = c < p.C;
    double s1 = 0.0, s2 = 0.0;
    if (MODE == PG_PREP_BNRELU) {
        if (cok) for (int t = pl; t < p.parts; t += 8) { s1 += (double)p.stats[(int64_t)t * 2 * p.C + c]; s2 += (double)p.stats[((int64_t)t * 2 + 1) * p.C + c]; }
    } else if (p.red) {
        if (cok) for (int t = pl; t < p.red_parts; t += 8) { s1 += (double)p.red[(int64_t)t * 2 * p.C + c]; s2 += (double)p.red[((int64_t)t * 2 + 1) * p.C + c]; }
    } else if (MODE == PG_PREP_DY_MAX) {   // p is non-zero at one row per (group, channel): sum it straight from the [G, C] arrays
        if (cok) {
            const float sc = p.scale[c], sh = p.shift[c], mu = p.mean[c], is = p.invstd[c];
            const int64_t G = p.M / p.K;
            for (int64_t g = pl; g < G; g += 8) {
                const float yv = p.ysel[g * p.C + c];
                const float pv = fmaf(sc, yv, sh) > 0.f ? p.gout[g * p.C + c] : 0.f;
                s1 += (double)pv;
                s2 += (double)(pv * ((yv - mu) * is));
            }
        }
    }
    rd[0][pl][cl] = s1; rd[1][pl][cl] = s2;
    __syncthreads();
    if (threadIdx.x < 32) {
        s1 = 0.0; s2 = 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q) { s1 += rd[0][q][cl]; s2 += rd[1][q][cl]; }
        const bool first = blockIdx.y == 0;
        if (MODE == PG_PREP_BNRELU) {
            float sc = 0.f, sh = 0.f;
            if (cok) {
                const double mu = s1 / (double)p.M;
                double var = s2 / (double)p.M - mu * mu;   // biased variance (paddle BatchNorm training), as bn_finalize_kernel
                if (var < 0.0) var = 0.0;
                const double is = 1.0 / sqrt(var + (double)p.eps);
                const double scd = (p.gamma ? (double)p.gamma[c] : 1.0) * is;
                sc = (float)scd;
                sh = (float)((p.beta ? (double)p.beta[c] : 0.0) - mu * scd);
                if (first) {
                    p.mean[c] = (float)mu; p.invstd[c] = (float)is; p.scale[c] = sc; p.shift[c] = sh;
                    if (p.rmean) p.rmean[c] = p.momentum * p.rmean[c] + (1.f - p.momentum) * (float)mu;
                    if (p.rvar) p.rvar[c] = p.momentum * p.rvar[c] + (1.f - p.momentum) * (float)var;
                }
            }
            cs[0][cl] = sc; cs[1][cl] = sh;
        } else {
            float c1 = 0.f, c2 = 0.f;
            if (cok) {
                c1 = (float)(s1 / (double)p.M); c2 = (float)(s2 / (double)p.M);
                if (first) {
                    if (p.dbeta) p.dbeta[c] = p.accumulate ? p.dbeta[c] + (float)s1 : (float)s1;
                    if (p.dgamma) p.dgamma[c] = p.accumulate ? p.dgamma[c] + (float)s2 : (float)s2;
                }
                cs[0][cl] = p.scale[c]; cs[1][cl] = p.shift[c]; cs[2][cl] = p.mean[c]; cs[3][cl] = p.invstd[c];
            } else {
                cs[0][cl] = 0.f; cs[1][cl] = 0.f; cs[2][cl] = 0.f; cs[3][cl] = 0.f;
            }
            cs[4][cl] = c1; cs[5][cl] = c2;
        }
    }
    __syncthreads();
}

// grid (channel blocks of 32, row blocks of 128); a wave owns one 32 x 32 tile
template <int MODE>
__global__ __launch_bounds__(256) void pg_prep_kernel(PrepArgs p)
{
    __shared__ float tile[4][32][33];
    __shared__ float cs[6][32];
    __shared__ double rd[2][8][32];
    const int cb = blockIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r_in = lane & 31, h = lane >> 5;
    const int mb = blockIdx.y * 4 + wv;
    const int64_t m = (int64_t)mb * 32 + r_in;
    const bool mok = m < p.M;
    const int64_t mm = mok ? m : 0;
    // the raw operands are requested BEFORE the constants are folded (prep_consts: <= 64 partial rows in double precision behind two barriers):
    // one exposed round trip per launch instead of two (these launches are 9-16 us of which 4.7 is the launch itself)
    float4 ra[2][2], rdz[2][2];
    int4 rix[2][2];
    int64_t gsel = 0;
    int kinsel = 0;
    if (MODE != PG_PREP_CONCAT) {
        if (MODE == PG_PREP_DY_MAX) { gsel = mm / p.K; kinsel = (int)(mm - gsel * p.K); }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int k0 = cb * 32 + 16 * j + 8 * h;
            const int kk = k0 < p.C ? k0 : 0;
            ra[j][0] = ld4(p.x + mm * p.ldx + kk); ra[j][1] = ld4(p.x + mm * p.ldx + kk + 4);
            if (MODE == PG_PREP_DY_DENSE) { rdz[j][0] = ld4(p.dz + mm * p.C + kk); rdz[j][1] = ld4(p.dz + mm * p.C + kk + 4); }
            if (MODE == PG_PREP_DY_MAX) {
                rdz[j][0] = ld4(p.gout + gsel * p.C + kk); rdz[j][1] = ld4(p.gout + gsel * p.C + kk + 4);
                rix[j][0] = *reinterpret_cast<const int4 *>(p.argmax + gsel * p.C + kk); rix[j][1] = *reinterpret_cast<const int4 *>(p.argmax + gsel * p.C + kk + 4);
            }
        }
    }
    if (MODE == PG_PREP_BNRELU || MODE == PG_PREP_DY_DENSE || MODE == PG_PREP_DY_MAX) prep_consts<MODE>(p, cb, cs, rd);
    float v[2][8];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int kl = 16 * j + 8 * h;            // first of this lane's 8 channels inside the block
        const int k0 = cb * 32 + kl;
        const bool kok = k0 < p.C;                // (C % 8 == 0 outside CONCAT: an octet is all in or all out)
        if (MODE == PG_PREP_CONCAT) {
            const int b = (int)(mm / p.N), n = (int)(mm - (int64_t)b * p.N);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int k = k0 + i;
                float e = 0.f;
                if (mok && k < p.C) {
                    const int kx = p.xyz_first ? k : k - p.D;        // coordinate index when in [0, 3)
                    const int kf = p.xyz_first ? k - 3 : k;          // feature index when in [0, D)
                    if (kx >= 0 && kx < 3) e = p.xyz[(int64_t)b * p.sb + (int64_t)n * p.sn + (int64_t)kx * p.sc];
                    else if (kf >= 0 && kf < p.D) e = p.feats[mm * p.D + kf];
                }
                v[j][i] = e;
            }
        } else {
            const float4 a0 = ra[j][0], a1 = ra[j][1];
            float e[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
            if (MODE == PG_PREP_BNRELU) {
#pragma unroll
                for (int i = 0; i < 8; ++i) e[i] = fmaxf(fmaf(cs[0][kl + i], e[i], cs[1][kl + i]), 0.f);
            } else if (MODE == PG_PREP_DY_DENSE) {
                const float4 d0 = rdz[j][0], d1 = rdz[j][1];
                const float d[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
#pragma unroll
                for (int i = 0; i < 8; ++i) e[i] = dy_elem(d[i], e[i], cs[0][kl + i], cs[1][kl + i], cs[2][kl + i], cs[3][kl + i], cs[4][kl + i], cs[5][kl + i]);
            } else if (MODE == PG_PREP_DY_MAX) {
                const int kin = kinsel;
                const float4 g0 = rdz[j][0], g1 = rdz[j][1];
                const int4 i0 = rix[j][0], i1 = rix[j][1];
                const float d[8] = {i0.x == kin ? g0.x : 0.f, i0.y == kin ? g0.y : 0.f, i0.z == kin ? g0.z : 0.f, i0.w == kin ? g0.w : 0.f,
                                    i1.x == kin ? g1.x : 0.f, i1.y == kin ? g1.y : 0.f, i1.z == kin ? g1.z : 0.f, i1.w == kin ? g1.w : 0.f};
#pragma unroll
                for (int i = 0; i < 8; ++i) e[i] = dy_elem(d[i], e[i], cs[0][kl + i], cs[1][kl + i], cs[2][kl + i], cs[3][kl + i], cs[4][kl + i], cs[5][kl + i]);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) v[j][i] = (mok && kok) ? e[i] : 0.f;
        }
    }
    if (p.P) {
#pragma unroll
        for (int j = 0; j < 2; ++j) store_planes8(p.P + ((int64_t)mb * p.KBp + cb * 2 + j) * 3072, lane, v[j]);
    }
    if (p.PT) {
        // transpose through LDS: PT row = channel cb*32 + (lane & 31), contraction = the tile's 32 rows
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i) tile[wv][r_in][16 * j + 8 * h + i] = v[j][i];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float u[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) u[i] = tile[wv][16 * j + 8 * h + i][r_in];
            store_planes8(p.PT + ((int64_t)cb * p.KBt + mb * 2 + j) * 3072, lane, u);
        }
    }
}

// ---- the matrix kernel: C[i, j] = sum_k A[i, k] B[j, k] on two plane sets -------------------------------------------------------
struct PgArgs {
    const char *a, *b; int KB;        // planes; k16 blocks per 32-row block (the same padded contraction length for both)
    int nst;                          // k32 stages per workgroup; blockIdx.z selects the stage range (split K)
    int R1, R2;                       // valid rows of A / of B = rows / columns of C
    float *c; int64_t ldc, zstride;   // C row-major (+ blockIdx.z * zstride)
    const float *bias;                // FWD
    float *stats;                     // FWD: [gridDim.x][2][R2] column sums / sums of squares; RED: sums of p, p * xhat
    float *gmax, *gmin; int32_t *amax, *amin;   // FWD_GMAX: [gridDim.x][R2]
    const float *y_prev, *mean, *invstd, *scale, *shift;   // RED: the layer below ([R1, R2] and its BN constants)
    int dbg;                          // PAPC_PG_DBG phase-removal bits (timing experiments only)
    int g1, g2;                       // row / column tiles (the grid is 1-D: g1 * g2 * splits workgroups)
};

// one 1 KiB fragment global -> LDS: the LDS address is M0 + lane * 16 (wave-uniform base), the global address per lane
__device__ __forceinline__ void glds16(const char *g, unsigned lds_addr)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(g), "s"(lds_addr)
                 : "memory");
}
template <int N>
__device__ __forceinline__ void pg_wait_barrier()
{
    // this wave's fragments of the stage have landed (only the N loads of the next stage may still be in flight), its LDS reads of
    // the previous stage have returned; then the workgroup barrier publishes both to the other waves
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

// Epilogue from the accumulator layout: lane = column (lane & 31) of a 32-column block, register r = row (r >> 2) * 8 + (lane >> 5) * 4 + (r & 3)
// of a 32-row block.  FULL: the tile lies inside C (no masks).
template <int EPI, int NB, bool FULL>
__device__ __forceinline__ void pg_epilogue(const PgArgs &p, floatx16 (&acc)[2][NB], char *smem, int lane, int wr, int wc, int bi, int bj, int bz)
{
    const int cl = lane & 31, h = lane >> 5;
    constexpr bool FWD = (EPI == PG_EPI_FWD || EPI == PG_EPI_FWD_GMAX);
    constexpr bool SUMS = FWD || EPI == PG_EPI_RED;
    float *rs = reinterpret_cast<float *>(smem);           // cross-wave exchange (the stage ring is free after the barrier below)
    if (SUMS) __syncthreads();
    const int it0 = 2 * wr * 32 + h * 4;                    // first row of this lane inside the tile
    const int ldc = (int)p.ldc;
#pragma unroll
    for (int ib = 0; ib < NB; ++ib) {
        const int jt = (wc * NB + ib) * 32 + cl;            // column inside the tile
        const int j = bj * (NB * 64) + jt;
        const bool jok = FULL || j < p.R2;
        const int jj = jok ? j : 0;
        float *cp = p.c + (int64_t)bz * p.zstride + (int64_t)(bi * 128 + it0) * p.ldc + jj;
        float bias = 0.f, sc = 0.f, sh = 0.f, mu = 0.f, is = 0.f;
        if (FWD && p.bias) bias = p.bias[jj];
        if (EPI == PG_EPI_RED) { sc = p.scale[jj]; sh = p.shift[jj]; mu = p.mean[jj]; is = p.invstd[jj]; }
        float yp[2][16];
        if (EPI == PG_EPI_RED) {                            // the layer below: all 32 loads in flight before the first use
            const float *yq = p.y_prev + (int64_t)(bi * 128 + it0) * p.ldc + jj;
#pragma unroll
            for (int ia = 0; ia < 2; ++ia)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ro = ia * 32 + (r >> 2) * 8 + (r & 3);
                    yp[ia][r] = (FULL || (jok && bi * 128 + it0 + ro < p.R1)) ? yq[ro * ldc] : 0.f;
                }
        }
        if (FWD) {
#pragma unroll
            for (int ia = 0; ia < 2; ++ia)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ia][ib][r] += bias;
        }
#pragma unroll
        for (int ia = 0; ia < 2; ++ia)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ro = ia * 32 + (r >> 2) * 8 + (r & 3);
                if (FULL) cp[ro * ldc] = acc[ia][ib][r];
                else if (jok && bi * 128 + it0 + ro < p.R1) cp[ro * ldc] = acc[ia][ib][r];
            }
        if (!SUMS) continue;
        float s1 = 0.f, s2 = 0.f, mx = -INFINITY, mn = INFINITY;
        int amx = 0, amn = 0;
#pragma unroll
        for (int ia = 0; ia < 2; ++ia)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ro = ia * 32 + (r >> 2) * 8 + (r & 3);
                const bool ok = FULL || (jok && bi * 128 + it0 + ro < p.R1);
                const float v = acc[ia][ib][r];
                if (FWD) {
                    s1 += ok ? v : 0.f;
                    s2 = ok ? fmaf(v, v, s2) : s2;
                    if (EPI == PG_EPI_FWD_GMAX) {
                        if (v > mx) { mx = v; amx = it0 + ro; }
                        if (v < mn) { mn = v; amn = it0 + ro; }
                    }
                } else {
                    const float pv = (ok && fmaf(sc, yp[ia][r], sh) > 0.f) ? v : 0.f;
                    s1 += pv;
                    s2 = fmaf(pv, (yp[ia][r] - mu) * is, s2);
                }
            }
        // the other 32 lanes hold the other half of this wave's 64 rows of the same column
        s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
        if (EPI == PG_EPI_FWD_GMAX) {
            const float omx = __shfl_xor(mx, 32), omn = __shfl_xor(mn, 32);
            const int oamx = __shfl_xor(amx, 32), oamn = __shfl_xor(amn, 32);
            if (omx > mx || (omx == mx && oamx < amx)) { mx = omx; amx = oamx; }
            if (omn < mn || (omn == mn && oamn < amn)) { mn = omn; amn = oamn; }
        }
        if (wr == 1 && h == 0) {
            float *q = rs + jt * 6;
            q[0] = s1; q[1] = s2;
            if (EPI == PG_EPI_FWD_GMAX) { q[2] = mx; q[3] = __int_as_float(amx); q[4] = mn; q[5] = __int_as_float(amn); }
        }
        __syncthreads();
        if (wr == 0 && h == 0 && jok) {
            const float *q = rs + jt * 6;
            p.stats[((int64_t)bi * 2 + 0) * p.R2 + j] = s1 + q[0];
            p.stats[((int64_t)bi * 2 + 1) * p.R2 + j] = s2 + q[1];
            if (EPI == PG_EPI_FWD_GMAX) {     // rows of the wr = 1 waves come after ours: they win only when strictly better
                const float omx = q[2], omn = q[4];
                if (omx > mx) { mx = omx; amx = __float_as_int(q[3]); }
                if (omn < mn) { mn = omn; amn = __float_as_int(q[5]); }
                p.gmax[(int64_t)bi * p.R2 + j] = mx; p.amax[(int64_t)bi * p.R2 + j] = amx;
                p.gmin[(int64_t)bi * p.R2 + j] = mn; p.amin[(int64_t)bi * p.R2 + j] = amn;
            }
        }
    }
}

// NS = stages in the LDS ring (NS - 1 of them in flight): 3 stages x 48 KiB is one workgroup per CU; (NB = 1, NS = 2) is 72 KiB, two
// workgroups per CU whose waves fill each other's LDS-read and barrier gaps.
template <int EPI, int NB, int NS>
__global__ __launch_bounds__(256, 1) void pg_gemm_kernel(PgArgs p)
{
    constexpr int NRB = 4 + 2 * NB;              // 32-row blocks per stage: 4 of A, then 2 NB of B
    constexpr int STAGE = NRB * 6 * 1024;        // a (row block, k32) piece is 6 KiB: [2 k16][3 planes][1 KiB]
    constexpr int NLW = NRB * 6 / 4;             // fragment loads per wave and stage
    __shared__ __attribute__((aligned(1024))) char smem[NS * STAGE];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wr = w >> 1, wc = w & 1;
    // workgroup -> (split z, row tile bi, column tile bj).  Consecutive workgroup ids go round-robin to the 8 XCDs (one L2 each): give
    // every XCD a CONTIGUOUS range of the (z, bi, bj) order, so that it streams one k range of few row tiles against all column tiles
    // (forward / dX) or all tiles of one k range (dW) through its own L2 instead of every XCD pulling every operand from the fabric.
    int t = blockIdx.x;
    const int nwg = gridDim.x;
    if ((nwg & 7) == 0) t = (t & 7) * (nwg >> 3) + (t >> 3);
    const int tiles = p.g1 * p.g2;
    const int bz = t / tiles;
    t -= bz * tiles;
    const int bi = t / p.g2, bj = t - bi * p.g2;
    const int nst = p.nst;
    const int64_t st0 = (int64_t)bz * nst;
    const unsigned lds0 = (unsigned)(uintptr_t)smem;

    // this wave's share of a stage: A row block w (6 fragments) and, of B, row block w (NB = 2) or half of row block w >> 1 (NB = 1)
    const char *ga = p.a + (((int64_t)bi * 4 + w) * p.KB + st0 * 2) * 3072 + lane * 16;
    const char *gb;
    unsigned sa = (unsigned)(w * 6) * 1024, sbq;
    if (NB == 2) {
        gb = p.b + (((int64_t)bj * 4 + w) * p.KB + st0 * 2) * 3072 + lane * 16;
        sbq = (unsigned)((4 + w) * 6) * 1024;
    } else {
        gb = p.b + (((int64_t)bj * 2 + (w >> 1)) * p.KB + st0 * 2) * 3072 + (w & 1) * 3072 + lane * 16;
        sbq = (unsigned)((4 + (w >> 1)) * 6 + (w & 1) * 3) * 1024;
    }
    auto issue = [&](int s, int buf) {
        const unsigned base = lds0 + (unsigned)buf * STAGE;
        const char *a = ga + (int64_t)s * 6144, *b = gb + (int64_t)s * 6144;
#pragma unroll
        for (int q = 0; q < 6; ++q) glds16(a + q * 1024, __builtin_amdgcn_readfirstlane(base + sa + q * 1024));
#pragma unroll
        for (int q = 0; q < (NB == 2 ? 6 : 3); ++q) glds16(b + q * 1024, __builtin_amdgcn_readfirstlane(base + sbq + q * 1024));
    };

    floatx16 acc[2][NB];
#pragma unroll
    for (int ia = 0; ia < 2; ++ia)
#pragma unroll
        for (int ib = 0; ib < NB; ++ib)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ia][ib][r] = 0.f;

    if (!(p.dbg & 2)) {
        issue(0, 0);
        if (NS == 3 && nst > 1) issue(1, 1);
    }
    int buf = 0;
    for (int s = 0; s < nst; ++s) {
        if (NS == 3 && s + 1 < nst) pg_wait_barrier<NLW>();
        else pg_wait_barrier<0>();
        // refill the buffer every wave finished reading before this barrier
        if (s + NS - 1 < nst && !(p.dbg & 2)) issue(s + NS - 1, buf == 0 ? NS - 1 : buf - 1);
        const char *sb = smem + buf * STAGE + lane * 16;
#pragma unroll
        for (int kbl = 0; kbl < 2; ++kbl) {
            if (p.dbg & 4) continue;
            bf16x8 af[2][3], bq[NB][3];
#pragma unroll
            for (int ia = 0; ia < 2; ++ia)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) af[ia][pl] = *reinterpret_cast<const bf16x8 *>(sb + ((2 * wr + ia) * 6 + kbl * 3 + pl) * 1024);
#pragma unroll
            for (int ib = 0; ib < NB; ++ib)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) bq[ib][pl] = *reinterpret_cast<const bf16x8 *>(sb + ((4 + wc * NB + ib) * 6 + kbl * 3 + pl) * 1024);
            // a*b = a0b0 + (a0b1 + a1b0) + (a0b2 + a1b1 + a2b0), smallest terms first
            constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
            if (p.dbg & 1) {   // keep the LDS reads alive without the matrix work
#pragma unroll
                for (int ia = 0; ia < 2; ++ia)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) asm volatile("" ::"v"(af[ia][pl]));
#pragma unroll
                for (int ib = 0; ib < NB; ++ib)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) asm volatile("" ::"v"(bq[ib][pl]));
                continue;
            }
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int ia = 0; ia < 2; ++ia)
#pragma unroll
                    for (int ib = 0; ib < NB; ++ib)
                        acc[ia][ib] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ia][PA[t]], bq[ib][PB[t]], acc[ia][ib], 0, 0, 0);
        }
        buf = buf == NS - 1 ? 0 : buf + 1;
    }

    // ---- epilogue (full tiles take the straight-line flavour: a per-element predicate puts every store in its own basic block
    // behind an s_waitcnt vmcnt(0))
    if (p.dbg & 8) return;
    const bool full = bi * 128 + 128 <= p.R1 && (bj + 1) * (NB * 64) <= p.R2;
    if (full) pg_epilogue<EPI, NB, true>(p, acc, smem, lane, wr, wc, bi, bj, bz);
    else pg_epilogue<EPI, NB, false>(p, acc, smem, lane, wr, wc, bi, bj, bz);
}

// ---- last forward layer: batch statistics -> BN constants, then BN + ReLU on the selected extreme of every (group, channel) ------
__global__ __launch_bounds__(256) void pg_final_kernel(const float *__restrict__ stats, int parts, int64_t M, int C, const float *__restrict__ gamma,
                                                       const float *__restrict__ beta, float eps, float momentum, float *mean, float *invstd,
                                                       float *scale, float *shift, float *rmean, float *rvar, float *__restrict__ gmax,
                                                       const float *__restrict__ gmin, const int32_t *__restrict__ amax,
                                                       const int32_t *__restrict__ amin, int64_t G, float *__restrict__ out, int32_t *__restrict__ argmax,
                                                       int tpg, float *__restrict__ ysel)
{
    __shared__ double rd[2][4][64];
    __shared__ float cs[2][64];
    const int cl = threadIdx.x & 63, gl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const bool cok = c < C;
    double s1 = 0.0, s2 = 0.0;
    if (cok) for (int t = gl; t < parts; t += 4) { s1 += (double)stats[(int64_t)t * 2 * C + c]; s2 += (double)stats[((int64_t)t * 2 + 1) * C + c]; }
    rd[0][gl][cl] = s1; rd[1][gl][cl] = s2;
    __syncthreads();
    if (gl == 0) {
        s1 = (rd[0][0][cl] + rd[0][1][cl]) + (rd[0][2][cl] + rd[0][3][cl]);
        s2 = (rd[1][0][cl] + rd[1][1][cl]) + (rd[1][2][cl] + rd[1][3][cl]);
        float sc = 0.f, sh = 0.f;
        if (cok) {
            const double mu = s1 / (double)M;
            double var = s2 / (double)M - mu * mu;
            if (var < 0.0) var = 0.0;
            const double is = 1.0 / sqrt(var + (double)eps);
            const double scd = (gamma ? (double)gamma[c] : 1.0) * is;
            sc = (float)scd;
            sh = (float)((beta ? (double)beta[c] : 0.0) - mu * scd);
            mean[c] = (float)mu; invstd[c] = (float)is; scale[c] = sc; shift[c] = sh;
            if (rmean) rmean[c] = momentum * rmean[c] + (1.f - momentum) * (float)mu;
            if (rvar) rvar[c] = momentum * rvar[c] + (1.f - momentum) * (float)var;
        }
        cs[0][cl] = sc; cs[1][cl] = sh;
    }
    __syncthreads();
    if (!cok) return;
    const float sc = cs[0][cl], sh = cs[1][cl];
    const bool up = sc >= 0.f;      // relu(sc*y+sh) is non-decreasing in y for sc >= 0 (bn_select_max_kernel)
    if (tpg > 1) {
        // a group spans tpg consecutive 128-row tiles (nsample = 128 tpg: PointNet-Basic's max over the N points of a cloud, pointnet_base.py:44):
        // the extreme of the tiles' extrema, the first tile winning ties (= the first row of the group); ysel goes to its own [G, C] array
        const float *gsel = up ? gmax : gmin;             // (the extremum the ReLU'd BatchNorm can select: one array, one index array)
        const int32_t *asel = up ? amax : amin;
        for (int64_t g = gl; g < G; g += 4) {
            float best = up ? -INFINITY : INFINITY;
            int arg = 0;
            for (int t0 = 0; t0 < tpg; t0 += 8) {          // eight tiles' values and offsets in flight (a dependent chain of tpg round trips otherwise)
                float v[8];
                int a[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int t = t0 + j < tpg ? t0 + j : t0;
                    const int64_t e = (g * tpg + t) * C + c;
                    v[j] = gsel[e]; a[j] = asel[e];
                }
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (t0 + j < tpg && (up ? v[j] > best : v[j] < best)) { best = v[j]; arg = (t0 + j) * 128 + a[j]; }
            }
            out[g * C + c] = relu_np(fmaf(sc, best, sh));
            ysel[g * C + c] = best;
            argmax[g * C + c] = arg;
        }
        return;
    }
    for (int64_t g = gl; g < G; g += 4) {
        const int64_t e = g * C + c;
        const float sel = up ? gmax[e] : gmin[e];
        out[e] = relu_np(fmaf(sc, sel, sh));
        gmax[e] = sel;               // ysel: the raw value behind out[e]
        argmax[e] = up ? amax[e] : amin[e];
    }
}

// ---- split-K partials -> gradient (fixed order) --------------------------------------------------------------------------------
struct FoldJob {
    const float *part; int nsplit; int64_t stride, n; float *out; int accumulate;
};
struct FoldJobs {
    FoldJob j[8];
};
__global__ __launch_bounds__(256) void pg_fold_kernel(FoldJobs jobs)
{
    const FoldJob jb = jobs.j[blockIdx.y];
    const int64_t n4 = jb.n >> 2;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n4; e += (int64_t)gridDim.x * 256) {
        float4 s = ld4(jb.part + e * 4);
        for (int t = 1; t < jb.nsplit; ++t) {
            const float4 v = ld4(jb.part + t * jb.stride + e * 4);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        float4 *o = reinterpret_cast<float4 *>(jb.out + e * 4);
        if (jb.accumulate) { const float4 a = *o; s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w; }
        *o = s;
    }
    if (blockIdx.x == 0) {           // ragged tail (n % 4 elements)
        for (int64_t e = (n4 << 2) + threadIdx.x; e < jb.n; e += 256) {
            float s = jb.part[e];
            for (int t = 1; t < jb.nsplit; ++t) s += jb.part[t * jb.stride + e];
            jb.out[e] = jb.accumulate ? jb.out[e] + s : s;
        }
    }
}

}  // namespace papc

using namespace papc;

extern "C" {

size_t papc_pg_planes_bytes(int64_t R, int64_t K)
{
    if (R < 1 || K < 1) return 0;
    return (size_t)(ru64(R, 128) * ru64(K, 32) * 6);
}

int papc_pg_prep_weights_f32(const papc_pg_wjob *jobs, int count, papc_stream_t stream)
{
    PAPC_REQUIRE(jobs && count >= 1 && count <= 8, PAPC_E_INVALID, "papc_pg_prep_weights_f32: 1..8 jobs");
    WJobs wj;
    memset(&wj, 0, sizeof(wj));
    int64_t most = 0;
    for (int i = 0; i < count; ++i) {
        PAPC_REQUIRE(jobs[i].src && jobs[i].planes && jobs[i].R >= 1 && jobs[i].K >= 1, PAPC_E_INVALID, "papc_pg_prep_weights_f32: bad job %d", i);
        PAPC_REQUIRE(aligned16(jobs[i].planes), PAPC_E_INVALID, "papc_pg_prep_weights_f32: planes must be 16-byte aligned");
        wj.j[i].src = jobs[i].src; wj.j[i].sr = jobs[i].row_stride; wj.j[i].sc = jobs[i].col_stride; wj.j[i].R = jobs[i].R; wj.j[i].K = jobs[i].K;
        wj.j[i].dst = reinterpret_cast<char *>(jobs[i].planes);
        most = std::max<int64_t>(most, (ru64(jobs[i].R, 32) / 32) * (ru64(jobs[i].K, 32) / 16));
    }
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_MISC, st);
    hipLaunchKernelGGL(pg_prep_w_kernel, dim3((unsigned)std::min<int64_t>(cdiv(most, 4), 512), (unsigned)count), dim3(256), 0, st, wj);
    return check_launch("papc_pg_prep_weights_f32");
}

int papc_pg_prep_rows_f32(const papc_pg_prep *a, papc_stream_t stream)
{
    PAPC_REQUIRE(a, PAPC_E_INVALID, "papc_pg_prep_rows_f32: null args");
    PAPC_REQUIRE(a->M >= 1 && a->M < (1ll << 31) && a->C >= 1, PAPC_E_INVALID, "papc_pg_prep_rows_f32: bad sizes");
    PAPC_REQUIRE(a->planes || a->planes_t, PAPC_E_INVALID, "papc_pg_prep_rows_f32: no output");
    PAPC_REQUIRE((!a->planes || aligned16(a->planes)) && (!a->planes_t || aligned16(a->planes_t)), PAPC_E_INVALID, "papc_pg_prep_rows_f32: planes must be 16-byte aligned");
    PAPC_REQUIRE(!a->planes_t || a->M % 32 == 0, PAPC_E_UNSUPPORTED, "papc_pg_prep_rows_f32: transposed planes need M %% 32 == 0");
    PrepArgs p;
    memset(&p, 0, sizeof(p));
    p.M = a->M; p.C = a->C;
    p.P = reinterpret_cast<char *>(a->planes); p.KBp = (int)(ru64(a->C, 32) / 16);
    p.PT = reinterpret_cast<char *>(a->planes_t); p.KBt = (int)(ru64(a->M, 32) / 16);
    const int mode = a->mode;
    if (mode == PG_PREP_CONCAT) {
        PAPC_REQUIRE(a->xyz && a->N >= 1 && a->M % a->N == 0 && a->D >= 0 && a->C == a->D + 3 && (a->D == 0 || a->feats), PAPC_E_INVALID,
                     "papc_pg_prep_rows_f32: CONCAT needs xyz, feats [M, D], C == D + 3, N | M");
        p.xyz = a->xyz; p.sb = a->sb; p.sn = a->sn; p.sc = a->sc; p.feats = a->feats; p.N = a->N; p.D = a->D; p.xyz_first = a->xyz_first;
    } else {
        PAPC_REQUIRE(mode == PG_PREP_PLAIN || mode == PG_PREP_BNRELU || mode == PG_PREP_DY_DENSE || mode == PG_PREP_DY_MAX, PAPC_E_INVALID,
                     "papc_pg_prep_rows_f32: bad mode %d", mode);
        PAPC_REQUIRE(a->x && a->ldx >= a->C, PAPC_E_INVALID, "papc_pg_prep_rows_f32: x null or ldx < C");
        PAPC_REQUIRE(a->C % 8 == 0 && a->ldx % 4 == 0 && aligned16(a->x), PAPC_E_UNSUPPORTED, "papc_pg_prep_rows_f32: needs C %% 8 == 0 and 16-byte aligned rows");
        p.x = a->x; p.ldx = a->ldx;
    }
    if (mode == PG_PREP_BNRELU) {
        PAPC_REQUIRE(a->stats && a->parts >= 1 && a->mean && a->invstd && a->scale && a->shift, PAPC_E_INVALID, "papc_pg_prep_rows_f32: BNRELU needs stats and the four constant vectors");
        p.stats = a->stats; p.parts = a->parts; p.gamma = a->gamma; p.beta = a->beta; p.eps = a->eps; p.momentum = a->momentum;
        p.rmean = a->running_mean; p.rvar = a->running_var;
    }
    if (mode == PG_PREP_DY_DENSE || mode == PG_PREP_DY_MAX) {
        PAPC_REQUIRE(a->mean && a->invstd && a->scale && a->shift, PAPC_E_INVALID, "papc_pg_prep_rows_f32: DY needs the layer's BN constants");
        PAPC_REQUIRE(a->ldx == a->C, PAPC_E_INVALID, "papc_pg_prep_rows_f32: DY needs dense y (ldx == C)");
        if (mode == PG_PREP_DY_DENSE) {
            PAPC_REQUIRE(a->dz && aligned16(a->dz) && a->red && a->red_parts >= 1, PAPC_E_INVALID, "papc_pg_prep_rows_f32: DY_DENSE needs dz and the red partials");
        } else {
            PAPC_REQUIRE(a->gout && a->argmax && a->K >= 1 && a->M % a->K == 0 && aligned16(a->gout) && aligned16(a->argmax), PAPC_E_INVALID,
                         "papc_pg_prep_rows_f32: DY_MAX needs gout / argmax [M/K, C] and K | M");
            PAPC_REQUIRE(a->red || a->ysel, PAPC_E_INVALID, "papc_pg_prep_rows_f32: DY_MAX needs red partials or ysel");
        }
        p.dz = a->dz; p.gout = a->gout; p.ysel = a->ysel; p.argmax = a->argmax; p.K = a->K > 0 ? a->K : 1;
        p.red = a->red; p.red_parts = a->red_parts; p.dgamma = a->dgamma; p.dbeta = a->dbeta; p.accumulate = a->accumulate;
    }
    p.mean = a->mean; p.invstd = a->invstd; p.scale = a->scale; p.shift = a->shift;
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_MISC, st);
    const dim3 grid((unsigned)(ru64(a->C, 32) / 32), (unsigned)(ru64(a->M, 128) / 128));
    switch (mode) {
    case PG_PREP_PLAIN: hipLaunchKernelGGL(pg_prep_kernel<PG_PREP_PLAIN>, grid, dim3(256), 0, st, p); break;
    case PG_PREP_CONCAT: hipLaunchKernelGGL(pg_prep_kernel<PG_PREP_CONCAT>, grid, dim3(256), 0, st, p); break;
    case PG_PREP_BNRELU: hipLaunchKernelGGL(pg_prep_kernel<PG_PREP_BNRELU>, grid, dim3(256), 0, st, p); break;
    case PG_PREP_DY_DENSE: hipLaunchKernelGGL(pg_prep_kernel<PG_PREP_DY_DENSE>, grid, dim3(256), 0, st, p); break;
    default: hipLaunchKernelGGL(pg_prep_kernel<PG_PREP_DY_MAX>, grid, dim3(256), 0, st, p); break;
    }
    return check_launch("papc_pg_prep_rows_f32");
}

int papc_pg_gemm_f32(const papc_pg_gemm *g, papc_stream_t stream)
{
    PAPC_REQUIRE(g && g->a && g->b && g->c, PAPC_E_INVALID, "papc_pg_gemm_f32: null pointer");
    PAPC_REQUIRE(g->R1 >= 1 && g->R2 >= 1 && g->K >= 1 && g->ldc >= g->R2, PAPC_E_INVALID, "papc_pg_gemm_f32: bad sizes");
    PAPC_REQUIRE(aligned16(g->a) && aligned16(g->b), PAPC_E_INVALID, "papc_pg_gemm_f32: planes must be 16-byte aligned");
    const int nst_all = (int)(ru64(g->K, 32) / 32);
    const int split = g->split >= 1 ? g->split : 1;
    PAPC_REQUIRE(nst_all % split == 0, PAPC_E_INVALID, "papc_pg_gemm_f32: split %d does not divide the %d k32 stages", split, nst_all);
    PAPC_REQUIRE(split == 1 || g->epi == PG_EPI_STORE, PAPC_E_INVALID, "papc_pg_gemm_f32: split K only with the plain store epilogue");
    PgArgs p;
    memset(&p, 0, sizeof(p));
    p.a = reinterpret_cast<const char *>(g->a); p.b = reinterpret_cast<const char *>(g->b); p.KB = nst_all * 2; p.nst = nst_all / split;
    p.R1 = g->R1; p.R2 = g->R2; p.c = g->c; p.ldc = g->ldc; p.zstride = g->split_stride;
    p.bias = g->bias; p.stats = g->stats; p.dbg = knob(KNOB_PG_DBG);
    const int epi = g->epi;
    if (epi == PG_EPI_FWD || epi == PG_EPI_FWD_GMAX || epi == PG_EPI_RED) PAPC_REQUIRE(g->stats, PAPC_E_INVALID, "papc_pg_gemm_f32: this epilogue needs stats");
    if (epi == PG_EPI_FWD_GMAX) {
        PAPC_REQUIRE(g->gmax && g->gmin && g->amax && g->amin && g->R1 % 128 == 0, PAPC_E_INVALID, "papc_pg_gemm_f32: GMAX needs its four arrays and whole 128-row groups");
        p.gmax = g->gmax; p.gmin = g->gmin; p.amax = g->amax; p.amin = g->amin;
    }
    if (epi == PG_EPI_RED) {
        PAPC_REQUIRE(g->y_prev && g->mean && g->invstd && g->scale && g->shift && g->ldc == g->R2, PAPC_E_INVALID, "papc_pg_gemm_f32: RED needs the layer below (dense)");
        p.y_prev = g->y_prev; p.mean = g->mean; p.invstd = g->invstd; p.scale = g->scale; p.shift = g->shift;
    }
    hipStream_t st = as_stream(stream);
    ProfScope prof(g->family >= 0 && g->family < PAPC_K_COUNT ? g->family : PAPC_K_MISC, st);
    // Tile flavours.  (NB = 2, NS = 3): 128 x 128 tiles, 144 KiB of LDS, one workgroup (one wave per SIMD) per CU -- the least operand
    // traffic, but every LDS-read wait of its lone wave is exposed.  (NB = 1, NS = 2): 128 x 64 tiles, 72 KiB, two workgroups per CU.
    // PAPC_PG_NB / PAPC_PG_NS force one (experiments).
    const int64_t t1 = cdiv(g->R1, 128);
    int nb = knob(KNOB_PG_NB), ns = knob(KNOB_PG_NS);
    if (nb == 0) nb = (g->R2 > 64 && t1 * cdiv(g->R2, 128) * split >= 200) ? 2 : 1;
    if (g->R2 <= 64) nb = 1;
    if (ns == 0) ns = 2;   // (96 / 72 KiB: the workgroup also fits beside a 49 KiB farthest-point-sampling workgroup of the sampling branch)
    p.g1 = (int)t1; p.g2 = (int)cdiv(g->R2, nb == 2 ? 128 : 64);
    const dim3 grid((unsigned)(p.g1 * p.g2 * split));
#define PG_GO(E)                                                                                      \
    do {                                                                                              \
        if (nb == 2 && ns == 3) hipLaunchKernelGGL((pg_gemm_kernel<E, 2, 3>), grid, dim3(256), 0, st, p);      \
        else if (nb == 2) hipLaunchKernelGGL((pg_gemm_kernel<E, 2, 2>), grid, dim3(256), 0, st, p);            \
        else if (ns == 3) hipLaunchKernelGGL((pg_gemm_kernel<E, 1, 3>), grid, dim3(256), 0, st, p);            \
        else hipLaunchKernelGGL((pg_gemm_kernel<E, 1, 2>), grid, dim3(256), 0, st, p);                         \
    } while (0)
    switch (epi) {
    case PG_EPI_STORE: PG_GO(PG_EPI_STORE); break;
    case PG_EPI_FWD: PG_GO(PG_EPI_FWD); break;
    case PG_EPI_FWD_GMAX: PG_GO(PG_EPI_FWD_GMAX); break;
    case PG_EPI_RED: PG_GO(PG_EPI_RED); break;
    default: set_error("papc_pg_gemm_f32: bad epilogue %d", epi); return PAPC_E_INVALID;
    }
#undef PG_GO
    return check_launch("papc_pg_gemm_f32");
}

int papc_pg_final_f32(const float *stats, int parts, int64_t M, int C, const float *gamma, const float *beta, float eps, float momentum,
                      float *mean, float *invstd, float *scale, float *shift, float *running_mean, float *running_var, float *gmax,
                      const float *gmin, const int32_t *amax, const int32_t *amin, int64_t G, float *out, int32_t *argmax, papc_stream_t stream)
{
    PAPC_REQUIRE(stats && mean && invstd && scale && shift && gmax && gmin && amax && amin && out && argmax, PAPC_E_INVALID, "papc_pg_final_f32: null pointer");
    PAPC_REQUIRE(parts >= 1 && M >= 1 && C >= 1 && G >= 1, PAPC_E_INVALID, "papc_pg_final_f32: bad sizes");
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_BN_RELU_MAX, st);
    hipLaunchKernelGGL(pg_final_kernel, dim3((unsigned)cdiv(C, 64)), dim3(256), 0, st, stats, parts, M, C, gamma, beta, eps, momentum, mean, invstd,
                       scale, shift, running_mean, running_var, gmax, gmin, amax, amin, G, out, argmax, 1, (float *)nullptr);
    return check_launch("papc_pg_final_f32");
}

int papc_pg_final_groups_f32(const float *stats, int parts, int64_t M, int C, const float *gamma, const float *beta, float eps, float momentum,
                             float *mean, float *invstd, float *scale, float *shift, float *running_mean, float *running_var, const float *gmax,
                             const float *gmin, const int32_t *amax, const int32_t *amin, int64_t G, int tiles_per_group, float *out, int32_t *argmax,
                             float *ysel, papc_stream_t stream)
{
    PAPC_REQUIRE(stats && mean && invstd && scale && shift && gmax && gmin && amax && amin && out && argmax && ysel, PAPC_E_INVALID, "papc_pg_final_groups_f32: null pointer");
    PAPC_REQUIRE(parts >= 1 && M >= 1 && C >= 1 && G >= 1 && tiles_per_group >= 2, PAPC_E_INVALID, "papc_pg_final_groups_f32: bad sizes (tiles_per_group >= 2; 1 is papc_pg_final_f32)");
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_BN_RELU_MAX, st);
    hipLaunchKernelGGL(pg_final_kernel, dim3((unsigned)cdiv(C, 64)), dim3(256), 0, st, stats, parts, M, C, gamma, beta, eps, momentum, mean, invstd,
                       scale, shift, running_mean, running_var, const_cast<float *>(gmax), gmin, amax, amin, G, out, argmax, tiles_per_group, ysel);
    return check_launch("papc_pg_final_groups_f32");
}

int papc_pg_fold_f32(const papc_pg_fold_job *jobs, int count, papc_stream_t stream)
{
    PAPC_REQUIRE(jobs && count >= 1 && count <= 8, PAPC_E_INVALID, "papc_pg_fold_f32: 1..8 jobs");
    FoldJobs fj;
    memset(&fj, 0, sizeof(fj));
    int64_t most = 0;
    for (int i = 0; i < count; ++i) {
        PAPC_REQUIRE(jobs[i].partial && jobs[i].out && jobs[i].nsplit >= 1 && jobs[i].n >= 1, PAPC_E_INVALID, "papc_pg_fold_f32: bad job %d", i);
        PAPC_REQUIRE(aligned16(jobs[i].partial) && aligned16(jobs[i].out) && jobs[i].stride % 4 == 0, PAPC_E_INVALID, "papc_pg_fold_f32: 16-byte alignment");
        fj.j[i].part = jobs[i].partial; fj.j[i].nsplit = jobs[i].nsplit; fj.j[i].stride = jobs[i].stride; fj.j[i].n = jobs[i].n;
        fj.j[i].out = jobs[i].out; fj.j[i].accumulate = jobs[i].accumulate;
        most = std::max<int64_t>(most, jobs[i].n);
    }
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_BWD_DW, st);
    hipLaunchKernelGGL(pg_fold_kernel, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(cdiv(most, 1024), 1024)), (unsigned)count), dim3(256), 0, st, fj);
    return check_launch("papc_pg_fold_f32");
}

}  // extern "C"
