// mlp_gemm.hip -- fp32-MFMA row GEMM with fused operand producers and epilogues (gfx950).
//
//   forward  : y[M,Cout]  = A(x)[M,Cin] . w[Cout,Cin]^T + bias      (+ per-channel sum / sum-of-squares partials)
//   backward : dX[M,Cin]  = dY[M,Cout] . w[Cout,Cin]                 (dY recomputed in the load; optional scatter-add)
//
// Replaces nn.Conv2D(cin,cout,1) on the [B,C,K,S] tensors of
// /root/reference/PAPC/models/layers/pointnet2_basic_layers.py:189,215-217 (rows = (b,s,k), channel-contiguous).
//
// Shape of the problem: M is huge (up to 1M rows), K and N are small (3..1024).  Tiling for CDNA4:
//   * 256 threads = 4 waves per workgroup, 128-row M tile, BN in {128,64,32}; each wave owns WMxWN 32x32
//     accumulator tiles of v_mfma_f32_32x32x2_f32 (exact fp32, k-ordered fma chain).
//   * both operands sit in LDS K-contiguous with a 36-float row stride (9 x 16 B: odd number of 16-B slots ->
//     conflict-free ds_read_b128 for the 16-lane groups of the instruction); one ds_read_b128 per operand tile
//     feeds FOUR MFMAs: lanes 0-31 carry k = kk..kk+3, lanes 32-63 carry k = kk+4..kk+7.
//   * workgroups are persistent over row tiles (grid.x <= 512 = one residency wave), so the BN-statistics partials are one
//     deterministic row per workgroup, no atomics.
//   * software pipeline: the (row tile, k-chunk) pairs of a workgroup form one flat sequence of stages; stage
//     s+1's global loads are issued (raw, into registers) BEFORE the MFMAs of stage s and are transformed + written
//     to the other LDS buffer after them, so HBM/L2 latency hides under the matrix pipe and one barrier per stage
//     suffices (writes of stage s go to the buffer last read in stage s-1, which every wave left before the
//     previous barrier).  Two workgroups per CU: one computes while the other writes / stores.
#include "mlp_gemm.h"
#include <mutex>
#include <unordered_map>

namespace papc {

constexpr int LDT = 36;  // LDS row stride (floats)
constexpr int BK = 32;

// The thread's float4 of the weight tile: row n (output channel), internal channels k..k+3.  fetch_w4 ONLY issues
// loads (unconditional, clamped indices) so they stay in flight across the MFMA phase; mask_w4 zeroes the
// out-of-range elements when the registers are written to LDS.  WMAP / NMAP are compile-time (GROUP forward /
// scatter dX): internal channel order [feats, xyz] -> the caller's weight columns / rows through gk().
template <bool VEC, bool WMAP, bool NMAP>
__device__ __forceinline__ float4 fetch_w4(const GemmArgs &p, int n, int k)
{
    const int nn = n < p.Nout ? n : 0;
    const int nrow = NMAP ? gk(p.a.g, nn) : nn;
    const float *wr = p.w + (int64_t)nrow * p.ldw;
    float4 v;
    if (WMAP) {
        v.x = wr[k < p.Kin ? gk(p.a.g, k) : 0];
        v.y = wr[k + 1 < p.Kin ? gk(p.a.g, k + 1) : 0];
        v.z = wr[k + 2 < p.Kin ? gk(p.a.g, k + 2) : 0];
        v.w = wr[k + 3 < p.Kin ? gk(p.a.g, k + 3) : 0];
    } else if (VEC) {
        v = ld4(wr + (k < p.Kin ? k : 0));
    } else {
        v.x = wr[k < p.Kin ? k : 0];
        v.y = wr[k + 1 < p.Kin ? k + 1 : 0];
        v.z = wr[k + 2 < p.Kin ? k + 2 : 0];
        v.w = wr[k + 3 < p.Kin ? k + 3 : 0];
    }
    return v;
}
__device__ __forceinline__ float4 mask_w4(const GemmArgs &p, int n, int k, float4 v)
{
    const bool nok = n < p.Nout;
    if (!(nok && k < p.Kin)) v.x = 0.f;
    if (!(nok && k + 1 < p.Kin)) v.y = 0.f;
    if (!(nok && k + 2 < p.Kin)) v.z = 0.f;
    if (!(nok && k + 3 < p.Kin)) v.w = 0.f;
    return v;
}

// Per iteration (stage s):   issue raw loads of stage s+1  ->  MFMAs of stage s (LDS buffer s&1)  ->  consume the loads
// (transform, write LDS buffer (s+1)&1)  ->  epilogue stores if s closed a row tile  ->  LDS-only barrier.
// The consume step sits BEFORE the stores on purpose: gfx9-family vmcnt counts loads and stores together and, with both
// kinds pending, a wait for a load degenerates to vmcnt(0); ordered this way the stores are the youngest VMEM ops of
// the iteration and drain under the next stage's MFMAs (the barrier does not wait for them: lds_barrier()).
//
// BF3 = true: the operands are split into three bf16 planes when they are written to LDS and the products run on
// v_mfma_f32_32x32x16_bf16 (six per 32x32x16 block, see split3 in mlp_loaders.h) -- fp32 accuracy at 2.7x the fp32 matrix
// rate.  A stage is then one 16-wide k block: LDS rows are [plane0 | plane1 | plane2] x 32 B + 16 B pad = 112 B (7 x 16 B,
// odd -> the 16-lane groups of a ds_read_b128 hit 16 distinct 16-B slots); a lane's 16-B read at plane*32 + 16*(lane>>5) is
// its 8 consecutive k of row lane&31, exactly the instruction's A/B operand.  BF3 = false is the exact fp32 path
// (v_mfma_f32_32x32x2_f32), kept selectable (PAPC_GEMM_F32=1) as the A/B reference.
//
// WS = G > 0 (wave-specialised, BF3 only): the workgroup has 4 CONSUMER waves that own the accumulators and do nothing but
// ds_read + MFMA (+ the epilogue), and G groups of 4 PRODUCER waves that do nothing but global loads, the operand
// transform, the bf16 split and the LDS writes.  A consumer and G producer waves share each SIMD, so the VALU work of the
// transform runs in the shadow of another wave's MFMAs by construction (no reliance on the compiler interleaving one
// wave's instruction stream).  Producer group g owns the stages s = g (mod G): it issues the loads of its next stage right
// after finishing the previous one and only touches them G - 1 slots later, so the HBM latency is covered by plain
// issue -> wait -> transform code per wave (register rings across a loop back-edge do not survive hipcc's waitcnt
// insertion: it drains to vmcnt(0)).  Same two LDS buffers and the same single barrier per stage as the unspecialised loop.
//
// TL = true (dX kernels with a dense store): the MFMA operands are swapped, so the accumulator tile is held TRANSPOSED --
// a lane owns one output ROW (m = tile row lane&31) and 16 columns in four groups of 4 consecutive ones.  The epilogue then
// moves 16 bytes per lane per instruction (global_store_dwordx4, and dwordx4 loads of the previous layer's y for the fused
// BN-backward sums) instead of 4: a quarter of the memory instructions, which is what the dX epilogues were bound by
// (DESIGN.md 3.7).  The per-column sums become 16 per-lane accumulators per 32-column tile, reduced across lanes once at
// the end of the kernel (DPP tree, fixed order).
template <int AMODE, int EPI, bool VEC, int WGM, int WGN, int WM, int WN, int DEPTH, bool BF3, int WS, bool TL, int KB = 1>
__global__ __launch_bounds__((1 + WS) * WGM * WGN * 64, WS ? (1 + WS) : WGM * WGN / 2) void gemm_kernel(GemmArgs p)
{
    static_assert(!TL || ((AMODE == A_DY_DENSE || AMODE == A_DY_MAX) && (EPI == EPI_STORE || EPI == EPI_STORE_RED) && VEC), "TL: dense dX stores");
    constexpr int BM = WGM * WM * 32, BN = WGN * WN * 32;
    static_assert(KB == 1 || (BF3 && !WS && KB == 2), "KB: k blocks of 16 per stage (split-bf16 kernel)");
    constexpr int BKS = BF3 ? 16 * KB : BK;   // k extent of one stage (KB = 2: half as many, fatter stages for the latency-bound
                                              // few-row-tile problems that run one workgroup per CU anyway)
    constexpr int KQN = BKS / 4;              // threads along k (one float4 each)
    constexpr int NC = WGM * WGN * 64;        // threads that own accumulators (4 or 8 waves)
    constexpr int NT = (1 + WS) * NC;         // threads per workgroup
    constexpr int RPI = NC / KQN;             // rows covered by one load iteration of the loader threads
    static_assert(!WS || (BF3 && WGM * WGN == 4 && DEPTH == 1 && WS <= 3), "wave specialisation: BF3, 4 + 4G waves");
    constexpr int NAI = BM / RPI;             // A float4 loads per thread per stage
    constexpr int NWL = (BN + RPI - 1) / RPI; // weight float4 loads per thread per stage
    constexpr bool WPART = BN < RPI;          // only the first BN row-threads carry a weight row (wave-uniform)
    constexpr int PLB = 32 * KB;              // bytes of one bf16 plane of a row
    constexpr int ROWB = 3 * PLB + 16;        // BF3 LDS row stride in bytes (an odd number of 16-byte slots: 112 / 208)
    constexpr int ROWF = BF3 ? ROWB / 4 : LDT;
    constexpr bool WMAP = (AMODE == A_GROUP);
    constexpr bool NMAP = (EPI == EPI_SCATTER);
    static_assert(WGM * WGN == 4 || WGM * WGN == 8, "4 or 8 waves");
    static_assert(BM == 128, "row tile is 128");
    constexpr int STAGE = (BM + BN) * ROWF;
    constexpr bool TLRED = TL && EPI == EPI_STORE_RED;
#ifndef PAPC_GEMM_EARLY
#define PAPC_GEMM_EARLY 1
#endif
    constexpr bool EARLY = PAPC_GEMM_EARLY && DEPTH == 1;
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE + 2 * WGM * BN + (TLRED ? 4 * BN : 0)];
    float *red = smem + 2 * STAGE;
    float *cst = red + 2 * WGM * BN;   // TLRED: previous layer's scale / shift / mean / invstd of this column block

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hi = lane >> 5, l31 = lane & 31;
    const bool producer = WS && wave >= WGM * WGN;          // wave-uniform role
    const int pgrp = producer ? (wave - WGM * WGN) / (WGM * WGN) : 0;   // producer group
    const int ltid = producer ? tid - NC * (1 + pgrp) : tid;            // loader thread index within its group
    const int cwave = producer ? 0 : wave;
    const int wgm = cwave / WGN, wgn = cwave % WGN;
    const int n0 = blockIdx.y * BN;
    const int64_t n_mtiles = (p.M + BM - 1) / BM;
    const int Kpad = BF3 ? ((p.Kin + 15) & ~15) : ((p.Kin + 7) & ~7);
    const int n_kc = (Kpad + BKS - 1) / BKS;
    const int kq = (ltid % KQN) * 4;
    const int r0 = ltid / KQN;  // 0..RPI-1
    const bool wrow = !WPART || r0 < BN;
    const bool use_jpre = (AMODE == A_GROUP) && p.a.g.idx != nullptr;

    float s1[WN], s2[WN], biasv[WN];
    bool cok[WN];
#pragma unroll
    for (int wn = 0; wn < WN; ++wn) {
        s1[wn] = 0.f; s2[wn] = 0.f;
        const int col = n0 + (wgn * WN + wn) * 32 + l31;
        cok[wn] = col < p.Nout;
        biasv[wn] = ((EPI == EPI_STORE || EPI == EPI_STORE_GMAX || EPI == EPI_STORE_RED) && p.bias) ? p.bias[cok[wn] ? col : 0] : 0.f;
    }
    float rsc[WN], rsh[WN], rmu[WN], ris[WN];  // EPI_STORE_RED: previous layer's BN constants of this lane's columns
#pragma unroll
    for (int wn = 0; wn < WN; ++wn) {
        const int col = n0 + (wgn * WN + wn) * 32 + l31;
        const int cc = cok[wn] ? col : 0;
        rsc[wn] = rsh[wn] = rmu[wn] = ris[wn] = 0.f;
        if (EPI == EPI_STORE_RED) { rsc[wn] = p.rd.scale[cc]; rsh[wn] = p.rd.shift[cc]; rmu[wn] = p.rd.mean[cc]; ris[wn] = p.rd.invstd[cc]; }
    }

    constexpr int NS = TLRED ? 16 : 1;
    float st1[WN][NS], st2[WN][NS];   // TLRED: per-lane column sums (column = the lane's r-th accumulator column)
#pragma unroll
    for (int wn = 0; wn < WN; ++wn)
#pragma unroll
        for (int r = 0; r < NS; ++r) { st1[wn][r] = 0.f; st2[wn][r] = 0.f; }
    if (TLRED) {
        for (int i = tid; i < BN; i += NT) {
            const int cc = n0 + i < p.Nout ? n0 + i : 0;
            cst[0 * BN + i] = p.rd.scale[cc]; cst[1 * BN + i] = p.rd.shift[cc]; cst[2 * BN + i] = p.rd.mean[cc]; cst[3 * BN + i] = p.rd.invstd[cc];
        }
    }

    floatx16 acc[WM][WN];
#pragma unroll
    for (int wm = 0; wm < WM; ++wm)
#pragma unroll
        for (int wn = 0; wn < WN; ++wn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[wm][wn][r] = 0.f;

    // ---- pipeline state.  A `Stg` is the register image of one in-flight stage (raw loads + what is needed to finish them).
    struct Stg {
        RowCtx rows[NAI];
        Raw3 ra[NAI];
        float4 rw[NWL];
        KConst kc;
        int kf;          // this thread's first channel of the stage's k-chunk
        int kci;         // k-chunk index
        int64_t tile;    // row tile
        bool ok;         // the stage exists
    };
    Stg sa = {}, sb = {};
    RowCtx rows_f[NAI] = {};   // row contexts of the fetch cursor's tile
    int64_t tile_f = blockIdx.x;  // fetch cursor: next stage to issue = (tile_f, kc_f)
    int kc_f = 0;
    int jcur[NAI], jpre[NAI];  // GROUP: neighbour indices of the fetch cursor's tile and of the tile after it (loaded one tile early)
#pragma unroll
    for (int i = 0; i < NAI; ++i) { jcur[i] = -2; jpre[i] = -2; }

    auto load_j = [&](int64_t tile) {  // issue the idx loads of `tile`'s rows (clamped: always in range)
        if (use_jpre) {
#pragma unroll
            for (int i = 0; i < NAI; ++i) {
                const int64_t m = tile * BM + r0 + RPI * i;
                jpre[i] = p.a.g.idx[m < p.M ? m : 0];
            }
        }
    };
    // issue the loads of the stage under the fetch cursor into `s` (loads only: nothing here touches the values) and
    // advance the cursor
    auto issue = [&](Stg &s) {
        s.ok = tile_f < n_mtiles;
        s.tile = tile_f; s.kci = kc_f; s.kf = kc_f * BKS + kq;
        if (s.ok) {
            if (kc_f == 0) {   // row contexts (base pointers, group / validity) are per tile: the other k-chunks reuse them
#pragma unroll
                for (int i = 0; i < NAI; ++i) rows_f[i] = make_row<AMODE>(p.a, tile_f * BM + r0 + RPI * i, p.M, use_jpre ? jcur[i] : -2);
            }
#pragma unroll
            for (int i = 0; i < NAI; ++i) s.rows[i] = rows_f[i];   // (only `valid` / `kin` stay live in the stage image)
            s.kc = make_kconst<AMODE, VEC>(p.a, s.kf, p.Kin);
#pragma unroll
            for (int i = 0; i < NAI; ++i) s.ra[i] = fetch_a4<AMODE, VEC>(p.a, rows_f[i], s.kf, p.Kin);
            if (wrow) {
#pragma unroll
                for (int i = 0; i < NWL; ++i) s.rw[i] = fetch_w4<VEC, WMAP, NMAP>(p, n0 + r0 + RPI * i, s.kf);
            }
        }
        kc_f += 1;
        if (kc_f == n_kc) {
            kc_f = 0; tile_f += gridDim.x;
#pragma unroll
            for (int i = 0; i < NAI; ++i) jcur[i] = jpre[i];
            load_j(tile_f + gridDim.x);
        }
    };
    // consume the landed registers of `s`: transform + write one LDS buffer
    auto consume = [&](const Stg &s, float *As) {
        if (BF3) {
            char *Ab = reinterpret_cast<char *>(As), *Wb = Ab + BM * ROWB;
#pragma unroll
            for (int i = 0; i < NAI; ++i) {
                uint2 q0, q1, q2;
                split3(finish_a4<AMODE, VEC>(p.a, s.rows[i], s.kf, p.Kin, s.kc, s.ra[i]), q0, q1, q2);
                char *d = Ab + (r0 + RPI * i) * ROWB + kq * 2;
                *reinterpret_cast<uint2 *>(d) = q0; *reinterpret_cast<uint2 *>(d + PLB) = q1; *reinterpret_cast<uint2 *>(d + 2 * PLB) = q2;
            }
            if (wrow) {
#pragma unroll
                for (int i = 0; i < NWL; ++i) {
                    uint2 q0, q1, q2;
                    split3(mask_w4(p, n0 + r0 + RPI * i, s.kf, s.rw[i]), q0, q1, q2);
                    char *d = Wb + (r0 + RPI * i) * ROWB + kq * 2;
                    *reinterpret_cast<uint2 *>(d) = q0; *reinterpret_cast<uint2 *>(d + PLB) = q1; *reinterpret_cast<uint2 *>(d + 2 * PLB) = q2;
                }
            }
            return;
        }
        float *Ws = As + BM * LDT;
#pragma unroll
        for (int i = 0; i < NAI; ++i)
            *reinterpret_cast<float4 *>(&As[(r0 + RPI * i) * LDT + kq]) = finish_a4<AMODE, VEC>(p.a, s.rows[i], s.kf, p.Kin, s.kc, s.ra[i]);
#pragma unroll
        for (int i = 0; i < NWL; ++i)
            *reinterpret_cast<float4 *>(&Ws[(r0 + RPI * i) * LDT + kq]) = mask_w4(p, n0 + r0 + RPI * i, s.kf, s.rw[i]);
    };

    // ---- prologue: stage 0 goes through LDS synchronously; DEPTH - 1 further stages are put in flight
    if (use_jpre && !WS) {
        load_j(tile_f);
#pragma unroll
        for (int i = 0; i < NAI; ++i) jcur[i] = jpre[i];
        load_j(tile_f + gridDim.x);
    }
    if (!WS) issue(sa);
    bool have = sa.ok;
    int64_t tile_c = sa.tile;  // tile / chunk of the stage being computed
    int kc_c = sa.kci;
    if (have && !WS) consume(sa, smem);
    if (DEPTH == 2 || (EARLY && !WS)) issue(sa);
    if (!WS) __syncthreads();

    // ---- MFMAs of one stage from LDS buffer `buf` (kc = its k-chunk index)
    auto mfma_stage = [&](int buf, int kc_c) {
        if (BF3) {
            const char *Ab = reinterpret_cast<const char *>(smem + buf * STAGE), *Wb = Ab + BM * ROWB;
            // smallest terms first; consecutive MFMAs go to different accumulators
            constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                bf16x8 af[WM][3], bq[WN][3];
#pragma unroll
                for (int wm = 0; wm < WM; ++wm)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        af[wm][pl] = *reinterpret_cast<const bf16x8 *>(Ab + ((wgm * WM + wm) * 32 + l31) * ROWB + pl * PLB + kb * 32 + hi * 16);
#pragma unroll
                for (int wn = 0; wn < WN; ++wn)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        bq[wn][pl] = *reinterpret_cast<const bf16x8 *>(Wb + ((wgn * WN + wn) * 32 + l31) * ROWB + pl * PLB + kb * 32 + hi * 16);
#pragma unroll
                for (int t = 0; t < 6; ++t)
#pragma unroll
                    for (int wm = 0; wm < WM; ++wm)
#pragma unroll
                        for (int wn = 0; wn < WN; ++wn)
                            acc[wm][wn] = TL ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[wn][PB[t]], af[wm][PA[t]], acc[wm][wn], 0, 0, 0)
                                             : __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[wm][PA[t]], bq[wn][PB[t]], acc[wm][wn], 0, 0, 0);
            }
        } else {
            const float *As = smem + buf * STAGE, *Ws = As + BM * LDT;
            const int kend = min(BK, Kpad - kc_c * BK);
            for (int kk = 0; kk < kend; kk += 8) {
                float4 af[WM], bf[WN];
#pragma unroll
                for (int wm = 0; wm < WM; ++wm)
                    af[wm] = *reinterpret_cast<const float4 *>(&As[((wgm * WM + wm) * 32 + l31) * LDT + kk + 4 * hi]);
#pragma unroll
                for (int wn = 0; wn < WN; ++wn)
                    bf[wn] = *reinterpret_cast<const float4 *>(&Ws[((wgn * WN + wn) * 32 + l31) * LDT + kk + 4 * hi]);
#pragma unroll
                for (int wm = 0; wm < WM; ++wm)
#pragma unroll
                    for (int wn = 0; wn < WN; ++wn) {
                        if (TL) {
                            acc[wm][wn] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[wn].x, af[wm].x, acc[wm][wn], 0, 0, 0);
                            acc[wm][wn] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[wn].y, af[wm].y, acc[wm][wn], 0, 0, 0);
                            acc[wm][wn] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[wn].z, af[wm].z, acc[wm][wn], 0, 0, 0);
                            acc[wm][wn] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[wn].w, af[wm].w, acc[wm][wn], 0, 0, 0);
                        } else {
                        acc[wm][wn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[wm].x, bf[wn].x, acc[wm][wn], 0, 0, 0);
                        acc[wm][wn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[wm].y, bf[wn].y, acc[wm][wn], 0, 0, 0);
                        acc[wm][wn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[wm].z, bf[wn].z, acc[wm][wn], 0, 0, 0);
                        acc[wm][wn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[wm].w, bf[wn].w, acc[wm][wn], 0, 0, 0);
                        }
                    }
            }
        }

    };
    // ---- epilogue of row tile `tile_c` (its last k-chunk has been accumulated).  C/D layout: col = lane&31,
    // row = (r&3) + 8*(r>>2) + 4*(lane>>5).  Straight-line stores (no per-element branch) for full tiles; the ragged last
    // tile takes the predicated path.
    auto epilogue = [&](int64_t tile_c) {
        if (TL) {
            // transposed accumulators: lane = output row, register r = column (r&3) + 8*(r>>2) + 4*(lane>>5) of the 32-column tile
            const int64_t m0 = tile_c * BM;
            const bool full = m0 + BM <= p.M && n0 + BN <= p.Nout;   // whole tile inside the matrix: no predicates at all
#pragma unroll
            for (int wm = 0; wm < WM; ++wm) {
                const int64_t m = m0 + (wgm * WM + wm) * 32 + l31;
                const bool rok = m < p.M;
                const int64_t mc = rok ? m : m0;
#pragma unroll
                for (int wn = 0; wn < WN; ++wn) {
                    const int cl0 = (wgn * WN + wn) * 32 + 4 * hi;   // column (inside the block) of accumulator 0
                    float4 yv[4];
                    if (EPI == EPI_STORE_RED) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int col = n0 + cl0 + 8 * q;
                            yv[q] = ld4(p.rd.y + mc * p.ldy + (col < p.Nout ? col : 0));
                        }
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int col = n0 + cl0 + 8 * q;
                        const float4 v = make_float4(acc[wm][wn][4 * q], acc[wm][wn][4 * q + 1], acc[wm][wn][4 * q + 2], acc[wm][wn][4 * q + 3]);
                        if (EPI == EPI_STORE_RED) {
                            const float4 sc = *reinterpret_cast<const float4 *>(&cst[0 * BN + cl0 + 8 * q]);
                            const float4 sh = *reinterpret_cast<const float4 *>(&cst[1 * BN + cl0 + 8 * q]);
                            const float4 mu = *reinterpret_cast<const float4 *>(&cst[2 * BN + cl0 + 8 * q]);
                            const float4 is = *reinterpret_cast<const float4 *>(&cst[3 * BN + cl0 + 8 * q]);
                            // rows / columns outside the matrix carry v = 0 (zero operand rows / zero weight rows): they add nothing
                            float pp;
                            pp = fmaf(sc.x, yv[q].x, sh.x) > 0.f ? v.x : 0.f; st1[wn][4 * q + 0] += pp; st2[wn][4 * q + 0] = fmaf(pp, (yv[q].x - mu.x) * is.x, st2[wn][4 * q + 0]);
                            pp = fmaf(sc.y, yv[q].y, sh.y) > 0.f ? v.y : 0.f; st1[wn][4 * q + 1] += pp; st2[wn][4 * q + 1] = fmaf(pp, (yv[q].y - mu.y) * is.y, st2[wn][4 * q + 1]);
                            pp = fmaf(sc.z, yv[q].z, sh.z) > 0.f ? v.z : 0.f; st1[wn][4 * q + 2] += pp; st2[wn][4 * q + 2] = fmaf(pp, (yv[q].z - mu.z) * is.z, st2[wn][4 * q + 2]);
                            pp = fmaf(sc.w, yv[q].w, sh.w) > 0.f ? v.w : 0.f; st1[wn][4 * q + 3] += pp; st2[wn][4 * q + 3] = fmaf(pp, (yv[q].w - mu.w) * is.w, st2[wn][4 * q + 3]);
                        }
                        if (full || (rok && col < p.Nout)) *reinterpret_cast<float4 *>(p.y + m * p.ldy + col) = v;
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[wm][wn][r] = 0.f;
                }
            }
            return;
        }

            const int64_t m0 = tile_c * BM;
            const bool full = m0 + BM <= p.M;
#pragma unroll
            for (int wn = 0; wn < WN; ++wn) {
                const int col = n0 + (wgn * WN + wn) * 32 + l31;
                if (cok[wn]) {
                    float dsum = 0.f;                    // scatter epilogue: pending sum of padding duplicates ...
                    int dgrp = -1, djf = -1, dbat = 0;   // ... of group dgrp (first neighbour djf, cloud dbat)
                    if (EPI == EPI_STORE_GMAX) {
                        // store y, accumulate the BN statistics, and reduce max / min (+ first offsets) over each group of K rows.
                        // A lane holds 16 of a 32-row tile's rows, ascending in r; lane^32 holds the other 16.
                        float tmx[WM], tmn[WM];
                        int tix[WM], tin[WM];
#pragma unroll
                        for (int wm = 0; wm < WM; ++wm) {
                            const int64_t rb = m0 + (wgm * WM + wm) * 32 + 4 * hi;
                            float *yp = p.y + rb * p.ldy + col;
                            float vmx = -INFINITY, vmn = INFINITY;
                            int imx = 0, imn = 0;
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const int ro = (r & 3) + 8 * (r >> 2);
                                const float v = acc[wm][wn][r] + biasv[wn];
                                yp[(int64_t)ro * p.ldy] = v;
                                s1[wn] += v;
                                s2[wn] = fmaf(v, v, s2[wn]);
                                if (v > vmx) { vmx = v; imx = ro + 4 * hi; }
                                if (v < vmn) { vmn = v; imn = ro + 4 * hi; }
                            }
                            const float omx = __shfl_xor(vmx, 32), omn = __shfl_xor(vmn, 32);
                            const int oix = __shfl_xor(imx, 32), oin = __shfl_xor(imn, 32);
                            if (omx > vmx || (omx == vmx && oix < imx)) { vmx = omx; imx = oix; }
                            if (omn < vmn || (omn == vmn && oin < imn)) { vmn = omn; imn = oin; }
                            tmx[wm] = vmx; tix[wm] = imx; tmn[wm] = vmn; tin[wm] = imn;   // per 32-row tile, both halves agree
                        }
                        const int K = p.gm.K;
                        if (K == 32) {
                            if (hi == 0) {
#pragma unroll
                                for (int wm = 0; wm < WM; ++wm) {
                                    const int64_t g = (m0 >> 5) + (wgm * WM + wm);
                                    p.gm.gmax[g * p.Nout + col] = tmx[wm]; p.gm.amax[g * p.Nout + col] = tix[wm];
                                    p.gm.gmin[g * p.Nout + col] = tmn[wm]; p.gm.amin[g * p.Nout + col] = tin[wm];
                                }
                            }
                        } else {
                            // K = 64: this wave's WM = 2 tiles form one group; K = 128: plus the partner wave (other wgm) through LDS
                            float gx = tmx[0], gn = tmn[0];
                            int ix = tix[0], in_ = tin[0];
#pragma unroll
                            for (int wm = 1; wm < WM; ++wm) {
                                if (tmx[wm] > gx) { gx = tmx[wm]; ix = tix[wm] + 32 * wm; }
                                if (tmn[wm] < gn) { gn = tmn[wm]; in_ = tin[wm] + 32 * wm; }
                            }
                            if (K == 64) {
                                if (hi == 0) {
                                    const int64_t g = (m0 >> 6) + wgm;
                                    p.gm.gmax[g * p.Nout + col] = gx; p.gm.amax[g * p.Nout + col] = ix;
                                    p.gm.gmin[g * p.Nout + col] = gn; p.gm.amin[g * p.Nout + col] = in_;
                                }
                            } else {  // K == 128 (WGM == 2): rows of wgm = 1 come after those of wgm = 0
                                const int cl = (wgn * WN + wn) * 32 + l31;
                                float *xf = red;                                   // [4][BN]: max, min, (int) imax, imin of wgm = 1
                                if (wgm == 1 && hi == 0) {
                                    xf[0 * BN + cl] = gx; xf[1 * BN + cl] = gn;
                                    xf[2 * BN + cl] = __int_as_float(ix + 64); xf[3 * BN + cl] = __int_as_float(in_ + 64);
                                }
                                lds_barrier();
                                if (wgm == 0 && hi == 0) {
                                    const float ox = xf[0 * BN + cl], on = xf[1 * BN + cl];
                                    if (ox > gx) { gx = ox; ix = __float_as_int(xf[2 * BN + cl]); }
                                    if (on < gn) { gn = on; in_ = __float_as_int(xf[3 * BN + cl]); }
                                    const int64_t g = m0 >> 7;
                                    p.gm.gmax[g * p.Nout + col] = gx; p.gm.amax[g * p.Nout + col] = ix;
                                    p.gm.gmin[g * p.Nout + col] = gn; p.gm.amin[g * p.Nout + col] = in_;
                                }
                            }
                        }
                    }
                    if (EPI == EPI_STORE_RED) {
                        // per 32-row tile: issue the previous layer's y for all 16 of this lane's rows first (one exposed
                        // latency), then store dz_prev and accumulate p = dz*[z>0], p*xhat
                        // RB 32-row tiles at a time (all of the lane's rows of the column on the 4-wave kernel, which has the registers)
                        constexpr int RB = (WGM * WGN == 4 && WN == 2 && !WS) ? WM : 1;   // (measured: the 64-wide tile is faster unbatched)
#pragma unroll
                        for (int w0 = 0; w0 < WM; w0 += RB) {
                            float yv[RB][16];
#pragma unroll
                            for (int wi = 0; wi < RB; ++wi) {
                                const int64_t rb = m0 + (wgm * WM + w0 + wi) * 32 + 4 * hi;
                                const float *qp = p.rd.y + rb * p.ldy + col;
#pragma unroll
                                for (int r = 0; r < 16; ++r) {
                                    const int ro = (r & 3) + 8 * (r >> 2);
                                    yv[wi][r] = qp[(int64_t)((full || rb + ro < p.M) ? ro : 0) * p.ldy];
                                }
                            }
#pragma unroll
                            for (int wi = 0; wi < RB; ++wi) {
                                const int wm = w0 + wi;
                                const int64_t rb = m0 + (wgm * WM + wm) * 32 + 4 * hi;
                                float *yp = p.y + rb * p.ldy + col;
#pragma unroll
                                for (int r = 0; r < 16; ++r) {
                                    const int ro = (r & 3) + 8 * (r >> 2);
                                    if (full || rb + ro < p.M) {
                                        const float v = acc[wm][wn][r] + biasv[wn];   // (bias: the A_MAXCAT dX only, else 0)
                                        const float pp = fmaf(rsc[wn], yv[wi][r], rsh[wn]) > 0.f ? v : 0.f;
                                        yp[(int64_t)ro * p.ldy] = p.rd.masked ? pp : v;     // (masked: the value the sums are formed from)
                                        s1[wn] += pp;
                                        s2[wn] = fmaf(pp, (yv[wi][r] - rmu[wn]) * ris[wn], s2[wn]);
                                    }
                                }
                            }
                        }
                    }
#pragma unroll
                    for (int wm = 0; wm < WM; ++wm) {
                        const int64_t rb = m0 + (wgm * WM + wm) * 32 + 4 * hi;
                        if (EPI == EPI_STORE_RED || EPI == EPI_STORE_GMAX) {
                            // handled above (all row tiles of this column at once)
                        } else if (EPI == EPI_STORE) {
                            float *yp = p.y + rb * p.ldy + col;
                            if (full) {
#pragma unroll
                                for (int r = 0; r < 16; ++r) {
                                    const float v = acc[wm][wn][r] + biasv[wn];
                                    yp[(int64_t)((r & 3) + 8 * (r >> 2)) * p.ldy] = v;
                                    s1[wn] += v;
                                    s2[wn] = fmaf(v, v, s2[wn]);
                                }
                            } else {
#pragma unroll
                                for (int r = 0; r < 16; ++r) {
                                    const int ro = (r & 3) + 8 * (r >> 2);
                                    if (rb + ro < p.M) {
                                        const float v = acc[wm][wn][r] + biasv[wn];
                                        yp[(int64_t)ro * p.ldy] = v;
                                        s1[wn] += v;
                                        s2[wn] = fmaf(v, v, s2[wn]);
                                    }
                                }
                            }
                        } else {
                            // gradient of index_points: grad_feats[b, idx[m], col] += dX[m, col] (feature columns only).
                            // Ball-query padding repeats each group's FIRST neighbour (pointnet2_basic_layers.py:118-124),
                            // often for half the nsample slots: those rows would hammer one L2 line with dependent atomics,
                            // so they are summed in a register per group and flushed with a single atomic.
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const int64_t row = rb + (r & 3) + 8 * (r >> 2);
                                if (row < p.M) {
                                    const float v = acc[wm][wn][r];
                                    const int b = (int)fdiv((uint32_t)row, p.sc.divSK);
                                    if (p.sc.idx) {
                                        const int g = (int)fdiv((uint32_t)row, p.sc.divK);
                                        const int jf = p.sc.idx[(int64_t)g * p.sc.K];
                                        const int j = p.sc.idx[row];
                                        if (j == jf && (int)row != g * p.sc.K) {
                                            if (g != dgrp) {
                                                if (dgrp >= 0 && djf >= 0 && djf < p.sc.N)
                                                    unsafeAtomicAdd(&p.sc.gf[((int64_t)dbat * p.sc.N + djf) * p.sc.D + col], dsum);
                                                dsum = 0.f; dgrp = g; djf = jf; dbat = b;
                                            }
                                            dsum += v;
                                        } else if (j >= 0 && j < p.sc.N) {
                                            unsafeAtomicAdd(&p.sc.gf[((int64_t)b * p.sc.N + j) * p.sc.D + col], v);
                                        }
                                    } else {
                                        const int j = (int)row - b * p.sc.S * p.sc.K;
                                        unsafeAtomicAdd(&p.sc.gf[((int64_t)b * p.sc.N + j) * p.sc.D + col], v);
                                    }
                                }
                            }
                        }
                    }
                    if (EPI == EPI_SCATTER && dgrp >= 0 && djf >= 0 && djf < p.sc.N)
                        unsafeAtomicAdd(&p.sc.gf[((int64_t)dbat * p.sc.N + djf) * p.sc.D + col], dsum);
                }
#pragma unroll
                for (int wm = 0; wm < WM; ++wm)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[wm][wn][r] = 0.f;
            }
            };

    if (WS) {
        // every wave runs the same number of stage slots, hence the same number of barriers
        const int64_t my_tiles = n_mtiles > (int64_t)blockIdx.x ? (n_mtiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
        const int64_t n_stages = my_tiles * n_kc;
        const bool xbar = (EPI == EPI_STORE_GMAX) && p.gm.K == 128;   // the K = 128 group-max exchange has its own barriers
        if (producer) {
            // this group's next stage: index ns, tile nt, k-chunk nk; `last` = tile whose row contexts are in rows_f
            int64_t ns = pgrp, nt = blockIdx.x, last = -1;
            int nk = pgrp;
            while (nk >= n_kc) { nk -= n_kc; nt += gridDim.x; }
            auto issue_ws = [&](Stg &st) {
                st.ok = ns < n_stages;
                st.tile = nt; st.kci = nk; st.kf = nk * BKS + kq;
                if (st.ok) {
                    if (nt != last) {   // row contexts (base pointers, group / validity) are per tile
#pragma unroll
                        for (int i = 0; i < NAI; ++i) rows_f[i] = make_row<AMODE>(p.a, nt * BM + r0 + RPI * i, p.M, -2);
                        last = nt;
                    }
#pragma unroll
                    for (int i = 0; i < NAI; ++i) st.rows[i] = rows_f[i];
                    st.kc = make_kconst<AMODE, VEC>(p.a, st.kf, p.Kin);
#pragma unroll
                    for (int i = 0; i < NAI; ++i) st.ra[i] = fetch_a4<AMODE, VEC>(p.a, rows_f[i], st.kf, p.Kin);
                    if (wrow) {
#pragma unroll
                        for (int i = 0; i < NWL; ++i) st.rw[i] = fetch_w4<VEC, WMAP, NMAP>(p, n0 + r0 + RPI * i, st.kf);
                    }
                }
            };
            auto advance = [&]() {   // to this group's next stage
                ns += WS; nk += WS;
                while (nk >= n_kc) { nk -= n_kc; nt += gridDim.x; }
            };
            issue_ws(sa);
            if (pgrp == 0) {         // stage 0 goes through LDS before the first slot
                if (sa.ok) consume(sa, smem);
                advance();
                issue_ws(sa);
            }
            __syncthreads();
            int kcs = 0;
            unsigned long long tw = 0, tc = 0, ti = 0, tb = 0;
            for (int64_t t = 0; t < n_stages; ++t) {
                const unsigned long long c0 = p.dbg ? __builtin_readcyclecounter() : 0;
                unsigned long long c1 = c0, c2 = c0, c3 = c0;
                if (ns == t + 1) {   // this group's stage is the one the consumers need next: finish it into the other buffer
                    if (p.dbg) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); c1 = __builtin_readcyclecounter(); }
                    if (sa.ok) consume(sa, smem + (int)((t + 1) & 1) * STAGE);
                    if (p.dbg) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); c2 = __builtin_readcyclecounter(); }
                    advance();
                    issue_ws(sa);    // ... and put the group's next stage in flight for the coming WS - 1 slots
                    if (p.dbg) c3 = __builtin_readcyclecounter();
                    tw += c1 - c0; tc += c2 - c1; ti += c3 - c2;
                }
                if (kcs == n_kc - 1) {
                    kcs = 0;
                    if (xbar) {
#pragma unroll
                        for (int wn = 0; wn < WN; ++wn) lds_barrier();
                    }
                } else ++kcs;
                const unsigned long long c4 = p.dbg ? __builtin_readcyclecounter() : 0;
                lds_barrier();
                if (p.dbg) tb += __builtin_readcyclecounter() - c4;
            }
            if (p.dbg && blockIdx.y == 0 && (tid & 63) == 0 && ((wave - 4) & 3) == 0) {
                unsigned long long *d = p.dbg + ((int64_t)blockIdx.x * 4 + 1 + pgrp) * 8;
                d[0] = tw; d[1] = tc; d[2] = ti; d[3] = tb; d[4] = (unsigned long long)n_stages;
            }
        } else {
            __syncthreads();
            int buf = 0, kcc = 0;
            int64_t tile_c = blockIdx.x;
            unsigned long long tm = 0, te = 0, tb = 0;
            for (int64_t t = 0; t < n_stages; ++t) {
                const unsigned long long c0 = p.dbg ? __builtin_readcyclecounter() : 0;
                mfma_stage(buf, kcc);
                unsigned long long c1 = c0;
                if (p.dbg) { asm volatile("s_nop 0" ::: "memory"); c1 = __builtin_readcyclecounter(); }
                if (kcc == n_kc - 1) { epilogue(tile_c); tile_c += gridDim.x; kcc = 0; } else ++kcc;
                const unsigned long long c2 = p.dbg ? __builtin_readcyclecounter() : 0;
                lds_barrier();
                if (p.dbg) { const unsigned long long c3 = __builtin_readcyclecounter(); tm += c1 - c0; te += c2 - c1; tb += c3 - c2; }
                buf ^= 1;
            }
            if (p.dbg && blockIdx.y == 0 && tid == 0) {
                unsigned long long *d = p.dbg + ((int64_t)blockIdx.x * 4) * 8;
                d[0] = tm; d[1] = te; d[2] = tb; d[4] = (unsigned long long)n_stages;
            }
        }
    }

    int buf = 0;
    // one pipeline step: issue a new stage into `si`, compute the current stage from LDS, then finish the OLDEST in-flight
    // stage `sc` (the next one to compute) into the other LDS buffer.  DEPTH 1: si == sc; DEPTH 2: the two sets alternate, so
    // every stage's loads get two MFMA phases to land.
    unsigned long long t_is = 0, t_mf = 0, t_co = 0, t_ep = 0, t_ba = 0, n_st = 0;
    auto step = [&](Stg &si, Stg &sc) -> bool {
        if (!have) return false;
        const unsigned long long c0 = p.dbg ? __builtin_readcyclecounter() : 0;
        if (!EARLY) issue(si);
        const unsigned long long c1 = p.dbg ? __builtin_readcyclecounter() : 0;
        mfma_stage(buf, kc_c);
        const unsigned long long c2 = p.dbg ? __builtin_readcyclecounter() : 0;
        // ---- consume the next stage's loads: transform, write the other LDS buffer
        const int64_t nx_tile = sc.tile; const int nx_kc = sc.kci; const bool nx_ok = sc.ok;
        if (sc.ok) consume(sc, smem + (buf ^ 1) * STAGE);
        // EARLY: the stage after next is issued right here, into the registers consume() just released -- its loads then have
        // the epilogue, the barrier and the whole next MFMA phase to land (one register set, no ring for hipcc to drain)
        if (EARLY) issue(si);
        unsigned long long c3 = c2;
        if (p.dbg) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); c3 = __builtin_readcyclecounter(); }
        if (kc_c == n_kc - 1) epilogue(tile_c);
        const unsigned long long c4 = p.dbg ? __builtin_readcyclecounter() : 0;
        lds_barrier();  // LDS-only: the epilogue's global stores and the idx prefetch stay in flight across it
        if (p.dbg) { const unsigned long long c5 = __builtin_readcyclecounter(); t_is += c1 - c0; t_mf += c2 - c1; t_co += c3 - c2; t_ep += c4 - c3; t_ba += c5 - c4; ++n_st; }
        buf ^= 1;
        tile_c = nx_tile;
        kc_c = nx_kc;
        have = nx_ok;
        return true;
    };
    if (!WS) {
        if (DEPTH == 2) { while (step(sb, sa) && step(sa, sb)) {} }
        else { while (step(sa, sa)) {} }
        if (p.dbg && blockIdx.y == 0 && tid == 0) {
            unsigned long long *d = p.dbg + (int64_t)blockIdx.x * 32;
            d[0] = t_is; d[1] = t_mf; d[2] = t_co; d[3] = t_ep; d[4] = n_st; d[5] = t_ba;
        }
    }

    if ((EPI == EPI_STORE || EPI == EPI_STORE_RED || EPI == EPI_STORE_GMAX) && p.stats) {
        if (TLRED) {
            // per-lane column sums -> sums over the 32 rows of each half wave (fixed DPP tree); lanes 31 / 63 hold them
#pragma unroll
            for (int wn = 0; wn < WN; ++wn)
#pragma unroll
                for (int r = 0; r < NS; ++r) {
                    const float a = half_sum_f32(st1[wn][r]), b = half_sum_f32(st2[wn][r]);
                    if (l31 == 31 && !producer) {
                        const int cl = (wgn * WN + wn) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                        red[(0 * WGM + wgm) * BN + cl] = a;
                        red[(1 * WGM + wgm) * BN + cl] = b;
                    }
                }
        } else {
#pragma unroll
        for (int wn = 0; wn < WN; ++wn) {
            s1[wn] += __shfl_xor(s1[wn], 32);
            s2[wn] += __shfl_xor(s2[wn], 32);
            if (hi == 0 && !producer) {
                red[(0 * WGM + wgm) * BN + (wgn * WN + wn) * 32 + l31] = s1[wn];
                red[(1 * WGM + wgm) * BN + (wgn * WN + wn) * 32 + l31] = s2[wn];
            }
        }
        }
        __syncthreads();
        for (int i = tid; i < 2 * BN; i += NT) {
            const int which = i / BN, c = i - which * BN;
            float t = 0.f;
#pragma unroll
            for (int g = 0; g < WGM; ++g) t += red[(which * WGM + g) * BN + c];
            if (n0 + c < p.Nout) {
                p.stats[((int64_t)blockIdx.x * 2 + which) * p.Nout + n0 + c] = t;
                for (int r = blockIdx.x + gridDim.x; r < p.parts; r += gridDim.x) p.stats[((int64_t)r * 2 + which) * p.Nout + n0 + c] = 0.f;
            }
        }
    }
}

static int gemm_max_parts()
{
    return knob(KNOB_PARTS);
}
static int gemm_parts(int64_t M) { return (int)std::min<int64_t>((M + 127) / 128, gemm_max_parts()); }

// Prefetch depth: the kernel supports DEPTH = 2 (two stages of loads in flight, register sets alternating), but a
// same-box A/B on MI355X showed no gain over DEPTH = 1 (loaded-memory latency is not what limits these kernels), so only
// DEPTH = 1 is instantiated.
static bool gemm_f32_exact()
{
    return knob(KNOB_GEMM_F32) == 1;
}
static bool gemm_waves8(int amode, int epi)   // 8-wave flavour of the 128x128 tile (PAPC_GEMM_WAVES=4|8 forces one; default per epilogue)
{
    const int v = knob(KNOB_GEMM_WAVES);
    if (v == 4) return false;
    if (v == 8) return true;
    if (amode == A_MAXCAT) return knob(KNOB_MAXCAT_WAVES) == 8;
    return !(amode == A_DY_MAX && epi == EPI_STORE_RED);   // measured per kernel (MI355X): only that one is faster on 4 waves (178 vs 223 us)
}
static int gemm_ws()   // producer groups of the wave-specialised 128x128 kernel (0 = unspecialised)
{
    return knob(KNOB_GEMM_WS) == 3 ? 3 : 0;   // opt-in: see DESIGN.md 3.7
}
static unsigned long long *g_dbg = nullptr;
static void dbg_report0(const GemmArgs &p, int amode, int epi, unsigned gx, int waves)
{
    hipDeviceSynchronize();
    static unsigned long long h[512 * 4 * 8];
    hipMemcpy(h, g_dbg, sizeof(h), hipMemcpyDeviceToHost);
    double c[6] = {0, 0, 0, 0, 0, 0};
    for (unsigned b = 0; b < gx; ++b) for (int i = 0; i < 6; ++i) c[i] += (double)h[b * 32 + i];
    const double n = c[4] > 0 ? c[4] : 1;
    fprintf(stderr, "[gemm dbg0] amode %d epi %d waves %d M %lld K %d N %d: stages/wg %.0f | wave 0 cyc/stage: issue %.0f mfma %.0f consume %.0f epilogue %.0f barrier %.0f\n",
            amode, epi, waves, (long long)p.M, p.Kin, p.Nout, n / gx, c[0] / n, c[1] / n, c[2] / n, c[3] / n, c[5] / n);
}

// persistent grid = one residency wave of THIS kernel: CUs x resident workgroups per CU (registers / LDS), capped by the row tiles
static unsigned persist_grid(const void *kern, int threads, unsigned parts)
{
    static std::mutex mu;
    static std::unordered_map<const void *, int> cache;
    static int ncu = 0;
    std::lock_guard<std::mutex> lk(mu);
    if (!ncu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ncu = prop.multiProcessorCount;
        else ncu = 256;
    }
    auto it = cache.find(kern);
    int per_cu;
    if (it == cache.end()) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, threads, 0) != hipSuccess || n < 1) n = 2;
        per_cu = std::min(n, 3);
        cache[kern] = per_cu;
        if (knob(KNOB_GEMM_OCC)) fprintf(stderr, "[gemm occ] kernel %p threads %d: %d resident workgroups per CU (API %d), %d CUs\n", kern, threads, per_cu, n, ncu);
    } else {
        per_cu = it->second;
    }
    return std::min<unsigned>(parts, (unsigned)(ncu * per_cu));
}
#define GEMM_GO(KERN, THREADS)                                                                                         \
    do {                                                                                                               \
        auto kfn = KERN;                                                                                               \
        grid.x = persist_grid(reinterpret_cast<const void *>(kfn), THREADS, gx);                                       \
        hipLaunchKernelGGL(kfn, grid, dim3(THREADS), 0, st, p);                                                        \
    } while (0)

static void dbg_report(const GemmArgs &p, int amode, int epi, unsigned gx)
{
    hipDeviceSynchronize();
    static unsigned long long h[512 * 4 * 8];
    hipMemcpy(h, g_dbg, sizeof(h), hipMemcpyDeviceToHost);
    double c[3] = {0, 0, 0}, pr[3][4] = {{0}};
    unsigned long long ns = 0;
    for (unsigned b = 0; b < gx; ++b) {
        for (int i = 0; i < 3; ++i) c[i] += (double)h[(b * 4) * 8 + i];
        ns += h[(b * 4) * 8 + 4];
        for (int g = 0; g < 3; ++g) for (int i = 0; i < 4; ++i) pr[g][i] += (double)h[(b * 4 + 1 + g) * 8 + i];
    }
    const double n = (double)ns;   // total slots over workgroups
    fprintf(stderr, "[gemm dbg] amode %d epi %d M %lld K %d N %d: slots/wg %.0f | consumer cyc/slot: mfma-issue %.0f epilogue %.0f barrier %.0f | producer grp0 cyc/slot: wait %.0f transform %.0f issue %.0f barrier %.0f\n",
            amode, epi, (long long)p.M, p.Kin, p.Nout, n / gx, c[0] / n, c[1] / n, c[2] / n, pr[0][0] / n, pr[0][1] / n, pr[0][2] / n, pr[0][3] / n);
}

#define GEMM_LAUNCH(a, b, c, d)                                                                                        \
    do {                                                                                                               \
        if (gemm_f32_exact()) GEMM_GO((gemm_kernel<AMODE, EPI, VEC, a, b, c, d, 1, false, 0, TL>), 256);                \
        else if (kb2) GEMM_GO((gemm_kernel<AMODE, EPI, VEC, a, b, c, d, 1, true, 0, TL, 2>), a * b * 64);               \
        else GEMM_GO((gemm_kernel<AMODE, EPI, VEC, a, b, c, d, 1, true, 0, TL>), a * b * 64);                           \
        if (dbg_on) dbg_report0(p, AMODE, EPI, gx, a * b);                                                             \
    } while (0)

template <int AMODE, int EPI, bool VEC, bool TL>
static int launch_gemm_v(const GemmArgs &p_in, hipStream_t st)
{
    GemmArgs p = p_in;
    const int dbg_on = knob(KNOB_GEMM_DBG);
    if (dbg_on) {
        if (!g_dbg) hipMalloc(&g_dbg, 512 * 4 * 8 * sizeof(unsigned long long));
        hipMemsetAsync(g_dbg, 0, 512 * 4 * 8 * sizeof(unsigned long long), st);
        p.dbg = g_dbg;
    }
    const unsigned gx = (unsigned)gemm_parts(p.M);
    p.parts = (int)gx;
    // few row tiles (group_all layers: M = B*N ~ 4096): 128-wide column tiles would leave most CUs idle, so use
    // 64- (or 32-) wide ones to get >= ~256 workgroups; the A tile is then re-read from L2 by more workgroups
    const int64_t wg128 = (int64_t)gx * cdiv(p.Nout, 128);
    // few row tiles also means ONE tile per workgroup: the k loop is a serial chain of load -> transform -> MFMA stages with nothing
    // to overlap it, so those problems take 32-wide k stages (half as many): kb2
    const int kb_env = knob(KNOB_GEMM_KB);
    const bool kb2 = !TL && (kb_env == 2 || (kb_env != 1 && gx <= 64));   // PAPC_GEMM_KB=1: never, =2: always (experiment)
    const int minwg = knob(KNOB_GEMM_MINWG);
    if (p.Nout > 64 && wg128 < minwg && (int64_t)gx * cdiv(p.Nout, 64) < 1024) {
        if ((int64_t)gx * cdiv(p.Nout, 64) >= minwg || p.Nout <= 64) {
            dim3 grid(gx, (unsigned)cdiv(p.Nout, 64));
            GEMM_LAUNCH(2, 2, 2, 1);
        } else {
            dim3 grid(gx, (unsigned)cdiv(p.Nout, 32));
            GEMM_LAUNCH(4, 1, 1, 1);
        }
    } else if (p.Nout > 64) {
        dim3 grid(gx, (unsigned)cdiv(p.Nout, 128));
        if (!gemm_f32_exact() && gemm_ws() == 3)        // 4 consumer + 12 producer waves on the same 128x128 tile
        {
            GEMM_GO((gemm_kernel<AMODE, EPI, VEC, 2, 2, 2, 2, 1, true, 3, TL>), 1024);
            if (dbg_on) dbg_report(p, AMODE, EPI, gx);
        }
        else if (!gemm_f32_exact() && gemm_waves8(AMODE, EPI))   // same 128x128 tile on 8 waves of 64x32: 4 waves per SIMD
        {
            if (kb2) GEMM_GO((gemm_kernel<AMODE, EPI, VEC, 2, 4, 2, 1, 1, true, 0, TL, 2>), 512);
            else GEMM_GO((gemm_kernel<AMODE, EPI, VEC, 2, 4, 2, 1, 1, true, 0, TL>), 512);
            if (dbg_on) dbg_report0(p, AMODE, EPI, gx, 8);
        }
        else
            GEMM_LAUNCH(2, 2, 2, 2);
    } else if (p.Nout > 32) {
        dim3 grid(gx, 1);
        GEMM_LAUNCH(2, 2, 2, 1);
    } else {
        dim3 grid(gx, 1);
        GEMM_LAUNCH(4, 1, 1, 1);
    }
    return check_launch("mlp gemm");
}

template <int AMODE, int EPI>
static int launch_gemm(const GemmArgs &p, bool vec, hipStream_t st)
{
    if constexpr (AMODE != A_GROUP && EPI != EPI_SCATTER) {
        // the big streaming shapes (M >> K, N) run barrier-free with the weights resident in LDS: mlp_stream.hip
        GemmArgs q = p;
        q.parts = gemm_parts(p.M);
        const int rc = stream_gemm_try(q, AMODE, EPI, vec, st);
        if (rc != 0) return rc < 0 ? rc : PAPC_OK;
    }
    if (p.wstat || p.rows_dev || ((AMODE == A_DY_DENSE || AMODE == A_DY_MAX) && (p.a.d.wrow || p.a.d.rows_dev))) {
        set_error("mlp gemm: a device-side row count / compacted dY source is only built for the row-streaming kernel's flavours (M=%lld Kin=%d Nout=%d)",
                  (long long)p.M, p.Kin, p.Nout);
        return PAPC_E_UNSUPPORTED;
    }
    constexpr bool TLOK = (AMODE == A_DY_DENSE || AMODE == A_DY_MAX) && (EPI == EPI_STORE || EPI == EPI_STORE_RED);
    if constexpr (TLOK) {
        if (vec && p.tl) return launch_gemm_v<AMODE, EPI, true, true>(p, st);
    }
    return vec ? launch_gemm_v<AMODE, EPI, true, false>(p, st) : launch_gemm_v<AMODE, EPI, false, false>(p, st);
}

static void fill_group(GroupSrc &g, const papc_group_src *s)
{
    g.xyz = s->xyz; g.sb = s->sb; g.sn = s->sn; g.sc = s->sc; g.new_xyz = s->new_xyz; g.feats = s->feats;
    g.idx = s->idx; g.N = s->N; g.S = s->S; g.K = s->K; g.D = s->D; g.xyz_first = s->xyz_first;
    g.divK = make_fastdiv((uint32_t)s->K);
    g.divS = make_fastdiv((uint32_t)s->S);
}

void fill_dy(DySrc &d, const papc_bwd_dy *s)
{
    d.dz = s->dz; d.gout = s->gout; d.argmax = s->argmax; d.K = s->K > 0 ? s->K : 1; d.y = s->y; d.mean = s->mean;
    d.invstd = s->invstd; d.scale = s->scale; d.shift = s->shift; d.c1 = s->c1; d.c2 = s->c2;
    d.divK = make_fastdiv((uint32_t)d.K);
    d.C = 0;  // set by the entry points (channels of the layer)
    d.wrow = s->wrow; d.seg_grp = s->seg_grp; d.rows_dev = s->rows_dev; d.psel = s->psel;
}

// validates a dY descriptor and says whether its VEC flavour is legal
int check_dy(const papc_bwd_dy *dy, int64_t M, int C, bool *vec, const char *who)
{
    PAPC_REQUIRE(dy && dy->y && dy->mean && dy->invstd && dy->scale && dy->shift && dy->c1 && dy->c2, PAPC_E_INVALID,
                 "%s: null pointer in dy", who);
    const bool cst = aligned16(dy->mean) && aligned16(dy->invstd) && aligned16(dy->scale) && aligned16(dy->shift) &&
                     aligned16(dy->c1) && aligned16(dy->c2);
    if (dy->dz_mode == PAPC_DZ_DENSE) {
        PAPC_REQUIRE(dy->dz, PAPC_E_INVALID, "%s: DENSE needs dz", who);
        *vec = cst && aligned16(dy->dz) && aligned16(dy->y) && (C % 4 == 0);
    } else if (dy->dz_mode == PAPC_DZ_MAX) {
        PAPC_REQUIRE(dy->gout && dy->argmax && dy->K >= 1 && M % dy->K == 0, PAPC_E_INVALID, "%s: MAX needs gout/argmax and K | M", who);
        *vec = cst && aligned16(dy->gout) && aligned16(dy->argmax) && aligned16(dy->y) && (C % 4 == 0);
    } else {
        set_error("%s: bad dz_mode %d", who, dy->dz_mode);
        return PAPC_E_INVALID;
    }
    return PAPC_OK;
}

int fill_asrc(ASrc &a, int a_mode, const float *x, int64_t ldx, const papc_group_src *grp, const float *sc,
              const float *sh, int Cin, const char *who)
{
    memset(&a, 0, sizeof(a));
    if (a_mode == A_PLAIN || a_mode == A_BNRELU) {
        PAPC_REQUIRE(x && ldx >= Cin, PAPC_E_INVALID, "%s: x null or ldx < Cin", who);
        PAPC_REQUIRE(a_mode == A_PLAIN || (sc && sh), PAPC_E_INVALID, "%s: BNRELU needs bn_scale/bn_shift", who);
        a.x = x; a.ldx = ldx; a.sc = sc; a.sh = sh;
        a.vec = aligned16(x) && (ldx % 4 == 0) && (Cin % 4 == 0) && (a_mode == A_PLAIN || (aligned16(sc) && aligned16(sh)));
    } else if (a_mode == A_XYZ) {
        PAPC_REQUIRE(x && ldx == 4 && sc, PAPC_E_INVALID, "%s: XYZ needs xc [M, 4] (ldx = 4) and the folded first layer wf [Cin, 4] as bn_scale", who);
        a.x = x; a.ldx = 4; a.sc = sc; a.sh = nullptr;
        a.vec = aligned16(x) && aligned16(sc) && Cin % 4 == 0;
    } else if (a_mode == A_GROUP) {
        PAPC_REQUIRE(grp && grp->xyz && grp->new_xyz, PAPC_E_INVALID, "%s: GROUP needs grp->xyz/new_xyz", who);
        PAPC_REQUIRE(grp->D == 0 || grp->feats, PAPC_E_INVALID, "%s: GROUP D=%d but feats null", who, grp->D);
        PAPC_REQUIRE(Cin == grp->D + 3, PAPC_E_INVALID, "%s: GROUP Cin=%d != D+3=%d", who, Cin, grp->D + 3);
        PAPC_REQUIRE(grp->N >= 1 && grp->S >= 1 && grp->K >= 1, PAPC_E_INVALID, "%s: GROUP bad N/S/K", who);
        fill_group(a.g, grp);
        a.vec = (grp->D == 0) || (aligned16(grp->feats) && (grp->D % 4 == 0));
    } else {
        set_error("%s: bad a_mode %d", who, a_mode);
        return PAPC_E_INVALID;
    }
    return PAPC_OK;
}

}  // namespace papc

using namespace papc;

extern "C" {

int papc_mlp_gemm_parts(int64_t M) { return gemm_parts(M); }

int papc_mlp_gemm_gmax_ok(int64_t M, int Cout, int K)
{
    // the fused epilogue needs whole groups inside one 128-row tile, full tiles only, and a 2x2-wave tile configuration;
    // with few row tiles the launcher switches to the 4x1 narrow configuration (see launch_gemm_v), which is excluded too
    if (!(K == 32 || K == 64 || K == 128) || M % 128 != 0 || Cout <= 32 || Cout % 32 != 0) return 0;  // (whole 32-column wave tiles: the K = 128 exchange has a barrier)
    if (K == 128 && !(Cout == 64 || Cout % 128 == 0)) return 0;   // every wave of a column block must take part in the exchange
    const int64_t gx = gemm_parts(M);
    if (Cout > 64 && gx * cdiv(Cout, 128) < 192 && gx * cdiv(Cout, 64) < 1024 && !(gx * cdiv(Cout, 64) >= 192)) return 0;
    return 1;
}

int papc_mlp_gemm_f32(int a_mode, const float *x, int64_t ldx, const papc_group_src *grp,
                      const float *bn_scale, const float *bn_shift, const float *w, const float *bias,
                      int64_t M, int Cin, int Cout, float *y, float *stats_partial, const papc_group_max *gmax,
                      papc_stream_t stream)
{
    return papc_mlp_gemm_rows_f32(a_mode, x, ldx, grp, bn_scale, bn_shift, w, bias, M, Cin, Cout, y, stats_partial, gmax, nullptr, stream);
}

int papc_mlp_gemm_rows_w_f32(int a_mode, const float *x, int64_t ldx, const papc_group_src *grp,
                           const float *bn_scale, const float *bn_shift, const float *w, const float *bias,
                           int64_t M, int Cin, int Cout, float *y, float *stats_partial, const papc_group_max *gmax,
                           const int32_t *rows_dev, const float *wrow, papc_stream_t stream)
{
    PAPC_REQUIRE(w && (y || gmax), PAPC_E_INVALID, "papc_mlp_gemm_f32: null w/y");
    PAPC_REQUIRE(!rows_dev || (!gmax && (a_mode == A_PLAIN || a_mode == A_BNRELU)), PAPC_E_UNSUPPORTED,
                 "papc_mlp_gemm_rows_f32: a device-side row count goes with plain / BN+ReLU rows and no fused group max");
    PAPC_REQUIRE(M >= 1 && Cin >= 1 && Cout >= 1, PAPC_E_INVALID, "papc_mlp_gemm_f32: M=%lld Cin=%d Cout=%d", (long long)M, Cin, Cout);
    PAPC_REQUIRE(M < (1ll << 31), PAPC_E_UNSUPPORTED, "papc_mlp_gemm_f32: M=%lld >= 2^31 rows", (long long)M);
    GemmArgs p;
    memset(&p, 0, sizeof(p));
    int rc = fill_asrc(p.a, a_mode, x, ldx, grp, bn_scale, bn_shift, Cin, "papc_mlp_gemm_f32");
    if (rc) return rc;
    p.w = w; p.ldw = Cin; p.bias = bias; p.M = M; p.Kin = Cin; p.Nout = Cout; p.y = y; p.ldy = Cout; p.stats = stats_partial;
    p.rows_dev = rows_dev;
    p.wstat = rows_dev ? wrow : nullptr;
    p.wmap = (a_mode == A_GROUP) ? 1 : 0;  // GROUP: internal order [feats, xyz] -> caller's columns through gk()
    const bool vec = p.a.vec && (p.wmap || (aligned16(w) && Cin % 4 == 0));
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_MLP_GEMM, st);
    PAPC_REQUIRE(!(gmax && a_mode == A_XYZ), PAPC_E_UNSUPPORTED, "papc_mlp_gemm_f32: PAPC_A_XYZ has no fused group-max flavour");
    if (gmax) {
        PAPC_REQUIRE(gmax->gmax && gmax->gmin && gmax->amax && gmax->amin, PAPC_E_INVALID, "papc_mlp_gemm_f32: null pointer in gmax");
        PAPC_REQUIRE(papc_mlp_gemm_gmax_ok(M, Cout, gmax->K), PAPC_E_UNSUPPORTED,
                     "papc_mlp_gemm_f32: fused group max needs K in {32,64,128}, M %% 128 == 0, Cout > 32 (got K=%d M=%lld Cout=%d)",
                     gmax->K, (long long)M, Cout);
        p.gm.gmax = gmax->gmax; p.gm.gmin = gmax->gmin; p.gm.amax = gmax->amax; p.gm.amin = gmax->amin; p.gm.K = gmax->K; p.gm.sgn = gmax->sign_src;
        if (!y) {   // the output exists only as its per-group extrema (papc_mlp_max_nostore_ok says where): row-streaming kernel or nothing
            GemmArgs q = p;
            q.parts = gemm_parts(p.M);
            const int rc2 = a_mode == A_BNRELU ? stream_gemm_try(q, A_BNRELU, EPI_GMAX, vec, st) : 0;
            if (rc2 == 0) { set_error("papc_mlp_gemm_f32: y = NULL (no stored output under the max) is not built for a_mode=%d M=%lld Cin=%d Cout=%d K=%d", a_mode, (long long)M, Cin, Cout, gmax->K); return PAPC_E_UNSUPPORTED; }
            return rc2 < 0 ? rc2 : PAPC_OK;
        }
        switch (a_mode) {
        case A_PLAIN: return launch_gemm<A_PLAIN, EPI_STORE_GMAX>(p, vec, st);
        case A_BNRELU: return launch_gemm<A_BNRELU, EPI_STORE_GMAX>(p, vec, st);
        default: return launch_gemm<A_GROUP, EPI_STORE_GMAX>(p, vec, st);
        }
    }
    if (a_mode == A_XYZ) {   // only the row-streaming kernel computes this operand (papc_mlp_xyz_ok says where it applies)
        GemmArgs q = p;
        q.parts = gemm_parts(p.M);
        const int rc2 = stream_gemm_try(q, A_XYZ, EPI_STORE, vec, st);
        if (rc2 == 0) { set_error("papc_mlp_gemm_f32: PAPC_A_XYZ is not built for M=%lld Cin=%d Cout=%d", (long long)M, Cin, Cout); return PAPC_E_UNSUPPORTED; }
        return rc2 < 0 ? rc2 : PAPC_OK;
    }
    switch (a_mode) {
    case A_PLAIN: return launch_gemm<A_PLAIN, EPI_STORE>(p, vec, st);
    case A_BNRELU: return launch_gemm<A_BNRELU, EPI_STORE>(p, vec, st);
    default: return launch_gemm<A_GROUP, EPI_STORE>(p, vec, st);
    }
}

int papc_mlp_gemm_rows_f32(int a_mode, const float *x, int64_t ldx, const papc_group_src *grp,
                           const float *bn_scale, const float *bn_shift, const float *w, const float *bias,
                           int64_t M, int Cin, int Cout, float *y, float *stats_partial, const papc_group_max *gmax,
                           const int32_t *rows_dev, papc_stream_t stream)
{
    return papc_mlp_gemm_rows_w_f32(a_mode, x, ldx, grp, bn_scale, bn_shift, w, bias, M, Cin, Cout, y, stats_partial, gmax, rows_dev, nullptr, stream);
}

/* 1 when a stack whose first layer is fed by coordinates only (D = 0) can run that layer through its input moments (xyz1.hip): the
 * second layer's forward and dW must both have their A_XYZ flavours (row-streaming kernels: M >= 65 536 rows, 64- or 128-channel widths) */
int papc_mlp_xyz_ok(int64_t M, int C1, int C2)
{
    if (!knob(KNOB_STREAM) || knob(KNOB_GEMM_F32) || !knob(KNOB_DW_ROWS) || knob(KNOB_DW_F32)) return 0;
    if (M % 64 != 0 || M / 32 < knob(KNOB_STREAM_MINTILES)) return 0;
    if (C1 != 64 || !(C2 == 64 || C2 == 128)) return 0;
    if (M * (int64_t)C2 * 4 >= (1ll << 32)) return 0;      // (the dW kernel addresses with 32-bit byte offsets)
    if ((C1 / 64) * (C2 / 64) > knob(KNOB_DW_ROWS_BLOCKS)) return 0;
    return 1;
}

/* 1 when the max-pooled last layer of a stack (Cin -> Cout over groups of K rows, BN+ReLU input) can run without ever storing its
 * [M, Cout] output: forward with y = NULL (extrema only), dX by papc_mlp_bwd_dx_max_f32 and dW by papc_mlp_bwd_dw_max_f32, all three on
 * their row-streaming flavours. */
int papc_mlp_max_nostore_ok(int64_t M, int Cin, int Cout, int K)
{
    if (!knob(KNOB_STREAM) || knob(KNOB_GEMM_F32) || !knob(KNOB_STREAM_ASM) || !knob(KNOB_STREAM_MAXCAT) || !knob(KNOB_DW_ROWS) || knob(KNOB_DW_F32)) return 0;
    if (!knob(KNOB_MAX_NOSTORE)) return 0;
    if (M % 128 != 0 || M / 32 < knob(KNOB_STREAM_MINTILES) || M * (int64_t)Cout * 4 >= (1ll << 32)) return 0;   // (the dW kernel addresses with 32-bit byte offsets)
    if (!(K == 32 || K == 64 || K == 128) || !papc_mlp_gemm_gmax_ok(M, Cout, K)) return 0;
    if (!(Cin == 64 && Cout == 128)) return 0;
    return 1;
}

/* 1 when a grouped stack whose first layer is the gather-add kernel (papc_lingather_*) can run COMPACTED (papc_compact_plan_f32): every
 * later layer needs the device-row-count flavours of the row-streaming forward, dX and dW kernels -- 128 -> 128 dense layers and a
 * 128 -> 256 layer under the max, on a capacity of M >= 65 536 rows (M % 128 == 0), nsample % 8 == 0. */
int papc_mlp_compact_ok(int64_t M, int K, int n_layers, const int *couts)
{
    if (!knob(KNOB_STREAM) || knob(KNOB_GEMM_F32) || !knob(KNOB_STREAM_ASM) || !knob(KNOB_DW_ROWSX) || knob(KNOB_DW_F32)) return 0;
    if (!couts || n_layers < 2 || M % 128 != 0 || M / 32 < knob(KNOB_STREAM_MINTILES) || K < 8 || K % 8 != 0) return 0;
    if (M * 256 * 4 >= (1ll << 32)) return 0;              // (32-bit byte offsets in the dW kernel)
    if (couts[0] != 128) return 0;
    for (int l = 1; l < n_layers; ++l)
        if (couts[l] != (l == n_layers - 1 ? 256 : 128)) return 0;
    return 1;
}

int papc_mlp_bwd_dx_f32(const papc_bwd_dy *dy, const float *wt, int64_t M, int Cin, int Cout, float *dx,
                        const papc_scatter_dst *scatter, const papc_bwd_red *next_red, papc_stream_t stream)
{
    PAPC_REQUIRE(wt, PAPC_E_INVALID, "papc_mlp_bwd_dx_f32: null wt");
    PAPC_REQUIRE(dx || scatter, PAPC_E_INVALID, "papc_mlp_bwd_dx_f32: need dx or scatter");
    PAPC_REQUIRE(M >= 1 && Cin >= 1 && Cout >= 1, PAPC_E_INVALID, "papc_mlp_bwd_dx_f32: bad sizes");
    PAPC_REQUIRE(M < (1ll << 31), PAPC_E_UNSUPPORTED, "papc_mlp_bwd_dx_f32: M=%lld >= 2^31 rows", (long long)M);
    bool vec = false;
    int rc = check_dy(dy, M, Cout, &vec, "papc_mlp_bwd_dx_f32");
    if (rc) return rc;
    GemmArgs p;
    memset(&p, 0, sizeof(p));
    fill_dy(p.a.d, dy);
    p.a.d.C = Cout;
    // GEMM view: rows M, reduction over Cout, outputs Cin; weights = wt [Cin][Cout]
    p.w = wt; p.ldw = Cout; p.M = M; p.Kin = Cout; p.Nout = Cin; p.y = dx; p.ldy = Cin;
    vec = vec && aligned16(wt);
    if (scatter) {
        PAPC_REQUIRE(scatter->grad_feats && scatter->D >= 1 && scatter->D + 3 == Cin, PAPC_E_INVALID,
                     "papc_mlp_bwd_dx_f32: scatter needs grad_feats and D+3 == Cin");
        PAPC_REQUIRE(scatter->N >= 1 && scatter->S >= 1 && scatter->K >= 1, PAPC_E_INVALID, "papc_mlp_bwd_dx_f32: scatter bad N/S/K");
        p.sc.gf = scatter->grad_feats; p.sc.idx = scatter->idx; p.sc.N = scatter->N; p.sc.S = scatter->S;
        p.sc.K = scatter->K; p.sc.D = scatter->D;
        p.sc.divSK = make_fastdiv((uint32_t)scatter->S * (uint32_t)scatter->K);
        p.sc.divK = make_fastdiv((uint32_t)scatter->K);
        // output column n is internal order [feats, xyz]; weight row = caller's channel order
        p.nmap = scatter->col0 ? 1 : 0;
        p.a.g.D = scatter->D; p.a.g.xyz_first = scatter->col0 ? 1 : 0;
        p.Nout = scatter->D;  // xyz columns carry no gradient: skip them entirely
    }
    if (next_red) {
        PAPC_REQUIRE(!scatter && dx, PAPC_E_INVALID, "papc_mlp_bwd_dx_f32: next_red needs a dense dx");
        PAPC_REQUIRE(next_red->y && next_red->mean && next_red->invstd && next_red->scale && next_red->shift && next_red->red_partial,
                     PAPC_E_INVALID, "papc_mlp_bwd_dx_f32: null pointer in next_red");
        p.rd.y = next_red->y; p.rd.mean = next_red->mean; p.rd.invstd = next_red->invstd; p.rd.scale = next_red->scale;
        p.rd.shift = next_red->shift; p.stats = next_red->red_partial; p.rd.masked = next_red->store_masked ? 1 : 0;
    }
    // opt-in (PAPC_GEMM_TL=1): measured on MI355X the transposed epilogue itself is 30-40 % shorter, but the kernel's load-issue
    // phase grows by more (9140-9180 vs 9220-9260 clouds/s end to end), so the column-lane epilogue stays the default
    const int tl_on = knob(KNOB_GEMM_TL);
    // (with the fused BN-backward sums only the 64-column tile has the registers for the 32 per-lane accumulators)
    p.tl = tl_on && !scatter && dx && aligned16(dx) && Cin % 4 == 0 && (!next_red || (aligned16(next_red->y) && Cin <= 64 && !next_red->store_masked));
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_BWD_DX, st);
    if (next_red) {
        return dy->dz_mode == PAPC_DZ_DENSE ? launch_gemm<A_DY_DENSE, EPI_STORE_RED>(p, vec, st) : launch_gemm<A_DY_MAX, EPI_STORE_RED>(p, vec, st);
    }
    if (scatter) {
        return dy->dz_mode == PAPC_DZ_DENSE ? launch_gemm<A_DY_DENSE, EPI_SCATTER>(p, vec, st) : launch_gemm<A_DY_MAX, EPI_SCATTER>(p, vec, st);
    }
    return dy->dz_mode == PAPC_DZ_DENSE ? launch_gemm<A_DY_DENSE, EPI_STORE>(p, vec, st) : launch_gemm<A_DY_MAX, EPI_STORE>(p, vec, st);
}

int papc_mlp_bwd_dx_xyz_ok(int64_t M, int Cin, int Cout)
{
    // the row-streaming kernel's (Kin = Cout in {64, 128}) x (Nout = Cin = 64) dX flavours, whole 32-row tiles, enough of them
    return knob(KNOB_STREAM) && knob(KNOB_STREAM_ASM) && !knob(KNOB_GEMM_F32) && Cin == 64 && (Cout == 64 || Cout == 128) && M % 32 == 0 &&
           M / 32 >= knob(KNOB_STREAM_MINTILES);
}

int papc_mlp_bwd_dx_xyz_f32(const papc_bwd_dy *dy, const float *wt, int64_t M, int Cin, int Cout, const float *xc, const float *wf,
                            float *partial, papc_stream_t stream)
{
    PAPC_REQUIRE(wt && xc && wf && partial, PAPC_E_INVALID, "papc_mlp_bwd_dx_xyz_f32: null pointer");
    PAPC_REQUIRE(M >= 1 && M < (1ll << 31) && Cin >= 1 && Cout >= 1, PAPC_E_INVALID, "papc_mlp_bwd_dx_xyz_f32: bad sizes");
    PAPC_REQUIRE(dy && dy->dz_mode == PAPC_DZ_DENSE, PAPC_E_UNSUPPORTED, "papc_mlp_bwd_dx_xyz_f32: a dense dY source only (the layer above the first is never the pooled one)");
    PAPC_REQUIRE(papc_mlp_bwd_dx_xyz_ok(M, Cin, Cout), PAPC_E_UNSUPPORTED, "papc_mlp_bwd_dx_xyz_f32: not built for M=%lld Cin=%d Cout=%d (papc_mlp_bwd_dx_xyz_ok)", (long long)M, Cin, Cout);
    PAPC_REQUIRE(aligned16(xc) && aligned16(wf) && aligned16(wt), PAPC_E_INVALID, "papc_mlp_bwd_dx_xyz_f32: 16-byte alignment");
    bool vec = false;
    int rc = check_dy(dy, M, Cout, &vec, "papc_mlp_bwd_dx_xyz_f32");
    if (rc) return rc;
    GemmArgs p;
    memset(&p, 0, sizeof(p));
    fill_dy(p.a.d, dy);
    p.a.d.C = Cout;
    p.w = wt; p.ldw = Cout; p.M = M; p.Kin = Cout; p.Nout = Cin; p.y = nullptr; p.ldy = Cin;
    p.rd.y = xc; p.rd.scale = wf; p.stats = partial; p.parts = gemm_parts(M);
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_BWD_DX, st);
    rc = stream_gemm_try(p, A_DY_DENSE, EPI_XYZ_RED, vec, st);
    if (rc < 0) return rc;
    if (rc == 0) { set_error("papc_mlp_bwd_dx_xyz_f32: the row-streaming kernel declined M=%lld Cin=%d Cout=%d", (long long)M, Cin, Cout); return PAPC_E_UNSUPPORTED; }
    return PAPC_OK;
}

int papc_mlp_bwd_dx_max_f32(const float *psel, const int32_t *argmax, int K, const float *x, int64_t ldx, const float *bn_scale,
                            const float *bn_shift, const float *wcat, const float *hbias, int64_t M, int Cin, int Cout, float *dx,
                            const papc_bwd_red *next_red, papc_stream_t stream)
{
    PAPC_REQUIRE(psel && argmax && x && bn_scale && bn_shift && wcat && hbias && dx, PAPC_E_INVALID, "papc_mlp_bwd_dx_max_f32: null pointer");
    PAPC_REQUIRE(M >= 1 && Cin >= 1 && Cout >= 1 && K >= 1 && M % K == 0, PAPC_E_INVALID, "papc_mlp_bwd_dx_max_f32: bad sizes");
    PAPC_REQUIRE(M < (1ll << 31), PAPC_E_UNSUPPORTED, "papc_mlp_bwd_dx_max_f32: M=%lld >= 2^31 rows", (long long)M);
    PAPC_REQUIRE(Cout % 16 == 0 && Cin % 4 == 0 && ldx % 4 == 0 && aligned16(psel) && aligned16(argmax) && aligned16(x) && aligned16(bn_scale) &&
                     aligned16(bn_shift) && aligned16(wcat) && aligned16(dx),
                 PAPC_E_UNSUPPORTED, "papc_mlp_bwd_dx_max_f32: needs Cout %% 16 == 0, Cin %% 4 == 0 and 16-byte aligned operands");
    GemmArgs p;
    memset(&p, 0, sizeof(p));
    p.a.x = x; p.a.ldx = ldx; p.a.sc = bn_scale; p.a.sh = bn_shift; p.a.vec = 1;
    p.a.d.gout = psel; p.a.d.argmax = argmax; p.a.d.K = K; p.a.d.divK = make_fastdiv((uint32_t)K); p.a.d.C = Cout;
    // GEMM view: rows M, reduction over the concatenated Cout + Cin channels, outputs Cin; weights wcat [Cin][Cout + Cin]
    p.w = wcat; p.ldw = Cout + Cin; p.bias = hbias; p.M = M; p.Kin = Cout + Cin; p.Nout = Cin; p.y = dx; p.ldy = Cin;
    if (next_red) {
        PAPC_REQUIRE(next_red->y && next_red->mean && next_red->invstd && next_red->scale && next_red->shift && next_red->red_partial,
                     PAPC_E_INVALID, "papc_mlp_bwd_dx_max_f32: null pointer in next_red");
        p.rd.y = next_red->y; p.rd.mean = next_red->mean; p.rd.invstd = next_red->invstd; p.rd.scale = next_red->scale;
        p.rd.shift = next_red->shift; p.stats = next_red->red_partial; p.rd.masked = next_red->store_masked ? 1 : 0;
    }
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_BWD_DX, st);
    if (next_red && knob(KNOB_STREAM_MAXCAT)) {      // the row-streaming flavour where it exists (Cout = 2 Cin = 128 / 256, big M)
        GemmArgs q = p;
        q.parts = gemm_parts(p.M);
        const int rc = stream_gemm_try(q, A_MAXCAT, EPI_STORE_RED, true, st);
        if (rc != 0) return rc < 0 ? rc : PAPC_OK;
    }
    if (next_red) return launch_gemm_v<A_MAXCAT, EPI_STORE_RED, true, false>(p, st);
    return launch_gemm_v<A_MAXCAT, EPI_STORE, true, false>(p, st);
}

}  // extern "C"
