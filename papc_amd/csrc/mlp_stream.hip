// mlp_stream.hip -- barrier-free "row streaming" form of the pointwise-MLP GEMMs (gfx950).
//
//   forward  : y[M,Cout]  = A(x)[M,Cin] . w[Cout,Cin]^T + bias      (+ per-channel sum / sum-of-squares partials, + group max)
//   backward : dX[M,Cin]  = dY[M,Cout] . w[Cout,Cin]                 (dY recomputed in the load, + BN-backward sums of the layer below)
//
// Same contract as gemm_kernel (mlp_gemm.hip) -- nn.Conv2D(cin,cout,1) + the BatchNorm2D statistics of
// /root/reference/PAPC/models/layers/pointnet2_basic_layers.py:189,215-219 on channel-contiguous rows -- for the shapes that
// dominate a training step: M in the hundreds of thousands, K and N between 32 and 256.  What the tiled kernel is bound by on
// those shapes is not bytes but its per-stage sequence (issue -> wait -> transform -> LDS write -> barrier -> MFMA):
// two lock-stepped workgroups per CU cannot fill each other's gaps.  This kernel removes the sequence instead of tuning it:
//
//   * The weights are split into their three bf16 planes ONCE per (persistent) workgroup and stay in LDS for all of its rows
//     (N tile x K x 6 bytes: 25-100 KB).  After that prologue there is no barrier and no LDS write in the main loop.
//   * A wave owns whole 32-row tiles.  The fp32 row-major operand already has the MFMA A-fragment shape -- lane (row = lane & 31,
//     half = lane >> 5) needs the 8 consecutive channels 16 kb + 8 half .. + 7 of its row, i.e. two 16-byte loads -- so the operand
//     goes global -> VGPR directly (each 128-byte line is fetched once; both halves of a row are touched by the same instruction
//     pair), is transformed (BN+ReLU of the previous layer / the BN+ReLU+max backward) and split in registers, and meets the
//     weight fragments read from LDS in v_mfma_f32_32x32x16_bf16 (six products per block, fp32 accumulate: the exact 3-way split
//     of mlp_loaders.h).  No A tile in LDS, no transposes.
//   * Waves are independent streams: 8 per CU (2 per SIMD), each with its own prefetch -- the operand loads of k chunk c + 2 are
//     issued when chunk c has been consumed, into the registers it frees, and waited for with a COUNTED s_waitcnt (only the
//     younger chunk's loads may remain outstanding; loads return in order, so `vmcnt(#younger loads)` is exact).  The loads are
//     inline asm so hipcc's waitcnt pass, which drains such rings to vmcnt(0) at every loop back-edge, does not see them;
//     the destination registers are tied ("+v") through the load and named again by the wait (cdna guide 5.7, form ii).
//     One wave's epilogue / load wait overlaps the other seven's MFMAs by ordinary wave scheduling.
//   * Epilogue from the accumulator layout (lane = column): every store instruction writes two full 128-byte row segments; the
//     BN statistics are two running sums per lane; the neighbourhood max (EPI_STORE_GMAX) is a running max / min / first-offset
//     per lane over the K/32 consecutive tiles of a group, which one wave processes back to back (no exchange between waves).
#include "mlp_gemm.h"
#include <type_traits>
#include <utility>

namespace papc {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int I, int N, class F>
__device__ __forceinline__ void sfor(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        sfor<I + 1, N>(f);
    }
}

// one 16-byte operand load: address = wave-uniform base (SGPR pair) + per-lane byte offset + immediate
template <bool ASM, int OFF, bool FIRST>
__device__ __forceinline__ void gload4(f32x4 &d, unsigned voff, const char *sbase)
{
    if constexpr (ASM) {
        // FIRST: the base may have been produced by v_readfirstlane (VALU write of an SGPR -> VMEM read: 5 wait states)
        if constexpr (FIRST) asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:%3" : "+v"(d) : "v"(voff), "s"(sbase), "n"(OFF));
        else asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "+v"(d) : "v"(voff), "s"(sbase), "n"(OFF));
    } else {
        d = *reinterpret_cast<const f32x4 *>(sbase + voff + OFF);
    }
}
template <bool ASM, int N>
__device__ __forceinline__ void wait_vm()
{
    if constexpr (ASM) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N));
}
template <bool ASM>
__device__ __forceinline__ void touch(f32x4 &d)
{
    if constexpr (ASM) asm volatile("" : "+v"(d));
}

struct StreamGeo {
    int n_units;     // units of U consecutive 32-row tiles
    int ushift;      // U = 1 << ushift
    int kgshift;     // DY_MAX: rows per group = 1 << kgshift (>= 32)
    // compacted stack (compact.hip): the physical row count lives in device memory (a multiple of 128; n_units is its upper bound), a DY
    // operand weighs its BatchNorm-backward term with the row's multiplicity, and groups are ragged (multiples of 8 rows)
    const int *rows_dev;
    const float *wrow;       // [rows] (CP flavours)
    const int *seg_grp;      // [rows / 8] group of every 8-row segment (CP, DY_MAX)
    const float *wstat;      // forward of a compacted stack: the rows' multiplicity weights -> statistics of the padded tensor (see the kernel's tail)
};

// AMODE: A_PLAIN, A_BNRELU, A_DY_DENSE, A_DY_MAX, A_MAXCAT, A_XYZ.  EPI: EPI_STORE, EPI_STORE_GMAX / EPI_GMAX (forward), EPI_STORE_RED (dX).
// KB16 = K / 16 k blocks, CK blocks per prefetch chunk, WN 32-column tiles per wave (N tile = 32 WN columns per workgroup).
// ---- packed-f32 flavour of the operand transform + split (PAPC_STREAM_PK, default on).  A lane's 8 k values are 4 register pairs
// (consecutive k, as the dwordx4 loads deliver them), so v_pk_fma_f32 / v_pk_add_f32 need no moves to form their 64-bit operands:
// the BN fma, the BN-backward fmas and the two subtractions of the split each cost one instruction per PAIR.
#ifndef PAPC_STREAM_PK
#define PAPC_STREAM_PK 1
#endif
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
// three bf16 planes of one pair: x = p0 + p1 + p2 exactly (see split3 in mlp_loaders.h)
__device__ __forceinline__ void split3_pair(f32x2 x, unsigned &p0, unsigned &p1, unsigned &p2)
{
    p0 = pack_bf16x2(x.x, x.y);
    x = x - f32x2{bf16_lo(p0), bf16_hi(p0)};
    p1 = pack_bf16x2(x.x, x.y);
    x = x - f32x2{bf16_lo(p1), bf16_hi(p1)};
    p2 = pack_bf16x2(x.x, x.y);
}

// CP (dX of a compacted stack: DY operands, STORE_RED epilogue): per-row weight w in the BatchNorm-backward term, dy = sc p - w (A + B' (y - mean));
// DY_MAX: the row's group comes from seg_grp (ragged groups), argmax holds absolute rows.  The weight of tile j + 1 and the group of tile
// j + 2 are fetched between the full drain that precedes tile j's epilogue (LATE1 flavours) and the epilogue itself: no new wait in the loop.
// KV (ragged k: the MSG segmenter's 196-channel layer, pointnet2.py:63): the operand rows hold KV = 16 (KB16 - 1) + 4 channels.  The LDS image of the
// weights and the constants are zero beyond KV; the last k block is a PARTIAL block of one 16-byte load per streamed array (channels KV - 4 ..
// KV - 1, read by BOTH half-waves -- the upper half's copy meets zero weights) that rides on the last chunk of full blocks.
// NR (ragged n): Nout is not a multiple of the column block; weight rows beyond it are zero in LDS, their stores and statistics are masked.
// CBW (with NR): a column block owns CBW < 32 WN columns and keeps CBW + 1 weight rows in LDS -- the last one zeros, which the lanes of the
// boundary tile beyond column CBW read -- so that 196 columns are TWO blocks of 98 (four tiles each, two passes over the operand) where
// whole tiles would need three passes (K = 256: 96 whole-tile rows are the most that fit beside the constants).
// PS (CP + A_DY_MAX): the streamed [G, C] operand is psel = scale * p (papc_bwd_dy::psel), not gout: no ReLU test, no scale per row.
// NWT: waves per workgroup.  8 (two per SIMD, up to 256 registers each) everywhere but the flavours that fit 168 registers: those run TWELVE
// (three per SIMD: these kernels wait on their loads, not on an execution unit), with the wave-major tile numbering of the device-row-count
// flavours so that the ragged last round is spread over all CUs.
// GS (group-max epilogues): ONE extremum per channel -- the max of a channel whose BatchNorm weight is >= 0, the min of the others (GmaxDst::sgn,
// papc_group_max::sign_src) -- followed as the maximum of the value with its sign bit flipped where the minimum is wanted; written to both pairs
// of arrays, so papc_bn_select_max_f32 reads the same value whichever it picks.  Half the compare / select instructions of the epilogue.
template <int AMODE, int EPI, int KB16, int CK, int WN, bool ASM, bool CP = false, int KV = KB16 * 16, bool NR = false, int CBW = WN * 32, bool PS = false, int NWT = 8,
          bool GS = false>
__global__ __launch_bounds__(NWT * 64, NWT / 4) void stream_kernel(GemmArgs p, StreamGeo geo)
{
    static_assert(!GS || EPI == EPI_STORE_GMAX || EPI == EPI_GMAX, "GS: group-max epilogues only");
    static_assert(NWT == 8 || NWT == 12, "two or three waves per SIMD");
    static_assert(!PS || (CP && AMODE == A_DY_MAX), "PS: the compacted max-layer dX only (the one flavour where the result is bit-identical)");
    static_assert(!CP || ((AMODE == A_DY_DENSE || AMODE == A_DY_MAX) && EPI == EPI_STORE_RED && ASM && PAPC_STREAM_PK), "CP: dX flavours of the asm ring only");
    constexpr int NW = NWT;
    constexpr bool KR = (KV != KB16 * 16);
    // K = extent of the LDS images (weight planes, constants).  Ragged k: KV + 4 -- the partial block's fragment reads of the UPPER half-wave
    // (k = KV + 4 .. KV + 11) run into the next plane / the row's zeroed pad, and that half's operand is forced to zero in the transform
    constexpr int K = KR ? KV + 4 : KB16 * 16;
    constexpr int NT = WN * 32;
    constexpr int KBF = KR ? KB16 - 1 : KB16;      // full k blocks
    static_assert(!KR || (KV == KB16 * 16 - 12 && !CP && PAPC_STREAM_PK && (AMODE == A_BNRELU || AMODE == A_DY_DENSE || AMODE == A_PLAIN)), "ragged k: four channels in the last block");
    static_assert(!NR || (EPI == EPI_STORE || EPI == EPI_STORE_RED), "ragged n: storing epilogues without the group max");
    constexpr int NCH = KBF / CK;
    static_assert(KBF % CK == 0 && NCH >= 2 && NCH % 2 == 0, "an even number of chunks per tile");
    constexpr bool DY = (AMODE == A_DY_DENSE || AMODE == A_DY_MAX);
    constexpr bool GM = (EPI == EPI_STORE_GMAX || EPI == EPI_GMAX);   // per-group max / min of the raw output in the epilogue
    constexpr bool XYZ = (AMODE == A_XYZ);   // operand computed from the row's centred coordinates: no streamed operand, no asm ring
    // A_MAXCAT (dX of a max-pooled last layer without its output y, papc_mlp_bwd_dx_max_f32): k blocks [0, KS) are the sparse max-backward
    // operand (psel + argmax of the row's group: 4 loads), blocks [KS, KB16) the layer's BN+ReLU input (2 loads).  Cout = 2 Cin: KS = 2/3 KB16.
    constexpr bool MC = (AMODE == A_MAXCAT);
    constexpr int KS = MC ? KB16 * 2 / 3 : 0;
    static_assert(!MC || (KB16 % 3 == 0 && CK == 1), "MAXCAT: Cout = 2 Cin, one k block per chunk");
    constexpr int NLD = (AMODE == A_DY_MAX) ? 6 : ((AMODE == A_DY_DENSE || MC) ? 4 : 2);   // 16-byte loads per lane and k block (MAXCAT: at most)
    constexpr int NLDP = KR ? NLD / 2 : 0;                                            // ... of the partial block (one per streamed array)
    constexpr int CLB = CK * NLD;                                                     // ... per chunk of full blocks = registers of a ring buffer
    constexpr int CL = CLB + NLDP;                                                    // ... per chunk at most
    static_assert(CL <= 60, "vmcnt is a 6-bit field");
    constexpr int ROWB = 6 * K + (KR ? 32 : 16);   // LDS bytes of one weight row: [plane 0 | plane 1 | plane 2] bf16 + pad (odd number of 16-B slots)
    static_assert((ROWB / 16) % 2 == 1 && ROWB % 16 == 0, "conflict-free fragment reads");
    constexpr int NCST = (AMODE == A_BNRELU || MC) ? 2 : (DY ? 5 : (XYZ ? 4 : 0));
    constexpr bool CB = (CBW != NT);
    constexpr int TB = CBW / 32;                  // CB: the boundary tile
    static_assert(!CB || (NR && TB == WN - 1 && CBW % 32 != 0), "a narrow column block ends inside its last tile");
    constexpr int NROW = CB ? CBW + 1 : NT;       // weight rows in LDS
    constexpr int W_BYTES = NROW * ROWB;
    constexpr int CST_BYTES = NCST * K * 4 + (KR ? 32 : 0);      // (ragged k: the upper half-wave reads 8 constants past the last array; never used)
    constexpr bool XR = (EPI == EPI_XYZ_RED);   // dX folded straight into the four sums of a coordinates-only first layer's backward: nothing stored
    constexpr int NSUM = XR ? 4 : 2;
    constexpr int RED_BYTES = NSUM * NW * NT * 4;
    constexpr int SMEM = (W_BYTES + CST_BYTES) > RED_BYTES ? (W_BYTES + CST_BYTES) : RED_BYTES;
    __shared__ __attribute__((aligned(16))) char smem[SMEM];
    char *cstb = smem + W_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, l31 = lane & 31;
    const int n0 = blockIdx.y * CBW;
    const int ldx = (int)(DY ? p.a.d.C : p.a.ldx);     // row stride (floats) of the streamed operand(s)
    // loads of chunk ci (compile time): uniform except MAXCAT, whose sparse blocks take 4 and dense blocks 2
    auto nld_of = [](int ci) constexpr -> int { return MC ? ((ci % NCH) < KS ? 4 : 2) : (CK * NLD + ((ci % NCH) == NCH - 1 ? NLDP : 0)); };

    // ---- prologue: weights -> three bf16 planes in LDS (once per workgroup); folded per-channel constants
    {
        const float *wb = p.w + (int64_t)n0 * p.ldw;
        // all of the thread's pieces are loaded before the first is used: one exposed L2 round trip, not NI of them
        constexpr int NP = NROW * (K / 4);                // float4 pieces of the block's weights
        constexpr int NI = (NP + NW * 64 - 1) / (NW * 64);
        constexpr bool PG = (NP % (NW * 64) != 0);        // (guarded last round)
        static_assert(KR || CB || !PG || NW != 8, "whole float4 pieces per thread");
        float4 wv[NI];
#pragma unroll
        for (int q = 0; q < NI; ++q) {
            const int i = tid + q * NW * 64;
            const int n = i / (K / 4), k4 = (i - n * (K / 4)) * 4;
            bool ok = true;
            if constexpr (KR) ok = ok && k4 < KV;
            if constexpr (PG) ok = ok && i < NP;
            if constexpr (NR) ok = ok && n0 + n < p.Nout;
            if constexpr (CB) ok = ok && n < CBW;
            wv[q] = ok ? ld4(wb + (int64_t)n * p.ldw + k4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int q = 0; q < NI; ++q) {
            const int i = tid + q * NW * 64;
            const int n = i / (K / 4), k4 = (i - n * (K / 4)) * 4;
            uint2 q0, q1, q2;
            split3(wv[q], q0, q1, q2);
            char *d = smem + n * ROWB + k4 * 2;
            if (!PG || i < NP) {
                *reinterpret_cast<uint2 *>(d) = q0;
                *reinterpret_cast<uint2 *>(d + 2 * K) = q1;
                *reinterpret_cast<uint2 *>(d + 4 * K) = q2;
            }
        }
        if constexpr (KR) {      // the rows' pads are READ (plane 2, upper half-wave of the partial block): zeros
            for (int n = tid; n < NROW; n += NW * 64) {
                *reinterpret_cast<uint4 *>(smem + n * ROWB + 6 * K) = make_uint4(0u, 0u, 0u, 0u);
                *reinterpret_cast<uint4 *>(smem + n * ROWB + 6 * K + 16) = make_uint4(0u, 0u, 0u, 0u);
            }
        }
        float *cf = reinterpret_cast<float *>(cstb);
        for (int k = tid; k < K; k += NW * 64) {
            if (KR && k >= KV) {     // (zero constants: the transform of whatever a lane holds there is 0)
#pragma unroll
                for (int q = 0; q < NCST; ++q) cf[q * K + k] = 0.f;
            } else if (AMODE == A_BNRELU) {
                cf[k] = p.a.sc[k]; cf[K + k] = p.a.sh[k];
            } else if (MC) {         // BN+ReLU constants of the dense region, stored at the concatenated k
                const bool dn = k >= KS * 16;
                cf[k] = dn ? p.a.sc[k - KS * 16] : 0.f; cf[K + k] = dn ? p.a.sh[k - KS * 16] : 0.f;
            } else if (XYZ) {        // folded first layer wf [K][4] -> four per-channel arrays
                const float4 q = ld4(p.a.sc + 4 * k);
                cf[k] = q.x; cf[K + k] = q.y; cf[2 * K + k] = q.z; cf[3 * K + k] = q.w;
            } else if (DY) {
                // dy = sc (p - c1 - xhat c2), p = dz [sc y + sh > 0], xhat = (y - mean) invstd   ==   sc p - (A + Bp (y - mean))
                const float sc = p.a.d.scale[k];
                cf[k] = sc; cf[K + k] = p.a.d.shift[k]; cf[2 * K + k] = p.a.d.mean[k];
                cf[3 * K + k] = sc * p.a.d.c1[k];
                cf[4 * K + k] = sc * p.a.d.c2[k] * p.a.d.invstd[k];
            }
        }
    }
    __syncthreads();

    // ---- this wave's tiles: units gw, gw + TW, ...; U consecutive tiles per unit
    const int TW = (int)gridDim.x * NW;
    // (device-side row count: the tile count is whatever the batch gives, so the last round is ragged -- wave-major numbering hands its few
    // tiles to one wave each of as many different workgroups, where a lone wave runs at twice the issue rate, instead of to all eight waves
    // of the first few workgroups while the other CUs idle)
    const int gw = (geo.rows_dev || NW != 8) ? wave * (int)gridDim.x + (int)blockIdx.x : (int)blockIdx.x * NW + wave;
    const int U = 1 << geo.ushift;
    const int n_units = geo.rows_dev ? min(geo.n_units, (__builtin_amdgcn_readfirstlane(*geo.rows_dev) >> 5) >> geo.ushift) : geo.n_units;
    const int my_units = n_units > gw ? (n_units - gw + TW - 1) / TW : 0;
    const int my_tiles = my_units << geo.ushift;
    auto tile_row0 = [&](int j) -> int {   // first row of this wave's j-th tile (clamped to its last one: the prefetch runs ahead)
        const int jj = j < my_tiles ? j : my_tiles - 1;
        const int u = gw + (jj >> geo.ushift) * TW;
        return ((u << geo.ushift) + (jj & (U - 1))) * 32;
    };
    struct SB { const char *p0, *p1, *p2; unsigned vg; };
    const unsigned voff_grp0 = (unsigned)(8 * hi * 4);
    auto bases = [&](int row0) -> SB {
        SB s;
        s.p1 = nullptr; s.p2 = nullptr; s.vg = voff_grp0;
        if (DY) {
            s.p0 = reinterpret_cast<const char *>(p.a.d.y + (int64_t)row0 * ldx);
            if (AMODE == A_DY_DENSE) s.p1 = reinterpret_cast<const char *>(p.a.d.dz + (int64_t)row0 * ldx);
            else if (CP) {     // ragged groups: the per-lane group offset is SB::vg (set by the caller from seg_grp)
                s.p1 = reinterpret_cast<const char *>(p.a.d.gout);
                s.p2 = reinterpret_cast<const char *>(p.a.d.argmax);
            } else {
                const int64_t g = row0 >> geo.kgshift;
                s.p1 = reinterpret_cast<const char *>(p.a.d.gout + g * ldx);
                s.p2 = reinterpret_cast<const char *>(p.a.d.argmax + g * ldx);
            }
        } else {
            s.p0 = reinterpret_cast<const char *>(p.a.x + (int64_t)row0 * ldx);
            if (MC) {
                const int64_t g = row0 >> geo.kgshift;
                s.p1 = reinterpret_cast<const char *>(p.a.d.gout + g * p.a.d.C);
                s.p2 = reinterpret_cast<const char *>(p.a.d.argmax + g * p.a.d.C);
            }
        }
        return s;
    };
    const unsigned voff_row = (unsigned)((l31 * ldx + 8 * hi) * 4);   // this lane's row and k half inside a tile
    const unsigned voff_grp = voff_grp0;                               // ... inside a per-group row (DY_MAX: gout / argmax)
    const unsigned voff_lo = (unsigned)(l31 * ldx * 4);                // KR: the partial k block, channels KV - 4 .. KV - 1 of the lane's row for both halves

    f32x4 buf[2][CLB];
    f32x4 pbuf[NLDP > 0 ? NLDP : 1];       // KR: the partial block's registers (loaded with the last chunk only)
#pragma unroll
    for (int i = 0; i < (NLDP > 0 ? NLDP : 1); ++i) {
        if constexpr (KR) asm volatile("" : "=v"(pbuf[i]));
        else pbuf[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int i = 0; i < CLB; ++i) {
            // (CP flavours sit at the register limit: their ring registers are DEFINED here, behind the prologue's barrier, by an empty volatile asm
            // instead of a zero the compiler hoists above the weight split and then spills across it)
            if constexpr (CP || KR) asm volatile("" : "=v"(buf[b][i]));
            else buf[b][i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }

    // issue the loads of chunk `ci` (compile-time) of the tile whose bases are `s` into buffer `bi`
    auto issue = [&](auto bi_, auto ci_, const SB &s) {
        constexpr int bi = decltype(bi_)::value, ci = decltype(ci_)::value;
        sfor<0, CK>([&](auto blk_) {
            constexpr int blk = decltype(blk_)::value;
            constexpr int off = (ci * CK + blk) * 64;
            if constexpr (MC) {
                if constexpr (ci < KS) {       // sparse block: 8 psel values and their argmax offsets of this row's group
                    gload4<ASM, off, true>(buf[bi][0], voff_grp, s.p1);
                    gload4<ASM, off + 16, false>(buf[bi][1], voff_grp, s.p1);
                    gload4<ASM, off, true>(buf[bi][2], voff_grp, s.p2);
                    gload4<ASM, off + 16, false>(buf[bi][3], voff_grp, s.p2);
                } else {                       // dense block: the layer's input row
                    gload4<ASM, (ci - KS) * 64, true>(buf[bi][0], voff_row, s.p0);
                    gload4<ASM, (ci - KS) * 64 + 16, false>(buf[bi][1], voff_row, s.p0);
                }
                return;
            }
            gload4<ASM, off, blk == 0>(buf[bi][blk * NLD + 0], voff_row, s.p0);
            gload4<ASM, off + 16, false>(buf[bi][blk * NLD + 1], voff_row, s.p0);
            if constexpr (AMODE == A_DY_DENSE) {
                gload4<ASM, off, blk == 0>(buf[bi][blk * NLD + 2], voff_row, s.p1);
                gload4<ASM, off + 16, false>(buf[bi][blk * NLD + 3], voff_row, s.p1);
            }
            if constexpr (AMODE == A_DY_MAX) {
                const unsigned vg = CP ? s.vg : voff_grp;
                gload4<ASM, off, blk == 0>(buf[bi][blk * NLD + 2], vg, s.p1);
                gload4<ASM, off + 16, false>(buf[bi][blk * NLD + 3], vg, s.p1);
                gload4<ASM, off, blk == 0>(buf[bi][blk * NLD + 4], vg, s.p2);
                gload4<ASM, off + 16, false>(buf[bi][blk * NLD + 5], vg, s.p2);
            }
        });
        if constexpr (KR && ci == NCH - 1) {       // the partial block rides on the last chunk
            gload4<ASM, KBF * 64, false>(pbuf[0], voff_lo, s.p0);
            if constexpr (AMODE == A_DY_DENSE) gload4<ASM, KBF * 64, false>(pbuf[1], voff_lo, s.p1);
        }
    };
    auto touch_buf = [&](auto bi_) {
        constexpr int bi = decltype(bi_)::value;
        sfor<0, CLB>([&](auto i_) { touch<ASM>(buf[bi][decltype(i_)::value]); });
    };
    auto touch_p = [&]() {
        if constexpr (KR) sfor<0, NLDP>([&](auto i_) { touch<ASM>(pbuf[decltype(i_)::value]); });
    };

    floatx16 acc[WN];
#pragma unroll
    for (int wn = 0; wn < WN; ++wn)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[wn][r] = 0.f;

    const char *wl = smem + l31 * ROWB + hi * 16;        // this lane's weight-fragment row (tile 0, plane 0, k block 0)
    const char *wlb = smem + min(TB * 32 + l31, CBW) * ROWB + hi * 16;   // CB: ... in the boundary tile (the zero row beyond column CBW)
    const char *cl = cstb + hi * 32;                     // this lane's constants (k block 0)
    int kin = 0;                                         // DY_MAX: this lane's row offset inside its group (CP: its absolute row)
    float wcur = -1.f;                                   // CP: MINUS the multiplicity weight of this lane's row in the current tile

    // k block kb of a tile whose operand comes from the lane's centred coordinates (A_XYZ): relu(wf . x + t) for the lane's 8 channels
    auto compute_xyz = [&](auto kb_, const float4 &xv) {
        constexpr int kb = decltype(kb_)::value;
        f32x4 c[4][2];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            c[q][0] = *reinterpret_cast<const f32x4 *>(cl + q * K * 4 + kb * 64);
            c[q][1] = *reinterpret_cast<const f32x4 *>(cl + q * K * 4 + kb * 64 + 16);
        }
        unsigned q0[4], q1[4], q2[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int h = j >> 1, i = 2 * (j & 1);
// (plain fmas: the packed form needs the row's coordinate broadcast into both halves -- hipcc encodes that as v_pk_fma_f32 with
            // op_sel / op_sel_hi on the (x, y) register pair, and that instruction returned wrong values for a varying ~15 % of the rows
            // on gfx950, ROCm 7.2: tools/probe/dbg_xyz2.py; the natural-pair pk_fma of the other operand flavours is unaffected)
            f32x2 t;
            t.x = fmaf(c[2][h][i], xv.z, fmaf(c[1][h][i], xv.y, fmaf(c[0][h][i], xv.x, c[3][h][i])));
            t.y = fmaf(c[2][h][i + 1], xv.z, fmaf(c[1][h][i + 1], xv.y, fmaf(c[0][h][i + 1], xv.x, c[3][h][i + 1])));
            split3_pair(f32x2{fmaxf(t.x, 0.f), fmaxf(t.y, 0.f)}, q0[j], q1[j], q2[j]);
        }
        bf16x8 af[3];
        af[0] = __builtin_bit_cast(bf16x8, make_uint4(q0[0], q0[1], q0[2], q0[3]));
        af[1] = __builtin_bit_cast(bf16x8, make_uint4(q1[0], q1[1], q1[2], q1[3]));
        af[2] = __builtin_bit_cast(bf16x8, make_uint4(q2[0], q2[1], q2[2], q2[3]));
        bf16x8 bq[WN][3];
#pragma unroll
        for (int wn = 0; wn < WN; ++wn)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                bq[wn][pl] = *reinterpret_cast<const bf16x8 *>(wl + wn * 32 * ROWB + pl * 2 * K + kb * 32);
        constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int wn = 0; wn < WN; ++wn)
                acc[wn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PA[t]], bq[wn][PB[t]], acc[wn], 0, 0, 0);
    };

    // MFMAs of chunk `ci` from buffer `bi`
    auto compute_block = [&](auto kb_, const f32x4 *r) {
        {
            constexpr int kb = decltype(kb_)::value;
            constexpr int NJ = (KR && kb == KBF) ? 2 : 4;      // value pairs of the lane that exist (partial block: channels KV - 4 .. KV - 1)
            bf16x8 af[3];
#if PAPC_STREAM_PK
            {
                f32x2 v2[4];   // pairs (k, k+1): r[0] = k 0..3, r[1] = k 4..7 of this lane's half block
                if constexpr (AMODE == A_PLAIN) {
#pragma unroll
                    for (int j = 0; j < NJ; ++j) v2[j] = f32x2{r[j >> 1][2 * (j & 1)], r[j >> 1][2 * (j & 1) + 1]};
                } else if constexpr (MC && kb < KS) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int h = j >> 1, i = 2 * (j & 1);
                        v2[j] = f32x2{(__float_as_int(r[2 + h][i]) == kin) ? r[h][i] : 0.f, (__float_as_int(r[2 + h][i + 1]) == kin) ? r[h][i + 1] : 0.f};
                    }
                } else if constexpr (AMODE == A_BNRELU || MC) {
                    const f32x4 s0 = *reinterpret_cast<const f32x4 *>(cl + kb * 64), s1 = *reinterpret_cast<const f32x4 *>(cl + kb * 64 + 16);
                    const f32x4 h0 = *reinterpret_cast<const f32x4 *>(cl + K * 4 + kb * 64), h1 = *reinterpret_cast<const f32x4 *>(cl + K * 4 + kb * 64 + 16);
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const int h = j >> 1, i = 2 * (j & 1);
                        const f32x4 sc = h ? s1 : s0, sh = h ? h1 : h0;
                        const f32x2 t = pk_fma(f32x2{sc[i], sc[i + 1]}, f32x2{r[h][i], r[h][i + 1]}, f32x2{sh[i], sh[i + 1]});
                        v2[j] = f32x2{fmaxf(t.x, 0.f), fmaxf(t.y, 0.f)};
                    }
                } else {
                    f32x4 c[5][2];
#pragma unroll
                    for (int q = PS ? 2 : 0; q < 5; ++q) {      // (PS: scale and shift are not needed)
                        c[q][0] = *reinterpret_cast<const f32x4 *>(cl + q * K * 4 + kb * 64);
                        c[q][1] = *reinterpret_cast<const f32x4 *>(cl + q * K * 4 + kb * 64 + 16);
                    }
                    if constexpr (PS) __builtin_amdgcn_sched_barrier(0);      // (without it this flavour's schedule needs 256 registers + scratch; with it 222)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const int h = j >> 1, i = 2 * (j & 1);
                        const f32x2 y = f32x2{r[h][i], r[h][i + 1]};
                        f32x2 dz = f32x2{r[2 + h][i], r[2 + h][i + 1]};
                        if constexpr (AMODE == A_DY_MAX) {
                            dz.x = (__float_as_int(r[4 + h][i]) == kin) ? dz.x : 0.f;
                            dz.y = (__float_as_int(r[4 + h][i + 1]) == kin) ? dz.y : 0.f;
                        }
                        f32x2 c0 = f32x2{0.f, 0.f}, pp = dz;
                        if constexpr (!PS) {
                            c0 = f32x2{c[0][h][i], c[0][h][i + 1]};
                            const f32x2 z = pk_fma(c0, y, f32x2{c[1][h][i], c[1][h][i + 1]});
                            pp = f32x2{z.x > 0.f ? dz.x : 0.f, z.y > 0.f ? dz.y : 0.f};
                        }
                        if constexpr (CP) {
                            const f32x2 t = pk_fma(f32x2{c[4][h][i], c[4][h][i + 1]}, y - f32x2{c[2][h][i], c[2][h][i + 1]}, f32x2{c[3][h][i], c[3][h][i + 1]});
                            // PS: dz IS scale * p at the group's argmax row (the reduction that made the BN-backward sums formed the same product of
                            // the same two floats): what the other path computes as c0 * pp
                            if constexpr (PS) v2[j] = pk_fma(f32x2{wcur, wcur}, t, dz);
                            else v2[j] = pk_fma(f32x2{wcur, wcur}, t, c0 * pp);
                        } else {
                            const f32x2 inner = pk_fma(c0, pp, -f32x2{c[3][h][i], c[3][h][i + 1]});
                            v2[j] = pk_fma(-f32x2{c[4][h][i], c[4][h][i + 1]}, y - f32x2{c[2][h][i], c[2][h][i + 1]}, inner);
                        }
                    }
                }
                if constexpr (NJ == 2) {      // (partial block: the upper half-wave holds a copy of the same four channels -- its k does not exist)
                    if (hi) { v2[0] = f32x2{0.f, 0.f}; v2[1] = f32x2{0.f, 0.f}; }
                }
                unsigned q0[4], q1[4], q2[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (j < NJ) split3_pair(v2[j], q0[j], q1[j], q2[j]);
                    else { q0[j] = 0u; q1[j] = 0u; q2[j] = 0u; }      // (partial block: k beyond KV)
                }
                af[0] = __builtin_bit_cast(bf16x8, make_uint4(q0[0], q0[1], q0[2], q0[3]));
                af[1] = __builtin_bit_cast(bf16x8, make_uint4(q1[0], q1[1], q1[2], q1[3]));
                af[2] = __builtin_bit_cast(bf16x8, make_uint4(q2[0], q2[1], q2[2], q2[3]));
            }
#else
            float v[8];
            if constexpr (AMODE == A_PLAIN) {
#pragma unroll
                for (int i = 0; i < 4; ++i) { v[i] = r[0][i]; v[4 + i] = r[1][i]; }
            } else if constexpr (AMODE == A_BNRELU) {
                const f32x4 s0 = *reinterpret_cast<const f32x4 *>(cl + kb * 64), s1 = *reinterpret_cast<const f32x4 *>(cl + kb * 64 + 16);
                const f32x4 h0 = *reinterpret_cast<const f32x4 *>(cl + K * 4 + kb * 64), h1 = *reinterpret_cast<const f32x4 *>(cl + K * 4 + kb * 64 + 16);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    v[i] = fmaxf(fmaf(s0[i], r[0][i], h0[i]), 0.f);
                    v[4 + i] = fmaxf(fmaf(s1[i], r[1][i], h1[i]), 0.f);
                }
            } else {
                f32x4 c[5][2];
#pragma unroll
                for (int q = 0; q < 5; ++q) {
                    c[q][0] = *reinterpret_cast<const f32x4 *>(cl + q * K * 4 + kb * 64);
                    c[q][1] = *reinterpret_cast<const f32x4 *>(cl + q * K * 4 + kb * 64 + 16);
                }
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float y = r[h][i];
                        float dz = r[2 + h][i];
                        if constexpr (AMODE == A_DY_MAX) dz = (__float_as_int(r[4 + h][i]) == kin) ? dz : 0.f;
                        const float z = fmaf(c[0][h][i], y, c[1][h][i]);
                        const float pp = z > 0.f ? dz : 0.f;
                        v[4 * h + i] = fmaf(-c[4][h][i], y - c[2][h][i], fmaf(c[0][h][i], pp, -c[3][h][i]));
                    }
            }
            uint2 a0, a1, a2, b0, b1, b2;
            split3(make_float4(v[0], v[1], v[2], v[3]), a0, a1, a2);
            split3(make_float4(v[4], v[5], v[6], v[7]), b0, b1, b2);
            af[0] = __builtin_bit_cast(bf16x8, make_uint4(a0.x, a0.y, b0.x, b0.y));
            af[1] = __builtin_bit_cast(bf16x8, make_uint4(a1.x, a1.y, b1.x, b1.y));
            af[2] = __builtin_bit_cast(bf16x8, make_uint4(a2.x, a2.y, b2.x, b2.y));
#endif
            bf16x8 bq[WN][3];
#pragma unroll
            for (int wn = 0; wn < WN; ++wn)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    bq[wn][pl] = *reinterpret_cast<const bf16x8 *>(((CB && wn == TB) ? wlb : wl + wn * 32 * ROWB) + pl * 2 * K + kb * 32);
            // smallest terms first; consecutive MFMAs go to different accumulators
            constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int wn = 0; wn < WN; ++wn)
                    acc[wn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PA[t]], bq[wn][PB[t]], acc[wn], 0, 0, 0);
        }
    };
    auto compute = [&](auto bi_, auto ci_) {
        constexpr int bi = decltype(bi_)::value, ci = decltype(ci_)::value;
        sfor<0, CK>([&](auto blk_) {
            constexpr int blk = decltype(blk_)::value;
            compute_block(std::integral_constant<int, ci * CK + blk>{}, &buf[bi][blk * NLD]);
        });
        if constexpr (KR && ci == NCH - 1) {
            // the partial block: the lane's first four values are channels KV - 4 .. KV - 1 (upper half-wave: a copy that meets zero weights), the rest zeros
            const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
            f32x4 rr[4];
            rr[0] = pbuf[0]; rr[1] = z4;
            rr[2] = (AMODE == A_DY_DENSE) ? pbuf[NLDP - 1] : z4; rr[3] = z4;
            compute_block(std::integral_constant<int, KBF>{}, rr);
        }
    };

    // ---- per-lane epilogue state: this lane's columns are n0 + 32 wn + l31 for every tile
    float s1[WN], s2[WN], biasv[WN];
    float rsc[WN], rsh[WN], rmu[WN], ris[WN];
    float gmx[WN], gmn[WN];
    int gix[WN], gin[WN];
    float s3[WN], s4[WN];          // XR: (s1, s2, s3, s4) = sums of p x, p y, p z, p
    float4 kq[WN];                 // XR: the folded first layer of this lane's column
#pragma unroll
    for (int wn = 0; wn < WN; ++wn) {
        const int colr = n0 + wn * 32 + l31;
        const int col = NR ? min(colr, p.Nout - 1) : colr;       // (NR: masked lanes read a valid column's constants and never store)
        s1[wn] = 0.f; s2[wn] = 0.f; s3[wn] = 0.f; s4[wn] = 0.f;
        kq[wn] = XR ? ld4(p.rd.scale + 4 * col) : make_float4(0.f, 0.f, 0.f, 0.f);
        biasv[wn] = (!CP && !(KR && DY) && p.bias && colr == col && (!CB || wn * 32 + l31 < CBW)) ? p.bias[col] : 0.f;     // (CP, ragged-k dX: no bias in a dX product; a compile-time zero frees WN registers at the limit)
        rsc[wn] = rsh[wn] = rmu[wn] = ris[wn] = 0.f;
        if (EPI == EPI_STORE_RED) { rsc[wn] = p.rd.scale[col]; rsh[wn] = p.rd.shift[col]; rmu[wn] = p.rd.mean[col]; ris[wn] = p.rd.invstd[col]; }
        gmx[wn] = -INFINITY; gmn[wn] = INFINITY; gix[wn] = 0; gin[wn] = 0;
        if constexpr (GS) gin[wn] = (p.gm.sgn[col] < 0.f) ? (int)0x80000000u : 0;      // (GS: gin holds the channel's sign flip, gmn is unused)
    }

    // epilogue of the tile starting at row0 (sub = its index inside the unit).  C/D layout: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 hi.
    auto epilogue = [&](int row0, int sub) {
        const int64_t ldy = p.ldy;
        if constexpr (XR) {
            // the rows' centred coordinates: one float4 per row, the same address for the 32 lanes of a half-wave (one broadcast line each)
            const float4 *xq = reinterpret_cast<const float4 *>(p.rd.y) + row0 + 4 * hi;
            // (four rows at a time, fenced: with all 16 coordinate loads hoisted to the top the flavour needs 256 registers and spills)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 xv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) xv[i] = xq[8 * g + i];
#pragma unroll
                for (int wn = 0; wn < WN; ++wn) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int r = 4 * g + i;       // C/D layout: row = (r & 3) + 8 (r >> 2) + 4 hi
                        const float z = fmaf(kq[wn].z, xv[i].z, fmaf(kq[wn].y, xv[i].y, fmaf(kq[wn].x, xv[i].x, kq[wn].w)));   // (as xyz_l1_bwd_kernel)
                        const float pp = z > 0.f ? acc[wn][r] : 0.f;
                        s1[wn] = fmaf(pp, xv[i].x, s1[wn]); s2[wn] = fmaf(pp, xv[i].y, s2[wn]); s3[wn] = fmaf(pp, xv[i].z, s3[wn]);
                        s4[wn] += pp;
                        acc[wn][r] = 0.f;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            return;
        }
#pragma unroll
        for (int wn = 0; wn < WN; ++wn) {
            const int col = n0 + wn * 32 + l31;
            const bool cok = (!NR || col < p.Nout) && (!CB || wn * 32 + l31 < CBW);
            float *yp = p.y + (int64_t)(row0 + 4 * hi) * ldy + col;
            if (EPI == EPI_STORE_RED) {
                if (cok) {
                const float *qp = p.rd.y + (int64_t)(row0 + 4 * hi) * ldy + col;
                float yv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) yv[r] = qp[(int64_t)((r & 3) + 8 * (r >> 2)) * ldy];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = acc[wn][r] + biasv[wn];
                    const float pp = fmaf(rsc[wn], yv[r], rsh[wn]) > 0.f ? v : 0.f;
                    // (masked: the value the sums are formed from, papc_bwd_red.store_masked.  The compacted flavours ALWAYS store it: every consumer
                    // of a dX applies the mask itself, so the two are interchangeable, and these flavours have no register left for the choice)
                    if constexpr (CP) yp[(int64_t)((r & 3) + 8 * (r >> 2)) * ldy] = pp;
                    else yp[(int64_t)((r & 3) + 8 * (r >> 2)) * ldy] = v;      // (a masked store of the padded layout goes to the tiled kernel: stream_go)
                    s1[wn] += pp;
                    s2[wn] = fmaf(pp, (yv[r] - rmu[wn]) * ris[wn], s2[wn]);
                    // (the stored value now comes out of a select: without a fence the scheduler forms all sixteen of them ahead of the stores and this
                    // flavour -- at its register limit -- spills; four rows at a time keep the temporaries at four)
                    if constexpr (CP) { if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0); }
                }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ro = (r & 3) + 8 * (r >> 2);
                    const float v = acc[wn][r] + biasv[wn];
                    if (EPI != EPI_GMAX && cok) yp[(int64_t)ro * ldy] = v;   // (EPI_GMAX: under the max the output itself stays unwritten)
                    s1[wn] += v;
                    s2[wn] = fmaf(v, v, s2[wn]);
                    if (GM) {
                        const int off = sub * 32 + ro + 4 * hi;
                        if constexpr (GS) {
                            const float key = __int_as_float(__float_as_int(v) ^ gin[wn]);      // (v < best  <=>  -v > -best: strict either way, first row wins)
                            if (key > gmx[wn]) { gmx[wn] = key; gix[wn] = off; }
                        } else {
                            if (v > gmx[wn]) { gmx[wn] = v; gix[wn] = off; }
                            if (v < gmn[wn]) { gmn[wn] = v; gin[wn] = off; }
                        }
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[wn][r] = 0.f;
            if (GM && sub == U - 1) {
                // close the group: merge the two half-waves (first offset wins ties), lanes 0-31 write
                if constexpr (GS) {
                    float vmx = gmx[wn];
                    int imx = gix[wn];
                    const float omx = __shfl_xor(vmx, 32);
                    const int oix = __shfl_xor(imx, 32);
                    if (omx > vmx || (omx == vmx && oix < imx)) { vmx = omx; imx = oix; }
                    if (hi == 0) {
                        const int64_t g = (int64_t)(row0 >> 5) >> geo.ushift;
                        const float sel = __int_as_float(__float_as_int(vmx) ^ gin[wn]);
                        p.gm.gmax[g * p.Nout + col] = sel; p.gm.amax[g * p.Nout + col] = imx;
                        p.gm.gmin[g * p.Nout + col] = sel; p.gm.amin[g * p.Nout + col] = imx;
                    }
                    gmx[wn] = -INFINITY; gix[wn] = 0;
                } else {
                float vmx = gmx[wn], vmn = gmn[wn];
                int imx = gix[wn], imn = gin[wn];
                const float omx = __shfl_xor(vmx, 32), omn = __shfl_xor(vmn, 32);
                const int oix = __shfl_xor(imx, 32), oin = __shfl_xor(imn, 32);
                if (omx > vmx || (omx == vmx && oix < imx)) { vmx = omx; imx = oix; }
                if (omn < vmn || (omn == vmn && oin < imn)) { vmn = omn; imn = oin; }
                if (hi == 0) {
                    const int64_t g = (int64_t)(row0 >> 5) >> geo.ushift;
                    p.gm.gmax[g * p.Nout + col] = vmx; p.gm.amax[g * p.Nout + col] = imx;
                    p.gm.gmin[g * p.Nout + col] = vmn; p.gm.amin[g * p.Nout + col] = imn;
                }
                gmx[wn] = -INFINITY; gmn[wn] = INFINITY; gix[wn] = 0; gin[wn] = 0;
                }
            }
        }
    };

    // ---- A_XYZ: the operand is 12 bytes per row; one float4 load per lane and tile, prefetched one tile ahead (a compiler-visible load:
    // this flavour issues no hidden ones)
    if constexpr (XYZ) {
        if (my_tiles > 0) {
            const float4 *xc = reinterpret_cast<const float4 *>(p.a.x);
            int row0 = tile_row0(0);
            float4 xn = xc[row0 + l31];
            for (int j = 0; j < my_tiles; ++j) {
                const float4 xv = xn;
                const int row0n = tile_row0(j + 1);
                xn = xc[row0n + l31];
                asm volatile("" ::: "memory");   // (keeps the weight / constant LDS reads inside the tile loop: hoisted, they are 250+ registers)
                sfor<0, KB16>([&](auto kb_) { compute_xyz(kb_, xv); });
                epilogue(row0, j & (U - 1));
                row0 = row0n;
            }
        }
    }
    // ---- main loop.  Chunk c of the flat (tile, chunk) sequence lives in buffer c & 1 and is prefetched two chunks ahead.
    if (!XYZ && my_tiles > 0) {
        int row0 = tile_row0(0);
        SB sa = bases(row0);
        // CP: (w, group offset) of this lane's row in tiles j, j + 1, j + 2
        unsigned vgnxt = voff_grp0;
        auto vg_of = [&](int g) -> unsigned { return (unsigned)((g * ldx + 8 * hi) * 4); };
        if constexpr (CP) {
            const int r1 = tile_row0(1);
            wcur = -geo.wrow[row0 + l31];            // (kept negated: the transform multiplies by -w)
            if constexpr (AMODE == A_DY_MAX) {
                sa.vg = vg_of(geo.seg_grp[(row0 + l31) >> 3]);
                vgnxt = vg_of(geo.seg_grp[(r1 + l31) >> 3]);
            }
        }
        issue(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, sa);
        issue(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, sa);
        wait_vm<ASM, nld_of(1)>();
        touch_buf(std::integral_constant<int, 0>{});
        for (int j = 0; j < my_tiles; ++j) {
            const int row0n = tile_row0(j + 1);
            SB sn = bases(row0n);
            if constexpr (CP) sn.vg = vgnxt;
            if (AMODE == A_DY_MAX || MC) kin = CP ? row0 + l31 : (row0 & ((1 << geo.kgshift) - 1)) + l31;
            // LATE1: the epilogue of the dX kernels issues compiler-visible loads (the layer below's y).  hipcc's own counted waits for
            // them proved unsound on hardware while asm loads it cannot see are in flight (late data landed in registers it had
            // already reused: wrong rows, timing dependent), so for that epilogue nothing hidden is outstanding: chunk 1 of the next
            // tile is issued AFTER the epilogue instead of before it (it still has chunk 0's whole compute phase to land).
#ifdef PAPC_STREAM_NO_LATE1
            constexpr bool LATE1 = false;    // (diagnostic build: reproduces the hazard described above; tools/probe/late1_isa.py)
#else
            constexpr bool LATE1 = (EPI == EPI_STORE_RED || EPI == EPI_XYZ_RED);   // (both epilogues issue compiler-visible loads)
#endif
            sfor<0, NCH>([&](auto c_) {
                constexpr int c = decltype(c_)::value;
                constexpr int bi = c & 1;
                if constexpr (c > 0) {   // (chunk 0 was waited for before the previous tile's stores went out)
                    wait_vm<ASM, nld_of(c + 1)>();   // (LATE1: chunk 1 was issued behind the epilogue's stores; chunk c + 1's loads are still the only younger ones)
                    touch_buf(std::integral_constant<int, bi>{});
                    if constexpr (c == NCH - 1) touch_p();
                }
                compute(std::integral_constant<int, bi>{}, c_);
                if constexpr (c + 2 < NCH) issue(std::integral_constant<int, bi>{}, std::integral_constant<int, c + 2>{}, sa);
                else if constexpr (!(LATE1 && c == NCH - 1)) issue(std::integral_constant<int, bi>{}, std::integral_constant<int, c + 2 - NCH>{}, sn);
            });
            // chunk 0 of the next tile: in flight since chunk NCH - 2 was consumed; only chunk 1's loads are younger.  Waiting
            // here, BEFORE the stores, keeps fresh stores out of every counted wait (vmcnt counts them too).
            wait_vm<ASM, LATE1 ? 0 : nld_of(1)>();
            touch_buf(std::integral_constant<int, 0>{});
            // CP: the weight of tile j + 1 and the group of tile j + 2, as ordinary (compiler-visible) loads issued HERE -- everything hidden has
            // landed (LATE1 drains above), and the epilogue below waits for its own, younger loads, so these two have landed by its end at no
            // extra wait; they are first read behind the epilogue
            float wld = 0.f;
            int gld = 0;
            if constexpr (CP) {
                static_assert(!CP || LATE1, "CP relies on the full drain ahead of the STORE_RED epilogue");
                wld = geo.wrow[row0n + l31];
                if constexpr (AMODE == A_DY_MAX) gld = geo.seg_grp[(tile_row0(j + 2) + l31) >> 3];
            }
            epilogue(row0, j & (U - 1));
            if constexpr (CP) {
                wcur = -wld;
                if constexpr (AMODE == A_DY_MAX) vgnxt = vg_of(gld);
            }
            if constexpr (LATE1) issue(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, sn);
            row0 = row0n;
            sa = sn;
        }
        wait_vm<ASM, 0>();
        touch_buf(std::integral_constant<int, 0>{});
        touch_buf(std::integral_constant<int, 1>{});
        touch_p();
    }

    // ---- compacted stack, forward: the copies' share of the statistics.  The ball query's padding copies are not rows of a compacted stack;
    // what is left of a group's nsample slots rides as a weight w = 1 + copies on the group's FIRST row (compact.hip), which starts an 8-row
    // segment -- rows 0 / 8 / 16 / 24 of a tile.  This wave re-reads those four rows of every tile it has just written (its own stores,
    // acknowledged: nothing hidden is in flight any more) and adds (w - 1) y, (w - 1) y^2: the partial row then carries sum w y, sum w y^2,
    // the padded tensor's statistics, and the stack needs no correction launch per layer (bn_stats_corr_kernel: 6-11 us on the serial chain).
    if constexpr (EPI == EPI_STORE && (AMODE == A_BNRELU || AMODE == A_PLAIN) && !NR && !KR) {
        if (geo.wstat && p.stats && my_tiles > 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            for (int j = 0; j < my_tiles; ++j) {
                const int row0 = tile_row0(j);
                float cfq[2], vq[2][WN];
#pragma unroll
                for (int q = 0; q < 2; ++q) {           // this half-wave's two candidate rows: segments hi and hi + 2 of the tile
                    const int64_t row = row0 + 8 * (2 * q + hi);
                    cfq[q] = geo.wstat[row] - 1.f;
#pragma unroll
                    for (int wn = 0; wn < WN; ++wn) vq[q][wn] = p.y[row * p.ldy + n0 + wn * 32 + l31];
                }
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int wn = 0; wn < WN; ++wn) {
                        s1[wn] = fmaf(cfq[q], vq[q][wn], s1[wn]);
                        s2[wn] = fmaf(cfq[q] * vq[q][wn], vq[q][wn], s2[wn]);
                    }
            }
        }
    }

    // ---- BN statistics (or BN-backward sums): one deterministic partial row per row-workgroup
    if (p.stats) {
        __syncthreads();   // every wave has left the weights: the reduction scratch may overwrite them
        float *red = reinterpret_cast<float *>(smem);
#pragma unroll
        for (int wn = 0; wn < WN; ++wn) {
            s1[wn] += __shfl_xor(s1[wn], 32);
            s2[wn] += __shfl_xor(s2[wn], 32);
            if (XR) { s3[wn] += __shfl_xor(s3[wn], 32); s4[wn] += __shfl_xor(s4[wn], 32); }
            if (hi == 0) {
                red[(0 * NW + wave) * NT + wn * 32 + l31] = s1[wn];
                red[(1 * NW + wave) * NT + wn * 32 + l31] = s2[wn];
                if (XR) {
                    red[(2 * NW + wave) * NT + wn * 32 + l31] = s3[wn];
                    red[(3 * NW + wave) * NT + wn * 32 + l31] = s4[wn];
                }
            }
        }
        __syncthreads();
        for (int i = tid; i < NSUM * NT; i += NW * 64) {
            const int which = i / NT, c = i - which * NT;
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) t += red[(which * NW + w) * NT + c];
            // statistics: [part][which][Nout]; XR: [part][Nout][4], the layout papc_xyz_l1_bwd_finalize_f32 reads
            const int64_t col_off = XR ? (int64_t)(n0 + c) * 4 + which : (int64_t)which * p.Nout + n0 + c;
            const int64_t row_ld = (int64_t)NSUM * p.Nout;
            if ((NR && n0 + c >= p.Nout) || (CB && c >= CBW)) continue;
            p.stats[(int64_t)blockIdx.x * row_ld + col_off] = t;
            for (int r = blockIdx.x + gridDim.x; r < p.parts; r += gridDim.x) p.stats[(int64_t)r * row_ld + col_off] = 0.f;
        }
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------------------
static int stream_ncu()
{
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ncu = prop.multiProcessorCount;
        else ncu = 256;
    }
    return ncu;
}

template <int AMODE, int EPI, int KB16, int WN, bool CP = false, int KV = KB16 * 16, bool NR = false, int CBW = WN * 32, bool PS = false>
static int stream_go(const GemmArgs &p, const StreamGeo &geo, hipStream_t st)
{
    // k blocks per prefetch chunk: two, unless the flavour's registers do not allow it (an asm-loaded buffer must never spill)
    constexpr bool KR = (KV != KB16 * 16);
    constexpr int KBF = KR ? KB16 - 1 : KB16;
    constexpr int CK2 = (KBF == 6) ? 1 : 2;   // (six k blocks: an even chunk count needs 1 or 3 per chunk, and 3 spills the asm-loaded ring)
    constexpr int CK = (KBF < 4 || AMODE == A_DY_MAX || AMODE == A_MAXCAT || (AMODE == A_DY_DENSE && (WN == 4 || KR)) || (AMODE == A_PLAIN && KB16 == 8 && WN == 4)) ? 1 : CK2;
    const int ncb = (p.Nout + CBW - 1) / CBW;
    // one workgroup per CU in total (weights + 8 waves of up to 256 registers fill it); column blocks of the same rows are
    // gridDim.x apart in the flat id, i.e. on the same XCD when gridDim.x % 8 == 0: the second reader of a row finds it in L2
    int gx = std::max(8, (stream_ncu() / ncb) & ~7);
    // ... except the two flavours light enough for TWO per CU (<= 128 registers, <= 76 KB of LDS: the [P | A] dX of a max layer without stored
    // output and the coordinates-operand forward): four waves per SIMD overlap their memory, VALU and matrix phases better than two
    // (round 5, same box: 96-102 -> 88 us and 39.6 -> 35.0 us per launch, step -6 us)
    if (!CP && (AMODE == A_MAXCAT || AMODE == A_XYZ)) gx *= 2;
    gx = std::min(gx, std::max(1, (geo.n_units + 7) / 8));
    if (gx > p.parts) gx = p.parts;
    dim3 grid((unsigned)gx, (unsigned)ncb);
    if constexpr (!CP && EPI == EPI_STORE_RED) { if (p.rd.masked) return 0; }      // (only the compacted flavours store the masked value; a select in the others spills)
    if constexpr (CP) {
        if (!knob(KNOB_STREAM_ASM)) return 0;
        if constexpr (PS) {
            GemmArgs q = p;
            q.a.d.gout = p.a.d.psel;       // (the kernel streams the [G, C] operand through the gout pointer)
            hipLaunchKernelGGL((stream_kernel<AMODE, EPI, KB16, CK, WN, true, true, KB16 * 16, false, WN * 32, true>), grid, dim3(512), 0, st, q, geo);
        } else {
            hipLaunchKernelGGL((stream_kernel<AMODE, EPI, KB16, CK, WN, true, true>), grid, dim3(512), 0, st, p, geo);
        }
        const int rc = check_launch("mlp stream gemm (compacted)");
        return rc ? rc : 1;
    }
    if constexpr (AMODE == A_XYZ) hipLaunchKernelGGL((stream_kernel<AMODE, EPI, KB16, CK, WN, false>), grid, dim3(512), 0, st, p, geo);   // (no streamed operand: no asm ring)
    else if (knob(KNOB_STREAM_ASM)) {
        // (two dX flavours hold 150-166 registers and run three waves per SIMD: the one folded into a coordinates-only first layer's sums, 1.558 ->
        // 1.551 ms per step, and the padded max layer's 256 -> 128, padded step 1.834 -> 1.818; the dense 64- and 96-channel dX kernels of config 3
        // qualify too and measured SLOWER there -- 5.58 -> 5.63 ms: its three branch streams already fill the SIMDs.  PAPC_STREAM_NW12=0: two waves)
        if constexpr ((EPI == EPI_XYZ_RED || (EPI == EPI_STORE_RED && AMODE == A_DY_MAX && KB16 == 16)) && WN == 2 && !NR && KV == KB16 * 16) {
            if (knob(KNOB_STREAM_NW12)) hipLaunchKernelGGL((stream_kernel<AMODE, EPI, KB16, CK, WN, true, false, KV, NR, CBW, false, 12>), grid, dim3(768), 0, st, p, geo);
            else hipLaunchKernelGGL((stream_kernel<AMODE, EPI, KB16, CK, WN, true, false, KV, NR, CBW>), grid, dim3(512), 0, st, p, geo);
        } else if constexpr ((EPI == EPI_GMAX || EPI == EPI_STORE_GMAX) && !NR && KV == KB16 * 16) {
            if (p.gm.sgn) hipLaunchKernelGGL((stream_kernel<AMODE, EPI, KB16, CK, WN, true, false, KV, NR, CBW, false, 8, true>), grid, dim3(512), 0, st, p, geo);
            else hipLaunchKernelGGL((stream_kernel<AMODE, EPI, KB16, CK, WN, true, false, KV, NR, CBW>), grid, dim3(512), 0, st, p, geo);
        } else {
            hipLaunchKernelGGL((stream_kernel<AMODE, EPI, KB16, CK, WN, true, false, KV, NR, CBW>), grid, dim3(512), 0, st, p, geo);
        }
    }
    else if constexpr (AMODE == A_MAXCAT || EPI == EPI_GMAX || EPI == EPI_XYZ_RED || KR || NR) return 0;   // (the compiler-scheduled ring of these flavours spills / is not built: the caller falls back)
    else hipLaunchKernelGGL((stream_kernel<AMODE, EPI, KB16, CK, WN, false>), grid, dim3(512), 0, st, p, geo);
    const int rc = check_launch("mlp stream gemm");
    return rc ? rc : 1;
}

template <int AMODE, int EPI>
static int stream_pick(const GemmArgs &p, const StreamGeo &geo, hipStream_t st)
{
    const int kb = p.Kin / 16;
    if (geo.wrow) {      // dX of a compacted stack: the two flavours SA2-shaped stacks need (128 -> 128 dense, 256 -> 128 under the max)
        if constexpr (AMODE == A_DY_DENSE && EPI == EPI_STORE_RED) { if (kb == 8 && p.Nout == 128) return stream_go<AMODE, EPI, 8, 4, true>(p, geo, st); }
        if constexpr (AMODE == A_DY_MAX && EPI == EPI_STORE_RED) {
            if (kb == 16 && p.Nout == 128) {
                if (p.a.d.psel) return stream_go<AMODE, EPI, 16, 2, true, 256, false, 64, true>(p, geo, st);
                return stream_go<AMODE, EPI, 16, 2, true>(p, geo, st);
            }
        }
        return 0;
    }
    // the MSG segmenter's 196-channel layer pair (segment/pointnet2/pointnet2.py:63, [128, 196, 256]): ragged k (196 = 12 k blocks + 4 channels) and
    // ragged n (196 of the last column block's columns exist); the column blocks are as wide as the weights' LDS image allows (<= 160 KB here)
    if (geo.wstat && (p.Kin == 196 || p.Nout == 196)) return 0;     // (weighted statistics are not built for the ragged flavours)
    if (p.Kin == 196 || p.Nout == 196) {
        if constexpr (AMODE == A_BNRELU && EPI == EPI_STORE) { if (p.Kin == 128 && p.Nout == 196) return stream_go<AMODE, EPI, 8, 4, false, 128, true>(p, geo, st); }
        if constexpr (AMODE == A_BNRELU && (EPI == EPI_STORE_GMAX || EPI == EPI_STORE)) { if (p.Kin == 196 && p.Nout % 128 == 0) return stream_go<AMODE, EPI, 13, 4, false, 196>(p, geo, st); }
        if constexpr (AMODE == A_DY_MAX && EPI == EPI_STORE_RED) { if (p.Kin == 256 && p.Nout == 196) return stream_go<AMODE, EPI, 16, 4, false, 256, true, 98>(p, geo, st); }
        if constexpr (AMODE == A_DY_DENSE && EPI == EPI_STORE_RED) { if (p.Kin == 196 && p.Nout % 128 == 0) return stream_go<AMODE, EPI, 13, 4, false, 196>(p, geo, st); }
        return 0;
    }
    // N tile: as many columns as the weights' LDS image allows (6 K + 16 bytes per column, <= ~100 KB), at most 128
    const int nt = p.Kin > 128 ? 64 : 128;
    int wn = std::min(p.Nout, nt) / 32;
    if (p.Nout % (32 * wn) != 0) {       // 96 = 3 column tiles (the MSG branches' 96-channel layers); otherwise whole 64- or 128-wide blocks only
        if (p.Nout == 96 && nt >= 96) wn = 3;
        else return 0;
    }
#define STREAM_CASE(KB, WNN) if (kb == KB && wn == WNN) return stream_go<AMODE, EPI, KB, WNN>(p, geo, st)
    if constexpr (EPI == EPI_GMAX) {
        STREAM_CASE(4, 4); STREAM_CASE(8, 4);            // 64 -> 128 and 128 -> 256 under the max
    } else if constexpr (AMODE == A_MAXCAT) {
        STREAM_CASE(12, 2); STREAM_CASE(24, 2);          // [P (2 C) | relu(bn(x)) (C)] with C = 64 / 128
    } else if constexpr (AMODE == A_XYZ) {
        // the layer above a coordinates-only first layer (xyz1.hip, papc_mlp_xyz_ok): 64 channels below, 64 or 128 above
        STREAM_CASE(4, 2); STREAM_CASE(4, 4);
    } else if constexpr (AMODE == A_DY_DENSE && (EPI == EPI_STORE || EPI == EPI_XYZ_RED)) {
        STREAM_CASE(4, 2); STREAM_CASE(8, 2);
    } else {
        STREAM_CASE(2, 2); STREAM_CASE(2, 4);
        STREAM_CASE(4, 2); STREAM_CASE(4, 4);
        STREAM_CASE(8, 2); STREAM_CASE(8, 4);
        STREAM_CASE(4, 3); STREAM_CASE(8, 3);          // 96 output channels
        STREAM_CASE(6, 2); STREAM_CASE(6, 4);          // 96 input channels
        if constexpr (AMODE == A_DY_DENSE || AMODE == A_DY_MAX || AMODE == A_BNRELU) { STREAM_CASE(16, 2); }
    }
#undef STREAM_CASE
    return 0;
}

static int ilog2_exact(int v)
{
    int s = 0;
    while ((1 << s) < v) ++s;
    return (1 << s) == v ? s : -1;
}

int stream_gemm_try(const GemmArgs &p, int amode, int epi, bool vec, hipStream_t st)
{
    auto why = [&](int site) -> int { if (getenv("PAPC_STREAM_WHY")) fprintf(stderr, "[stream_gemm_try] declined at site %d: amode %d epi %d M %lld Kin %d Nout %d vec %d\n", site, amode, epi, (long long)p.M, p.Kin, p.Nout, (int)vec); return 0; };
    if (!knob(KNOB_STREAM) || knob(KNOB_GEMM_F32) || !vec) return why(1);
    if (p.M % 32 != 0 || p.M / 32 < knob(KNOB_STREAM_MINTILES)) return why(2);
    if (((p.Kin % 32 != 0 || p.Nout % 32 != 0) && p.Kin != 196 && p.Nout != 196) || p.Kin % 4 != 0 || p.Kin < 32 || p.Kin > (amode == A_MAXCAT ? 384 : 256) || p.Nout < 64) return why(3);
    if (amode == A_MAXCAT && (p.a.d.C != 2 * p.Nout || p.Kin != 3 * p.Nout || p.a.ldx != p.Nout)) return why(4);   // Cout = 2 Cin, dense input rows
    if (p.wmap || p.nmap || p.ldy != p.Nout) return why(5);
    if (!(p.stats || epi == EPI_STORE)) return why(6);
    StreamGeo geo;
    geo.ushift = 0; geo.kgshift = 5;
    geo.rows_dev = p.rows_dev; geo.wrow = nullptr; geo.seg_grp = nullptr;
    geo.wstat = (p.rows_dev && epi == EPI_STORE && (amode == A_BNRELU || amode == A_PLAIN)) ? p.wstat : nullptr;
    if (p.wstat && !geo.wstat) return why(12);      // (weighted statistics: the storing forward flavours under a device-side row count only)
    if (amode == A_DY_DENSE || amode == A_DY_MAX) { geo.wrow = p.a.d.wrow; geo.seg_grp = p.a.d.seg_grp; geo.rows_dev = p.a.d.rows_dev; }
    if (geo.rows_dev && (p.M % 128 != 0 || epi == EPI_STORE_GMAX || epi == EPI_GMAX)) return why(7);   // (ragged groups: no fused group max)
    if (geo.wrow && (!geo.rows_dev || (amode == A_DY_MAX && !geo.seg_grp))) return why(8);
    if (epi == EPI_STORE_GMAX || epi == EPI_GMAX) {
        const int s = ilog2_exact(p.gm.K);
        if (s < 5 || p.M % p.gm.K != 0) return why(9);
        geo.ushift = s - 5;
    }
    if ((amode == A_DY_MAX && !geo.wrow) || amode == A_MAXCAT) {
        const int s = ilog2_exact(p.a.d.K);
        if (s < 5) return why(10);
        geo.kgshift = s;
    }
    geo.n_units = (int)((p.M / 32) >> geo.ushift);
    if (amode == A_BNRELU && epi == EPI_STORE) return stream_pick<A_BNRELU, EPI_STORE>(p, geo, st);
    if (amode == A_BNRELU && epi == EPI_GMAX) return knob(KNOB_STREAM_ASM) ? stream_pick<A_BNRELU, EPI_GMAX>(p, geo, st) : 0;
    if (amode == A_BNRELU && epi == EPI_STORE_GMAX) return stream_pick<A_BNRELU, EPI_STORE_GMAX>(p, geo, st);
    if (amode == A_PLAIN && epi == EPI_STORE) return stream_pick<A_PLAIN, EPI_STORE>(p, geo, st);
    if (amode == A_DY_DENSE && epi == EPI_STORE_RED) return stream_pick<A_DY_DENSE, EPI_STORE_RED>(p, geo, st);
    if (amode == A_DY_DENSE && epi == EPI_STORE) return stream_pick<A_DY_DENSE, EPI_STORE>(p, geo, st);   // (layer above a Gram-path first layer: no BN-backward sums)
    if (amode == A_DY_DENSE && epi == EPI_XYZ_RED) return (knob(KNOB_STREAM_ASM) && !geo.rows_dev) ? stream_pick<A_DY_DENSE, EPI_XYZ_RED>(p, geo, st) : 0;   // (the same layer, its dX folded into the first layer's sums)
    if (amode == A_XYZ && epi == EPI_STORE) return stream_pick<A_XYZ, EPI_STORE>(p, geo, st);
    if (amode == A_MAXCAT && epi == EPI_STORE_RED) return stream_pick<A_MAXCAT, EPI_STORE_RED>(p, geo, st);
    if (amode == A_DY_MAX && epi == EPI_STORE_RED) return stream_pick<A_DY_MAX, EPI_STORE_RED>(p, geo, st);
    return why(11);
}

}  // namespace papc
