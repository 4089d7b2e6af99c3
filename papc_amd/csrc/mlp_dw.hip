// mlp_dw.hip -- weight gradient of a conv1x1 layer on rows: dW[Cout,Cin] = sum_m dY[m,:]^T A(x)[m,:]   (gfx950)
//
// Both operands are recomputed in the load (dY from (dz|gout+argmax, y) and the BN constants; A(x) from the
// previous layer's pre-BN output, or gathered rows) -- see mlp_loaders.h.  The reduction runs over M (up to 1M
// rows), the output is tiny, so the grid is split over row chunks; every workgroup owns one <=128x128 output
// tile for its chunk and writes a partial that papc_reduce_partials_f32 sums in fixed order (deterministic).
//
// MFMA mapping (v_mfma_f32_32x32x2_f32): A operand = dY^T (i = cout, k = row), B operand = X (k = row, j = cin).
// Row stages sit in LDS row-major [RS rows][TOp | TIp channels]; lane l reads element [2*ks + (l>>5)][tile*32 + (l&31)]
// with ds_read_b32: the 32 lanes of each half read 32 consecutive banks -> conflict-free without padding.
//
// Software pipeline (same scheme as mlp_gemm.hip): the raw global loads of row stage s+1 are issued before the
// MFMAs of stage s and transformed + written to the other LDS buffer after them; one barrier per stage; two
// workgroups per CU.  For gathered rows the neighbour indices of stage s+2 are prefetched as well, so the
// idx -> address -> data dependency never sits in front of the matrix pipe.
#include <type_traits>
#include "mlp_loaders.h"

namespace papc {

typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int I, int N, class F>
__device__ __forceinline__ void dw_sfor(F &&f)       // f(std::integral_constant<int, I>{}) for I in [I, N): compile-time indices
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        dw_sfor<I + 1, N>(f);
    }
}

void fill_dy(DySrc &d, const papc_bwd_dy *s);
int check_dy(const papc_bwd_dy *dy, int64_t M, int C, bool *vec, const char *who);
int fill_asrc(ASrc &a, int a_mode, const float *x, int64_t ldx, const papc_group_src *grp, const float *sc,
              const float *sh, int Cin, const char *who);

struct DwArgs {
    ASrc x;       // A(x) producer  (PLAIN / BNRELU / GROUP)
    ASrc dy;      // dY producer    (DY_DENSE / DY_MAX)
    int64_t M; int Cin; int Cout; int rows_per_chunk;
    float *dw_partial;  // [n_chunks][Cout][Cin]
    float *db_partial;  // [n_chunks][Cout] or null
    int64_t part_ld;    // row stride (floats) of both partial buffers
    int xmap;           // map internal cin -> caller's column (GROUP)
    int TOp, TIp;       // padded tile widths (32 / 64 / 128)
    int RS;             // rows per stage (32 or 64)
    unsigned long long *dbg;  // PAPC_DW_DBG=1: cycle counters of workgroup 0 (development aid)
};

constexpr int DW_T = 128;              // output tile edge (channels)
constexpr int DW_TI_WIDE = 160;           // gather layers (Cin = D+3 in (128,160]) keep all of cin in ONE tile
constexpr int DW_STAGE_FLOATS = 9216;  // RS * (TOp + TIp) <= 32*(128+160) floats = 36 KiB per buffer

// Waves are arranged WO x WI over the output tile; each owns NTO x NTI 32x32 tiles, so one k-step needs NTO + NTI
// LDS operand reads for NTO*NTI MFMAs.  Absent tiles (narrow layers) are computed on clamped coordinates and not stored.
template <int XMODE, int DYMODE, bool VEC, int WO, int WI, int NTO, int NTI, int NSX>
__global__ __launch_bounds__(256, 2) void dw_kernel(DwArgs p)
{
    static_assert(WO * WI == 4, "4 waves");
    __shared__ __attribute__((aligned(16))) float smem[2 * DW_STAGE_FLOATS + DW_T];
    float *dbred = smem + 2 * DW_STAGE_FLOATS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hi = lane >> 5, l31 = lane & 31;
    const int wo = wave / WI, wi = wave % WI;
    const int o0 = blockIdx.y * DW_T, i0 = blockIdx.z * (NSX > 4 ? DW_TI_WIDE : DW_T);
    const int TOp = p.TOp, TIp = p.TIp, RS = p.RS;
    const int nto = min(TOp / 32, (p.Cout - o0 + 31) / 32);  // 32-wide tiles actually present
    const int nti = min(TIp / 32, (p.Cin - i0 + 31) / 32);
    // loader mapping: slot = tid + 256*i -> row = slot / cg, channel group = slot % cg.  cg | 256 for every tile width
    // except the 160-wide gather tile, so (outside that case) all slots of a thread share one channel group -- which the
    // per-thread BN constants rely on.  Slots past the stage (narrow tiles) are clamped duplicates: they re-load and
    // re-write the thread's last real slot, which keeps the fetch free of branches.
    const int cgy = TOp / 4, cgx = TIp / 4;
    const int nsy = (RS * cgy + 255) / 256, nsx = (RS * cgx + 255) / 256;
    int yr[4], xr[NSX], xk[NSX];
    const int kqy = (tid % cgy) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) yr[i] = (tid + 256 * (i < nsy ? i : nsy - 1)) / cgy;
#pragma unroll
    for (int i = 0; i < NSX; ++i) {
        const int s = tid + 256 * (i < nsx ? i : nsx - 1);
        xr[i] = s / cgx;
        xk[i] = (s - xr[i] * cgx) * 4;
    }
    const int kqx = xk[0];
    const bool use_jpre = (XMODE == A_GROUP) && p.x.g.idx != nullptr;

    floatx16 acc[NTO][NTI];
#pragma unroll
    for (int a = 0; a < NTO; ++a)
#pragma unroll
        for (int b = 0; b < NTI; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    int toff[NTO], tiff[NTI];
    bool tok_o[NTO], tok_i[NTI];
#pragma unroll
    for (int a = 0; a < NTO; ++a) { const int to = wo * NTO + a; tok_o[a] = to < nto; toff[a] = (tok_o[a] ? to : 0) * 32; }
#pragma unroll
    for (int b = 0; b < NTI; ++b) { const int ti = wi * NTI + b; tok_i[b] = ti < nti; tiff[b] = (tok_i[b] ? ti : 0) * 32; }

    const KConst kcy = make_kconst<DYMODE, VEC>(p.dy, o0 + kqy, p.Cout);
    const KConst kcx = make_kconst<XMODE, VEC>(p.x, i0 + kqx, p.Cin);
    float4 dbs = make_float4(0.f, 0.f, 0.f, 0.f);

    const int64_t mbeg = (int64_t)blockIdx.x * p.rows_per_chunk;
    const int64_t mend = min(p.M, mbeg + p.rows_per_chunk);

    RowCtx rowy[4], rowx[NSX];
    Raw3 rawy[4], rawx[NSX];
    int jpre[NSX];
#pragma unroll
    for (int i = 0; i < NSX; ++i) jpre[i] = -2;  // gathered rows: neighbour indices of the NEXT fetch, loaded one stage early

    auto prefetch_j = [&](int64_t m0) {
        if (use_jpre) {
#pragma unroll
            for (int i = 0; i < NSX; ++i) {
                const int64_t m = m0 + xr[i];
                jpre[i] = p.x.g.idx[m < p.M ? m : 0];
            }
        }
    };
    auto fetch = [&](int64_t m0) {  // loads only
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            rowy[i] = make_row<DYMODE>(p.dy, m0 + yr[i], mend);
            rawy[i] = fetch_a4<DYMODE, VEC>(p.dy, rowy[i], o0 + kqy, p.Cout);
        }
#pragma unroll
        for (int i = 0; i < NSX; ++i) {
            rowx[i] = make_row<XMODE>(p.x, m0 + xr[i], mend, use_jpre ? jpre[i] : -2);
            rawx[i] = fetch_a4<XMODE, VEC>(p.x, rowx[i], i0 + xk[i], p.Cin);
        }
    };
    auto finish = [&](float *Ys, float *Xs) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float4 vy = finish_a4<DYMODE, VEC>(p.dy, rowy[i], o0 + kqy, p.Cout, kcy, rawy[i]);
            if (i < nsy) { dbs.x += vy.x; dbs.y += vy.y; dbs.z += vy.z; dbs.w += vy.w; }
            *reinterpret_cast<float4 *>(&Ys[yr[i] * TOp + kqy]) = vy;
        }
#pragma unroll
        for (int i = 0; i < NSX; ++i) {
            const float4 vx = finish_a4<XMODE, VEC>(p.x, rowx[i], i0 + xk[i], p.Cin, kcx, rawx[i]);
            *reinterpret_cast<float4 *>(&Xs[xr[i] * TIp + xk[i]]) = vx;
        }
    };

    // ---- prologue
    bool have = mbeg < mend;
    if (have) {
        prefetch_j(mbeg);
        fetch(mbeg);
        prefetch_j(mbeg + RS);
        finish(smem, smem + RS * TOp);
    }
    __syncthreads();
    int buf = 0;
    int64_t m0 = mbeg;
    while (have) {
        const int64_t mn = m0 + RS;
        const bool have_next = mn < mend;
        if (have_next) {
            fetch(mn);                // uses jpre loaded during the previous stage
            prefetch_j(mn + RS);      // indices for the stage after next
        }
        {
            // MFMA phase with register double-buffered LDS operands: the reads of k-step ks+1 are issued before the
            // MFMAs of k-step ks, so the matrix pipe never waits on LDS latency
            const float *Ys = smem + buf * DW_STAGE_FLOATS + hi * TOp + l31;
            const float *Xs = smem + buf * DW_STAGE_FLOATS + RS * TOp + hi * TIp + l31;
            const int nks = RS / 2;  // even
            float a0[NTO], b0[NTI], a1[NTO], b1[NTI];
#pragma unroll
            for (int a = 0; a < NTO; ++a) a0[a] = Ys[toff[a]];
#pragma unroll
            for (int b = 0; b < NTI; ++b) b0[b] = Xs[tiff[b]];
            for (int ks = 0; ks < nks; ks += 2) {
                const float *y1 = Ys + (2 * ks + 2) * TOp, *x1 = Xs + (2 * ks + 2) * TIp;
#pragma unroll
                for (int a = 0; a < NTO; ++a) a1[a] = y1[toff[a]];
#pragma unroll
                for (int b = 0; b < NTI; ++b) b1[b] = x1[tiff[b]];
#pragma unroll
                for (int a = 0; a < NTO; ++a)
#pragma unroll
                    for (int b = 0; b < NTI; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[a], b0[b], acc[a][b], 0, 0, 0);
                const int k2 = (ks + 2 < nks) ? ks + 2 : ks;  // last pair: harmless re-read of valid rows
                const float *y2 = Ys + (2 * k2) * TOp, *x2 = Xs + (2 * k2) * TIp;
#pragma unroll
                for (int a = 0; a < NTO; ++a) a0[a] = y2[toff[a]];
#pragma unroll
                for (int b = 0; b < NTI; ++b) b0[b] = x2[tiff[b]];
#pragma unroll
                for (int a = 0; a < NTO; ++a)
#pragma unroll
                    for (int b = 0; b < NTI; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[a], b1[b], acc[a][b], 0, 0, 0);
            }
        }
        if (have_next) {
            float *Ys = smem + (buf ^ 1) * DW_STAGE_FLOATS;
            finish(Ys, Ys + RS * TOp);
        }
        lds_barrier();  // LDS-only barrier: the idx prefetch of stage s+2 stays in flight
        buf ^= 1;
        m0 = mn;
        have = have_next;
    }

    // ---- store the partial tile: row (cout) = (r&3)+8*(r>>2)+4*hi, col (cin) = l31
    float *out = p.dw_partial + (int64_t)blockIdx.x * p.part_ld;
#pragma unroll
    for (int b = 0; b < NTI; ++b) {
        const int ci = i0 + tiff[b] + l31;
        if (tok_i[b] && ci < p.Cin) {
            const int cig = p.xmap ? gk(p.x.g, ci) : ci;
#pragma unroll
            for (int a = 0; a < NTO; ++a) {
                if (tok_o[a]) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int co = o0 + toff[a] + (r & 3) + 8 * (r >> 2) + 4 * hi;
                        if (co < p.Cout) out[(int64_t)co * p.Cin + cig] = acc[a][b][r];
                    }
                }
            }
        }
    }

    // ---- bias gradient partial: column sums of dY over this chunk (only the first cin tile writes it).
    // Threads with the same channel group differ in their row lane (tid / cgy): add them in that fixed order.
    if (p.db_partial && blockIdx.z == 0) {
        const int rsy = 256 / cgy, ry0 = tid / cgy;
        if (tid < DW_T) dbred[tid] = 0.f;
        __syncthreads();
        for (int g = 0; g < rsy; ++g) {
            if (ry0 == g) { dbred[kqy + 0] += dbs.x; dbred[kqy + 1] += dbs.y; dbred[kqy + 2] += dbs.z; dbred[kqy + 3] += dbs.w; }
            __syncthreads();
        }
        if (tid < TOp && o0 + tid < p.Cout) p.db_partial[(int64_t)blockIdx.x * p.part_ld + o0 + tid] = dbred[tid];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Wave-specialised dW on the bf16 matrix pipe (fp32 operands as exact 3-way bf16 splits, see split3 in mlp_loaders.h).
//
// 16 waves per workgroup: 4 CONSUMER waves (2 x 2 over the <=128x128 output tile, NTO x NTI 32x32 accumulators each) that
// only ds_read + MFMA, and 3 groups of 4 PRODUCER waves that only load, transform, split and write LDS.  A stage is 16
// rows (one k block of v_mfma_f32_32x32x16_bf16); producer group g owns the stages s = g (mod 3): it issues the loads of
// its next stage right after finishing one and touches them two slots later, so the HBM latency is covered by plain
// issue -> wait -> transform code.  dW has no per-tile epilogue (one partial store per workgroup at the very end), so the
// consumers run the matrix pipe back to back: this is where the specialised structure pays.
//
// LDS image per stage: channel-major, K(=row)-contiguous: [TOp + TIp channels][plane0 | plane1 | plane2] x 32 B + 16 B pad
// (112 B, odd number of 16-B slots).  The MFMA reduces over ROWS, so the row-major global data is transposed on the way
// in: a producer thread owns a 4-row x 4-channel block, builds per channel the 4 consecutive-row values, splits them and
// writes 8 bytes per plane.  Lane l of a consumer reads its operand (channel tile*32 + (l&31), rows 8*(l>>5)..+7) with
// one ds_read_b128 per plane.
template <int XMODE, int DYMODE, int NTO, int NTI, int RS = 32>
__global__ __launch_bounds__(768, 3) void dw_ws_kernel(DwArgs p)
{
    constexpr int TO = NTO * 64, TI = NTI * 64;   // padded tile widths handled by this instantiation
    // a stage is RS = 32 rows = two k blocks; BOTH producer groups work in every slot (group g on rows 16g..16g+15 of the stage):
    // a SIMD needs two active waves to keep its VALU busy, a lone producer wave issues only every ~8-10 cycles.
    // RS = 64 (64 x 64 tiles only): with 128 operand channels a 32-row stage has work for HALF of the 512 producer threads, so the
    // narrow layers take 64-row stages (each group 32 rows: all producers busy, half as many barriers)
    static_assert(RS == 32 || (RS == 64 && TO == 64 && TI == 64), "64-row stages: 64 x 64 tiles");
    constexpr int G = 2, ROWB = 3 * RS * 2 + 16;
    constexpr int HG = RS / G;          // rows of a stage handled by one producer group
    constexpr int RQ = HG / 4;          // row quads per group (4 or 8)
    constexpr int STAGE_B = (TO + TI) * ROWB;
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_B];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hi = lane >> 5, l31 = lane & 31;
    const bool producer = wave >= 4;
    const int pgrp = producer ? (wave - 4) >> 2 : 0;
    const int lt = producer ? tid - 256 * (1 + pgrp) : tid;   // thread index within the producer group
    const int o0 = blockIdx.y * DW_T, i0 = blockIdx.z * DW_T;
    const int64_t mbeg = (int64_t)blockIdx.x * p.rows_per_chunk;
    const int64_t mend = min(p.M, mbeg + p.rows_per_chunk);
    const int n_stages = mbeg < mend ? (int)((mend - mbeg + RS - 1) / RS) : 0;

    if (producer) {
        // block assignment: threads [0, TO) carry dY blocks, [TO, TO + TI) carry X blocks, the rest idle (wave-uniform split).
        // A block = 4 consecutive rows x 4 consecutive channels; the channel quad is fixed for the whole kernel, so the BN
        // constants are loaded (and pre-folded) once.
        // (TO and TI are multiples of 64, so the role is the same for a whole wave: make that visible to the compiler)
        constexpr int NTY = TO * RQ / 4, NTX = TI * RQ / 4;   // threads carrying dY / X blocks (multiples of 64)
        static_assert(NTY + NTX <= 256, "a producer group has 256 threads");
        const bool isy = __builtin_amdgcn_readfirstlane((int)(lt < NTY)) != 0;
        const bool isx = !isy && __builtin_amdgcn_readfirstlane((int)(lt < NTY + NTX)) != 0;
        const int lb = isy ? lt : lt - NTY;
        // a wave covers 16 channel quads x all 4 row quads: its 8-byte LDS writes then spread over 32 of the 64 banks
        // (the 112-byte channel stride maps channel quads to only 4 distinct bank offsets; the row quads supply the rest)
        const int cq = (lb & 15) + 16 * (lb / (16 * RQ)), rq = (lb >> 4) & (RQ - 1);   // channel quad, row quad (0..RQ-1)
        const int ch = (isy ? o0 : i0) + cq * 4;           // first global channel of the block
        const int C = isy ? p.Cout : p.Cin;
        const bool chok = ch < C;                          // VEC shapes: C % 4 == 0, so a quad is all in or all out
        const int chc = chok ? ch : 0;
        // constants: dY = sc*p - (A + Bp*(y - mean)) with p = [sc*y + sh > 0] * dz, A = sc*c1, Bp = sc*c2*invstd (the same
        // value as dy_elem, 6 instead of 9 operations per element); X = relu(sc*x + sh).  Channels past C get all-zero
        // constants, which makes their values exactly 0 without a per-element select.
        float4 ksc = make_float4(0.f, 0.f, 0.f, 0.f), ksh = ksc, kmu = ksc, kA = ksc, kB = ksc;
        if (isy) {
            const DySrc &d = p.dy.d;
            ksc = ld4(d.scale + chc); ksh = ld4(d.shift + chc); kmu = ld4(d.mean + chc);
            const float4 is = ld4(d.invstd + chc), c1 = ld4(d.c1 + chc), c2 = ld4(d.c2 + chc);
            kA = make_float4(ksc.x * c1.x, ksc.y * c1.y, ksc.z * c1.z, ksc.w * c1.w);
            kB = make_float4(ksc.x * c2.x * is.x, ksc.y * c2.y * is.y, ksc.z * c2.z * is.z, ksc.w * c2.w * is.w);
        } else if (XMODE == A_BNRELU) {
            ksc = ld4(p.x.sc + chc); ksh = ld4(p.x.sh + chc);
        }
        if (!chok) { ksc = ksh = kmu = kA = kB = make_float4(0.f, 0.f, 0.f, 0.f); }
        char *const wbase = smem + ((isy ? 0 : TO) + cq * 4) * ROWB + pgrp * (HG * 2) + rq * 8;   // plane-relative: row HG*pgrp + 4*rq

        // addressing: workgroup-uniform chunk bases + 32-bit per-thread byte offsets (the host guarantees rows_per_chunk * ld * 4 < 2^31)
        const int64_t ld = isy ? (int64_t)p.Cout : p.x.ldx;
        const char *const b0 = reinterpret_cast<const char *>(isy ? p.dy.d.y + mbeg * p.Cout : p.x.x + mbeg * p.x.ldx);
        const char *const b1 = reinterpret_cast<const char *>(p.dy.d.dz ? p.dy.d.dz + mbeg * p.Cout : p.dy.d.y);   // DENSE only
        const uint32_t ldb = (uint32_t)ld * 4u;
        uint32_t off = (uint32_t)(pgrp * HG + rq * 4) * ldb + (uint32_t)chc * 4u;   // row 0 of this thread's block in its next stage
        const uint32_t adv = (uint32_t)RS * ldb;

        float4 ry[4], rz[4];   // raw rows: x | y, and dz (DENSE)
        float4 rg = make_float4(0.f, 0.f, 0.f, 0.f);   // MAX: gout of the block's group
        int4 ra = make_int4(-1, -1, -1, -1);           // MAX: argmax of the block's group
        int kin0 = 0;                                   // MAX: position of the block's first row inside its group
        int s_next = 0;      // next stage (every group takes part in every stage)
        int nvalid = 4;      // rows of the block that exist (ragged last stage only)
        auto issue = [&]() {
            if (s_next < n_stages && (isy || isx)) {
                const int64_t m0 = mbeg + (int64_t)s_next * RS + pgrp * HG + rq * 4;
                const int64_t left = mend - m0;
                nvalid = left >= 4 ? 4 : (left > 0 ? (int)left : 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint32_t o = off + (i < nvalid ? (uint32_t)i * ldb : 0u);   // rows past the end re-read a valid row
                    ry[i] = *reinterpret_cast<const float4 *>(b0 + (nvalid ? o : (uint32_t)chc * 4u));
                    if (isy && DYMODE == A_DY_DENSE) rz[i] = *reinterpret_cast<const float4 *>(b1 + (nvalid ? o : (uint32_t)chc * 4u));
                }
                if (isy && DYMODE == A_DY_MAX) {   // K % 4 == 0 (host-checked): the block's 4 rows share one group
                    const uint32_t mc = (uint32_t)(nvalid ? m0 : mbeg);
                    const int grp = (int)fdiv(mc, p.dy.d.divK);
                    kin0 = (int)mc - grp * p.dy.d.K;
                    rg = ld4(p.dy.d.gout + (int64_t)grp * p.Cout + chc);
                    ra = *reinterpret_cast<const int4 *>(p.dy.d.argmax + (int64_t)grp * p.Cout + chc);
                }
                off += adv;
            }
        };
        auto dyv = [&](float dz, float y, float sc, float sh, float mu, float A, float Bp) {
            const float z = fmaf(sc, y, sh);
            const float pp = z > 0.f ? dz : 0.f;
            return fmaf(sc, pp, -fmaf(Bp, y - mu, A));
        };
        auto finish = [&](char *buf) {
            if (!(isy || isx)) return;
            float4 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (isy) {
                    float4 dz;
                    if (DYMODE == A_DY_DENSE) dz = rz[i];
                    else {
                        const int kin = kin0 + i;
                        dz.x = ra.x == kin ? rg.x : 0.f; dz.y = ra.y == kin ? rg.y : 0.f;
                        dz.z = ra.z == kin ? rg.z : 0.f; dz.w = ra.w == kin ? rg.w : 0.f;
                    }
                    v[i].x = dyv(dz.x, ry[i].x, ksc.x, ksh.x, kmu.x, kA.x, kB.x);
                    v[i].y = dyv(dz.y, ry[i].y, ksc.y, ksh.y, kmu.y, kA.y, kB.y);
                    v[i].z = dyv(dz.z, ry[i].z, ksc.z, ksh.z, kmu.z, kA.z, kB.z);
                    v[i].w = dyv(dz.w, ry[i].w, ksc.w, ksh.w, kmu.w, kA.w, kB.w);
                } else if (XMODE == A_BNRELU) {
                    v[i].x = fmaxf(fmaf(ksc.x, ry[i].x, ksh.x), 0.f); v[i].y = fmaxf(fmaf(ksc.y, ry[i].y, ksh.y), 0.f);
                    v[i].z = fmaxf(fmaf(ksc.z, ry[i].z, ksh.z), 0.f); v[i].w = fmaxf(fmaf(ksc.w, ry[i].w, ksh.w), 0.f);
                } else {
                    v[i] = chok ? ry[i] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
            if (nvalid < 4) {   // ragged last stage of the last chunk
#pragma unroll
                for (int i = 0; i < 4; ++i) if (i >= nvalid) v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            // transpose in registers: per channel the 4 consecutive rows, split, 8 bytes per plane
            char *d = buf + (wbase - smem);
            uint2 q0, q1, q2;
            split3(make_float4(v[0].x, v[1].x, v[2].x, v[3].x), q0, q1, q2);
            constexpr int PL = RS * 2;   // bytes of one plane of a channel
            *reinterpret_cast<uint2 *>(d) = q0; *reinterpret_cast<uint2 *>(d + PL) = q1; *reinterpret_cast<uint2 *>(d + 2 * PL) = q2;
            split3(make_float4(v[0].y, v[1].y, v[2].y, v[3].y), q0, q1, q2);
            *reinterpret_cast<uint2 *>(d + ROWB) = q0; *reinterpret_cast<uint2 *>(d + ROWB + PL) = q1; *reinterpret_cast<uint2 *>(d + ROWB + 2 * PL) = q2;
            split3(make_float4(v[0].z, v[1].z, v[2].z, v[3].z), q0, q1, q2);
            *reinterpret_cast<uint2 *>(d + 2 * ROWB) = q0; *reinterpret_cast<uint2 *>(d + 2 * ROWB + PL) = q1; *reinterpret_cast<uint2 *>(d + 2 * ROWB + 2 * PL) = q2;
            split3(make_float4(v[0].w, v[1].w, v[2].w, v[3].w), q0, q1, q2);
            *reinterpret_cast<uint2 *>(d + 3 * ROWB) = q0; *reinterpret_cast<uint2 *>(d + 3 * ROWB + PL) = q1; *reinterpret_cast<uint2 *>(d + 3 * ROWB + 2 * PL) = q2;
        };

        issue();                               // stage 0 goes through LDS before the first slot
        if (n_stages > 0) finish(smem);
        s_next = 1;
        issue();
        __syncthreads();
        unsigned long long tw = 0, tf = 0, ti = 0, tb = 0;
        for (int t = 0; t < n_stages; ++t) {
            const unsigned long long c0 = p.dbg ? __builtin_readcyclecounter() : 0;
            if (s_next < n_stages) {         // s_next == t + 1: finish it into the other buffer, put stage t + 2 in flight
                unsigned long long c1 = c0, c2 = c0;
                if (p.dbg) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); c1 = __builtin_readcyclecounter(); }
                finish(smem + ((t + 1) & 1) * STAGE_B);
                if (p.dbg) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); c2 = __builtin_readcyclecounter(); }
                s_next += 1;
                issue();
                if (p.dbg) { const unsigned long long c3 = __builtin_readcyclecounter(); tw += c1 - c0; tf += c2 - c1; ti += c3 - c2; }
            }
            const unsigned long long c4 = p.dbg ? __builtin_readcyclecounter() : 0;
            lds_barrier();
            if (p.dbg) tb += __builtin_readcyclecounter() - c4;
        }
        if (p.dbg && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && lane == 0 && ((wave - 4) & 3) == 0) {
            unsigned long long *d = p.dbg + (1 + pgrp) * 8;
            d[0] = tw; d[1] = tf; d[2] = ti; d[3] = tb; d[4] = (unsigned long long)n_stages;
        }
    } else {
        const int wo = wave >> 1, wi = wave & 1;
        floatx16 acc[NTO][NTI];
#pragma unroll
        for (int a = 0; a < NTO; ++a)
#pragma unroll
            for (int b = 0; b < NTI; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
        __syncthreads();
        unsigned long long tm = 0, tb = 0;
        for (int t = 0; t < n_stages; ++t) {
            const unsigned long long c0 = p.dbg ? __builtin_readcyclecounter() : 0;
            const char *Yb = smem + (t & 1) * STAGE_B, *Xb = Yb + TO * ROWB;
            constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};   // smallest terms first
#pragma unroll
            for (int kb = 0; kb < RS / 16; ++kb) {
                bf16x8 ya[NTO][3], xb[NTI][3];
#pragma unroll
                for (int a = 0; a < NTO; ++a)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        ya[a][pl] = *reinterpret_cast<const bf16x8 *>(Yb + ((wo * NTO + a) * 32 + l31) * ROWB + pl * (RS * 2) + kb * 32 + hi * 16);
#pragma unroll
                for (int b = 0; b < NTI; ++b)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        xb[b][pl] = *reinterpret_cast<const bf16x8 *>(Xb + ((wi * NTI + b) * 32 + l31) * ROWB + pl * (RS * 2) + kb * 32 + hi * 16);
#pragma unroll
                for (int tt = 0; tt < 6; ++tt)
#pragma unroll
                    for (int a = 0; a < NTO; ++a)
#pragma unroll
                        for (int b = 0; b < NTI; ++b)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ya[a][PA[tt]], xb[b][PB[tt]], acc[a][b], 0, 0, 0);
            }
            const unsigned long long c1 = p.dbg ? __builtin_readcyclecounter() : 0;
            lds_barrier();
            if (p.dbg) { tm += c1 - c0; tb += __builtin_readcyclecounter() - c1; }
        }
        if (p.dbg && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && tid == 0) { p.dbg[0] = tm; p.dbg[1] = tb; p.dbg[4] = (unsigned long long)n_stages; }
        // ---- store the partial tile: row (cout) = (r&3)+8*(r>>2)+4*hi, col (cin) = l31
        float *out = p.dw_partial + (int64_t)blockIdx.x * p.part_ld;
#pragma unroll
        for (int b = 0; b < NTI; ++b) {
            const int ci = i0 + (wi * NTI + b) * 32 + l31;
            if (ci < p.Cin) {
#pragma unroll
                for (int a = 0; a < NTO; ++a) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int co = o0 + (wo * NTO + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                        if (co < p.Cout) out[(int64_t)co * p.Cin + ci] = acc[a][b][r];
                    }
                }
            }
        }
    }
    // Bias gradient: the bias feeds a train-mode BatchNorm, whose backward removes the per-channel mean of its input
    // gradient -- sum_m dY[m, c] = sc * (sum p - M c1 - c2 sum xhat) = 0 identically.  The exact value is written instead
    // of accumulating 1e-8-sized rounding noise.
    if (p.db_partial && blockIdx.z == 0) {
        if (tid < TO && o0 + tid < p.Cout) p.db_partial[(int64_t)blockIdx.x * p.part_ld + o0 + tid] = 0.f;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// dW of a first layer with a handful of input channels: a GROUP input with D = 0 (coordinates only, Cin = 3: SA1 of the SSG / MSG
// classifiers) or D = 4 (coordinates + normals padded to four, Cin = 7: SA1 of the part-segmentation models,
// segment/pointnet2/pointnet2.py:28-30).  A 128x32 MFMA tile would carry 3 (7) useful columns, so this is a streaming reduction
// instead: a lane owns 4 output channels (float4 loads of dz and y), 16 lanes cover a 64-channel row, and each lane keeps
// 4 x NC running sums of dY[m, c] * x[m, k] (k: the D feature columns, then xyz_j - centre); the 256-thread workgroup reduces its row
// slots in LDS in fixed order.  HBM-bound on the (dz, y) stream.
// ---------------------------------------------------------------------------------------------------------------------
template <int NF>      // float4 feature slots ahead of the coordinate slot (D = 4 NF)
__global__ __launch_bounds__(256) void dw_xyz_kernel(DwArgs p)
{
    constexpr int NC = 4 * NF + 3;           // input channels, internal order [feats, xyz]
    constexpr int NS = NF + 1;               // float4 slots of a row
    __shared__ float red[256 * 4 * NC];
    const int tid = threadIdx.x;
    const int CQ = p.Cout >> 2;              // channel quads per row (<= 64)
    const int RSL = 256 / CQ;                // row slots of the workgroup
    const int cq = tid % CQ, slot = tid / CQ;
    const bool act = slot < RSL;
    const int c = cq * 4;
    const DySrc &d = p.dy.d;
    const float4 ksc = ld4(d.scale + c), ksh = ld4(d.shift + c), kmu = ld4(d.mean + c);
    const float4 is = ld4(d.invstd + c), c1 = ld4(d.c1 + c), c2 = ld4(d.c2 + c);
    const float4 kA = make_float4(ksc.x * c1.x, ksc.y * c1.y, ksc.z * c1.z, ksc.w * c1.w);
    const float4 kB = make_float4(ksc.x * c2.x * is.x, ksc.y * c2.y * is.y, ksc.z * c2.z * is.z, ksc.w * c2.w * is.w);
    const int64_t mbeg = (int64_t)blockIdx.x * p.rows_per_chunk;
    const int64_t mend = min(p.M, mbeg + p.rows_per_chunk);
    float a[4][NC];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NC; ++j) a[i][j] = 0.f;
    auto dyv = [&](float dz, float y, float sc, float sh, float mu, float A, float Bp) {
        const float z = fmaf(sc, y, sh);
        const float pp = z > 0.f ? dz : 0.f;
        return fmaf(sc, pp, -fmaf(Bp, y - mu, A));
    };
    constexpr int U = 4;                     // rows in flight per lane
    constexpr int XR = 256;                  // rows whose input is staged at a time
    // the row's input (gathered features, centred coordinates) is fetched ONCE per row by one thread and staged in LDS: the Cout / 4 lanes
    // of a row would otherwise each repeat the same index / feature / coordinate / centroid loads (ten memory instructions per lane and row
    // against two useful ones)
    __shared__ float4 xrow[XR][NS];
    for (int64_t base = mbeg; base < mend; base += XR) {
        const int nrows = (int)min((int64_t)XR, mend - base);
        __syncthreads();
        for (int rr = tid; rr < nrows; rr += 256) {
            const RowCtx r = make_row<A_GROUP>(p.x, base + rr, p.M);
#pragma unroll
            for (int sl = 0; sl < NS; ++sl) {     // slots 0 .. NF-1: features; slot NF: (x - cx, y - cy, z - cz, 0); zero for a no-hit row
                const Raw3 w = fetch_a4<A_GROUP, true>(p.x, r, 4 * sl, NC);
                xrow[rr][sl] = finish_a4<A_GROUP, true>(p.x, r, 4 * sl, NC, KConst{}, w);
            }
        }
        __syncthreads();
        if (!act) continue;
        for (int r0 = slot; r0 < nrows; r0 += U * RSL) {
            float4 vy[U], vz[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int rr = min(r0 + u * RSL, nrows - 1);
                vy[u] = ld4(d.y + (base + rr) * p.Cout + c);
                vz[u] = ld4(d.dz + (base + rr) * p.Cout + c);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int rr = r0 + u * RSL;
                if (rr >= nrows) continue;
                float g[4];
                g[0] = dyv(vz[u].x, vy[u].x, ksc.x, ksh.x, kmu.x, kA.x, kB.x);
                g[1] = dyv(vz[u].y, vy[u].y, ksc.y, ksh.y, kmu.y, kA.y, kB.y);
                g[2] = dyv(vz[u].z, vy[u].z, ksc.z, ksh.z, kmu.z, kA.z, kB.z);
                g[3] = dyv(vz[u].w, vy[u].w, ksc.w, ksh.w, kmu.w, kA.w, kB.w);
#pragma unroll
                for (int sl = 0; sl < NS; ++sl) {
                    const float4 x = xrow[rr][sl];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        a[i][4 * sl + 0] = fmaf(g[i], x.x, a[i][4 * sl + 0]);
                        a[i][4 * sl + 1] = fmaf(g[i], x.y, a[i][4 * sl + 1]);
                        a[i][4 * sl + 2] = fmaf(g[i], x.z, a[i][4 * sl + 2]);
                        if (sl < NF) a[i][4 * sl + 3] = fmaf(g[i], x.w, a[i][4 * sl + 3]);     // (the coordinate slot's 4th element is a structural zero)
                    }
                }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NC; ++j) red[tid * (4 * NC) + i * NC + j] = act ? a[i][j] : 0.f;
    __syncthreads();
    // thread t < Cout * NC: (channel, column); sum over the row slots in slot order
    float *out = p.dw_partial + (int64_t)blockIdx.x * p.part_ld;
    for (int t = tid; t < p.Cout * NC; t += 256) {
        const int ch = t / NC, j = t - ch * NC;
        const int q = ch >> 2, i = ch & 3;
        float sacc = 0.f;
        for (int sl = 0; sl < RSL; ++sl) sacc += red[(sl * CQ + q) * (4 * NC) + i * NC + j];
        out[(int64_t)ch * NC + gk(p.x.g, j)] = sacc;       // internal column j -> the caller's weight column
    }
    if (p.db_partial) {
        for (int t = tid; t < p.Cout; t += 256) p.db_partial[(int64_t)blockIdx.x * p.part_ld + t] = 0.f;   // exact (see dw_ws_kernel)
    }
}

static bool dw_f32_exact()
{
    return knob(KNOB_DW_F32) == 1;
}

// ---------------------------------------------------------------------------------------------------------------------
// Row-streaming dW for narrow inputs (Cin = 64; PAPC_DW_ROWS): no LDS staging, no producer / consumer split, no barrier
// in the main loop.  The MFMA contracts over ROWS, and lane (channel = lane & 31, half = lane >> 5) wants 8 CONSECUTIVE ROWS of its
// channel: with row-major data that is 8 dword loads whose 32 lanes read 32 consecutive floats of one row -- coalesced 128-byte
// segments straight into the operand layout (a dwordx4 load would deliver 4 channels of one row: the transpose dw_ws_kernel does
// through LDS).  A lane's channel is fixed per tile, so the BN / BN-backward constants sit in registers.  A workgroup owns a row
// chunk and a 64-channel block of Cout; its 8 waves take the chunk's 16-row blocks round-robin, each accumulating the whole
// 64 x 64 tile (4 accumulators), and fold them in a fixed 3-level tree through LDS at the end (one partial row per workgroup, as
// the staged kernels write).  What this buys: every thread loads, transforms and multiplies -- no idle consumers, no barrier skew.
// XYZ: the x operand is the activation of a coordinates-only first layer, recomputed from the row's centred coordinates (xyz1.hip):
// relu(wf_c . x[m] + t_c) with xc [M, 4] = p.x.x and the folded layer wf [Cin, 4] = p.x.sc -- 12 bytes per row instead of 4 per element.
template <int DYMODE, bool XYZ = false>
__global__ __launch_bounds__(512, 2) void dw_rows_kernel(DwArgs p)
{
    constexpr int NTO = 2, NTI = 2;
    __shared__ float slab[4][NTO * NTI * 16][64];   // 64 KB: the upper half of the waves parks its accumulators here
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hi = lane >> 5, l31 = lane & 31;
    const int o0 = blockIdx.y * 64, i0 = blockIdx.z * 64;
    const int64_t mbeg = (int64_t)blockIdx.x * p.rows_per_chunk;
    const int64_t mend = min(p.M, mbeg + p.rows_per_chunk);
    const int n_kb = mbeg < mend ? (int)((mend - mbeg) >> 4) : 0;       // (whole 16-row blocks only: host-checked)
    const DySrc &d = p.dy.d;
    const int Cout = p.Cout, ldx = (int)p.x.ldx;

    float ksc[NTO], ksh[NTO], kmu[NTO], kA[NTO], kB[NTO], xs[NTI], xh[NTI];
#pragma unroll
    for (int a = 0; a < NTO; ++a) {
        const int c = o0 + 32 * a + l31;
        ksc[a] = d.scale[c]; ksh[a] = d.shift[c]; kmu[a] = d.mean[c];
        kA[a] = ksc[a] * d.c1[c];
        kB[a] = ksc[a] * d.c2[c] * d.invstd[c];
    }
#pragma unroll
    for (int b = 0; b < NTI; ++b) {
        if (XYZ) { xs[b] = 0.f; xh[b] = 0.f; }
        else { xs[b] = p.x.sc[i0 + 32 * b + l31]; xh[b] = p.x.sh[i0 + 32 * b + l31]; }
    }
    float4 wfk[NTI];
#pragma unroll
    for (int b = 0; b < NTI; ++b) wfk[b] = XYZ ? ld4(p.x.sc + 4 * (i0 + 32 * b + l31)) : make_float4(0.f, 0.f, 0.f, 0.f);

    floatx16 acc[NTO][NTI];
#pragma unroll
    for (int a = 0; a < NTO; ++a)
#pragma unroll
        for (int b = 0; b < NTI; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    struct Raw { float y[NTO][8], z[NTO][8], x[XYZ ? 3 : NTI][8]; int am[NTO]; };
    // addresses: wave-uniform base + a 32-bit byte offset per lane (host-checked: M * max(Cout, ldx) * 4 < 2^32, whole 16-row blocks only):
    // one 32-bit add per load instead of a 64-bit multiply-add and a row clamp
    auto ldg = [](const float *base, uint32_t byte_off) { return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + byte_off); };
    auto ldgi = [](const int *base, uint32_t byte_off) { return *reinterpret_cast<const int *>(reinterpret_cast<const char *>(base) + byte_off); };
    const uint32_t ystride = (uint32_t)Cout * 4u, xstride = (uint32_t)(XYZ ? 4 : ldx) * 4u;
    const uint32_t oy_lane = (uint32_t)((mbeg + 8 * hi) * Cout + o0 + l31) * 4u;
    const uint32_t ox_lane = XYZ ? (uint32_t)((mbeg + 8 * hi) * 4) * 4u : (uint32_t)((mbeg + 8 * hi) * ldx + i0 + l31) * 4u;
    auto fetch = [&](int kb, Raw &w) {
        const uint32_t oy = oy_lane + (uint32_t)kb * 16u * ystride, ox = ox_lane + (uint32_t)kb * 16u * xstride;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#pragma unroll
            for (int a = 0; a < NTO; ++a) {
                w.y[a][j] = ldg(d.y, oy + (uint32_t)j * ystride + 128u * a);
                if (DYMODE == A_DY_DENSE) w.z[a][j] = ldg(d.dz, oy + (uint32_t)j * ystride + 128u * a);
            }
            if (XYZ) {                                       // (the same 12 bytes for the 32 lanes of a half: one broadcast line)
                w.x[0][j] = ldg(p.x.x, ox + (uint32_t)j * 16u); w.x[1][j] = ldg(p.x.x, ox + (uint32_t)j * 16u + 4u); w.x[2][j] = ldg(p.x.x, ox + (uint32_t)j * 16u + 8u);
            } else {
#pragma unroll
                for (int b = 0; b < NTI; ++b) w.x[b][j] = ldg(p.x.x, ox + (uint32_t)j * xstride + 128u * b);
            }
        }
        if (DYMODE == A_DY_MAX) {   // K % 16 == 0 (host-checked): the block's 16 rows share one group
            const uint32_t g = fdiv((uint32_t)mbeg + 16u * (uint32_t)kb, d.divK);     // (no 64-bit division between the loads)
            const uint32_t og = (g * (uint32_t)Cout + (uint32_t)(o0 + l31)) * 4u;
#pragma unroll
            for (int a = 0; a < NTO; ++a) {
                w.z[a][0] = ldg(d.gout, og + 128u * a);
                w.am[a] = ldgi(d.argmax, og + 128u * a);
            }
        }
    };
    auto split8 = [&](const float (&vin)[8], bf16x8 (&pl)[3]) {
        uint2 a0, a1, a2, b0, b1, b2;
        split3(make_float4(vin[0], vin[1], vin[2], vin[3]), a0, a1, a2);
        split3(make_float4(vin[4], vin[5], vin[6], vin[7]), b0, b1, b2);
        pl[0] = __builtin_bit_cast(bf16x8, make_uint4(a0.x, a0.y, b0.x, b0.y));
        pl[1] = __builtin_bit_cast(bf16x8, make_uint4(a1.x, a1.y, b1.x, b1.y));
        pl[2] = __builtin_bit_cast(bf16x8, make_uint4(a2.x, a2.y, b2.x, b2.y));
    };
    auto compute = [&](int kb, const Raw &w) {
        int kin0 = 0;
        if (DYMODE == A_DY_MAX) {
            const uint32_t b0 = (uint32_t)mbeg + 16u * (uint32_t)kb;
            kin0 = (int)(b0 - fdiv(b0, d.divK) * (uint32_t)d.K) + 8 * hi;
        }
        bf16x8 pa[NTO][3], pb[NTI][3];
#pragma unroll
        for (int a = 0; a < NTO; ++a) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float y = w.y[a][j];
                float dz;
                if (DYMODE == A_DY_DENSE) dz = w.z[a][j];
                else dz = (w.am[a] == kin0 + j) ? w.z[a][0] : 0.f;
                const float z = fmaf(ksc[a], y, ksh[a]);
                const float pp = z > 0.f ? dz : 0.f;
                v[j] = fmaf(ksc[a], pp, -fmaf(kB[a], y - kmu[a], kA[a]));
            }
            split8(v, pa[a]);
        }
#pragma unroll
        for (int b = 0; b < NTI; ++b) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (XYZ) v[j] = fmaxf(fmaf(wfk[b].z, w.x[2][j], fmaf(wfk[b].y, w.x[1][j], fmaf(wfk[b].x, w.x[0][j], wfk[b].w))), 0.f);
                else v[j] = fmaxf(fmaf(xs[b], w.x[b][j], xh[b]), 0.f);
            }
            split8(v, pb[b]);
        }
        constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};   // smallest terms first
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int a = 0; a < NTO; ++a)
#pragma unroll
                for (int b = 0; b < NTI; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[a][PA[t]], pb[b][PB[t]], acc[a][b], 0, 0, 0);
    };

    // this wave's blocks: wave, wave + 8, ...; the raw operands of the next PF - 1 blocks are in flight while one is computed (a ring of PF
    // register buffers, as deep as the flavour's registers allow).  No branch sits around a load (block indices past the chunk are clamped to
    // the wave's last block: re-read, never used), no 64-bit division between them, and scheduling barriers keep loads ahead of the block's
    // arithmetic: otherwise the compiler collapses the ring (loads sunk behind the next block's MFMAs, every wait a vmcnt(0))
    constexpr int PF = (DYMODE == A_DY_MAX && !XYZ) ? 3 : 2;
    const int nb = (n_kb - wave + 7) >> 3;          // blocks of this wave
    if (nb > 0) {
        const int last = wave + 8 * (nb - 1);
        Raw rr[PF];
#pragma unroll
        for (int i = 0; i < PF - 1; ++i) fetch(min(wave + 8 * i, last), rr[i]);
        const int n_main = nb - nb % PF;
        for (int j0 = 0; j0 < n_main; j0 += PF) {
#pragma unroll
            for (int i = 0; i < PF; ++i) {
                const int kb = wave + 8 * (j0 + i);
                fetch(min(kb + 8 * (PF - 1), last), rr[(i + PF - 1) % PF]);
                __builtin_amdgcn_sched_barrier(0);
                compute(kb, rr[i]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int i = 0; i < PF - 1; ++i)
            if (n_main + i < nb) compute(wave + 8 * (n_main + i), rr[i]);
    }

    // ---- fold the 8 waves' tiles: 4 -> LDS, +4; 2 -> LDS, +2; 1 -> LDS, +1 (fixed order); slab[w][reg][lane]: conflict-free
#pragma unroll
    for (int half = 4; half >= 1; half >>= 1) {
        if (wave >= half && wave < 2 * half) {
#pragma unroll
            for (int a = 0; a < NTO; ++a)
#pragma unroll
                for (int b = 0; b < NTI; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) slab[wave - half][(a * NTI + b) * 16 + r][lane] = acc[a][b][r];
        }
        __syncthreads();
        if (wave < half) {
#pragma unroll
            for (int a = 0; a < NTO; ++a)
#pragma unroll
                for (int b = 0; b < NTI; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[a][b][r] += slab[wave][(a * NTI + b) * 16 + r][lane];
        }
        __syncthreads();
    }
    if (wave == 0) {   // C/D layout: row (cout) = (r & 3) + 8 (r >> 2) + 4 half, col (cin) = lane & 31
        float *out = p.dw_partial + (int64_t)blockIdx.x * p.part_ld;
#pragma unroll
        for (int a = 0; a < NTO; ++a)
#pragma unroll
            for (int b = 0; b < NTI; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = o0 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    out[(int64_t)co * p.Cin + i0 + 32 * b + l31] = acc[a][b][r];
                }
    }
    // (bias gradient: exactly 0 under a train-mode BN, see dw_ws_kernel)
    if (p.db_partial && blockIdx.z == 0 && tid < 64) p.db_partial[(int64_t)blockIdx.x * p.part_ld + o0 + tid] = 0.f;
}

// ---------------------------------------------------------------------------------------------------------------------
// Row-streaming dW for 128-channel inputs and 256-channel blocks of Cout (PAPC_DW_ROWSX): the hybrid of dw_rows_kernel and the staged
// kernels.  Each of the 8 waves owns 32 channels of Cout and streams ITS dY operand straight into the MFMA layout (coalesced dword
// loads, constants in registers: nothing is transformed twice, nothing of dY touches LDS).  The input operand x is common to all
// waves: the workgroup transforms each 16-row block of it once (4 values per thread), splits it and parks the three bf16 planes in
// LDS ([plane][channel][16 rows], 48-byte channel stride: conflict-free ds_read_b128), double-buffered, ONE barrier per block --
// and since all waves do identical work the barrier costs little skew.  A wave accumulates 32 x 128 of dW (4 accumulators) and
// stores its rows of the partial itself: no cross-wave fold.
// CP (compacted stack, compact.hip): the row count comes from device memory (the chunk size is derived from it here), dY carries the row's
// multiplicity weight in its BatchNorm-backward term, groups are ragged multiples of 8 rows (a lane's 8 rows share a group: seg_grp) and
// argmax holds absolute rows.
// RG (ragged widths: the MSG segmenter's 196-channel pair, pointnet2.py:63): Cout need not fill the workgroup's channel block (lanes beyond it read
// its last channel and store nothing), and the input may be wider than 128 channels -- blockIdx.z walks 128-channel blocks of it, row stride
// p.x.ldx, lanes beyond Cin read its last channel and their columns of the partial are not stored.
template <int DYMODE, int NW, bool CP = false, bool RG = false>      // NW waves = NW 32-channel groups of Cout per workgroup: 8 (256-channel blocks) or 4 (128-channel blocks, two workgroups per CU)
__global__ __launch_bounds__(64 * NW, 2) void dw_rowsx_kernel(DwArgs p)
{
    static_assert(!(CP && RG), "ragged widths: padded stacks only");
    constexpr int NTI = 4, CI = 128, CHS = 48, PLB = CI * CHS, STG = 3 * PLB;   // bytes: channel stride, plane, stage
    constexpr int CB = 32 * NW;              // channels of Cout per workgroup
    constexpr int RPT = 32 / NW;             // x rows per thread and block (the 64 NW threads share 16 rows x 128 channels): 4 or 8
    constexpr int NX4 = RPT / 4;             // ... as float4 groups
    __shared__ __attribute__((aligned(16))) char xs_lds[2 * STG];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hi = lane >> 5, l31 = lane & 31;
    const int co = RG ? min((int)blockIdx.y * CB + wave * 32 + l31, p.Cout - 1) : blockIdx.y * CB + wave * 32 + l31;        // this lane's dY channel
    const DySrc &d = p.dy.d;
    const int i0 = RG ? (int)blockIdx.z * CI : 0;            // first input channel of this workgroup
    const uint32_t LDX = RG ? (uint32_t)p.x.ldx : (uint32_t)CI;   // row stride of x (floats)
    int64_t rows_all = p.M, rpc = p.rows_per_chunk;
    if constexpr (CP) {      // one chunk per workgroup column of the grid, sized from the device-side row count (a multiple of 128)
        rows_all = __builtin_amdgcn_readfirstlane(*d.rows_dev);
        rpc = (((rows_all + gridDim.x - 1) / gridDim.x) + 15) & ~(int64_t)15;
    }
    const int64_t mbeg = (int64_t)blockIdx.x * rpc;
    const int64_t mend = min(rows_all, mbeg + rpc);
    const int n_kb = mbeg < mend ? (int)((mend - mbeg) >> 4) : 0;       // (whole 16-row blocks only: host-checked)
    const int Cout = p.Cout;

    const float ksc = d.scale[co], ksh = d.shift[co], kmu = d.mean[co];
    const float kA = ksc * d.c1[co], kB = ksc * d.c2[co] * d.invstd[co];
    // x producer role: channel xc, rows RPT xq .. RPT xq + RPT - 1 of the block
    const int xc = tid & 127, xq = tid >> 7;
    const int xg = RG ? min(i0 + xc, p.Cin - 1) : xc;        // ... as a channel of the input
    const float xsc = p.x.sc[xg], xsh = p.x.sh[xg];

    floatx16 acc[NTI];
#pragma unroll
    for (int b = 0; b < NTI; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;

    struct Raw { float y[8], z[8]; int am; float4 w0, w1; int g; };
    struct RawX { float x[RPT]; };
    // addresses: wave-uniform base + a 32-bit byte offset per lane (host-checked: M * Cout * 4 < 2^32 and whole 16-row blocks only), so a
    // load costs one 32-bit add instead of a 64-bit multiply-add and a row clamp -- address arithmetic was a third of the loop's VALU work
    auto ldg = [](const float *base, uint32_t byte_off) { return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + byte_off); };
    auto ldgi = [](const int *base, uint32_t byte_off) { return *reinterpret_cast<const int *>(reinterpret_cast<const char *>(base) + byte_off); };
    const uint32_t ystride = (uint32_t)Cout * 4u;
    const uint32_t oy_lane = (uint32_t)((mbeg + 8 * hi) * Cout + co) * 4u;        // row (mbeg + 8 hi) of this lane's channel
    const uint32_t ox_lane = (uint32_t)((mbeg + RPT * xq) * LDX + xg) * 4u;
    const uint32_t ow_lane = (uint32_t)(mbeg + 8 * hi) * 4u;                      // CP: weights of this lane's 8 rows
    const uint32_t og_lane = (uint32_t)(((mbeg >> 3) + hi)) * 4u;                 // CP: group of this lane's 8-row segment
    // CP, DY_MAX: the group of block kb's segment, fetched one ring round ahead of the block itself (its gout / argmax addresses depend on it)
    auto fetch_g = [&](int kb, Raw &w) {
        if constexpr (CP && DYMODE == A_DY_MAX) w.g = ldgi(d.seg_grp, og_lane + (uint32_t)kb * 8u);
    };
    auto fetch = [&](int kb, Raw &w) {
        const uint32_t o0 = oy_lane + (uint32_t)kb * 16u * ystride;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            w.y[j] = ldg(d.y, o0 + (uint32_t)j * ystride);
            if (DYMODE == A_DY_DENSE) w.z[j] = ldg(d.dz, o0 + (uint32_t)j * ystride);
        }
        if (DYMODE == A_DY_MAX) {
            uint32_t g;
            if constexpr (CP) g = (uint32_t)w.g;
            else g = fdiv((uint32_t)mbeg + 16u * (uint32_t)kb, d.divK);      // (no 64-bit division: a branchy call sequence between the loads)
            const uint32_t og = (g * (uint32_t)Cout + (uint32_t)co) * 4u;
            w.z[0] = ldg(d.gout, og);
            w.am = ldgi(d.argmax, og);
        }
        if constexpr (CP) {
            const char *wb = reinterpret_cast<const char *>(d.wrow) + ow_lane + (uint32_t)kb * 64u;
            w.w0 = *reinterpret_cast<const float4 *>(wb);
            w.w1 = *reinterpret_cast<const float4 *>(wb + 16);
        }
    };
    auto wof = [](const Raw &w, int j) -> float {
        const float a[8] = {w.w0.x, w.w0.y, w.w0.z, w.w0.w, w.w1.x, w.w1.y, w.w1.z, w.w1.w};
        return a[j];
    };
    auto fetch_x = [&](int kb, RawX &w) {
        const uint32_t o0 = ox_lane + (uint32_t)kb * (16u * LDX * 4u);
#pragma unroll
        for (int j = 0; j < RPT; ++j) w.x[j] = ldg(p.x.x, o0 + (uint32_t)j * (LDX * 4u));
    };
    auto stage_x = [&](int kb, const RawX &w, char *stg) {   // this thread's RPT rows of channel xc -> three (2 RPT)-byte plane pieces
        (void)kb;
#pragma unroll
        for (int h = 0; h < NX4; ++h) {
            float4 v;
            v.x = fmaxf(fmaf(xsc, w.x[4 * h + 0], xsh), 0.f); v.y = fmaxf(fmaf(xsc, w.x[4 * h + 1], xsh), 0.f);
            v.z = fmaxf(fmaf(xsc, w.x[4 * h + 2], xsh), 0.f); v.w = fmaxf(fmaf(xsc, w.x[4 * h + 3], xsh), 0.f);
            uint2 q0, q1, q2;
            split3(v, q0, q1, q2);
            char *dst = stg + xc * CHS + xq * (2 * RPT) + 8 * h;
            *reinterpret_cast<uint2 *>(dst) = q0;
            *reinterpret_cast<uint2 *>(dst + PLB) = q1;
            *reinterpret_cast<uint2 *>(dst + 2 * PLB) = q2;
        }
    };
    // prep: dY of block kb (this lane's channel, its 8 rows) -> the three bf16 planes of the MFMA's row operand
    auto prep = [&](int kb, const Raw &w, bf16x8 (&pa)[3]) {
        int kin0 = 0;
        if (DYMODE == A_DY_MAX) {
            const uint32_t b0 = (uint32_t)mbeg + 16u * (uint32_t)kb;
            kin0 = CP ? (int)b0 + 8 * hi : (int)(b0 - fdiv(b0, d.divK) * (uint32_t)d.K) + 8 * hi;
        }
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float y = w.y[j];
            float dz;
            if (DYMODE == A_DY_DENSE) dz = w.z[j];
            else dz = (w.am == kin0 + j) ? w.z[0] : 0.f;
            const float z = fmaf(ksc, y, ksh);
            const float pp = z > 0.f ? dz : 0.f;
            if constexpr (CP) v[j] = fmaf(-wof(w, j), fmaf(kB, y - kmu, kA), ksc * pp);
            else v[j] = fmaf(ksc, pp, -fmaf(kB, y - kmu, kA));
        }
        uint2 a0, a1, a2, b0, b1, b2;
        split3(make_float4(v[0], v[1], v[2], v[3]), a0, a1, a2);
        split3(make_float4(v[4], v[5], v[6], v[7]), b0, b1, b2);
        pa[0] = __builtin_bit_cast(bf16x8, make_uint4(a0.x, a0.y, b0.x, b0.y));
        pa[1] = __builtin_bit_cast(bf16x8, make_uint4(a1.x, a1.y, b1.x, b1.y));
        pa[2] = __builtin_bit_cast(bf16x8, make_uint4(a2.x, a2.y, b2.x, b2.y));
    };
    // mma: the staged x planes of a block against this wave's dY planes
    auto mma = [&](const bf16x8 (&pa)[3], const char *stg) {
        bf16x8 pb[NTI][3];
#pragma unroll
        for (int b = 0; b < NTI; ++b)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                pb[b][pl] = *reinterpret_cast<const bf16x8 *>(stg + pl * PLB + (32 * b + l31) * CHS + hi * 16);
        constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};   // smallest terms first
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int b = 0; b < NTI; ++b)
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[PA[t]], pb[b][PB[t]], acc[b], 0, 0, 0);
    };

    // every wave walks ALL blocks of the chunk (it owns channels, not rows).  The HBM round trip (~2 us) is several blocks long (a block
    // is ~0.6 us of issue), so the raw operands sit in a ring of PF register buffers: iteration kb puts dY and x of block kb + PF - 1 in
    // flight, stages x of block kb + 1 (fetched PF - 2 iterations ago) into the other LDS stage, computes block kb, one barrier.
    // one block of MFMAs (this wave's dY planes pa against the staged x planes) with the NEXT block's operand work dealt out between them:
    // after each of the 24 MFMAs one slice of VALU (fenced with scheduling barriers: the scheduler's own order puts all VALU behind the
    // MFMAs) -- 8 slices transform dY, 5 split it into planes, 4 transform / split / store this thread's share of x
    auto weave = [&](const bf16x8 (&pc)[3], const char *stg, int kbn, const Raw &w, bf16x8 (&pn)[3], const RawX &wx, char *stgn) {
        bf16x8 pb[NTI][3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)          // (plane-major: the first four MFMAs need the leading planes only, the rest is still in flight)
#pragma unroll
            for (int b = 0; b < NTI; ++b)
                pb[b][pl] = *reinterpret_cast<const bf16x8 *>(stg + pl * PLB + (32 * b + l31) * CHS + hi * 16);
        int kin0 = 0;
        if (DYMODE == A_DY_MAX) {
            const uint32_t b0 = (uint32_t)mbeg + 16u * (uint32_t)kbn;
            kin0 = CP ? (int)b0 + 8 * hi : (int)(b0 - fdiv(b0, d.divK) * (uint32_t)d.K) + 8 * hi;
        }
        float v[8];
        float4 ra, rb, rx;
        uint2 a0, a1, a2, b0, b1, b2, q0, q1, q2;
        // (pure arithmetic is sunk to its first use at the IR level, i.e. behind all 24 MFMAs, whatever the scheduling barriers say: an empty
        // volatile asm on a slice's results pins the slice where it is written)
        auto pin = [](float &x) { asm volatile("" : "+v"(x)); };
        auto pin4 = [&](float4 &r) { pin(r.x); pin(r.y); pin(r.z); pin(r.w); };
        auto pinu = [](uint2 &u) { asm volatile("" : "+v"(u.x), "+v"(u.y)); };
        auto level = [](float4 &r, uint2 &pl) {      // one level of the exact split: the leading bf16 of each value, and what is left
            pl.x = pack_bf16x2(r.x, r.y); pl.y = pack_bf16x2(r.z, r.w);
            r.x -= bf16_lo(pl.x); r.y -= bf16_hi(pl.x); r.z -= bf16_lo(pl.y); r.w -= bf16_hi(pl.y);
        };
        constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};   // smallest terms first
        dw_sfor<0, 24>([&](auto g_) {
            constexpr int g = decltype(g_)::value, t = g / NTI, bb = g % NTI;
            acc[bb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pc[PA[t]], pb[bb][PB[t]], acc[bb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (g < 8) {
                const float y = w.y[g];
                float dz;
                if (DYMODE == A_DY_DENSE) dz = w.z[g];
                else dz = (w.am == kin0 + g) ? w.z[0] : 0.f;
                const float z = fmaf(ksc, y, ksh);
                const float pp = z > 0.f ? dz : 0.f;
                if constexpr (CP) v[g] = fmaf(-wof(w, g), fmaf(kB, y - kmu, kA), ksc * pp);
                else v[g] = fmaf(ksc, pp, -fmaf(kB, y - kmu, kA));
                pin(v[g]);
            } else if constexpr (g == 8) { ra = make_float4(v[0], v[1], v[2], v[3]); level(ra, a0); pin4(ra); pinu(a0); }
            else if constexpr (g == 9) { level(ra, a1); pin4(ra); pinu(a1); }
            else if constexpr (g == 10) {
                a2.x = pack_bf16x2(ra.x, ra.y); a2.y = pack_bf16x2(ra.z, ra.w); rb = make_float4(v[4], v[5], v[6], v[7]); level(rb, b0);
                pinu(a2); pin4(rb); pinu(b0);
            } else if constexpr (g == 11) { level(rb, b1); pin4(rb); pinu(b1); }
            else if constexpr (g == 12) {
                b2.x = pack_bf16x2(rb.x, rb.y); b2.y = pack_bf16x2(rb.z, rb.w);
                pinu(b2);
                pn[0] = __builtin_bit_cast(bf16x8, make_uint4(a0.x, a0.y, b0.x, b0.y));
                pn[1] = __builtin_bit_cast(bf16x8, make_uint4(a1.x, a1.y, b1.x, b1.y));
                pn[2] = __builtin_bit_cast(bf16x8, make_uint4(a2.x, a2.y, b2.x, b2.y));
            } else if constexpr (g >= 13 && g < 13 + 4 * NX4) {          // this thread's share of x: four slices per float4 group
                constexpr int h = (g - 13) / 4, step = (g - 13) % 4;
                if constexpr (step == 0) {
                    rx.x = fmaxf(fmaf(xsc, wx.x[4 * h + 0], xsh), 0.f); rx.y = fmaxf(fmaf(xsc, wx.x[4 * h + 1], xsh), 0.f);
                    rx.z = fmaxf(fmaf(xsc, wx.x[4 * h + 2], xsh), 0.f); rx.w = fmaxf(fmaf(xsc, wx.x[4 * h + 3], xsh), 0.f);
                    pin4(rx);
                } else if constexpr (step == 1) { level(rx, q0); pin4(rx); pinu(q0); }
                else if constexpr (step == 2) { level(rx, q1); pin4(rx); pinu(q1); }
                else {
                    q2.x = pack_bf16x2(rx.x, rx.y); q2.y = pack_bf16x2(rx.z, rx.w);
                    char *dst = stgn + xc * CHS + xq * (2 * RPT) + 8 * h;
                    *reinterpret_cast<uint2 *>(dst) = q0;
                    *reinterpret_cast<uint2 *>(dst + PLB) = q1;
                    *reinterpret_cast<uint2 *>(dst + 2 * PLB) = q2;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    // Inside a wave the NEXT block's transform (VALU) is woven into the CURRENT block's MFMAs: the matrix pipe runs an MFMA for ~8 issue
    // slots, and all 8 waves of the workgroup are in the same phase behind the per-block barrier, so without the weave VALU time and matrix
    // time add up on every SIMD.
    constexpr int PF = 3;
    Raw dr[PF];
    RawX xr[PF];
    bf16x8 pa[PF][3];       // (dY planes: a ring like the raw buffers -- block kb's planes sit in slot kb % PF whatever the parity of PF)
    const int last = n_kb - 1;
    if (n_kb > 0) {
        // (no branch sits around a load anywhere below: block indices past the chunk are clamped to its last block -- re-read, never used --
        // so the compiler keeps COUNTED waits (vmcnt(n > 0)) in the steady loop instead of draining the ring at every control-flow merge)
#pragma unroll
        for (int i = 0; i < PF; ++i) fetch_g(min(i, last), dr[i]);
#pragma unroll
        for (int i = 0; i < PF - 1; ++i) { fetch(min(i, last), dr[i]); fetch_x(min(i, last), xr[i]); }
        stage_x(0, xr[0], xs_lds);
        prep(0, dr[0], pa[0]);
        lds_barrier();
    }
    const int n_main = n_kb - n_kb % PF;
    for (int kb0 = 0; kb0 < n_main; kb0 += PF) {
#pragma unroll
        for (int i = 0; i < PF; ++i) {          // (the ring indices are compile-time: the buffers stay in registers)
            const int kb = kb0 + i;
            constexpr int NXT = PF - 1;
            // (CP: dr[i]'s block is already in its planes pa[i], so the slot is free for the group of block kb + PF -- one ring round ahead of
            // that block's own loads; issued BEFORE this iteration's loads, so the wait for it next iteration leaves them in flight)
            if constexpr (CP && DYMODE == A_DY_MAX) {
                const int gnew = ldgi(d.seg_grp, og_lane + (uint32_t)min(kb + PF, last) * 8u);
                fetch(min(kb + NXT, last), dr[(i + NXT) % PF]);
                dr[i].g = gnew;
            } else
            fetch(min(kb + NXT, last), dr[(i + NXT) % PF]);
            fetch_x(min(kb + NXT, last), xr[(i + NXT) % PF]);
            __builtin_amdgcn_sched_barrier(0);
            weave(pa[i], xs_lds + (kb & 1) * STG, min(kb + 1, last), dr[(i + 1) % PF], pa[(i + 1) % PF], xr[(i + 1) % PF],
                  xs_lds + ((kb + 1) & 1) * STG);
            lds_barrier();
        }
    }
    // the last n_kb % PF blocks: their operands are already in the ring (slots 0 .. n_kb % PF - 1 of this round)
#pragma unroll
    for (int i = 0; i < PF - 1; ++i) {
        const int kb = n_main + i;
        if (kb < n_kb) {
            mma(pa[i], xs_lds + (kb & 1) * STG);
            if (kb + 1 < n_kb) {
                prep(kb + 1, dr[(i + 1) % PF], pa[(i + 1) % PF]);
                stage_x(kb + 1, xr[(i + 1) % PF], xs_lds + ((kb + 1) & 1) * STG);
            }
            lds_barrier();
        }
    }

    // ---- this wave's 32 rows of the partial: row (cout) = (r & 3) + 8 (r >> 2) + 4 half, col (cin) = lane & 31
    float *out = p.dw_partial + (int64_t)blockIdx.x * p.part_ld;
    const int cbase = blockIdx.y * CB + wave * 32;
#pragma unroll
    for (int b = 0; b < NTI; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = cbase + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if constexpr (RG) {
                if (c < p.Cout && i0 + 32 * b + l31 < p.Cin) out[(int64_t)c * p.Cin + i0 + 32 * b + l31] = acc[b][r];
            } else
            out[(int64_t)c * CI + 32 * b + l31] = acc[b][r];
        }
    if (p.db_partial && tid < CB && (!RG || (blockIdx.z == 0 && (int)blockIdx.y * CB + tid < p.Cout))) p.db_partial[(int64_t)blockIdx.x * p.part_ld + blockIdx.y * CB + tid] = 0.f;
}

static unsigned long long *g_dw_dbg = nullptr;
// ---------------------------------------------------------------------------------------------------------------------
// dW of a max-pooled layer whose [M, Cout] output was never stored (papc_mlp_max_nostore_ok; 64 -> 128 channels).  With a = relu(bn(x))
// the layer's input rows, P' the max-backward values (sc * gout at the winning row where the pooled output is alive, psel / argmax:
// [G, Cout]) and z - mean = W (a - abar):
//     dW = sum_m dz[m]^T a[m] = P'^T A  -  (sc c1) (x) S  -  diag(e) W (A^T A - S S^T / M),      S = sum_m a[m],  e = sc c2 invstd,
// so one pass over the INPUT rows yields everything: T = P'^T A ([Cout, 64], the one-hot expansion of (psel, argmax) as the MFMA's row
// operand), the Gram matrix A^T A ([64, 64], the transformed input against itself) and the column sums S.  Same streaming scheme as
// dw_rows_kernel; a workgroup owns a row chunk, 64 channels of Cout and 32 rows of the Gram matrix (6 accumulator tiles per wave).
// The pair of workgroups of one chunk is 8 apart in the flat id: same XCD, so the second reader of the rows finds them in its L2.
struct DwMaxArgs {
    const float *x, *xsc, *xsh, *psel;
    const int *argmax;
    float *partial;      // [chunks][part_ld]: T [Cout, 64] | G [64, 64] | S [64]
    int64_t M, part_ld;
    int K, Cout, rows_per_chunk, n_chunks;
};

__global__ __launch_bounds__(512, 2) void dw_rows_max_kernel(DwMaxArgs p)
{
    constexpr int NT = 3, NTI = 2, CI = 64;
    __shared__ float slab[4][NT * NTI * 16][64];   // 96 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hi = lane >> 5, l31 = lane & 31;
    const int yb = (blockIdx.x >> 3) & 1, chunk = (int)(((blockIdx.x >> 4) << 3) | (blockIdx.x & 7));
    const int o0 = 64 * yb, Cout = p.Cout;
    const int64_t mbeg = (int64_t)chunk * p.rows_per_chunk;
    const int64_t mend = min(p.M, mbeg + p.rows_per_chunk);
    const int n_kb = mbeg < mend ? (int)((mend - mbeg) >> 4) : 0;       // (M, rows_per_chunk % 16 == 0: whole blocks only)
    const int kshift = 31 - __clz(p.K);

    float xs[NTI], xh[NTI], xsum[NTI];
#pragma unroll
    for (int b = 0; b < NTI; ++b) { xs[b] = p.xsc[32 * b + l31]; xh[b] = p.xsh[32 * b + l31]; xsum[b] = 0.f; }
    floatx16 acc[NT][NTI];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < NTI; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    struct Raw { float x[NTI][8]; float ps[2]; int am[2]; };
    // (32-bit byte offsets from wave-uniform bases -- M * 128 channels * 4 bytes < 2^32 is part of papc_mlp_max_nostore_ok's M < 2^31 / 4:
    // host-checked -- instead of 64-bit index arithmetic per load)
    auto ldg = [](const float *base, uint32_t byte_off) { return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + byte_off); };
    auto ldgi = [](const int *base, uint32_t byte_off) { return *reinterpret_cast<const int *>(reinterpret_cast<const char *>(base) + byte_off); };
    const uint32_t ox_lane = (uint32_t)((mbeg + 8 * hi) * CI + l31) * 4u;
    const uint32_t og_lane = (uint32_t)(o0 + l31) * 4u;
    auto fetch = [&](int kb, Raw &w) {
        const uint32_t o = ox_lane + (uint32_t)kb * (16u * CI * 4u);
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int b = 0; b < NTI; ++b) w.x[b][j] = ldg(p.x, o + (uint32_t)(j * CI + 32 * b) * 4u);
        const uint32_t g = ((uint32_t)mbeg + 16u * (uint32_t)kb) >> kshift;           // K % 16 == 0: the block's 16 rows share one group (K a power of two)
        const uint32_t og = g * (uint32_t)Cout * 4u + og_lane;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            w.ps[a] = ldg(p.psel, og + 128u * a);
            w.am[a] = ldgi(p.argmax, og + 128u * a);
        }
    };
    auto compute = [&](int kb, const Raw &w) {
        const int kin0 = (int)(((uint32_t)mbeg + 16u * (uint32_t)kb) & (uint32_t)(p.K - 1)) + 8 * hi;
        bf16x8 pa[2][3], pb[NTI][3];
#pragma unroll
        for (int b = 0; b < NTI; ++b) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { v[j] = fmaxf(fmaf(xs[b], w.x[b][j], xh[b]), 0.f); xsum[b] += v[j]; }
            uint2 a0, a1, a2, b0, b1, b2;
            split3(make_float4(v[0], v[1], v[2], v[3]), a0, a1, a2);
            split3(make_float4(v[4], v[5], v[6], v[7]), b0, b1, b2);
            pb[b][0] = __builtin_bit_cast(bf16x8, make_uint4(a0.x, a0.y, b0.x, b0.y));
            pb[b][1] = __builtin_bit_cast(bf16x8, make_uint4(a1.x, a1.y, b1.x, b1.y));
            pb[b][2] = __builtin_bit_cast(bf16x8, make_uint4(a2.x, a2.y, b2.x, b2.y));
        }
        {   // the one-hot rows: the value's three planes, parked at row (argmax - kin0) of this lane's eight
            uint2 q0, q1, q2;
            split3(make_float4(w.ps[0], w.ps[1], 0.f, 0.f), q0, q1, q2);   // .x = (plane of ps[0]) | (plane of ps[1]) << 16
            const unsigned pl[3] = {q0.x, q1.x, q2.x};
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int idx = w.am[a] - kin0;
                const int dsel = (idx >= 0 && idx < 8) ? (idx >> 1) : -1;
                const int sh = (idx & 1) * 16;
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const unsigned bits = ((a ? pl[t] >> 16 : pl[t]) & 0xffffu) << sh;
                    pa[a][t] = __builtin_bit_cast(bf16x8, make_uint4(dsel == 0 ? bits : 0u, dsel == 1 ? bits : 0u, dsel == 2 ? bits : 0u, dsel == 3 ? bits : 0u));
                }
            }
        }
        bf16x8 pg[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) pg[t] = yb ? pb[1][t] : pb[0][t];
        constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};   // smallest terms first
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int b = 0; b < NTI; ++b) {
                acc[0][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[0][PA[t]], pb[b][PB[t]], acc[0][b], 0, 0, 0);
                acc[1][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[1][PA[t]], pb[b][PB[t]], acc[1][b], 0, 0, 0);
                acc[2][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pg[PA[t]], pb[b][PB[t]], acc[2][b], 0, 0, 0);
            }
    };

    // this wave's blocks: wave, wave + 8, ...  The raw operands of the next PF - 1 blocks are in flight while one is computed.  No branch
    // sits around a load (block indices past the chunk are clamped to the wave's last block: re-read, never used) and no 64-bit division
    // between them: at every control-flow merge the compiler drains its register ring to vmcnt(0) instead of counting
    constexpr int PF = 3;
    const int nb = (n_kb - wave + 7) >> 3;          // blocks of this wave
    if (nb > 0) {
        const int last = wave + 8 * (nb - 1);
        Raw rr[PF];
#pragma unroll
        for (int i = 0; i < PF - 1; ++i) fetch(min(wave + 8 * i, last), rr[i]);
        const int n_main = nb - nb % PF;
        for (int j0 = 0; j0 < n_main; j0 += PF) {
#pragma unroll
            for (int i = 0; i < PF; ++i) {
                const int kb = wave + 8 * (j0 + i);
                fetch(min(kb + 8 * (PF - 1), last), rr[(i + PF - 1) % PF]);
                __builtin_amdgcn_sched_barrier(0);      // (keep the loads ahead of the block's arithmetic and the blocks in order: the scheduler
                compute(kb, rr[i]);                     //  otherwise sinks loads behind MFMAs of the next block and waits them out at vmcnt(0))
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int i = 0; i < PF - 1; ++i)
            if (n_main + i < nb) compute(wave + 8 * (n_main + i), rr[i]);
    }

    // ---- column sums of the transformed input: (wave, half) partials -> LDS, fixed-order sum (chunk's first workgroup only)
    float *out = p.partial + (int64_t)chunk * p.part_ld;
    if (yb == 0) {
        float *sred = &slab[0][0][0];
#pragma unroll
        for (int b = 0; b < NTI; ++b) sred[(wave * 2 + hi) * CI + 32 * b + l31] = xsum[b];
        __syncthreads();
        if (tid < CI) {
            float sacc = 0.f;
            for (int q = 0; q < 16; ++q) sacc += sred[q * CI + tid];
            out[(int64_t)(Cout + CI) * CI + tid] = sacc;
        }
        __syncthreads();
    }
    // ---- fold the 8 waves' tiles (fixed order, as dw_rows_kernel)
#pragma unroll
    for (int half = 4; half >= 1; half >>= 1) {
        if (wave >= half && wave < 2 * half) {
#pragma unroll
            for (int a = 0; a < NT; ++a)
#pragma unroll
                for (int b = 0; b < NTI; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) slab[wave - half][(a * NTI + b) * 16 + r][lane] = acc[a][b][r];
        }
        __syncthreads();
        if (wave < half) {
#pragma unroll
            for (int a = 0; a < NT; ++a)
#pragma unroll
                for (int b = 0; b < NTI; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[a][b][r] += slab[wave][(a * NTI + b) * 16 + r][lane];
        }
        __syncthreads();
    }
    if (wave == 0) {   // C/D layout: row = (r & 3) + 8 (r >> 2) + 4 half, col (cin) = lane & 31
#pragma unroll
        for (int a = 0; a < NT; ++a)
#pragma unroll
            for (int b = 0; b < NTI; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rr = (r & 3) + 8 * (r >> 2) + 4 * hi;
                    const int row = a < 2 ? o0 + 32 * a + rr : Cout + 32 * yb + rr;
                    out[(int64_t)row * CI + 32 * b + l31] = acc[a][b][r];
                }
    }
}

// chunk partials -> double sums: one wave per 16 elements, four quarter-ranges of the chunks in parallel (loads unrolled: a serial chain of
// 128 L2 round trips otherwise), quarters added in fixed order.  No LDS and one wave, so that the launch finds room beside another
// stream's persistent GEMM (config 3: the 256-thread form waited 135 us on average for a CU to drain)
__global__ __launch_bounds__(64) void dw_max_fold_kernel(const float *__restrict__ partial, int n_chunks, int64_t part_ld, int n, double *__restrict__ out)
{
    const int l = threadIdx.x & 15, q = threadIdx.x >> 4;
    const int e = blockIdx.x * 16 + l;
    const int per = (n_chunks + 3) / 4, c0 = q * per, c1 = min(n_chunks, c0 + per);
    double s = 0.0;
    if (e < n) {
        int c = c0;
        for (; c + 8 <= c1; c += 8) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = partial[(int64_t)(c + j) * part_ld + e];
#pragma unroll
            for (int j = 0; j < 8; ++j) s += (double)v[j];
        }
        for (; c < c1; ++c) s += (double)partial[(int64_t)c * part_ld + e];
    }
    const double s1 = __shfl(s, l + 16), s2 = __shfl(s, l + 32), s3 = __shfl(s, l + 48);
    if (q == 0 && e < n) out[e] = ((s + s1) + s2) + s3;
}

// dW[c, i] = T[c, i] - sc_c c1_c S_i - e_c sum_j W[c, j] (G[j, i] - S_j S_i / M); four rows of dW per workgroup
__global__ __launch_bounds__(256) void dw_max_finalize_kernel(const double *__restrict__ fold, const float *__restrict__ w, const float *__restrict__ e,
                                                              const float *__restrict__ scale, const float *__restrict__ c1, double inv_m, int Cout,
                                                              float *__restrict__ dw, int accumulate)
{
    constexpr int CI = 64;
    __shared__ double gc[CI][CI];
    const int tid = threadIdx.x;
    const double *G = fold + (int64_t)Cout * CI, *S = G + CI * CI;
    for (int q = tid; q < CI * CI; q += 256) {
        const int j = q >> 6, i = q & 63;
        gc[j][i] = G[q] - S[j] * S[i] * inv_m;
    }
    __syncthreads();
    const int i = tid & 63, c = 4 * blockIdx.x + (tid >> 6);
    if (c >= Cout) return;
    double dot = 0.0;
    for (int j = 0; j < CI; ++j) dot += (double)w[(int64_t)c * CI + j] * gc[j][i];
    const double v = fold[(int64_t)c * CI + i] - (double)scale[c] * (double)c1[c] * S[i] - (double)e[c] * dot;
    float *o = dw + (int64_t)c * CI + i;
    *o = accumulate ? *o + (float)v : (float)v;
}

static void dw_dbg_report(const DwArgs &p, int xm, int dm)
{
    hipDeviceSynchronize();
    unsigned long long h[32];
    hipMemcpy(h, g_dw_dbg, sizeof(h), hipMemcpyDeviceToHost);
    const double n = (double)h[4];
    fprintf(stderr, "[dw dbg] x %d dy %d M %lld Cout %d Cin %d: slots %.0f | consumer cyc/slot: reads+mfma-issue %.0f barrier %.0f | producer grp0 per ACTIVE slot: wait %.0f finish %.0f issue %.0f ; barrier/slot %.0f\n",
            xm, dm, (long long)p.M, p.Cout, p.Cin, n, h[0] / n, h[1] / n, h[8] / n, h[9] / n, h[10] / n, h[11] / n);
}

// row-streaming kernel (dw_rows_kernel): which layers take it (PAPC_DW_ROWS=0: none).  Measured on config 2's SA1 with one workgroup
// per CU (papc_mlp_bwd_dw_chunk_hint): 64 -> 64 dense 99.5 -> 81 us; 64 -> 128 under the max 119 -> ~110 us although its two 64-channel
// blocks of Cout transform the input twice; dW family 0.85 -> 0.82 ms/step.
static bool dw_rowsx_ragged(int Cin, int Cout)      // the ragged flavours (RG): the 196-channel layer pair [128, 196, 256], the 96-channel pair [64, 96, 128]
{
    return (Cin == 128 && Cout == 196) || (Cin == 196 && Cout == 256) || (Cin == 96 && Cout == 128) || (Cin == 64 && Cout == 96);
}
static bool dw_rowsx_eligible(int Cin, int Cout, bool dense, int K)   // dw_rowsx_kernel: 128-channel input, 256-channel blocks of Cout
{
    return knob(KNOB_DW_ROWSX) != 0 && ((Cin == 128 && Cout % 128 == 0) || dw_rowsx_ragged(Cin, Cout)) && !dw_f32_exact() && (dense || (K >= 16 && K % 16 == 0));   // (+ rowsx_rows_ok at the launch)
}
static bool dw_rows_eligible(int Cin, int Cout, bool dense, int K)
{
    // (64 x 64 blocks of the output: every block transforms its operands again, so at most four of them)
    return knob(KNOB_DW_ROWS) != 0 && Cin % 64 == 0 && Cout % 64 == 0 && (Cin / 64) * (Cout / 64) <= knob(KNOB_DW_ROWS_BLOCKS) && !dw_f32_exact() && (dense || (K >= 16 && K % 16 == 0));
}

template <int XMODE, int DYMODE, bool VEC>
static int launch_dw_v(const DwArgs &p_in, hipStream_t st)
{
    DwArgs p = p_in;
    const int dbg_on = knob(KNOB_DW_DBG);
    if (dbg_on) {
        if (!g_dw_dbg) hipMalloc(&g_dw_dbg, 32 * sizeof(unsigned long long));
        hipMemsetAsync(g_dw_dbg, 0, 32 * sizeof(unsigned long long), st);
        p.dbg = g_dw_dbg;
    }
    const bool wide = p.TIp == DW_TI_WIDE;
    dim3 grid((unsigned)cdiv(p.M, p.rows_per_chunk), (unsigned)cdiv(p.Cout, DW_T), (unsigned)cdiv(p.Cin, wide ? DW_TI_WIDE : DW_T));
    const int to = p.TOp / 32, ti = p.TIp / 32;  // 32-wide tiles per workgroup (upper bound)
    const bool off32 = (int64_t)p.rows_per_chunk * std::max<int64_t>(p.Cout, XMODE == A_GROUP ? 1 : p.x.ldx) * 4 < (1ll << 31);
    const bool k4 = DYMODE != A_DY_MAX || p.dy.d.K % 4 == 0;
    // (the kernel addresses with 32-bit byte offsets and walks whole 16-row blocks)
    const bool rowsx_rows_ok = p.M % 16 == 0 && p.M * (int64_t)p.Cout * 4 < (1ll << 32);
    if (p.dy.d.wrow) {     // compacted stack: dw_rowsx_kernel's CP flavours or nothing
        const bool ok = VEC && XMODE == A_BNRELU && rowsx_rows_ok && p.Cin == 128 && p.Cout % 128 == 0 && p.x.ldx == p.Cin && !dw_f32_exact() &&
                        p.dy.d.rows_dev && (DYMODE != A_DY_MAX || p.dy.d.seg_grp) && p.M % 128 == 0;
        if (!ok) {
            set_error("papc_mlp_bwd_dw_f32: a compacted dY source needs a 128-channel BN+ReLU input and Cout %% 128 == 0 (got Cin=%d Cout=%d)", p.Cin, p.Cout);
            return PAPC_E_UNSUPPORTED;
        }
        if (p.Cout % 256 == 0) {
            dim3 g2((unsigned)cdiv(p.M, p.rows_per_chunk), (unsigned)(p.Cout / 256));
            hipLaunchKernelGGL((dw_rowsx_kernel<DYMODE, 8, true>), g2, dim3(512), 0, st, p);
        } else {
            dim3 g2((unsigned)cdiv(p.M, p.rows_per_chunk), (unsigned)(p.Cout / 128));
            hipLaunchKernelGGL((dw_rowsx_kernel<DYMODE, 4, true>), g2, dim3(256), 0, st, p);
        }
        return check_launch("papc_mlp_bwd_dw_f32 (compacted)");
    }
    if (VEC && XMODE == A_BNRELU && rowsx_rows_ok && dw_rowsx_eligible(p.Cin, p.Cout, DYMODE == A_DY_DENSE, p.dy.d.K) && p.x.ldx == p.Cin && p.rows_per_chunk % 16 == 0 &&
        (!dw_rowsx_ragged(p.Cin, p.Cout) || p.M * (int64_t)p.Cin * 4 < (1ll << 32))) {      // (ragged: 32-bit byte offsets into x as well)
        if (dw_rowsx_ragged(p.Cin, p.Cout) && p.Cout <= 128) {     // one 128-channel block of Cout: four waves, two workgroups per CU
            dim3 g2((unsigned)cdiv(p.M, p.rows_per_chunk), 1u, (unsigned)cdiv(p.Cin, 128));
            hipLaunchKernelGGL((dw_rowsx_kernel<DYMODE, 4, false, true>), g2, dim3(256), 0, st, p);
        } else if (dw_rowsx_ragged(p.Cin, p.Cout)) {
            dim3 g2((unsigned)cdiv(p.M, p.rows_per_chunk), (unsigned)cdiv(p.Cout, 256), (unsigned)cdiv(p.Cin, 128));
            hipLaunchKernelGGL((dw_rowsx_kernel<DYMODE, 8, false, true>), g2, dim3(512), 0, st, p);
        } else if (p.Cout % 256 == 0) {
            dim3 g2((unsigned)cdiv(p.M, p.rows_per_chunk), (unsigned)(p.Cout / 256));
            hipLaunchKernelGGL((dw_rowsx_kernel<DYMODE, 8>), g2, dim3(512), 0, st, p);
        } else {        // 128-channel blocks: four waves per workgroup, two workgroups per CU
            dim3 g2((unsigned)cdiv(p.M, p.rows_per_chunk), (unsigned)(p.Cout / 128));
            hipLaunchKernelGGL((dw_rowsx_kernel<DYMODE, 4>), g2, dim3(256), 0, st, p);
        }
        return check_launch("papc_mlp_bwd_dw_f32");
    }
    const bool rows_rows_ok = p.M % 16 == 0 && p.M * (int64_t)std::max(p.Cout, p.Cin) * 4 < (1ll << 32);
    if (VEC && XMODE == A_BNRELU && rows_rows_ok && dw_rows_eligible(p.Cin, p.Cout, DYMODE == A_DY_DENSE, p.dy.d.K) && p.x.ldx == p.Cin && p.rows_per_chunk % 16 == 0) {
        // narrow input (64 channels): row-streaming kernel, every thread loads + transforms + multiplies
        dim3 g2((unsigned)cdiv(p.M, p.rows_per_chunk), (unsigned)(p.Cout / 64), (unsigned)(p.Cin / 64));
        hipLaunchKernelGGL((dw_rows_kernel<DYMODE>), g2, dim3(512), 0, st, p);
        return check_launch("papc_mlp_bwd_dw_f32");
    }
    if (VEC && XMODE != A_GROUP && !wide && to >= 2 && ti >= 2 && off32 && k4 && !dw_f32_exact()) {
        // dense layers with >= 64-wide tiles: wave-specialised bf16x3 kernel
        if (to > 2 && ti > 2) hipLaunchKernelGGL((dw_ws_kernel<XMODE, DYMODE, 2, 2>), grid, dim3(768), 0, st, p);
        else if (to > 2) hipLaunchKernelGGL((dw_ws_kernel<XMODE, DYMODE, 2, 1>), grid, dim3(768), 0, st, p);
        else if (ti > 2) hipLaunchKernelGGL((dw_ws_kernel<XMODE, DYMODE, 1, 2>), grid, dim3(768), 0, st, p);
        else if (knob(KNOB_DW_RS64)) hipLaunchKernelGGL((dw_ws_kernel<XMODE, DYMODE, 1, 1, 64>), grid, dim3(768), 0, st, p);   // 64 x 64 tiles: 64-row stages
        else hipLaunchKernelGGL((dw_ws_kernel<XMODE, DYMODE, 1, 1>), grid, dim3(768), 0, st, p);
        if (dbg_on) dw_dbg_report(p, XMODE, DYMODE);
        return check_launch("papc_mlp_bwd_dw_f32");
    }
    if (wide) {
        if (XMODE == A_GROUP) hipLaunchKernelGGL((dw_kernel<A_GROUP, DYMODE, VEC, 4, 1, 1, 5, 5>), grid, dim3(256), 0, st, p);
    } else if (!VEC) {
        hipLaunchKernelGGL((dw_kernel<XMODE, DYMODE, false, 2, 2, 2, 2, 4>), grid, dim3(256), 0, st, p);  // ragged shapes: one generic variant
    } else if (to > 2 && ti > 2) hipLaunchKernelGGL((dw_kernel<XMODE, DYMODE, VEC, 2, 2, 2, 2, 4>), grid, dim3(256), 0, st, p);
    else if (to > 2) hipLaunchKernelGGL((dw_kernel<XMODE, DYMODE, VEC, 2, 2, 2, 1, 4>), grid, dim3(256), 0, st, p);
    else if (ti > 2) hipLaunchKernelGGL((dw_kernel<XMODE, DYMODE, VEC, 2, 2, 1, 2, 4>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((dw_kernel<XMODE, DYMODE, VEC, 2, 2, 1, 1, 4>), grid, dim3(256), 0, st, p);
    return check_launch("papc_mlp_bwd_dw_f32");
}

template <int XMODE, int DYMODE>
static int launch_dw(const DwArgs &p, bool vec, hipStream_t st)
{
    return vec ? launch_dw_v<XMODE, DYMODE, true>(p, st) : launch_dw_v<XMODE, DYMODE, false>(p, st);
}

static int pad_tile(int c) { return c <= 32 ? 32 : (c <= 64 ? 64 : 128); }

}  // namespace papc

using namespace papc;

/* rows_per_chunk the dW entry point would like for this layer, or 0 for "the caller's own rule": the row-streaming kernel keeps a whole
 * 64 x 64 tile per wave (241 registers: one workgroup per CU), so it wants ONE residency wave of workgroups -- ncu row chunks in all */
extern "C" int papc_mlp_bwd_dw_chunk_hint(int64_t M, int Cin, int Cout, int a_mode, int dz_mode, int K)
{
    if (a_mode == PAPC_A_GROUP && dz_mode == PAPC_DZ_DENSE && (Cin == 3 || Cin == 7) && knob(KNOB_DW_XYZ) && Cout % 4 == 0 && Cout <= 256 && M >= 1) {
        // the streaming reduction of a 3- / 7-input first layer (dw_xyz_kernel) is bound by the (dz, y) stream: enough workgroups for
        // four waves per SIMD (~2048), chunks of at least 256 rows
        const int64_t rpc = std::max<int64_t>(256, cdiv(cdiv(M, 2048), 64) * 64);
        return (int)std::min<int64_t>(rpc, 1 << 24);
    }
    if ((a_mode != PAPC_A_BNRELU && a_mode != PAPC_A_XYZ) || M < 1) return 0;
    const bool xk = dw_rowsx_eligible(Cin, Cout, dz_mode == PAPC_DZ_DENSE, K);
    if (!xk && !dw_rows_eligible(Cin, Cout, dz_mode == PAPC_DZ_DENSE, K)) return 0;
    hipDeviceProp_t prop;
    int dev = 0;
    static int ncu = 0;
    if (!ncu) { ncu = 256; if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ncu = prop.multiProcessorCount; }
    const int64_t want = xk ? (dw_rowsx_ragged(Cin, Cout) ? std::max<int64_t>(1, (Cout <= 128 ? 2 : 1) * ncu / (cdiv(Cout, 256) * cdiv(Cin, 128)))
                                  : Cout % 256 == 0 ? std::max<int64_t>(1, ncu / (Cout / 256)) : std::max<int64_t>(1, 2 * ncu / (Cout / 128)))
                            : std::max<int64_t>(1, ncu / ((Cout / 64) * (Cin / 64)));
    int64_t rpc = cdiv(M, want);
    rpc = std::max<int64_t>(64, cdiv(rpc, 64) * 64);
    return (int)std::min<int64_t>(rpc, 1 << 24);
}

extern "C" int papc_mlp_bwd_dw_f32(const papc_bwd_dy *dy, int a_mode, const float *x, int64_t ldx,
                                   const papc_group_src *grp, const float *bn_scale, const float *bn_shift, int64_t M,
                                   int Cin, int Cout, int rows_per_chunk, float *dw_partial, float *db_partial,
                                   int64_t part_ld, papc_stream_t stream)
{
    PAPC_REQUIRE(dw_partial, PAPC_E_INVALID, "papc_mlp_bwd_dw_f32: null dw_partial");
    PAPC_REQUIRE(M >= 1 && Cin >= 1 && Cout >= 1, PAPC_E_INVALID, "papc_mlp_bwd_dw_f32: bad sizes");
    PAPC_REQUIRE(M < (1ll << 31), PAPC_E_UNSUPPORTED, "papc_mlp_bwd_dw_f32: M=%lld >= 2^31 rows", (long long)M);
    PAPC_REQUIRE(part_ld >= (int64_t)Cout * Cin, PAPC_E_INVALID, "papc_mlp_bwd_dw_f32: part_ld=%lld < Cout*Cin", (long long)part_ld);
    PAPC_REQUIRE(rows_per_chunk >= 64 && rows_per_chunk % 64 == 0, PAPC_E_INVALID,
                 "papc_mlp_bwd_dw_f32: rows_per_chunk=%d must be a positive multiple of 64", rows_per_chunk);
    bool vdy = false;
    int rc = check_dy(dy, M, Cout, &vdy, "papc_mlp_bwd_dw_f32");
    if (rc) return rc;
    DwArgs p;
    memset(&p, 0, sizeof(p));
    rc = fill_asrc(p.x, a_mode, x, ldx, grp, bn_scale, bn_shift, Cin, "papc_mlp_bwd_dw_f32");
    if (rc) return rc;
    if (a_mode == A_XYZ) {   // recomputed first-layer activations: only the row-streaming kernel has the flavour
        const bool dense_ = dy->dz_mode == PAPC_DZ_DENSE;
        PAPC_REQUIRE(vdy && p.x.vec && dw_rows_eligible(Cin, Cout, dense_, dy->K) && rows_per_chunk % 16 == 0 && M % 16 == 0 &&
                     M * (int64_t)Cout * 4 < (1ll << 32), PAPC_E_UNSUPPORTED,
                     "papc_mlp_bwd_dw_f32: PAPC_A_XYZ is not built for Cin=%d Cout=%d (see papc_mlp_xyz_ok)", Cin, Cout);
        fill_dy(p.dy.d, dy);
        p.dy.d.C = Cout;
        p.M = M; p.Cin = Cin; p.Cout = Cout; p.rows_per_chunk = rows_per_chunk; p.dw_partial = dw_partial; p.db_partial = db_partial; p.part_ld = part_ld;
        hipStream_t st2 = as_stream(stream);
        ProfScope prof2(PAPC_K_BWD_DW, st2);
        dim3 g2((unsigned)cdiv(M, rows_per_chunk), (unsigned)(Cout / 64), (unsigned)(Cin / 64));
        if (dense_) hipLaunchKernelGGL((dw_rows_kernel<A_DY_DENSE, true>), g2, dim3(512), 0, st2, p);
        else hipLaunchKernelGGL((dw_rows_kernel<A_DY_MAX, true>), g2, dim3(512), 0, st2, p);
        return check_launch("papc_mlp_bwd_dw_f32");
    }
    fill_dy(p.dy.d, dy);
    p.dy.d.C = Cout;
    const bool vec = vdy && p.x.vec;
    p.M = M; p.Cin = Cin; p.Cout = Cout; p.rows_per_chunk = rows_per_chunk; p.dw_partial = dw_partial; p.db_partial = db_partial; p.part_ld = part_ld;
    p.xmap = (a_mode == A_GROUP) ? 1 : 0;
    p.TOp = pad_tile(Cout); p.TIp = pad_tile(Cin);
    if (a_mode == A_GROUP && Cin > DW_T && Cin <= DW_TI_WIDE && p.TOp == 128) p.TIp = DW_TI_WIDE;  // D+3 with D = 128
    p.RS = (p.TOp + p.TIp <= 128) ? 64 : 32;
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_BWD_DW, st);
    const bool dense = dy->dz_mode == PAPC_DZ_DENSE;
    const int xyz_on = knob(KNOB_DW_XYZ);
    if (xyz_on && a_mode == A_GROUP && dense && vec && ((grp->D == 0 && Cin == 3) || (grp->D == 4 && Cin == 7)) && Cout % 4 == 0 && Cout >= 4 && Cout <= 256) {
        // a first layer with 3 (coordinates) or 7 (+ four feature columns) inputs: streaming reduction instead of a 3- / 7-of-32-column MFMA tile
        if (grp->D == 0) hipLaunchKernelGGL(dw_xyz_kernel<0>, dim3((unsigned)cdiv(M, rows_per_chunk)), dim3(256), 0, st, p);
        else hipLaunchKernelGGL(dw_xyz_kernel<1>, dim3((unsigned)cdiv(M, rows_per_chunk)), dim3(256), 0, st, p);
        return check_launch("papc_mlp_bwd_dw_f32");
    }
    switch (a_mode) {
    case A_PLAIN: return dense ? launch_dw<A_PLAIN, A_DY_DENSE>(p, vec, st) : launch_dw<A_PLAIN, A_DY_MAX>(p, vec, st);
    case A_BNRELU: return dense ? launch_dw<A_BNRELU, A_DY_DENSE>(p, vec, st) : launch_dw<A_BNRELU, A_DY_MAX>(p, vec, st);
    default: return dense ? launch_dw<A_GROUP, A_DY_DENSE>(p, vec, st) : launch_dw<A_GROUP, A_DY_MAX>(p, vec, st);
    }
}

static int dw_max_chunks()
{
    static int n = 0;
    if (!n) {
        hipDeviceProp_t prop;
        int dev = 0, ncu = 256;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ncu = prop.multiProcessorCount;
        n = std::max(8, (ncu / 2) & ~7);     // two workgroups per chunk, one residency wave, whole groups of 8
    }
    return n;
}

/* floats of workspace papc_mlp_bwd_dw_max_f32 needs (chunk partials + their double sums) */
extern "C" int64_t papc_mlp_bwd_dw_max_ws_floats(int64_t M, int Cin, int Cout)
{
    (void)M;
    const int64_t n = (int64_t)(Cout + Cin) * Cin + Cin;
    return (int64_t)dw_max_chunks() * n + 2 * n + 2;
}

extern "C" int papc_mlp_bwd_dw_max_f32(const float *psel, const int32_t *argmax, int K, const float *x, const float *bn_scale, const float *bn_shift,
                                       const float *w, const float *e, const float *scale, const float *c1, int64_t M, int Cin, int Cout,
                                       float *workspace, float *dw, int accumulate, papc_stream_t stream)
{
    PAPC_REQUIRE(psel && argmax && x && bn_scale && bn_shift && w && e && scale && c1 && workspace && dw, PAPC_E_INVALID, "papc_mlp_bwd_dw_max_f32: null pointer");
    PAPC_REQUIRE(papc_mlp_max_nostore_ok(M, Cin, Cout, K), PAPC_E_UNSUPPORTED,
                 "papc_mlp_bwd_dw_max_f32: not built for M=%lld Cin=%d Cout=%d K=%d (see papc_mlp_max_nostore_ok)", (long long)M, Cin, Cout, K);
    PAPC_REQUIRE(aligned16(x) && aligned16(workspace), PAPC_E_INVALID, "papc_mlp_bwd_dw_max_f32: x / workspace must be 16-byte aligned");
    DwMaxArgs p;
    memset(&p, 0, sizeof(p));
    const int n = (Cout + Cin) * Cin + Cin;
    p.x = x; p.xsc = bn_scale; p.xsh = bn_shift; p.psel = psel; p.argmax = argmax; p.partial = workspace;
    p.M = M; p.part_ld = n; p.K = K; p.Cout = Cout; p.n_chunks = dw_max_chunks();
    p.rows_per_chunk = (int)(cdiv(cdiv(M, p.n_chunks), 64) * 64);
    double *fold = reinterpret_cast<double *>(workspace + (((int64_t)p.n_chunks * n + 1) & ~(int64_t)1));
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_BWD_DW, st);
    hipLaunchKernelGGL(dw_rows_max_kernel, dim3((unsigned)(2 * p.n_chunks)), dim3(512), 0, st, p);
    hipLaunchKernelGGL(dw_max_fold_kernel, dim3((unsigned)cdiv(n, 16)), dim3(64), 0, st, workspace, p.n_chunks, (int64_t)n, n, fold);
    hipLaunchKernelGGL(dw_max_finalize_kernel, dim3((unsigned)cdiv(Cout, 4)), dim3(256), 0, st, fold, w, e, scale, c1, 1.0 / (double)M, Cout, dw, accumulate);
    return check_launch("papc_mlp_bwd_dw_max_f32");
}
