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
#include "mlp_loaders.h"

namespace papc {

typedef float floatx16 __attribute__((ext_vector_type(16)));

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

template <int XMODE, int DYMODE, bool VEC>
static int launch_dw_v(const DwArgs &p, hipStream_t st)
{
    const bool wide = p.TIp == DW_TI_WIDE;
    dim3 grid((unsigned)cdiv(p.M, p.rows_per_chunk), (unsigned)cdiv(p.Cout, DW_T), (unsigned)cdiv(p.Cin, wide ? DW_TI_WIDE : DW_T));
    const int to = p.TOp / 32, ti = p.TIp / 32;  // 32-wide tiles per workgroup (upper bound)
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
    switch (a_mode) {
    case A_PLAIN: return dense ? launch_dw<A_PLAIN, A_DY_DENSE>(p, vec, st) : launch_dw<A_PLAIN, A_DY_MAX>(p, vec, st);
    case A_BNRELU: return dense ? launch_dw<A_BNRELU, A_DY_DENSE>(p, vec, st) : launch_dw<A_BNRELU, A_DY_MAX>(p, vec, st);
    default: return dense ? launch_dw<A_GROUP, A_DY_DENSE>(p, vec, st) : launch_dw<A_GROUP, A_DY_MAX>(p, vec, st);
    }
}
