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
// Every wave owns NT (1, 2 or 4) 32x32 output tiles; absent tiles (narrow layers) are computed on clamped
// coordinates and simply not stored, so the MFMA loop has no branches.
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
    int xmap;           // map internal cin -> caller's column (GROUP)
    int TOp, TIp;       // padded tile widths (32 / 64 / 128)
    int RS;             // rows per stage (32 or 64)
};

constexpr int DW_T = 128;              // output tile edge (channels)
constexpr int DW_TI_WIDE = 160;           // gather layers (Cin = D+3 in (128,160]) keep all of cin in ONE tile
constexpr int DW_STAGE_FLOATS = 9216;  // RS * (TOp + TIp) <= 32*(128+160) floats = 36 KiB per buffer

template <int XMODE, int DYMODE, bool VEC, int NT, int NSX>
__global__ __launch_bounds__(256, 2) void dw_kernel(DwArgs p)
{
    __shared__ __attribute__((aligned(16))) float smem[2 * DW_STAGE_FLOATS + DW_T];
    float *dbred = smem + 2 * DW_STAGE_FLOATS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hi = lane >> 5, l31 = lane & 31;
    const int o0 = blockIdx.y * DW_T, i0 = blockIdx.z * (NSX > 4 ? DW_TI_WIDE : DW_T);
    const int TOp = p.TOp, TIp = p.TIp, RS = p.RS;
    const int nto = min(TOp / 32, (p.Cout - o0 + 31) / 32);  // 32-wide tiles actually present
    const int nti = min(TIp / 32, (p.Cin - i0 + 31) / 32);
    const int ntiles = nto * nti;
    // loader mapping: a stage of the dY tile is RS x TOp floats = RS*TOp/4 float4 slots; thread t owns slots
    // t, t+256, ... -> channel group fixed per thread (TOp/4 divides 256), rows advance by 256/(TOp/4)
    const int cgy = TOp / 4, cgx = TIp / 4;
    const int kqy = (tid % cgy) * 4, ry0 = tid / cgy, rsy = 256 / cgy, nsy = RS / rsy;  // nsy <= 4
    // X slots are mapped generically (slot = tid + 256*i -> row = slot / cgx, channel group = slot % cgx) so the
    // 160-wide gather tile (cgx = 40, which does not divide 256) works too; when cgx | 256 every slot of a thread
    // has the same channel group, which the per-thread BN constants of the BNRELU producer rely on.
    const int nsx = (RS * cgx + 255) / 256;
    int xr[NSX], xk[NSX];
#pragma unroll
    for (int i = 0; i < NSX; ++i) {
        const int s = tid + 256 * i;
        xr[i] = s / cgx;
        xk[i] = (s - xr[i] * cgx) * 4;
        if (xr[i] >= RS) { xr[i] = RS - 1; }  // clamped duplicate slot (same value written twice: harmless)
    }
    const int kqx = xk[0];
    const bool use_jpre = (XMODE == A_GROUP) && p.x.g.idx != nullptr;

    floatx16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    // tile ids of this wave: wave, wave+4, ... ; absent ids are clamped onto tile 0 (computed, never stored)
    int toff[NT], tiff[NT];
    bool tok[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int id = wave + 4 * t;
        tok[t] = id < ntiles;
        const int idc = tok[t] ? id : 0;
        const int to = idc / nti, ti = idc - to * nti;
        toff[t] = to * 32; tiff[t] = ti * 32;
    }

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
    auto fetch = [&](int64_t m0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i < nsy) {
                rowy[i] = make_row<DYMODE>(p.dy, m0 + ry0 + rsy * i, mend);
                rawy[i] = fetch_a4<DYMODE, VEC>(p.dy, rowy[i], o0 + kqy, p.Cout);
            }
        }
#pragma unroll
        for (int i = 0; i < NSX; ++i) {
            if (i < nsx) {
                rowx[i] = make_row<XMODE>(p.x, m0 + xr[i], mend, use_jpre ? jpre[i] : -2);
                rawx[i] = fetch_a4<XMODE, VEC>(p.x, rowx[i], i0 + xk[i], p.Cin);
            }
        }
    };
    auto finish = [&](float *Ys, float *Xs) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i < nsy) {
                const float4 vy = finish_a4<DYMODE, VEC>(p.dy, rowy[i], o0 + kqy, p.Cout, kcy, rawy[i]);
                dbs.x += vy.x; dbs.y += vy.y; dbs.z += vy.z; dbs.w += vy.w;
                *reinterpret_cast<float4 *>(&Ys[(ry0 + rsy * i) * TOp + kqy]) = vy;
            }
        }
#pragma unroll
        for (int i = 0; i < NSX; ++i) {
            if (i < nsx) {
                const float4 vx = finish_a4<XMODE, VEC>(p.x, rowx[i], i0 + xk[i], p.Cin, kcx, rawx[i]);
                *reinterpret_cast<float4 *>(&Xs[xr[i] * TIp + xk[i]]) = vx;
            }
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
            const float *Ys = smem + buf * DW_STAGE_FLOATS + hi * TOp + l31;
            const float *Xs = smem + buf * DW_STAGE_FLOATS + RS * TOp + hi * TIp + l31;
            const int nks = RS / 2;
#pragma unroll 4
            for (int ks = 0; ks < nks; ++ks) {
                const float *yr = Ys + 2 * ks * TOp;
                const float *xr = Xs + 2 * ks * TIp;
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(yr[toff[t]], xr[tiff[t]], acc[t], 0, 0, 0);
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
    float *out = p.dw_partial + (int64_t)blockIdx.x * p.Cout * p.Cin;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int ci = i0 + tiff[t] + l31;
        if (tok[t] && ci < p.Cin) {
            const int cig = p.xmap ? gk(p.x.g, ci) : ci;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = o0 + toff[t] + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (co < p.Cout) out[(int64_t)co * p.Cin + cig] = acc[t][r];
            }
        }
    }

    // ---- bias gradient partial: column sums of dY over this chunk (only the first cin tile writes it).
    // Threads with the same channel group differ in ry0 (0 .. rsy-1): add them in that fixed order (deterministic).
    if (p.db_partial && blockIdx.z == 0) {
        if (tid < DW_T) dbred[tid] = 0.f;
        __syncthreads();
        for (int g = 0; g < rsy; ++g) {
            if (ry0 == g) { dbred[kqy + 0] += dbs.x; dbred[kqy + 1] += dbs.y; dbred[kqy + 2] += dbs.z; dbred[kqy + 3] += dbs.w; }
            __syncthreads();
        }
        if (tid < TOp && o0 + tid < p.Cout) p.db_partial[(int64_t)blockIdx.x * p.Cout + o0 + tid] = dbred[tid];
    }
}

template <int XMODE, int DYMODE, bool VEC>
static int launch_dw_v(const DwArgs &p, hipStream_t st)
{
    const bool wide = p.TIp == DW_TI_WIDE;
    dim3 grid((unsigned)cdiv(p.M, p.rows_per_chunk), (unsigned)cdiv(p.Cout, DW_T), (unsigned)cdiv(p.Cin, wide ? DW_TI_WIDE : DW_T));
    const int tiles = (p.TOp / 32) * (p.TIp / 32);  // upper bound of 32x32 tiles per workgroup
    if (wide) {
        if (XMODE == A_GROUP) hipLaunchKernelGGL((dw_kernel<A_GROUP, DYMODE, VEC, 5, 5>), grid, dim3(256), 0, st, p);
    } else if (tiles <= 4) hipLaunchKernelGGL((dw_kernel<XMODE, DYMODE, VEC, 1, 4>), grid, dim3(256), 0, st, p);
    else if (tiles <= 8) hipLaunchKernelGGL((dw_kernel<XMODE, DYMODE, VEC, 2, 4>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((dw_kernel<XMODE, DYMODE, VEC, 4, 4>), grid, dim3(256), 0, st, p);
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
                                   papc_stream_t stream)
{
    PAPC_REQUIRE(dw_partial, PAPC_E_INVALID, "papc_mlp_bwd_dw_f32: null dw_partial");
    PAPC_REQUIRE(M >= 1 && Cin >= 1 && Cout >= 1, PAPC_E_INVALID, "papc_mlp_bwd_dw_f32: bad sizes");
    PAPC_REQUIRE(M < (1ll << 31), PAPC_E_UNSUPPORTED, "papc_mlp_bwd_dw_f32: M=%lld >= 2^31 rows", (long long)M);
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
    const bool vec = vdy && p.x.vec;
    p.M = M; p.Cin = Cin; p.Cout = Cout; p.rows_per_chunk = rows_per_chunk; p.dw_partial = dw_partial; p.db_partial = db_partial;
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
