// mlp_dw.hip -- weight gradient of a conv1x1 layer on rows: dW[Cout,Cin] = sum_m dY[m,:]^T A(x)[m,:]   (gfx950)
//
// Both operands are recomputed in the load (dY from (dz|gout+argmax, y) and the BN constants; A(x) from the
// previous layer's pre-BN output, or gathered rows) -- see mlp_loaders.h.  The reduction runs over M (up to 1M
// rows), the output is tiny, so the grid is split over row chunks; every workgroup owns one <=128x128 output
// tile for its chunk and writes a partial that papc_reduce_partials_f32 sums in fixed order (deterministic).
//
// MFMA mapping (v_mfma_f32_32x32x2_f32): A operand = dY^T (i = cout, k = row), B operand = X (k = row, j = cin).
// Tiles sit in LDS row-major [32 rows][128 ch]; lane l reads element [2*ks + (l>>5)][tile*32 + (l&31)] with
// ds_read_b32: the 32 lanes of each half read 32 consecutive banks -> conflict-free without padding.
#include "mlp_loaders.h"

namespace papc {

typedef float floatx16 __attribute__((ext_vector_type(16)));

void fill_dy(DySrc &d, const papc_bwd_dy *s);
int fill_asrc(ASrc &a, int a_mode, const float *x, int64_t ldx, const papc_group_src *grp, const float *sc,
              const float *sh, int Cin, const char *who);

struct DwArgs {
    ASrc x;       // A(x) producer  (PLAIN / BNRELU / GROUP)
    ASrc dy;      // dY producer    (DY_DENSE / DY_MAX)
    int64_t M; int Cin; int Cout; int rows_per_chunk;
    float *dw_partial;  // [n_chunks][Cout][Cin]
    float *db_partial;  // [n_chunks][Cout] or null
    int xmap;           // map internal cin -> caller's column (GROUP with xyz_first)
};

constexpr int DW_RS = 32;    // rows per LDS stage
constexpr int DW_T = 128;    // output tile edge (channels)

template <int XMODE, int DYMODE>
__global__ __launch_bounds__(256) void dw_kernel(DwArgs p)
{
    __shared__ __attribute__((aligned(16))) float smem[2 * DW_RS * DW_T + 8 * DW_T];
    float *Ys = smem;                    // dY tile  [32][128]
    float *Xs = smem + DW_RS * DW_T;     // X tile   [32][128]
    float *dbred = smem + 2 * DW_RS * DW_T;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hi = lane >> 5, l31 = lane & 31;
    const int o0 = blockIdx.y * DW_T, i0 = blockIdx.z * DW_T;
    const int nto = min(4, (p.Cout - o0 + 31) / 32);  // 32-wide tiles in this block's output tile
    const int nti = min(4, (p.Cin - i0 + 31) / 32);
    const int ntiles = nto * nti;
    const int kq = (tid & 31) * 4;   // channel group within the 128-wide tile
    const int rr = tid >> 5;         // 0..7

    floatx16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    // tile ids handled by this wave: wave, wave+4, wave+8, wave+12  -> (to = id / nti, ti = id % nti)
    int to_[4], ti_[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { const int id = wave + 4 * t; to_[t] = id / nti; ti_[t] = id - to_[t] * nti; }

    const KConst kcy = make_kconst<DYMODE>(p.dy, o0 + kq, p.Cout);
    const KConst kcx = make_kconst<XMODE>(p.x, i0 + kq, p.Cin);
    float4 dbs = make_float4(0.f, 0.f, 0.f, 0.f);

    const int64_t mbeg = (int64_t)blockIdx.x * p.rows_per_chunk;
    const int64_t mend = min(p.M, mbeg + p.rows_per_chunk);
    for (int64_t m0 = mbeg; m0 < mend; m0 += DW_RS) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = rr + 8 * i;
            const int64_t m = m0 + row;
            const int64_t mlim = mend;  // rows past the chunk end contribute zero
            RowCtx ry = make_row<DYMODE>(p.dy, m, mlim);
            float4 vy = load_a4<DYMODE>(p.dy, ry, o0 + kq, p.Cout, kcy);
            dbs.x += vy.x; dbs.y += vy.y; dbs.z += vy.z; dbs.w += vy.w;
            *reinterpret_cast<float4 *>(&Ys[row * DW_T + kq]) = vy;
            RowCtx rx = make_row<XMODE>(p.x, m, mlim);
            float4 vx = load_a4<XMODE>(p.x, rx, i0 + kq, p.Cin, kcx);
            *reinterpret_cast<float4 *>(&Xs[row * DW_T + kq]) = vx;
        }
        __syncthreads();
#pragma unroll 4
        for (int ks = 0; ks < DW_RS / 2; ++ks) {
            const float *yr = Ys + (2 * ks + hi) * DW_T + l31;
            const float *xr = Xs + (2 * ks + hi) * DW_T + l31;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (wave + 4 * t < ntiles) {
                    const float a = yr[to_[t] * 32];
                    const float b = xr[ti_[t] * 32];
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }

    // ---- store the partial tile: row (cout) = (r&3)+8*(r>>2)+4*hi, col (cin) = l31
    float *out = p.dw_partial + (int64_t)blockIdx.x * p.Cout * p.Cin;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if (wave + 4 * t < ntiles) {
            const int ci = i0 + ti_[t] * 32 + l31;
            if (ci < p.Cin) {
                const int cig = p.xmap ? gk(p.x.g, ci) : ci;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = o0 + to_[t] * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (co < p.Cout) out[(int64_t)co * p.Cin + cig] = acc[t][r];
                }
            }
        }
    }

    // ---- bias gradient partial: column sums of dY over this chunk (only the first cin tile writes it)
    if (p.db_partial && blockIdx.z == 0) {
        *reinterpret_cast<float4 *>(&dbred[rr * DW_T + kq]) = dbs;
        __syncthreads();
        if (tid < DW_T) {
            float s = 0.f;
#pragma unroll
            for (int g = 0; g < 8; ++g) s += dbred[g * DW_T + tid];
            if (o0 + tid < p.Cout) p.db_partial[(int64_t)blockIdx.x * p.Cout + o0 + tid] = s;
        }
    }
}

template <int XMODE, int DYMODE>
static int launch_dw(const DwArgs &p, hipStream_t st)
{
    dim3 grid((unsigned)cdiv(p.M, p.rows_per_chunk), (unsigned)cdiv(p.Cout, DW_T), (unsigned)cdiv(p.Cin, DW_T));
    hipLaunchKernelGGL((dw_kernel<XMODE, DYMODE>), grid, dim3(256), 0, st, p);
    return check_launch("papc_mlp_bwd_dw_f32");
}

}  // namespace papc

using namespace papc;

extern "C" int papc_mlp_bwd_dw_f32(const papc_bwd_dy *dy, int a_mode, const float *x, int64_t ldx,
                                   const papc_group_src *grp, const float *bn_scale, const float *bn_shift, int64_t M,
                                   int Cin, int Cout, int rows_per_chunk, float *dw_partial, float *db_partial,
                                   papc_stream_t stream)
{
    PAPC_REQUIRE(dy && dw_partial && dy->y && dy->mean && dy->invstd && dy->scale && dy->shift && dy->c1 && dy->c2,
                 PAPC_E_INVALID, "papc_mlp_bwd_dw_f32: null pointer");
    PAPC_REQUIRE(M >= 1 && Cin >= 1 && Cout >= 1, PAPC_E_INVALID, "papc_mlp_bwd_dw_f32: bad sizes");
    PAPC_REQUIRE(rows_per_chunk >= DW_RS && rows_per_chunk % DW_RS == 0, PAPC_E_INVALID,
                 "papc_mlp_bwd_dw_f32: rows_per_chunk=%d must be a positive multiple of %d", rows_per_chunk, DW_RS);
    DwArgs p;
    memset(&p, 0, sizeof(p));
    int rc = fill_asrc(p.x, a_mode, x, ldx, grp, bn_scale, bn_shift, Cin, "papc_mlp_bwd_dw_f32");
    if (rc) return rc;
    fill_dy(p.dy.d, dy);
    if (dy->dz_mode == PAPC_DZ_DENSE) {
        PAPC_REQUIRE(dy->dz, PAPC_E_INVALID, "papc_mlp_bwd_dw_f32: DENSE needs dz");
        p.dy.vec = aligned16(dy->dz) && aligned16(dy->y) && (Cout % 4 == 0);
    } else {
        PAPC_REQUIRE(dy->gout && dy->argmax && dy->K >= 1 && M % dy->K == 0, PAPC_E_INVALID, "papc_mlp_bwd_dw_f32: MAX needs gout/argmax/K | M");
        p.dy.vec = aligned16(dy->gout) && aligned16(dy->y) && (Cout % 4 == 0);
    }
    p.M = M; p.Cin = Cin; p.Cout = Cout; p.rows_per_chunk = rows_per_chunk; p.dw_partial = dw_partial; p.db_partial = db_partial;
    p.xmap = (a_mode == A_GROUP && grp->xyz_first) ? 1 : 0;
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_BWD_DW, st);
    const bool dense = dy->dz_mode == PAPC_DZ_DENSE;
    switch (a_mode) {
    case A_PLAIN: return dense ? launch_dw<A_PLAIN, A_DY_DENSE>(p, st) : launch_dw<A_PLAIN, A_DY_MAX>(p, st);
    case A_BNRELU: return dense ? launch_dw<A_BNRELU, A_DY_DENSE>(p, st) : launch_dw<A_BNRELU, A_DY_MAX>(p, st);
    default: return dense ? launch_dw<A_GROUP, A_DY_DENSE>(p, st) : launch_dw<A_GROUP, A_DY_MAX>(p, st);
    }
}
