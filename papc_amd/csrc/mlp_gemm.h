// mlp_gemm.h -- argument block shared by the two row-GEMM kernel families (mlp_gemm.hip: LDS-staged tiles with a barrier per
// k stage; mlp_stream.hip: barrier-free row streaming with the weights resident in LDS).
#pragma once
#include "mlp_loaders.h"

namespace papc {

typedef float floatx16 __attribute__((ext_vector_type(16)));

enum { EPI_STORE = 0, EPI_SCATTER = 1, EPI_STORE_RED = 2, EPI_STORE_GMAX = 3, EPI_GMAX = 4, EPI_XYZ_RED = 5 };   // (EPI_GMAX: EPI_STORE_GMAX without the output itself; row-streaming kernel only)

// EPI_STORE_RED (dX only): besides storing dz_prev = dX, accumulate the BN-backward reductions of the PREVIOUS layer
// (p = dz_prev * [scale*y_prev + shift > 0]; sum p and sum p*xhat per channel) into the stats partials, so no separate
// pass has to re-read dz_prev.
// EPI_XYZ_RED (dX of the layer above a coordinates-only first layer, row-streaming kernel only): dX is never stored -- the first layer's
// whole backward needs four sums per channel of p = dX [wf . x + t > 0] against the row's centred coordinates (xyz1.hip), accumulated here:
// RedSrc::y = xc [M][4], RedSrc::scale = the folded layer wf [Nout][4]; GemmArgs::stats = partial [parts][Nout][4] (T0, T1, T2, S).
struct RedSrc {
    const float *y; const float *mean, *invstd, *scale, *shift;  // previous layer: pre-BN output [M,Nout] and BN constants
    int masked;     // EPI_STORE_RED: store p (the ReLU-masked dX the sums are formed from) instead of dX (papc_bwd_red.store_masked)
};

struct ScatterDst {
    float *gf; const int32_t *idx; int N, S, K, D;
    FastDiv divSK, divK;
};

// EPI_STORE_GMAX (last forward layer): besides y and the statistics partials, write per group of K consecutive rows the
// max and min of y and the first row offset attaining each.  relu(scale*y+shift) is monotone in y (direction = sign of
// scale), so max_k relu(bn(y)) = relu(scale * (scale >= 0 ? max y : min y) + shift): the neighbourhood max
// (pointnet2_basic_layers.py:219) no longer needs a separate pass over y.  K in {32, 64, 128}, M % 128 == 0.
struct GmaxDst {
    float *gmax, *gmin; int32_t *amax, *amin; int K;
    const float *sgn;      // papc_group_max::sign_src (row-streaming kernel: one extremum per channel instead of two)
};

struct GemmArgs {
    ASrc a;
    const float *w; int64_t ldw;  // weights [Nout][Kin]
    int wmap;                      // map internal k -> weight column with gk() (GROUP forward)
    int nmap;                      // map internal n -> weight row with gk() (GROUP dX)
    const float *bias;
    int64_t M; int Kin; int Nout;
    float *y; int64_t ldy;
    float *stats;                  // [parts][2][Nout] or null
    int parts;                     // rows of `stats` the caller reduces (>= gridDim.x; the surplus rows are written as zeros)
    ScatterDst sc;
    RedSrc rd;
    GmaxDst gm;
    const int32_t *rows_dev;       // compacted stack (compact.hip): the physical row count in device memory (<= M, a multiple of 128); row-streaming kernel only
    const float *wstat;            // ... its rows' multiplicity weights: the forward statistics are those of the padded tensor (row-streaming kernel only)
    unsigned long long *dbg;       // PAPC_GEMM_DBG=1: per-workgroup cycle counters (development aid)
    int tl;                        // host: the transposed-accumulator epilogue is legal (16-byte aligned dense dX store)
};

constexpr int GEMM_MAX_PARTS = 768;  // rows of the per-workgroup partial buffers: up to 256 CUs x 3 resident workgroups (a kernel that
                                     // fits fewer per CU launches fewer and zero-fills the rows it does not own)

// mlp_stream.hip: returns 1 when the row-streaming kernel took the launch, 0 when the shape is not one of its flavours
// (the caller then runs the tiled kernel), < 0 on a launch error.
int stream_gemm_try(const GemmArgs &p, int amode, int epi, bool vec, hipStream_t st);

}  // namespace papc
