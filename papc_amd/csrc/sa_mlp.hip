// sa_mlp.hip -- ONE call per direction for a whole shared-MLP stack: papc_sa_mlp_plan / _fwd / _bwd (host code; gfx950 kernels live in
// the other files of this directory).
//
// What the reference does in PointNetSetAbstraction.forward after the grouping -- relu(bn(conv(.))) x L, max over the neighbourhood
// (PAPC/models/layers/pointnet2_basic_layers.py:214-219; the Msg variant :271-276; PointNetFeaturePropagation's Conv1D stack :330-333;
// PointNet-Basic's, classify/pointnet_base/pointnet_base.py:7-25, :44) -- is here a sequence of 10-25 launches whose choice depends on
// the shapes: gather-add first layer, coordinates-only first layer through its input moments, row-streaming or tiled GEMMs, fused or
// separate neighbourhood max, a max layer that never stores its output, the compacted (distinct-neighbours) form.  This file owns
// that choice and the scratch layout, so a host binds three functions instead of re-implementing the sequence:
//
//   papc_sa_mlp_plan   descriptor -> which path each layer takes + bytes of `saved` (forward -> backward) and of scratch per direction
//   papc_sa_mlp_fwd    inputs + parameters -> out (+ running statistics), fills `saved`
//   papc_sa_mlp_bwd    gout + `saved` -> parameter gradients (written or accumulated in place), grad_feats / grad_x
//
// All buffers are caller-owned device memory; nothing is allocated, nothing synchronises, every launch goes to the given stream.
#include "common.h"

#define SA_CALL(expr)                 \
    do {                              \
        const int rc__ = (expr);      \
        if (rc__ != PAPC_OK) return rc__; \
    } while (0)

namespace papc {
// (pfn.hip: the fused launches of the PillarFeatureNet entry points below)
int pfn_gram_stats(const float *features, const int32_t *num_voxels, const int32_t *coors, int P, int T, float vx, float vy, float x_offset, float y_offset, int zero_padded,
                   double *gram_partial, const float *w, int C, const float *gamma, const float *beta, float eps, float momentum, float *mean, float *invstd,
                   float *scale, float *shift, float *running_mean, float *running_var, double *gram, unsigned *ticket, hipStream_t st);
int pfn_gram_impl(const float *features, const int32_t *num_voxels, const int32_t *coors, int P, int T, float vx, float vy, float x_offset, float y_offset,
                  double *gram_partial, int zero_padded, papc_stream_t stream);
int pfn_apply_impl(const float *features, const int32_t *num_voxels, const int32_t *coors, int P, int T, float vx, float vy, float x_offset, float y_offset,
                   const float *w, int C, const float *scale, const float *shift, float *out, int32_t *argmax, int zero_padded, papc_stream_t stream);
int pfn_bwd_sparse_impl(const float *features, const int32_t *num_voxels, const int32_t *coors, int P, int T, float vx, float vy, float x_offset, float y_offset,
                        const float *w, int C, const float *gout, const int32_t *argmax, const float *mean, const float *invstd, const float *scale,
                        const float *shift, float *partial, int zero_padded, papc_stream_t stream);
int pfn_bwd_fold_finalize(const float *partial, int n_chunks, float *sums, int64_t M, const float *w, int C, const double *gram, const float *mean,
                          const float *invstd, const float *scale, float *dgamma, float *dbeta, float *dw, int flags, unsigned *ticket, hipStream_t st);

constexpr int A_PLAIN_ = PAPC_A_PLAIN, A_BNRELU_ = PAPC_A_BNRELU, A_GROUP_ = PAPC_A_GROUP, A_XYZ_ = PAPC_A_XYZ;

struct Carver {                       // hands out 256-byte aligned pieces of one buffer; base == nullptr: sizes only
    char *base;
    size_t off;
    template <class T>
    T *take(size_t n)
    {
        off = (off + 255) & ~(size_t)255;
        T *p = base ? reinterpret_cast<T *>(base + off) : nullptr;
        off += n * sizeof(T);
        return p;
    }
};

struct SavedPtrs {
    float *y[PAPC_SA_MAX_LAYERS];       // pre-BN outputs [M, c_l] (nullptr where a layer's output is never stored)
    float *cst[PAPC_SA_MAX_LAYERS];     // [4, c_l]: mean, invstd, scale, shift
    int32_t *argmax;                    // [G, c_L]
    float *gbuf_f;                      // [2, G, c_L]: per-group max | min of the raw output; [0] becomes ysel (fused / ragged max)
    float *xc; double *gram; float *wf; // coordinates-only first layer: grouped coordinates [M, 4] (unless handed in), moments [16], folded layer [c_0, 4]
    float *P;                           // gather-add first layer: P = feats W_f^T [B * N, c_0] (the list backward without y needs it again)
};
struct FwdPtrs {
    float *stats;                       // [max parts + corr rows, 2, max c]
    int32_t *gbuf_i;                    // [2, G, c_L]
    float *P, *wfeat;                   // gather-add first layer: per-point products [B*N, c_0], feature block of its weight [c_0, D]
    double *gpart;                      // coordinates-only first layer: partial moments
};
struct BwdPtrs {
    float *c12[PAPC_SA_MAX_LAYERS];     // [2, c_l]
    float *wt[PAPC_SA_MAX_LAYERS];      // W^T [cin_l, c_l] where the caller gave none
    float *part[PAPC_SA_MAX_LAYERS];    // dW / db partials [n_chunks, c_l * cin_l + c_l]
    float *red, *fused_red[2];          // BN-backward partial sums
    float *dz[2];                       // ping-pong [M, max cin]
    float *psel, *wcat, *hb, *eq, *dwmax_ws;
    float *bpart;                       // xyz1
    float *dwx_part, *Gs, *part_g, *wft;   // gather-add first layer
    float *tmp_gb;                      // [2, max c] throw-away gamma / beta gradients
};

static inline int64_t rows_of(const papc_sa_desc &d) { return (int64_t)d.B * d.S * d.K; }

// deferred folds (papc_sa_grads.defer, include/papc_hip.h): append to the caller's list; false = no list or no room (the caller of this
// helper then launches the fold itself, as without a list)
static bool defer_fold(const papc_sa_grads &gr, const float *partial, int n_chunks, int64_t ld, int rows, int cols, float *out, int64_t out_ld, int accumulate)
{
    papc_fold_list *fl = gr.defer;
    if (!fl || !fl->jobs || fl->count >= fl->capacity) return false;
    papc_fold_job &j = fl->jobs[fl->count++];
    j.partial = partial; j.n_chunks = n_chunks; j.accumulate = accumulate; j.ld = ld; j.rows = rows; j.cols = cols; j.out = out; j.out_ld = out_ld;
    return true;
}
static bool defer_room(const papc_sa_grads &gr, int n) { return gr.defer && gr.defer->jobs && gr.defer->count + n <= gr.defer->capacity; }
static inline int cin_of(const papc_sa_plan &p, int l) { return l == 0 ? p.cin0 : p.d.cout[l - 1]; }

// rows per dW chunk, at least 256 -- except the gather-add layer's dW_f = G^T feats over the B N source points: 64 (at 256 its 16 384 rows ran on 64
// workgroups, a quarter of the chip, for 24 us: step 1.482 -> 1.464 ms; the same floor for every small-M launch cost config 3 0.03 ms)
static int dw_rows_per_chunk(int64_t M, int cout, int cin, int min_rows = 256)      // (mlp.py::_dw_rows_per_chunk: one residency wave of workgroups in total)
{
    const bool wide = cin > 128 && cin <= 160;
    const int tiles = ((cout + 127) / 128) * (wide ? 1 : (cin + 127) / 128);
    const int want = std::max(1, 512 / tiles);
    int64_t rpc = (M + want - 1) / want;
    rpc = std::max<int64_t>(min_rows, ((rpc + 63) / 64) * 64);
    return (int)rpc;
}

static int dw_chunk(const papc_sa_plan &p, int l, int a_mode, int dz_mode)
{
    const int64_t M = rows_of(p.d);
    int rpc = 0;
    const bool plain = p.d.input == PAPC_SA_IN_ROWS;
    if (l > 0 || !plain) rpc = papc_mlp_bwd_dw_chunk_hint(M, cin_of(p, l), p.d.cout[l], a_mode, dz_mode, dz_mode == PAPC_DZ_MAX ? p.d.K : 0);
    if (rpc <= 0) rpc = dw_rows_per_chunk(M, p.d.cout[l], cin_of(p, l));
    return rpc;
}

static size_t layout_saved(const papc_sa_plan &p, void *base, SavedPtrs &s)
{
    Carver c{reinterpret_cast<char *>(base), 0};
    const int L = p.d.n_layers;
    const int64_t M = rows_of(p.d), G = (int64_t)p.d.B * p.d.S;
    memset(&s, 0, sizeof(s));
    for (int l = 0; l < L; ++l) {
        const bool stored = !(l == 0 && p.xyz1) && !(l == L - 1 && p.nostore);
        s.y[l] = stored ? c.take<float>((size_t)M * p.d.cout[l] + 64) : nullptr;
        s.cst[l] = c.take<float>(4 * (size_t)p.d.cout[l]);
    }
    if (p.d.pool) {
        s.argmax = c.take<int32_t>((size_t)G * p.d.cout[L - 1]);
        if (p.gmax || p.compact) s.gbuf_f = c.take<float>(2 * (size_t)G * p.d.cout[L - 1]);
    }
    if (p.xyz1) {
        s.xc = c.take<float>((size_t)M * 4);
        s.gram = c.take<double>(16);
        s.wf = c.take<float>(4 * (size_t)p.d.cout[0]);
    }
    if (p.lin0) s.P = c.take<float>((size_t)p.d.B * p.d.N * p.d.cout[0]);      // (last: the offsets ahead of it are part of papc_sa_plan)
    return c.off;
}

static size_t layout_fwd(const papc_sa_plan &p, void *base, FwdPtrs &f)
{
    Carver c{reinterpret_cast<char *>(base), 0};
    const int L = p.d.n_layers;
    const int64_t M = rows_of(p.d), G = (int64_t)p.d.B * p.d.S;
    memset(&f, 0, sizeof(f));
    int cmax = 0;
    for (int l = 0; l < L; ++l) cmax = std::max(cmax, p.d.cout[l]);
    const int rows = std::max(papc_mlp_gemm_parts(M), p.lin0 ? papc_lingather_parts(M) : 0) + (p.compact ? papc_compact_corr_parts() : 0);
    f.stats = c.take<float>((size_t)rows * 2 * cmax * L);          // one region per layer: a layer's finalize may still read while the next GEMM writes
    if (p.gmax) f.gbuf_i = c.take<int32_t>(2 * (size_t)G * p.d.cout[L - 1]);
    if (p.lin0) f.wfeat = c.take<float>((size_t)p.d.cout[0] * p.d.D);     // (P itself lives in `saved`: the backward reads it again)
    if (p.xyz1) f.gpart = c.take<double>((size_t)papc_xyz_parts(M) * 16 + 16);
    return c.off;
}

static size_t layout_bwd(const papc_sa_plan &p, void *base, BwdPtrs &b)
{
    Carver c{reinterpret_cast<char *>(base), 0};
    const int L = p.d.n_layers;
    const int64_t M = rows_of(p.d), G = (int64_t)p.d.B * p.d.S;
    memset(&b, 0, sizeof(b));
    int cmax = p.cin0;
    for (int l = 0; l < L; ++l) cmax = std::max(cmax, p.d.cout[l]);
    const int n_parts = (int)std::min<int64_t>(512, (M + 127) / 128);
    const int gparts = papc_mlp_gemm_parts(M);
    for (int l = 0; l < L; ++l) {
        const int cin = cin_of(p, l), cout = p.d.cout[l];
        b.c12[l] = c.take<float>(2 * (size_t)cout);
        b.wt[l] = c.take<float>((size_t)cin * cout);
        // (dW partials: sized for the larger of the two operand flavours a layer can take)
        const int dzm = (l == L - 1 && p.d.pool) ? PAPC_DZ_MAX : PAPC_DZ_DENSE;
        const int am = l == 0 ? (p.d.input == PAPC_SA_IN_ROWS ? A_PLAIN_ : A_GROUP_) : ((l == 1 && p.xyz1) ? A_XYZ_ : A_BNRELU_);
        const int rpc = dw_chunk(p, l, am, dzm);
        const int64_t n_chunks = (M + rpc - 1) / rpc;
        b.part[l] = c.take<float>((size_t)n_chunks * ((size_t)cout * cin + cout));
    }
    b.red = c.take<float>((size_t)n_parts * 2 * cmax);
    b.fused_red[0] = c.take<float>((size_t)gparts * 2 * cmax);
    b.fused_red[1] = c.take<float>((size_t)gparts * 2 * cmax);
    if (L > 1 || p.d.input == PAPC_SA_IN_ROWS) {
        b.dz[0] = c.take<float>((size_t)M * cmax + 64);
        if (L > 2) b.dz[1] = c.take<float>((size_t)M * cmax + 64);
    }
    if (p.d.pool && L > 1) {
        const int cL = p.d.cout[L - 1], cLi = p.d.cout[L - 2];
        b.psel = c.take<float>((size_t)G * cL);
        b.wcat = c.take<float>((size_t)cLi * (cL + cLi));
        b.hb = c.take<float>((size_t)cLi);
        b.eq = c.take<float>(2 * (size_t)cL);
        if (p.nostore) b.dwmax_ws = c.take<float>((size_t)papc_mlp_bwd_dw_max_ws_floats(M, cLi, cL) + 4);
    }
    if (p.xyz1) b.bpart = c.take<float>((size_t)std::max(papc_xyz_bwd_parts(M), papc_mlp_gemm_parts(M)) * p.d.cout[0] * 4);
    if (p.lin0) {
        const int c0 = p.d.cout[0];
        const int64_t BN = (int64_t)p.d.B * p.d.N;
        b.dwx_part = c.take<float>((size_t)std::max(papc_lingather_parts(M), papc_lingather_list_parts(BN)) * c0 * 3);
        b.Gs = c.take<float>((size_t)BN * c0);
        const int rpc_g = dw_rows_per_chunk(BN, c0, p.d.D, 64);
        b.part_g = c.take<float>((size_t)((BN + rpc_g - 1) / rpc_g) * ((size_t)c0 * p.d.D + c0));
        b.wft = c.take<float>((size_t)p.d.D * c0);
    }
    b.tmp_gb = c.take<float>(2 * (size_t)cmax);
    return c.off;
}

// ---- the planes path (csrc/smallm.hip): few-row stacks -- sample_and_group_all (pointnet2_basic_layers.py:160-176), point-wise stacks ------
constexpr int PG_GROUP = 128, PG_MAX_ROWS = 16384;
struct PlanesSaved {
    float *y[PAPC_SA_MAX_LAYERS], *cst[PAPC_SA_MAX_LAYERS];
    int32_t *argmax; float *gbuf_f;     // argmax [G, c_L]; per-TILE max | min [2, T, c_L]
    float *ysel;                        // [G, c_L] raw y at the argmax (= gbuf_f when a group is one tile)
    void *PT[PAPC_SA_MAX_LAYERS];       // input^T planes of every layer (dW operands)
    void *wtp[PAPC_SA_MAX_LAYERS];      // W^T planes (dX operands)
};
struct PlanesFwd { void *wp[PAPC_SA_MAX_LAYERS]; void *P; float *stats[PAPC_SA_MAX_LAYERS]; int32_t *gbuf_i; };
struct PlanesBwd { void *dyp, *dypt; float *dz[2], *red[3], *part[PAPC_SA_MAX_LAYERS], *tmp_gb; };

static int pg_split_for(int R1, int R2, int nst)      // (smallm.py::_split_for: about one workgroup per CU, >= 8 k32 stages each -- 2 for one- or two-tile products --, a power of two dividing nst)
{
    const int tiles = ((R1 + 127) / 128) * ((R2 + 127) / 128);
    int s = 1;
    // (a product of one or two output tiles -- PointNet-Basic's 64-channel layers -- is split down to 2 stages per workgroup: 128 workgroups instead of 32,
    // config 0 0.359 -> 0.335 ms; the same for the wide layers of config 2's SA3 costs more in partial traffic than it gains: 1.512 -> 1.533 ms)
    const int min_stages = tiles <= 2 ? 2 : 8;
    while (s * 2 * tiles <= 256 && nst % (s * 2) == 0 && nst / (s * 2) >= min_stages) s *= 2;
    return s;
}
static inline int pg_n_in(const papc_sa_plan &p) { return p.d.input == PAPC_SA_IN_ROWS ? p.cin0 : p.d.D; }

static size_t layout_planes_saved(const papc_sa_plan &p, void *base, PlanesSaved &s)
{
    Carver c{reinterpret_cast<char *>(base), 0};
    const int L = p.d.n_layers;
    const int64_t M = rows_of(p.d), T = M / PG_GROUP;
    memset(&s, 0, sizeof(s));
    for (int l = 0; l < L; ++l) {
        s.y[l] = c.take<float>((size_t)M * p.d.cout[l]);
        s.cst[l] = c.take<float>(4 * (size_t)p.d.cout[l]);
    }
    if (p.d.pool) {
        s.argmax = c.take<int32_t>((size_t)T * p.d.cout[L - 1]);
        s.gbuf_f = c.take<float>(2 * (size_t)T * p.d.cout[L - 1]);
        s.ysel = p.d.K > PG_GROUP ? c.take<float>((size_t)(M / p.d.K) * p.d.cout[L - 1]) : s.gbuf_f;
    }
    if (!p.d.inference)
        for (int l = 0; l < L; ++l) {
            s.PT[l] = c.take<char>(papc_pg_planes_bytes(cin_of(p, l), M));
            if (l > 0) s.wtp[l] = c.take<char>(papc_pg_planes_bytes(cin_of(p, l), p.d.cout[l]));
            else if (p.d.want_input_grad) s.wtp[0] = c.take<char>(papc_pg_planes_bytes(pg_n_in(p), p.d.cout[0]));
        }
    return c.off;
}

static size_t layout_planes_fwd(const papc_sa_plan &p, void *base, PlanesFwd &f)
{
    Carver c{reinterpret_cast<char *>(base), 0};
    const int L = p.d.n_layers;
    const int64_t M = rows_of(p.d), T = M / PG_GROUP;
    memset(&f, 0, sizeof(f));
    size_t pmax = papc_pg_planes_bytes(M, p.cin0);
    for (int l = 0; l < L; ++l) {
        f.wp[l] = c.take<char>(papc_pg_planes_bytes(p.d.cout[l], cin_of(p, l)));
        f.stats[l] = c.take<float>((size_t)T * 2 * p.d.cout[l]);
        if (l < L - 1) pmax = std::max(pmax, papc_pg_planes_bytes(M, p.d.cout[l]));
    }
    f.P = c.take<char>(pmax);
    if (p.d.pool) f.gbuf_i = c.take<int32_t>(2 * (size_t)T * p.d.cout[L - 1]);
    return c.off;
}

static size_t layout_planes_bwd(const papc_sa_plan &p, void *base, PlanesBwd &b)
{
    Carver c{reinterpret_cast<char *>(base), 0};
    const int L = p.d.n_layers;
    const int64_t M = rows_of(p.d), T = M / PG_GROUP;
    memset(&b, 0, sizeof(b));
    int cmax = p.cin0;
    size_t pm = 0, pmt = 0;
    for (int l = 0; l < L; ++l) {
        cmax = std::max(cmax, p.d.cout[l]);
        pm = std::max(pm, papc_pg_planes_bytes(M, p.d.cout[l]));
        pmt = std::max(pmt, papc_pg_planes_bytes(p.d.cout[l], M));
        b.part[l] = c.take<float>((size_t)pg_split_for(p.d.cout[l], cin_of(p, l), (int)(M / 32)) * p.d.cout[l] * cin_of(p, l));
    }
    b.dyp = c.take<char>(pm);
    b.dypt = c.take<char>(pmt);
    b.dz[0] = c.take<float>((size_t)M * cmax);
    b.dz[1] = c.take<float>((size_t)M * cmax);
    for (int i = 0; i < 3; ++i) b.red[i] = c.take<float>((size_t)T * 2 * cmax);
    b.tmp_gb = c.take<float>(2 * (size_t)cmax);
    return c.off;
}

static int planes_fwd(const papc_sa_plan &p, const papc_sa_io &io, papc_stream_t st)
{
    const papc_sa_desc &d = p.d;
    const int L = d.n_layers;
    const int64_t M = rows_of(d), T = M / PG_GROUP;
    const bool plain = d.input == PAPC_SA_IN_ROWS, want_bwd = !d.inference;
    PlanesSaved s; PlanesFwd f;
    layout_planes_saved(p, io.saved, s);
    layout_planes_fwd(p, io.scratch, f);
    // ---- weights -> planes: W_l [c_l x c_(l-1)] for the forward, W_l^T [c_(l-1) x c_l] for dX (layer 1: the gradient-carrying columns)
    {
        papc_pg_wjob jobs[2 * PAPC_SA_MAX_LAYERS];
        int n = 0;
        const int fcol0 = plain ? 0 : (d.xyz_first ? 3 : 0);
        for (int l = 0; l < L; ++l) {
            const int cin = cin_of(p, l), cout = d.cout[l];
            jobs[n++] = papc_pg_wjob{io.layer[l].w, cin, 1, cout, cin, f.wp[l]};
            if (s.wtp[l]) {
                if (l > 0) jobs[n++] = papc_pg_wjob{io.layer[l].w, 1, cin, cin, cout, s.wtp[l]};
                else jobs[n++] = papc_pg_wjob{io.layer[l].w + fcol0, 1, cin, pg_n_in(p), cout, s.wtp[l]};
            }
        }
        for (int j0 = 0; j0 < n; j0 += 8) SA_CALL(papc_pg_prep_weights_f32(jobs + j0, std::min(8, n - j0), st));
    }
    // ---- layer 1 operand: the rows of sample_and_group_all (or the caller's rows)
    papc_pg_prep a;
    memset(&a, 0, sizeof(a));
    a.M = M; a.C = p.cin0; a.planes = f.P; a.planes_t = want_bwd ? s.PT[0] : nullptr;
    if (plain) { a.mode = PAPC_PG_PLAIN; a.x = io.x_rows; a.ldx = p.cin0; }
    else { a.mode = PAPC_PG_CONCAT; a.xyz = io.xyz; a.sb = io.sb; a.sn = io.sn; a.sc = io.sc; a.feats = io.feats; a.N = d.N; a.D = d.D; a.xyz_first = d.xyz_first; }
    SA_CALL(papc_pg_prep_rows_f32(&a, st));
    for (int l = 0; l < L; ++l) {
        const papc_sa_layer &ly = io.layer[l];
        const int cout = d.cout[l], cin = cin_of(p, l);
        float *cst = s.cst[l];
        const bool last_pool = l == L - 1 && d.pool;
        papc_pg_gemm g;
        memset(&g, 0, sizeof(g));
        g.epi = last_pool ? PAPC_PG_FWD_GMAX : PAPC_PG_FWD;
        g.a = f.P; g.b = f.wp[l]; g.R1 = (int)M; g.R2 = cout; g.K = cin;
        g.c = s.y[l]; g.ldc = cout; g.split = 1; g.bias = ly.b; g.stats = f.stats[l]; g.family = PAPC_K_MLP_GEMM;
        if (last_pool) { g.gmax = s.gbuf_f; g.gmin = s.gbuf_f + T * cout; g.amax = f.gbuf_i; g.amin = f.gbuf_i + T * cout; }
        SA_CALL(papc_pg_gemm_f32(&g, st));
        if (l < L - 1) {
            // BN statistics of this layer folded in the prologue of the NEXT layer's operand prep (relu(bn(y)) -> planes)
            papc_pg_prep b;
            memset(&b, 0, sizeof(b));
            b.mode = PAPC_PG_BNRELU; b.M = M; b.C = cout; b.x = s.y[l]; b.ldx = cout; b.stats = f.stats[l]; b.parts = (int)T;
            b.gamma = ly.gamma; b.beta = ly.beta; b.eps = d.eps; b.momentum = d.momentum; b.running_mean = ly.running_mean; b.running_var = ly.running_var;
            b.mean = cst; b.invstd = cst + cout; b.scale = cst + 2 * cout; b.shift = cst + 3 * cout;
            b.planes = f.P; b.planes_t = want_bwd ? s.PT[l + 1] : nullptr;
            SA_CALL(papc_pg_prep_rows_f32(&b, st));
        } else if (!d.pool) {
            SA_CALL(papc_bn_finalize_f32(f.stats[l], (int)T, M, cout, ly.gamma, ly.beta, d.eps, d.momentum, cst, cst + cout, cst + 2 * cout, cst + 3 * cout,
                                         ly.running_mean, ly.running_var, st));
            SA_CALL(papc_bn_relu_f32(s.y[l], cst + 2 * cout, cst + 3 * cout, M, cout, io.out, st));
        } else if (d.K > PG_GROUP) {     // a group spans several 128-row tiles: the extreme of the tiles' extrema
            SA_CALL(papc_pg_final_groups_f32(f.stats[l], (int)T, M, cout, ly.gamma, ly.beta, d.eps, d.momentum, cst, cst + cout, cst + 2 * cout, cst + 3 * cout,
                                             ly.running_mean, ly.running_var, s.gbuf_f, s.gbuf_f + T * cout, f.gbuf_i, f.gbuf_i + T * cout, M / d.K, d.K / PG_GROUP,
                                             io.out, s.argmax, s.ysel, st));
        } else {
            SA_CALL(papc_pg_final_f32(f.stats[l], (int)T, M, cout, ly.gamma, ly.beta, d.eps, d.momentum, cst, cst + cout, cst + 2 * cout, cst + 3 * cout,
                                      ly.running_mean, ly.running_var, s.gbuf_f, s.gbuf_f + T * cout, f.gbuf_i, f.gbuf_i + T * cout, T, io.out, s.argmax, st));
        }
    }
    return PAPC_OK;
}

static int planes_bwd(const papc_sa_plan &p, const papc_sa_io &io, const papc_sa_grads &gr, papc_stream_t st)
{
    const papc_sa_desc &d = p.d;
    const int L = d.n_layers;
    const int64_t M = rows_of(d), T = M / PG_GROUP;
    const bool plain = d.input == PAPC_SA_IN_ROWS;
    PAPC_REQUIRE(!d.inference, PAPC_E_INVALID, "papc_sa_mlp_bwd: the forward ran with desc.inference set");
    PlanesSaved s; PlanesBwd b;
    layout_planes_saved(p, io.saved, s);
    layout_planes_bwd(p, io.scratch, b);
    int cmaxc = 0;
    for (int l = 0; l < L; ++l) cmaxc = std::max(cmaxc, d.cout[l]);
    float *grad_in = plain ? gr.grad_x : gr.grad_feats;
    PAPC_REQUIRE(!grad_in || s.wtp[0], PAPC_E_INVALID, "papc_sa_mlp_bwd: an input gradient needs desc.want_input_grad at the forward");
    papc_pg_fold_job fold[PAPC_SA_MAX_LAYERS];
    int n_fold = 0;
    const float *dz = nullptr, *red = nullptr;
    int flip = 0;
    for (int l = L - 1; l >= 0; --l) {
        const int cout = d.cout[l], cin = cin_of(p, l);
        const float *cst = s.cst[l];
        const bool acc_w = gr.acc_w[l] != 0, acc_gb = gr.acc_gb[l] != 0;
        float *dgamma = gr.dgamma[l] ? gr.dgamma[l] : b.tmp_gb, *dbeta = gr.dbeta[l] ? gr.dbeta[l] : b.tmp_gb + cmaxc;
        PAPC_REQUIRE(gr.dw[l], PAPC_E_INVALID, "papc_sa_mlp_bwd: dw[%d] is NULL", l);
        const bool need_dx = l > 0 || grad_in;
        // dY of this layer as planes, both orientations; c1 / c2 / dgamma / dbeta folded in the prologue
        papc_pg_prep a;
        memset(&a, 0, sizeof(a));
        a.M = M; a.C = cout; a.x = s.y[l]; a.ldx = cout; a.mean = const_cast<float *>(cst); a.invstd = const_cast<float *>(cst + cout);
        a.scale = const_cast<float *>(cst + 2 * cout); a.shift = const_cast<float *>(cst + 3 * cout);
        a.dgamma = dgamma; a.dbeta = dbeta; a.accumulate = (acc_gb && gr.dgamma[l]) ? 1 : 0; a.planes = need_dx ? b.dyp : nullptr; a.planes_t = b.dypt;
        if (l == L - 1 && !d.pool) {
            // dense upstream gradient: the layer's BN-backward sums in a pass of their own (T partial rows, folded in the prep's prologue)
            SA_CALL(papc_bn_bwd_reduce_f32(PAPC_DZ_DENSE, gr.gout, nullptr, nullptr, 1, s.y[l], cst, cst + cout, cst + 2 * cout, cst + 3 * cout, M, cout, (int)T, b.red[2], st));
            a.mode = PAPC_PG_DY_DENSE; a.dz = gr.gout; a.red = b.red[2]; a.red_parts = (int)T;
        } else if (l == L - 1) {
            a.mode = PAPC_PG_DY_MAX; a.gout = gr.gout; a.ysel = s.ysel; a.argmax = s.argmax; a.K = d.K;
        } else {
            a.mode = PAPC_PG_DY_DENSE; a.dz = dz; a.red = red; a.red_parts = (int)T;
        }
        SA_CALL(papc_pg_prep_rows_f32(&a, st));
        // ---- dX
        float *dz_prev = nullptr, *red_prev = nullptr;
        if (l > 0) {
            const float *pc = s.cst[l - 1];
            dz_prev = b.dz[flip]; red_prev = b.red[flip];
            flip ^= 1;
            papc_pg_gemm g;
            memset(&g, 0, sizeof(g));
            g.epi = PAPC_PG_RED; g.a = b.dyp; g.b = s.wtp[l]; g.R1 = (int)M; g.R2 = cin; g.K = cout;
            g.c = dz_prev; g.ldc = cin; g.split = 1; g.stats = red_prev; g.family = PAPC_K_BWD_DX;
            g.y_prev = s.y[l - 1]; g.mean = pc; g.invstd = pc + cin; g.scale = pc + 2 * cin; g.shift = pc + 3 * cin;
            SA_CALL(papc_pg_gemm_f32(&g, st));
        } else if (grad_in) {
            const int n_in = pg_n_in(p);
            papc_pg_gemm g;
            memset(&g, 0, sizeof(g));
            g.epi = PAPC_PG_STORE; g.a = b.dyp; g.b = s.wtp[0]; g.R1 = (int)M; g.R2 = n_in; g.K = cout;
            g.c = grad_in; g.ldc = n_in; g.split = 1; g.family = PAPC_K_BWD_DX;
            SA_CALL(papc_pg_gemm_f32(&g, st));
        }
        // ---- dW = dY^T . input: contraction over the M rows, split over workgroups, partials folded at the end
        const int split = pg_split_for(cout, cin, (int)(M / 32));
        papc_pg_gemm g;
        memset(&g, 0, sizeof(g));
        g.epi = PAPC_PG_STORE; g.a = b.dypt; g.b = s.PT[l]; g.R1 = cout; g.R2 = cin; g.K = (int)M;
        g.c = b.part[l]; g.ldc = cin; g.split = split; g.split_stride = (int64_t)cout * cin; g.family = PAPC_K_BWD_DW;
        SA_CALL(papc_pg_gemm_f32(&g, st));
        if (!defer_fold(gr, b.part[l], split, (int64_t)cout * cin, 1, cout * cin, gr.dw[l], (int64_t)cout * cin, acc_w ? 1 : 0))
            fold[n_fold++] = papc_pg_fold_job{b.part[l], split, (int64_t)cout * cin, (int64_t)cout * cin, gr.dw[l], acc_w ? 1 : 0};
        if (gr.db[l] && !acc_w) SA_CALL(papc_fill_f32(gr.db[l], cout, 0.f, st));      // (a bias feeding a train-mode BN: gradient exactly 0)
        if (l > 0) { dz = dz_prev; red = red_prev; }
    }
    for (int j0 = 0; j0 < n_fold; j0 += 8) SA_CALL(papc_pg_fold_f32(fold + j0, std::min(8, n_fold - j0), st));
    return PAPC_OK;
}

static void fill_grp(papc_group_src &g, const papc_sa_desc &d, const papc_sa_io &io, bool compact)
{
    memset(&g, 0, sizeof(g));
    g.xyz = io.xyz; g.sb = io.sb; g.sn = io.sn; g.sc = io.sc; g.new_xyz = io.new_xyz; g.feats = io.feats; g.idx = io.idx;
    g.N = d.N; g.S = d.S; g.K = d.K; g.D = d.D; g.xyz_first = d.xyz_first;
    if (compact && io.compact) { g.cidx = io.compact->cidx; g.seg_grp = io.compact->seg_grp; g.rows_dev = io.compact->rows; }
    g.plists = io.plists;     // (papc_lingather_bwd_f32 uses them only when they index the row layout this stack runs)
}

__global__ void mul_vec_kernel(const float *__restrict__ a, const float *__restrict__ b, int n, float *__restrict__ out, int accumulate)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = accumulate ? out[i] + a[i] * b[i] : a[i] * b[i];
}

}  // namespace papc

using namespace papc;

extern "C" {

int papc_sa_mlp_plan(const papc_sa_desc *desc, const papc_sa_io *io, papc_sa_plan *plan)
{
    PAPC_REQUIRE(desc && plan, PAPC_E_INVALID, "papc_sa_mlp_plan: null pointer");
    const papc_sa_desc &d = *desc;
    PAPC_REQUIRE(d.n_layers >= 1 && d.n_layers <= PAPC_SA_MAX_LAYERS, PAPC_E_UNSUPPORTED, "papc_sa_mlp_plan: %d layers (1..%d)", d.n_layers, PAPC_SA_MAX_LAYERS);
    PAPC_REQUIRE(d.B >= 1 && d.N >= 1 && d.S >= 1 && d.K >= 1 && d.D >= 0, PAPC_E_INVALID, "papc_sa_mlp_plan: bad B/N/S/K/D");
    const int64_t M = rows_of(d);
    PAPC_REQUIRE(M < (1ll << 31), PAPC_E_UNSUPPORTED, "papc_sa_mlp_plan: >= 2^31 rows");
    for (int l = 0; l < d.n_layers; ++l) PAPC_REQUIRE(d.cout[l] >= 1, PAPC_E_INVALID, "papc_sa_mlp_plan: cout[%d] = %d", l, d.cout[l]);
    memset(plan, 0, sizeof(*plan));
    plan->d = d;
    const int L = d.n_layers;
    const bool plain = d.input == PAPC_SA_IN_ROWS, ev = d.eval_bn != 0;
    plan->cin0 = plain ? d.cin : d.D + 3;
    PAPC_REQUIRE(plan->cin0 >= 1, PAPC_E_INVALID, "papc_sa_mlp_plan: no input channels");
    const bool has_idx = !plain && d.identity_rows == 0, has_feats = !plain && d.D > 0;
    {   // few rows in groups of exactly 128 (one GEMM tile), no neighbour index, widths in multiples of 8: the planes kernels (smallm.hip)
        bool ok = !(d.disable & PAPC_SA_NO_PLANES) && !ev && !has_idx && M % PG_GROUP == 0 && M >= PG_GROUP && M <= PG_MAX_ROWS && L >= 2;
        ok = ok && (d.pool ? (d.K % PG_GROUP == 0 && d.K / PG_GROUP <= 64) : (plain && !(d.disable & PAPC_SA_NO_PLANES_POINTWISE)));
        for (int l = 0; l < L; ++l) ok = ok && d.cout[l] % 8 == 0;
        ok = ok && (plain ? plan->cin0 % 8 == 0 : d.S == 1);
        plan->planes = ok ? 1 : 0;
    }
    if (plan->planes) {
        PlanesSaved ps; PlanesFwd pf; PlanesBwd pb;
        char *fake = reinterpret_cast<char *>((uintptr_t)1 << 40);
        layout_planes_saved(*plan, fake, ps);
        for (int l = 0; l < PAPC_SA_MAX_LAYERS; ++l) {
            plan->off_y[l] = l < L ? (int64_t)(reinterpret_cast<char *>(ps.y[l]) - fake) : -1;
            plan->off_cst[l] = l < L ? (int64_t)(reinterpret_cast<char *>(ps.cst[l]) - fake) : -1;
        }
        plan->off_argmax = ps.argmax ? (int64_t)(reinterpret_cast<char *>(ps.argmax) - fake) : -1;
        plan->saved_bytes = (int64_t)layout_planes_saved(*plan, nullptr, ps) + 256;
        plan->fwd_scratch_bytes = (int64_t)layout_planes_fwd(*plan, nullptr, pf) + 256;
        plan->bwd_scratch_bytes = (int64_t)layout_planes_bwd(*plan, nullptr, pb) + 256;
        return PAPC_OK;
    }
    plan->lin0 = !(d.disable & PAPC_SA_NO_LINGATHER) && !ev && !plain && has_idx && has_feats && L >= 2 && d.D % 4 == 0 && d.D >= 16 &&
                 d.cout[0] % 4 == 0 && d.cout[0] <= 256;
    plan->xyz1 = !(d.disable & PAPC_SA_NO_XYZ1) && !ev && !plain && has_idx && !has_feats && d.D == 0 && L >= 3 &&
                 papc_mlp_xyz_ok(M, d.cout[0], d.cout[1]);
    plan->compact = plan->lin0 && d.pool && !(d.disable & PAPC_SA_NO_COMPACT) && io && io->compact && papc_mlp_compact_ok(M, d.K, L, d.cout);
    const int cL = d.cout[L - 1], cLi = L > 1 ? d.cout[L - 2] : plan->cin0;
    plan->gmax = !plan->compact && d.pool && !(d.disable & PAPC_SA_NO_GMAX) && !ev && papc_mlp_gemm_gmax_ok(M, cL, d.K);
    plan->nostore = plan->gmax && !(d.disable & PAPC_SA_NO_NOSTORE) && L >= 2 && !(L == 2 && plan->xyz1) && papc_mlp_max_nostore_ok(M, cLi, cL, d.K);
    plan->sparse_max = plan->nostore;
    SavedPtrs s; FwdPtrs f; BwdPtrs b;
    {   // offsets for the caller: lay `saved` out on a fake base
        char *fake = reinterpret_cast<char *>((uintptr_t)1 << 40);
        layout_saved(*plan, fake, s);
        for (int l = 0; l < PAPC_SA_MAX_LAYERS; ++l) {
            plan->off_y[l] = (l < L && s.y[l]) ? (int64_t)(reinterpret_cast<char *>(s.y[l]) - fake) : -1;
            plan->off_cst[l] = (l < L && s.cst[l]) ? (int64_t)(reinterpret_cast<char *>(s.cst[l]) - fake) : -1;
        }
        plan->off_argmax = s.argmax ? (int64_t)(reinterpret_cast<char *>(s.argmax) - fake) : -1;
    }
    plan->saved_bytes = (int64_t)layout_saved(*plan, nullptr, s) + 256;
    plan->fwd_scratch_bytes = (int64_t)layout_fwd(*plan, nullptr, f) + 256;
    plan->bwd_scratch_bytes = (int64_t)layout_bwd(*plan, nullptr, b) + 256;
    return PAPC_OK;
}

int papc_sa_mlp_fwd(const papc_sa_plan *plan, const papc_sa_io *io, papc_stream_t st)
{
    PAPC_REQUIRE(plan && io && io->out && io->saved && io->scratch, PAPC_E_INVALID, "papc_sa_mlp_fwd: null pointer");
    const papc_sa_plan &p = *plan;
    const papc_sa_desc &d = p.d;
    const int L = d.n_layers;
    const int64_t M = rows_of(d), G = (int64_t)d.B * d.S;
    const bool plain = d.input == PAPC_SA_IN_ROWS, ev = d.eval_bn != 0;
    PAPC_REQUIRE(plain ? io->x_rows != nullptr : (io->xyz && io->new_xyz), PAPC_E_INVALID, "papc_sa_mlp_fwd: missing input");
    PAPC_REQUIRE(plain || d.identity_rows || io->idx, PAPC_E_INVALID, "papc_sa_mlp_fwd: grouped input without idx (set identity_rows for sample_and_group_all)");
    PAPC_REQUIRE(plain || d.D == 0 || io->feats, PAPC_E_INVALID, "papc_sa_mlp_fwd: D = %d but feats is NULL", d.D);
    PAPC_REQUIRE(!p.compact || io->compact, PAPC_E_INVALID, "papc_sa_mlp_fwd: the plan is compacted but io->compact is NULL");
    if (p.planes) {
        for (int l = 0; l < L; ++l) PAPC_REQUIRE(io->layer[l].w && io->layer[l].gamma && io->layer[l].beta, PAPC_E_INVALID, "papc_sa_mlp_fwd: layer %d lacks w / gamma / beta", l);
        return planes_fwd(p, *io, st);
    }
    SavedPtrs s; FwdPtrs f;
    layout_saved(p, io->saved, s);
    layout_fwd(p, io->scratch, f);
    const int parts = papc_mlp_gemm_parts(M);
    // compacted stack: the producers weigh their statistics with the rows' multiplicities themselves (papc_mlp_gemm_rows_w_f32, papc_group_src.wstat);
    // PAPC_SA_NO_WSTATS: unweighted sums + one papc_bn_stats_corr_f32 launch per layer
    const bool wstats = p.compact && !(d.disable & PAPC_SA_NO_WSTATS);
    const int R_c = (p.compact && !wstats) ? papc_compact_corr_parts() : 0;
    papc_group_src grp;
    if (!plain) fill_grp(grp, d, *io, p.compact);
    if (!plain && wstats) grp.wstat = io->compact->wrow;
    int cmax = 0;
    for (int l = 0; l < L; ++l) cmax = std::max(cmax, d.cout[l]);
    const size_t stats_stride = (size_t)(std::max(parts, p.lin0 ? papc_lingather_parts(M) : 0) + R_c) * 2 * cmax;

    const float *prev_y = nullptr, *prev_sc = nullptr, *prev_sh = nullptr;
    int cin = p.cin0;
    const float *xc = nullptr;
    papc_group_max gm;
    memset(&gm, 0, sizeof(gm));
    for (int l = 0; l < L; ++l) {
        const papc_sa_layer &ly = io->layer[l];
        PAPC_REQUIRE(ly.w && ly.gamma && ly.beta, PAPC_E_INVALID, "papc_sa_mlp_fwd: layer %d lacks w / gamma / beta", l);
        PAPC_REQUIRE(!ev || (ly.running_mean && ly.running_var), PAPC_E_INVALID, "papc_sa_mlp_fwd: eval_bn needs the running statistics");
        const int cout = d.cout[l];
        float *stats = ev ? nullptr : f.stats + (size_t)l * stats_stride;
        int parts_l = parts;
        const papc_group_max *gm_ref = nullptr;
        if (l == L - 1 && p.gmax) {
            gm.gmax = s.gbuf_f; gm.gmin = s.gbuf_f + G * cout; gm.amax = f.gbuf_i; gm.amin = f.gbuf_i + G * cout; gm.K = d.K;
            gm.sign_src = (d.disable & PAPC_SA_NO_GSIGN) ? nullptr : ly.gamma;      // (one extremum per channel in the row-streaming kernel's epilogue)
            gm_ref = &gm;
        }
        float *y = s.y[l];
        float *cst = s.cst[l];
        if (l == 0 && p.xyz1) {
            const double *gpart1;
            if (io->xc && io->xc_gram) { xc = io->xc; gpart1 = io->xc_gram; }
            else {
                SA_CALL(papc_xyz_group_f32(&grp, d.B, s.xc, f.gpart + 16, st));
                SA_CALL(papc_xyz_gram_fold_f32(f.gpart + 16, papc_xyz_parts(M), f.gpart, st));
                xc = s.xc; gpart1 = f.gpart;
            }
            SA_CALL(papc_xyz_l1_finalize_f32(gpart1, 1, M, ly.w, cin, 0, ly.b, ly.gamma, ly.beta, d.eps, d.momentum, cout, cst, cst + cout, cst + 2 * cout,
                                             cst + 3 * cout, ly.running_mean, ly.running_var, s.wf, s.gram, st));
            prev_y = nullptr; prev_sc = cst + 2 * cout; prev_sh = cst + 3 * cout;
            cin = cout;
            continue;
        }
        if (l == 1 && p.xyz1) {
            SA_CALL(papc_mlp_gemm_f32(A_XYZ_, xc, 4, nullptr, s.wf, nullptr, ly.w, ly.b, M, cin, cout, y, stats, gm_ref, st));
        } else if (l == 0 && plain) {
            SA_CALL(papc_mlp_gemm_f32(A_PLAIN_, io->x_rows, cin, nullptr, nullptr, nullptr, ly.w, ly.b, M, cin, cout, y, stats, gm_ref, st));
        } else if (l == 0 && p.lin0) {
            // W_f feats_j depends on the source point only: one [B*N, D] x [D, cout] product on the feature block of the weight, then a gather-add
            const int fcol0 = d.xyz_first ? 3 : 0;
            const float *wfeat = io->wfeat;       // (the caller's copy, made with the step's weight transposes: one launch less on the chain)
            if (!wfeat) {
                SA_CALL(papc_copy2d_f32(ly.w + fcol0, cin, f.wfeat, d.D, cout, d.D, 0, st));
                wfeat = f.wfeat;
            }
            const int64_t BN = (int64_t)d.B * d.N;
            SA_CALL(papc_mlp_gemm_f32(A_PLAIN_, io->feats, d.D, nullptr, nullptr, nullptr, wfeat, nullptr, BN, d.D, cout, s.P, nullptr, nullptr, st));
            parts_l = papc_lingather_parts(M);
            SA_CALL(papc_lingather_fwd_f32(s.P, &grp, d.B, ly.w, cin, d.xyz_first ? 0 : d.D, ly.b, cout, y, stats, st));
        } else if (l == 0) {
            SA_CALL(papc_mlp_gemm_f32(A_GROUP_, nullptr, 0, &grp, nullptr, nullptr, ly.w, ly.b, M, cin, cout, y, stats, gm_ref, st));
        } else if (p.compact) {
            SA_CALL(papc_mlp_gemm_rows_w_f32(A_BNRELU_, prev_y, cin, nullptr, prev_sc, prev_sh, ly.w, ly.b, M, cin, cout, y, stats, nullptr, io->compact->rows,
                                             wstats ? io->compact->wrow : nullptr, st));
        } else {
            SA_CALL(papc_mlp_gemm_f32(A_BNRELU_, prev_y, cin, nullptr, prev_sc, prev_sh, ly.w, ly.b, M, cin, cout, y, stats, gm_ref, st));
        }
        if (p.compact && !wstats) {      // what the copies add to this layer's statistics: extra partial rows behind the kernel's own
            SA_CALL(papc_bn_stats_corr_f32(y, cout, io->compact->start, io->compact->coef, io->compact->G, stats + (size_t)parts_l * 2 * cout, st));
            parts_l += R_c;
        }
        if (ev) SA_CALL(papc_bn_eval_consts_f32(ly.running_mean, ly.running_var, ly.gamma, ly.beta, d.eps, cout, cst, cst + cout, cst + 2 * cout, cst + 3 * cout, st));
        else SA_CALL(papc_bn_finalize_f32(stats, parts_l, M, cout, ly.gamma, ly.beta, d.eps, d.momentum, cst, cst + cout, cst + 2 * cout, cst + 3 * cout,
                                          ly.running_mean, ly.running_var, st));
        prev_y = y; prev_sc = cst + 2 * cout; prev_sh = cst + 3 * cout;
        cin = cout;
    }
    if (!d.pool) return papc_bn_relu_f32(prev_y, prev_sc, prev_sh, M, cin, io->out, st);
    if (p.compact) return papc_bn_relu_max_seg_f32(prev_y, cin, io->compact->start, prev_sc, prev_sh, (int)G, io->out, s.argmax, s.gbuf_f, st);
    if (p.gmax) return papc_bn_select_max_f32(s.gbuf_f, s.gbuf_f + G * cin, f.gbuf_i, f.gbuf_i + G * cin, prev_sc, prev_sh, G, cin, io->out, s.argmax, st);
    return papc_bn_relu_max_f32(prev_y, prev_sc, prev_sh, G, d.K, cin, io->out, s.argmax, st);
}

int papc_sa_mlp_bwd(const papc_sa_plan *plan, const papc_sa_io *io, const papc_sa_grads *gr, papc_stream_t st)
{
    PAPC_REQUIRE(plan && io && gr && gr->gout && io->saved && io->scratch, PAPC_E_INVALID, "papc_sa_mlp_bwd: null pointer");
    const papc_sa_plan &p = *plan;
    const papc_sa_desc &d = p.d;
    const int L = d.n_layers;
    const int64_t M = rows_of(d), G = (int64_t)d.B * d.S;
    const bool plain = d.input == PAPC_SA_IN_ROWS, ev = d.eval_bn != 0;
    if (p.planes) return planes_bwd(p, *io, *gr, st);
    SavedPtrs s; BwdPtrs b;
    layout_saved(p, io->saved, s);
    layout_bwd(p, io->scratch, b);
    const float *xc = p.xyz1 ? ((io->xc && io->xc_gram) ? io->xc : s.xc) : nullptr;
    papc_group_src grp;
    if (!plain) fill_grp(grp, d, *io, p.compact);
    const int n_parts = (int)std::min<int64_t>(512, (M + 127) / 128);
    const int gemm_parts = papc_mlp_gemm_parts(M);
    const bool x_needs = plain && gr->grad_x, f_needs = !plain && d.D > 0 && gr->grad_feats && !d.cut_gather_grad;
    const float *ysel = (p.gmax || p.compact) ? s.gbuf_f : nullptr;
    const float *ones = io->consts3, *zeros = nullptr, *big = nullptr;
    int cmaxc = 0;
    for (int l = 0; l < L; ++l) cmaxc = std::max(cmaxc, d.cout[l]);
    if (p.lin0) {
        PAPC_REQUIRE(io->consts3 && io->consts3_ld >= d.cout[0], PAPC_E_INVALID, "papc_sa_mlp_bwd: the gather-add first layer needs consts3 (ones | zeros | 1e30 rows of >= %d floats)", d.cout[0]);
        zeros = io->consts3 + io->consts3_ld; big = io->consts3 + 2 * (int64_t)io->consts3_ld;
    }

    // ---- W^T operands of the dX GEMMs: from the caller's table, else one batched transpose
    bool need_wt[PAPC_SA_MAX_LAYERS];
    const float *wts[PAPC_SA_MAX_LAYERS];
    for (int l = 0; l < L; ++l) {
        need_wt[l] = l > 0 || x_needs || f_needs;
        wts[l] = nullptr;
    }
    if (p.sparse_max) need_wt[L - 1] = false;
    if (p.lin0) need_wt[0] = false;
    {
        const float *srcs[8]; float *dsts[8]; int rws[8], cls[8];
        int n = 0;
        for (int l = 0; l < L; ++l) {
            if (!need_wt[l]) continue;
            if (gr->wt[l]) { wts[l] = gr->wt[l]; continue; }
            srcs[n] = io->layer[l].w; dsts[n] = b.wt[l]; rws[n] = d.cout[l]; cls[n] = cin_of(p, l);
            wts[l] = b.wt[l];
            if (++n == 8) { SA_CALL(papc_transpose_batch_f32(srcs, dsts, rws, cls, n, st)); n = 0; }
        }
        if (n) SA_CALL(papc_transpose_batch_f32(srcs, dsts, rws, cls, n, st));
    }

    papc_reduce_job jobs[PAPC_SA_MAX_LAYERS];
    int n_jobs = 0;
    bool xyz_fused = false;
    bool dz0_masked = false;      // the dX launch of layer 1 stored the gather-add layer's dz with its ReLU mask applied (papc_bwd_red.store_masked)
    const float *dz = nullptr;
    const float *fused_red = nullptr;
    int flip = 0;
    for (int l = L - 1; l >= 0; --l) {
        const papc_sa_layer &ly = io->layer[l];
        const int cout = d.cout[l], cin = cin_of(p, l);
        const float *cst = s.cst[l];
        float *c12 = b.c12[l];
        const bool acc_w = gr->acc_w[l] != 0, acc_gb = gr->acc_gb[l] != 0 && !ev;
        float *dgamma = gr->dgamma[l] ? gr->dgamma[l] : b.tmp_gb, *dbeta = gr->dbeta[l] ? gr->dbeta[l] : b.tmp_gb + cmaxc;
        PAPC_REQUIRE(gr->dw[l], PAPC_E_INVALID, "papc_sa_mlp_bwd: dw[%d] is NULL", l);
        papc_bwd_dy dy;
        memset(&dy, 0, sizeof(dy));
        if (p.compact) { dy.wrow = io->compact->wrow; dy.seg_grp = io->compact->seg_grp; dy.rows_dev = io->compact->rows; }
        if (l == L - 1 && !d.pool) { dy.dz_mode = PAPC_DZ_DENSE; dy.dz = gr->gout; dy.K = 1; }
        else if (l == L - 1) { dy.dz_mode = PAPC_DZ_MAX; dy.gout = gr->gout; dy.argmax = s.argmax; dy.K = d.K; }
        else { dy.dz_mode = PAPC_DZ_DENSE; dy.dz = dz; dy.K = 1; }
        if (l == 0 && p.xyz1) {
            // the whole backward of the coordinates-only layer from one pass over dz and the inputs' moments
            PAPC_REQUIRE(!gr->dgamma[l] || (gr->acc_gb[l] != 0) == (gr->acc_w[l] != 0), PAPC_E_UNSUPPORTED,
                         "papc_sa_mlp_bwd: the coordinates-only first layer takes ONE accumulate flag for dw / dgamma / dbeta");
            int nbp = papc_xyz_bwd_parts(M);
            if (xyz_fused) nbp = papc_mlp_gemm_parts(M);      // (the partial sums came out of the layer above's dX kernel)
            else SA_CALL(papc_xyz_l1_bwd_f32(dz, xc, s.wf, M, cout, b.bpart, st));
            SA_CALL(papc_xyz_l1_bwd_finalize_f32(b.bpart, nbp, M, cout, s.gram, ly.w, cin, 0, ly.b, cst, cst + cout, cst + 2 * cout, dgamma, dbeta, gr->dw[l],
                                                 acc_w ? 1 : 0, st));
            if (gr->db[l] && !acc_w) SA_CALL(papc_fill_f32(gr->db[l], cout, 0.f, st));
            break;
        }
        dy.y = s.y[l];
        dy.mean = cst; dy.invstd = cst + cout; dy.scale = cst + 2 * cout; dy.shift = cst + 3 * cout;
        dy.c1 = c12; dy.c2 = c12 + cout;
        const float *red;
        int red_parts;
        if (!fused_red) {     // (sum p, sum p*xhat): separate pass, unless the dX kernel of layer l+1 already produced it
            red = b.red; red_parts = n_parts;
            if (p.sparse_max && l == L - 1)
                SA_CALL(papc_bn_bwd_reduce_max_f32(ysel, dy.gout, dy.K, dy.mean, dy.invstd, dy.scale, dy.shift, M, cout, n_parts, b.red, b.psel, st));
            else if (p.compact && l == L - 1 && dy.dz_mode == PAPC_DZ_MAX && ysel && b.psel && !(d.disable & PAPC_SA_NO_PSEL)) {
                // the same reduction, which also leaves scale * p per (group, channel): the compacted dX kernel streams that instead of gout and
                // repeats neither the ReLU test nor the scale on every row (bit-identical: the same product of the same two floats)
                SA_CALL(papc_bn_bwd_reduce_max_f32(ysel, dy.gout, dy.K, dy.mean, dy.invstd, dy.scale, dy.shift, M, cout, n_parts, b.red, b.psel, st));
                dy.psel = b.psel;
            } else
                SA_CALL(papc_bn_bwd_reduce_f32(dy.dz_mode, dy.dz_mode == PAPC_DZ_MAX ? ysel : dy.dz, dy.gout, dy.argmax, dy.K, dy.y, dy.mean, dy.invstd, dy.scale,
                                               dy.shift, M, cout, n_parts, b.red, st));
        } else { red = fused_red; red_parts = gemm_parts; }
        SA_CALL(papc_bn_bwd_finalize_f32(red, red_parts, M, cout, dgamma, dbeta, c12, c12 + cout, (acc_gb ? 1 : 0) | (ev ? 2 : 0), st));
        if (l == 0 && p.lin0) {
            // G[j] = sum of the dY rows that gathered point j (+ the xyz columns of dW, streamed); the D-wide products run on B*N rows
            const int64_t BN = (int64_t)d.B * d.N;
            const int parts_l = papc_lingather_bwd_parts(&grp, d.B, cout);
            // (with the grouping's point lists every row of G is written in a fixed summation order: no atomics, nothing to pre-zero)
            if (!papc_lingather_bwd_lists_ok(&grp, cout)) SA_CALL(papc_fill_f32(b.Gs, BN * cout, 0.f, st));
            // (dz masked by the dX launch above + the point lists: the backward that gathers dz alone -- y's share comes from P and the lists' moments)
            if (dz0_masked && papc_lingather_bwd_pp_ok(&grp, d.B, cout))
                SA_CALL(papc_lingather_bwd_pp_f32(&dy, &grp, d.B, cout, s.P, ly.w, cin, d.xyz_first ? 0 : d.D, ly.b, b.Gs, b.dwx_part, st));
            else
                SA_CALL(papc_lingather_bwd_f32(&dy, &grp, d.B, cout, b.Gs, b.dwx_part, st));
            const int fcol0 = d.xyz_first ? 3 : 0, xcol0 = d.xyz_first ? 0 : d.D;
            float *dw = gr->dw[l];
            const int acc = acc_w ? 1 : 0;
            if (!defer_fold(*gr, b.dwx_part, parts_l, (int64_t)cout * 3, cout, 3, dw + xcol0, cin, acc))
                SA_CALL(papc_reduce_partials_strided_f32(b.dwx_part, parts_l, (int64_t)cout * 3, cout, 3, dw + xcol0, cin, acc, st));
            // dW_f = G^T feats on the library's own dW kernel: G plays dY with BN constants that make dY = dz (scale 1, shift huge, c1 = c2 = 0)
            papc_bwd_dy dyg;
            memset(&dyg, 0, sizeof(dyg));
            dyg.dz_mode = PAPC_DZ_DENSE; dyg.dz = b.Gs; dyg.K = 1; dyg.y = b.Gs;
            dyg.mean = zeros; dyg.invstd = ones; dyg.scale = ones; dyg.shift = big; dyg.c1 = zeros; dyg.c2 = zeros;
            const int rpc_g = dw_rows_per_chunk(BN, cout, d.D, 64);
            const int n_chunks_g = (int)((BN + rpc_g - 1) / rpc_g);
            const int64_t pld_g = (int64_t)cout * d.D + cout;
            SA_CALL(papc_mlp_bwd_dw_f32(&dyg, A_PLAIN_, io->feats, d.D, nullptr, nullptr, nullptr, BN, d.D, cout, rpc_g, b.part_g, b.part_g + (int64_t)cout * d.D, pld_g, st));
            if (!defer_fold(*gr, b.part_g, n_chunks_g, pld_g, cout, d.D, dw + fcol0, cin, acc))
                SA_CALL(papc_reduce_partials_strided_f32(b.part_g, n_chunks_g, pld_g, cout, d.D, dw + fcol0, cin, acc, st));
            if (gr->db[l] && !acc_w) SA_CALL(papc_fill_f32(gr->db[l], cout, 0.f, st));      // (a bias feeding a train-mode BN: gradient exactly 0)
            if (f_needs) {
                const float *wft;
                if (gr->wt[l]) wft = gr->wt[l] + (int64_t)fcol0 * cout;     // rows of the precomputed W^T [cin, cout]: the feature block, contiguous
                else {
                    SA_CALL(papc_copy2d_f32(ly.w + fcol0, cin, b.wft, cout, cout, d.D, 1, st));
                    wft = b.wft;
                }
                SA_CALL(papc_mlp_gemm_f32(A_PLAIN_, b.Gs, cout, nullptr, nullptr, nullptr, wft, nullptr, BN, cout, d.D, gr->grad_feats, nullptr, nullptr, st));
            }
            break;
        }
        if (p.sparse_max && l == L - 1) {
            SA_CALL(papc_bn_max_prep_f32(nullptr, nullptr, cst + 2 * cout, cst + 3 * cout, cst, cst + cout, c12, c12 + cout, ly.w, ly.b, G, cout, cin, nullptr,
                                         b.wcat, b.hb, b.eq, b.eq + cout, st));     // (psel: written by the reduction above)
        }
        const bool x1 = p.xyz1 && l == 1;     // the input of this layer is the recomputed activation of the coordinates-only first layer
        // ---- dW, db
        if (p.nostore && l == L - 1) {
            const float *pc = s.cst[l - 1];
            SA_CALL(papc_mlp_bwd_dw_max_f32(b.psel, s.argmax, d.K, s.y[l - 1], pc + 2 * cin, pc + 3 * cin, ly.w, b.eq, cst + 2 * cout, c12, M, cin, cout, b.dwmax_ws,
                                            gr->dw[l], acc_w ? 1 : 0, st));
            if (gr->db[l] && !acc_w) SA_CALL(papc_fill_f32(gr->db[l], cout, 0.f, st));
        } else {
            const int am = l == 0 ? (plain ? A_PLAIN_ : A_GROUP_) : (x1 ? A_XYZ_ : A_BNRELU_);
            const int rpc = dw_chunk(p, l, am, dy.dz_mode);
            const int n_chunks = (int)((M + rpc - 1) / rpc);
            const int64_t pld = (int64_t)cout * cin + cout;
            float *dwp = b.part[l], *dbp = b.part[l] + (int64_t)cout * cin;
            if (l == 0 && plain) SA_CALL(papc_mlp_bwd_dw_f32(&dy, A_PLAIN_, io->x_rows, cin, nullptr, nullptr, nullptr, M, cin, cout, rpc, dwp, dbp, pld, st));
            else if (l == 0) SA_CALL(papc_mlp_bwd_dw_f32(&dy, A_GROUP_, nullptr, 0, &grp, nullptr, nullptr, M, cin, cout, rpc, dwp, dbp, pld, st));
            else if (x1) SA_CALL(papc_mlp_bwd_dw_f32(&dy, A_XYZ_, xc, 4, nullptr, s.wf, nullptr, M, cin, cout, rpc, dwp, dbp, pld, st));
            else {
                const float *pc = s.cst[l - 1];
                SA_CALL(papc_mlp_bwd_dw_f32(&dy, A_BNRELU_, s.y[l - 1], cin, nullptr, pc + 2 * cin, pc + 3 * cin, M, cin, cout, rpc, dwp, dbp, pld, st));
            }
            // the partials of all layers are folded in ONE launch once the stack's last dW kernel is enqueued
            const bool want_db = gr->db[l] && !ev;
            if (defer_room(*gr, 2)) {
                defer_fold(*gr, b.part[l], n_chunks, pld, 1, cout * cin, gr->dw[l], (int64_t)cout * cin, acc_w ? 1 : 0);
                if (want_db) defer_fold(*gr, b.part[l] + (int64_t)cout * cin, n_chunks, pld, 1, cout, gr->db[l], cout, acc_w ? 1 : 0);
            } else {
                papc_reduce_job &j = jobs[n_jobs++];
                j.partial = b.part[l]; j.n_chunks = n_chunks; j.accumulate = acc_w ? 1 : 0; j.ld = pld; j.n1 = (int64_t)cout * cin; j.out1 = gr->dw[l];
                j.n2 = want_db ? cout : 0; j.out2 = want_db ? gr->db[l] : nullptr;
            }
            if (ev && gr->db[l]) {     // no batch-mean term removes the bias direction: db = sum_m dy = scale * sum_m p
                hipLaunchKernelGGL(mul_vec_kernel, dim3((unsigned)cdiv(cout, 256)), dim3(256), 0, as_stream(st), cst + 2 * cout, dbeta, cout, gr->db[l], acc_w ? 1 : 0);
                SA_CALL(check_launch("papc_sa_mlp_bwd (eval-mode bias gradient)"));
            }
        }
        // ---- dX
        fused_red = nullptr;
        if (l > 0) {
            float *dz_prev = b.dz[flip];
            flip ^= (b.dz[1] ? 1 : 0);
            papc_bwd_red nr;
            const papc_bwd_red *nr_ref = nullptr;
            if (!(d.disable & PAPC_SA_NO_FUSED_RED) && !x1) {     // (x1: the layer below takes its BN-backward sums from its own pass over dz)
                const float *pc = s.cst[l - 1];
                nr.y = s.y[l - 1]; nr.mean = pc; nr.invstd = pc + cin; nr.scale = pc + 2 * cin; nr.shift = pc + 3 * cin;
                float *fr = b.fused_red[l & 1];
                nr.red_partial = fr;
                nr.store_masked = 0;
                if (l == 1 && p.lin0 && papc_lingather_bwd_pp_ok(&grp, d.B, p.d.cout[0])) {
                    nr.store_masked = 1;      // the gather-add layer below then gathers dz alone (papc_lingather_bwd_pp_f32)
                    dz0_masked = true;
                }
                nr_ref = &nr;
                fused_red = fr;
            }
            if (p.sparse_max && l == L - 1) {
                const float *pc = s.cst[l - 1];
                SA_CALL(papc_mlp_bwd_dx_max_f32(b.psel, s.argmax, d.K, s.y[l - 1], cin, pc + 2 * cin, pc + 3 * cin, b.wcat, b.hb, M, cin, cout, dz_prev, nr_ref, st));
            } else {
                if (x1 && !(d.disable & PAPC_SA_NO_XYZ_FUSE) && dy.dz_mode == PAPC_DZ_DENSE && !dy.wrow && papc_mlp_bwd_dx_xyz_ok(M, cin, cout)) {
                    // the coordinates-only layer below needs four sums per channel of this dX, not dX: the kernel folds them, nothing is stored
                    const int rc = papc_mlp_bwd_dx_xyz_f32(&dy, wts[l], M, cin, cout, xc, s.wf, b.bpart, st);
                    if (rc == PAPC_OK) xyz_fused = true;
                    else if (rc != PAPC_E_UNSUPPORTED) return rc;      // (declined, e.g. an operand off a 16-byte boundary: the stored form below)
                }
                if (!xyz_fused) SA_CALL(papc_mlp_bwd_dx_f32(&dy, wts[l], M, cin, cout, dz_prev, nullptr, nr_ref, st));
            }
            dz = dz_prev;
        } else if (x_needs) {
            SA_CALL(papc_mlp_bwd_dx_f32(&dy, wts[l], M, cin, cout, gr->grad_x, nullptr, nullptr, st));
        } else if (f_needs) {
            SA_CALL(papc_fill_f32(gr->grad_feats, (int64_t)d.B * d.N * d.D, 0.f, st));
            papc_scatter_dst sc;
            memset(&sc, 0, sizeof(sc));
            sc.grad_feats = gr->grad_feats; sc.idx = io->idx; sc.N = d.N; sc.S = d.S; sc.K = d.K; sc.D = d.D; sc.col0 = d.xyz_first ? 3 : 0;
            SA_CALL(papc_mlp_bwd_dx_f32(&dy, wts[l], M, cin, cout, nullptr, &sc, nullptr, st));
        }
    }
    if (n_jobs) SA_CALL(papc_reduce_partials_batch_f32(jobs, n_jobs, st));
    return PAPC_OK;
}

/* ---- PillarFeatureNet with its single (last) PFNLayer: pillars.py:79-108 over PFNLayer :29-37 -- decorate, mask, Linear(9 -> C, no bias),
 * train-mode BatchNorm1D, ReLU, max over the T points of a pillar -- in ONE call per direction (the Gram-matrix path of csrc/pfn.hip) */
static void pfn_layout(const papc_pfn_desc &d, void *saved, void *scratch, float *&cst, int32_t *&argmax, double *&gram, double *&gpart, float *&part, float *&sums)
{
    Carver cs{reinterpret_cast<char *>(saved), 0};
    cst = cs.take<float>(4 * (size_t)d.C);
    argmax = cs.take<int32_t>((size_t)d.P * d.C);
    gram = cs.take<double>(256);
    Carver cw{reinterpret_cast<char *>(scratch), 0};
    gpart = cw.take<double>((size_t)papc_pfn_gram_blocks(d.P) * 256);
    part = cw.take<float>((size_t)papc_pfn_num_blocks(d.P) * 11 * d.C);
    sums = cw.take<float>(11 * (size_t)d.C);
}

int papc_pfn_workspace(const papc_pfn_desc *desc, int64_t *saved_bytes, int64_t *scratch_bytes)
{
    PAPC_REQUIRE(desc && saved_bytes && scratch_bytes, PAPC_E_INVALID, "papc_pfn_workspace: null pointer");
    PAPC_REQUIRE(desc->P >= 1 && desc->T >= 1 && desc->C >= 1 && desc->C <= 64, PAPC_E_UNSUPPORTED, "papc_pfn_workspace: P=%d T=%d C=%d (C in 1..64)", desc->P, desc->T, desc->C);
    const papc_pfn_desc &d = *desc;
    *saved_bytes = 256 + ((4 * (int64_t)d.C * 4 + 255) & ~255ll) + (((int64_t)d.P * d.C * 4 + 255) & ~255ll) + 2048;
    *scratch_bytes = 256 + (((int64_t)papc_pfn_gram_blocks(d.P) * 2048 + 255) & ~255ll) + (((int64_t)papc_pfn_num_blocks(d.P) * 11 * d.C * 4 + 255) & ~255ll) + 11 * d.C * 4 + 256;
    return PAPC_OK;
}

int papc_pfn_fwd(const papc_pfn_desc *desc, const papc_pfn_io *io, papc_stream_t st)
{
    PAPC_REQUIRE(desc && io && io->features && io->num_voxels && io->coors && io->w && io->out && io->saved && io->scratch, PAPC_E_INVALID, "papc_pfn_fwd: null pointer");
    const papc_pfn_desc &d = *desc;
    float *cst, *part, *sums; int32_t *argmax; double *gram, *gpart;
    pfn_layout(d, io->saved, io->scratch, cst, argmax, gram, gpart, part, sums);
    const int C = d.C;
    if (d.training) {
        // batch statistics from the inputs' Gram matrix: one float64-MFMA pass over the points instead of a C-channel pass over [P*T, C]
        const int ng = papc_pfn_gram_blocks(d.P);
        unsigned *tk = knob(KNOB_PFN_FUSED_TAILS) ? io->tickets : nullptr;      // (caller-owned ticket words: no process-global state)
        if (tk) {    // the Gram pass's last-arriving workgroup writes the BatchNorm constants (pfn.hip): one launch
            SA_CALL(papc::pfn_gram_stats(io->features, io->num_voxels, io->coors, d.P, d.T, d.vx, d.vy, d.x_offset, d.y_offset, d.zero_padded, gpart, io->w, C, io->gamma, io->beta,
                                         d.eps, d.momentum, cst, cst + C, cst + 2 * C, cst + 3 * C, io->running_mean, io->running_var, gram, tk, as_stream(st)));
        } else {
            SA_CALL(papc::pfn_gram_impl(io->features, io->num_voxels, io->coors, d.P, d.T, d.vx, d.vy, d.x_offset, d.y_offset, gpart, d.zero_padded, st));
            SA_CALL(papc_pfn_gram_finalize_f32(gpart, ng, (int64_t)d.P * d.T, io->w, C, io->gamma, io->beta, d.eps, d.momentum, cst, cst + C, cst + 2 * C, cst + 3 * C,
                                               io->running_mean, io->running_var, gram, st));
        }
    } else {
        PAPC_REQUIRE(io->running_mean && io->running_var, PAPC_E_INVALID, "papc_pfn_fwd: eval mode needs the running statistics");
        SA_CALL(papc_bn_eval_consts_f32(io->running_mean, io->running_var, io->gamma, io->beta, d.eps, C, cst, cst + C, cst + 2 * C, cst + 3 * C, st));
    }
    return papc::pfn_apply_impl(io->features, io->num_voxels, io->coors, d.P, d.T, d.vx, d.vy, d.x_offset, d.y_offset, io->w, C, cst + 2 * C, cst + 3 * C, io->out, argmax, d.zero_padded, st);
}

int papc_pfn_bwd(const papc_pfn_desc *desc, const papc_pfn_io *io, const float *gout, float *dw, float *dgamma, float *dbeta, int accumulate, papc_stream_t st)
{
    PAPC_REQUIRE(desc && io && io->features && io->num_voxels && io->coors && io->w && io->saved && io->scratch && gout && dw && dgamma && dbeta, PAPC_E_INVALID,
                 "papc_pfn_bwd: null pointer");
    const papc_pfn_desc &d = *desc;
    float *cst, *part, *sums; int32_t *argmax; double *gram, *gpart;
    pfn_layout(d, io->saved, io->scratch, cst, argmax, gram, gpart, part, sums);
    const int C = d.C, nb = papc_pfn_num_blocks(d.P);
    if (!d.training) SA_CALL(papc_fill_f32(reinterpret_cast<float *>(gram), 512, 0.f, st));     // eval-mode BN: the Gram terms carry zero weight
    // sparse pass (one argmax row per (pillar, channel)): sum p, sum p*xhat, sum p*x_k; the dense part of dW comes from the Gram matrix
    SA_CALL(papc::pfn_bwd_sparse_impl(io->features, io->num_voxels, io->coors, d.P, d.T, d.vx, d.vy, d.x_offset, d.y_offset, io->w, C, gout, argmax, cst, cst + C,
                                      cst + 2 * C, cst + 3 * C, part, d.zero_padded, st));
    unsigned *tk = knob(KNOB_PFN_FUSED_TAILS) ? io->tickets : nullptr;
    if (tk)      // the fold of the partial rows and the finalize in one launch (last-arriving workgroup, pfn.hip)
        return papc::pfn_bwd_fold_finalize(part, nb, sums, (int64_t)d.P * d.T, io->w, C, gram, cst, cst + C, cst + 2 * C, dgamma, dbeta, dw,
                                           (d.training ? 0 : 1) | (accumulate ? 2 : 0), tk + 1, as_stream(st));
    SA_CALL(papc_reduce_partials_f32(part, nb, 11 * (int64_t)C, sums, 0, st));
    return papc_pfn_bwd_finalize_f32(sums, (int64_t)d.P * d.T, io->w, C, gram, cst, cst + C, cst + 2 * C, dgamma, dbeta, dw, (d.training ? 0 : 1) | (accumulate ? 2 : 0), st);
}

}  // extern "C"
