/*
 * papc_hip.h -- C ABI of libpapc_hip.so: the MI355X (gfx950) hot path of AgentMaker/PAPC.
 *
 * The reference has no FFI on this path: every entry point below replaces a plain Python function or
 * paddle.nn.Layer method (cited per function, paths relative to /root/reference/).  The reference's one
 * native-plugin precedent (pybind11 .so loaded by load_pb11,
 * PAPC/models/detect/pointpillars/libs/tools/buildtools/pybind11_build.py:76-115; callee convention
 * libs/ops/cc/nms/nms.h:17-30: caller pre-allocates outputs, function returns a status/count) is the
 * model for the conventions here:
 *
 *   - every pointer is DEVICE memory owned by the caller (PyTorch tensors); the library allocates
 *     nothing, keeps no global state besides the optional event profiler, never calls hipSetDevice;
 *   - every call only enqueues work on `stream` (a hipStream_t passed as void*), never synchronises;
 *   - returns PAPC_OK (0) or a negative PAPC_E_* code; papc_last_error_string() (thread-local) says why;
 *   - fp32 everywhere ("f32" suffix), indices int32 unless a function says int64.
 *
 * Row layout used by the MLP entry points: activations are row-major [M, C] with rows ordered
 * (b, s, k) -- i.e. the reference's [B, C, K, S] tensors (pointnet2_basic_layers.py:214) stored
 * point-major / channel-contiguous.
 */
#ifndef PAPC_HIP_H
#define PAPC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PAPC_OK 0
#define PAPC_E_INVALID (-1)     /* bad argument (null pointer, size out of range, misalignment) */
#define PAPC_E_UNSUPPORTED (-2) /* valid request outside what the kernels were built for        */
#define PAPC_E_LAUNCH (-3)      /* hipLaunchKernel / runtime error                               */

typedef void *papc_stream_t; /* hipStream_t */

/* library version (major*10000 + minor*100 + patch) */
int papc_version(void);
/* ABI of the descriptor structs below.  They carry no size field: optional trailing fields have been appended round by round (papc_group_src.wstat,
 * papc_group_max.sign_src, papc_bwd_dy.psel, papc_sa_grads.defer, papc_pfn_io.tickets, papc_pfn_desc.zero_padded, ...), and a kernel dereferences
 * or writes through whatever such a pointer holds.  The contract therefore is:
 *   (1) ZERO-INITIALISE every descriptor struct (memset / = {0}) before setting fields -- NULL / 0 in an optional field always selects the
 *       behaviour the library had before the field existed;
 *   (2) a caller compiled against this header checks papc_abi_version() == PAPC_ABI_VERSION once after loading the library (the number changes
 *       whenever a struct's layout or a function's signature does; papc_amd/_lib.py refuses a mismatching library);
 *   (3) a binding that mirrors the structs by hand (ctypes, cgo, JNA) can compare its sizes with papc_abi_sizeof("papc_sa_io") etc.
 *       (-1 for an unknown name; tests/test_abi.py does this for every ctypes mirror in papc_amd/). */
#define PAPC_ABI_VERSION 8
int papc_abi_version(void);
int64_t papc_abi_sizeof(const char *struct_name);
/* text of the last error raised on this thread ("" if none) */
const char *papc_last_error_string(void);

/* ------------------------------------------------------------------------------------------------
 * Sampling / grouping (PAPC/models/layers/pointnet2_basic_layers.py)
 * ---------------------------------------------------------------------------------------------- */

/* farthest_point_sample(xyz, npoint)  -- pointnet2_basic_layers.py:65-95.
 * xyz is addressed as xyz[b*sb + n*sn + c*sc] (elements): [B,N,3] -> (3N,3,1); planar [B,3,N] -> (3N,1,N).
 * start_idx[b] replaces paddle.randint (:76); init_dist is 1.0f for reference parity (:75).
 * out_idx [B,npoint] int32; out_new_xyz [B,npoint,3] (may be NULL) = index_points(xyz, out_idx) (:144).
 * Bit-exact contract: dist = (dx*dx+dy*dy)+dz*dz, strict-< update, lowest index on ties.
 * N <= 16384 runs the register-resident kernel; up to N = 131072 a slower kernel that re-reads the cloud from L2. */
int papc_fps_f32(const float *xyz, int64_t sb, int64_t sn, int64_t sc, int B, int N, int npoint,
                 const int64_t *start_idx, float init_dist, int32_t *out_idx, float *out_new_xyz,
                 papc_stream_t stream);

/* query_ball_point(radius, nsample, xyz, new_xyz) for n_radii radii in ONE scan
 * -- pointnet2_basic_layers.py:98-126 (MSG loop :260-262).
 * xyz strided as above; new_xyz [B,S,3] contiguous.  thr[r] = (float)((double)radius*radius), computed by
 * the host (python scalar promoted to fp32, :112).  out_idx[r] -> [B,S,nsample[r]], int64 when idx64 != 0
 * (the reference's dtype) else int32.  thr/nsample/out_idx are HOST arrays of length n_radii (<= 4).
 * Semantics: first nsample[r] indices j ascending with !(sqdist > thr[r]), padded with the first hit; N
 * everywhere when there is no hit.  sqdist = ((-2*fma-chain dot) + |q|^2) + |p|^2 exactly as :36-38. */
int papc_ball_query_f32(const float *xyz, int64_t sb, int64_t sn, int64_t sc, const float *new_xyz, int B,
                        int N, int S, int n_radii, const float *thr, const int *nsample,
                        void *const *out_idx, int idx64, papc_stream_t stream);

/* square_distance(src, dst) -- pointnet2_basic_layers.py:26-40. src [B,N,3], dst [B,M,3] -> out [B,N,M]. */
int papc_square_distance_f32(const float *src, const float *dst, int B, int N, int M, float *out,
                             papc_stream_t stream);

/* index_points(points, idx) -- pointnet2_basic_layers.py:43-62.  points [B,N,C], idx [B,S] (int32, or
 * int64 when idx64) -> out [B,S,C].  Indices outside [0,N) write zeros (the reference raises IndexError). */
int papc_index_points_f32(const float *points, const void *idx, int idx64, int B, int N, int C, int S,
                          float *out, papc_stream_t stream);
/* gradient of index_points: grad_points[b, idx[b,s], :] += grad_out[b,s,:]  (grad_points pre-zeroed) */
int papc_index_points_bwd_f32(const float *grad_out, const void *idx, int idx64, int B, int N, int C, int S,
                              float *grad_points, papc_stream_t stream);

/* the gather/centre/concat of sample_and_group -- pointnet2_basic_layers.py:146-153 (and the MSG order
 * :263-269).  xyz strided; new_xyz [B,S,3]; feats [B,N,D] or NULL (D=0); idx [B,S,K] int32.
 * out [B,S,K,3+D]: xyz_first != 0 -> [xyz[idx]-new_xyz, feats[idx]] (SSG :151) else [feats[idx], xyz-..] (:267). */
int papc_group_points_f32(const float *xyz, int64_t sb, int64_t sn, int64_t sc, const float *new_xyz,
                          const float *feats, const int32_t *idx, int B, int N, int S, int K, int D,
                          int xyz_first, float *out, papc_stream_t stream);
/* gradient of papc_group_points_f32 (SURVEY 8b's papc_group_gather_bwd; the reference's index_points is where ITS autograd stops,
 * :57-60 -- this is for hosts that let the gradient through).  grad_out [B,S,K,3+D] in the same column order; each output may be
 * NULL: grad_feats [B,N,D] and grad_xyz [B,N,3] are ACCUMULATED into (float atomics: zero them first),
 * grad_new_xyz [B,S,3] = -sum_k of the coordinate columns is written. */
int papc_group_points_bwd_f32(const float *grad_out, const int32_t *idx, int B, int N, int S, int K, int D, int xyz_first,
                              float *grad_feats, float *grad_xyz, float *grad_new_xyz, papc_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Feature propagation (PointNetFeaturePropagation, pointnet2_basic_layers.py:284-335) -- the interpolation half;
 * its Conv1D/BN/ReLU stack runs on the MLP entry points below.
 * ---------------------------------------------------------------------------------------------- */

/* For every query point of xyz1 [B,N,3] (strided like papc_fps_f32) the three smallest squared distances to the
 * support points xyz2 [B,S,3] (strided), ascending, ties in ascending index order (:315-318 with
 * square_distance's canonical arithmetic); weight3 = (1/(d+1e-8)) / sum (:320-322); idx3 = the TRUE neighbour
 * indices.  (The reference sorts `dists` before calling argsort on it (:316-317), so ITS idx is always 0,1,2 --
 * papc_amd.layers reproduces that by default and offers the true indices as an option.)  S >= 3. */
int papc_three_nn_f32(const float *xyz1, int64_t sb1, int64_t sn1, int64_t sc1, const float *xyz2, int64_t sb2,
                      int64_t sn2, int64_t sc2, int B, int N, int S, float *dist3, int32_t *idx3, float *weight3,
                      papc_stream_t stream);
/* out[b,n,:] = (points2[b,idx3[b,n,0],:]*w0 + points2[b,idx3[b,n,1],:]*w1) + points2[b,idx3[b,n,2],:]*w2   (:323).
 * points2 [B,S,D], idx3/weight3 [B,N,3] -> out [B,N,D]. */
int papc_three_interpolate_f32(const float *points2, const int32_t *idx3, const float *weight3, int B, int N, int S,
                               int D, float *out, papc_stream_t stream);
/* gradient w.r.t. points2: grad_points2[b, idx3[b,n,j], :] += w_j * grad_out[b,n,:]   (grad_points2 pre-zeroed) */
int papc_three_interpolate_bwd_f32(const float *grad_out, const int32_t *idx3, const float *weight3, int B, int N,
                                   int S, int D, float *grad_points2, papc_stream_t stream);
/* The same backward when idx3 is the constant (0, 1, 2) of every query -- the neighbours the reference's sort-then-argsort yields
 * (pointnet2_basic_layers.py:316-317): a per-cloud column reduction, no atomics, deterministic; writes ALL of grad_points2 [B,S,D]
 * (rows 0..2 the sums, zeros elsewhere: no pre-zeroing).  S >= 3. */
int papc_three_interpolate_bwd_first3_f32(const float *grad_out, const float *weight3, int B, int N, int S, int D, float *grad_points2,
                                          papc_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Shared pointwise MLP: relu(bn(conv1x1(x))) stacks + max  (pointnet2_basic_layers.py:215-219, :271-276;
 * PAPC/models/classify/pointnet_base/pointnet_base.py:7-25,44)
 * ---------------------------------------------------------------------------------------------- */

/* A-operand sources of papc_mlp_gemm_f32 */
#define PAPC_A_PLAIN 0   /* x[m,k]                                                       */
#define PAPC_A_BNRELU 1  /* relu(scale[k]*x[m,k] + shift[k])   (previous layer's BN+ReLU folded into the load) */
#define PAPC_A_GROUP 2   /* rows gathered on the fly: [xyz[idx]-new_xyz, feats[idx]] (see papc_group_points_f32) */
#define PAPC_A_XYZ 6     /* the activation of a coordinates-only FIRST layer, recomputed: relu(wf[k][0] x + wf[k][1] y + wf[k][2] z + wf[k][3])
                          * with (x, y, z) = xc[m]; x = xc [M, 4] (ldx = 4, papc_xyz_group_f32), bn_scale = wf [Cin, 4] (papc_xyz_l1_finalize_f32).
                          * Forward (no group max) and dW only, where papc_mlp_xyz_ok(M, Cin, Cout) != 0 */

/* Point lists of a grouping (papc_point_lists_f32): for every source point of every cloud the physical rows that gathered it, ascending -- the
 * inverse of the ball-query lists.  Weight-independent like the lists themselves, so a training loop builds them with the sampling pyramid of the
 * next batch; with them the gather-add first layer's backward sums a point's rows in FIXED order (no float atomics: bit-reproducible gradients, no
 * pre-zeroed G).  prange [B*N][2] = (first entry, one past the last) into prow / pmeta; prow [cap] physical row per entry; pmeta [cap][4] per entry:
 * xyz_j - centre (3 floats) and the entry's weight: a compacted grouping's row weight (papc_compact_plan_f32: wrow); for a padded grouping 1 --
 * except that a group's padding copies of its first neighbour (rows with identical input, hence identical gradients) are ONE entry, the first
 * copy's row with weight = their number: both list backwards multiply the row's contribution by it.  cap = rows of the grouping (G * K, rounded up
 * to 128 for a compacted one).  `compact` says which row numbering the lists index: 1 = the compacted layout of papc_compact_plan_f32, 0 = the padded
 * [B,S,K] lists; a consumer whose stack runs the other layout ignores the lists (and takes the atomic path). */
typedef struct papc_point_lists {
    const int32_t *prange;
    const int32_t *prow;
    const float *pmeta;
    int32_t compact;
    const float *pmom;     /* optional (NULL: not built): [B*N][12] per-point moments of the list's entries (w, d) = pmeta -- w_j = sum w | D_j = sum w d (3) |
                              M2_j = sum w d d^T as (00, 01, 02, 11, 12, 22) | 2 floats of padding.  What papc_lingather_bwd_pp_f32 needs instead of y */
} papc_point_lists;

typedef struct papc_group_src {
    const float *xyz;      /* strided cloud */
    int64_t sb, sn, sc;
    const float *new_xyz;  /* [B,S,3] */
    const float *feats;    /* [B,N,D] or NULL */
    const int32_t *idx;    /* [B,S,K] */
    int N, S, K, D;
    int xyz_first;
    /* compacted grouping (papc_compact_plan_f32; all NULL for the padded [B,S,K] lists): point index of every physical row, group of
     * every 8-row segment, physical row count in device memory.  Read by the gather-add first layer (papc_lingather_*) only. */
    const int32_t *cidx, *seg_grp, *rows_dev;
    /* optional, compacted grouping only: the rows' multiplicity weights (papc_compact_plan_f32: wrow, 1 + the group's padding copies on its
     * first row).  With it papc_lingather_fwd_f32 writes the statistics of the PADDED tensor (sum w y, sum w y^2) itself, without it the
     * unweighted sums of the physical rows (the copies' share then comes from papc_bn_stats_corr_f32). */
    const float *wstat;
    /* optional (papc_lingather_bwd_f32 only): the grouping's point lists -- deterministic segmented sums instead of float atomics */
    const papc_point_lists *plists;
} papc_group_src;

/* One conv1x1 layer on rows with fp32 MFMA: y[M,Cout] = A(x)[M,Cin] . w[Cout,Cin]^T + bias.
 * a_mode selects how A is produced (above); x is [M,ldx] for PLAIN/BNRELU, `grp` for GROUP.
 * bn_scale/bn_shift [Cin] fold the previous layer's train-mode BN (from papc_bn_finalize_f32).
 * stats_partial (may be NULL): [papc_mlp_gemm_parts(M), 2, Cout] per-workgroup column sums and sums of
 * squares of y (deterministic, no atomics) for this layer's batch statistics. */
/* rows of stats_partial written by papc_mlp_gemm_f32 for M rows: min(row tiles, 768).  The kernel launches one residency wave of
 * its own occupancy (<= that many workgroups) and writes the rows no workgroup owns as zeros. */
int papc_mlp_gemm_parts(int64_t M);
/* gmax (optional; last layer of a stack): fuse the neighbourhood max (paddle.max(new_points, 2), :219) into the
 * epilogue.  Per group of K consecutive rows the kernel writes max and min of y and the first row offset attaining
 * each ([M/K, Cout] arrays); papc_bn_select_max_f32 then applies BN+ReLU to the max (scale >= 0) or the min
 * (scale < 0).  Only where papc_mlp_gemm_gmax_ok(M, Cout, K) != 0; otherwise use papc_bn_relu_max_f32. */
typedef struct papc_group_max {
    float *gmax, *gmin;      /* [M/K, Cout] */
    int32_t *amax, *amin;    /* [M/K, Cout] */
    int K;
    /* optional (NULL: not given; zero-initialise the struct): [Cout] values with the SIGN of this layer's BatchNorm scale -- its weight gamma,
     * since scale = gamma * invstd and invstd > 0.  With it a kernel may follow only the extremum the ReLU'd BatchNorm can select (the max of a
     * channel with sign_src[c] >= 0, the min of the others) and then writes that one to BOTH (gmax, amax) and (gmin, amin):
     * papc_bn_select_max_f32 finds the same value whichever it reads.  Half the epilogue's compare / select instructions. */
    const float *sign_src;
} papc_group_max;
int papc_mlp_gemm_gmax_ok(int64_t M, int Cout, int K);
int papc_mlp_gemm_f32(int a_mode, const float *x, int64_t ldx, const papc_group_src *grp,
                      const float *bn_scale, const float *bn_shift, const float *w, const float *bias,
                      int64_t M, int Cin, int Cout, float *y, float *stats_partial, const papc_group_max *gmax,
                      papc_stream_t stream);
/* papc_mlp_gemm_f32 over the first rows_dev[0] rows only (a device-side count, <= M, a multiple of 128: the compacted stack of
 * papc_compact_plan_f32; NULL = all M rows).  PLAIN / BNRELU operands on the row-streaming kernel's shapes (M >= 65 536 rows of capacity,
 * Cin in 32..256, Cout % 64 == 0), no fused group max; PAPC_E_UNSUPPORTED elsewhere.  Rows beyond the count are neither read nor written. */
int papc_mlp_gemm_rows_f32(int a_mode, const float *x, int64_t ldx, const papc_group_src *grp,
                           const float *bn_scale, const float *bn_shift, const float *w, const float *bias,
                           int64_t M, int Cin, int Cout, float *y, float *stats_partial, const papc_group_max *gmax,
                           const int32_t *rows_dev, papc_stream_t stream);
/* ... with the rows' multiplicity weights `wrow` (papc_compact_plan_f32): stats_partial then holds the statistics of the PADDED tensor -- sum
 * w y, sum w y^2 over the physical rows -- so a compacted stack needs no papc_bn_stats_corr_f32 launch per layer.  (The kernel adds the
 * copies' share itself: only a group's first row has w != 1 and it starts an 8-row segment, so each wave re-reads rows 0 / 8 / 16 / 24 of the
 * tiles it has just written.)  wrow == NULL: papc_mlp_gemm_rows_f32. */
int papc_mlp_gemm_rows_w_f32(int a_mode, const float *x, int64_t ldx, const papc_group_src *grp,
                             const float *bn_scale, const float *bn_shift, const float *w, const float *bias,
                             int64_t M, int Cin, int Cout, float *y, float *stats_partial, const papc_group_max *gmax,
                             const int32_t *rows_dev, const float *wrow, papc_stream_t stream);
/* out[g,c] = relu(scale*(scale >= 0 ? gmax : gmin) + shift), argmax[g,c] = the matching row offset.  On return gmax holds the
 * SELECTED raw value (ysel: y at the argmax) -- pass it to papc_bn_bwd_reduce_f32 (MAX) so the backward need not gather y. */
int papc_bn_select_max_f32(float *gmax, const float *gmin, const int32_t *amax, const int32_t *amin,
                           const float *scale, const float *shift, int64_t G, int C, float *out, int32_t *argmax,
                           papc_stream_t stream);

/* Reduce stats_partial [n_tiles,2,C] (n_tiles = papc_mlp_gemm_parts(M)) -> train-mode BatchNorm constants (BatchNorm2D :190; paddle default
 * eps 1e-5, biased variance): mean, invstd, and the folded affine scale = gamma*invstd,
 * shift = beta - mean*scale.  running_mean/var (may be NULL) get paddle's momentum update
 * r = momentum*r + (1-momentum)*batch. */
int papc_bn_finalize_f32(const float *stats_partial, int n_tiles, int64_t M, int C, const float *gamma,
                         const float *beta, float eps, float momentum, float *mean, float *invstd,
                         float *scale, float *shift, float *running_mean, float *running_var,
                         papc_stream_t stream);

/* out[g,c] = max_{k<K} relu(scale[c]*y[g*K+k, c] + shift[c]); argmax[g,c] = first k attaining it
 * (paddle.max(new_points, 2) :219).  y [G*K, C] -> out [G,C], argmax [G,C] int32 (may be NULL). */
int papc_bn_relu_max_f32(const float *y, const float *scale, const float *shift, int64_t G, int K, int C,
                         float *out, int32_t *argmax, papc_stream_t stream);

/* z = relu(scale*y+shift) materialised ([M,C]); used where the reference returns activations (PFN non-last
 * layers, PointNet-Basic intermediate checks). */
int papc_bn_relu_f32(const float *y, const float *scale, const float *shift, int64_t M, int C, float *z,
                     papc_stream_t stream);

/* ---- backward of the stack ------------------------------------------------------------------ */

/* dZ sources for the backward kernels */
#define PAPC_DZ_DENSE 0 /* dz[m,c] given densely                                            */
#define PAPC_DZ_MAX 1   /* dz[m,c] = (m%K == argmax[m/K,c]) ? gout[m/K,c] : 0  (backward of the max over K) */

/* Per-channel reductions of the BN+ReLU backward: with p = dz * (scale*y+shift > 0),
 * partial[t] = (sum p, sum p*xhat) over the t-th of n_parts contiguous row ranges, xhat = (y-mean)*invstd.
 * DENSE: dz [M,C]; MAX: gout [M/K,C] + argmax, and dz = NULL (y is gathered at the argmax) or dz = ysel [M/K,C], the raw y at
 * the argmax as left in gmax by papc_bn_select_max_f32.  red_partial [n_parts,2,C] (caller picks n_parts <= 1024). */
int papc_bn_bwd_reduce_f32(int dz_mode, const float *dz, const float *gout, const int32_t *argmax, int K,
                           const float *y, const float *mean, const float *invstd, const float *scale,
                           const float *shift, int64_t M, int C, int n_parts, float *red_partial,
                           papc_stream_t stream);
/* The MAX-mode reduction from the selected raw values alone (ysel, gout: [M/K, C]; y itself is not needed), optionally also writing
 * psel [M/K, C] = scale * p -- the sparse operand papc_bn_max_prep_f32 would otherwise compute in a pass of its own (pass psel = NULL there). */
int papc_bn_bwd_reduce_max_f32(const float *ysel, const float *gout, int K, const float *mean, const float *invstd, const float *scale,
                               const float *shift, int64_t M, int C, int n_parts, float *red_partial, float *psel, papc_stream_t stream);

/* Reduce red_partial -> dgamma[c] = sum p*xhat, dbeta[c] = sum p, and the two per-channel constants of
 * dy = scale*(p - c1 - xhat*c2): c1 = dbeta/M, c2 = dgamma/M.  bit 0 of accumulate adds into dgamma/dbeta; bit 1: eval-mode BN (below). */
int papc_bn_bwd_finalize_f32(const float *red_partial, int n_tiles, int64_t M, int C, float *dgamma,
                             float *dbeta, float *c1, float *c2, int accumulate, papc_stream_t stream);

/* Eval-mode BatchNorm (the source's registered norms under model.eval(): PointNet-Basic's mlp_1/mlp_2,
 * pointnet_base.py:8-24, PFNLayer.norm, pillars.py:24): mean / invstd / scale / shift from the RUNNING statistics,
 * the four vectors papc_bn_finalize_f32 derives from a batch.  The backward of such a layer passes
 * `accumulate | 2` to papc_bn_bwd_finalize_f32 (c1 = c2 = 0: no batch-mean terms). */
int papc_bn_eval_consts_f32(const float *running_mean, const float *running_var, const float *gamma, const float *beta,
                            float eps, int C, float *mean, float *invstd, float *scale, float *shift,
                            papc_stream_t stream);

/* Backward of a max over groups of K consecutive rows (PFNLayer's paddle.max(x, axis=1), pillars.py:34, when the
 * activations x are themselves an output): dx [G*K,C] = gout [G,C] at row argmax [G,C] of each group, 0 elsewhere. */
int papc_group_max_bwd_f32(const float *gout, const int32_t *argmax, int64_t G, int K, int C, float *dx,
                           papc_stream_t stream);

typedef struct papc_bwd_dy {
    int dz_mode;           /* PAPC_DZ_* */
    const float *dz;       /* [M,C] (DENSE) */
    const float *gout;     /* [M/K,C] (MAX) */
    const int32_t *argmax; /* [M/K,C] (MAX) */
    int K;
    const float *y;        /* [M,C] pre-BN output of this layer (saved by forward) */
    const float *mean, *invstd, *scale, *shift, *c1, *c2; /* [C] */
    /* compacted stack (papc_compact_plan_f32 below; all NULL for a padded one): per-row multiplicity weight of the BatchNorm-backward term,
     * dy = scale p - wrow (A + B (y - mean)); group of every 8-row segment (ragged groups: gout / argmax rows are looked up through it
     * and argmax holds ABSOLUTE rows); physical row count in device memory (M then is the capacity, a multiple of 128). */
    const float *wrow;
    const int32_t *seg_grp;
    const int32_t *rows_dev;
    /* optional (MAX): scale * p per (group, channel), p = gout where relu(bn(y at the argmax)) is alive, 0 elsewhere -- what
     * papc_bn_bwd_reduce_max_f32 writes as `psel`.  The compacted row-streaming dX kernel then takes the value as it is instead of repeating
     * the ReLU test and the scale on every row (same products, bit-identical result; 10.5 -> 8 VALU instructions per MFMA).  NULL: from gout. */
    const float *psel;
} papc_bwd_dy;

/* dX[M,Cin] = dY[M,Cout] . w[Cout,Cin], dY produced on the fly from `dy` (never materialised).
 * wt is w transposed: [Cin,Cout].  If scatter != NULL the result rows are instead accumulated
 * (atomicAdd) into grad_feats[b, idx[m], :] for the feature columns of a GROUP layer (gradient of
 * index_points, autograd-correct -- the reference cuts it, pointnet2_basic_layers.py:57-60); xyz columns
 * carry no gradient.  col0/ncols select the slice of dX columns kept (feature part). */
typedef struct papc_scatter_dst {
    float *grad_feats;     /* [B,N,D], pre-zeroed */
    const int32_t *idx;    /* [B,S,K] or NULL for identity rows (group_all) */
    int N, S, K, D;
    int col0;              /* first dX column that maps to feature 0 */
} papc_scatter_dst;
/* next_red (optional, dense dx only): fuse the previous layer's BN-backward reductions into this kernel's epilogue.
 * dx is that layer's dz; with y/mean/invstd/scale/shift of the PREVIOUS layer (channel count Cin) the kernel also
 * writes red_partial [papc_mlp_gemm_parts(M), 2, Cin] = (sum p, sum p*xhat), exactly what papc_bn_bwd_reduce_f32
 * (DENSE) would produce in a separate pass over dx -- feed it to papc_bn_bwd_finalize_f32 with n_tiles =
 * papc_mlp_gemm_parts(M). */
typedef struct papc_bwd_red {
    const float *y;                                  /* [M,Cin] pre-BN output of the previous layer */
    const float *mean, *invstd, *scale, *shift;      /* [Cin] */
    float *red_partial;                              /* [papc_mlp_gemm_parts(M), 2, Cin] */
    int32_t store_masked;                            /* != 0: dx is stored with the previous layer's ReLU mask already applied (p = dx where scale y + shift > 0, else
                                                        0 -- the value the sums above are formed from).  Every consumer of dx applies that mask itself, so the
                                                        stored tensor is interchangeable; papc_lingather_bwd_pp_f32 relies on it (it never reads y) */
} papc_bwd_red;
int papc_mlp_bwd_dx_f32(const papc_bwd_dy *dy, const float *wt, int64_t M, int Cin, int Cout, float *dx,
                        const papc_scatter_dst *scatter, const papc_bwd_red *next_red, papc_stream_t stream);

/* dW partials: dw_partial[t, Cout, Cin] = sum over the rows of chunk t of dY[m,:]^T A(x)[m,:], and
 * db_partial[t, Cout] = sum dY[m,:].  A(x) as in papc_mlp_gemm_f32 (recomputed, not stored).
 * rows_per_chunk is a multiple of 64; n_chunks = ceil(M/rows_per_chunk).  Chunk rows are addressed with the row
 * stride part_ld (>= Cout*Cin for dw_partial, the same stride for db_partial), so both partials may share one buffer:
 * pass db_partial = dw_partial + Cout*Cin and part_ld = Cout*Cin + Cout, then reduce with papc_reduce_partials2_f32. */
int papc_mlp_bwd_dw_f32(const papc_bwd_dy *dy, int a_mode, const float *x, int64_t ldx,
                        const papc_group_src *grp, const float *bn_scale, const float *bn_shift, int64_t M,
                        int Cin, int Cout, int rows_per_chunk, float *dw_partial, float *db_partial,
                        int64_t part_ld, papc_stream_t stream);

/* rows_per_chunk papc_mlp_bwd_dw_f32 would like for this layer (a multiple of 64), or 0 for "any": the row-streaming flavour it runs on
 * layers with a 64-channel BN+ReLU input wants exactly one residency wave of workgroups.  K: rows per group (PAPC_DZ_MAX), else 0. */
int papc_mlp_bwd_dw_chunk_hint(int64_t M, int Cin, int Cout, int a_mode, int dz_mode, int K);

/* out[i] (+)= sum_t partial[t, i]  (fixed order -> deterministic); n = elements per chunk; accumulate != 0 adds
 * into out (gradient accumulation straight into a parameter's .grad) */
int papc_reduce_partials_f32(const float *partial, int n_chunks, int64_t n, float *out, int accumulate,
                             papc_stream_t stream);
/* two outputs from ONE partial buffer whose chunk rows are [n1 | n2] floats with row stride ld (the dW and db
 * partials of papc_mlp_bwd_dw_f32 laid out back to back): out1[i] (+)= sum_t partial[t*ld + i], out2[j] likewise */
/* count <= 8 reductions of the papc_reduce_partials2_f32 kind in ONE launch (`jobs` is a HOST array read during the call): the dW / db
 * partials of every layer of a stack folded at the end of its backward instead of one launch-latency-sized kernel per layer */
typedef struct papc_reduce_job {
    const float *partial;   /* [n_chunks][ld] */
    int32_t n_chunks;
    int32_t accumulate;     /* != 0: add into out1 / out2 */
    int64_t ld, n1, n2;     /* row stride; elements [0,n1) -> out1, [n1,n1+n2) -> out2 */
    float *out1, *out2;     /* out2 may be NULL when n2 == 0 */
} papc_reduce_job;
int papc_reduce_partials_batch_f32(const papc_reduce_job *jobs, int count, papc_stream_t stream);
int papc_reduce_partials2_f32(const float *partial, int n_chunks, int64_t ld, int64_t n1, float *out1, int64_t n2,
                              float *out2, int accumulate, papc_stream_t stream);

/* Deferred folds: up to PAPC_FOLD_MAX partial reductions of ANY of the kinds above (a stack's dW / db partials, the strided xyz / feature
 * column blocks of the gather-add first layer, the split-K partials of the planes path) in ONE launch:
 *     out[r * out_ld + c] (+)= sum_{t < n_chunks} partial[t * ld + r * cols + c]      r < rows, c < cols      (fixed order)
 * A training step folds the partials of ALL its stacks once, behind the last backward kernel, instead of 5-6 launch-latency-sized
 * kernels spread over the backward: papc_sa_mlp_bwd appends its jobs to papc_sa_grads.defer (a HOST list owned by the caller, who keeps
 * the backward scratch buffers alive until papc_fold_jobs_f32 has been enqueued) instead of launching them.  `jobs` is a HOST array. */
#define PAPC_FOLD_MAX 24
typedef struct papc_fold_job {
    const float *partial;   /* [n_chunks][ld] */
    int32_t n_chunks;
    int32_t accumulate;     /* != 0: add into out */
    int64_t ld;             /* chunk stride (floats) */
    int32_t rows, cols;     /* the folded block: rows x cols contiguous in a chunk */
    float *out; int64_t out_ld;
} papc_fold_job;
typedef struct papc_fold_list { papc_fold_job *jobs; int32_t capacity, count; } papc_fold_list;
int papc_fold_jobs_f32(const papc_fold_job *jobs, int count, papc_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * A whole shared-MLP stack in ONE call per direction (csrc/sa_mlp.hip) -- the boundary SURVEY 8b names `papc_sa_mlp_{fwd,bwd}`.
 * Replaces, after the grouping, the body of PointNetSetAbstraction.forward (PAPC/models/layers/pointnet2_basic_layers.py:214-219:
 * relu(bn(conv(.))) x L, max over nsample), of PointNetSetAbstractionMsg's branches (:271-276), of PointNetFeaturePropagation's
 * Conv1D stack (:330-333, pool = 0) and of PointNet-Basic's (classify/pointnet_base/pointnet_base.py:7-25, :44).  The library picks
 * the kernels per layer (gather-add first layer, coordinates-only first layer through its input moments, row-streaming / tiled
 * GEMMs, fused neighbourhood max, a max layer that never stores its output, the compacted form) and lays out its own scratch; the
 * caller owns three buffers: `saved` (forward -> backward), one scratch per direction.  Train-mode BatchNorm unless eval_bn.
 *
 *   papc_sa_desc   the stack: B clouds x S groups x K rows (M = B*S*K rows), layer widths; input = grouped rows [xyz_j - centre | feats_j]
 *                  (or [feats | xyz] with xyz_first = 0; identity_rows: sample_and_group_all, S = 1, K = N, no idx) or plain rows [M, cin]
 *   papc_sa_io     device pointers: inputs, per-layer parameters (+ running statistics, updated in place), out, the three buffers
 *   papc_sa_plan   filled by papc_sa_mlp_plan: the path per layer and the byte sizes of saved / scratch (host struct; keep it for _bwd)
 *   papc_sa_grads  backward: gout [G, c_L] (pool) or [M, c_L]; per layer dw [c_l, cin_l], db, dgamma, dbeta (db / dgamma / dbeta may be
 *                  NULL = not wanted; acc_* != 0 adds into the buffer: gradient accumulation straight into a parameter's .grad);
 *                  optional W^T operands [cin_l, c_l] made by the caller for this forward pass; grad_feats [B,N,D] / grad_x [M,cin] or NULL.
 * Errors: PAPC_E_* as everywhere; nothing is launched after the first failing call. */
#define PAPC_SA_MAX_LAYERS 8
#define PAPC_SA_IN_GROUP 0
#define PAPC_SA_IN_ROWS 1
#define PAPC_SA_NO_LINGATHER 1u   /* `disable` bits: paths NOT to take (A/B, tests) */
#define PAPC_SA_NO_XYZ1 2u
#define PAPC_SA_NO_NOSTORE 4u
#define PAPC_SA_NO_GMAX 8u
#define PAPC_SA_NO_FUSED_RED 16u
#define PAPC_SA_NO_COMPACT 32u
#define PAPC_SA_NO_PLANES 64u            /* few-row stacks (sample_and_group_all, M <= 16 384) on the row kernels instead of the planes kernels */
#define PAPC_SA_NO_PLANES_POINTWISE 128u /* ... only the un-pooled point-wise stacks */
#define PAPC_SA_NO_XYZ_FUSE 256u         /* the dX above a coordinates-only first layer stored + papc_xyz_l1_bwd_f32 instead of papc_mlp_bwd_dx_xyz_f32 */
#define PAPC_SA_NO_PSEL 1024u           /* compacted max layer: its dX kernel recomputes scale * p per row from gout instead of streaming the reduction's psel */
#define PAPC_SA_NO_GSIGN 2048u          /* fused group max: both extrema per channel instead of the one sign(gamma) selects */
#define PAPC_SA_NO_WSTATS 512u          /* compacted stack: unweighted statistics + one papc_bn_stats_corr_f32 launch per layer instead of weighted ones */
typedef struct papc_sa_desc {
    int32_t B, N, S, K, D;
    int32_t n_layers;
    int32_t cin;                          /* PAPC_SA_IN_ROWS: channels of x_rows (grouped input: D + 3) */
    int32_t cout[PAPC_SA_MAX_LAYERS];
    int32_t input;                        /* PAPC_SA_IN_GROUP / PAPC_SA_IN_ROWS */
    int32_t identity_rows;                /* grouped input without idx: row m of cloud b is point m (sample_and_group_all) */
    int32_t xyz_first, pool, eval_bn, cut_gather_grad;
    float eps, momentum;
    uint32_t disable;
    int32_t inference;                    /* != 0: no backward will follow (the planes path then skips the transposed operands it keeps for dW) */
    int32_t want_input_grad;              /* != 0: papc_sa_mlp_bwd will be asked for grad_feats / grad_x (the planes path prepares W_1^T with the forward) */
} papc_sa_desc;
typedef struct papc_sa_layer {
    const float *w, *b, *gamma, *beta;    /* [cout, cin], [cout] x 3 */
    float *running_mean, *running_var;    /* [cout] or NULL */
} papc_sa_layer;
typedef struct papc_compact_src {        /* the tensors of papc_compact_plan_f32 */
    const int32_t *start, *rows, *cidx, *seg_grp;
    const float *wrow, *coef;
    int32_t G;
} papc_compact_src;
typedef struct papc_sa_io {
    const float *xyz; int64_t sb, sn, sc; /* strided cloud [B, N, 3] */
    const float *new_xyz;                 /* [B, S, 3] */
    const float *feats;                   /* [B, N, D] or NULL */
    const int32_t *idx;                   /* [B, S, K] or NULL (identity_rows) */
    const float *x_rows;                  /* PAPC_SA_IN_ROWS: [M, cin] */
    const float *xc; const double *xc_gram;   /* optional: grouped centred coordinates [M, 4] and their folded moments [16] (papc_xyz_group_f32 +
                                                 papc_xyz_gram_fold_f32), e.g. computed with the sampling pyramid one step ahead */
    const papc_compact_src *compact;      /* optional: compacted grouping of idx */
    const float *consts3; int32_t consts3_ld;   /* three rows of consts3_ld >= max cout floats: ones | zeros | 1e30 (identity BatchNorm constants) */
    papc_sa_layer layer[PAPC_SA_MAX_LAYERS];
    float *out;
    void *saved, *scratch;
    const papc_point_lists *plists;       /* optional: point lists of idx (papc_point_lists_f32), used by the gather-add first layer's backward when they
                                             index the row layout the stack runs (plists->compact == plan.compact) */
    const float *wfeat;                   /* optional: the first layer's feature block W_f [c_1, D], contiguous, prepared by the caller for THIS forward
                                             (papc_transpose_batch_ld_f32 with copy = 1); NULL: a gather-add first layer copies it out itself (one launch) */
} papc_sa_io;
typedef struct papc_sa_plan {
    papc_sa_desc d;
    int32_t cin0;
    int32_t lin0, xyz1, gmax, nostore, compact, sparse_max, planes;
    int64_t saved_bytes, fwd_scratch_bytes, bwd_scratch_bytes;
    /* where forward leaves what a caller may want to look at, as byte offsets into `saved` (-1: not stored on this path):
     * pre-BN outputs y_l [M, c_l], BatchNorm constants [4, c_l] (mean | invstd | scale | shift), argmax [G, c_L] int32 */
    int64_t off_y[PAPC_SA_MAX_LAYERS], off_cst[PAPC_SA_MAX_LAYERS], off_argmax;
} papc_sa_plan;
typedef struct papc_sa_grads {
    const float *gout;
    float *dw[PAPC_SA_MAX_LAYERS], *db[PAPC_SA_MAX_LAYERS], *dgamma[PAPC_SA_MAX_LAYERS], *dbeta[PAPC_SA_MAX_LAYERS];
    int32_t acc_w[PAPC_SA_MAX_LAYERS], acc_gb[PAPC_SA_MAX_LAYERS];
    const float *wt[PAPC_SA_MAX_LAYERS];
    float *grad_feats, *grad_x;
    struct papc_fold_list *defer;         /* optional: the partial folds of this backward are appended here instead of launched (papc_fold_jobs_f32);
                                             the parameter gradients are complete only after the caller has run the list, and `scratch` must
                                             stay alive until then */
} papc_sa_grads;
int papc_sa_mlp_plan(const papc_sa_desc *desc, const papc_sa_io *io, papc_sa_plan *plan);
int papc_sa_mlp_fwd(const papc_sa_plan *plan, const papc_sa_io *io, papc_stream_t stream);
int papc_sa_mlp_bwd(const papc_sa_plan *plan, const papc_sa_io *io, const papc_sa_grads *grads, papc_stream_t stream);

/* PillarFeatureNet.forward with its single last PFNLayer (PAPC/models/detect/pointpillars/models/bones/pillars.py:79-108 over :29-37; the
 * shipped configuration num_filters: [64], with_distance = false) in ONE call per direction: decorate + mask + Linear(9 -> C, no bias) +
 * BatchNorm1D(eps, paddle momentum) + ReLU + max over the T points.  features [P,T,4] f32, num_voxels [P] i32, coors [P,4] i32
 * (batch, z, y, x), w [C,9], gamma / beta [C]; out [P,C].  `saved` / `scratch`: papc_pfn_workspace bytes (saved carries forward ->
 * backward).  papc_pfn_bwd: gout [P,C] -> dw [C,9], dgamma, dbeta (accumulate != 0 adds in place).  training = 0: running statistics. */
typedef struct papc_pfn_desc {
    int32_t P, T, C;
    float vx, vy, x_offset, y_offset;     /* pillars.py:74-77 */
    float eps, momentum;
    int32_t training;
    int32_t zero_padded;                  /* != 0: the caller states that rows t >= num_voxels[p] of `features` are zero -- what the reference's
                                             voxeliser hands over (its buffers are zero-initialised, libs/ops/point_cloud/point_cloud_ops.py:148).
                                             The passes then load only the real rows (a KITTI pillar holds ~12 of its 100 slots); with non-zero
                                             padding the result would differ from pillars.py:82 (the cluster mean sums ALL T rows), hence opt-in */
} papc_pfn_desc;
typedef struct papc_pfn_io {
    const float *features; const int32_t *num_voxels, *coors;
    const float *w, *gamma, *beta;
    float *running_mean, *running_var;    /* [C] or NULL (training) */
    float *out;
    void *saved, *scratch;
    uint32_t *tickets;                    /* optional: 2 words of device memory owned by the caller, ZERO at first use; the library leaves them zero
                                             behind every launch.  With them the statistics ride as the last-arriving workgroup's tail of the Gram
                                             pass and the dW / dgamma / dbeta finalize as the tail of the backward fold (5 launches per frame
                                             instead of 8).  One pair per stream on which PFN calls may be in flight at the same time; NULL = the
                                             separate finalize launches */
} papc_pfn_io;
int papc_pfn_workspace(const papc_pfn_desc *desc, int64_t *saved_bytes, int64_t *scratch_bytes);
int papc_pfn_fwd(const papc_pfn_desc *desc, const papc_pfn_io *io, papc_stream_t stream);
int papc_pfn_bwd(const papc_pfn_desc *desc, const papc_pfn_io *io, const float *gout, float *dw, float *dgamma, float *dbeta, int accumulate,
                 papc_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * PointPillars PillarFeatureNet (PAPC/models/detect/pointpillars/models/bones/pillars.py)
 * ---------------------------------------------------------------------------------------------- */

/* PillarFeatureNet.forward with a single (last) PFNLayer -- pillars.py:79-108 + PFNLayer :29-37 --
 * the reference's shipped configuration (num_filters [64]).  features [P,T,4], num_voxels [P] i32,
 * coors [P,4] i32 (batch,z,y,x).  Decoration :82-95 (9 channels), padding mask :99-102,
 * Linear(9->C, no bias) :30, BatchNorm1D(train, eps) :31, ReLU :32, max over T :34.
 * pass 1 writes stats_partial [n_blocks,2,C]; papc_bn_finalize_f32 turns them into scale/shift;
 * pass 2 recomputes the linear layer and writes out [P,C] (+argmax [P,C] or NULL).
 * w is [C,9] ([out,in]; paddle's Linear stores [in,out]).  C <= 64. */
/* decoration only (pillars.py:82-102): out [P,T,9] = masked [x,y,z,r, xyz-mean, x-cx, y-cy] rows */
int papc_pfn_decorate_f32(const float *features, const int32_t *num_voxels, const int32_t *coors, int P, int T,
                          float vx, float vy, float x_offset, float y_offset, int with_distance, float *out,
                          papc_stream_t stream);
/* the same decoration for points of ANY width F >= 3 ([x, y, z, ...]: pillars.py:79-102 only reads the first three / two columns) and
 * any T: features [P,T,F] -> out [P,T,F+5(+1 with_distance)] = [F raw | xyz - cluster mean | x, y - pillar centre | (norm)], masked.
 * PillarFeatureNet(num_input_features != 4) decorates here and runs its PFNLayers on the shared-MLP entry points. */
int papc_pfn_decorate_nf_f32(const float *features, int F, const int32_t *num_voxels, const int32_t *coors, int P, int T,
                             float vx, float vy, float x_offset, float y_offset, int with_distance, float *out, papc_stream_t stream);
int papc_pfn_stats_f32(const float *features, const int32_t *num_voxels, const int32_t *coors, int P, int T,
                       float vx, float vy, float x_offset, float y_offset, const float *w, int C,
                       float *stats_partial, int *n_blocks_out, papc_stream_t stream);
int papc_pfn_apply_f32(const float *features, const int32_t *num_voxels, const int32_t *coors, int P, int T,
                       float vx, float vy, float x_offset, float y_offset, const float *w, int C,
                       const float *scale, const float *shift, float *out, int32_t *argmax,
                       papc_stream_t stream);
/* number of stats blocks papc_pfn_stats_f32 will write for P pillars */
int papc_pfn_num_blocks(int P);
/* backward: given gout [P,C] and argmax, the BN constants and w: dgamma/dbeta partials, dW [C,9].
 * Two passes like the forward (red pass, then dw pass with c1,c2). */
int papc_pfn_bwd_reduce_f32(const float *features, const int32_t *num_voxels, const int32_t *coors, int P,
                            int T, float vx, float vy, float x_offset, float y_offset, const float *w, int C,
                            const float *gout, const int32_t *argmax, const float *mean, const float *invstd,
                            const float *scale, const float *shift, float *red_partial, papc_stream_t stream);
int papc_pfn_bwd_dw_f32(const float *features, const int32_t *num_voxels, const int32_t *coors, int P, int T,
                        float vx, float vy, float x_offset, float y_offset, const float *w, int C,
                        const float *gout, const int32_t *argmax, const float *mean, const float *invstd,
                        const float *scale, const float *shift, const float *c1, const float *c2,
                        float *dw_partial, papc_stream_t stream);

/* The Gram path of the same layer (what papc_amd.pillars.PillarFeatureNet calls; pillars.py:29-37, :79-108).  With 9 input
 * channels the train-mode BatchNorm statistics and the weight gradient depend on the dense [P*T, C] activations only through the
 * inputs' Gram matrix G = sum_rows [x | 1]^T [x | 1] (x = the decorated, masked row; y = x W^T, so sum y_c = W_c . colsum,
 * sum y_c^2 = W_c G W_c^T, sum y_c x_k = (W G)_ck): no 64-channel pass is needed for either.
 *   papc_pfn_gram_f32            one pass over the input: gram_partial [papc_pfn_gram_blocks(P)][256] float64 (16x16 tiles, row-major;
 *                                rows/cols 0..8 = decorated channels, 9 = distance slot, 10 = the constant 1), accumulated in float64
 *                                on the matrix pipe (the quadratic forms cancel: coordinates are O(70 m), a channel's spread O(1));
 *   papc_pfn_gram_finalize_f32   partials -> gram [256] float64 and, when mean != NULL, the BN constants of the layer
 *                                (as papc_bn_finalize_f32: biased variance, paddle momentum rule; M = P*T);
 *   papc_pfn_bwd_sparse_f32      partial [papc_pfn_num_blocks(P)][11][C]: sum p, sum p*xhat and T[c][k] = sum p * x_k of the argmax
 *                                rows (p = gout where the ReLU is alive) -- one row per (pillar, channel), nothing dense;
 *                                reduce over the blocks with papc_reduce_partials_f32 (n = 11*C);
 *   papc_pfn_bwd_finalize_f32    sums [11][C] + gram -> dgamma, dbeta and
 *                                dW_ck = sc_c (T_ck - c1_c colsum_k - c2_c invstd_c ((W G)_ck - mean_c colsum_k)), c1 = sum p / M,
 *                                c2 = sum p*xhat / M.  flags bit 0: eval-mode BN (running statistics, c1 = c2 = 0); bit 1: ADD into dgamma / dbeta / dw. */
int papc_pfn_gram_blocks(int P);
int papc_pfn_gram_f32(const float *features, const int32_t *num_voxels, const int32_t *coors, int P, int T,
                      float vx, float vy, float x_offset, float y_offset, double *gram_partial, papc_stream_t stream);
int papc_pfn_gram_finalize_f32(const double *gram_partial, int n_blocks, int64_t M, const float *w, int C, const float *gamma,
                               const float *beta, float eps, float momentum, float *mean, float *invstd, float *scale, float *shift,
                               float *running_mean, float *running_var, double *gram, papc_stream_t stream);
int papc_pfn_bwd_sparse_f32(const float *features, const int32_t *num_voxels, const int32_t *coors, int P,
                            int T, float vx, float vy, float x_offset, float y_offset, const float *w, int C,
                            const float *gout, const int32_t *argmax, const float *mean, const float *invstd,
                            const float *scale, const float *shift, float *partial, papc_stream_t stream);
int papc_pfn_bwd_finalize_f32(const float *sums, int64_t M, const float *w, int C, const double *gram, const float *mean,
                              const float *invstd, const float *scale, float *dgamma, float *dbeta, float *dw, int flags,
                              papc_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Harness helpers (the reference's Adam step, PAPC/train.py:62-65,113-116, on one flat buffer)
 * ---------------------------------------------------------------------------------------------- */
/* ------------------------------------------------------------------------------------------------
 * Steps either side of the PillarFeatureNet
 * ---------------------------------------------------------------------------------------------- */

/* points_to_voxel (pointpillars/libs/ops/point_cloud/point_cloud_ops.py:8-53 zyx kernel, :56-103 xyz kernel, wrapper
 * :106-166), exact and deterministic on the device: a cell becomes voxel v when its first point (lowest index) is met,
 * points keep their input order inside a voxel (first max_points), and everything from the first point that would open
 * voxel number max_voxels onwards is dropped (the source's `break`, :44-45).
 *   points [N, ndim] fp32 (xyz first); voxel_size[3], coors_range[6] HOST floats (xyz, xyzxyz min/max)
 *   voxels [max_voxels, max_points, ndim] (zero-filled here), coors [max_voxels, 3] int32 (zyx when reverse_index),
 *   num_points [max_voxels] int32, voxel_num [1] int32 (device) -- rows >= *voxel_num stay zero.
 *   workspace: papc_points_to_voxel_workspace(N) bytes of device memory. */
size_t papc_points_to_voxel_workspace(int N);
int papc_points_to_voxel_f32(const float *points, int N, int ndim, const float *voxel_size, const float *coors_range,
                             int max_points, int max_voxels, int reverse_index, float *voxels, int32_t *coors,
                             int32_t *num_points, int32_t *voxel_num, void *workspace, size_t workspace_bytes,
                             papc_stream_t stream);

/* PointPillarsScatter.forward (pointpillars/models/bones/pillars.py:122-142): canvas [B, C, ny, nx] (zero-filled here),
 * canvas[b, :, y, x] = voxel_features[p, :] for coords[p] = (b, z, y, x); a repeated cell keeps the LAST pillar, like the
 * numpy assignment of select_change (libs/functional.py:35-38).  owner [B, ny, nx] int32 is written here (winning pillar
 * per cell, -1 = empty) and read by the backward. */
int papc_pillar_scatter_f32(const float *voxel_features, const int32_t *coords, int P, int C, int batch_size, int ny,
                            int nx, float *canvas, int32_t *owner, papc_stream_t stream);
/* grad_features[p, :] = grad_canvas[b, :, y, x] if pillar p owns its cell, else 0 */
int papc_pillar_scatter_bwd_f32(const float *grad_canvas, const int32_t *coords, const int32_t *owner, int P, int C,
                                int batch_size, int ny, int nx, float *grad_features, papc_stream_t stream);

/* Classifier head (classify/pointnet2/pointnet2.py:17-23, :37-39 / :51-57, :71-73): one launch per layer each way.
 * papc_head_fc_f32: out = [dropout(relu(bn_train(]x W^T + bias[)))]  for x [B,Cin], W [Cout,Cin] (nn.Linear as [out,in]), B <= 256,
 * Cin % 4 == 0.  has_bn = 2: ReLU + dropout without a norm (the PointNet-Basic head, classify/pointnet_base/pointnet_base.py:26-33;
 * gamma / beta / mean / invstd unused, y optional).  has_bn = 1: train-mode BatchNorm1D over the B rows (biased variance, eps), running_mean/var updated with `momentum`
 * (new = (1-m) old + m batch, BIASED batch variance -- the paddle.nn.BatchNorm1D convention; nullable), num_batches_tracked[0] += 1 (nullable); y [B,Cout] (linear output),
 * mean/invstd [Cout] are saved for the backward.  has_bn = 3: EVAL-mode BatchNorm1D (model.eval(): the head's norms are registered layers): running_mean /
 * running_var (required, read only) normalise, nothing is updated, mean / invstd (nullable) receive running_mean and 1/sqrt(running_var + eps);
 * papc_head_bwd_f32 with has_bn = 3 is the backward through the frozen norm (no batch-mean terms).  Dropout p = drop_p in upscale_in_train mode from a counter-based hash of
 * rng_state = {seed, counter} (device int64[2]; null or drop_p == 0: none); keep [B,Cout] (uint8, nullable) receives the mask;
 * rng_bump (nullable, the same int64[2]): counter += 1 at the end of this launch -- pass it on the head's last layer.
 * papc_head_bwd_f32 for layer l: g = gnext . wnext (gnext [B,Cn] = dY of layer l+1, wnext [Cn,Cout]; wnext null: g = gnext),
 * has_bn: dropout/ReLU/BN backward from the saved forward -> dY_l (written to dy when non-null; dgamma, dbeta), else dY_l = g;
 * x non-null: dw [Cout,Cin] = dY_l^T x, db [Cout] = column sums.  accumulate != 0 adds into dw/db/dgamma/dbeta.
 * papc_softmax_xent_f32: loss[0] = mean cross-entropy of logits [B,C] with int64 labels, dlogits = (softmax - onehot)/B. */
int papc_head_fc_f32(const float *x, const float *w, const float *bias, const float *gamma, const float *beta, int B, int Cin, int Cout,
                     int has_bn, float eps, float momentum, float *running_mean, float *running_var, int64_t *num_batches_tracked,
                     float drop_p, const int64_t *rng_state, int layer_tag, int64_t *rng_bump, float *y, float *mean, float *invstd,
                     uint8_t *keep, float *out, papc_stream_t stream);
int papc_head_bwd_f32(const float *gnext, const float *wnext, int Cn, const float *out, const float *y, const float *mean,
                      const float *invstd, const float *gamma, float drop_p, int has_bn, const float *x, int B, int Cin, int Cout,
                      float *dy, float *dw, float *db, float *dgamma, float *dbeta, int accumulate, papc_stream_t stream);
int papc_softmax_xent_f32(const float *logits, const int64_t *labels, int B, int C, float *loss, float *dlogits, papc_stream_t stream);

/* The same layers as PHASES OF ONE LAUNCH each way (csrc/head.hip, head_chain_*_kernel): inside a replayed graph a dependent launch costs
 * about 5 us before its first instruction, and a head is eight of them around ~3 us of work each.  A grid barrier separates the phases;
 * what one phase hands the next is written and read with agent-scope accesses (no fence).  Results are bit-identical to the per-layer calls.
 *   papc_head_chain_fwd_f32   layers[i] = the arguments of papc_head_fc_f32 for layer i (layers[i].x must be layers[i-1].out); with `labels`
 *                             the mean softmax cross-entropy of the last layer's output and its gradient are computed by the same launch
 *                             (classify/pointnet2/pointnet2.py:37-39 + train.py:106-109 in one launch)
 *   papc_head_chain_bwd_f32   jobs[i] = the arguments of papc_head_bwd_f32; jobs of one `phase` run side by side, a job that reads the dy of
 *                             another one belongs to a later phase (phases non-decreasing over the list)
 *   sync                      2 words of device memory owned by the caller, ZERO at first use; the library leaves them zero behind every
 *                             launch.  One pair per stream on which chain launches may be in flight at the same time.
 * At most 4 layers / jobs, 64 / 128 workgroups per phase (all resident at once: the barrier spins); wider heads take the per-layer calls. */
typedef struct papc_head_fc_layer {
    const float *x, *w, *bias, *gamma, *beta;
    int32_t Cin, Cout, has_bn;
    float eps, momentum;
    float *running_mean, *running_var; int64_t *num_batches_tracked;
    float drop_p; int32_t layer_tag;
    float *y, *mean, *invstd; uint8_t *keep; float *out;
} papc_head_fc_layer;
typedef struct papc_head_bwd_job {
    const float *gnext, *wnext; int32_t Cn;
    const float *out, *y, *mean, *invstd, *gamma; float drop_p; int32_t has_bn;
    const float *x; int32_t Cin, Cout;
    float *dy, *dw, *db, *dgamma, *dbeta; int32_t accumulate;
    int32_t phase;
} papc_head_bwd_job;
int papc_head_chain_fwd_f32(const papc_head_fc_layer *layers, int n_layers, int B, const int64_t *rng_state, int64_t *rng_bump,
                            const int64_t *labels, float *loss, float *dlogits, uint32_t *sync, papc_stream_t stream);
int papc_head_chain_bwd_f32(const papc_head_bwd_job *jobs, int n_jobs, int B, uint32_t *sync, papc_stream_t stream);

/* Axis-aligned bitmask NMS: nms_gpu of pointpillars/libs/ops/non_max_suppression/nms_gpu.py:130-164 (CUDA twin
 * libs/ops/cc/nms/nms_kernel.cu.cc:38-157), all on the device.  dets [N,5] = (x1, y1, x2, y2, score) fp32, N <= 65536.
 * keep [N] int32 receives the ORIGINAL indices of the kept boxes in descending-score order (ties: higher index first, the
 * order of a stable argsort reversed), num_out [1] their count.  workspace: papc_nms_workspace(N) bytes. */
size_t papc_nms_workspace(int N);
int papc_nms_f32(const float *dets, int N, float nms_overlap_thresh, int32_t *keep, int32_t *num_out, void *workspace,
                 size_t workspace_bytes, papc_stream_t stream);
/* Rotated boxes, nms_gpu.py:179-653 (numba.cuda in the reference).  The intersection of two rectangles is computed by clipping one
 * against the other's four half-planes (Sutherland-Hodgman, ordered vertex list in registers, fp64 distances / crossings / shoelace on
 * the source's fp32 corners :366-389) -- not by the source's candidate list + angular sort; results agree with it to rounding wherever
 * the source's strict edge tests are not ties (coincident edges, e.g. identical boxes at a general angle, are rounding noise there and
 * exact here).
 * NOT OFFERED, deliberately: a `reference_quirks` mode for that coincident-edge case.  The source's value for two boxes that share an edge
 * line is whatever its fp32 candidate list / angular sort leaves after ties between strict `>` tests (:235-278, :323-339) -- anything between 0
 * and the true IoU for the SAME pair depending on the angle's last bit (oracle/reference_np.py::quad_inter reproduces it) -- so rotated NMS
 * there cannot suppress an exact duplicate.  Matching it would mean shipping the source's routine instruction for instruction; the value has
 * no geometric meaning to preserve, callers get the geometric one (identical boxes -> 1) and tests/test_gpu_nms.py states the divergence.
 * papc_rotate_nms_f32: rotate_nms_gpu (:453-488), dets [N,6] = (x, y, x_d, y_d, angle, score); outputs and workspace as papc_nms_f32.
 * papc_rotate_iou_f32: rotate_iou_gpu / rotate_iou_gpu_eval (:524-653), boxes [N,5], query_boxes [K,5] = (x, y, x_d, y_d, angle) ->
 * iou [N,K]; criterion -1: intersection over union, 0: over area(query), 1: over area(box), 2: the intersection area. */
int papc_rotate_nms_f32(const float *dets, int N, float nms_overlap_thresh, int32_t *keep, int32_t *num_out, void *workspace,
                        size_t workspace_bytes, papc_stream_t stream);
int papc_rotate_iou_f32(const float *boxes, const float *query_boxes, int N, int K, int criterion, float *iou, papc_stream_t stream);
/* rbbox_iou of pointpillars/libs/ops/cc/box_ops.h:23-80 (boost::geometry on the host there): box_corners [N,4,2], qbox_corners [K,4,2]
 * convex quadrilaterals, standup_iou [N,K] (may be NULL: computed from the corners' bounding boxes, iou_jit with eps = 0) ->
 * overlaps [N,K] = |P n Q| / |P u Q| where standup_iou > standup_thresh, 0 elsewhere.
 * papc_riou_f32: riou_cc of libs/ops/box_np_ops.py:16-27 in one launch -- rbboxes [N,5], qrbboxes [K,5] = (x, y, w, l, angle); corners
 * (center_to_corner_box2d :363-383), standup boxes (:236-241) and their IoU (:654-682) are formed on the device. */
int papc_rbbox_iou_f32(const float *box_corners, const float *qbox_corners, const float *standup_iou, float standup_thresh, int N, int K,
                       float *overlaps, papc_stream_t stream);
int papc_riou_f32(const float *rbboxes, const float *qrbboxes, float standup_thresh, int N, int K, float *overlaps, papc_stream_t stream);

/* First layer of a GROUPED stack with the linear map taken before the gather (pointnet2_basic_layers.py:146-153 + conv1 :215-217):
 * a row is [xyz_j - centre | feats_j], so y[m] = P[j] + W_x (xyz_j - centre) + b with P = feats W_f^T [B*N, C] computed once per
 * SOURCE POINT by the caller (a B*N-row GEMM) instead of once per (group, neighbour) row.
 * papc_lingather_fwd_f32: y [M,C] and the per-workgroup column sums / sums of squares stats_partial [papc_lingather_parts(M), 2, C]
 * (feed papc_bn_finalize_f32 with that row count).  w [C, ldw] is the layer weight, its xyz columns are xcol0..xcol0+2; grp gives
 * xyz / new_xyz / idx / N / S / K (feats, D unused); bias may be NULL.  C % 4 == 0.
 * papc_lingather_bwd_f32: from the layer's dense dY source (papc_bwd_dy, DENSE), G [B*N, C] += sum of dY rows per source point
 * (pre-zeroed by the caller, float atomics) and dwx_partial [papc_lingather_parts(M), C, 3] = partial sums of dY^T (xyz_j - centre);
 * with grp->plists (matching the row layout, 256 % (C / 4) == 0): G [B*N, C] = the same sums in fixed order, every row of G WRITTEN (no pre-zeroing),
 * dwx_partial [papc_lingather_list_parts(B * N), C, 3] -- papc_lingather_bwd_parts(grp, B, C) says how many partial rows the call will write;
 * the caller finishes with grad_feats = G W_f, dW_f = G^T feats, dW_x = sum of the partials. */
int papc_lingather_parts(int64_t M);
/* The COMPACTED form of a grouped stack: distinct neighbours only (csrc/compact.hip).  query_ball_point pads every neighbourhood to
 * nsample slots with copies of its first hit (pointnet2_basic_layers.py:118-124); copies are identical rows through every layer of the
 * stack (:214-217), do not change the max (:219) and enter the train-mode BatchNorm statistics and the backward sums only through their
 * multiplicity -- so the stack is the same function, with the same gradients, on the distinct rows plus one weight per group.
 * papc_compact_plan_f32: idx [G, K] int32 (ball-query lists, K % 8 == 0) -> start [G+1] (first physical row of each group; groups are
 * the distinct neighbours in list order, then copies of the first one up to a multiple of 8 rows), rows [2] = {physical rows rounded up
 * to 128 (the last group takes the tail), their exact count}, cidx [cap] point index per row, seg_grp [cap / 8], wrow [cap] = 1 +
 * coef[g] on a group's first row and 1 elsewhere, coef [G] = nsample - rows of the group; cnt8 [G] scratch; cap = G * K rounded up to 128.  Everything
 * stays on the device: consumers take the row count from rows[0] (papc_mlp_gemm_rows_f32, papc_bwd_dy.rows_dev, papc_group_src.rows_dev).
 * papc_bn_stats_corr_f32: the copies' share of a layer's statistics -- writes papc_compact_corr_parts() extra partial rows [r][2][C] =
 * sum_g coef[g] (y, y^2)[start[g]] behind the kernel-written rows of stats_partial; papc_bn_finalize_f32 then takes n_tiles + that many
 * rows and M = G * K.
 * papc_bn_relu_max_seg_f32: the neighbourhood max over ragged groups, out [G,C] = max relu(bn(y)) over the rows of a group, argmax [G,C] =
 * the first row attaining it (ABSOLUTE row), ysel [G,C] = y there. */
/* 1 when a grouped stack of these widths can run compacted: first layer on the gather-add kernel (couts[0] == 128), then 128 -> 128 dense
 * layers and a 128 -> 256 layer under the max (the flavours built so far: SA2 of the SSG classifier), capacity M = G * K >= 65 536 rows. */
int papc_mlp_compact_ok(int64_t M, int K, int n_layers, const int *couts);
int papc_compact_plan_f32(const int32_t *idx, int G, int K, int32_t *cnt8, int32_t *start, int32_t *rows, int32_t *cidx, int32_t *seg_grp,
                          float *wrow, float *coef, papc_stream_t stream);
int papc_compact_corr_parts(void);
int papc_bn_stats_corr_f32(const float *y, int C, const int32_t *start, const float *coef, int G, float *stats_rows, papc_stream_t stream);
int papc_bn_relu_max_seg_f32(const float *y, int C, const int32_t *start, const float *scale, const float *shift, int G, float *out,
                             int32_t *argmax, float *ysel, papc_stream_t stream);
int papc_lingather_fwd_f32(const float *P, const papc_group_src *grp, int B, const float *w, int ldw, int xcol0, const float *bias, int C,
                           float *y, float *stats_partial, papc_stream_t stream);
int papc_lingather_bwd_f32(const papc_bwd_dy *dy, const papc_group_src *grp, int B, int C, float *G, float *dwx_partial, papc_stream_t stream);
/* The list backward WITHOUT the layer's own output: with dz already masked (papc_bwd_red.store_masked on the dX launch that wrote it) everything the
 * gradient needs from y[m] = P[j] + W_x (xyz_j - centre) + b is linear in per-point sums of the list entries' (weight, xyz_j - centre) -- which pmeta
 * carries -- and in P[j] itself:
 *   G[j]       = scale S1[j] - kB (w_j (P[j] + b - mean) + W_x D_j) - kA w_j,       S1 = sum of the masked dz rows of j, w_j = sum of weights, D_j = sum w d
 *   dW_x[c, t] = scale sum_m p d_t - kB (sum_j (P[j] + b - mean)_c D_j[t] + sum_s W_x[c, s] M2[s, t]) - kA Dtot[t],   M2 = sum w d d^T
 * (kA = scale c1, kB = scale c2 invstd) -- so the kernel gathers ONE [rows, C] stream (dz) instead of two (y and dz), and the moments w_j, D_j, M2
 * come with the lists (papc_point_lists.pmom).  Needs grp->plists with pmom (any layout they match), P [B*N, C] (the table papc_lingather_fwd_f32
 * was given), the layer weight's xyz columns and bias as in the forward; C in {64, 128, 256} (papc_lingather_bwd_pp_ok != 0, else
 * PAPC_E_UNSUPPORTED).  Results equal papc_lingather_bwd_f32's up to fp32 rounding (the forward's own rounding of y is not re-read); same
 * partial-row count. */
int papc_lingather_bwd_pp_ok(const papc_group_src *grp, int B, int C);
int papc_lingather_bwd_pp_f32(const papc_bwd_dy *dy, const papc_group_src *grp, int B, int C, const float *P, const float *w, int ldw, int xcol0,
                              const float *bias, float *G, float *dwx_partial, papc_stream_t stream);
int papc_lingather_list_parts(int64_t BN);
int papc_lingather_bwd_parts(const papc_group_src *grp, int B, int C);
int papc_lingather_bwd_lists_ok(const papc_group_src *grp, int C);     /* 1: papc_lingather_bwd_f32 will take the point-list path (G need not be zeroed) */
/* Build the point lists of a grouping: grp gives xyz / new_xyz / idx / N / S / K and, for a compacted grouping, cidx / seg_grp / rows_dev / wstat
 * (= wrow) plus `start` [G + 1] (papc_compact_plan_f32); start == NULL: the padded lists.  One workgroup per cloud, its waves walking runs of the cloud's groups in order, so every
 * list is ascending in the row index.  Entries whose index is outside [0, N) (the no-hit sentinel) are in no list.  N <= 8192.
 * pmom (may be NULL): [B*N][12], the lists' per-point moments (papc_point_lists.pmom; a second small launch, one lane per point in list order). */
int papc_point_lists_f32(const papc_group_src *grp, int B, const int32_t *start, int32_t *prange, int32_t *prow, float *pmeta, float *pmom,
                         papc_stream_t stream);

/* dX of the layer ABOVE a coordinates-only first layer (papc_mlp_xyz_ok), folded into that first layer's backward: dX [M, Cin] is never
 * stored -- per channel c the four sums of p = dX[m, c] [wf_c . x_m + t_c > 0] against (x, y, z, 1) of the row's centred coordinates are all
 * papc_xyz_l1_bwd_finalize_f32 needs.  xc [M,4], wf [Cin,4] (papc_xyz_l1_finalize_f32), partial [papc_mlp_gemm_parts(M)][Cin][4] = what
 * papc_xyz_l1_bwd_f32 writes (pass papc_mlp_gemm_parts(M) as `parts` to the finalize).  Replaces papc_mlp_bwd_dx_f32 + papc_xyz_l1_bwd_f32
 * (one [M, Cin] store and one read less).  papc_mlp_bwd_dx_xyz_ok: where it is built (else PAPC_E_UNSUPPORTED). */
int papc_mlp_bwd_dx_xyz_ok(int64_t M, int Cin, int Cout);
int papc_mlp_bwd_dx_xyz_f32(const papc_bwd_dy *dy, const float *wt, int64_t M, int Cin, int Cout, const float *xc, const float *wf,
                            float *partial, papc_stream_t stream);

/* Backward of a max-pooled LAST layer without reading its dense output y [M,Cout] (the largest tensor of a stack).  With the BN+ReLU
 * backward expanded, dy = s*p - e*y + f (s = scale, e = s*c2*invstd, f = e*mean - s*c1; c1, c2 from papc_bn_bwd_finalize_f32) and
 * y = A W^T + b, A = relu(bn(y_prev)):      dX = (s*p) W - A (W^T E W) + (f - e*b) W.
 * p is non-zero only at the argmax row of each (group, channel), so the first product is a GEMM over a sparse operand that is built
 * from [M/K, Cout] arrays, and the second one is a GEMM over the layer's INPUT (Cin channels).
 * papc_bn_max_prep_f32: psel [G,Co] = s*[s*ysel+shift > 0]*gout (ysel: papc_bn_select_max_f32; NULL: skipped), wcat [Ci, Co+Ci] = [W^T | -W^T E W],
 * hbias [Ci] = (f - e*b) W, e [Co], q [Co] = f - e*b.   (w [Co,Ci]; bias may be NULL.)
 * papc_mlp_bwd_dx_max_f32: dx [M,Cin] = [P | relu(bn_scale*x + bn_shift)] . wcat^T + hbias, P[m,c] = (m%K == argmax[m/K,c]) ? psel : 0;
 * x [M,ldx] is the previous layer's pre-BN output; next_red as in papc_mlp_bwd_dx_f32.  Needs Cout % 16 == 0, Cin % 4 == 0 and 16-byte
 * aligned operands (PAPC_E_UNSUPPORTED otherwise: use papc_mlp_bwd_dx_f32). */
int papc_bn_max_prep_f32(const float *gout, const float *ysel, const float *scale, const float *shift, const float *mean,
                         const float *invstd, const float *c1, const float *c2, const float *w, const float *bias, int64_t G, int Co,
                         int Ci, float *psel, float *wcat, float *hbias, float *e_out, float *q_out, papc_stream_t stream);
int papc_mlp_bwd_dx_max_f32(const float *psel, const int32_t *argmax, int K, const float *x, int64_t ldx, const float *bn_scale,
                            const float *bn_shift, const float *wcat, const float *hbias, int64_t M, int Cin, int Cout, float *dx,
                            const papc_bwd_red *next_red, papc_stream_t stream);

/* The same layer without EVER storing its output (conv -> BN -> ReLU -> max over the K rows of a group,
 * pointnet2_basic_layers.py:215-219 with the last mlp entry): what the pooled result and its backward need of y [M,Cout] is the
 * per-group extrema (papc_group_max), its BN statistics, and sums over the layer's INPUT rows.
 * papc_mlp_max_nostore_ok: 1 where all three row-streaming flavours exist (64 -> 128 channels, K in {32,64,128}, M % 128 == 0 and
 * large).  There: papc_mlp_gemm_f32 accepts y = NULL with gmax != NULL (PAPC_A_BNRELU); dX is papc_mlp_bwd_dx_max_f32; and
 * papc_mlp_bwd_dw_max_f32 forms dW [Cout,Cin] (accumulate != 0: += into dw) in one pass over x [M,Cin] (previous layer's pre-BN output):
 *   dW = P'^T A - (scale*c1) (x) S - diag(e) W (A^T A - S S^T / M),  A = relu(bn_scale*x + bn_shift), S = column sums of A,
 * psel / e from papc_bn_max_prep_f32, (scale, c1) the layer's BN scale and papc_bn_bwd_finalize_f32's c1.  workspace:
 * papc_mlp_bwd_dw_max_ws_floats(M, Cin, Cout) floats, 16-byte aligned.  The bias gradient of such a layer is exactly 0. */
int papc_mlp_max_nostore_ok(int64_t M, int Cin, int Cout, int K);
int64_t papc_mlp_bwd_dw_max_ws_floats(int64_t M, int Cin, int Cout);
int papc_mlp_bwd_dw_max_f32(const float *psel, const int32_t *argmax, int K, const float *x, const float *bn_scale, const float *bn_shift,
                            const float *w, const float *e, const float *scale, const float *c1, int64_t M, int Cin, int Cout,
                            float *workspace, float *dw, int accumulate, papc_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * First layer of a stack fed by coordinates only (sample_and_group with points = None, pointnet2_basic_layers.py:152-153; SA1 of the
 * classifiers, classify/pointnet2/pointnet2.py:11,33): y[m, c] = W[c] . x[m] + b[c], x[m] = xyz[idx[m]] - new_xyz[m / K], 3 channels in.
 * The [M, C] output is never materialised (csrc/xyz1.hip): train-mode BatchNorm statistics and the whole backward of the layer are
 * closed forms of the inputs' second moments, and the layer folds into the next one's operand (PAPC_A_XYZ).
 *   papc_xyz_group_f32            xc [M, 4] = (x, y, z, 0) rows, M = B*S*K (grp: xyz / new_xyz / idx / N / S / K), and the float64
 *                                 partial moments gram_partial [papc_xyz_parts(M), 16] (sum x, y, z, xx, xy, xz, yy, yz, zz, count)
 *   papc_xyz_l1_finalize_f32      moments -> gram [16]; per channel mean / invstd / scale / shift (+ running statistics, paddle momentum)
 *                                 and the folded layer wf [C, 4]: relu(scale (W x + b) + shift) = relu(wf[c][0..2] . x + wf[c][3]);
 *                                 w [C, ldw] with the coordinate columns at xcol0..xcol0+2
 *   papc_xyz_l1_bwd_f32           one pass over dz [M, C] (the gradient w.r.t. the layer's activation): partial [papc_xyz_bwd_parts(M), C, 4]
 *                                 = sums of p x, p y, p z, p with p = dz [activation > 0]
 *   papc_xyz_l1_bwd_finalize_f32  partials + gram -> dgamma, dbeta, dW (coordinate columns of dw [C, ldw]); accumulate != 0 adds in place
 * ---------------------------------------------------------------------------------------------- */
int papc_mlp_xyz_ok(int64_t M, int C1, int C2);
int papc_xyz_parts(int64_t M);
int papc_xyz_bwd_parts(int64_t M);
int papc_xyz_group_f32(const papc_group_src *grp, int B, float *xc, double *gram_partial, papc_stream_t stream);
/* gram_partial [parts, 16] -> gram [16] in fixed order; papc_xyz_l1_finalize_f32 accepts the result as a one-row partial buffer (parts = 1) */
int papc_xyz_gram_fold_f32(const double *gram_partial, int parts, double *gram, papc_stream_t stream);
int papc_xyz_l1_finalize_f32(const double *gram_partial, int parts, int64_t M, const float *w, int ldw, int xcol0, const float *bias,
                             const float *gamma, const float *beta, float eps, float momentum, int C, float *mean, float *invstd, float *scale,
                             float *shift, float *running_mean, float *running_var, float *wf, double *gram, papc_stream_t stream);
int papc_xyz_l1_bwd_f32(const float *dz, const float *xc, const float *wf, int64_t M, int C, float *partial, papc_stream_t stream);
int papc_xyz_l1_bwd_finalize_f32(const float *partial, int parts, int64_t M, int C, const double *gram, const float *w, int ldw, int xcol0,
                                 const float *bias, const float *mean, const float *invstd, const float *scale, float *dgamma, float *dbeta,
                                 float *dw, int accumulate, papc_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Stacks with FEW rows (the group_all set-abstraction layer: sample_and_group_all + conv/BN/ReLU x L + max,
 * pointnet2_basic_layers.py:160-176, :215-219; PointNet2_SSG_Clas.sa3, classify/pointnet2/pointnet2.py:16 -- M = B*128 rows,
 * up to 1024 channels).  The operand transforms are taken out of the GEMMs (csrc/smallm.hip): prep kernels write every operand
 * once as the three bf16 PLANES of the exact fp32 product (x = p0 + p1 + p2, six MFMA products per fp32 product) in MFMA fragment
 * order, and one matrix kernel multiplies two plane sets:  C[i, j] = sum_k A[i, k] * B[j, k].
 *
 * planes of an [R x K] matrix (K = the contraction): R padded to 128, K to 32; papc_pg_planes_bytes(R, K) bytes, 16-byte aligned.
 *   forward  y[M, Co]   : A = planes of the layer input [M x Ci],  B = planes of W [Co x Ci]
 *   dX       dz[M, Ci]  : A = planes of dY [M x Co],               B = planes of W^T [Ci x Co]
 *   dW       dW[Co, Ci] : A = planes of dY^T [Co x M],             B = planes of input^T [Ci x M]   (split over the M contraction)
 * ---------------------------------------------------------------------------------------------- */
size_t papc_pg_planes_bytes(int64_t R, int64_t K);

/* count <= 8 strided fp32 matrices -> planes in one launch: element (r, k) = src[r*row_stride + k*col_stride], r < R, k < K
 * (W: row_stride = Ci, col_stride = 1; W^T: row_stride = 1, col_stride = Ci).  jobs is a HOST array. */
typedef struct papc_pg_wjob {
    const float *src;
    int64_t row_stride, col_stride;
    int R, K;
    void *planes;
} papc_pg_wjob;
int papc_pg_prep_weights_f32(const papc_pg_wjob *jobs, int count, papc_stream_t stream);

/* Rows [M x C] -> planes (contraction over the C channels) and / or planes_t (the transpose: rows = channels, contraction over M;
 * needs M % 32 == 0), with the elementwise part of the layer applied on the way: */
#define PAPC_PG_PLAIN 0    /* x[m, k]                                   (x [M, ldx])                                          */
#define PAPC_PG_CONCAT 1   /* sample_and_group_all rows [xyz | feats] (xyz_first) or [feats | xyz]: xyz strided [B, N, 3], feats [M, D], C = D + 3 */
#define PAPC_PG_BNRELU 2   /* relu(bn(x)): x = the previous layer's pre-BN output; the train-mode batch statistics are folded here from
                            * stats [parts, 2, C] (papc_pg_gemm_f32 FWD epilogue) -- mean / invstd / scale / shift and the running
                            * statistics are WRITTEN (what papc_bn_finalize_f32 does as a separate launch)                      */
#define PAPC_PG_DY_DENSE 3 /* dY of papc_bwd_dy (DENSE): x = this layer's y, dz [M, C]; c1 / c2 are folded here from red [red_parts, 2, C]
                            * (papc_pg_gemm_f32 RED epilogue) and dgamma / dbeta written or accumulated (papc_bn_bwd_finalize_f32)  */
#define PAPC_PG_DY_MAX 4   /* the same under the max over groups of K rows: gout / argmax [M/K, C]; red may be NULL, then the sums
                            * are taken from gout and ysel [M/K, C] (the raw y at the argmax, papc_pg_final_f32)                 */
typedef struct papc_pg_prep {
    int mode;
    int64_t M; int C;
    const float *x; int64_t ldx;
    const float *xyz; int64_t sb, sn, sc; const float *feats; int N, D, xyz_first;
    const float *dz;
    const float *gout, *ysel; const int32_t *argmax; int K;
    const float *stats; int parts;
    const float *gamma, *beta; float eps, momentum; float *running_mean, *running_var;
    float *mean, *invstd, *scale, *shift;
    const float *red; int red_parts;
    float *dgamma, *dbeta; int accumulate;
    void *planes, *planes_t;
} papc_pg_prep;
int papc_pg_prep_rows_f32(const papc_pg_prep *args, papc_stream_t stream);

/* C[i, j] (+ bias[j]) = sum_k A[i, k] B[j, k], i < R1, j < R2, K = contraction length (both plane sets padded alike).
 * Epilogues: */
#define PAPC_PG_STORE 0     /* c [R1, ldc]; split > 1: split-K partials c + z*split_stride (fold with papc_pg_fold_f32)          */
#define PAPC_PG_FWD 1       /* + bias, + stats [ceil(R1/128), 2, R2]: column sums / sums of squares per 128-row tile             */
#define PAPC_PG_FWD_GMAX 2  /* + per 128-row tile (= one group: nsample = 128) max / min of the column and the first row offset  *
                             * attaining each: gmax / gmin / amax / amin [R1/128, R2] (papc_group_max with K = 128)               */
#define PAPC_PG_RED 3       /* dX: + the BN-backward sums of the layer BELOW over the dz just produced (papc_bwd_red): y_prev    *
                             * [R1, R2] and its constants; stats [ceil(R1/128), 2, R2] = sums of p and p*xhat                     */
typedef struct papc_pg_gemm {
    int epi;
    const void *a, *b;
    int R1, R2, K;
    float *c; int64_t ldc;
    int split; int64_t split_stride;
    const float *bias;
    float *stats;
    float *gmax, *gmin; int32_t *amax, *amin;
    const float *y_prev, *mean, *invstd, *scale, *shift;
    int family;             /* PAPC_K_* family the launch is timed under (event profiler) */
} papc_pg_gemm;
int papc_pg_gemm_f32(const papc_pg_gemm *args, papc_stream_t stream);

/* Last forward layer of such a stack: stats [parts, 2, C] -> mean / invstd / scale / shift (+ running statistics), then
 * out[g, c] = relu(scale*(scale >= 0 ? gmax : gmin) + shift), argmax = the matching row offset, and gmax is left holding the
 * selected raw value ysel (papc_bn_finalize_f32 + papc_bn_select_max_f32 in one launch). */
int papc_pg_final_f32(const float *stats, int parts, int64_t M, int C, const float *gamma, const float *beta, float eps, float momentum,
                      float *mean, float *invstd, float *scale, float *shift, float *running_mean, float *running_var, float *gmax,
                      const float *gmin, const int32_t *amax, const int32_t *amin, int64_t G, float *out, int32_t *argmax, papc_stream_t stream);
/* The same when a group spans tiles_per_group >= 2 consecutive 128-row tiles (nsample = 128 tiles_per_group: the max over the N = 1024 points
 * of a cloud in PointNet-Basic, classify/pointnet_base/pointnet_base.py:44): gmax / gmin / amax / amin are per TILE [M/128, C] (the
 * PAPC_PG_FWD_GMAX epilogue), out / argmax / ysel per GROUP [G, C], G = M / (128 tiles_per_group); argmax = row offset inside the group. */
int papc_pg_final_groups_f32(const float *stats, int parts, int64_t M, int C, const float *gamma, const float *beta, float eps, float momentum,
                             float *mean, float *invstd, float *scale, float *shift, float *running_mean, float *running_var, const float *gmax,
                             const float *gmin, const int32_t *amax, const int32_t *amin, int64_t G, int tiles_per_group, float *out, int32_t *argmax,
                             float *ysel, papc_stream_t stream);

/* count <= 8 split-K partial sets folded in one launch, fixed order: out[e] (+)= sum_t partial[t*stride + e], e < n.  HOST array. */
typedef struct papc_pg_fold_job {
    const float *partial; int nsplit; int64_t stride, n;
    float *out; int accumulate;
} papc_pg_fold_job;
int papc_pg_fold_f32(const papc_pg_fold_job *jobs, int count, papc_stream_t stream);

/* count <= 8 row-major fp32 matrices transposed in one launch: dst[i] [cols[i], rows[i]] = src[i] [rows[i], cols[i]]^T.  The four
 * arrays are HOST arrays (read during the call); the matrices are device memory.  Used for the W^T operands of a stack's dX GEMMs. */
int papc_transpose_batch_f32(const float *const *src, float *const *dst, const int *rows, const int *cols, int count, papc_stream_t stream);
/* The same launch with a source row stride per matrix (src_ld, NULL: cols) and, where copy[i] != 0, a plain copy instead of a transpose:
 * dst[i] [rows[i], cols[i]] = the block itself -- a column slice of a wider matrix made contiguous.  A training step uses it to prepare the feature
 * block W_f [Cout, D] of a gather-add first layer (papc_sa_io.wfeat) in the launch that transposes the step's weights. */
int papc_transpose_batch_ld_f32(const float *const *src, const int *src_ld, float *const *dst, const int *rows, const int *cols, const int *copy, int count,
                                papc_stream_t stream);

/* Small data-movement entry points, so that a training step launches nothing but this library's kernels (the
 * reference's layers interleave paddle.zeros / concat / slicing with the math, pointnet2_basic_layers.py:151,170-173):
 *   papc_fill_f32                    p[0..n) = value
 *   papc_copy2d_f32                  dst[r,c] = src[r,c] (transpose: dst[c,r]) for a [rows,cols] block, arbitrary row strides
 *                                    (the feature / coordinate column blocks of a grouped layer's weight, and their transposes)
 *   papc_reduce_partials_strided_f32 out[r*out_ld + c] (+)= sum_t partial[t*ld + r*cols + c]: partial weight gradients summed
 *                                    straight into a column block of the [Cout, Cin] gradient (no concat, no add)
 *   papc_scale_by_f32                out[i] = x[i] * scalar[0], scalar on the device (the upstream gradient of a scalar loss) */
int papc_fill_f32(float *p, int64_t n, float value, papc_stream_t stream);
int papc_copy2d_f32(const float *src, int64_t src_ld, float *dst, int64_t dst_ld, int rows, int cols, int transpose,
                    papc_stream_t stream);
int papc_reduce_partials_strided_f32(const float *partial, int n_chunks, int64_t ld, int rows, int cols, float *out,
                                     int64_t out_ld, int accumulate, papc_stream_t stream);
/* count <= 8 strided 3-D copies in ONE launch (`jobs` is a HOST array read during the call):
 *   dst[b*db + r*dr + c*dc] = src[b*sb + r*sr + c*sc]   for b < B, r < R, c < C   (strides in elements; a stride of 0 broadcasts)
 * -- the concatenations / transposed copies around the layers (pointnet2_basic_layers.py:205, :280, :326-327;
 * segment/pointnet2/pointnet2.py:45) and their backward splits, as one kernel instead of one library copy per piece. */
typedef struct papc_copy_job {
    const float *src;
    float *dst;
    int32_t B, R, C;
    int64_t sb, sr, sc, db, dr, dc;
} papc_copy_job;
int papc_copy_strided_batch_f32(const papc_copy_job *jobs, int count, papc_stream_t stream);
int papc_scale_by_f32(const float *x, const float *scalar, int64_t n, float *out, papc_stream_t stream);

/* Adam with paddle semantics (L2 `weight_decay` added to the gradient): n contiguous params.  The betas are doubles:
 * 1 - beta and the bias corrections are formed in double on the host (as float, 1 - 0.999f is already 1.3e-5 off). */
int papc_adam_step_f32(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n,
                       float lr, double beta1, double beta2, float eps, float weight_decay, int step,
                       float grad_scale, papc_stream_t stream);
/* the same step, and grad[i] = 0 behind it: the next step's in-place gradient accumulation starts from zeros without an
 * optimizer.clear_grad() launch of its own (PAPC/train.py:87) */
int papc_adam_step_zero_f32(float *param, float *grad, float *exp_avg, float *exp_avg_sq, int64_t n,
                            float lr, double beta1, double beta2, float eps, float weight_decay, int step,
                            float grad_scale, papc_stream_t stream);
/* The same update with the step count in DEVICE memory, so that the launch can be part of a captured hipGraph (no host scalar changes from
 * step to step; an eager launch behind a graph replay starts 8-20 us after the graph's last kernel).  step_dev: one int64, the number of the
 * step being applied (1 for the first); papc_adam_tick adds 1 to it -- enqueue it anywhere EARLIER in the step (e.g. on the sampling branch),
 * never concurrently with the update.  zero_grad: bit 0 also clears the gradient bucket (papc_adam_step_zero_f32); bit 1 = the launch ticks
 * for itself: step_dev is then TWO int64 ([0] the number of the last step applied, [1] zero), the kernel applies step [0] + 1 and its
 * last-finishing block stores that number (no papc_adam_tick launch: for a step that has no side branch to hide one on). */
int papc_adam_tick(int64_t *step_dev, papc_stream_t stream);
/* A device-side gate between two streams (bench.py --side-graph).  `flag` = FOUR zero-initialised uint32 words owned by the gate: [0] openings so far,
 * [1] openings waited for so far, [2] sticky count of waits that gave up, [3] reserved.  papc_flag_set adds `value` (1) to [0] (release, agent
 * scope); papc_flag_wait (one lane) spins, bounded by max_spins sleeps of ~1 us, until opening number [1] + 1 has happened, then advances [1].  A
 * wait that gives up increments [2] and still advances [1], so the late opening behind it does not let the NEXT wait through early; the host
 * reads [2] (a plain copy of the word) and must treat a non-zero value as an error.  For two hipGraphs on two streams that must not be joined
 * by a graph edge (a forked branch costs the main chain ~60 us per replay on MI355X) but where the second has to start behind a point of the first.
 * The gate only PLACES work; whatever buffers the two streams share must also be ordered by stream events (bench.py does: the gated graph waits
 * for the end-of-step event of the previous step before it is replayed).  The launch that opens must be enqueued BEFORE the one that waits when
 * both streams may share a hardware queue.  papc_flag_wait_slot: a SECOND waiter on the same gate (slot 1 keeps its count of openings waited for
 * in word [3]; slot 0 is papc_flag_wait) -- two streams released by the same opening. */
int papc_flag_set(uint32_t *flag, uint32_t value, int64_t *counter, papc_stream_t stream);
int papc_flag_wait(uint32_t *flag, int64_t max_spins, papc_stream_t stream);
int papc_flag_wait_slot(uint32_t *flag, int slot, int64_t max_spins, papc_stream_t stream);
int papc_adam_step_dev_f32(float *param, float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, float lr, double beta1, double beta2,
                           float eps, float weight_decay, const int64_t *step_dev, float grad_scale, int zero_grad, papc_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Event profiler (bench.py's live per-kernel durations).  Off by default.
 * papc_prof_enable(mask): bit i enables event pairs around launches of kernel family i
 * (PAPC_K_* below).  papc_prof_read: total ms and launch count since the last reset (synchronises the
 * recorded events).  Uses hipEventRecord on the caller's stream.
 * ---------------------------------------------------------------------------------------------- */
#define PAPC_K_FPS 0
#define PAPC_K_BALL_QUERY 1
#define PAPC_K_GROUP 2
#define PAPC_K_MLP_GEMM 3
#define PAPC_K_BN_RELU_MAX 4
#define PAPC_K_BWD_REDUCE 5
#define PAPC_K_BWD_DX 6
#define PAPC_K_BWD_DW 7
#define PAPC_K_PFN 8
#define PAPC_K_MISC 9
#define PAPC_K_COUNT 10
int papc_prof_enable(unsigned mask);
int papc_prof_reset(void);
int papc_prof_read(int kernel, double *total_ms, int64_t *launches);

/* ------------------------------------------------------------------------------------------------
 * Tuning knobs.  Every PAPC_* environment variable (README.md) is read ONCE, when the library is
 * loaded; entry points never call getenv.  papc_knob_set / papc_knob_get let a tuning harness flip
 * one at run time (process-wide, not thread-safe against concurrent launches).  Unknown names and
 * out-of-range values are PAPC_E_INVALID.
 * ---------------------------------------------------------------------------------------------- */
int papc_knob_set(const char *name, int value);
int papc_knob_get(const char *name, int *value);

#ifdef __cplusplus
}
#endif
#endif /* PAPC_HIP_H */
