// nms.hip -- axis-aligned bitmask NMS for gfx950.
//
// Reference: /root/reference/PAPC/models/detect/pointpillars/libs/ops/non_max_suppression/nms_gpu.py:22-34 (iou_device),
// :73-108 (nms_kernel), :111-127 (nms_postprocess), :130-164 (nms_gpu); C++/CUDA twin libs/ops/cc/nms/nms_kernel.cu.cc:38-157.
//
//   order  = argsort(score) descending                       (radix sort of (ordered score, index) keys, rocPRIM)
//   mask   = per (row box i, 64-column block) one 64-bit word: bit j set when IoU(box_i, box_j) > thr and j comes after i
//            -- the reference builds the word with a 64-iteration loop per thread (threadsPerBlock = 64 = one mask word);
//            on a 64-lane wave the word IS one __ballot over the block's columns
//   sweep  = the reference's sequential host loop (keep i unless an earlier kept box removed it), one wave on the device
// IoU in fp32 with the source's "+1" box convention, -ffp-contract=off: the keep decisions are bit-identical to the oracle.
#include "common.h"
#include <rocprim/rocprim.hpp>

namespace papc {

__device__ __forceinline__ uint32_t ordered_f32(float f)   // order-preserving float -> uint32
{
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(256) void nms_key_kernel(const float *__restrict__ dets, int N, int W, uint64_t *__restrict__ keys)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < N) keys[i] = ((uint64_t)ordered_f32(dets[(int64_t)i * W + W - 1]) << 32) | (uint32_t)i;   // score = last column
}

__global__ __launch_bounds__(256) void nms_gather_kernel(const float *__restrict__ dets, const uint64_t *__restrict__ keys, int N, int W,
                                                         float *__restrict__ sorted, int32_t *__restrict__ order)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const int src = (int)(uint32_t)keys[i];
    order[i] = src;
    for (int c = 0; c < W; ++c) sorted[(int64_t)i * W + c] = dets[(int64_t)src * W + c];
}

__device__ __forceinline__ float iou_dev(const float *a, const float *b)   // nms_gpu.py:22-34
{
    const float left = fmaxf(a[0], b[0]), right = fminf(a[2], b[2]);
    const float top = fmaxf(a[1], b[1]), bottom = fminf(a[3], b[3]);
    const float width = fmaxf(right - left + 1.f, 0.f), height = fmaxf(bottom - top + 1.f, 0.f);
    const float interS = width * height;
    const float Sa = (a[2] - a[0] + 1.f) * (a[3] - a[1] + 1.f);
    const float Sb = (b[2] - b[0] + 1.f) * (b[3] - b[1] + 1.f);
    return interS / (Sa + Sb - interS);
}

// one wave per (row block, column block); lanes = columns
__global__ __launch_bounds__(64) void nms_mask_kernel(const float *__restrict__ boxes, int N, float thr, int col_blocks,
                                                      unsigned long long *__restrict__ mask)
{
    const int row_start = blockIdx.y, col_start = blockIdx.x, lane = threadIdx.x;
    __shared__ float rows[64 * 5];
    const int row_size = min(N - row_start * 64, 64), col_size = min(N - col_start * 64, 64);
    if (lane < row_size) {
#pragma unroll
        for (int c = 0; c < 5; ++c) rows[lane * 5 + c] = boxes[((int64_t)row_start * 64 + lane) * 5 + c];
    }
    float cb[4] = {0.f, 0.f, 0.f, 0.f};
    if (lane < col_size) {
#pragma unroll
        for (int c = 0; c < 4; ++c) cb[c] = boxes[((int64_t)col_start * 64 + lane) * 5 + c];
    }
    __syncthreads();
    unsigned long long mine = 0;
    for (int i = 0; i < row_size; ++i) {
        const int start = (row_start == col_start) ? i + 1 : 0;                                   // :96-98
        const bool hit = lane >= start && lane < col_size && iou_dev(&rows[i * 5], cb) > thr;      // :99-102
        const unsigned long long t = __ballot(hit);
        if (lane == i) mine = t;
    }
    if (lane < row_size) mask[((int64_t)row_start * 64 + lane) * col_blocks + col_start] = mine;  // :105
}

// nms_postprocess (:111-127) on one wave: remv words live in LDS, lanes OR a kept row's words in parallel
__global__ __launch_bounds__(64) void nms_sweep_kernel(const unsigned long long *__restrict__ mask, const int32_t *__restrict__ order, int N,
                                                       int col_blocks, int32_t *__restrict__ keep, int32_t *__restrict__ num_out)
{
    extern __shared__ unsigned long long remv[];
    const int lane = threadIdx.x;
    for (int j = lane; j < col_blocks; j += 64) remv[j] = 0ull;
    __syncthreads();
    int n_keep = 0;
    for (int i = 0; i < N; ++i) {
        const int nblock = i >> 6, inblock = i & 63;
        const bool removed = (remv[nblock] >> inblock) & 1ull;      // uniform (same LDS word for every lane)
        if (!removed) {
            if (lane == 0) keep[n_keep] = order[i];                 // list(order[keep])  (:164)
            ++n_keep;
            for (int j = nblock + lane; j < col_blocks; j += 64) remv[j] |= mask[(int64_t)i * col_blocks + j];
            __syncthreads();
        }
    }
    if (lane == 0) num_out[0] = n_keep;
}

// ---------------------------------------------------------------------------------------------------------------------
// rotated boxes (x, y, x_d, y_d, angle): IoU of two rectangles.  What the reference computes (nms_gpu.py:179-414, numba.cuda) is the
// area of a convex quadrilateral intersection; it gets there by collecting corner-in-box hits and edge-edge crossings, sorting them
// by angle around their centroid and summing a triangle fan.  This file does not follow that route.  Here box A is CLIPPED against
// the four half-planes of box B (Sutherland-Hodgman): the polygon stays an ordered vertex list throughout (at most 4 + 4 vertices),
// so there is no candidate list, no angular sort and no scratch array, and touching / coincident edges are ordinary cases of the
// signed-distance test instead of ties between strict comparisons (the source's IoU of two identical boxes at a general angle is
// rounding noise; here it is 1).  Corners are formed in fp32 exactly as the source forms them (:366-389: they define the rectangles);
// signed distances, crossing points and the shoelace sum are fp64, as are the area accumulator and the quotient in the source
// (`area_val = 0.0`, `/ 2.0` are Python floats under numba).  -ffp-contract=off.
// ---------------------------------------------------------------------------------------------------------------------
struct Quad { float x[4], y[4]; };

__device__ __forceinline__ Quad corners_of(const float *b)   // (x, y, x_d, y_d, angle) -> corners, the source's order and arithmetic (:366-389)
{
    const float c = cosf(b[4]), s = sinf(b[4]);
    const float hx = b[2] / 2.f, hy = b[3] / 2.f;
    Quad q;
    q.x[0] = c * -hx + s * -hy + b[0]; q.y[0] = -s * -hx + c * -hy + b[1];
    q.x[1] = c * -hx + s * hy + b[0];  q.y[1] = -s * -hx + c * hy + b[1];
    q.x[2] = c * hx + s * hy + b[0];   q.y[2] = -s * hx + c * hy + b[1];
    q.x[3] = c * hx + s * -hy + b[0];  q.y[3] = -s * hx + c * -hy + b[1];
    return q;
}

// twice the signed area of a quadrilateral (shoelace); its sign is the winding of the corner order
__device__ __forceinline__ double quad_area2(const Quad &q)
{
    double a = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int j = (i + 1) & 3;
        a += (double)q.x[i] * (double)q.y[j] - (double)q.x[j] * (double)q.y[i];
    }
    return a;
}

// Area of (convex quadrilateral A) n (convex quadrilateral B).  The vertex list lives in eight named slots per coordinate; every index
// below is a compile-time constant after unrolling (a slot is picked with selects), so the lists stay in registers.
__device__ double quad_intersection_area(const Quad &A, const Quad &B)
{
    const double wind = quad_area2(B) < 0.0 ? -1.0 : 1.0;      // inside = left of every edge for counter-clockwise B, right for clockwise
    double px[8], py[8];
    int n = 4;
#pragma unroll
    for (int i = 0; i < 8; ++i) { px[i] = i < 4 ? (double)A.x[i] : 0.0; py[i] = i < 4 ? (double)A.y[i] : 0.0; }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const double cx = (double)B.x[e], cy = (double)B.y[e];
        const double ex = (double)B.x[(e + 1) & 3] - cx, ey = (double)B.y[(e + 1) & 3] - cy;
        // signed distance (times |edge|) of every live vertex to the edge's line, positive inside
        double sd[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) sd[i] = wind * (ex * (py[i] - cy) - ey * (px[i] - cx));
        double qx[8], qy[8];
        int m = 0;
        // the vertex before slot 0 is the last live one
        double lx = px[0], ly = py[0], ls = sd[0];
#pragma unroll
        for (int i = 1; i < 8; ++i) if (i == n - 1) { lx = px[i]; ly = py[i]; ls = sd[i]; }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (i < n) {
                const double vx = px[i], vy = py[i], vs = sd[i];
                const bool in_v = vs >= 0.0, in_l = ls >= 0.0;
                if (in_v != in_l) {                            // the boundary is crossed between the previous vertex and this one
                    const double t = ls / (ls - vs);
                    const double ix = lx + t * (vx - lx), iy = ly + t * (vy - ly);
#pragma unroll
                    for (int k = 0; k < 8; ++k) if (k == m) { qx[k] = ix; qy[k] = iy; }
                    ++m;
                }
                if (in_v) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) if (k == m) { qx[k] = vx; qy[k] = vy; }
                    ++m;
                }
                lx = vx; ly = vy; ls = vs;
            }
        }
        n = m < 8 ? m : 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) { px[i] = qx[i]; py[i] = qy[i]; }
        if (n == 0) return 0.0;
    }
    // shoelace over the live vertices, relative to vertex 0 (keeps the products small)
    double a2 = 0.0;
#pragma unroll
    for (int i = 1; i < 7; ++i)
        if (i + 1 < n) a2 += (px[i] - px[0]) * (py[i + 1] - py[0]) - (px[i + 1] - px[0]) * (py[i] - py[0]);
    return fabs(a2) * 0.5;
}

__device__ __forceinline__ double rotate_inter(const float *b1, const float *b2)   // inter(), :392-406
{
    return quad_intersection_area(corners_of(b1), corners_of(b2));
}

__device__ __forceinline__ double rotate_iou_eval(const float *b1, const float *b2, int criterion)   // :409-414, :562-574
{
    const float area1 = b1[2] * b1[3], area2 = b2[2] * b2[3];
    const double ai = rotate_inter(b1, b2);
    if (criterion == -1) return ai / ((double)(area1 + area2) - ai);
    if (criterion == 0) return ai / (double)area1;
    if (criterion == 1) return ai / (double)area2;
    return ai;
}

// rotate_nms_kernel (:417-450) in the ballot form of nms_mask_kernel; boxes [N,6] = (x, y, x_d, y_d, angle, score)
__global__ __launch_bounds__(64) void rotate_nms_mask_kernel(const float *__restrict__ boxes, int N, float thr, int col_blocks,
                                                             unsigned long long *__restrict__ mask)
{
    const int row_start = blockIdx.y, col_start = blockIdx.x, lane = threadIdx.x;
    __shared__ float rows[64 * 5];
    const int row_size = min(N - row_start * 64, 64), col_size = min(N - col_start * 64, 64);
    if (lane < row_size) {
#pragma unroll
        for (int c = 0; c < 5; ++c) rows[lane * 5 + c] = boxes[((int64_t)row_start * 64 + lane) * 6 + c];
    }
    float cb[5] = {0.f, 0.f, 1.f, 1.f, 0.f};
    if (lane < col_size) {
#pragma unroll
        for (int c = 0; c < 5; ++c) cb[c] = boxes[((int64_t)col_start * 64 + lane) * 6 + c];
    }
    __syncthreads();
    unsigned long long mine = 0;
    for (int i = 0; i < row_size; ++i) {
        const int start = (row_start == col_start) ? i + 1 : 0;
        bool hit = false;
        if (lane >= start && lane < col_size) hit = rotate_iou_eval(&rows[i * 5], cb, -1) > (double)thr;   // devRotateIoU(cur, block_box) (:443-446)
        const unsigned long long t = __ballot(hit);
        if (lane == i) mine = t;
    }
    if (lane < row_size) mask[((int64_t)row_start * 64 + lane) * col_blocks + col_start] = mine;
}

// rotate_iou_kernel(_eval) (:491-521, :577-615): iou[n, k] = devRotateIoUEval(query[k], boxes[n], criterion)
__global__ __launch_bounds__(256) void rotate_iou_kernel(const float *__restrict__ boxes, const float *__restrict__ query, int N, int K,
                                                         int criterion, float *__restrict__ iou)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)N * K) return;
    const int n = (int)(e / K), k = (int)(e - (int64_t)n * K);
    float q[5], b[5];
#pragma unroll
    for (int c = 0; c < 5; ++c) { q[c] = query[(int64_t)k * 5 + c]; b[c] = boxes[(int64_t)n * 5 + c]; }
    iou[e] = (float)rotate_iou_eval(q, b, criterion);
}

// rbbox_iou (cc/box_ops.h:23-80, boost::geometry on the host in the reference): overlaps[n, k] = area(P_n n Q_k) / area(P_n u Q_k) for the
// pairs whose axis-aligned "standup" IoU exceeds standup_thresh, 0 elsewhere.  Convex quadrilaterals: the union's area is
// |P| + |Q| - |P n Q|.  CORNERS = false: the boxes arrive as (x, y, w, l, angle) and the corners (center_to_corner_box2d, box_np_ops.py:363-383:
// the same rotation as rbbox_to_corners), the standup boxes (:236-241) and their IoU (iou_jit with eps = 0, :654-682) are formed here --
// riou_cc (:16-27) in one launch.
template <bool CORNERS>
__global__ __launch_bounds__(256) void rbbox_iou_kernel(const float *__restrict__ boxes, const float *__restrict__ qboxes,
                                                        const float *__restrict__ standup_iou, float standup_thresh, int N, int K,
                                                        float *__restrict__ overlaps)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)N * K) return;
    const int n = (int)(e / K), k = (int)(e - (int64_t)n * K);
    Quad P, Q;
    if (CORNERS) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            P.x[i] = boxes[(int64_t)n * 8 + 2 * i]; P.y[i] = boxes[(int64_t)n * 8 + 2 * i + 1];
            Q.x[i] = qboxes[(int64_t)k * 8 + 2 * i]; Q.y[i] = qboxes[(int64_t)k * 8 + 2 * i + 1];
        }
    } else {
        float b[5], q[5];
#pragma unroll
        for (int c = 0; c < 5; ++c) { b[c] = boxes[(int64_t)n * 5 + c]; q[c] = qboxes[(int64_t)k * 5 + c]; }
        P = corners_of(b);
        Q = corners_of(q);
    }
    float su;
    if (CORNERS && standup_iou) su = standup_iou[e];
    else {
        float p0 = P.x[0], p1 = P.y[0], p2 = P.x[0], p3 = P.y[0], q0 = Q.x[0], q1 = Q.y[0], q2 = Q.x[0], q3 = Q.y[0];
#pragma unroll
        for (int i = 1; i < 4; ++i) {
            p0 = fminf(p0, P.x[i]); p1 = fminf(p1, P.y[i]); p2 = fmaxf(p2, P.x[i]); p3 = fmaxf(p3, P.y[i]);
            q0 = fminf(q0, Q.x[i]); q1 = fminf(q1, Q.y[i]); q2 = fmaxf(q2, Q.x[i]); q3 = fmaxf(q3, Q.y[i]);
        }
        su = 0.f;
        const float iw = fminf(p2, q2) - fmaxf(p0, q0);
        if (iw > 0.f) {
            const float ih = fminf(p3, q3) - fmaxf(p1, q1);
            if (ih > 0.f) su = iw * ih / ((p2 - p0) * (p3 - p1) + (q2 - q0) * (q3 - q1) - iw * ih);
        }
    }
    float out = 0.f;
    if (su > standup_thresh) {
        const double ai = quad_intersection_area(P, Q);
        if (ai > 0.0) {
            const double un = fabs(quad_area2(P)) * 0.5 + fabs(quad_area2(Q)) * 0.5 - ai;
            if (un > 0.0) out = (float)(ai / un);
        }
    }
    overlaps[e] = out;
}

}  // namespace papc

using namespace papc;

extern "C" {

size_t papc_nms_workspace(int N)
{
    if (N < 1) return 0;
    size_t sort_bytes = 0;
    (void)rocprim::radix_sort_keys_desc(nullptr, sort_bytes, (uint64_t *)nullptr, (uint64_t *)nullptr, (size_t)N, 0, 64, (hipStream_t)0);
    const size_t a = 256;
    auto up = [&](size_t x) { return (x + a - 1) / a * a; };
    const size_t cb = (size_t)cdiv(N, 64);
    return up((size_t)N * 8) * 2 + up((size_t)N * 24) + up((size_t)N * 4) + up((size_t)N * cb * 8) + up(sort_bytes) + a;
}

static int nms_driver(const float *dets, int N, int W, float thr, int32_t *keep, int32_t *num_out, void *workspace, size_t workspace_bytes,
                      papc_stream_t stream, const char *who)
{
    PAPC_REQUIRE(dets && keep && num_out && workspace, PAPC_E_INVALID, "%s: null pointer", who);
    PAPC_REQUIRE(N >= 1 && N <= 65536, PAPC_E_INVALID, "%s: N=%d not in [1, 65536]", who, N);
    PAPC_REQUIRE(workspace_bytes >= papc_nms_workspace(N), PAPC_E_INVALID, "%s: workspace too small", who);
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_MISC, st);
    const size_t a = 256;
    auto up = [&](size_t x) { return (x + a - 1) / a * a; };
    const int cb = cdiv(N, 64);
    char *w = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(workspace) + a - 1) / a * a);
    uint64_t *keys = reinterpret_cast<uint64_t *>(w); w += up((size_t)N * 8);
    uint64_t *sorted_keys = reinterpret_cast<uint64_t *>(w); w += up((size_t)N * 8);
    float *boxes = reinterpret_cast<float *>(w); w += up((size_t)N * 24);
    int32_t *order = reinterpret_cast<int32_t *>(w); w += up((size_t)N * 4);
    unsigned long long *mask = reinterpret_cast<unsigned long long *>(w); w += up((size_t)N * cb * 8);
    void *tmp = w;
    size_t sort_bytes = 0;
    (void)rocprim::radix_sort_keys_desc(nullptr, sort_bytes, keys, sorted_keys, (size_t)N, 0, 64, st);
    const unsigned nb = (unsigned)cdiv(N, 256);
    hipLaunchKernelGGL(nms_key_kernel, dim3(nb), dim3(256), 0, st, dets, N, W, keys);
    // descending (score, index): scores.argsort()[::-1] with a stable sort (:145, :469)
    if (rocprim::radix_sort_keys_desc(tmp, sort_bytes, keys, sorted_keys, (size_t)N, 0, 64, st) != hipSuccess) return check_launch(who);
    hipLaunchKernelGGL(nms_gather_kernel, dim3(nb), dim3(256), 0, st, dets, sorted_keys, N, W, boxes, order);
    if (W == 5) hipLaunchKernelGGL(nms_mask_kernel, dim3((unsigned)cb, (unsigned)cb), dim3(64), 0, st, boxes, N, thr, cb, mask);
    else hipLaunchKernelGGL(rotate_nms_mask_kernel, dim3((unsigned)cb, (unsigned)cb), dim3(64), 0, st, boxes, N, thr, cb, mask);
    hipLaunchKernelGGL(nms_sweep_kernel, dim3(1), dim3(64), (size_t)cb * 8, st, mask, order, N, cb, keep, num_out);
    return check_launch(who);
}

int papc_nms_f32(const float *dets, int N, float nms_overlap_thresh, int32_t *keep, int32_t *num_out, void *workspace,
                 size_t workspace_bytes, papc_stream_t stream)
{
    return nms_driver(dets, N, 5, nms_overlap_thresh, keep, num_out, workspace, workspace_bytes, stream, "papc_nms_f32");
}

int papc_rotate_nms_f32(const float *dets, int N, float nms_overlap_thresh, int32_t *keep, int32_t *num_out, void *workspace,
                        size_t workspace_bytes, papc_stream_t stream)
{
    return nms_driver(dets, N, 6, nms_overlap_thresh, keep, num_out, workspace, workspace_bytes, stream, "papc_rotate_nms_f32");
}

int papc_rotate_iou_f32(const float *boxes, const float *query_boxes, int N, int K, int criterion, float *iou, papc_stream_t stream)
{
    PAPC_REQUIRE(boxes && query_boxes && iou, PAPC_E_INVALID, "papc_rotate_iou_f32: null pointer");
    PAPC_REQUIRE(N >= 1 && K >= 1 && criterion >= -1 && criterion <= 2, PAPC_E_INVALID, "papc_rotate_iou_f32: N=%d K=%d criterion=%d", N, K, criterion);
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_MISC, st);
    hipLaunchKernelGGL(rotate_iou_kernel, dim3((unsigned)cdiv((int64_t)N * K, 256)), dim3(256), 0, st, boxes, query_boxes, N, K, criterion, iou);
    return check_launch("papc_rotate_iou_f32");
}

int papc_rbbox_iou_f32(const float *box_corners, const float *qbox_corners, const float *standup_iou, float standup_thresh, int N, int K,
                       float *overlaps, papc_stream_t stream)
{
    PAPC_REQUIRE(box_corners && qbox_corners && overlaps, PAPC_E_INVALID, "papc_rbbox_iou_f32: null pointer");
    PAPC_REQUIRE(N >= 1 && K >= 1, PAPC_E_INVALID, "papc_rbbox_iou_f32: N=%d K=%d", N, K);
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_MISC, st);
    hipLaunchKernelGGL(rbbox_iou_kernel<true>, dim3((unsigned)cdiv((int64_t)N * K, 256)), dim3(256), 0, st, box_corners, qbox_corners, standup_iou,
                       standup_thresh, N, K, overlaps);
    return check_launch("papc_rbbox_iou_f32");
}

int papc_riou_f32(const float *rbboxes, const float *qrbboxes, float standup_thresh, int N, int K, float *overlaps, papc_stream_t stream)
{
    PAPC_REQUIRE(rbboxes && qrbboxes && overlaps, PAPC_E_INVALID, "papc_riou_f32: null pointer");
    PAPC_REQUIRE(N >= 1 && K >= 1, PAPC_E_INVALID, "papc_riou_f32: N=%d K=%d", N, K);
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_MISC, st);
    hipLaunchKernelGGL(rbbox_iou_kernel<false>, dim3((unsigned)cdiv((int64_t)N * K, 256)), dim3(256), 0, st, rbboxes, qrbboxes, (const float *)nullptr,
                       standup_thresh, N, K, overlaps);
    return check_launch("papc_riou_f32");
}

}  // extern "C"
