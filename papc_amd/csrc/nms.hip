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
// rotated boxes (x, y, x_d, y_d, angle): IoU by convex clipping, nms_gpu.py:179-414 (numba.cuda in the reference).  Typing
// follows the numba source: corner / intersection arithmetic in fp32, the triangle-fan area and the IoU quotient in fp64
// (``area_val = 0.0`` and ``/ 2.0`` are Python floats there).  -ffp-contract=off: no fused multiply-adds.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void rbbox_to_corners(float *c, const float *b)   // :366-389
{
    const float a_cos = cosf(b[4]), a_sin = sinf(b[4]);
    const float hx = b[2] / 2.f, hy = b[3] / 2.f;
    const float cx[4] = {-hx, -hx, hx, hx}, cy[4] = {-hy, hy, hy, -hy};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        c[2 * i] = a_cos * cx[i] + a_sin * cy[i] + b[0];
        c[2 * i + 1] = -a_sin * cx[i] + a_cos * cy[i] + b[1];
    }
}

__device__ __forceinline__ bool point_in_quad(float px, float py, const float *c)   // :323-339
{
    const float ab0 = c[2] - c[0], ab1 = c[3] - c[1], ad0 = c[6] - c[0], ad1 = c[7] - c[1];
    const float ap0 = px - c[0], ap1 = py - c[1];
    const float abab = ab0 * ab0 + ab1 * ab1, abap = ab0 * ap0 + ab1 * ap1;
    const float adad = ad0 * ad0 + ad1 * ad1, adap = ad0 * ap0 + ad1 * ap1;
    return abab >= abap && abap >= 0.f && adad >= adap && adap >= 0.f;
}

__device__ __forceinline__ bool seg_intersection(const float *p1, const float *p2, int i, int j, float *out)   // :235-278
{
    const float A0 = p1[2 * i], A1 = p1[2 * i + 1], B0 = p1[2 * ((i + 1) & 3)], B1 = p1[2 * ((i + 1) & 3) + 1];
    const float C0 = p2[2 * j], C1 = p2[2 * j + 1], D0 = p2[2 * ((j + 1) & 3)], D1 = p2[2 * ((j + 1) & 3) + 1];
    const float BA0 = B0 - A0, BA1 = B1 - A1, DA0 = D0 - A0, CA0 = C0 - A0, DA1 = D1 - A1, CA1 = C1 - A1;
    const bool acd = DA1 * CA0 > CA1 * DA0;
    const bool bcd = (D1 - B1) * (C0 - B0) > (C1 - B1) * (D0 - B0);
    if (acd != bcd) {
        const bool abc = CA1 * BA0 > BA1 * CA0, abd = DA1 * BA0 > BA1 * DA0;
        if (abc != abd) {
            const float DC0 = D0 - C0, DC1 = D1 - C1;
            const float ABBA = A0 * B1 - B0 * A1, CDDC = C0 * D1 - D0 * C1;
            const float DH = BA1 * DC0 - BA0 * DC1;
            const float Dx = ABBA * DC0 - BA0 * CDDC, Dy = ABBA * DC1 - BA1 * CDDC;
            out[0] = Dx / DH; out[1] = Dy / DH;
            return true;
        }
    }
    return false;
}

__device__ double rotate_inter(const float *b1, const float *b2)   // inter(), :392-406
{
    float c1[8], c2[8], pts[16 * 2];   // (the source sizes int_pts at 16 floats = 8 points, the most a quad-quad clip yields generically;
                                       //  degenerate overlaps can list up to 24 candidates, so the scratch here is larger)
    rbbox_to_corners(c1, b1);
    rbbox_to_corners(c2, b2);
    int n = 0;
    for (int i = 0; i < 4; ++i) {      // quadrilateral_intersection, :342-363
        if (point_in_quad(c1[2 * i], c1[2 * i + 1], c2)) { if (n < 16) { pts[2 * n] = c1[2 * i]; pts[2 * n + 1] = c1[2 * i + 1]; } ++n; }
        if (point_in_quad(c2[2 * i], c2[2 * i + 1], c1)) { if (n < 16) { pts[2 * n] = c2[2 * i]; pts[2 * n + 1] = c2[2 * i + 1]; } ++n; }
    }
    float t[2];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            if (seg_intersection(c1, c2, i, j, t)) { if (n < 16) { pts[2 * n] = t[0]; pts[2 * n + 1] = t[1]; } ++n; }
    if (n > 16) n = 16;
    if (n > 0) {                       // sort_vertex_in_convex_polygon, :195-232
        float ctr0 = 0.f, ctr1 = 0.f;
        for (int i = 0; i < n; ++i) { ctr0 += pts[2 * i]; ctr1 += pts[2 * i + 1]; }
        ctr0 /= (float)n; ctr1 /= (float)n;
        float vs[16];
        for (int i = 0; i < n; ++i) {
            float v0 = pts[2 * i] - ctr0, v1 = pts[2 * i + 1] - ctr1;
            const float d = sqrtf(v0 * v0 + v1 * v1);
            v0 = v0 / d; v1 = v1 / d;
            if (v1 < 0.f) v0 = -2.f - v0;
            vs[i] = v0;
        }
        for (int i = 1; i < n; ++i) {
            if (vs[i - 1] > vs[i]) {
                const float temp = vs[i], tx = pts[2 * i], ty = pts[2 * i + 1];
                int j = i;
                while (j > 0 && vs[j - 1] > temp) {
                    vs[j] = vs[j - 1]; pts[2 * j] = pts[2 * j - 2]; pts[2 * j + 1] = pts[2 * j - 1];
                    --j;
                }
                vs[j] = temp; pts[2 * j] = tx; pts[2 * j + 1] = ty;
            }
        }
    }
    double area = 0.0;                 // area(), :185-192
    for (int i = 0; i < n - 2; ++i) {
        const float a0 = pts[0], a1 = pts[1], b0 = pts[2 * i + 2], b1v = pts[2 * i + 3], c0 = pts[2 * i + 4], c1v = pts[2 * i + 5];
        const float num = (a0 - c0) * (b1v - c1v) - (a1 - c1v) * (b0 - c0);      // trangle_area numerator in fp32 (:179-182)
        area += fabs((double)num / 2.0);
    }
    return area;
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

}  // extern "C"
