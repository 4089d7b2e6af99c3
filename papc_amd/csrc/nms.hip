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

__global__ __launch_bounds__(256) void nms_key_kernel(const float *__restrict__ dets, int N, uint64_t *__restrict__ keys)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < N) keys[i] = ((uint64_t)ordered_f32(dets[(int64_t)i * 5 + 4]) << 32) | (uint32_t)i;
}

__global__ __launch_bounds__(256) void nms_gather_kernel(const float *__restrict__ dets, const uint64_t *__restrict__ keys, int N,
                                                         float *__restrict__ sorted, int32_t *__restrict__ order)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const int src = (int)(uint32_t)keys[i];
    order[i] = src;
#pragma unroll
    for (int c = 0; c < 5; ++c) sorted[(int64_t)i * 5 + c] = dets[(int64_t)src * 5 + c];
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
    return up((size_t)N * 8) * 2 + up((size_t)N * 20) + up((size_t)N * 4) + up((size_t)N * cb * 8) + up(sort_bytes) + a;
}

int papc_nms_f32(const float *dets, int N, float nms_overlap_thresh, int32_t *keep, int32_t *num_out, void *workspace,
                 size_t workspace_bytes, papc_stream_t stream)
{
    PAPC_REQUIRE(dets && keep && num_out && workspace, PAPC_E_INVALID, "papc_nms_f32: null pointer");
    PAPC_REQUIRE(N >= 1 && N <= 65536, PAPC_E_INVALID, "papc_nms_f32: N=%d not in [1, 65536]", N);
    PAPC_REQUIRE(workspace_bytes >= papc_nms_workspace(N), PAPC_E_INVALID, "papc_nms_f32: workspace too small");
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_MISC, st);
    const size_t a = 256;
    auto up = [&](size_t x) { return (x + a - 1) / a * a; };
    const int cb = cdiv(N, 64);
    char *w = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(workspace) + a - 1) / a * a);
    uint64_t *keys = reinterpret_cast<uint64_t *>(w); w += up((size_t)N * 8);
    uint64_t *sorted_keys = reinterpret_cast<uint64_t *>(w); w += up((size_t)N * 8);
    float *boxes = reinterpret_cast<float *>(w); w += up((size_t)N * 20);
    int32_t *order = reinterpret_cast<int32_t *>(w); w += up((size_t)N * 4);
    unsigned long long *mask = reinterpret_cast<unsigned long long *>(w); w += up((size_t)N * cb * 8);
    void *tmp = w;
    size_t sort_bytes = 0;
    (void)rocprim::radix_sort_keys_desc(nullptr, sort_bytes, keys, sorted_keys, (size_t)N, 0, 64, st);
    const unsigned nb = (unsigned)cdiv(N, 256);
    hipLaunchKernelGGL(nms_key_kernel, dim3(nb), dim3(256), 0, st, dets, N, keys);
    // descending (score, index): scores.argsort()[::-1] with a stable sort (:145)
    if (rocprim::radix_sort_keys_desc(tmp, sort_bytes, keys, sorted_keys, (size_t)N, 0, 64, st) != hipSuccess) return check_launch("papc_nms_f32: sort");
    hipLaunchKernelGGL(nms_gather_kernel, dim3(nb), dim3(256), 0, st, dets, sorted_keys, N, boxes, order);
    hipLaunchKernelGGL(nms_mask_kernel, dim3((unsigned)cb, (unsigned)cb), dim3(64), 0, st, boxes, N, nms_overlap_thresh, cb, mask);
    hipLaunchKernelGGL(nms_sweep_kernel, dim3(1), dim3(64), (size_t)cb * 8, st, mask, order, N, cb, keep, num_out);
    return check_launch("papc_nms_f32");
}

}  // extern "C"
