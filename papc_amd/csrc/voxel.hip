// voxel.hip -- the two steps either side of the PillarFeatureNet, for gfx950:
//
//   points_to_voxel      : /root/reference/PAPC/models/detect/pointpillars/libs/ops/point_cloud/point_cloud_ops.py:8-53 (zyx
//                          "reverse" kernel), :56-103 (xyz kernel), wrapper :106-166
//   PointPillarsScatter  : /root/reference/PAPC/models/detect/pointpillars/models/bones/pillars.py:110-142
//
// The reference voxeliser is a sequential first-come loop: a cell becomes voxel number v when its FIRST point is met,
// points are appended in input order up to max_points, and the loop BREAKS at the first point that would open voxel
// number max_voxels (so every later point is dropped, also those of existing voxels).  The device formulation is exact
// and order-independent:
//   1. key_i = (cell_i << 32) | i  (cell = the index into the reference's coor_to_voxelidx map; out-of-range points get
//      the all-ones cell and sort last);
//   2. sort the keys (rocPRIM radix sort, only the occupied bits): points of a cell become one segment, ascending i;
//   3. a cell's first point marks flag[i] = 1; exclusive scan of flag over the POINT index = number of cells opened before
//      point i = the reference's voxel number of the cell; the point that finds max_voxels cells before it is the break;
//   4. one thread per segment walks its points: voxel row t = t-th point of the cell with index below the break.
// Integer / byte work bounded by HBM and the sort; nothing here wants the matrix cores.
#include "common.h"
#include <rocprim/rocprim.hpp>

namespace papc {

struct VoxGrid {
    float lo[3], vs[3];
    int g[3];          // grid size along x, y, z (reference: round((hi - lo) / vs))
    int reverse;       // coordinates stored z,y,x (point_cloud_ops.py:39) instead of x,y,z (:87)
};

__device__ __forceinline__ uint32_t cell_of(const float *pt, const VoxGrid &v, int c[3])
{
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float f = floorf((pt[j] - v.lo[j]) / v.vs[j]);      // :35 / :82, float32 like the jitted source
        if (f < 0.f || f >= (float)v.g[j]) return 0xFFFFFFFFu;    // :36-38
        c[j] = (int)f;
    }
    // index into coor_to_voxelidx: [z][y][x] for the reverse kernel (shape reversed, :134-135), [x][y][z] otherwise
    return v.reverse ? (uint32_t)((c[2] * v.g[1] + c[1]) * v.g[0] + c[0]) : (uint32_t)((c[0] * v.g[1] + c[1]) * v.g[2] + c[2]);
}

__global__ __launch_bounds__(256) void vox_key_kernel(const float *__restrict__ pts, int N, int ndim, VoxGrid v,
                                                      uint64_t *__restrict__ keys)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    int c[3];
    const uint32_t cell = cell_of(pts + (int64_t)i * ndim, v, c);
    keys[i] = ((uint64_t)cell << 32) | (uint32_t)i;
}

__global__ __launch_bounds__(256) void vox_mark_kernel(const uint64_t *__restrict__ keys, int N, int32_t *__restrict__ flag)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= N) return;
    const uint32_t cell = (uint32_t)(keys[k] >> 32);
    if (cell == 0xFFFFFFFFu) return;
    if (k == 0 || (uint32_t)(keys[k - 1] >> 32) != cell) flag[(uint32_t)keys[k]] = 1;   // the cell's first point (lowest index)
}

// the break point (:44-45): the first point that would open voxel number max_voxels; N when there is none
__global__ __launch_bounds__(256) void vox_cut_kernel(const int32_t *__restrict__ flag, const int32_t *__restrict__ before, int N,
                                                      int max_voxels, int32_t *__restrict__ cut_and_num)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    if (flag[i] && before[i] == max_voxels) cut_and_num[0] = i;   // exactly one such point
    if (i == N - 1) { const int total = before[i] + flag[i]; cut_and_num[1] = total < max_voxels ? total : max_voxels; }
}

__global__ __launch_bounds__(256) void vox_emit_kernel(const float *__restrict__ pts, const uint64_t *__restrict__ keys, int N, int ndim,
                                                       VoxGrid v, const int32_t *__restrict__ before,
                                                       const int32_t *__restrict__ cut_and_num, int max_points, int max_voxels,
                                                       float *__restrict__ voxels, int32_t *__restrict__ coors,
                                                       int32_t *__restrict__ num_points)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= N) return;
    const uint32_t cell = (uint32_t)(keys[k] >> 32);
    if (cell == 0xFFFFFFFFu) return;
    if (k > 0 && (uint32_t)(keys[k - 1] >> 32) == cell) return;       // not a segment start
    const int first = (int)(uint32_t)keys[k];
    const int cut = cut_and_num[0];
    const int vid = before[first];
    if (vid >= max_voxels || first >= cut) return;                     // opened at / after the break
    int c[3];
    cell_of(pts + (int64_t)first * ndim, v, c);
    if (v.reverse) { coors[vid * 3 + 0] = c[2]; coors[vid * 3 + 1] = c[1]; coors[vid * 3 + 2] = c[0]; }   // coor[ndim_minus_1 - j] = c  (:39)
    else { coors[vid * 3 + 0] = c[0]; coors[vid * 3 + 1] = c[1]; coors[vid * 3 + 2] = c[2]; }
    int cnt = 0;
    for (int t = k; t < N; ++t) {
        const uint64_t key = keys[t];
        if ((uint32_t)(key >> 32) != cell) break;
        const int i = (int)(uint32_t)key;
        if (i >= cut) break;                                           // ascending inside the segment: all later ones too
        if (cnt < max_points) {                                        // :48-50
            const float *s = pts + (int64_t)i * ndim;
            float *d = voxels + ((int64_t)vid * max_points + cnt) * ndim;
            for (int e = 0; e < ndim; ++e) d[e] = s[e];
            ++cnt;
        } else break;
    }
    num_points[vid] = cnt;
}

// ---- PointPillarsScatter
__global__ __launch_bounds__(256) void scatter_owner_kernel(const int32_t *__restrict__ coords, int P, int B, int ny, int nx,
                                                            int32_t *__restrict__ owner)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const int b = coords[p * 4], y = coords[p * 4 + 2], x = coords[p * 4 + 3];
    if (b < 0 || b >= B || y < 0 || y >= ny || x < 0 || x >= nx) return;
    atomicMax(&owner[((int64_t)b * ny + y) * nx + x], p);   // numpy's x[:, index] = y keeps the LAST duplicate (functional.py:35-38)
}

template <bool BWD>
__global__ __launch_bounds__(256) void scatter_move_kernel(const float *__restrict__ src, const int32_t *__restrict__ coords,
                                                           const int32_t *__restrict__ owner, int P, int C, int B, int ny, int nx,
                                                           float *__restrict__ dst)
{
    // one wave per pillar, lanes over channels
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= P) return;
    const int lane = threadIdx.x & 63;
    const int b = coords[p * 4], y = coords[p * 4 + 2], x = coords[p * 4 + 3];
    const bool in = !(b < 0 || b >= B || y < 0 || y >= ny || x < 0 || x >= nx);
    const int64_t cell = in ? ((int64_t)b * ny + y) * nx + x : 0;
    const bool mine = in && owner[cell] == p;
    const int64_t plane = (int64_t)ny * nx;
    for (int c = lane; c < C; c += 64) {
        const int64_t ci = ((int64_t)(in ? b : 0) * C + c) * plane + (in ? (int64_t)y * nx + x : 0);
        if (BWD) dst[(int64_t)p * C + c] = mine ? src[ci] : 0.f;      // grad_features <- grad_canvas
        else if (mine) dst[ci] = src[(int64_t)p * C + c];             // canvas <- features
    }
}

}  // namespace papc

using namespace papc;

static int fill_grid(VoxGrid &v, const float *voxel_size, const float *coors_range, int reverse_index, int64_t *cells)
{
    for (int j = 0; j < 3; ++j) {
        v.lo[j] = coors_range[j]; v.vs[j] = voxel_size[j];
        const float g = (coors_range[3 + j] - coors_range[j]) / voxel_size[j];   // float32, like the numpy expression (:25)
        v.g[j] = (int)rintf(g);                                                   // np.round: half to even
        if (!(v.g[j] >= 1)) return 0;
    }
    v.reverse = reverse_index ? 1 : 0;
    *cells = (int64_t)v.g[0] * v.g[1] * v.g[2];
    return *cells < 0xFFFFFFFFll;
}

extern "C" {

size_t papc_points_to_voxel_workspace(int N)
{
    if (N < 1) return 0;
    size_t sort_bytes = 0, scan_bytes = 0;
    (void)rocprim::radix_sort_keys(nullptr, sort_bytes, (uint64_t *)nullptr, (uint64_t *)nullptr, (size_t)N, 0, 64, (hipStream_t)0);
    (void)rocprim::exclusive_scan(nullptr, scan_bytes, (int32_t *)nullptr, (int32_t *)nullptr, 0, (size_t)N, rocprim::plus<int32_t>(), (hipStream_t)0);
    const size_t a = 256;
    auto up = [&](size_t x) { return (x + a - 1) / a * a; };
    // keys in, keys out, flag, before, cut+num, rocPRIM temporary storage
    return up((size_t)N * 8) * 2 + up((size_t)N * 4) * 2 + a + up(std::max(sort_bytes, scan_bytes)) + a;
}

int papc_points_to_voxel_f32(const float *points, int N, int ndim, const float *voxel_size, const float *coors_range,
                             int max_points, int max_voxels, int reverse_index, float *voxels, int32_t *coors,
                             int32_t *num_points, int32_t *voxel_num, void *workspace, size_t workspace_bytes,
                             papc_stream_t stream)
{
    PAPC_REQUIRE(points && voxel_size && coors_range && voxels && coors && num_points && voxel_num && workspace, PAPC_E_INVALID,
                 "papc_points_to_voxel_f32: null pointer");
    PAPC_REQUIRE(N >= 1 && ndim >= 3 && max_points >= 1 && max_voxels >= 1, PAPC_E_INVALID, "papc_points_to_voxel_f32: bad sizes");
    PAPC_REQUIRE(workspace_bytes >= papc_points_to_voxel_workspace(N), PAPC_E_INVALID, "papc_points_to_voxel_f32: workspace too small");
    VoxGrid v;
    int64_t cells = 0;
    PAPC_REQUIRE(fill_grid(v, voxel_size, coors_range, reverse_index, &cells), PAPC_E_INVALID, "papc_points_to_voxel_f32: bad voxel grid");
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_PFN, st);
    const size_t a = 256;
    auto up = [&](size_t x) { return (x + a - 1) / a * a; };
    char *w = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(workspace) + a - 1) / a * a);
    uint64_t *keys = reinterpret_cast<uint64_t *>(w); w += up((size_t)N * 8);
    uint64_t *sorted = reinterpret_cast<uint64_t *>(w); w += up((size_t)N * 8);
    int32_t *flag = reinterpret_cast<int32_t *>(w); w += up((size_t)N * 4);
    int32_t *before = reinterpret_cast<int32_t *>(w); w += up((size_t)N * 4);
    int32_t *cut = reinterpret_cast<int32_t *>(w); w += a;
    void *tmp = w;
    size_t sort_bytes = 0, scan_bytes = 0;
    (void)rocprim::radix_sort_keys(nullptr, sort_bytes, keys, sorted, (size_t)N, 0, 64, st);
    (void)rocprim::exclusive_scan(nullptr, scan_bytes, flag, before, 0, (size_t)N, rocprim::plus<int32_t>(), st);

    const unsigned nb = (unsigned)cdiv(N, 256);
    if (hipMemsetAsync(flag, 0, (size_t)N * 4, st) != hipSuccess) return check_launch("papc_points_to_voxel_f32: memset");
    if (hipMemsetAsync(voxels, 0, (size_t)max_voxels * max_points * ndim * 4, st) != hipSuccess) return check_launch("papc_points_to_voxel_f32: memset");
    if (hipMemsetAsync(coors, 0, (size_t)max_voxels * 3 * 4, st) != hipSuccess) return check_launch("papc_points_to_voxel_f32: memset");
    if (hipMemsetAsync(num_points, 0, (size_t)max_voxels * 4, st) != hipSuccess) return check_launch("papc_points_to_voxel_f32: memset");
    if (hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(cut), N, 1, st) != hipSuccess) return check_launch("papc_points_to_voxel_f32: init");   // no break
    if (hipMemsetAsync(cut + 1, 0, 4, st) != hipSuccess) return check_launch("papc_points_to_voxel_f32: init");
    hipLaunchKernelGGL(vox_key_kernel, dim3(nb), dim3(256), 0, st, points, N, ndim, v, keys);
    if (rocprim::radix_sort_keys(tmp, sort_bytes, keys, sorted, (size_t)N, 0, 64, st) != hipSuccess) return check_launch("papc_points_to_voxel_f32: sort");
    hipLaunchKernelGGL(vox_mark_kernel, dim3(nb), dim3(256), 0, st, sorted, N, flag);
    if (rocprim::exclusive_scan(tmp, scan_bytes, flag, before, 0, (size_t)N, rocprim::plus<int32_t>(), st) != hipSuccess) return check_launch("papc_points_to_voxel_f32: scan");
    hipLaunchKernelGGL(vox_cut_kernel, dim3(nb), dim3(256), 0, st, flag, before, N, max_voxels, cut);
    hipLaunchKernelGGL(vox_emit_kernel, dim3(nb), dim3(256), 0, st, points, sorted, N, ndim, v, before, cut, max_points, max_voxels, voxels,
                       coors, num_points);
    if (hipMemcpyAsync(voxel_num, cut + 1, 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return check_launch("papc_points_to_voxel_f32: copy");
    return check_launch("papc_points_to_voxel_f32");
}

int papc_pillar_scatter_f32(const float *voxel_features, const int32_t *coords, int P, int C, int batch_size, int ny, int nx,
                            float *canvas, int32_t *owner, papc_stream_t stream)
{
    PAPC_REQUIRE(voxel_features && coords && canvas && owner, PAPC_E_INVALID, "papc_pillar_scatter_f32: null pointer");
    PAPC_REQUIRE(P >= 0 && C >= 1 && batch_size >= 1 && ny >= 1 && nx >= 1, PAPC_E_INVALID, "papc_pillar_scatter_f32: bad sizes");
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_PFN, st);
    const size_t cells = (size_t)batch_size * ny * nx;
    if (hipMemsetAsync(canvas, 0, cells * C * 4, st) != hipSuccess) return check_launch("papc_pillar_scatter_f32: memset");   // paddle.zeros (:125)
    if (hipMemsetAsync(owner, 0xFF, cells * 4, st) != hipSuccess) return check_launch("papc_pillar_scatter_f32: memset");     // -1
    if (P > 0) {
        hipLaunchKernelGGL(scatter_owner_kernel, dim3((unsigned)cdiv(P, 256)), dim3(256), 0, st, coords, P, batch_size, ny, nx, owner);
        hipLaunchKernelGGL(scatter_move_kernel<false>, dim3((unsigned)cdiv(P, 4)), dim3(256), 0, st, voxel_features, coords, owner, P, C,
                           batch_size, ny, nx, canvas);
    }
    return check_launch("papc_pillar_scatter_f32");
}

int papc_pillar_scatter_bwd_f32(const float *grad_canvas, const int32_t *coords, const int32_t *owner, int P, int C,
                                int batch_size, int ny, int nx, float *grad_features, papc_stream_t stream)
{
    PAPC_REQUIRE(grad_canvas && coords && owner && grad_features, PAPC_E_INVALID, "papc_pillar_scatter_bwd_f32: null pointer");
    PAPC_REQUIRE(P >= 1 && C >= 1 && batch_size >= 1 && ny >= 1 && nx >= 1, PAPC_E_INVALID, "papc_pillar_scatter_bwd_f32: bad sizes");
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_PFN, st);
    hipLaunchKernelGGL(scatter_move_kernel<true>, dim3((unsigned)cdiv(P, 4)), dim3(256), 0, st, grad_canvas, coords, owner, P, C, batch_size,
                       ny, nx, grad_features);
    return check_launch("papc_pillar_scatter_bwd_f32");
}

}  // extern "C"
