// sampling.hip -- farthest-point sampling, ball query, gather/group, square_distance for gfx950.
//
// Index-exact kernels: compiled with -ffp-contract=off; every FMA below is explicit (fmaf) and mirrors the
// canonical arithmetic of the reference's tensor expressions
// (/root/reference/PAPC/models/layers/pointnet2_basic_layers.py:26-40, :65-95, :98-126).
#include "common.h"

namespace papc {

typedef unsigned long long u64;
typedef unsigned int u32;

// =====================================================================================================
// FPS  (pointnet2_basic_layers.py:65-95)
//
// One workgroup per cloud (the npoint-long argmax chain is serial; a cloud never spans CUs), and the CU's vector ALUs are what
// bounds an iteration: N points x (distance, running minimum, argmax) on 4 x 16 lanes.  So the loop is written for instruction
// count.  Each lane keeps PPT points in VGPRs for the whole kernel, as PAIRS: the distance update runs on packed f32 instructions
// (v_pk_add_f32 / v_pk_mul_f32: two points per instruction, each element rounded exactly like the scalar op -- the file is built
// with -ffp-contract=off, nothing fuses).  A copy of xyz sits in LDS so the winner's coordinates are one broadcast ds_read away.
// Running distances live as the BIT PATTERNS of non-negative floats: unsigned min / max / compare equal the float ones.
//
// argmax(distance, -1) returns the FIRST maximum (:93) and that matters: ties at exactly 1.0 happen in the first iterations of
// every unit-sphere cloud because the running distance starts at 1.0 (:75).  Per iteration:
//   thread  max over its PPT points (v_max3_u32 tree) + the lowest j holding it -> candidate index j T + tid
//   row     a 16-lane DPP max (four steps); the lanes holding their row's maximum -- one per row unless distances tie -- fold
//           {max bits : ~index} into ONE 64-bit LDS word with ds_max_u64 (larger distance wins, then the smaller index)
//   group   one barrier, one broadcast read of the word.  Three words rotate: the word of iteration it + 2 is cleared after the
//           barrier of iteration it, when its last readers (iteration it - 1) are behind that barrier and its next writers two
//           barriers away.
//   (a single-wave group skips LDS: a 64-lane DPP max, a ballot of the lanes holding it -- almost always one lane, whose candidate
//   is one v_readlane away; several lanes: a DPP min over their candidates)
// =====================================================================================================
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int T, int PPT, bool LDS_XYZ>
__global__ __launch_bounds__(T) void fps_kernel(const float *__restrict__ xyz, int64_t sb, int64_t sn, int64_t sc,
                                                int N, int npoint, const int64_t *__restrict__ start,
                                                float init_dist, int32_t *__restrict__ out_idx,
                                                float *__restrict__ out_new_xyz)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u64 *key = reinterpret_cast<u64 *>(smem);        // [3] {max distance bits : ~index}
    float *sx = reinterpret_cast<float *>(smem + 256);
    float *sy = sx + N;
    float *sz = sy + N;

    constexpr int NW = T / 64;
    constexpr int NP = (PPT + 1) / 2, PP = 2 * NP;   // pairs of points per lane (PPT = 1: the second element is padding)
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    const float *p = xyz + (int64_t)b * sb;

    f32x2 x[NP], y[NP], z[NP];
    u32 d[PP];
#pragma unroll
    for (int j = 0; j < PP; ++j) {
        const int i = j * T + tid;
        float px = 0.f, py = 0.f, pz = 0.f;
        d[j] = 0u;    // padding: distance pinned at +0 and index >= N, so it never wins a max against a real point
        if (j < PPT && i < N) {
            px = p[(int64_t)i * sn];
            py = p[(int64_t)i * sn + sc];
            pz = p[(int64_t)i * sn + 2 * sc];
            d[j] = __float_as_uint(init_dist);
            if (LDS_XYZ) { sx[i] = px; sy[i] = py; sz[i] = pz; }
        }
        x[j >> 1][j & 1] = px; y[j >> 1][j & 1] = py; z[j >> 1][j & 1] = pz;
    }
    if (tid < 3) key[tid] = 0ull;
    int far = (int)start[b];
    __syncthreads();

    int s0 = 0, s2 = 2;      // it % 3, (it + 2) % 3
    const u32 key_lds = (u32)(uintptr_t)key;
    for (int it = 0; it < npoint; ++it) {
        float cx, cy, cz;
        if (LDS_XYZ) { cx = sx[far]; cy = sy[far]; cz = sz[far]; }
        else { cx = p[(int64_t)far * sn]; cy = p[(int64_t)far * sn + sc]; cz = p[(int64_t)far * sn + 2 * sc]; }
        if (tid == 0) {
            out_idx[(int64_t)b * npoint + it] = far;  // centroids[:, i] = farthest  (:80)
            if (out_new_xyz) {
                float *o = out_new_xyz + ((int64_t)b * npoint + it) * 3;
                o[0] = cx; o[1] = cy; o[2] = cz;
            }
        }
        const f32x2 c2x = {cx, cx}, c2y = {cy, cy}, c2z = {cz, cz};
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const f32x2 dx = x[q] - c2x, dy = y[q] - c2y, dz = z[q] - c2z;
            const f32x2 dd = (dx * dx + dy * dy) + dz * dz;          // sum((xyz - centroid) ** 2, -1)  (:86)
            const u32 a0 = __float_as_uint(dd[0]), a1 = __float_as_uint(dd[1]);
            d[2 * q] = a0 < d[2 * q] ? a0 : d[2 * q];                // where(dist < distance, dist, distance) (:87-92)
            d[2 * q + 1] = a1 < d[2 * q + 1] ? a1 : d[2 * q + 1];
        }
        // per-thread argmax: max tree + the lowest j holding the max (strict first-maximum rule, :93)
        u32 bmax = d[0];
#pragma unroll
        for (int j = 1; j < PP; ++j) bmax = d[j] > bmax ? d[j] : bmax;
        int bestj = PP - 1;
#pragma unroll
        for (int j = PP - 2; j >= 0; --j) bestj = d[j] == bmax ? j : bestj;
        const u32 cand = (u32)(bestj * T + tid);
        if (NW == 1) {
            // one wave: max distance, then the lowest candidate among the lanes holding it (almost always a single lane)
            const u32 mwb = readlane63_u32(wave_max_u32_fused_to_lane63(bmax));
            const u64 holders = __ballot(bmax == mwb);
            u32 iw;
            if (__builtin_popcountll(holders) == 1) iw = (u32)__builtin_amdgcn_readlane((int)cand, __builtin_ctzll(holders));
            else iw = readlane63_u32(wave_min_u32_fused_to_lane63(bmax == mwb ? cand : 0xFFFFFFFFu));
            far = (int)iw;
        } else {
            // 16-lane row maximum (four DPP steps, every lane of the row ends with it); the lanes holding it -- one per row unless
            // distances tie -- fold {distance bits : ~candidate} into the iteration's LDS word
            const u32 rmax = row_max_u32_fused(bmax);
            if (bmax == rmax) {
                const u64 kv = ((u64)bmax << 32) | (u64)(~cand);
                asm volatile("ds_max_u64 %0, %1" ::"v"(key_lds + 8u * (u32)s0), "v"(kv) : "memory");
            }
            lds_barrier();
            const u64 k = key[s0];
            far = (int)~(u32)__builtin_amdgcn_readfirstlane((int)(u32)k);
            if (tid == 0) key[s2] = 0ull;
            s2 = s0;                       // (it + 3) % 3
            s0 = s0 == 2 ? 0 : s0 + 1;
        }
    }
}

// ---- clouds beyond the register-resident kernel (N > 16 384, up to 131 072: a KITTI sweep) -----------------------------------------
// Same arithmetic, same first-maximum rule; what changes is where things live: 1024 threads per cloud, the running distances of a thread's PPT
// points in its registers (as bit patterns), the coordinates re-read every iteration (the cloud is L2-resident: 12 N bytes), the candidate
// {distance bits : ~index} keys folded by six 64-bit shuffle steps per wave and one LDS row per iteration parity (one barrier per iteration).
// A capability path, not a tuned one: ~PPT x 3 loads per thread and iteration.
template <int PPT>
__global__ __launch_bounds__(1024) void fps_big_kernel(const float *__restrict__ xyz, int64_t sb, int64_t sn, int64_t sc, int N, int npoint,
                                                       const int64_t *__restrict__ start, float init_dist, int32_t *__restrict__ out_idx,
                                                       float *__restrict__ out_new_xyz)
{
    __shared__ u64 wkey[2][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
    const float *p = xyz + (int64_t)b * sb;
    u32 d[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) d[j] = (j * 1024 + tid < N) ? __float_as_uint(init_dist) : 0u;
    int far = (int)start[b];
    for (int it = 0; it < npoint; ++it) {
        const float cx = p[(int64_t)far * sn], cy = p[(int64_t)far * sn + sc], cz = p[(int64_t)far * sn + 2 * sc];
        if (tid == 0) {
            out_idx[(int64_t)b * npoint + it] = far;
            if (out_new_xyz) {
                float *o = out_new_xyz + ((int64_t)b * npoint + it) * 3;
                o[0] = cx; o[1] = cy; o[2] = cz;
            }
        }
        u64 best = 0ull;
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const int i = j * 1024 + tid;
            if (i < N) {
                const float dx = p[(int64_t)i * sn] - cx, dy = p[(int64_t)i * sn + sc] - cy, dz = p[(int64_t)i * sn + 2 * sc] - cz;
                const u32 a = __float_as_uint((dx * dx + dy * dy) + dz * dz);
                d[j] = a < d[j] ? a : d[j];
                const u64 k = ((u64)d[j] << 32) | (u64)(u32)~(u32)i;       // larger distance wins, then the smaller index
                best = k > best ? k : best;
            }
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const u32 lo = (u32)__shfl_xor((int)(u32)best, m), hi = (u32)__shfl_xor((int)(u32)(best >> 32), m);
            const u64 o = ((u64)hi << 32) | lo;
            best = o > best ? o : best;
        }
        if (lane == 0) wkey[it & 1][wave] = best;
        __syncthreads();
        u64 k = wkey[it & 1][0];
#pragma unroll
        for (int w = 1; w < 16; ++w) { const u64 o = wkey[it & 1][w]; k = o > k ? o : k; }
        far = (int)~(u32)k;
    }
}

template <int T, int PPT>
static int launch_fps(const float *xyz, int64_t sb, int64_t sn, int64_t sc, int B, int N, int npoint,
                      const int64_t *start, float init_dist, int32_t *out_idx, float *out_new_xyz, hipStream_t st)
{
    const size_t lds_full = 256 + (size_t)N * 12;
    if (lds_full <= 150 * 1024) {
        auto kern = fps_kernel<T, PPT, true>;
        if (lds_full > 48 * 1024) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds_full) != hipSuccess)
                return check_launch("papc_fps_f32: hipFuncSetAttribute");
        }
        hipLaunchKernelGGL(kern, dim3(B), dim3(T), lds_full, st, xyz, sb, sn, sc, N, npoint, start, init_dist,
                           out_idx, out_new_xyz);
    } else {
        hipLaunchKernelGGL((fps_kernel<T, PPT, false>), dim3(B), dim3(T), 256, st, xyz, sb, sn, sc, N, npoint,
                           start, init_dist, out_idx, out_new_xyz);
    }
    return check_launch("papc_fps_f32");
}

// =====================================================================================================
// Ball query (pointnet2_basic_layers.py:98-126), all radii of an MSG layer in one scan (:260-262).
//
// The reference materialises a [B,S,N] int64 tile, masks it and SORTS it; the net result is "first nsample
// in-radius indices, ascending, padded with the first".  Here: one 64-lane wave per query tests 64 points
// per step against every radius, __ballot + mbcnt give each in-radius lane its output slot (ordered
// compaction), a popcount advances the per-radius counter, and the wave leaves as soon as all radii are
// full.  The cloud is staged once per workgroup into LDS as float4 (x,y,z,|p|^2) and shared by the
// workgroup's 64 queries.  No [B,S,N] matrix, no sort.
// =====================================================================================================
struct BQParams {
    float thr[4];
    int ns[4];
    void *out[4];
};

constexpr int BQ_T = 1024;       // threads per workgroup (16 waves)
constexpr int BQ_QPW = 4;        // queries per wave
constexpr int BQ_QPB = (BQ_T / 64) * BQ_QPW;
constexpr int BQ_CHUNK = 8192;   // points staged per pass (128 KiB of float4)

template <int NR, typename IdxT>
__global__ __launch_bounds__(BQ_T) void ball_query_kernel(const float *__restrict__ xyz, int64_t sb, int64_t sn,
                                                          int64_t sc, const float *__restrict__ new_xyz, int N,
                                                          int S, BQParams prm, int chunk)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4 *pts = reinterpret_cast<float4 *>(smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int b = blockIdx.y;
    const int q0 = blockIdx.x * BQ_QPB + wave * BQ_QPW;
    const float *p = xyz + (int64_t)b * sb;

    float qx[BQ_QPW], qy[BQ_QPW], qz[BQ_QPW], aa[BQ_QPW];
    int cnt[BQ_QPW][NR], first[BQ_QPW][NR];
#pragma unroll
    for (int qi = 0; qi < BQ_QPW; ++qi) {
        const int q = q0 + qi;
        if (q < S) {
            const float *qp = new_xyz + ((int64_t)b * S + q) * 3;
            qx[qi] = qp[0]; qy[qi] = qp[1]; qz[qi] = qp[2];
        } else { qx[qi] = 0.f; qy[qi] = 0.f; qz[qi] = 0.f; }
        aa[qi] = (qx[qi] * qx[qi] + qy[qi] * qy[qi]) + qz[qi] * qz[qi];  // sum(src ** 2, -1)  (:37)
#pragma unroll
        for (int r = 0; r < NR; ++r) { cnt[qi][r] = 0; first[qi][r] = N; }
    }

    for (int c0 = 0; c0 < N; c0 += chunk) {
        const int n_in = min(chunk, N - c0);
        if (c0 > 0) __syncthreads();
        for (int i = tid; i < n_in; i += BQ_T) {
            const int64_t g = (int64_t)(c0 + i) * sn;
            const float x = p[g], y = p[g + sc], z = p[g + 2 * sc];
            pts[i] = make_float4(x, y, z, (x * x + y * y) + z * z);  // sum(dst ** 2, -1)  (:38)
        }
        __syncthreads();

#pragma unroll
        for (int qi = 0; qi < BQ_QPW; ++qi) {
            const int q = q0 + qi;
            if (q >= S) continue;
            bool full = true;
#pragma unroll
            for (int r = 0; r < NR; ++r) full = full && (cnt[qi][r] >= prm.ns[r]);
            if (full) continue;
            for (int base = 0; base < n_in; base += 64) {
                const int i = base + lane;
                const bool valid = i < n_in;
                const float4 pt = pts[valid ? i : n_in - 1];
                // dist = -2 * matmul(src, dst^T); dist += |src|^2; dist += |dst|^2   (:36-38)
                const float dot = fmaf(qz[qi], pt.z, fmaf(qy[qi], pt.y, qx[qi] * pt.x));
                float dist = -2.0f * dot;
                dist = dist + aa[qi];
                dist = dist + pt.w;
                bool all_full = true;
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    if (cnt[qi][r] < prm.ns[r]) {
                        const bool in = valid && !(dist > prm.thr[r]);  // mask = sqrdists > radius ** 2 (:112)
                        const u64 m = __ballot(in);
                        if (m) {
                            const int pos = cnt[qi][r] +
                                            (int)__builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0));
                            if (in && pos < prm.ns[r])
                                reinterpret_cast<IdxT *>(prm.out[r])[((int64_t)b * S + q) * prm.ns[r] + pos] = (IdxT)(c0 + i);
                            if (cnt[qi][r] == 0) first[qi][r] = c0 + base + (int)__builtin_ctzll(m);
                            cnt[qi][r] += (int)__builtin_popcountll(m);
                        }
                        all_full = all_full && (cnt[qi][r] >= prm.ns[r]);
                    }
                }
                if (all_full) break;
            }
        }
    }

    // group_idx[mask] = group_first[mask]  (:118-124); with no hit at all every slot keeps N (:115)
#pragma unroll
    for (int qi = 0; qi < BQ_QPW; ++qi) {
        const int q = q0 + qi;
        if (q >= S) continue;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int have = min(cnt[qi][r], prm.ns[r]);
            const int fill = cnt[qi][r] ? first[qi][r] : N;
            IdxT *o = reinterpret_cast<IdxT *>(prm.out[r]) + ((int64_t)b * S + q) * prm.ns[r];
            for (int pos = have + lane; pos < prm.ns[r]; pos += 64) o[pos] = (IdxT)fill;
        }
    }
}

template <int NR, typename IdxT>
static int launch_bq(const float *xyz, int64_t sb, int64_t sn, int64_t sc, const float *new_xyz, int B, int N, int S,
                     const BQParams &prm, hipStream_t st)
{
    const int chunk = N < BQ_CHUNK ? N : BQ_CHUNK;
    const size_t lds = (size_t)chunk * 16;
    auto kern = ball_query_kernel<NR, IdxT>;
    if (lds > 48 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return check_launch("papc_ball_query_f32: hipFuncSetAttribute");
    }
    dim3 grid((unsigned)cdiv(S, BQ_QPB), (unsigned)B);
    hipLaunchKernelGGL(kern, grid, dim3(BQ_T), lds, st, xyz, sb, sn, sc, new_xyz, N, S, prm, chunk);
    return check_launch("papc_ball_query_f32");
}

// =====================================================================================================
// square_distance (:26-40) -- API parity only; the hot path never materialises [B,N,M].
// =====================================================================================================
__global__ void square_distance_kernel(const float *__restrict__ src, const float *__restrict__ dst, int N, int M,
                                       float *__restrict__ out)
{
    const int b = blockIdx.z, i = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= M) return;
    const float *a = src + ((int64_t)b * N + i) * 3;
    const float *q = dst + ((int64_t)b * M + j) * 3;
    const float a0 = a[0], a1 = a[1], a2 = a[2], b0 = q[0], b1 = q[1], b2 = q[2];
    const float dot = fmaf(a2, b2, fmaf(a1, b1, a0 * b0));
    float d = -2.0f * dot;
    d = d + ((a0 * a0 + a1 * a1) + a2 * a2);
    d = d + ((b0 * b0 + b1 * b1) + b2 * b2);
    out[((int64_t)b * N + i) * M + j] = d;
}

// =====================================================================================================
// index_points (:43-62) and its gradient
// =====================================================================================================
template <typename IdxT>
__global__ void index_points_kernel(const float *__restrict__ points, const IdxT *__restrict__ idx, int N, int C,
                                    int S, int64_t total, float *__restrict__ out)
{
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = e / C;
        const int c = (int)(e - row * C);
        const int64_t b = row / S;
        const int64_t j = (int64_t)idx[row];
        out[e] = (j >= 0 && j < N) ? points[(b * N + j) * C + c] : 0.f;
    }
}

template <typename IdxT>
__global__ void index_points_bwd_kernel(const float *__restrict__ gout, const IdxT *__restrict__ idx, int N, int C,
                                        int S, int64_t total, float *__restrict__ gpoints)
{
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = e / C;
        const int c = (int)(e - row * C);
        const int64_t b = row / S;
        const int64_t j = (int64_t)idx[row];
        if (j >= 0 && j < N) atomicAdd(&gpoints[(b * N + j) * C + c], gout[e]);
    }
}

// =====================================================================================================
// sample_and_group's gather / centre / concat (:146-153; MSG order :263-269)
// =====================================================================================================
__global__ void group_points_kernel(const float *__restrict__ xyz, int64_t sb, int64_t sn, int64_t sc,
                                    const float *__restrict__ new_xyz, const float *__restrict__ feats,
                                    const int32_t *__restrict__ idx, int N, int S, int K, int D, int xyz_first,
                                    int64_t total, float *__restrict__ out)
{
    const int C = 3 + D;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = e / C;  // (b,s,k)
        const int c = (int)(e - row * C);
        const int64_t bs = row / K;
        const int64_t b = bs / S;
        const int j = idx[row];
        const int cx = xyz_first ? c : c - D;  // coordinate index when this column is an xyz column
        const bool is_xyz = xyz_first ? (c < 3) : (c >= D);
        float v = 0.f;
        if (j >= 0 && j < N) {
            if (is_xyz) v = xyz[b * sb + (int64_t)j * sn + cx * sc] - new_xyz[bs * 3 + cx];  // grouped_xyz - new_xyz (:147)
            else v = feats[(b * N + j) * D + (xyz_first ? c - 3 : c)];
        }
        out[e] = v;
    }
}

// gradient of the gather / centre / concat: every (b, s, k) row scatters its feature columns to the point it gathered (float
// atomics: a point sits in many neighbourhoods), its coordinate columns to that point's xyz and, negated, to its centre
__global__ void group_points_bwd_kernel(const float *__restrict__ gout, const int32_t *__restrict__ idx, int N, int S, int K, int D,
                                        int xyz_first, int64_t total, float *__restrict__ gfeats, float *__restrict__ gxyz)
{
    const int C = 3 + D;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = e / C;
        const int c = (int)(e - row * C);
        const int64_t b = row / ((int64_t)S * K);
        const int j = idx[row];
        if (j < 0 || j >= N) continue;
        const bool is_xyz = xyz_first ? (c < 3) : (c >= D);
        if (is_xyz) {
            if (gxyz) atomicAdd(&gxyz[(b * N + j) * 3 + (xyz_first ? c : c - D)], gout[e]);
        } else if (gfeats) {
            atomicAdd(&gfeats[(b * N + j) * D + (xyz_first ? c - 3 : c)], gout[e]);
        }
    }
}
__global__ void group_centre_bwd_kernel(const float *__restrict__ gout, int K, int D, int xyz_first, int64_t total, float *__restrict__ gnew)
{
    const int C = 3 + D;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // (b, s, coordinate)
    if (e >= total) return;
    const int64_t bs = e / 3;
    const int x = (int)(e - bs * 3);
    const float *g = gout + bs * K * C + (xyz_first ? x : D + x);
    float acc = 0.f;
    for (int k = 0; k < K; ++k) acc += g[(int64_t)k * C];
    gnew[e] = -acc;
}

}  // namespace papc

using namespace papc;

extern "C" {

int papc_fps_f32(const float *xyz, int64_t sb, int64_t sn, int64_t sc, int B, int N, int npoint,
                 const int64_t *start_idx, float init_dist, int32_t *out_idx, float *out_new_xyz,
                 papc_stream_t stream)
{
    PAPC_REQUIRE(xyz && start_idx && out_idx, PAPC_E_INVALID, "papc_fps_f32: null pointer");
    PAPC_REQUIRE(B >= 1 && N >= 1 && npoint >= 1, PAPC_E_INVALID, "papc_fps_f32: B=%d N=%d npoint=%d must be >= 1", B, N, npoint);
    PAPC_REQUIRE(init_dist >= 0.f, PAPC_E_INVALID, "papc_fps_f32: init_dist must be >= 0");
    PAPC_REQUIRE(N <= 131072, PAPC_E_UNSUPPORTED, "papc_fps_f32: N=%d > 131072 points per cloud not supported", N);
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_FPS, st);
    if (N > 16384) {     // beyond the register-resident kernel: distances in registers, coordinates from L2 (fps_big_kernel)
        if (N <= 32768) hipLaunchKernelGGL(fps_big_kernel<32>, dim3(B), dim3(1024), 0, st, xyz, sb, sn, sc, N, npoint, start_idx, init_dist, out_idx, out_new_xyz);
        else if (N <= 65536) hipLaunchKernelGGL(fps_big_kernel<64>, dim3(B), dim3(1024), 0, st, xyz, sb, sn, sc, N, npoint, start_idx, init_dist, out_idx, out_new_xyz);
        else hipLaunchKernelGGL(fps_big_kernel<128>, dim3(B), dim3(1024), 0, st, xyz, sb, sn, sc, N, npoint, start_idx, init_dist, out_idx, out_new_xyz);
        return check_launch("papc_fps_f32 (large cloud)");
    }
    // geometry (measured on MI355X, tools/probe/fps_time.py; us per iteration): up to 512 points ONE wave with 8 points per lane (0.32; no
    // barrier, no LDS word); from 1024 points 512 threads (N = 1024: 0.32 against 0.35 at 256 threads; N = 2048: 0.40 / 0.42; N = 4096:
    // 0.53, two waves per SIMD hide each other's DPP / LDS latency -- 1024 threads measure the same, 256 are 12 % slower)
    int T = N <= 512 ? 64 : (N < 1024 ? 256 : 512);
    { const int t = knob(N <= 1024 ? KNOB_FPS_THREADS_SMALL : KNOB_FPS_THREADS);   // tuning knobs (small / large clouds)
      if (t == 64 || t == 128 || t == 256 || t == 512 || t == 1024) T = t; }
    int ppt = 1;
    while ((int64_t)T * ppt < N) ppt *= 2;
    while (ppt > 16 && T < 1024) { T *= 2; ppt = 1; while ((int64_t)T * ppt < N) ppt *= 2; }
    PAPC_REQUIRE(ppt <= 16, PAPC_E_UNSUPPORTED, "papc_fps_f32: geometry T=%d ppt=%d", T, ppt);
#define FPS_CASE(TT, PP) \
    if (T == TT && ppt == PP) return launch_fps<TT, PP>(xyz, sb, sn, sc, B, N, npoint, start_idx, init_dist, out_idx, out_new_xyz, st);
#define FPS_ROW(TT) FPS_CASE(TT, 1) FPS_CASE(TT, 2) FPS_CASE(TT, 4) FPS_CASE(TT, 8) FPS_CASE(TT, 16)
    FPS_ROW(64) FPS_ROW(128) FPS_ROW(256) FPS_ROW(512) FPS_ROW(1024)
#undef FPS_ROW
#undef FPS_CASE
    set_error("papc_fps_f32: no kernel for T=%d ppt=%d", T, ppt);
    return PAPC_E_UNSUPPORTED;
}

int papc_ball_query_f32(const float *xyz, int64_t sb, int64_t sn, int64_t sc, const float *new_xyz, int B,
                        int N, int S, int n_radii, const float *thr, const int *nsample,
                        void *const *out_idx, int idx64, papc_stream_t stream)
{
    PAPC_REQUIRE(xyz && new_xyz && thr && nsample && out_idx, PAPC_E_INVALID, "papc_ball_query_f32: null pointer");
    PAPC_REQUIRE(B >= 1 && N >= 1 && S >= 1, PAPC_E_INVALID, "papc_ball_query_f32: B=%d N=%d S=%d must be >= 1", B, N, S);
    PAPC_REQUIRE(n_radii >= 1 && n_radii <= 4, PAPC_E_INVALID, "papc_ball_query_f32: n_radii=%d not in [1,4]", n_radii);
    BQParams prm;
    for (int r = 0; r < 4; ++r) { prm.thr[r] = 0.f; prm.ns[r] = 0; prm.out[r] = nullptr; }
    for (int r = 0; r < n_radii; ++r) {
        PAPC_REQUIRE(nsample[r] >= 1 && out_idx[r], PAPC_E_INVALID, "papc_ball_query_f32: radius %d: nsample=%d / null out", r, nsample[r]);
        prm.thr[r] = thr[r]; prm.ns[r] = nsample[r]; prm.out[r] = out_idx[r];
    }
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_BALL_QUERY, st);
#define BQ_CASE(NR) \
    if (n_radii == NR) return idx64 ? launch_bq<NR, int64_t>(xyz, sb, sn, sc, new_xyz, B, N, S, prm, st) \
                                    : launch_bq<NR, int32_t>(xyz, sb, sn, sc, new_xyz, B, N, S, prm, st);
    BQ_CASE(1) BQ_CASE(2) BQ_CASE(3) BQ_CASE(4)
#undef BQ_CASE
    return PAPC_E_INVALID;
}

int papc_square_distance_f32(const float *src, const float *dst, int B, int N, int M, float *out, papc_stream_t stream)
{
    PAPC_REQUIRE(src && dst && out, PAPC_E_INVALID, "papc_square_distance_f32: null pointer");
    PAPC_REQUIRE(B >= 1 && N >= 1 && M >= 1 && N <= 65535 && B <= 65535, PAPC_E_INVALID, "papc_square_distance_f32: bad sizes");
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_MISC, st);
    dim3 grid((unsigned)cdiv(M, 256), (unsigned)N, (unsigned)B);
    hipLaunchKernelGGL(square_distance_kernel, grid, dim3(256), 0, st, src, dst, N, M, out);
    return check_launch("papc_square_distance_f32");
}

static inline unsigned ew_grid(int64_t total) { return (unsigned)std::min<int64_t>(cdiv(total, 256), 256 * 16); }

int papc_index_points_f32(const float *points, const void *idx, int idx64, int B, int N, int C, int S, float *out,
                          papc_stream_t stream)
{
    PAPC_REQUIRE(points && idx && out, PAPC_E_INVALID, "papc_index_points_f32: null pointer");
    PAPC_REQUIRE(B >= 1 && N >= 1 && C >= 1 && S >= 1, PAPC_E_INVALID, "papc_index_points_f32: bad sizes");
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_GROUP, st);
    const int64_t total = (int64_t)B * S * C;
    if (idx64) hipLaunchKernelGGL(index_points_kernel<int64_t>, dim3(ew_grid(total)), dim3(256), 0, st, points, (const int64_t *)idx, N, C, S, total, out);
    else hipLaunchKernelGGL(index_points_kernel<int32_t>, dim3(ew_grid(total)), dim3(256), 0, st, points, (const int32_t *)idx, N, C, S, total, out);
    return check_launch("papc_index_points_f32");
}

int papc_index_points_bwd_f32(const float *grad_out, const void *idx, int idx64, int B, int N, int C, int S,
                              float *grad_points, papc_stream_t stream)
{
    PAPC_REQUIRE(grad_out && idx && grad_points, PAPC_E_INVALID, "papc_index_points_bwd_f32: null pointer");
    PAPC_REQUIRE(B >= 1 && N >= 1 && C >= 1 && S >= 1, PAPC_E_INVALID, "papc_index_points_bwd_f32: bad sizes");
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_GROUP, st);
    const int64_t total = (int64_t)B * S * C;
    if (idx64) hipLaunchKernelGGL(index_points_bwd_kernel<int64_t>, dim3(ew_grid(total)), dim3(256), 0, st, grad_out, (const int64_t *)idx, N, C, S, total, grad_points);
    else hipLaunchKernelGGL(index_points_bwd_kernel<int32_t>, dim3(ew_grid(total)), dim3(256), 0, st, grad_out, (const int32_t *)idx, N, C, S, total, grad_points);
    return check_launch("papc_index_points_bwd_f32");
}

int papc_group_points_f32(const float *xyz, int64_t sb, int64_t sn, int64_t sc, const float *new_xyz,
                          const float *feats, const int32_t *idx, int B, int N, int S, int K, int D,
                          int xyz_first, float *out, papc_stream_t stream)
{
    PAPC_REQUIRE(xyz && new_xyz && idx && out, PAPC_E_INVALID, "papc_group_points_f32: null pointer");
    PAPC_REQUIRE(D == 0 || feats, PAPC_E_INVALID, "papc_group_points_f32: D=%d but feats is null", D);
    PAPC_REQUIRE(B >= 1 && N >= 1 && S >= 1 && K >= 1 && D >= 0, PAPC_E_INVALID, "papc_group_points_f32: bad sizes");
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_GROUP, st);
    const int64_t total = (int64_t)B * S * K * (3 + D);
    hipLaunchKernelGGL(group_points_kernel, dim3(ew_grid(total)), dim3(256), 0, st, xyz, sb, sn, sc, new_xyz, feats, idx, N, S, K, D, xyz_first, total, out);
    return check_launch("papc_group_points_f32");
}

int papc_group_points_bwd_f32(const float *grad_out, const int32_t *idx, int B, int N, int S, int K, int D, int xyz_first,
                              float *grad_feats, float *grad_xyz, float *grad_new_xyz, papc_stream_t stream)
{
    PAPC_REQUIRE(grad_out && idx, PAPC_E_INVALID, "papc_group_points_bwd_f32: null pointer");
    PAPC_REQUIRE(B >= 1 && N >= 1 && S >= 1 && K >= 1 && D >= 0, PAPC_E_INVALID, "papc_group_points_bwd_f32: bad sizes");
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_GROUP, st);
    if ((grad_feats && D > 0) || grad_xyz) {
        const int64_t total = (int64_t)B * S * K * (3 + D);
        hipLaunchKernelGGL(group_points_bwd_kernel, dim3(ew_grid(total)), dim3(256), 0, st, grad_out, idx, N, S, K, D, xyz_first, total,
                           D > 0 ? grad_feats : nullptr, grad_xyz);
        const int rc = check_launch("papc_group_points_bwd_f32");
        if (rc) return rc;
    }
    if (grad_new_xyz) {
        const int64_t total = (int64_t)B * S * 3;
        hipLaunchKernelGGL(group_centre_bwd_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, st, grad_out, K, D, xyz_first, total, grad_new_xyz);
        return check_launch("papc_group_points_bwd_f32 (centres)");
    }
    return PAPC_OK;
}

}  // extern "C"
