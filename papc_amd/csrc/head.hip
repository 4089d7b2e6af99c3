// head.hip -- the classifier head behind the set-abstraction pyramid, for gfx950.
//
// Reference: /root/reference/PAPC/models/classify/pointnet2/pointnet2.py:17-23, :37-39 (SSG) and :51-57, :71-73 (MSG):
//   x = drop1(relu(bn1(fc1(x))));  x = drop2(relu(bn2(fc2(x))));  x = fc3(x)      then softmax cross-entropy (mean).
// B (the batch) is tens of rows, so every library op of this chain is a launch-latency-sized kernel (~60 of them, 9 % of the
// training step).  Here a layer is ONE launch forward and ONE launch backward:
//   forward   workgroup = 32 output channels x all B rows.  Linear on v_mfma_f32_32x32x2_f32 (fp32 products, fp32 accumulate)
//             with the K range split over the 8 waves and summed in fixed order through LDS; train-mode BatchNorm statistics
//             are column-local, so the same workgroup normalises, applies ReLU and dropout and updates the running statistics.
//   backward  workgroup = 32 channels of layer l: g = dY_{l+1} . W_{l+1} (this tile's columns), dropout/ReLU/BN backward
//             (again column-local) -> dY_l, then dW_l rows = dY_l^T . X_l, all in the one launch.
// Dropout uses a counter-based hash of (seed, step counter, layer, element) read from device memory, so a captured hipGraph
// draws fresh masks on every replay.
#include "common.h"

namespace papc {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int HT = 512;             // threads (8 waves)
constexpr int HW = 8;               // waves: K split
constexpr int HROWS_MAX = 256;      // batch rows a workgroup can hold in LDS
constexpr int TP = 33;              // LDS tile pitch (floats)
constexpr int HC_STAGE_FLOATS = 16384;   // chain kernels: a [B, C] operand handed over by the previous phase is staged in LDS up to this size (64 KB)

__device__ __forceinline__ float hash_uniform(uint64_t seed, uint64_t counter, uint32_t tag, uint32_t idx)
{
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (counter + 1) + ((uint64_t)tag << 40) + idx;   // splitmix64 finaliser
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (float)(uint32_t)(z >> 40) * (1.0f / 16777216.0f);   // 24 bits -> [0,1)
}

// ---- one launch for a chain of layers (head_chain_fwd_kernel / head_chain_bwd_kernel below) ---------------------------------------------
// Inside a replayed graph a dependent launch costs >= 4.7 us before its first instruction (profiles/r05_*: a one-workgroup kernel with
// nothing to do), and the head is 8 of them around ~3 us of work each.  The chain kernels run the layers as PHASES of one launch with a
// grid barrier between them.  No fence: a device-scope release writes back everything dirty in the XCD's L2 (pfn.hip measured +60 us), so
// what one phase hands the next (activations, dY) is written and read with agent-scope accesses that bypass the per-XCD L2 (COH flavours
// of the loaders below), ordered against the barrier's counter by waiting for the stores' acknowledgements.  The workgroups of a chain
// launch (<= 64) are all resident at once on any part this library targets, so spinning on the counter cannot deadlock.
template <class T>
__device__ __forceinline__ void coh_st(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class T>
__device__ __forceinline__ T coh_ld(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <bool COH>
__device__ __forceinline__ float ld1(const float *p) { if constexpr (COH) return coh_ld(p); else return *p; }
template <bool COH>
__device__ __forceinline__ float4 ld4(const float *p)
{
    if constexpr (COH) return make_float4(coh_ld(p), coh_ld(p + 1), coh_ld(p + 2), coh_ld(p + 3));
    else return *reinterpret_cast<const float4 *>(p);
}
template <bool COH>
__device__ __forceinline__ void st1(float *p, float v) { if constexpr (COH) coh_st(p, v); else *p = v; }

__device__ __forceinline__ void grid_barrier(unsigned *bar, unsigned target)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this thread's agent-scope stores have been acknowledged
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}
// behind the last phase: the last workgroup to leave returns both words to zero for the next launch
__device__ __forceinline__ void grid_exit(unsigned *bar)
{
    if (threadIdx.x == 0) {
        const unsigned t = __hip_atomic_fetch_add(bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == gridDim.x - 1) { coh_st(bar, 0u); coh_st(bar + 1, 0u); }
    }
}

// acc[] += A[rows r0.., k] * Bm[cols c0.., k] over the wave's share of K; both operands K-contiguous (row-major [*, K]).
// lane: row/col = lane & 31, k half = lane >> 5; float4 loads give 4 MFMA steps (any k pairing is valid: the sum is over k).
// COH: A was written by another workgroup of this launch (chain kernels).
template <bool COH>
__device__ __forceinline__ void mfma_kcontig(f32x16 &acc, const float *__restrict__ A, int a_rows, int a_r0, const float *__restrict__ Bm,
                                             int b_rows, int b_r0, int K, int wave, int lane)
{
    const int r = lane & 31, h = lane >> 5;
    const int ar = a_r0 + r, br = b_r0 + r;
    const bool a_ok = ar < a_rows, b_ok = br < b_rows;
    const float *ap = A + (int64_t)(a_ok ? ar : 0) * K;
    const float *bp = Bm + (int64_t)(b_ok ? br : 0) * K;
    const int nblk = (K + 7) >> 3;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    constexpr int NB = 8;
    // a wave owns a CONTIGUOUS k range (its consecutive 32-byte pieces share cache lines); NB blocks per trip: all 2*NB loads
    // in flight before the first MFMA
    const int per = (nblk + HW - 1) / HW, blk_end = min(nblk, (wave + 1) * per);
    for (int blk0 = wave * per; blk0 < blk_end; blk0 += NB) {
        float4 a[NB], b[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int k = (blk0 + j < blk_end ? blk0 + j : nblk) * 8 + 4 * h;        // K % 4 == 0
            const int kc = k < K ? k : 0;
            a[j] = ld4<COH>(ap + kc);
            b[j] = *reinterpret_cast<const float4 *>(bp + kc);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int k = (blk0 + j < blk_end ? blk0 + j : nblk) * 8 + 4 * h;
            const float4 av = (k < K && a_ok) ? a[j] : zero4;
            const float4 bv = (k < K && b_ok) ? b[j] : zero4;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc, 0, 0, 0);
        }
    }
}

// acc[] += G[rows r0.., n] * Wn[n, cols c0..] over the wave's share of n < Cn: G [rows, Cn] row-major (K-contiguous), Wn [Cn, ldw]
template <bool COH>
__device__ __forceinline__ void mfma_g_times_w(f32x16 &acc, const float *__restrict__ G, int g_rows, int g_r0, const float *__restrict__ Wn,
                                               int Cn, int ldw, int c0, int wave, int lane)
{
    const int r = lane & 31, h = lane >> 5;
    const int gr = g_r0 + r, col = c0 + r;
    const bool g_ok = gr < g_rows, c_ok = col < ldw;
    const float *gp = G + (int64_t)(g_ok ? gr : 0) * Cn;
    const float *wp = Wn + (c_ok ? col : 0);
    const int nblk = (Cn + 7) >> 3;
    const int per = (nblk + HW - 1) / HW, blk_end = min(nblk, (wave + 1) * per);
    for (int blk0 = wave * per; blk0 < blk_end; blk0 += 2) {  // contiguous n range per wave, 2 blocks per trip, loads first
        float4 a[2];
        float b[2][4];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = (blk0 + j < blk_end ? blk0 + j : nblk) * 8 + 4 * h;        // Cn % 4 == 0
            const int nc = n < Cn ? n : 0;
            a[j] = ld4<COH>(gp + nc);
#pragma unroll
            for (int i = 0; i < 4; ++i) b[j][i] = wp[(int64_t)(nc + i) * ldw];
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = (blk0 + j < blk_end ? blk0 + j : nblk) * 8 + 4 * h;
            const bool ok = n < Cn;
            const float4 av = (ok && g_ok) ? a[j] : make_float4(0.f, 0.f, 0.f, 0.f);
            const bool bk = ok && c_ok;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bk ? b[j][0] : 0.f, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bk ? b[j][1] : 0.f, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bk ? b[j][2] : 0.f, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bk ? b[j][3] : 0.f, acc, 0, 0, 0);
        }
    }
}

// the 8 waves' partial 32x32 tiles -> tile[(row0 + r) * TP + c] summed in wave order (deterministic)
__device__ __forceinline__ void reduce_waves_to_tile(const f32x16 &acc, float *red, float *tile, int row0, int wave, int lane, int tid)
{
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
        red[(wave * 32 + row) * TP + (lane & 31)] = acc[i];
    }
    __syncthreads();
    for (int e = tid; e < 1024; e += HT) {
        const int row = e >> 5, c = e & 31;
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < HW; ++w) s += red[(w * 32 + row) * TP + c];
        tile[(row0 + row) * TP + c] = s;
    }
    __syncthreads();
}

struct HeadFwd {
    const float *x, *w, *bias, *gamma, *beta;
    int B, Cin, Cout, has_bn;
    float eps, momentum;
    float *running_mean, *running_var;
    int64_t *num_batches_tracked;
    float drop_p;
    const int64_t *rng_state;     // [seed, counter] on the device (null: no dropout)
    int layer_tag;
    int64_t *rng_bump;            // non-null: counter += 1 after this launch's reads (last layer of the head)
    float *y, *mean, *invstd, *out;
    uint8_t *keep;
};

// one workgroup's share (output channels 32 bx ..) of a layer.  CIN: x comes from another workgroup of this launch; COUT: `out` goes to one
template <bool CIN, bool COUT>
__device__ __forceinline__ void head_fwd_body(const HeadFwd &a, const int bx)
{
    extern __shared__ float smem[];
    float *red = smem;                          // [8*32][TP]
    float *tile = smem + HW * 32 * TP;          // [Bpad][TP]
    __shared__ float s_mean[32], s_scale[32], s_shift[32];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int c0 = bx * 32;
    const int nrb = (a.B + 31) >> 5;
    const float *xin = a.x;
    bool staged = false;
    if constexpr (CIN) {
        // x was written by other workgroups of this launch: read past the L2 (agent scope).  Such a load is a ~2 us round trip, so the whole
        // [B, Cin] operand is fetched at once into LDS (every thread's loads in flight together) when it fits, and the k loop then runs from
        // LDS; a larger operand is read in place, coherently, block by block
        const int Bpad0 = nrb * 32;
        if ((int64_t)a.B * a.Cin <= HC_STAGE_FLOATS) {
            float *xs = tile + Bpad0 * TP;
            const int n = a.B * a.Cin;
            for (int e = tid; e < n; e += HT) xs[e] = coh_ld(a.x + e);
            __syncthreads();
            xin = xs;
            staged = true;
        }
    }
    for (int rb = 0; rb < nrb; ++rb) {
        f32x16 acc = {0};
        if (CIN && !staged) mfma_kcontig<true>(acc, xin, a.B, rb * 32, a.w, a.Cout, c0, a.Cin, wave, lane);
        else mfma_kcontig<false>(acc, xin, a.B, rb * 32, a.w, a.Cout, c0, a.Cin, wave, lane);
        reduce_waves_to_tile(acc, red, tile, rb * 32, wave, lane, tid);
    }
    const int ncol = min(32, a.Cout - c0);
    {   // 8 threads per channel: bias, then mean and centred second moment in two passes over the LDS column
        __shared__ float s_part[8][32];
        const int t = tid & 31, q = tid >> 5;             // q < 16; the upper 8 groups idle
        const bool act = q < 8 && t < ncol;
        const float bv = (act && a.bias) ? a.bias[c0 + t] : 0.f;
        float sacc = 0.f;
        if (act) for (int b = q; b < a.B; b += 8) { const float v = tile[b * TP + t] + bv; tile[b * TP + t] = v; sacc += v; }
        if (q < 8) s_part[q][t] = sacc;
        __syncthreads();
        float mean = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) mean += s_part[i][t];
        mean /= (float)a.B;
        __syncthreads();
        float vacc = 0.f;
        if (act && a.has_bn == 1) for (int b = q; b < a.B; b += 8) { const float d = tile[b * TP + t] - mean; vacc += d * d; }
        if (q < 8) s_part[q][t] = vacc;
        __syncthreads();
        if (tid < ncol && a.has_bn == 1) {
            const int c = c0 + tid;
            float v = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) v += s_part[i][tid];
            const float var = v / (float)a.B;
            const float invstd = 1.0f / sqrtf(var + a.eps);
            a.mean[c] = mean;
            a.invstd[c] = invstd;
            if (a.running_mean) {     // paddle.nn.BatchNorm1D train step (classify/pointnet2/pointnet2.py:18,21): the BIASED batch variance goes
                                      // into the running estimate (torch would use the unbiased one) -- the convention of every other
                                      // BatchNorm in this library, and what a .pdparams exchanged with the reference carries
                a.running_mean[c] = (1.f - a.momentum) * a.running_mean[c] + a.momentum * mean;
                a.running_var[c] = (1.f - a.momentum) * a.running_var[c] + a.momentum * var;
            }
            s_mean[tid] = mean;
            s_scale[tid] = invstd * a.gamma[c];
            s_shift[tid] = a.beta[c];
        }
        if (tid < ncol && a.has_bn == 3) {
            // eval-mode BatchNorm1D (model.eval(): classify/pointnet2/pointnet2.py:18,21 are registered layers): the RUNNING statistics normalise,
            // nothing is updated; mean / invstd are handed out for a backward through the frozen norm
            const int c = c0 + tid;
            const float rm = a.running_mean[c];
            const float invstd = 1.0f / sqrtf(a.running_var[c] + a.eps);
            if (a.mean) a.mean[c] = rm;
            if (a.invstd) a.invstd[c] = invstd;
            s_mean[tid] = rm;
            s_scale[tid] = invstd * a.gamma[c];
            s_shift[tid] = a.beta[c];
        }
        if (tid == 0 && bx == 0 && a.num_batches_tracked && a.has_bn == 1) a.num_batches_tracked[0] += 1;
    }
    __syncthreads();
    uint64_t seed = 0, counter = 0;
    const bool drop = a.rng_state != nullptr && a.drop_p > 0.f;
    if (drop) { seed = (uint64_t)a.rng_state[0]; counter = (uint64_t)a.rng_state[1]; }
    const float keep_scale = drop ? 1.0f / (1.0f - a.drop_p) : 1.0f;
    for (int e = tid; e < a.B * 32; e += HT) {
        const int b = e >> 5, t = e & 31;
        if (t >= ncol) continue;
        const int64_t o = (int64_t)b * a.Cout + c0 + t;
        const float yv = tile[b * TP + t];
        if (a.y) a.y[o] = yv;
        float v = yv;
        if (a.has_bn) {                   // 1: BatchNorm + ReLU + dropout; 2: ReLU + dropout (pointnet_base.py:26-33 has no norm in its head); 3: eval-mode norm
            if (a.has_bn != 2) v = (yv - s_mean[t]) * s_scale[t] + s_shift[t];
            v = fmaxf(v, 0.f);
            bool k = true;
            if (drop) k = hash_uniform(seed, counter, (uint32_t)a.layer_tag, (uint32_t)o) >= a.drop_p;
            if (a.keep) a.keep[o] = k ? 1 : 0;
            v = k ? v * keep_scale : 0.f;
        }
        st1<COUT>(a.out + o, v);
    }
    if (a.rng_bump && bx == 0 && tid == 0) a.rng_bump[1] += 1;   // stream order: every reader of this step is done or is this launch
}

__global__ __launch_bounds__(HT) void head_fwd_kernel(HeadFwd a) { head_fwd_body<false, false>(a, (int)blockIdx.x); }

struct HeadBwd {
    const float *gnext;           // [B, Cn]: dY of the next layer (or dlogits)
    const float *wnext;           // [Cn, Cout] next layer's weight, or null: g = gnext itself (then Cn == Cout)
    int Cn;
    const float *out, *y, *mean, *invstd, *gamma;   // this layer's saved forward (has_bn)
    float keep_scale;
    int has_bn;
    const float *x;               // this layer's input [B, Cin], or null: no weight gradient
    int B, Cin, Cout;
    float *dy;                    // [B, Cout] or null
    float *dw, *db, *dgamma, *dbeta;
    int accumulate;
};

// one workgroup's share of a layer's backward: channels 32 bx .., dW column blocks by, by + ny, ...  CIN: gnext comes from another workgroup
// of this launch; COUT: dy goes to one
template <bool CIN, bool COUT>
__device__ __forceinline__ void head_bwd_body(const HeadBwd &a, const int bx, const int by, const int ny)
{
    extern __shared__ float smem[];
    float *red = smem;                          // [8*32][TP]
    float *gt = smem + HW * 32 * TP;            // [Bpad][TP]  g -> masked g -> dY (in place)
    float *xh = gt + ((a.B + 31) & ~31) * TP;   // [Bpad][TP]  x-hat
    __shared__ float s_k1[32], s_k2[32], s_k3[32];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int c0 = bx * 32;
    const int nrb = (a.B + 31) >> 5;
    const int Bpad = nrb * 32;
    const int ncol = min(32, a.Cout - c0);
    if (a.wnext) {
        const float *gin = a.gnext;
        bool staged = false;
        if constexpr (CIN) {      // (as in the forward body: one round trip for the whole [B, Cn] operand)
            if ((int64_t)a.B * a.Cn <= HC_STAGE_FLOATS) {
                float *gs = xh + Bpad * TP;
                const int n = a.B * a.Cn;
                for (int e = tid; e < n; e += HT) gs[e] = coh_ld(a.gnext + e);
                __syncthreads();
                gin = gs;
                staged = true;
            }
        }
        for (int rb = 0; rb < nrb; ++rb) {
            f32x16 acc = {0};
            if (CIN && !staged) mfma_g_times_w<true>(acc, gin, a.B, rb * 32, a.wnext, a.Cn, a.Cout, c0, wave, lane);
            else mfma_g_times_w<false>(acc, gin, a.B, rb * 32, a.wnext, a.Cn, a.Cout, c0, wave, lane);
            reduce_waves_to_tile(acc, red, gt, rb * 32, wave, lane, tid);
        }
    } else {
        for (int e = tid; e < Bpad * 32; e += HT) {
            const int b = e >> 5, t = e & 31;
            gt[b * TP + t] = (b < a.B && t < ncol) ? ld1<CIN>(a.gnext + (int64_t)b * a.Cout + c0 + t) : 0.f;
        }
        __syncthreads();
    }
    if (a.has_bn == 2) {          // ReLU + dropout only: dY = g * keep_scale where the output is alive
        for (int e = tid; e < Bpad * 32; e += HT) {
            const int b = e >> 5, t = e & 31;
            float g = 0.f;
            if (b < a.B && t < ncol) g = a.out[(int64_t)b * a.Cout + c0 + t] > 0.f ? gt[b * TP + t] * a.keep_scale : 0.f;
            gt[b * TP + t] = g;
        }
        __syncthreads();
    } else if (a.has_bn) {
        for (int e = tid; e < Bpad * 32; e += HT) {
            const int b = e >> 5, t = e & 31;
            float g = 0.f, h = 0.f;
            if (b < a.B && t < ncol) {
                const int64_t o = (int64_t)b * a.Cout + c0 + t;
                g = a.out[o] > 0.f ? gt[b * TP + t] * a.keep_scale : 0.f;      // dropout (upscale_in_train) + ReLU backward
                h = (a.y[o] - a.mean[c0 + t]) * a.invstd[c0 + t];
            }
            gt[b * TP + t] = g;
            xh[b * TP + t] = h;
        }
        __syncthreads();
        if (tid < 32) {
            float sg = 0.f, sgx = 0.f;
            for (int b = 0; b < a.B; ++b) { sg += gt[b * TP + tid]; sgx += gt[b * TP + tid] * xh[b * TP + tid]; }
            if (tid < ncol) {
                const int c = c0 + tid;
                if (by == 0) {
                    a.dbeta[c] = (a.accumulate ? a.dbeta[c] : 0.f) + sg;
                    a.dgamma[c] = (a.accumulate ? a.dgamma[c] : 0.f) + sgx;
                }
                s_k1[tid] = a.gamma[c] * a.invstd[c];
            } else {
                s_k1[tid] = 0.f;
            }
            // (3 = eval-mode norm: the statistics are constants, so the batch-mean terms of the train-mode backward vanish)
            s_k2[tid] = a.has_bn == 3 ? 0.f : sg / (float)a.B;
            s_k3[tid] = a.has_bn == 3 ? 0.f : sgx / (float)a.B;
        }
        __syncthreads();
        for (int e = tid; e < Bpad * 32; e += HT) {
            const int b = e >> 5, t = e & 31;
            float d = 0.f;
            if (b < a.B && t < ncol) d = s_k1[t] * ((gt[b * TP + t] - s_k2[t]) - xh[b * TP + t] * s_k3[t]);
            gt[b * TP + t] = d;
        }
        __syncthreads();
    }
    const bool first = by == 0;          // the column-split workgroups recompute dY; only the first one publishes it
    if (a.dy && first) {
        for (int e = tid; e < a.B * 32; e += HT) {
            const int b = e >> 5, t = e & 31;
            if (t < ncol) st1<COUT>(a.dy + (int64_t)b * a.Cout + c0 + t, gt[b * TP + t]);
        }
    }
    if (!a.x) return;
    if (first && tid < ncol && a.db) {
        float s = 0.f;
        for (int b = 0; b < a.B; ++b) s += gt[b * TP + tid];
        a.db[c0 + tid] = (a.accumulate ? a.db[c0 + tid] : 0.f) + s;
    }
    // dW[c0 + r, k] = sum_b dY[b, c0 + r] * X[b, k]: MFMA with the batch as K; one wave per 32-column block of Cin, the
    // column blocks dealt round-robin over (blockIdx.y, wave)
    const int r = lane & 31, h = lane >> 5;
    for (int kb = by * HW + wave; kb * 32 < a.Cin; kb += HW * ny) {
        const int kcol = kb * 32 + r;
        const bool k_ok = kcol < a.Cin;
        const float *xp = a.x + (k_ok ? kcol : 0);
        f32x16 acc = {0};
        float prev[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {          // the accumulate target: issued before the MFMA chain, consumed after it
            const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
            prev[i] = (a.accumulate && row < ncol && k_ok) ? a.dw[(int64_t)(c0 + row) * a.Cin + kcol] : 0.f;
        }
        for (int b0 = 0; b0 < Bpad; b0 += 32) {
            float bv[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int b = b0 + 2 * j + h;
                bv[j] = xp[(int64_t)(b < a.B ? b : 0) * a.Cin];
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int b = b0 + 2 * j + h;
                const float av = gt[b * TP + r];                                       // rows >= B hold zeros
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, (k_ok && b < a.B) ? bv[j] : 0.f, acc, 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
            if (row < ncol && k_ok) a.dw[(int64_t)(c0 + row) * a.Cin + kcol] = prev[i] + acc[i];
        }
    }
}

__global__ __launch_bounds__(HT) void head_bwd_kernel(HeadBwd a) { head_bwd_body<false, false>(a, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.y); }

// mean softmax cross-entropy + its gradient: loss = mean_b (logsumexp(z_b) - z_b[label_b]); dz = (softmax(z) - onehot) / B
// rows at z + b * ldz: the logits in global memory (COH: written by another workgroup of this launch) or a layer body's LDS tile
template <bool COH>
__device__ __forceinline__ void softmax_xent_body(const float *z, int ldz, const int64_t *__restrict__ label, int B, int C, float *__restrict__ loss,
                                                  float *__restrict__ dz, float *part /* [blockDim.x] LDS */)
{
    const int nt = blockDim.x;
    float acc = 0.f;
    for (int b = threadIdx.x; b < B; b += nt) {
        const float *row = z + (int64_t)b * ldz;
        float m = ld1<COH>(row);
        for (int c = 1; c < C; ++c) m = fmaxf(m, ld1<COH>(row + c));
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += expf(ld1<COH>(row + c) - m);
        const float lse = m + logf(s);
        const int64_t y = label[b];
        if (y >= 0 && y < C) acc += lse - ld1<COH>(row + y);
        const float inv = 1.0f / (s * (float)B);
        for (int c = 0; c < C; ++c) {
            float g = expf(ld1<COH>(row + c) - m) * inv;
            if (c == y) g -= 1.0f / (float)B;
            dz[(int64_t)b * C + c] = g;
        }
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    // fixed order, independent of the workgroup size: thread 0 adds the B rows' terms in row order groups (rows b, b + nt, ... sit in one thread)
    if (threadIdx.x == 0) {
        float t = 0.f;
        const int n = B < nt ? B : nt;
        for (int i = 0; i < n; ++i) t += part[i];
        loss[0] = t / (float)B;
    }
}

__global__ __launch_bounds__(256) void softmax_xent_kernel(const float *__restrict__ z, const int64_t *__restrict__ label, int B, int C,
                                                           float *__restrict__ loss, float *__restrict__ dz)
{
    __shared__ float part[256];
    softmax_xent_body<false>(z, C, label, B, C, loss, dz, part);
}

// the same for many rows (per-point segmentation logits: B*N = 32768 rows): one row per thread, any number of workgroups; the
// per-workgroup partial sums of the loss are added with one float atomic each (the loss value may differ in its last bits from
// run to run; the gradient does not)
__global__ __launch_bounds__(256) void softmax_xent_rows_kernel(const float *__restrict__ z, const int64_t *__restrict__ label, int B, int C,
                                                                float *__restrict__ loss, float *__restrict__ dz)
{
    __shared__ float part[256];
    float acc = 0.f;
    const float invB = 1.0f / (float)B;
    for (int b = blockIdx.x * 256 + threadIdx.x; b < B; b += gridDim.x * 256) {
        const float *row = z + (int64_t)b * C;
        float m = row[0];
        for (int c = 1; c < C; ++c) m = fmaxf(m, row[c]);
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += expf(row[c] - m);
        const float lse = m + logf(s);
        const int64_t y = label[b];
        if (y >= 0 && y < C) acc += lse - row[y];
        const float inv = invB / s;
        for (int c = 0; c < C; ++c) {
            float g = expf(row[c] - m) * inv;
            if (c == y) g -= invB;
            dz[(int64_t)b * C + c] = g;
        }
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(loss, part[0] * invB);
}

// ---- chains: the layers of a head as phases of ONE launch (grid barrier in between, see the top of the file) --------------------------------
constexpr int HC_MAX = 4;
struct HeadChainFwd {
    HeadFwd l[HC_MAX];
    int n;
    const int64_t *labels; float *loss, *dlogits;     // optional: mean softmax cross-entropy of the last layer's output, fused
    unsigned *bar;
};
__global__ __launch_bounds__(HT) void head_chain_fwd_kernel(HeadChainFwd c)
{
    const int bid = (int)blockIdx.x;
    for (int p = 0; p < c.n; ++p) {
        const HeadFwd &a = c.l[p];
        if (bid * 32 < a.Cout) {
            if (p == 0) head_fwd_body<false, true>(a, bid);
            else head_fwd_body<true, true>(a, bid);
        }
        if (p < c.n - 1) grid_barrier(c.bar, (unsigned)(p + 1) * gridDim.x);
    }
    if (c.labels) {
        extern __shared__ float smem[];
        const HeadFwd &a = c.l[c.n - 1];
        if (a.Cout <= 32) {
            // the one workgroup that made the logits still holds them (bias added, no norm on a last layer) in its LDS tile
            if (bid == 0) {
                __syncthreads();
                softmax_xent_body<false>(smem + HW * 32 * TP, TP, c.labels, a.B, a.Cout, c.loss, c.dlogits, smem);
            }
        } else {
            grid_barrier(c.bar, (unsigned)c.n * gridDim.x);
            if (bid == 0) softmax_xent_body<true>(a.out, a.Cout, c.labels, a.B, a.Cout, c.loss, c.dlogits, smem);
        }
    }
    if (c.n > 1 || (c.labels && c.l[c.n - 1].Cout > 32)) grid_exit(c.bar);     // (no barrier was taken: nothing to return to zero, no atomic round trip at the end)
}

struct HeadChainBwd {
    HeadBwd j[HC_MAX];
    int gx[HC_MAX], gy[HC_MAX], off[HC_MAX], phase[HC_MAX];
    int n, n_phases;
    unsigned *bar;
};
__global__ __launch_bounds__(HT) void head_chain_bwd_kernel(HeadChainBwd c)
{
    const int bid = (int)blockIdx.x;
    for (int ph = 0; ph < c.n_phases; ++ph) {
        for (int j = 0; j < c.n; ++j) {
            if (c.phase[j] != ph) continue;
            const int local = bid - c.off[j];
            if (local >= 0 && local < c.gx[j] * c.gy[j]) {
                if (ph == 0) head_bwd_body<false, true>(c.j[j], local % c.gx[j], local / c.gx[j], c.gy[j]);
                else head_bwd_body<true, true>(c.j[j], local % c.gx[j], local / c.gx[j], c.gy[j]);
            }
        }
        if (ph < c.n_phases - 1) grid_barrier(c.bar, (unsigned)(ph + 1) * gridDim.x);
    }
    if (c.n_phases > 1) grid_exit(c.bar);
}

}  // namespace papc

using namespace papc;

extern "C" {

int papc_head_fc_f32(const float *x, const float *w, const float *bias, const float *gamma, const float *beta, int B, int Cin, int Cout,
                     int has_bn, float eps, float momentum, float *running_mean, float *running_var, int64_t *num_batches_tracked,
                     float drop_p, const int64_t *rng_state, int layer_tag, int64_t *rng_bump, float *y, float *mean, float *invstd,
                     uint8_t *keep, float *out, papc_stream_t stream)
{
    PAPC_REQUIRE(x && w && out, PAPC_E_INVALID, "papc_head_fc_f32: null pointer");
    PAPC_REQUIRE(B >= 1 && B <= HROWS_MAX, PAPC_E_INVALID, "papc_head_fc_f32: B=%d not in [1, %d]", B, HROWS_MAX);
    PAPC_REQUIRE(Cin >= 4 && Cin % 4 == 0 && Cout >= 1, PAPC_E_INVALID, "papc_head_fc_f32: Cin=%d must be a multiple of 4", Cin);
    PAPC_REQUIRE(has_bn >= 0 && has_bn <= 3, PAPC_E_INVALID, "papc_head_fc_f32: has_bn=%d not in {0, 1, 2, 3}", has_bn);
    PAPC_REQUIRE(has_bn != 1 || (gamma && beta && mean && invstd && y), PAPC_E_INVALID, "papc_head_fc_f32: BatchNorm needs gamma/beta/mean/invstd/y");
    PAPC_REQUIRE(has_bn != 3 || (gamma && beta && running_mean && running_var), PAPC_E_INVALID, "papc_head_fc_f32: eval-mode BatchNorm needs gamma/beta/running_mean/running_var");
    PAPC_REQUIRE(drop_p >= 0.f && drop_p < 1.f, PAPC_E_INVALID, "papc_head_fc_f32: drop_p=%f", (double)drop_p);
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_MISC, st);
    HeadFwd a{x, w, bias, gamma, beta, B, Cin, Cout, has_bn, eps, momentum, running_mean, running_var, num_batches_tracked,
              drop_p, rng_state, layer_tag, rng_bump, y, mean, invstd, out, keep};
    const int Bpad = (B + 31) & ~31;
    const size_t lds = (size_t)(HW * 32 + Bpad) * TP * sizeof(float);
    if (lds > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void *>(head_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return check_launch("papc_head_fc_f32: hipFuncSetAttribute");
    hipLaunchKernelGGL(head_fwd_kernel, dim3((unsigned)cdiv(Cout, 32)), dim3(HT), lds, st, a);
    return check_launch("papc_head_fc_f32");
}

int papc_head_bwd_f32(const float *gnext, const float *wnext, int Cn, const float *out, const float *y, const float *mean,
                      const float *invstd, const float *gamma, float drop_p, int has_bn, const float *x, int B, int Cin, int Cout,
                      float *dy, float *dw, float *db, float *dgamma, float *dbeta, int accumulate, papc_stream_t stream)
{
    PAPC_REQUIRE(gnext, PAPC_E_INVALID, "papc_head_bwd_f32: null pointer");
    PAPC_REQUIRE(B >= 1 && B <= HROWS_MAX, PAPC_E_INVALID, "papc_head_bwd_f32: B=%d not in [1, %d]", B, HROWS_MAX);
    PAPC_REQUIRE(Cout >= 1 && (!wnext || (Cn >= 4 && Cn % 4 == 0)), PAPC_E_INVALID, "papc_head_bwd_f32: Cn=%d must be a multiple of 4", Cn);
    PAPC_REQUIRE(wnext || Cn == Cout, PAPC_E_INVALID, "papc_head_bwd_f32: without wnext, gnext must be [B, Cout]");
    PAPC_REQUIRE(has_bn >= 0 && has_bn <= 3, PAPC_E_INVALID, "papc_head_bwd_f32: has_bn=%d not in {0, 1, 2, 3}", has_bn);
    PAPC_REQUIRE((has_bn != 1 && has_bn != 3) || (out && y && mean && invstd && gamma && dgamma && dbeta), PAPC_E_INVALID, "papc_head_bwd_f32: BatchNorm backward needs the saved forward");
    PAPC_REQUIRE(has_bn != 2 || out, PAPC_E_INVALID, "papc_head_bwd_f32: ReLU backward needs the layer's output");
    PAPC_REQUIRE(!x || (dw && Cin >= 1), PAPC_E_INVALID, "papc_head_bwd_f32: x without dw");
    PAPC_REQUIRE(drop_p >= 0.f && drop_p < 1.f, PAPC_E_INVALID, "papc_head_bwd_f32: drop_p=%f", (double)drop_p);
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_MISC, st);
    HeadBwd a{gnext, wnext, Cn, out, y, mean, invstd, gamma, 1.0f / (1.0f - drop_p), has_bn, x, B, Cin, Cout, dy, dw, db, dgamma, dbeta, accumulate};
    const int Bpad = (B + 31) & ~31;
    const size_t lds = (size_t)(HW * 32 + 2 * Bpad) * TP * sizeof(float);
    if (lds > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void *>(head_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return check_launch("papc_head_bwd_f32: hipFuncSetAttribute");
    const int nsplit = x ? std::min(8, std::max(1, (int)cdiv(cdiv(Cin, 32), HW))) : 1;   // dW column blocks: one per wave where possible
    hipLaunchKernelGGL(head_bwd_kernel, dim3((unsigned)cdiv(Cout, 32), (unsigned)nsplit), dim3(HT), lds, st, a);
    return check_launch("papc_head_bwd_f32");
}

int papc_softmax_xent_f32(const float *logits, const int64_t *labels, int B, int C, float *loss, float *dlogits, papc_stream_t stream)
{
    PAPC_REQUIRE(logits && labels && loss && dlogits, PAPC_E_INVALID, "papc_softmax_xent_f32: null pointer");
    PAPC_REQUIRE(B >= 1 && C >= 1, PAPC_E_INVALID, "papc_softmax_xent_f32: B=%d C=%d", B, C);
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_MISC, st);
    if (B <= 1024) {   // a classifier batch: one workgroup, fixed summation order (bit-reproducible loss)
        hipLaunchKernelGGL(softmax_xent_kernel, dim3(1), dim3(256), 0, st, logits, labels, B, C, loss, dlogits);
    } else {
        if (hipMemsetAsync(loss, 0, sizeof(float), st) != hipSuccess) return check_launch("papc_softmax_xent_f32 (memset)");
        hipLaunchKernelGGL(softmax_xent_rows_kernel, dim3((unsigned)std::min(cdiv(B, 256), (int64_t)2048)), dim3(256), 0, st, logits, labels, B, C,
                           loss, dlogits);
    }
    return check_launch("papc_softmax_xent_f32");
}

int papc_head_chain_fwd_f32(const papc_head_fc_layer *layers, int n_layers, int B, const int64_t *rng_state, int64_t *rng_bump,
                            const int64_t *labels, float *loss, float *dlogits, uint32_t *sync, papc_stream_t stream)
{
    PAPC_REQUIRE(layers && sync, PAPC_E_INVALID, "papc_head_chain_fwd_f32: null pointer");
    PAPC_REQUIRE(n_layers >= 1 && n_layers <= HC_MAX, PAPC_E_UNSUPPORTED, "papc_head_chain_fwd_f32: %d layers (1..%d)", n_layers, HC_MAX);
    PAPC_REQUIRE(B >= 1 && B <= HROWS_MAX, PAPC_E_INVALID, "papc_head_chain_fwd_f32: B=%d not in [1, %d]", B, HROWS_MAX);
    PAPC_REQUIRE(!labels || (loss && dlogits), PAPC_E_INVALID, "papc_head_chain_fwd_f32: labels without loss / dlogits");
    HeadChainFwd c;
    memset(&c, 0, sizeof(c));
    int grid = 1;
    for (int i = 0; i < n_layers; ++i) {
        const papc_head_fc_layer &l = layers[i];
        PAPC_REQUIRE(l.x && l.w && l.out, PAPC_E_INVALID, "papc_head_chain_fwd_f32: layer %d: null pointer", i);
        PAPC_REQUIRE(l.Cin >= 4 && l.Cin % 4 == 0 && l.Cout >= 1, PAPC_E_INVALID, "papc_head_chain_fwd_f32: layer %d: Cin=%d must be a multiple of 4", i, l.Cin);
        PAPC_REQUIRE(l.has_bn >= 0 && l.has_bn <= 3, PAPC_E_INVALID, "papc_head_chain_fwd_f32: layer %d: has_bn=%d", i, l.has_bn);
        PAPC_REQUIRE(l.has_bn != 1 || (l.gamma && l.beta && l.mean && l.invstd && l.y), PAPC_E_INVALID, "papc_head_chain_fwd_f32: layer %d: BatchNorm needs gamma/beta/mean/invstd/y", i);
        PAPC_REQUIRE(l.has_bn != 3 || (l.gamma && l.beta && l.running_mean && l.running_var), PAPC_E_INVALID, "papc_head_chain_fwd_f32: layer %d: eval-mode BatchNorm needs gamma/beta/running statistics", i);
        PAPC_REQUIRE(l.drop_p >= 0.f && l.drop_p < 1.f, PAPC_E_INVALID, "papc_head_chain_fwd_f32: layer %d: drop_p=%f", i, (double)l.drop_p);
        PAPC_REQUIRE(i == 0 || (l.x == layers[i - 1].out && l.Cin == layers[i - 1].Cout), PAPC_E_INVALID, "papc_head_chain_fwd_f32: layer %d does not consume layer %d's output", i, i - 1);
        c.l[i] = HeadFwd{l.x, l.w, l.bias, l.gamma, l.beta, B, l.Cin, l.Cout, l.has_bn, l.eps, l.momentum, l.running_mean, l.running_var, l.num_batches_tracked,
                         l.drop_p, rng_state, l.layer_tag, (i == n_layers - 1) ? rng_bump : nullptr, l.y, l.mean, l.invstd, l.out, l.keep};
        grid = std::max(grid, (int)cdiv(l.Cout, 32));
    }
    PAPC_REQUIRE(!labels || layers[n_layers - 1].has_bn == 0, PAPC_E_INVALID, "papc_head_chain_fwd_f32: the fused loss wants a plain last layer (has_bn = 0)");
    PAPC_REQUIRE(grid <= 64, PAPC_E_UNSUPPORTED, "papc_head_chain_fwd_f32: %d workgroups (layers wider than 2048 channels take papc_head_fc_f32 per layer)", grid);
    c.n = n_layers; c.labels = labels; c.loss = loss; c.dlogits = dlogits; c.bar = sync;
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_MISC, st);
    const int Bpad = (B + 31) & ~31;
    int stage = 0;
    for (int i = 1; i < n_layers; ++i)
        if ((int64_t)B * layers[i].Cin <= HC_STAGE_FLOATS) stage = std::max(stage, B * layers[i].Cin);
    const size_t lds = ((size_t)(HW * 32 + Bpad) * TP + stage) * sizeof(float);
    if (lds > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void *>(head_chain_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return check_launch("papc_head_chain_fwd_f32: hipFuncSetAttribute");
    hipLaunchKernelGGL(head_chain_fwd_kernel, dim3((unsigned)grid), dim3(HT), lds, st, c);
    return check_launch("papc_head_chain_fwd_f32");
}

int papc_head_chain_bwd_f32(const papc_head_bwd_job *jobs, int n_jobs, int B, uint32_t *sync, papc_stream_t stream)
{
    PAPC_REQUIRE(jobs && sync, PAPC_E_INVALID, "papc_head_chain_bwd_f32: null pointer");
    PAPC_REQUIRE(n_jobs >= 1 && n_jobs <= HC_MAX, PAPC_E_UNSUPPORTED, "papc_head_chain_bwd_f32: %d jobs (1..%d)", n_jobs, HC_MAX);
    PAPC_REQUIRE(B >= 1 && B <= HROWS_MAX, PAPC_E_INVALID, "papc_head_chain_bwd_f32: B=%d not in [1, %d]", B, HROWS_MAX);
    HeadChainBwd c;
    memset(&c, 0, sizeof(c));
    int grid = 1, n_phases = 0;
    int used[HC_MAX] = {0, 0, 0, 0};          // workgroups handed out per phase
    for (int i = 0; i < n_jobs; ++i) {
        const papc_head_bwd_job &j = jobs[i];
        PAPC_REQUIRE(j.gnext, PAPC_E_INVALID, "papc_head_chain_bwd_f32: job %d: null gnext", i);
        PAPC_REQUIRE(j.Cout >= 1 && (!j.wnext || (j.Cn >= 4 && j.Cn % 4 == 0)), PAPC_E_INVALID, "papc_head_chain_bwd_f32: job %d: Cn=%d must be a multiple of 4", i, j.Cn);
        PAPC_REQUIRE(j.wnext || j.Cn == j.Cout, PAPC_E_INVALID, "papc_head_chain_bwd_f32: job %d: without wnext, gnext must be [B, Cout]", i);
        PAPC_REQUIRE(j.has_bn >= 0 && j.has_bn <= 3, PAPC_E_INVALID, "papc_head_chain_bwd_f32: job %d: has_bn=%d", i, j.has_bn);
        PAPC_REQUIRE((j.has_bn != 1 && j.has_bn != 3) || (j.out && j.y && j.mean && j.invstd && j.gamma && j.dgamma && j.dbeta), PAPC_E_INVALID, "papc_head_chain_bwd_f32: job %d: BatchNorm backward needs the saved forward", i);
        PAPC_REQUIRE(j.has_bn != 2 || j.out, PAPC_E_INVALID, "papc_head_chain_bwd_f32: job %d: ReLU backward needs the layer's output", i);
        PAPC_REQUIRE(!j.x || (j.dw && j.Cin >= 1), PAPC_E_INVALID, "papc_head_chain_bwd_f32: job %d: x without dw", i);
        PAPC_REQUIRE(j.drop_p >= 0.f && j.drop_p < 1.f, PAPC_E_INVALID, "papc_head_chain_bwd_f32: job %d: drop_p=%f", i, (double)j.drop_p);
        PAPC_REQUIRE(j.phase >= 0 && j.phase < HC_MAX && (i == 0 || j.phase >= jobs[i - 1].phase), PAPC_E_INVALID, "papc_head_chain_bwd_f32: job %d: phases must be 0..%d, non-decreasing", i, HC_MAX - 1);
        c.j[i] = HeadBwd{j.gnext, j.wnext, j.Cn, j.out, j.y, j.mean, j.invstd, j.gamma, 1.0f / (1.0f - j.drop_p), j.has_bn, j.x, B, j.Cin, j.Cout, j.dy, j.dw, j.db,
                         j.dgamma, j.dbeta, j.accumulate};
        c.gx[i] = (int)cdiv(j.Cout, 32);
        c.gy[i] = j.x ? std::min(8, std::max(1, (int)cdiv(cdiv(j.Cin, 32), HW))) : 1;      // (papc_head_bwd_f32's own split)
        c.phase[i] = j.phase;
        c.off[i] = used[j.phase];
        used[j.phase] += c.gx[i] * c.gy[i];
        grid = std::max(grid, used[j.phase]);
        n_phases = std::max(n_phases, j.phase + 1);
    }
    PAPC_REQUIRE(grid <= 128, PAPC_E_UNSUPPORTED, "papc_head_chain_bwd_f32: %d workgroups in one phase (wider layers take papc_head_bwd_f32 per layer)", grid);
    c.n = n_jobs; c.n_phases = n_phases; c.bar = sync;
    hipStream_t st = as_stream(stream);
    ProfScope prof(PAPC_K_MISC, st);
    const int Bpad = (B + 31) & ~31;
    int stage = 0;
    for (int i = 0; i < n_jobs; ++i)
        if (jobs[i].phase > 0 && jobs[i].wnext && (int64_t)B * jobs[i].Cn <= HC_STAGE_FLOATS) stage = std::max(stage, B * jobs[i].Cn);
    const size_t lds = ((size_t)(HW * 32 + 2 * Bpad) * TP + stage) * sizeof(float);
    if (lds > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void *>(head_chain_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return check_launch("papc_head_chain_bwd_f32: hipFuncSetAttribute");
    hipLaunchKernelGGL(head_chain_bwd_kernel, dim3((unsigned)grid), dim3(HT), lds, st, c);
    return check_launch("papc_head_chain_bwd_f32");
}

}  // extern "C"
