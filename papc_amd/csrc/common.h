// common.h -- shared host/device helpers for libpapc_hip (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>

#include "papc_hip.h"

namespace papc {

// ---- error plumbing (thread-local message, negative status codes) ---------------------------------
void set_error(const char *fmt, ...);
int check_launch(const char *what);

#define PAPC_REQUIRE(cond, code, ...)      \
    do {                                   \
        if (!(cond)) {                     \
            papc::set_error(__VA_ARGS__);  \
            return (code);                 \
        }                                  \
    } while (0)

// ---- tuning knobs: the PAPC_* environment variables are read ONCE, when the library is loaded (capi.hip); entry points only
// index this table.  papc_knob_set() (C ABI) lets a tuning harness flip one at run time.
enum Knob {
    KNOB_LG_PARTS, KNOB_DW_F32, KNOB_DW_DBG, KNOB_DW_XYZ, KNOB_PARTS, KNOB_GEMM_F32, KNOB_GEMM_WAVES, KNOB_MAXCAT_WAVES,
    KNOB_GEMM_WS, KNOB_GEMM_OCC, KNOB_GEMM_DBG, KNOB_GEMM_KB, KNOB_GEMM_MINWG, KNOB_GEMM_TL, KNOB_FPS_THREADS,
    KNOB_FPS_THREADS_SMALL, KNOB_STREAM, KNOB_STREAM_MINTILES, KNOB_STREAM_CK, KNOB_STREAM_ASM, KNOB_PFN_MFMA, KNOB_DW_RS64, KNOB_DW_ROWS, KNOB_DW_ROWS_BLOCKS, KNOB_DW_ROWSX, KNOB_PG_DBG, KNOB_PG_NB, KNOB_PG_NS, KNOB_STREAM_MAXCAT, KNOB_MAX_NOSTORE, KNOB_PFN_FUSED_TAILS, KNOB_STREAM_NW12, KNOB_LG_LISTS, KNOB_FOLD_WAVES, KNOB_LGL_VARIANT, KNOB_LG_PP, KNOB_COUNT
};
int knob(int id);

// ---- optional event profiler ----------------------------------------------------------------------
struct ProfScope {
    int kernel;
    hipStream_t stream;
    bool on;
    hipEvent_t e0;
    ProfScope(int kernel, hipStream_t stream);
    ~ProfScope();
};

static inline hipStream_t as_stream(papc_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// ---- device: wave64 DPP reductions ------------------------------------------------------------------
// DPP controls (gfx9 family): quad_perm [1,0,3,2]=0xB1, [2,3,0,1]=0x4E, row_ror:n = 0x120+n,
// row_bcast:15 = 0x142, row_bcast:31 = 0x143.  After the sequence lane 63 holds the wave result.
#define PAPC_DPP(v, ctrl, rmask) __builtin_amdgcn_update_dpp((int)(v), (int)(v), (ctrl), (rmask), 0xF, false)

// sum over the 64 lanes of a float; every lane of the last row (48..63) holds the total, lane 63 canonical
__device__ __forceinline__ float wave_sum_f32_to_lane63(float v)
{
    float t;
    t = __int_as_float(PAPC_DPP(__float_as_int(v), 0xB1, 0xF));  v += t;
    t = __int_as_float(PAPC_DPP(__float_as_int(v), 0x4E, 0xF));  v += t;
    t = __int_as_float(PAPC_DPP(__float_as_int(v), 0x124, 0xF)); v += t;
    t = __int_as_float(PAPC_DPP(__float_as_int(v), 0x128, 0xF)); v += t;
    // row_bcast adds lane 15 of the previous row into rows 1 and 3, then lane 31 into rows 2,3
    t = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xA, 0xF, false)); v += t;
    t = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xC, 0xF, false)); v += t;
    return v;
}

// sums over the two 32-lane halves of a wave: lanes 16..31 hold the sum of lanes 0..31, lanes 48..63 that of lanes 32..63
__device__ __forceinline__ float half_sum_f32(float v)
{
    float t;
    t = __int_as_float(PAPC_DPP(__float_as_int(v), 0xB1, 0xF));  v += t;
    t = __int_as_float(PAPC_DPP(__float_as_int(v), 0x4E, 0xF));  v += t;
    t = __int_as_float(PAPC_DPP(__float_as_int(v), 0x124, 0xF)); v += t;
    t = __int_as_float(PAPC_DPP(__float_as_int(v), 0x128, 0xF)); v += t;
    t = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xA, 0xF, false)); v += t;
    return v;
}

__device__ __forceinline__ float readlane63_f32(float v)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}


// ---- 32-bit wave / row reductions (full-rate VALU; v_max_f32 / v_min_u32 fuse with the DPP operand)
#define PAPC_DPPF(v, ctrl, rmask) __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), (ctrl), (rmask), 0xF, false))
#define PAPC_DPPU(v, ctrl, rmask) ((unsigned)__builtin_amdgcn_update_dpp((int)(v), (int)(v), (ctrl), (rmask), 0xF, false))

__device__ __forceinline__ float row_max_f32(float v)  // every lane of a 16-lane row ends with the row max
{
    v = fmaxf(v, PAPC_DPPF(v, 0xB1, 0xF));
    v = fmaxf(v, PAPC_DPPF(v, 0x4E, 0xF));
    v = fmaxf(v, PAPC_DPPF(v, 0x124, 0xF));
    v = fmaxf(v, PAPC_DPPF(v, 0x128, 0xF));
    return v;
}
__device__ __forceinline__ unsigned row_min_u32(unsigned v)
{
    unsigned t;
    t = PAPC_DPPU(v, 0xB1, 0xF);  v = t < v ? t : v;
    t = PAPC_DPPU(v, 0x4E, 0xF);  v = t < v ? t : v;
    t = PAPC_DPPU(v, 0x124, 0xF); v = t < v ? t : v;
    t = PAPC_DPPU(v, 0x128, 0xF); v = t < v ? t : v;
    return v;
}
__device__ __forceinline__ float wave_max_f32_to_lane63(float v)
{
    v = row_max_f32(v);
    v = fmaxf(v, PAPC_DPPF(v, 0x142, 0xA));
    v = fmaxf(v, PAPC_DPPF(v, 0x143, 0xC));
    return v;
}
__device__ __forceinline__ unsigned wave_min_u32_to_lane63(unsigned v)
{
    unsigned t;
    v = row_min_u32(v);
    t = PAPC_DPPU(v, 0x142, 0xA); v = t < v ? t : v;
    t = PAPC_DPPU(v, 0x143, 0xC); v = t < v ? t : v;
    return v;
}
// ---- the same reductions as single fused DPP instructions (inline asm).  hipcc compiles the builtin forms above to
// v_mov_b32 + s_nop + v_mov_b32_dpp + (canonicalising v_max_f32) + v_max per step, ~5 dependent issue slots; here a step
// is `s_nop 1` (the VALU-write -> DPP-read hazard) + one v_max_u32_dpp / v_min_u32_dpp.  Unsigned compares: callers pass
// the bit patterns of NON-NEGATIVE floats, which order like the floats and need no canonicalisation.
#define PAPC_DPP_STEP(op, v, ctrl) asm volatile("s_nop 1\n\t" op " %0, %0, %0 " ctrl : "+v"(v))
__device__ __forceinline__ unsigned row_max_u32_fused(unsigned v)   // every lane of a 16-lane row ends with the row max
{
    PAPC_DPP_STEP("v_max_u32_dpp", v, "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf");
    PAPC_DPP_STEP("v_max_u32_dpp", v, "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf");
    PAPC_DPP_STEP("v_max_u32_dpp", v, "row_ror:4 row_mask:0xf bank_mask:0xf");
    PAPC_DPP_STEP("v_max_u32_dpp", v, "row_ror:8 row_mask:0xf bank_mask:0xf");
    return v;
}
__device__ __forceinline__ unsigned row_min_u32_fused(unsigned v)
{
    PAPC_DPP_STEP("v_min_u32_dpp", v, "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf");
    PAPC_DPP_STEP("v_min_u32_dpp", v, "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf");
    PAPC_DPP_STEP("v_min_u32_dpp", v, "row_ror:4 row_mask:0xf bank_mask:0xf");
    PAPC_DPP_STEP("v_min_u32_dpp", v, "row_ror:8 row_mask:0xf bank_mask:0xf");
    return v;
}
__device__ __forceinline__ unsigned wave_max_u32_fused_to_lane63(unsigned v)   // lane 63 (all of row 3) holds the wave max
{
    v = row_max_u32_fused(v);
    PAPC_DPP_STEP("v_max_u32_dpp", v, "row_bcast:15 row_mask:0xa bank_mask:0xf");
    PAPC_DPP_STEP("v_max_u32_dpp", v, "row_bcast:31 row_mask:0xc bank_mask:0xf");
    asm volatile("s_nop 1" ::"v"(v));
    return v;
}
__device__ __forceinline__ unsigned wave_min_u32_fused_to_lane63(unsigned v)
{
    v = row_min_u32_fused(v);
    PAPC_DPP_STEP("v_min_u32_dpp", v, "row_bcast:15 row_mask:0xa bank_mask:0xf");
    PAPC_DPP_STEP("v_min_u32_dpp", v, "row_bcast:31 row_mask:0xc bank_mask:0xf");
    asm volatile("s_nop 1" ::"v"(v));
    return v;
}
__device__ __forceinline__ unsigned readlane63_u32(unsigned v) { return (unsigned)__builtin_amdgcn_readlane((int)v, 63); }
__device__ __forceinline__ float readlane0_f32(float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)); }
__device__ __forceinline__ unsigned readlane0_u32(unsigned v) { return (unsigned)__builtin_amdgcn_readlane((int)v, 0); }

// Workgroup barrier that orders LDS traffic only: waits for this wave's outstanding LDS ops (lgkmcnt) and joins the
// barrier, but -- unlike __syncthreads(), whose fence makes hipcc drain vmcnt(0) -- leaves global loads AND stores in
// flight.  On CDNA vmcnt counts stores too, so a plain __syncthreads() after an epilogue exposes the full store latency.
// Only valid where no global-memory hand-off between waves depends on the barrier.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ReLU that lets a NaN through (v_max_f32 / fmaxf return the non-NaN operand): used where a pooled layer writes its OUTPUT, so that a
// diverged stack (NaN weights -> NaN statistics -> NaN scale / shift) shows up as NaN downstream in the padded and the compacted form alike,
// the way the reference's relu(max(.)) does (pointnet2_basic_layers.py:217-219)
__device__ __forceinline__ float relu_np(float t) { return !(t <= 0.f) ? t : 0.f; }
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

}  // namespace papc
