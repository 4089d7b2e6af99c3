// capi.hip -- version, thread-local error string, event profiler of libpapc_hip.
#include <stdarg.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "common.h"

#define GEMM_PARTS_DEFAULT 768

namespace papc {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return PAPC_E_LAUNCH;
    }
    return PAPC_OK;
}

// ---- knobs --------------------------------------------------------------------------------------
struct KnobDef { const char *name; int def, lo, hi; };
static const KnobDef g_knob_defs[KNOB_COUNT] = {
    {"PAPC_LG_PARTS", 2048, 1, 8192},      // workgroups of the gather-add kernels
    {"PAPC_DW_F32", 0, 0, 1},              // dW on the exact-f32 MFMA
    {"PAPC_DW_DBG", 0, 0, 1},
    {"PAPC_DW_XYZ", 1, 0, 1},              // streamed dW of a coordinates-only first layer
    {"PAPC_PARTS", GEMM_PARTS_DEFAULT, 1, 1024},   // rows of the per-workgroup partial buffers
    {"PAPC_GEMM_F32", 0, 0, 1},            // forward / dX GEMMs on the exact-f32 MFMA
    {"PAPC_GEMM_WAVES", 0, 0, 8},
    {"PAPC_MAXCAT_WAVES", 8, 4, 8},
    {"PAPC_GEMM_WS", 0, 0, 3},
    {"PAPC_GEMM_OCC", 0, 0, 1},
    {"PAPC_GEMM_DBG", 0, 0, 1},
    {"PAPC_GEMM_KB", 0, 0, 2},
    {"PAPC_GEMM_MINWG", 192, 1, 4096},
    {"PAPC_GEMM_TL", 0, 0, 1},
    {"PAPC_FPS_THREADS", 0, 0, 1024},
    {"PAPC_FPS_THREADS_SMALL", 0, 0, 1024},
    {"PAPC_STREAM", 1, 0, 1},              // row-streaming GEMM (mlp_stream.hip) where a flavour fits
    {"PAPC_STREAM_MINTILES", 2048, 1, 1 << 30},   // ... for problems with at least this many 32-row tiles
    {"PAPC_STREAM_CK", 0, 0, 8},           // k blocks per prefetch chunk (0 = per shape)
    {"PAPC_STREAM_ASM", 1, 0, 1},          // operand loads hidden from hipcc's waitcnt pass (hand-counted vmcnt)
    {"PAPC_PFN_MFMA", 1, 0, 1},            // PillarFeatureNet apply pass on the bf16 matrix pipe (0: lanes-are-channels VALU flavour)
    {"PAPC_DW_RS64", 1, 0, 1},             // dW of 64 x 64 layers: 64-row stages (all producer threads busy)
    {"PAPC_DW_ROWS", 1, 0, 1},             // dW of layers with a 64-channel BN+ReLU input on the row-streaming kernel (dw_rows_kernel)
    {"PAPC_DW_ROWS_BLOCKS", 4, 1, 16},     // ... for layers of at most this many 64 x 64 output blocks (8: slower, each block transforms its operands again)
    {"PAPC_DW_ROWSX", 1, 0, 1},            // dW of 128 -> 256k layers with dY streamed per wave and x staged once per workgroup (dw_rowsx_kernel; 0: the staged kernel)
    {"PAPC_PG_DBG", 0, 0, 15},             // development aid for pg_gemm_kernel (timing only, results are garbage): 1 no MFMAs, 2 no fragment loads, 4 no LDS reads, 8 no epilogue
    {"PAPC_PG_NB", 0, 0, 2},               // pg_gemm_kernel column tile: 1 = 64, 2 = 128 columns (0 = per shape)
    {"PAPC_PG_NS", 0, 0, 3},               // ... stages of its LDS ring: 2 or 3 (0 = per tile flavour)
    {"PAPC_STREAM_MAXCAT", 1, 0, 1},       // papc_mlp_bwd_dx_max_f32 on the row-streaming kernel where it has the flavour (0: tiled kernel)
    {"PAPC_MAX_NOSTORE", 1, 0, 1},         // the max-pooled last layer without its stored output where all three kernels have the flavour (papc_mlp_max_nostore_ok)
    {"PAPC_PFN_FUSED_TAILS", 1, 0, 1},     // papc_pfn_fwd / _bwd: BatchNorm constants / dW finalize as the last-arriving workgroup's tail of the Gram pass / the fold (pfn.hip)
    {"PAPC_STREAM_NW12", 1, 0, 1},         // twelve waves per workgroup (three per SIMD) for the row-streaming dX flavours that fit 168 registers (0: eight)
    {"PAPC_LG_LISTS", 1, 0, 1},            // gather-add backward over the grouping's point lists where the caller supplies them (0: float atomics)
    {"PAPC_FOLD_WAVES", 16, 8, 16},        // papc_fold_jobs_f32: chunk lanes (waves) per workgroup, 8 or 16
    {"PAPC_LGL_VARIANT", 0, 0, 3},         // lingather_bwd_lists_kernel geometry (A/B): 0 = 4 entries per half-wave and pass; 1 = 2 entries; 3 = 8 entries
    {"PAPC_LG_PP", 1, 0, 1},               // gather-add backward over point lists WITHOUT re-reading y: dz stored masked by the dX above, y's share from P and the lists' moments (0: gathers y and dz)
};
static int g_knobs[KNOB_COUNT];
static int knob_parse(int id, const char *e)
{
    const KnobDef &d = g_knob_defs[id];
    if (!e || !*e) return d.def;
    const int v = atoi(e);
    return (v < d.lo || v > d.hi) ? d.def : v;
}
static const bool g_knobs_loaded = [] {
    for (int i = 0; i < KNOB_COUNT; ++i) g_knobs[i] = knob_parse(i, getenv(g_knob_defs[i].name));
    return true;
}();
int knob(int id) { return g_knobs[id]; }

// ---- profiler: event pairs per enabled kernel family, resolved lazily in papc_prof_read ---------
struct ProfState {
    std::mutex mu;
    unsigned mask = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending[PAPC_K_COUNT];
    std::vector<hipEvent_t> pool;
    double total_ms[PAPC_K_COUNT] = {0};
    int64_t launches[PAPC_K_COUNT] = {0};
};
static ProfState g_prof;

static hipEvent_t prof_get_event()
{
    if (!g_prof.pool.empty()) {
        hipEvent_t e = g_prof.pool.back();
        g_prof.pool.pop_back();
        return e;
    }
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

ProfScope::ProfScope(int k, hipStream_t s) : kernel(k), stream(s), on(false), e0(nullptr)
{
    if (!(g_prof.mask & (1u << k))) return;
    std::lock_guard<std::mutex> lk(g_prof.mu);
    e0 = prof_get_event();
    if (!e0) return;
    on = hipEventRecord(e0, stream) == hipSuccess;
}

ProfScope::~ProfScope()
{
    if (!on) return;
    std::lock_guard<std::mutex> lk(g_prof.mu);
    hipEvent_t e1 = prof_get_event();
    if (!e1) return;
    if (hipEventRecord(e1, stream) == hipSuccess) g_prof.pending[kernel].push_back({e0, e1});
}

static void prof_drain(int k)
{
    for (auto &pr : g_prof.pending[k]) {
        float ms = 0.f;
        if (hipEventSynchronize(pr.second) == hipSuccess && hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
            g_prof.total_ms[k] += ms;
            g_prof.launches[k] += 1;
        }
        g_prof.pool.push_back(pr.first);
        g_prof.pool.push_back(pr.second);
    }
    g_prof.pending[k].clear();
}

}  // namespace papc

extern "C" {

int papc_version(void) { return 600; /* 0.6.0 */ }

int papc_abi_version(void) { return PAPC_ABI_VERSION; }

int64_t papc_abi_sizeof(const char *struct_name)
{
    if (!struct_name) return -1;
#define PAPC_SIZEOF(T) if (strcmp(struct_name, #T) == 0) return (int64_t)sizeof(T);
    PAPC_SIZEOF(papc_group_src)
    PAPC_SIZEOF(papc_point_lists)
    PAPC_SIZEOF(papc_group_max)
    PAPC_SIZEOF(papc_bwd_dy)
    PAPC_SIZEOF(papc_scatter_dst)
    PAPC_SIZEOF(papc_bwd_red)
    PAPC_SIZEOF(papc_reduce_job)
    PAPC_SIZEOF(papc_fold_job)
    PAPC_SIZEOF(papc_fold_list)
    PAPC_SIZEOF(papc_sa_desc)
    PAPC_SIZEOF(papc_sa_layer)
    PAPC_SIZEOF(papc_compact_src)
    PAPC_SIZEOF(papc_sa_io)
    PAPC_SIZEOF(papc_sa_plan)
    PAPC_SIZEOF(papc_sa_grads)
    PAPC_SIZEOF(papc_pfn_desc)
    PAPC_SIZEOF(papc_pfn_io)
    PAPC_SIZEOF(papc_head_fc_layer)
    PAPC_SIZEOF(papc_head_bwd_job)
    PAPC_SIZEOF(papc_pg_wjob)
    PAPC_SIZEOF(papc_pg_prep)
    PAPC_SIZEOF(papc_pg_gemm)
    PAPC_SIZEOF(papc_pg_fold_job)
    PAPC_SIZEOF(papc_copy_job)
#undef PAPC_SIZEOF
    return -1;
}

const char *papc_last_error_string(void) { return papc::g_err; }

int papc_knob_set(const char *name, int value)
{
    PAPC_REQUIRE(name, PAPC_E_INVALID, "papc_knob_set: null name");
    for (int i = 0; i < papc::KNOB_COUNT; ++i)
        if (!strcmp(name, papc::g_knob_defs[i].name)) {
            PAPC_REQUIRE(value >= papc::g_knob_defs[i].lo && value <= papc::g_knob_defs[i].hi, PAPC_E_INVALID,
                         "papc_knob_set: %s = %d outside [%d, %d]", name, value, papc::g_knob_defs[i].lo, papc::g_knob_defs[i].hi);
            papc::g_knobs[i] = value;
            return PAPC_OK;
        }
    papc::set_error("papc_knob_set: unknown knob %s", name);
    return PAPC_E_INVALID;
}

int papc_knob_get(const char *name, int *value)
{
    PAPC_REQUIRE(name && value, PAPC_E_INVALID, "papc_knob_get: null argument");
    for (int i = 0; i < papc::KNOB_COUNT; ++i)
        if (!strcmp(name, papc::g_knob_defs[i].name)) { *value = papc::g_knobs[i]; return PAPC_OK; }
    papc::set_error("papc_knob_get: unknown knob %s", name);
    return PAPC_E_INVALID;
}

int papc_prof_enable(unsigned mask)
{
    std::lock_guard<std::mutex> lk(papc::g_prof.mu);
    papc::g_prof.mask = mask;
    return PAPC_OK;
}

int papc_prof_reset(void)
{
    std::lock_guard<std::mutex> lk(papc::g_prof.mu);
    for (int k = 0; k < PAPC_K_COUNT; ++k) {
        papc::prof_drain(k);
        papc::g_prof.total_ms[k] = 0;
        papc::g_prof.launches[k] = 0;
    }
    return PAPC_OK;
}

int papc_prof_read(int kernel, double *total_ms, int64_t *launches)
{
    PAPC_REQUIRE(kernel >= 0 && kernel < PAPC_K_COUNT, PAPC_E_INVALID, "papc_prof_read: bad kernel id %d", kernel);
    std::lock_guard<std::mutex> lk(papc::g_prof.mu);
    papc::prof_drain(kernel);
    if (total_ms) *total_ms = papc::g_prof.total_ms[kernel];
    if (launches) *launches = papc::g_prof.launches[kernel];
    return PAPC_OK;
}

}  // extern "C"
