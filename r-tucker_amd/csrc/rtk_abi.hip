// C-ABI entry points of librtucker_hip.so (declared in include/rtucker_hip.h):
// argument validation, workspace carving, stage orchestration.  No allocation, no
// synchronisation (except rtk_read_error_flag), no global state but the last-error text.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "rtk_common.h"
#include "rtk_pack.h"

static thread_local char g_err[512] = "";

void rtk_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int rtk_query_vectors_f32_impl(const float *core, int a, int b, int c, const float *R, int64_t n_rel,
                               const float *S, int64_t n_sub, const int64_t *rel_idx,
                               const int64_t *sub_idx, int64_t batch, float *v_out, void *q_packed,
                               const RtkWorkspace &ws, hipStream_t st);
int rtk_split_ksteps_supported(int c);
int rtk_query_vectors_bf16_impl(const void *core, int a, int b, int c, const void *R, int64_t n_rel,
                                const void *S, int64_t n_sub, const int64_t *rel_idx,
                                const int64_t *sub_idx, int64_t batch, float *v_out, void *q_packed,
                                const RtkWorkspace &ws, hipStream_t st);

static size_t packed_bytes(int dtype, int64_t batch, int c) {
    const int ks = (c + 15) / 16;
    return (size_t)rtk_cdiv(batch, 32) * (size_t)rtk_pack_tile_bytes(ks, dtype == RTK_F32 ? 2 : 1);
}

static RtkWorkspace carve(void *base, int dtype, int64_t batch, int64_t n_rel, int a, int b, int c) {
    RtkWorkspace w;
    unsigned char *p = (unsigned char *)base;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        unsigned char *q = p ? p + off : nullptr;
        off += rtk_align_up(bytes, 256);
        return q;
    };
    const int64_t n_u_max = n_rel > batch ? batch : n_rel;
    w.flags = (uint32_t *)take(256);
    w.slot_of_rel = (int32_t *)take((size_t)n_rel * 4);
    w.rel_list = (int32_t *)take((size_t)n_u_max * 4);
    w.tables = (float *)take((size_t)n_u_max * b * c * 4);
    w.v = (float *)take((size_t)batch * c * 4);
    w.q_packed = take(packed_bytes(dtype, batch, c));
    const bool big_a = dtype == RTK_BF16 && a > 32 && a <= 512;   // tables through the bf16 MFMA kernel
    w.core_t = big_a ? take((size_t)a * b * c * 2) : nullptr;
    w.r_packed = big_a ? take(packed_bytes(RTK_BF16, n_u_max, a)) : nullptr;
    // fp32, a > 32: tables through the split-fp16 GEMM -- operand bounds (256-byte header) + the gathered relation rows
    if (dtype == RTK_F32 && a > 32) w.r_packed = take(256 + (size_t)n_u_max * a * 4);
    w.grp_cnt = (int32_t *)take((size_t)n_u_max * 8);
    w.grp_order = (int32_t *)take((size_t)batch * 4);
    w.grp_work = (int32_t *)take((size_t)(batch / 4 + n_u_max + 1) * 16);
    w.grp_qinfo = (int64_t *)take((size_t)batch * 16);
    w.total = off;
    return w;
}

// ---- cached relation tables (rtk_relation_tables_* / rtk_query_vectors_from_tables_*) ----------
int rtk_relation_tables_f32_impl(const float *core, int a, int b, int c, const float *R, int64_t n_rel, float *tables,
                                 void *r_scratch, hipStream_t st);
int rtk_relation_tables_bf16_impl(const void *core, int a, int b, int c, const void *R, int64_t n_rel, float *tables,
                                  void *core_t, void *r_packed, hipStream_t st);
int rtk_from_tables_f32_impl(const float *tables, int64_t n_rel, int b, int c, const float *S, int64_t n_sub,
                             const int64_t *rel_idx, const int64_t *sub_idx, int64_t batch, float *v_out,
                             void *q_packed, const RtkWorkspace &ws, hipStream_t st, int rel_part = 0, int rel_parts = 1);
int rtk_from_tables_bf16_impl(const float *tables, int64_t n_rel, int b, int c, const void *S, int64_t n_sub,
                              const int64_t *rel_idx, const int64_t *sub_idx, int64_t batch, float *v_out,
                              void *q_packed, const RtkWorkspace &ws, hipStream_t st, int rel_part = 0, int rel_parts = 1);

// scratch of the tables build: only the bf16 large-relation-rank path needs any (transposed core, packed R rows)
struct TablesWs {
    void *core_t, *r_packed;
    size_t total;
};
static TablesWs carve_tables(void *base, int dtype, int64_t n_rel, int a, int b, int c) {
    TablesWs w;
    unsigned char *p = (unsigned char *)base;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        unsigned char *q = p ? p + off : nullptr;
        off += rtk_align_up(bytes, 256);
        return q;
    };
    (void)take(256);
    const bool big_a = dtype == RTK_BF16 && a > 32 && a <= 512;
    w.core_t = big_a ? take((size_t)a * b * c * 2) : nullptr;
    w.r_packed = big_a ? take(packed_bytes(RTK_BF16, n_rel, a)) : nullptr;
    if (dtype == RTK_F32 && a > 32) w.r_packed = take(256 + (size_t)n_rel * a * 4);   // (as in carve())
    w.total = off;
    return w;
}

// workspace of rtk_query_vectors_from_tables_*: header + the slot order of the queries (slots = relation ids)
static RtkWorkspace carve_ft(void *base, int64_t batch, int64_t n_rel) {
    RtkWorkspace w{};
    unsigned char *p = (unsigned char *)base;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        unsigned char *q = p ? p + off : nullptr;
        off += rtk_align_up(bytes, 256);
        return q;
    };
    w.flags = (uint32_t *)take(256);
    w.grp_cnt = (int32_t *)take((size_t)n_rel * 8);
    w.grp_order = (int32_t *)take((size_t)batch * 4);
    w.grp_work = (int32_t *)take((size_t)(batch / 4 + (n_rel < batch ? n_rel : batch) + 1) * 16);
    w.grp_qinfo = (int64_t *)take((size_t)batch * 16);
    w.total = off;
    return w;
}

extern "C" size_t rtk_relation_tables_bytes(int64_t n_rel, int b, int c) {
    if (n_rel <= 0 || b <= 0 || c <= 0) return 0;
    return rtk_align_up((size_t)n_rel * b * c * sizeof(float), 256);
}

extern "C" size_t rtk_relation_tables_workspace_bytes(int dtype, int64_t n_rel, int a, int b, int c) {
    if (n_rel <= 0 || a <= 0 || b <= 0 || c <= 0) return 0;
    return carve_tables(nullptr, dtype, n_rel, a, b, c).total;
}

extern "C" size_t rtk_from_tables_workspace_bytes(int64_t batch, int64_t n_rel) {
    if (batch <= 0 || n_rel <= 0) return 0;
    return carve_ft(nullptr, batch, n_rel).total;
}

static int check_tables(const char *fn, const void *core, int a, int b, int c, const void *R, int64_t n_rel, float *tables,
                        void *ws, size_t ws_bytes, int dtype) {
    RTK_REQUIRE(core && R && tables, RTK_ERR_BAD_ARG, "%s: null operand", fn);
    RTK_REQUIRE(a > 0 && b > 0 && c > 0 && n_rel > 0, RTK_ERR_BAD_ARG, "%s: sizes must be positive", fn);
    RTK_REQUIRE(b == c, RTK_ERR_BAD_ARG, "%s: subject rank b=%d must equal object rank c=%d (the reference's view(-1, b) raises otherwise)", fn, b, c);
    RTK_REQUIRE(n_rel < (1ll << 31), RTK_ERR_UNSUPPORTED, "%s: n_rel exceeds 2^31-1", fn);
    RTK_REQUIRE((reinterpret_cast<uintptr_t>(tables) & 255) == 0, RTK_ERR_BAD_ARG, "%s: tables must be 256-byte aligned", fn);
    const size_t need = rtk_relation_tables_workspace_bytes(dtype, n_rel, a, b, c);
    RTK_REQUIRE(ws && ws_bytes >= need, RTK_ERR_WORKSPACE, "%s: workspace of %zu bytes given, %zu needed", fn, ws_bytes, need);
    RTK_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 255) == 0, RTK_ERR_WORKSPACE, "%s: workspace must be 256-byte aligned", fn);
    return RTK_OK;
}

extern "C" int rtk_relation_tables_f32(const float *core, int a, int b, int c, const float *R, int64_t n_rel,
                                       float *tables, void *workspace, size_t workspace_bytes, void *stream) {
    int rc = check_tables("rtk_relation_tables_f32", core, a, b, c, R, n_rel, tables, workspace, workspace_bytes, RTK_F32);
    if (rc != RTK_OK) return rc;
    const TablesWs ws = carve_tables(workspace, RTK_F32, n_rel, a, b, c);
    return rtk_relation_tables_f32_impl(core, a, b, c, R, n_rel, tables, ws.r_packed, (hipStream_t)stream);
}

extern "C" int rtk_relation_tables_bf16(const void *core, int a, int b, int c, const void *R, int64_t n_rel,
                                        float *tables, void *workspace, size_t workspace_bytes, void *stream) {
    int rc = check_tables("rtk_relation_tables_bf16", core, a, b, c, R, n_rel, tables, workspace, workspace_bytes, RTK_BF16);
    if (rc != RTK_OK) return rc;
    const TablesWs ws = carve_tables(workspace, RTK_BF16, n_rel, a, b, c);
    return rtk_relation_tables_bf16_impl(core, a, b, c, R, n_rel, tables, ws.core_t, ws.r_packed, (hipStream_t)stream);
}

static int check_ft(const char *fn, const float *tables, int64_t n_rel, int b, int c, const void *S, int64_t n_sub,
                    const void *rel_idx, const void *sub_idx, int64_t batch, const void *v_out, const void *q_packed,
                    void *ws, size_t ws_bytes) {
    RTK_REQUIRE(tables && S && rel_idx && sub_idx, RTK_ERR_BAD_ARG, "%s: null operand", fn);
    RTK_REQUIRE(b > 0 && c > 0 && n_rel > 0 && n_sub > 0 && batch > 0, RTK_ERR_BAD_ARG, "%s: sizes must be positive", fn);
    RTK_REQUIRE(b == c, RTK_ERR_BAD_ARG, "%s: subject rank b=%d must equal object rank c=%d", fn, b, c);
    RTK_REQUIRE(n_rel < (1ll << 31) && batch < (1ll << 31), RTK_ERR_UNSUPPORTED, "%s: n_rel/batch exceed 2^31-1", fn);
    RTK_REQUIRE(v_out || q_packed, RTK_ERR_BAD_ARG, "%s: both outputs are NULL", fn);
    const size_t need = rtk_from_tables_workspace_bytes(batch, n_rel);
    RTK_REQUIRE(ws && ws_bytes >= need, RTK_ERR_WORKSPACE, "%s: workspace of %zu bytes given, %zu needed", fn, ws_bytes, need);
    RTK_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 255) == 0, RTK_ERR_WORKSPACE, "%s: workspace must be 256-byte aligned", fn);
    return RTK_OK;
}

extern "C" int rtk_query_vectors_from_tables_f32(const float *tables, int64_t n_rel, int b, int c, const float *S,
                                                 int64_t n_sub, const int64_t *rel_idx, const int64_t *sub_idx,
                                                 int64_t batch, float *v_out, void *q_packed, void *workspace,
                                                 size_t workspace_bytes, void *stream) {
    int rc = check_ft("rtk_query_vectors_from_tables_f32", tables, n_rel, b, c, S, n_sub, rel_idx, sub_idx, batch, v_out,
                      q_packed, workspace, workspace_bytes);
    if (rc != RTK_OK) return rc;
    return rtk_from_tables_f32_impl(tables, n_rel, b, c, S, n_sub, rel_idx, sub_idx, batch, v_out, q_packed,
                                    carve_ft(workspace, batch, n_rel), (hipStream_t)stream);
}

extern "C" int rtk_query_vectors_from_tables_bf16(const float *tables, int64_t n_rel, int b, int c, const void *S,
                                                  int64_t n_sub, const int64_t *rel_idx, const int64_t *sub_idx,
                                                  int64_t batch, float *v_out, void *q_packed, void *workspace,
                                                  size_t workspace_bytes, void *stream) {
    int rc = check_ft("rtk_query_vectors_from_tables_bf16", tables, n_rel, b, c, S, n_sub, rel_idx, sub_idx, batch, v_out,
                      q_packed, workspace, workspace_bytes);
    if (rc != RTK_OK) return rc;
    return rtk_from_tables_bf16_impl(tables, n_rel, b, c, S, n_sub, rel_idx, sub_idx, batch, v_out, q_packed,
                                     carve_ft(workspace, batch, n_rel), (hipStream_t)stream);
}

// Stage 1 split over ranks BY RELATION: only the queries whose relation id is congruent to `part` modulo `n_parts`
// are contracted, into their rows of v_out; every other row of v_out is left untouched.
extern "C" int rtk_query_vectors_from_tables_part_f32(const float *tables, int64_t n_rel, int b, int c, const float *S,
                                                      int64_t n_sub, const int64_t *rel_idx, const int64_t *sub_idx,
                                                      int64_t batch, int part, int n_parts, float *v_out, void *workspace,
                                                      size_t workspace_bytes, void *stream) {
    int rc = check_ft("rtk_query_vectors_from_tables_part_f32", tables, n_rel, b, c, S, n_sub, rel_idx, sub_idx, batch, v_out,
                      nullptr, workspace, workspace_bytes);
    if (rc != RTK_OK) return rc;
    RTK_REQUIRE(v_out && n_parts >= 1 && part >= 0 && part < n_parts, RTK_ERR_BAD_ARG,
                "rtk_query_vectors_from_tables_part_f32: v_out must be given, 0 <= part < n_parts");
    return rtk_from_tables_f32_impl(tables, n_rel, b, c, S, n_sub, rel_idx, sub_idx, batch, v_out, nullptr,
                                    carve_ft(workspace, batch, n_rel), (hipStream_t)stream, part, n_parts);
}

extern "C" int rtk_query_vectors_from_tables_part_bf16(const float *tables, int64_t n_rel, int b, int c, const void *S,
                                                       int64_t n_sub, const int64_t *rel_idx, const int64_t *sub_idx,
                                                       int64_t batch, int part, int n_parts, float *v_out, void *workspace,
                                                       size_t workspace_bytes, void *stream) {
    int rc = check_ft("rtk_query_vectors_from_tables_part_bf16", tables, n_rel, b, c, S, n_sub, rel_idx, sub_idx, batch, v_out,
                      nullptr, workspace, workspace_bytes);
    if (rc != RTK_OK) return rc;
    RTK_REQUIRE(v_out && n_parts >= 1 && part >= 0 && part < n_parts, RTK_ERR_BAD_ARG,
                "rtk_query_vectors_from_tables_part_bf16: v_out must be given, 0 <= part < n_parts");
    return rtk_from_tables_bf16_impl(tables, n_rel, b, c, S, n_sub, rel_idx, sub_idx, batch, v_out, nullptr,
                                     carve_ft(workspace, batch, n_rel), (hipStream_t)stream, part, n_parts);
}

// ---- kernel timer ------------------------------------------------------------------------------------------
namespace {
struct RtkTimer { hipEvent_t ev[2]; bool launched; };
thread_local RtkTimer *g_armed = nullptr;
}  // namespace
bool rtk_take_launch_events(hipEvent_t *start, hipEvent_t *stop) {
    RtkTimer *t = g_armed;
    if (!t) return false;
    g_armed = nullptr;
    t->launched = true;
    *start = t->ev[0];
    *stop = t->ev[1];
    return true;
}
extern "C" int rtk_timer_create(void **timer) {
    RTK_REQUIRE(timer, RTK_ERR_BAD_ARG, "rtk_timer_create: null argument");
    RtkTimer *t = new RtkTimer{};
    if (hipEventCreate(&t->ev[0]) != hipSuccess || hipEventCreate(&t->ev[1]) != hipSuccess) {
        delete t;
        rtk_set_error("rtk_timer_create: hipEventCreate failed");
        return RTK_ERR_LAUNCH;
    }
    *timer = t;
    return RTK_OK;
}
extern "C" int rtk_timer_arm(void *timer) {
    RTK_REQUIRE(timer, RTK_ERR_BAD_ARG, "rtk_timer_arm: null timer");
    g_armed = (RtkTimer *)timer;
    g_armed->launched = false;
    return RTK_OK;
}
extern "C" int rtk_timer_elapsed_ms(void *timer, float *ms) {
    RTK_REQUIRE(timer && ms, RTK_ERR_BAD_ARG, "rtk_timer_elapsed_ms: null argument");
    RtkTimer *t = (RtkTimer *)timer;
    RTK_REQUIRE(t->launched, RTK_ERR_BAD_ARG, "rtk_timer_elapsed_ms: no score kernel was launched on this timer (rtk_timer_arm, then a score call of the same thread)");
    hipError_t e = hipEventSynchronize(t->ev[1]);
    if (e == hipSuccess) e = hipEventElapsedTime(ms, t->ev[0], t->ev[1]);
    if (e != hipSuccess) {
        (void)hipGetLastError();            // (not left behind for the caller's next HIP call to trip over)
        rtk_set_error("rtk_timer_elapsed_ms: %s", hipGetErrorString(e));
        return RTK_ERR_LAUNCH;
    }
    return RTK_OK;
}
extern "C" int rtk_timer_destroy(void *timer) {
    RTK_REQUIRE(timer, RTK_ERR_BAD_ARG, "rtk_timer_destroy: null timer");
    RtkTimer *t = (RtkTimer *)timer;
    if (g_armed == t) g_armed = nullptr;
    (void)hipEventDestroy(t->ev[0]);
    (void)hipEventDestroy(t->ev[1]);
    delete t;
    return RTK_OK;
}

extern "C" int rtk_version(void) { return 211; }
extern "C" const char *rtk_last_error_string(void) { return g_err; }

extern "C" size_t rtk_workspace_bytes(int dtype, int64_t batch, int64_t n_rel, int a, int b, int c) {
    if (batch <= 0 || n_rel <= 0 || a <= 0 || b <= 0 || c <= 0) return 0;
    return carve(nullptr, dtype, batch, n_rel, a, b, c).total;
}

extern "C" size_t rtk_packed_query_bytes(int dtype, int64_t batch, int c) {
    if (batch <= 0 || c <= 0) return 0;
    return packed_bytes(dtype, batch, c);
}

extern "C" int rtk_read_error_flag(void *workspace, void *stream, uint32_t *host_flag_out) {
    RTK_REQUIRE(workspace && host_flag_out, RTK_ERR_BAD_ARG, "rtk_read_error_flag: null argument");
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemcpyAsync(host_flag_out, workspace, 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    // clear on the SAME stream, behind the read: later launches on it start from a clean word
    if (e == hipSuccess && *host_flag_out) e = hipMemsetAsync(workspace, 0, 4, st);
    if (e != hipSuccess) {
        rtk_set_error("rtk_read_error_flag: %s", hipGetErrorString(e));
        return RTK_ERR_LAUNCH;
    }
    return RTK_OK;
}

static int check_common(const char *fn, const void *core, int a, int b, int c, const void *R, int64_t n_rel,
                        const void *S, int64_t n_sub, const void *rel_idx, const void *sub_idx,
                        int64_t batch, void *ws, size_t ws_bytes, int dtype) {
    RTK_REQUIRE(core && R && S && rel_idx && sub_idx, RTK_ERR_BAD_ARG, "%s: null operand", fn);
    RTK_REQUIRE(a > 0 && b > 0 && c > 0 && n_rel > 0 && n_sub > 0 && batch > 0, RTK_ERR_BAD_ARG,
                "%s: sizes must be positive (a=%d b=%d c=%d n_rel=%lld n_sub=%lld batch=%lld)", fn, a, b, c,
                (long long)n_rel, (long long)n_sub, (long long)batch);
    // the reference's .view(-1, b) of a (B,1,c) tensor (asymmetric/R_TuckER.py:46) only works for b == c
    RTK_REQUIRE(b == c, RTK_ERR_BAD_ARG, "%s: subject rank b=%d must equal object rank c=%d (the reference's view(-1, b) raises otherwise)", fn, b, c);
    RTK_REQUIRE(n_rel < (1ll << 31) && batch < (1ll << 31), RTK_ERR_UNSUPPORTED, "%s: n_rel/batch exceed 2^31-1", fn);
    const size_t need = rtk_workspace_bytes(dtype, batch, n_rel, a, b, c);
    RTK_REQUIRE(ws && ws_bytes >= need, RTK_ERR_WORKSPACE, "%s: workspace of %zu bytes given, %zu needed", fn, ws_bytes, need);
    RTK_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 255) == 0, RTK_ERR_WORKSPACE, "%s: workspace must be 256-byte aligned", fn);
    return RTK_OK;
}

extern "C" int rtk_query_vectors_f32(const float *core, int a, int b, int c, const float *R, int64_t n_rel,
                                     const float *S, int64_t n_sub, const int64_t *rel_idx,
                                     const int64_t *sub_idx, int64_t batch, float *v_out, void *q_packed,
                                     void *workspace, size_t workspace_bytes, void *stream) {
    int rc = check_common("rtk_query_vectors_f32", core, a, b, c, R, n_rel, S, n_sub, rel_idx, sub_idx, batch,
                          workspace, workspace_bytes, RTK_F32);
    if (rc != RTK_OK) return rc;
    RTK_REQUIRE(v_out || q_packed, RTK_ERR_BAD_ARG, "rtk_query_vectors_f32: both outputs are NULL");
    RtkWorkspace ws = carve(workspace, RTK_F32, batch, n_rel, a, b, c);
    return rtk_query_vectors_f32_impl(core, a, b, c, R, n_rel, S, n_sub, rel_idx, sub_idx, batch, v_out,
                                      q_packed, ws, (hipStream_t)stream);
}

extern "C" int rtk_score_1vN_f32(const float *core, int a, int b, int c, const float *R, int64_t n_rel,
                                 const float *S, int64_t n_sub, const float *O, int64_t n_local,
                                 const int64_t *rel_idx, const int64_t *sub_idx, int64_t batch, float *out,
                                 int64_t ld_out, unsigned flags, void *workspace, size_t workspace_bytes,
                                 void *stream) {
    int rc = check_common("rtk_score_1vN_f32", core, a, b, c, R, n_rel, S, n_sub, rel_idx, sub_idx, batch,
                          workspace, workspace_bytes, RTK_F32);
    if (rc != RTK_OK) return rc;
    RTK_REQUIRE(O && out && n_local > 0 && ld_out >= n_local, RTK_ERR_BAD_ARG, "rtk_score_1vN_f32: bad O/out/n_local/ld_out");
    RtkWorkspace ws = carve(workspace, RTK_F32, batch, n_rel, a, b, c);
    const bool exact = (flags & RTK_SCORE_EXACT_F32) || !rtk_split_ksteps_supported(c);
    rc = rtk_query_vectors_f32_impl(core, a, b, c, R, n_rel, S, n_sub, rel_idx, sub_idx, batch,
                                    exact ? ws.v : nullptr, exact ? nullptr : ws.q_packed, ws, (hipStream_t)stream);
    if (rc != RTK_OK) return rc;
    if (exact) return rtk_score_f32(ws.v, batch, c, O, n_local, out, ld_out, flags & RTK_SCORE_SIGMOID, stream);
    return rtk_score_packed_f32(ws.q_packed, batch, c, O, n_local, out, ld_out,
                                flags & (RTK_SCORE_SIGMOID | RTK_SCORE_SIGMOID_FAST | RTK_SCORE_KERNEL_MASK), stream);
}

extern "C" int rtk_query_vectors_bf16(const void *core, int a, int b, int c, const void *R, int64_t n_rel,
                                      const void *S, int64_t n_sub, const int64_t *rel_idx,
                                      const int64_t *sub_idx, int64_t batch, float *v_out, void *q_packed,
                                      void *workspace, size_t workspace_bytes, void *stream) {
    int rc = check_common("rtk_query_vectors_bf16", core, a, b, c, R, n_rel, S, n_sub, rel_idx, sub_idx, batch,
                          workspace, workspace_bytes, RTK_BF16);
    if (rc != RTK_OK) return rc;
    RTK_REQUIRE(v_out || q_packed, RTK_ERR_BAD_ARG, "rtk_query_vectors_bf16: both outputs are NULL");
    RtkWorkspace ws = carve(workspace, RTK_BF16, batch, n_rel, a, b, c);
    return rtk_query_vectors_bf16_impl(core, a, b, c, R, n_rel, S, n_sub, rel_idx, sub_idx, batch, v_out,
                                       q_packed, ws, (hipStream_t)stream);
}

extern "C" int rtk_score_1vN_bf16(const void *core, int a, int b, int c, const void *R, int64_t n_rel,
                                  const void *S, int64_t n_sub, const void *O, int64_t n_local,
                                  const int64_t *rel_idx, const int64_t *sub_idx, int64_t batch, float *out,
                                  int64_t ld_out, unsigned flags, void *workspace, size_t workspace_bytes,
                                  void *stream) {
    int rc = check_common("rtk_score_1vN_bf16", core, a, b, c, R, n_rel, S, n_sub, rel_idx, sub_idx, batch,
                          workspace, workspace_bytes, RTK_BF16);
    if (rc != RTK_OK) return rc;
    RTK_REQUIRE(O && out && n_local > 0 && ld_out >= n_local, RTK_ERR_BAD_ARG, "rtk_score_1vN_bf16: bad O/out/n_local/ld_out");
    RTK_REQUIRE(c <= 512, RTK_ERR_UNSUPPORTED, "rtk_score_1vN_bf16: c=%d > 512 not supported", c);
    RtkWorkspace ws = carve(workspace, RTK_BF16, batch, n_rel, a, b, c);
    rc = rtk_query_vectors_bf16_impl(core, a, b, c, R, n_rel, S, n_sub, rel_idx, sub_idx, batch, nullptr,
                                     ws.q_packed, ws, (hipStream_t)stream);
    if (rc != RTK_OK) return rc;
    return rtk_score_packed_bf16(ws.q_packed, batch, c, O, n_local, out, ld_out,
                                 flags & (RTK_SCORE_SIGMOID | RTK_SCORE_SIGMOID_FAST | RTK_SCORE_OUT_BF16), stream);
}
