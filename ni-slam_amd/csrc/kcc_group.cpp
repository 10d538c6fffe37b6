// kcc_group.cpp -- multi-GPU entry points of the C ABI (include/nislam_kcc.h "nik_group").
//
// The KCC front end shards by independent units: frame pairs for tracking, candidate key frames for loop
// closure (reference loop: src/loop_closure.cc:36-73; the CorrelationFlow object MapBuilder and LoopClosure share:
// src/map_builder.cc:23-26).  No data moves between GPUs on the data path.  The two exchanges are
//   * one RCCL all-reduce of 4 doubles per batch -- [sum PSR_t, sum PSR_r, sum |t|^2, count], reduced on each device from
//     the raw surface results (nik_residual_stats_dev) and summed over the GPUs (xGMI);
//   * one RCCL all-gather of each GPU's best loop-closure candidate (8 doubles) followed by the reference's selection rule
//     (loop_closure.cc:61-65: a strictly larger response.sum() wins, so the first candidate in global order wins ties).
// Two deployments share the code: one process driving every GPU of the node (nik_group_create_local: what a C++ MapBuilder
// linking this library uses), and one process per GPU (nik_group_create_rank: torchrun / MPI style launchers).
//
// RCCL is loaded at run time (dlopen "librccl.so.1"): the core library keeps working where RCCL is absent, and a host
// process that already carries an RCCL (e.g. PyTorch's) shares it instead of loading a second copy.
#include "../../include/nislam_kcc.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>      // types and enums only; every function is resolved with dlsym
#else
// RCCL's headers are absent: declare the handful of types the entry points use (ABI of rccl.h / nccl.h 2.x), so that the
// library still builds; at run time rccl() then reports "RCCL not found" unless a librccl can be dlopen-ed after all
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1, ncclInt32 = 2, ncclInt = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5,
               ncclFloat16 = 6, ncclHalf = 6, ncclFloat32 = 7, ncclFloat = 7, ncclFloat64 = 8, ncclDouble = 8 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3, ncclAvg = 4 } ncclRedOp_t;
}
#endif

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
    std::string path;           // file the symbols were bound from (dladdr of ncclAllReduce)
    bool shared = false;        // the host process had already loaded it (RTLD_NOLOAD hit: e.g. PyTorch's RCCL)
    bool ok = false;
};

static void rccl_load(Rccl& r);
Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;                 // (two threads creating groups at once must not race the dlopen / dlsym)
    std::call_once(once, [] { rccl_load(r); });
    return r;
}
static void rccl_load(Rccl& r) {
    for (const char* name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" }) {
        r.lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);          // reuse a copy the host process already loaded
        r.shared = r.lib != nullptr;
        if (!r.lib) r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (r.lib) break;
    }
    if (!r.lib) { const char* de = dlerror(); r.err = std::string("RCCL not found: ") + (de ? de : "librccl.so.1"); return; }
#define SYM(field, sym) do { *(void**)(&r.field) = dlsym(r.lib, sym); if (!r.field) { r.err = std::string("RCCL symbol missing: ") + sym; return; } } while (0)
    SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommInitAll, "ncclCommInitAll");
    SYM(CommDestroy, "ncclCommDestroy"); SYM(CommCount, "ncclCommCount"); SYM(AllReduce, "ncclAllReduce"); SYM(AllGather, "ncclAllGather");
    SYM(GroupStart, "ncclGroupStart"); SYM(GroupEnd, "ncclGroupEnd"); SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    Dl_info info;
    if (dladdr((void*)r.AllReduce, &info) && info.dli_fname) r.path = info.dli_fname;
    r.ok = true;
}

// ncclCommInitRank with a deadline.  The first communicator of a multi-GPU job forms here -- inside a process that may already
// carry the launcher's own RCCL world -- and a rank that never arrives (or a bootstrap that cannot connect) makes the call
// block for good: a benchmark driver would then see a timeout instead of a line.  The call runs on a helper thread; past
// $NIK_GROUP_INIT_TIMEOUT seconds (default 90) the group creation FAILS with a clear message (the caller falls back or
// reports), and the helper thread -- still inside RCCL -- is left detached with its own copy of everything it touches.
// A communicator that forms AFTER its creator gave up (the peers arrived late) has an abandoned rank in it: the helper destroys
// it at once (ncclCommAbort where the library has it, else ncclCommDestroy), so that the peers' first collective fails fast
// instead of waiting for a rank that will never call it, and nothing leaks.
struct InitJob { std::mutex mu; std::condition_variable cv; bool done = false, abandoned = false; ncclResult_t res = ncclSuccess; ncclComm_t comm = nullptr; std::string hip_err; };
// $NIK_GROUP_INIT_TIMEOUT in seconds: a number > 0; "0" = no limit (explicitly); anything unparsable = the 90 s default
// (atof() of garbage is 0 and would have switched the guard off silently)
static double init_timeout_seconds() {
    const char* e = getenv("NIK_GROUP_INIT_TIMEOUT");
    if (!e || !*e) return 90.0;
    char* end = nullptr;
    const double v = strtod(e, &end);
    while (end && (*end == ' ' || *end == '\t')) ++end;
    if (end == e || (end && *end) || !(v >= 0.0)) return 90.0;
    return v;
}
static int comm_init_rank_deadline(int device, int world, const ncclUniqueId& u, int rank, ncclComm_t* out, std::string& err) {
    const double limit = init_timeout_seconds();
    auto job = std::make_shared<InitJob>();
    std::thread([job, device, world, u, rank] {
        ncclComm_t c = nullptr; ncclResult_t r = ncclSuccess; std::string he;
        const hipError_t h = hipSetDevice(device);
        if (h != hipSuccess) he = std::string("hipSetDevice: ") + hipGetErrorString(h);
        else r = rccl().CommInitRank(&c, world, u, rank);
        std::lock_guard<std::mutex> lk(job->mu);
        if (job->abandoned && c) {
            typedef ncclResult_t (*abort_fn)(ncclComm_t);
            abort_fn ab = (abort_fn)dlsym(rccl().lib, "ncclCommAbort");
            if (ab) (void)ab(c); else (void)rccl().CommDestroy(c);
            c = nullptr;
        }
        job->comm = c; job->res = r; job->hip_err = he; job->done = true;
        job->cv.notify_all();
    }).detach();
    std::unique_lock<std::mutex> lk(job->mu);
    const bool in_time = limit <= 0 ? (job->cv.wait(lk, [&] { return job->done; }), true)
                                    : job->cv.wait_for(lk, std::chrono::duration<double>(limit), [&] { return job->done; });
    if (!in_time) job->abandoned = true;                     // (under job->mu: the helper sees it before it publishes a communicator)
    if (!in_time) { err = "ncclCommInitRank did not return within " + std::to_string((int)limit) + " s (rank " + std::to_string(rank) + " of " + std::to_string(world) + "; $NIK_GROUP_INIT_TIMEOUT)"; return NIK_ERR_HIP; }
    if (!job->hip_err.empty()) { err = job->hip_err; return NIK_ERR_HIP; }
    if (job->res != ncclSuccess) { err = std::string("ncclCommInitRank: ") + rccl().GetErrorString(job->res); return NIK_ERR_HIP; }
    *out = job->comm;
    return NIK_OK;
}

thread_local std::string g_group_error;

struct Member {                 // one GPU of this process
    nik_ctx* ctx = nullptr;
    int device = 0, rank = 0;
    ncclComm_t comm = nullptr;
    double* d_buf = nullptr;    // [4] all-reduced statistics | [8] my best record | [8 * world] gathered records | [1] pose-graph cost
    double* h_buf = nullptr;    // pinned mirror
    hipStream_t stream = nullptr;   // the context's statistics stream once statistics were requested, else a private one
    hipStream_t own_stream = nullptr;
    hipEvent_t done = nullptr;
    uint8_t* d_stage = nullptr; size_t stage_cap = 0;      // upload staging of nik_group_track_batch
    std::string err;            // error of this member's share of a fan-out call
};

// run f(member index) for every local member concurrently (one host thread per GPU; member 0 on the caller's thread)
template <class F> void for_each_member(size_t n, F f) {
    std::vector<std::thread> th;
    for (size_t r = 1; r < n; ++r) th.emplace_back([&f, r] { f(r); });
    f(0);
    for (std::thread& t : th) t.join();
}

}  // namespace

struct nik_group {
    int world = 1;
    bool use_rccl = false;      // world > 1, or $NIK_GROUP_FORCE_RCCL (tests: the RCCL calls with a single rank)
    bool owns_ctx = false;
    std::vector<Member> m;      // local members (world of them in a local group, one in a rank group)
    std::string err;
    bool stats_inflight = false;
};

namespace {

int gfail(nik_group* g, int code, const std::string& msg) {
    if (g) g->err = msg; else g_group_error = msg;
    return code;
}
#define G_HIP(g, expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return gfail(g, NIK_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)
#define G_NCCL(g, expr) do { ncclResult_t r_ = (expr); if (r_ != ncclSuccess) return gfail(g, NIK_ERR_HIP, std::string(#expr) + ": " + rccl().GetErrorString(r_)); } while (0)

int member_init(nik_group* g, Member& mb) {
    G_HIP(g, hipSetDevice(mb.device));
    G_HIP(g, hipMalloc(&mb.d_buf, sizeof(double) * (size_t)(4 + 8 + 8 * g->world + 1)));
    G_HIP(g, hipHostMalloc(&mb.h_buf, sizeof(double) * (size_t)(4 + 8 + 8 * g->world + 1)));
    G_HIP(g, hipStreamCreateWithFlags(&mb.own_stream, hipStreamNonBlocking));
    G_HIP(g, hipEventCreateWithFlags(&mb.done, hipEventDisableTiming));
    mb.stream = mb.own_stream;
    int rc = nik_set_residual_stats(mb.ctx, 1);
    if (rc) return gfail(g, rc, std::string("nik_set_residual_stats: ") + nik_last_error(mb.ctx));
    return NIK_OK;
}

}  // namespace

extern "C" {

const char* nik_group_last_error(const nik_group* g) { return g ? g->err.c_str() : g_group_error.c_str(); }

void nik_group_shard(int n, int world, int rank, int* begin, int* end) {
    // contiguous shards whose sizes differ by at most one, order preserved (rank order == global unit order)
    const int base = world > 0 ? n / world : 0, rem = world > 0 ? n % world : 0;
    const int b = rank * base + std::min(rank, rem);
    if (begin) *begin = b;
    if (end) *end = b + base + (rank < rem ? 1 : 0);
}

// The reference's winner rule over the gathered records (loop_closure.cc:61-65): records[8 r] = response.sum() of rank r's best
// candidate, records[8 r + 1] = its global index (< 0: rank r has none).  A strictly larger score wins, so among equal scores
// the lowest rank -- ranks hold contiguous shards in candidate order -- i.e. the first candidate in global order.  Returns
// the winning RANK (-1: no candidate anywhere).  Host-only.
int nik_group_pick_best(const double* records, int world) {
    int bi = -1; double bs = -3.0;                                   // LoopClosureResult(): response(-1,-1,-1)  (loop_closure.h:14)
    for (int r = 0; r < world; ++r)
        if (records[8 * r + 1] >= 0 && records[8 * r] > bs) { bs = records[8 * r]; bi = r; }
    return bi;
}

// number of ranks the group's RCCL communicator really spans (ncclCommCount); 0 when the group runs without RCCL
// (a single member, $NIK_GROUP_FORCE_RCCL unset) -- a machine-checkable fact for bench.py's JSON line
int nik_group_comm_ranks(const nik_group* g) {
    if (!g || !g->use_rccl || g->m.empty() || !g->m[0].comm || !rccl().ok) return 0;
    int n = 0;
    return rccl().CommCount(g->m[0].comm, &n) == ncclSuccess ? n : -1;
}

// which RCCL the library bound (machine-checkable fact for bench.py's multi_gpu object): the file of ncclAllReduce, and whether
// it was the copy the host process had already loaded (PyTorch's) or one this library loaded itself.  NULL: RCCL not loaded.
const char* nik_group_rccl_library(int* shared_with_host) {
    Rccl& r = rccl();
    if (shared_with_host) *shared_with_host = r.shared ? 1 : 0;
    return r.ok ? r.path.c_str() : nullptr;
}

int nik_group_unique_id(uint8_t id[NIK_GROUP_ID_BYTES]) {
    if (!id) return NIK_ERR_INVALID_ARG;
    static_assert(sizeof(ncclUniqueId) <= NIK_GROUP_ID_BYTES, "unique id does not fit");
    if (!rccl().ok) return gfail(nullptr, NIK_ERR_HIP, rccl().err);
    ncclUniqueId u;
    G_NCCL(nullptr, rccl().GetUniqueId(&u));
    memset(id, 0, NIK_GROUP_ID_BYTES);
    memcpy(id, &u, sizeof(u));
    return NIK_OK;
}

void nik_group_destroy(nik_group* g) {
    if (!g) return;
    for (Member& mb : g->m) {
        (void)hipSetDevice(mb.device);
        if (mb.own_stream) (void)hipStreamSynchronize(mb.own_stream);
        if (mb.ctx) (void)nik_synchronize(mb.ctx);
        if (mb.comm && rccl().ok) (void)rccl().CommDestroy(mb.comm);
        if (mb.done) (void)hipEventDestroy(mb.done);
        if (mb.own_stream) (void)hipStreamDestroy(mb.own_stream);
        (void)hipFree(mb.d_buf); (void)hipFree(mb.d_stage);
        if (mb.h_buf) (void)hipHostFree(mb.h_buf);
        if (g->owns_ctx && mb.ctx) nik_destroy(mb.ctx);
    }
    delete g;
}

int nik_group_create_rank(nik_ctx* ctx, int rank, int world, const uint8_t id[NIK_GROUP_ID_BYTES], nik_group** out) {
    if (!ctx || !out || world <= 0 || rank < 0 || rank >= world || (world > 1 && !id)) return gfail(nullptr, NIK_ERR_INVALID_ARG, "bad argument");
    *out = nullptr;
    nik_group* g = new nik_group();
    g->world = world; g->owns_ctx = false;
    g->m.resize(1);
    Member& mb = g->m[0];
    mb.ctx = ctx; mb.rank = rank;
    mb.device = nik_device(ctx);
    int rc = member_init(g, mb);
    g->use_rccl = world > 1 || getenv("NIK_GROUP_FORCE_RCCL") != nullptr;
    if (!rc && g->use_rccl) {
        if (!rccl().ok) rc = gfail(g, NIK_ERR_HIP, rccl().err);
        else {
            ncclUniqueId u;
            if (id) memcpy(&u, id, sizeof(u));
            else if (rccl().GetUniqueId(&u) != ncclSuccess) memset(&u, 0, sizeof(u));
            std::string ierr;
            if (comm_init_rank_deadline(mb.device, world, u, rank, &mb.comm, ierr)) rc = gfail(g, NIK_ERR_HIP, ierr);
        }
    }
    if (rc) { g_group_error = g->err; nik_group_destroy(g); return rc; }
    *out = g;
    return NIK_OK;
}

int nik_group_create_local(const nik_config* cfg, int image_height, int image_width, int max_batch, int max_frames,
                           int n_devices, const int* devices, nik_group** out) {
    if (!cfg || !out || n_devices <= 0) return gfail(nullptr, NIK_ERR_INVALID_ARG, "bad argument");
    *out = nullptr;
    nik_group* g = new nik_group();
    g->world = n_devices; g->owns_ctx = true;
    g->m.resize(n_devices);
    int rc = NIK_OK;
    std::vector<int> devs(n_devices);
    for (int i = 0; i < n_devices && !rc; ++i) {
        Member& mb = g->m[i];
        mb.rank = i; mb.device = devs[i] = devices ? devices[i] : i;
        rc = nik_create(cfg, image_height, image_width, max_batch, max_frames, mb.device, &mb.ctx);
        if (rc) { gfail(g, rc, std::string("nik_create on device ") + std::to_string(mb.device) + ": " + nik_last_error(nullptr)); break; }
        rc = member_init(g, mb);
    }
    g->use_rccl = n_devices > 1 || getenv("NIK_GROUP_FORCE_RCCL") != nullptr;
    if (!rc && g->use_rccl) {
        if (!rccl().ok) rc = gfail(g, NIK_ERR_HIP, rccl().err);
        else {
            std::vector<ncclComm_t> comms(n_devices);
            ncclResult_t r = rccl().CommInitAll(comms.data(), n_devices, devs.data());
            if (r != ncclSuccess) rc = gfail(g, NIK_ERR_HIP, std::string("ncclCommInitAll: ") + rccl().GetErrorString(r));
            else for (int i = 0; i < n_devices; ++i) g->m[i].comm = comms[i];
        }
    }
    if (rc) { g_group_error = g->err; nik_group_destroy(g); return rc; }
    *out = g;
    return NIK_OK;
}

int nik_group_world(const nik_group* g) { return g ? g->world : 0; }
int nik_group_local_count(const nik_group* g) { return g ? (int)g->m.size() : 0; }
nik_ctx* nik_group_ctx(nik_group* g, int local_index) { return (g && local_index >= 0 && local_index < (int)g->m.size()) ? g->m[local_index].ctx : nullptr; }
int nik_group_rank(const nik_group* g, int local_index) { return (g && local_index >= 0 && local_index < (int)g->m.size()) ? g->m[local_index].rank : -1; }

// all-reduce of the latest batch's residual statistics: every member's 4 doubles are summed over the whole group, on the
// devices (each context's statistics stream; RCCL over xGMI between GPUs).  Asynchronous: the result is fetched by
// nik_group_residual_result (or here when out != NULL).
int nik_group_allreduce_residual(nik_group* g, double out[4]) {
    if (!g) return NIK_ERR_INVALID_ARG;
    const bool multi = g->use_rccl;
    std::vector<double*> src(g->m.size());
    for (size_t i = 0; i < g->m.size(); ++i) {
        Member& mb = g->m[i];
        G_HIP(g, hipSetDevice(mb.device));
        void* st = nullptr;
        int rc = nik_residual_stats_dev(mb.ctx, &src[i], &st);
        if (rc) return gfail(g, rc, std::string("nik_residual_stats_dev: ") + nik_last_error(mb.ctx));
        mb.stream = (hipStream_t)st;
    }
    // (an error inside the RCCL group must not leave it open -- every later collective of the process would be deferred
    // for ever: remember the first error, always close the group, then report)
    std::string first_err;
    if (multi) G_NCCL(g, rccl().GroupStart());
    for (size_t i = 0; i < g->m.size() && first_err.empty(); ++i) {
        Member& mb = g->m[i];
        hipError_t he = hipSetDevice(mb.device);
        if (he != hipSuccess) { first_err = std::string("hipSetDevice: ") + hipGetErrorString(he); break; }
        if (multi) {
            const ncclResult_t nr = rccl().AllReduce(src[i], mb.d_buf, 4, ncclDouble, ncclSum, mb.comm, mb.stream);
            if (nr != ncclSuccess) first_err = std::string("ncclAllReduce: ") + rccl().GetErrorString(nr);
        } else {
            he = hipMemcpyAsync(mb.d_buf, src[i], sizeof(double) * 4, hipMemcpyDeviceToDevice, mb.stream);
            if (he != hipSuccess) first_err = std::string("hipMemcpyAsync: ") + hipGetErrorString(he);
        }
    }
    if (multi) { const ncclResult_t nr = rccl().GroupEnd(); if (nr != ncclSuccess && first_err.empty()) first_err = std::string("ncclGroupEnd: ") + rccl().GetErrorString(nr); }
    if (!first_err.empty()) return gfail(g, NIK_ERR_HIP, first_err);
    Member& m0 = g->m[0];
    G_HIP(g, hipSetDevice(m0.device));
    G_HIP(g, hipMemcpyAsync(m0.h_buf, m0.d_buf, sizeof(double) * 4, hipMemcpyDeviceToHost, m0.stream));
    G_HIP(g, hipEventRecord(m0.done, m0.stream));
    g->stats_inflight = true;
    return out ? nik_group_residual_result(g, out) : NIK_OK;
}

int nik_group_residual_result(nik_group* g, double out[4]) {
    if (!g || !out) return NIK_ERR_INVALID_ARG;
    if (!g->stats_inflight) return gfail(g, NIK_ERR_NOT_READY, "no all-reduce in flight");
    G_HIP(g, hipEventSynchronize(g->m[0].done));
    memcpy(out, g->m[0].h_buf, sizeof(double) * 4);
    g->stats_inflight = false;                   // fetched: a second fetch without a new all-reduce is an error, not stale data
    return NIK_OK;
}

// Loop closure over a sharded candidate set: every member contributes its best candidate (global index, result), the
// records are all-gathered and the reference's rule picks the winner (loop_closure.cc:61-65; -1: none).
int nik_group_gather_best(nik_group* g, const int* global_index, const nik_pose_result* local_best, int* best_index, nik_pose_result* best) {
    if (!g || !global_index || !local_best || !best_index) return NIK_ERR_INVALID_ARG;
    const bool multi = g->use_rccl;
    const int W8 = 8 * g->world;
    for (size_t i = 0; i < g->m.size(); ++i) {
        Member& mb = g->m[i];
        G_HIP(g, hipSetDevice(mb.device));
        double* rec = mb.h_buf + 4;
        if (global_index[i] >= 0) {
            const nik_pose_result& r = local_best[i];
            rec[0] = r.info[0] + r.info[1] + r.info[2]; rec[1] = (double)global_index[i];
            for (int k = 0; k < 3; ++k) { rec[2 + k] = r.pose[k]; rec[5 + k] = r.info[k]; }
        } else {
            for (int k = 0; k < 8; ++k) rec[k] = 0.0;
            rec[1] = -1.0;
        }
        G_HIP(g, hipMemcpyAsync(mb.d_buf + 4, rec, sizeof(double) * 8, hipMemcpyHostToDevice, mb.own_stream));
    }
    std::string first_err;
    if (multi) G_NCCL(g, rccl().GroupStart());
    for (size_t i = 0; i < g->m.size() && first_err.empty(); ++i) {
        Member& mb = g->m[i];
        hipError_t he = hipSetDevice(mb.device);
        if (he != hipSuccess) { first_err = std::string("hipSetDevice: ") + hipGetErrorString(he); break; }
        if (multi) {
            const ncclResult_t nr = rccl().AllGather(mb.d_buf + 4, mb.d_buf + 12, 8, ncclDouble, mb.comm, mb.own_stream);
            if (nr != ncclSuccess) first_err = std::string("ncclAllGather: ") + rccl().GetErrorString(nr);
        } else {
            he = hipMemcpyAsync(mb.d_buf + 12, mb.d_buf + 4, sizeof(double) * 8, hipMemcpyDeviceToDevice, mb.own_stream);
            if (he != hipSuccess) first_err = std::string("hipMemcpyAsync: ") + hipGetErrorString(he);
        }
    }
    if (multi) { const ncclResult_t nr = rccl().GroupEnd(); if (nr != ncclSuccess && first_err.empty()) first_err = std::string("ncclGroupEnd: ") + rccl().GetErrorString(nr); }
    if (!first_err.empty()) return gfail(g, NIK_ERR_HIP, first_err);
    Member& m0 = g->m[0];
    G_HIP(g, hipSetDevice(m0.device));
    G_HIP(g, hipMemcpyAsync(m0.h_buf + 12, m0.d_buf + 12, sizeof(double) * W8, hipMemcpyDeviceToHost, m0.own_stream));
    for (Member& mb : g->m) { G_HIP(g, hipSetDevice(mb.device)); G_HIP(g, hipStreamSynchronize(mb.own_stream)); }
    const double* all = m0.h_buf + 12;
    const int bi = nik_group_pick_best(all, g->world);
    *best_index = bi >= 0 ? (int)all[8 * bi + 1] : -1;
    if (best && bi >= 0) {
        memset(best, 0, sizeof(*best));
        for (int k = 0; k < 3; ++k) { best->pose[k] = all[8 * bi + 2 + k]; best->info[k] = all[8 * bi + 5 + k]; }
        // (the arg-max details of the winner stay with the member that owns it; a member whose own record won returns it whole)
        for (size_t i = 0; i < g->m.size(); ++i) if (global_index[i] == *best_index) *best = local_best[i];
    }
    return NIK_OK;
}

// The pose graph's cost over constraints SHARDED across the group (north star: "RCCL all-reduce ... only for the final
// pose-graph residual sum"): member i holds shard shards[i] (nik_pg_shard_create on ITS device: the constraints its own frame
// pairs produced), evaluates 0.5 sum |r|^2 of its shard at `poses` on the device (kcc_posegraph_dev.hip: wave-shuffle
// reduction) and the group sums the per-member doubles with one ncclAllReduce.  poses: [n_poses][3] as given to the shards
// (NULL: the poses of the previous call).  Every member of a rank group must call it (a collective).
int nik_group_pose_graph_cost(nik_group* g, nik_pg_shard* const* shards, const double* poses, double* cost) {
    if (!g || !shards || !cost) return NIK_ERR_INVALID_ARG;
    const bool multi = g->use_rccl;
    const size_t PG = (size_t)(12 + 8 * g->world);           // its own slot: a residual all-reduce may be in flight in [0, 4)
    std::vector<double*> src(g->m.size(), nullptr);
    std::vector<hipStream_t> st(g->m.size(), nullptr);
    for (size_t i = 0; i < g->m.size(); ++i) {
        Member& mb = g->m[i];
        // shards[i] must live on member i's GPU: the collective below runs on the shard's own stream
        if (!shards[i] || nik_pg_shard_device(shards[i]) != mb.device)
            return gfail(g, NIK_ERR_INVALID_ARG, "nik_group_pose_graph_cost: shard " + std::to_string(i) + " is not on its member's device");
        G_HIP(g, hipSetDevice(mb.device));
        void* s = nullptr;
        int rc = nik_pg_shard_cost_dev(shards[i], poses, &src[i], &s);
        if (rc) return gfail(g, rc, "nik_pg_shard_cost_dev failed");
        st[i] = (hipStream_t)s;
    }
    std::string first_err;
    if (multi) G_NCCL(g, rccl().GroupStart());
    for (size_t i = 0; i < g->m.size() && first_err.empty(); ++i) {
        Member& mb = g->m[i];
        hipError_t he = hipSetDevice(mb.device);
        if (he != hipSuccess) { first_err = std::string("hipSetDevice: ") + hipGetErrorString(he); break; }
        if (multi) {
            const ncclResult_t nr = rccl().AllReduce(src[i], mb.d_buf + PG, 1, ncclDouble, ncclSum, mb.comm, st[i]);
            if (nr != ncclSuccess) first_err = std::string("ncclAllReduce: ") + rccl().GetErrorString(nr);
        } else {
            he = hipMemcpyAsync(mb.d_buf + PG, src[i], sizeof(double), hipMemcpyDeviceToDevice, st[i]);
            if (he != hipSuccess) first_err = std::string("hipMemcpyAsync: ") + hipGetErrorString(he);
        }
    }
    if (multi) { const ncclResult_t nr = rccl().GroupEnd(); if (nr != ncclSuccess && first_err.empty()) first_err = std::string("ncclGroupEnd: ") + rccl().GetErrorString(nr); }
    if (!first_err.empty()) return gfail(g, NIK_ERR_HIP, first_err);
    Member& m0 = g->m[0];
    G_HIP(g, hipSetDevice(m0.device));
    G_HIP(g, hipMemcpyAsync(m0.h_buf + PG, m0.d_buf + PG, sizeof(double), hipMemcpyDeviceToHost, st[0]));
    for (size_t i = 0; i < g->m.size(); ++i) { G_HIP(g, hipSetDevice(g->m[i].device)); G_HIP(g, hipStreamSynchronize(st[i])); }
    // (a local group without RCCL -- one member -- has nothing to add up; with several local members RCCL summed them)
    *cost = m0.h_buf[PG];
    return NIK_OK;
}

// ---- one process, every GPU: sharded batches ---------------------------------------------------------------

// n frame pairs (host u8 images, h_gray[n][H*W]) registered against resident key frames; pair i runs on member
// r = shard(i) with keys[i] / cur_dst[i] naming slots of THAT member's context.  The uploads and the kernels of the
// members overlap; res[n] is complete on return.
int nik_group_track_batch(nik_group* g, int n, const uint8_t* h_gray, const nik_frame* keys, const nik_frame* cur_dst,
                          int not_large_rotation, nik_pose_result* res) {
    if (!g || !h_gray || !keys || !cur_dst || !res || n < 0) return NIK_ERR_INVALID_ARG;
    if (g->m.size() != (size_t)g->world) return gfail(g, NIK_ERR_INVALID_ARG, "nik_group_track_batch needs a local group");
    int dims[6];
    nik_get_dims(g->m[0].ctx, dims);
    const size_t N = (size_t)dims[0] * dims[1];
    std::vector<int> rcs(g->m.size(), NIK_OK);
    for_each_member(g->m.size(), [&](size_t r) {
        int b, e; nik_group_shard(n, g->world, (int)r, &b, &e);
        if (e <= b) return;
        Member& mb = g->m[r];
        const size_t bytes = N * (size_t)(e - b);
        hipError_t he = hipSetDevice(mb.device);
        if (he == hipSuccess && bytes > mb.stage_cap) {
            (void)hipFree(mb.d_stage); mb.d_stage = nullptr; mb.stage_cap = 0;
            he = hipMalloc(&mb.d_stage, bytes);
            if (he == hipSuccess) mb.stage_cap = bytes;
        }
        if (he == hipSuccess) he = hipMemcpy(mb.d_stage, h_gray + N * (size_t)b, bytes, hipMemcpyHostToDevice);
        if (he != hipSuccess) { mb.err = hipGetErrorString(he); rcs[r] = NIK_ERR_HIP; return; }
        rcs[r] = nik_track_batch_dev(mb.ctx, e - b, mb.d_stage, keys + b, cur_dst + b, not_large_rotation, res + b, 1);
        if (rcs[r]) mb.err = nik_last_error(mb.ctx);
    });
    for (size_t r = 0; r < g->m.size(); ++r)
        if (rcs[r]) return gfail(g, rcs[r], std::string("member ") + std::to_string(r) + ": " + g->m[r].err);
    return NIK_OK;
}

// FindLoopClosure's candidate loop over a sharded key-frame store: the query frame (host u8 image) is uploaded to every
// member (slot query_slot), each member registers it against its resident candidates (nik_match), the members' best
// candidates are all-gathered and the reference's rule picks the winner.  cands[r][0..n_cands[r]) are slots of member r;
// global candidate index = position in the concatenation over members.  best_member / best_local: where the winner lives.
int nik_group_match(nik_group* g, const uint8_t* h_query, nik_frame query_slot, const int* n_cands, const nik_frame* const* cands,
                    int* best_member, int* best_local, nik_pose_result* best) {
    if (!g || !h_query || !n_cands || !cands || !best_member || !best_local) return NIK_ERR_INVALID_ARG;
    if (g->m.size() != (size_t)g->world) return gfail(g, NIK_ERR_INVALID_ARG, "nik_group_match needs a local group");
    int dims[6];
    nik_get_dims(g->m[0].ctx, dims);
    std::vector<int> gidx(g->m.size(), -1), first(g->m.size(), 0);
    std::vector<nik_pose_result> lbest(g->m.size());
    int off = 0;
    for (size_t r = 0; r < g->m.size(); ++r) { first[r] = off; off += n_cands[r]; }
    std::vector<int> rcs(g->m.size(), NIK_OK);
    for_each_member(g->m.size(), [&](size_t r) {
        Member& mb = g->m[r];
        if (hipSetDevice(mb.device) != hipSuccess) { mb.err = "hipSetDevice failed"; rcs[r] = NIK_ERR_HIP; return; }
        int rc = nik_intermedium_u8(mb.ctx, h_query, dims[1], query_slot);
        int bi = -1;
        if (!rc && n_cands[r] > 0) rc = nik_match(mb.ctx, query_slot, n_cands[r], cands[r], &bi, nullptr, &lbest[r]);
        if (rc) { mb.err = nik_last_error(mb.ctx); rcs[r] = rc; return; }
        gidx[r] = bi >= 0 ? first[r] + bi : -1;
    });
    for (size_t r = 0; r < g->m.size(); ++r)
        if (rcs[r]) return gfail(g, rcs[r], std::string("member ") + std::to_string(r) + ": " + g->m[r].err);
    int win = -1;
    int rc = nik_group_gather_best(g, gidx.data(), lbest.data(), &win, best);
    if (rc) return rc;
    *best_member = -1; *best_local = -1;
    for (size_t r = 0; r < g->m.size(); ++r)
        if (win >= first[r] && win < first[r] + n_cands[r]) { *best_member = (int)r; *best_local = win - first[r]; }
    return NIK_OK;
}

}  // extern "C"
