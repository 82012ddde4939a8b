// Native RCCL hook for the sharded fit (one process per GPU; round 4).
//
// north_star: "the fit is partitioned across the 8 GPUs of one node by sharding data points and RCCL-reducing the normal
// equations over xGMI", with the host code in Fortran.  splpak_plan_set_allreduce takes ANY sum-all-reduce; until round 3 the
// only one in the repository was a ctypes callback into torch.distributed, so a C / Fortran multi-process caller had no way to
// reach RCCL.  This file gives the library its own: librccl is opened at run time (dlopen: the library does not link against
// it, and a process that already carries an RCCL -- PyTorch ships one -- keeps using that one), the plan's hook becomes
// ncclAllReduce(buf, buf, count, ncclDouble, ncclSum, comm, stream) on the fit's stream, and it declares SPLPAK_AR_ANY_POINTER
// (ncclAllReduce takes any device pointer), so the nested-dissection factorisation is distributed by subtrees.
//
// A caller without RCCL headers (Fortran) can also have the communicator made here: rank 0 draws the ncclUniqueId and hands
// it to the other processes through a file (no MPI needed on one node), see INTEGRATION.md.
#include "plan.hpp"

#include <dlfcn.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>

using namespace splpak;

namespace {

struct UniqueId { char internal[128]; };          // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
using GetUniqueId_t = int (*)(UniqueId *);
using CommInitRank_t = int (*)(void **, int, UniqueId, int);
using CommDestroy_t = int (*)(void *);
using AllReduce_t = int (*)(const void *, void *, size_t, int, int, void *, hipStream_t);
using GetErrorString_t = const char *(*)(int);
using CommInitAll_t = int (*)(void **, int, const int *);

struct Rccl {
    void *handle = nullptr;
    GetUniqueId_t get_unique_id = nullptr;
    CommInitRank_t comm_init_rank = nullptr;
    CommDestroy_t comm_destroy = nullptr;
    AllReduce_t all_reduce = nullptr;
    GetErrorString_t error_string = nullptr;
    CommInitAll_t comm_init_all = nullptr;
    bool tried = false;
};
Rccl g_rccl;
std::mutex g_rccl_mu;

constexpr int NCCL_FLOAT64 = 8, NCCL_SUM = 0;    // rccl.h: ncclFloat64 = ncclDouble = 8, ncclSum = 0

bool rccl_load()
{
    std::lock_guard<std::mutex> lock(g_rccl_mu);
    if (g_rccl.tried) return g_rccl.all_reduce != nullptr;
    g_rccl.tried = true;
    void *h = nullptr;
    if (const char *e = splpak::opt_get("SPLPAK_RCCL_LIB")) {
        h = dlopen(e, RTLD_NOW | RTLD_GLOBAL);          // an explicit library is the ONLY candidate: no silent second choice
    } else {
        // an RCCL the process already carries (PyTorch's) first: two RCCLs in one process would not share their state
        for (const char *name : {"librccl.so.1", "librccl.so"})
            if (!h) h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"})
            if (!h) h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    }
    if (!h) {
        const char *de = dlerror();                     // ONE call: dlerror() clears the message it returns
        set_error(std::string("RCCL: librccl.so could not be opened (") + (de ? de : "not found") + "); set SPLPAK_RCCL_LIB");
        return false;
    }
    g_rccl.handle = h;
    g_rccl.get_unique_id = (GetUniqueId_t)dlsym(h, "ncclGetUniqueId");
    g_rccl.comm_init_rank = (CommInitRank_t)dlsym(h, "ncclCommInitRank");
    g_rccl.comm_destroy = (CommDestroy_t)dlsym(h, "ncclCommDestroy");
    g_rccl.all_reduce = (AllReduce_t)dlsym(h, "ncclAllReduce");
    g_rccl.error_string = (GetErrorString_t)dlsym(h, "ncclGetErrorString");
    g_rccl.comm_init_all = (CommInitAll_t)dlsym(h, "ncclCommInitAll");
    if (!g_rccl.get_unique_id || !g_rccl.comm_init_rank || !g_rccl.comm_destroy || !g_rccl.all_reduce) {
        set_error("RCCL: librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllReduce");
        g_rccl.all_reduce = nullptr;
        return false;
    }
    return true;
}

void rccl_fail(const char *what, int rc)
{
    char buf[256];
    snprintf(buf, sizeof buf, "RCCL: %s failed: %s (%d)", what, g_rccl.error_string ? g_rccl.error_string(rc) : "?", rc);
    set_error(buf);
}

struct RcclHook { void *comm; };

// the plan's sum-all-reduce: in place, enqueued on the fit's stream (SPLPAK_AR_STREAM_ORDERED: plan.hip does not synchronise
// the stream around it)
int32_t rccl_allreduce(void *buf, int64_t count, void *stream, void *user)
{
    RcclHook *h = static_cast<RcclHook *>(user);
    const int rc = g_rccl.all_reduce(buf, buf, (size_t)count, NCCL_FLOAT64, NCCL_SUM, h->comm, (hipStream_t)stream);
    if (rc != 0) { rccl_fail("ncclAllReduce", rc); return 1; }
    return 0;
}

}  // namespace

// ---- RCCL for the ONE-PROCESS multi-GPU plan (dist.hip; round 5, SPLPAK_MPLAN_RCCL=1): one communicator per device of the
// plan from ncclCommInitAll; the plan's sums over the ranks (histogram, normal equations, residuals, solution) then are
// ncclAllReduce calls, one per rank thread on that rank's stream, instead of the peer-copy reduce-scatter + all-gather.
int splpak::rccl_comms_for_devices(int n, const int *devices, void **comms)
{
    for (int i = 0; i < n; ++i)
        for (int j = i + 1; j < n; ++j)
            if (devices[i] == devices[j]) {
                set_error("SPLPAK_MPLAN_RCCL=1: RCCL needs one DISTINCT device per rank (ranks share a device: virtual GPUs)");
                return SPLPAK_E_UNSUPPORTED;
            }
    if (!rccl_load()) return SPLPAK_E_COMM;
    if (!g_rccl.comm_init_all) { set_error("RCCL: librccl.so lacks ncclCommInitAll"); return SPLPAK_E_COMM; }
    const int rc = g_rccl.comm_init_all(comms, n, devices);
    if (rc != 0) { rccl_fail("ncclCommInitAll", rc); return SPLPAK_E_COMM; }
    return 0;
}

int splpak::rccl_allreduce_sum(void *comm, double *buf, long long count, hipStream_t st)
{
    const int rc = g_rccl.all_reduce(buf, buf, (size_t)count, NCCL_FLOAT64, NCCL_SUM, comm, st);
    if (rc != 0) { rccl_fail("ncclAllReduce", rc); return SPLPAK_E_COMM; }
    return 0;
}

void splpak::rccl_comm_free(void *comm)
{
    if (comm && g_rccl.comm_destroy) (void)g_rccl.comm_destroy(comm);
}

extern "C" {

int32_t splpak_plan_set_rccl(splpak_plan *plan, void *nccl_comm, int32_t rank, int32_t world)
{
    if (!plan || !nccl_comm || world < 1 || rank < 0 || rank >= world) { set_error("splpak_plan_set_rccl: bad argument"); return SPLPAK_E_BADARG; }
    if (!rccl_load()) return SPLPAK_E_COMM;
    RcclHook *h = static_cast<RcclHook *>(std::malloc(sizeof(RcclHook)));
    if (!h) return SPLPAK_E_NOMEM;
    h->comm = nccl_comm;
    std::free(plan->ar_owned);
    plan->ar_owned = h;
    // SPLPAK_RCCL_ONE_RANK_CALLS=1 (smoke tests on a one-GPU box): the reductions of a one-rank fit go through RCCL too
    const int always = splpak::opt_get("SPLPAK_RCCL_ONE_RANK_CALLS") ? SPLPAK_AR_ALWAYS : 0;
    return splpak_plan_set_allreduce_ex(plan, rccl_allreduce, h, rank, world, SPLPAK_AR_ANY_POINTER | SPLPAK_AR_STREAM_ORDERED | always);
}

int32_t splpak_rccl_unique_id(char *id128)
{
    if (!id128) { set_error("null argument"); return SPLPAK_E_BADARG; }
    if (!rccl_load()) return SPLPAK_E_COMM;
    UniqueId id;
    const int rc = g_rccl.get_unique_id(&id);
    if (rc != 0) { rccl_fail("ncclGetUniqueId", rc); return SPLPAK_E_COMM; }
    std::memcpy(id128, id.internal, sizeof id.internal);
    return 0;
}

int32_t splpak_rccl_comm_create(const char *id128, int32_t rank, int32_t world, void **comm)
{
    if (!id128 || !comm || world < 1 || rank < 0 || rank >= world) { set_error("splpak_rccl_comm_create: bad argument"); return SPLPAK_E_BADARG; }
    *comm = nullptr;
    if (int r = device_ready()) return r;
    if (!rccl_load()) return SPLPAK_E_COMM;
    UniqueId id;
    std::memcpy(id.internal, id128, sizeof id.internal);
    const int rc = g_rccl.comm_init_rank(comm, world, id, rank);          // the communicator lives on the CURRENT device
    if (rc != 0) { rccl_fail("ncclCommInitRank", rc); *comm = nullptr; return SPLPAK_E_COMM; }
    return 0;
}

// The id file: { "SPLPAKID", tag, publish time [ns, CLOCK_REALTIME], ncclUniqueId } = 152 bytes.  The tag names the JOB: a
// file left behind by another run (rank 0 crashed before it could remove it, or a second job on the same path) carries
// another tag and is ignored, so nobody enters ncclCommInitRank with a stale id -- which would hang, and no timeout of ours
// covers that.  Rank 0 removes whatever lies at `path` before it publishes and removes its own file once ncclCommInitRank
// has returned (the call is collective: every rank has read the id by then).
struct IdFile { char magic[8]; uint64_t tag; int64_t published_ns; char id[128]; };
static_assert(sizeof(IdFile) == 152, "id file layout");

static uint64_t job_tag_of(const char *job)
{
    std::string key;
    if (job && *job) key = job;
    else if (const char *e = splpak::opt_get("SPLPAK_RCCL_JOB")) key = e;
    else {
        // what launchers give every rank of ONE run: torchrun / torch.distributed, Slurm, Open MPI
        for (const char *name : {"TORCHELASTIC_RUN_ID", "MASTER_ADDR", "MASTER_PORT", "SLURM_JOB_ID", "SLURM_STEP_ID", "OMPI_MCA_ess_base_jobid"})
            if (const char *e = splpak::opt_get(name)) { key += name; key += '='; key += e; key += ';'; }
    }
    uint64_t h = 1469598103934665603ull;              // FNV-1a
    for (unsigned char c : key) { h ^= c; h *= 1099511628211ull; }
    return h;
}

int32_t splpak_rccl_comm_create_from_file_ex(const char *path, const char *job, int32_t rank, int32_t world, double timeout_s, void **comm)
{
    if (!path || !comm) { set_error("null argument"); return SPLPAK_E_BADARG; }
    if (world < 1 || rank < 0 || rank >= world) { set_error("splpak_rccl_comm_create_from_file: bad rank / world"); return SPLPAK_E_BADARG; }
    const uint64_t tag = job_tag_of(job);
    const double tmax = timeout_s > 0 ? timeout_s : 60.0;
    IdFile rec;
    if (rank == 0) {
        std::remove(path);                              // a file of an earlier run must not meet this run's readers
        std::memcpy(rec.magic, "SPLPAKID", 8);
        rec.tag = tag;
        rec.published_ns = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::system_clock::now().time_since_epoch()).count();
        if (int r = splpak_rccl_unique_id(rec.id)) return r;
        const std::string tmp = std::string(path) + ".tmp";
        FILE *f = std::fopen(tmp.c_str(), "wb");
        if (!f || std::fwrite(&rec, 1, sizeof rec, f) != sizeof rec) { if (f) std::fclose(f); set_error("RCCL: cannot write the id file"); return SPLPAK_E_COMM; }
        std::fclose(f);
        if (std::rename(tmp.c_str(), path) != 0) { set_error("RCCL: cannot publish the id file"); return SPLPAK_E_COMM; }     // (atomic: readers never see half an id)
    } else {
        const auto t0 = std::chrono::steady_clock::now();
        const int64_t entered_ns = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::system_clock::now().time_since_epoch()).count();
        bool stale_seen = false;
        for (;;) {
            FILE *f = std::fopen(path, "rb");
            if (f) {
                const size_t n = std::fread(&rec, 1, sizeof rec, f);
                std::fclose(f);
                // this job's tag, and not older than the wait we would have granted it ourselves
                const bool fresh = n == sizeof rec && std::memcmp(rec.magic, "SPLPAKID", 8) == 0 && rec.tag == tag &&
                                   rec.published_ns >= entered_ns - (int64_t)(tmax * 1e9);
                if (fresh) break;
                stale_seen = true;
            }
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > tmax) {
                set_error(stale_seen ? "RCCL: timed out waiting for rank 0's id file (the file at this path belongs to another run: wrong job tag or too old)"
                                     : "RCCL: timed out waiting for rank 0's id file");
                return SPLPAK_E_COMM;
            }
            std::this_thread::sleep_for(std::chrono::milliseconds(20));
        }
    }
    const int32_t r = splpak_rccl_comm_create(rec.id, rank, world, comm);
    if (rank == 0) std::remove(path);                   // every rank has the id: ncclCommInitRank is collective
    return r;
}

int32_t splpak_rccl_comm_create_from_file(const char *path, int32_t rank, int32_t world, double timeout_s, void **comm)
{
    return splpak_rccl_comm_create_from_file_ex(path, nullptr, rank, world, timeout_s, comm);
}

void splpak_rccl_comm_destroy(void *comm)
{
    if (comm && rccl_load()) (void)g_rccl.comm_destroy(comm);
}

}  // extern "C"
