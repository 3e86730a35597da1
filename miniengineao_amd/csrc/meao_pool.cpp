// meao_pool.cpp -- native multi-device entry point of the C ABI (include/meao.h, "pool").
//
// The path shards across independent frames only (the reference keeps no temporal
// state, AO.cs:291-308): frame f of a batch goes to pool member f mod G, every member owns a
// complete context (all intermediates) and a stream on its device, and there is no data-path
// exchange.  One host thread calls the pool -- the launches are asynchronous -- so a C# host can
// bind these entry points directly ([DllImport]) instead of running one process per GPU.  Inside, the
// launch sequences of DEVICE batches are enqueued by one worker thread per member: a member's 4-5 launches cost
// 10-14 us of host time, and eight members fed one after the other (111 us per step) cannot keep up with one 4K
// frame per GPU (57-61 us of GPU time: BASELINE config 4, tools/pool_enqueue_cost.py); in parallel they can.
// The pipelined forms of the single-context API exist here too (meao_pool_prefetch_batch,
// meao_pool_composite_enqueue): the in-process host gets the same step the per-GPU processes of bench.py run.
// Results stay on the owning device (or go to host memory); meao_pool_gather_to_device copies them to
// one device (hipMemcpyPeerAsync; device-to-device over xGMI where peer access could be enabled at
// creation, meao_pool_gather_path says which) when a single consumer wants the whole batch.
#include <hip/hip_runtime.h>
#include <pthread.h>
#include <sched.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <exception>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "meao_kernels.hpp"

#ifndef MEAO_TESTING
#define MEAO_TESTING 0      // 1: the `testhooks` variant library (meao_test_* entry points), never the product
#endif

namespace {

// NUMA node of a HIP device and that node's CPUs, from sysfs (numa_node of the device's PCI function; -1 = the kernel does
// not know one: single-node hosts, most VMs).  Eight GPUs of one node hang off two sockets: a thread that enqueues for a GPU
// and waits on its events should run next to it.
struct NumaPlacement {
    int node = -1;
    std::string pci, cpulist;
    cpu_set_t cpus;
    int cpu_count = 0;
};

bool read_line(const std::string &path, std::string *out)
{
    FILE *f = std::fopen(path.c_str(), "r");
    if (!f) return false;
    char buf[4096];
    const bool ok = std::fgets(buf, sizeof buf, f) != nullptr;
    std::fclose(f);
    if (!ok) return false;
    out->assign(buf);
    while (!out->empty() && (out->back() == '\n' || out->back() == ' ')) out->pop_back();
    return true;
}

// "0-15,32-47" -> cpu_set_t
int parse_cpulist(const std::string &list, cpu_set_t *set)
{
    CPU_ZERO(set);
    int count = 0;
    size_t i = 0;
    while (i < list.size()) {
        char *end = nullptr;
        const long a = std::strtol(list.c_str() + i, &end, 10);
        if (end == list.c_str() + i) break;
        long b = a;
        i = static_cast<size_t>(end - list.c_str());
        if (i < list.size() && list[i] == '-') {
            b = std::strtol(list.c_str() + i + 1, &end, 10);
            i = static_cast<size_t>(end - list.c_str());
        }
        for (long c = a; c <= b && c < CPU_SETSIZE; ++c) {
            if (c >= 0) { CPU_SET(static_cast<int>(c), set); ++count; }
        }
        if (i < list.size() && list[i] == ',') ++i;
    }
    return count;
}

NumaPlacement numa_placement_of(int device)
{
    NumaPlacement np;
    CPU_ZERO(&np.cpus);
    char pci[64] = {};
    if (hipDeviceGetPCIBusId(pci, sizeof pci, device) != hipSuccess) { (void)hipGetLastError(); return np; }
    np.pci = pci;
    for (char &c : np.pci) c = static_cast<char>(std::tolower(static_cast<unsigned char>(c)));
    std::string node;
    if (!read_line("/sys/bus/pci/devices/" + np.pci + "/numa_node", &node)) return np;
    np.node = std::atoi(node.c_str());
    if (np.node < 0) return np;
    if (read_line("/sys/devices/system/node/node" + std::to_string(np.node) + "/cpulist", &np.cpulist))
        np.cpu_count = parse_cpulist(np.cpulist, &np.cpus);
    return np;
}

// Bind the calling thread to the CPUs of `np` that it is allowed to run on (cgroup / taskset masks are respected); false =
// nothing to bind to (unknown node, or no allowed CPU on it) and the thread keeps its mask.
bool bind_this_thread(const NumaPlacement &np)
{
    if (np.node < 0 || np.cpu_count == 0) return false;
    cpu_set_t allowed, target;
    if (pthread_getaffinity_np(pthread_self(), sizeof allowed, &allowed) != 0) return false;
    CPU_AND(&target, &allowed, &np.cpus);
    if (CPU_COUNT(&target) == 0) return false;
    return pthread_setaffinity_np(pthread_self(), sizeof target, &target) == 0;
}

}  // namespace

// One worker per member: runs the member's share of a pool call on the member's device.  The caller posts a job to
// every worker and waits for all of them, so a context is only ever touched by one thread at a time.  A worker spins for
// a while after its last job (a stream of steps keeps it hot: a condition-variable wake-up costs as much as the job),
// then sleeps.
struct PoolWorker {
    std::thread thread;
    std::mutex mu;
    std::condition_variable cv;
    std::atomic<int> state{0};            // 0 idle, 1 job posted, 2 job done, 3 quit
    std::function<int32_t()> job;
    int32_t rc = 0;
    int32_t device = 0;
    NumaPlacement numa;                   // of `device`; the worker binds itself to it when bind_numa is set
    bool bind_numa = true;
    std::atomic<int> bound{0};            // 1 once the thread runs on its device's node
    std::atomic<int> spin_us{100};        // MEAO_POOL_SPIN_US
    void loop()
    {
        (void)hipSetDevice(device);
        if (bind_numa && bind_this_thread(numa)) bound.store(1, std::memory_order_release);
        for (;;) {
            // (a stream of 4K steps posts a job every ~60 us per member; 100 us of spinning covers that and gives the core
            // back soon after the stream ends -- the host may be running CPU work next to the pool, ADVICE r4.  MEAO_POOL_SPIN_US)
            const auto spin_until = std::chrono::steady_clock::now() + std::chrono::microseconds(spin_us.load(std::memory_order_relaxed));
            int st;
            while ((st = state.load(std::memory_order_acquire)) != 1 && st != 3) {
                if (std::chrono::steady_clock::now() > spin_until) {
                    std::unique_lock<std::mutex> lock(mu);
                    cv.wait(lock, [&] { const int v = state.load(std::memory_order_acquire); return v == 1 || v == 3; });
                }
            }
            if (st == 3) return;
            rc = job();
            state.store(2, std::memory_order_release);
        }
    }
    void post(std::function<int32_t()> fn)
    {
        job = std::move(fn);
        {
            std::lock_guard<std::mutex> lock(mu);      // pairs with the predicate check of a worker about to sleep
            state.store(1, std::memory_order_release);
        }
        cv.notify_one();
    }
    int32_t wait()
    {
        while (state.load(std::memory_order_acquire) != 2) std::this_thread::yield();
        state.store(0, std::memory_order_release);
        return rc;
    }
    void quit()
    {
        {
            std::lock_guard<std::mutex> lock(mu);
            state.store(3, std::memory_order_release);
        }
        cv.notify_one();
        if (thread.joinable()) thread.join();
    }
};

struct meao_pool {
    std::vector<std::unique_ptr<PoolWorker>> worker;      // started by the first DEVICE batch of a pool with several members
    bool workers_unavailable = false;                     // thread creation failed once: this pool enqueues serially
    std::vector<meao_ctx *> ctx;
    std::vector<int32_t> device;
    std::vector<hipStream_t> stream;
    std::vector<int32_t> peer_ok;      // [member * device_count + dst_device]: peer access from the member's device enabled
    int32_t device_count = 0;
    int32_t max_batch = 1;
    uint64_t out_bytes = 0;
    int32_t spin_us = 100;             // meao_pool_configure
    bool bind_numa = true;
    std::vector<NumaPlacement> numa;   // per member: where its device hangs
#if MEAO_TESTING
    bool refuse_peer = false;          // meao_test_pool_refuse_peer: every gather copy goes through hipMemcpyPeerAsync as if no link were enabled
#endif
    std::string err;
};

namespace {

thread_local std::string g_pool_error;

int pool_fail(meao_pool *p, int status, const std::string &msg)
{
    if (p) p->err = msg;
    g_pool_error = msg;
    return status;
}

// Pool entry points switch devices; the calling thread's current device is put back on every exit path
// (an in-process host -- torch, the C# component -- has its own idea of the current device).
struct DeviceGuard {
    int prev = -1;
    DeviceGuard() { if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; } }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

// frames of member m in a batch of n dealt round-robin over G members
template <typename T>
int32_t share_of(int32_t m, int32_t G, int32_t n, T *const *all, T **mine)
{
    int32_t k = 0;
    for (int32_t f = m; f < n; f += G) mine[k++] = all[f];
    return k;
}

// fn(m) for every member: on the members' worker threads (all at once) when the pool has several members, else here.
// Returns the first failing member's status (every member is run and waited for either way).
int32_t for_each_member(meao_pool *p, const std::function<int32_t(int32_t)> &fn, bool threaded, int32_t *failed_member)
{
    const int32_t G = static_cast<int32_t>(p->ctx.size());
    int32_t status = MEAO_OK;
    *failed_member = -1;
    if (!threaded || G < 2) {
        for (int32_t m = 0; m < G; ++m) {
            const int32_t rc = fn(m);
            if (rc != MEAO_OK && status == MEAO_OK) { status = rc; *failed_member = m; }
        }
        return status;
    }
    if (p->worker.empty() && !p->workers_unavailable) {
        try {
            for (int32_t m = 0; m < G; ++m) {
                p->worker.emplace_back(new PoolWorker());
                p->worker.back()->device = p->device[m];
                p->worker.back()->numa = p->numa[m];
                p->worker.back()->bind_numa = p->bind_numa;
                p->worker.back()->spin_us.store(p->spin_us, std::memory_order_relaxed);
                PoolWorker *w = p->worker.back().get();
                w->thread = std::thread([w] { w->loop(); });
            }
        } catch (const std::exception &) {     // std::system_error (no thread to be had) / bad_alloc: keep working, serially
            for (auto &w : p->worker) w->quit();
            p->worker.clear();
            p->workers_unavailable = true;
        }
    }
    if (p->worker.empty()) {
        for (int32_t m = 0; m < G; ++m) {
            const int32_t rc = fn(m);
            if (rc != MEAO_OK && status == MEAO_OK) { status = rc; *failed_member = m; }
        }
        return status;
    }
    for (int32_t m = 0; m < G; ++m) p->worker[m]->post([&fn, m] { return fn(m); });
    for (int32_t m = 0; m < G; ++m) {
        const int32_t rc = p->worker[m]->wait();
        if (rc != MEAO_OK && status == MEAO_OK) { status = rc; *failed_member = m; }
    }
    return status;
}

}  // namespace

extern "C" {

int32_t meao_pool_create(const meao_config *cfg, const int32_t *devices, int32_t num_devices, meao_pool **out_pool)
{
    if (out_pool) *out_pool = nullptr;
    if (!cfg || !out_pool || num_devices < 1 || num_devices > 64)
        return pool_fail(nullptr, MEAO_ERR_INVALID_ARGUMENT, "meao_pool_create: bad argument (1..64 members)");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        (void)hipGetLastError();
        return pool_fail(nullptr, MEAO_ERR_NO_DEVICE, "meao_pool_create: no HIP device visible (no CPU fallback exists)");
    }
    DeviceGuard guard;
    meao_pool *p = new (std::nothrow) meao_pool();
    if (!p) return pool_fail(nullptr, MEAO_ERR_OUT_OF_MEMORY, "meao_pool_create: host allocation failed");
    p->max_batch = cfg->max_batch;
    p->device_count = count;
    p->out_bytes = static_cast<uint64_t>(cfg->width) * cfg->height * (cfg->ao_format == MEAO_AO_R8 ? 1 : 2);
    for (int32_t i = 0; i < num_devices; ++i) {
        // devices == NULL: members 0..n-1 on devices 0..n-1 (wrapping, so a 1-GPU box can host several members)
        const int32_t dev = devices ? devices[i] : i % count;
        meao_config c = *cfg;
        c.device = dev;
        meao_ctx *ctx = nullptr;
        int32_t rc = meao_create(&c, &ctx);
        std::string why = rc == MEAO_OK ? std::string() : std::string(meao_last_error(nullptr));
        hipStream_t s = nullptr;
        if (rc == MEAO_OK) {
            hipError_t e = hipSetDevice(dev);
            if (e == hipSuccess) e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
            if (e != hipSuccess) {
                (void)hipGetLastError();
                rc = MEAO_ERR_HIP;
                why = std::string("stream creation: ") + hipGetErrorString(e);
            }
        }
        if (rc != MEAO_OK) {
            if (ctx) meao_destroy(ctx);
            meao_pool_destroy(p);
            return pool_fail(nullptr, rc, std::string("meao_pool_create: member ") + std::to_string(i) + " on device " +
                                              std::to_string(dev) + ": " + why);
        }
        p->ctx.push_back(ctx);
        p->device.push_back(dev);
        p->stream.push_back(s);
        p->numa.push_back(numa_placement_of(dev));
    }
    // Peer access, both ways, between every pair of distinct member devices: without it hipMemcpyPeerAsync
    // still works but bounces through host memory.  "Already enabled" (another pool, the host itself) is fine.
    p->peer_ok.assign(p->ctx.size() * static_cast<size_t>(count), 0);
    for (size_t m = 0; m < p->ctx.size(); ++m) {
        const int32_t from = p->device[m];
        for (int32_t to = 0; to < count; ++to) {
            if (to == from) continue;
            bool member_device = false;
            for (int32_t d : p->device) member_device = member_device || d == to;
            if (!member_device) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, from, to) != hipSuccess) { (void)hipGetLastError(); can = 0; }
            if (!can) continue;
            if (hipSetDevice(from) != hipSuccess) { (void)hipGetLastError(); continue; }
            const hipError_t e = hipDeviceEnablePeerAccess(to, 0);
            if (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled) p->peer_ok[m * count + to] = 1;
            (void)hipGetLastError();
        }
    }
    *out_pool = p;
    return MEAO_OK;
}

int32_t meao_pool_destroy(meao_pool *p)
{
    if (!p) return MEAO_OK;
    DeviceGuard guard;
    for (auto &w : p->worker) w->quit();
    p->worker.clear();
    for (size_t i = 0; i < p->ctx.size(); ++i) {
        (void)hipSetDevice(p->device[i]);
        // The member's context has this stream as the stream of its last call (launches in flight,
        // meao_destroy synchronises them): the context goes first, its stream after it.
        if (p->stream[i]) (void)hipStreamSynchronize(p->stream[i]);
        meao_destroy(p->ctx[i]);
        (void)hipSetDevice(p->device[i]);
        if (p->stream[i]) (void)hipStreamDestroy(p->stream[i]);
    }
    delete p;
    return MEAO_OK;
}

int32_t meao_pool_size(const meao_pool *p) { return p ? static_cast<int32_t>(p->ctx.size()) : 0; }

meao_ctx *meao_pool_context(meao_pool *p, int32_t member)
{
    return (p && member >= 0 && member < static_cast<int32_t>(p->ctx.size())) ? p->ctx[member] : nullptr;
}

int32_t meao_pool_device_of_frame(const meao_pool *p, int32_t frame)
{
    if (!p || frame < 0 || p->ctx.empty()) return -1;
    return p->device[frame % p->ctx.size()];
}

const char *meao_pool_last_error(const meao_pool *p) { return p ? p->err.c_str() : g_pool_error.c_str(); }

int32_t meao_pool_set_params(meao_pool *p, const meao_params *prm)
{
    if (!p || !prm) return MEAO_ERR_INVALID_ARGUMENT;
    DeviceGuard guard;
    for (size_t i = 0; i < p->ctx.size(); ++i) {
        const int32_t rc = meao_set_params(p->ctx[i], prm);
        if (rc != MEAO_OK) return pool_fail(p, rc, std::string("meao_pool_set_params: ") + meao_last_error(p->ctx[i]));
    }
    return MEAO_OK;
}

int32_t meao_pool_execute_batch(meao_pool *p, int32_t n, const void *const *depth, int32_t depth_loc,
                                void *const *ao_out, int32_t out_loc)
{
    if (!p || !depth || !ao_out) return MEAO_ERR_INVALID_ARGUMENT;
    const int32_t G = static_cast<int32_t>(p->ctx.size());
    if (n < 1 || n > p->max_batch * G)
        return pool_fail(p, MEAO_ERR_INVALID_ARGUMENT, "meao_pool_execute_batch: n must be 1..max_batch * members");
    DeviceGuard guard;
    // frame f -> member f mod G; each member runs its share as ONE batched launch
    // sequence on its own stream.  With HOST memory every member's staged copies and launches are put in
    // flight first and all members are waited for afterwards: member m+1's upload overlaps member m's kernels.
    const bool host = depth_loc == MEAO_MEM_HOST || out_loc == MEAO_MEM_HOST;
    int32_t failed = -1;
    // DEVICE batches: every member's launch sequence is enqueued by its own worker thread (see the top of this file).
    // HOST batches are dominated by their staged copies and stay on the calling thread.
    int32_t status = for_each_member(p, [&](int32_t m) -> int32_t {
        const void *d[MEAO_MAX_BATCH];
        void *o[MEAO_MAX_BATCH];
        const int32_t k = share_of(m, G, n, depth, d);
        share_of(m, G, n, ao_out, o);
        if (k == 0) return MEAO_OK;
        return meao::execute_batch_internal(p->ctx[m], k, d, depth_loc, o, out_loc, p->stream[m], false);
    }, !host, &failed);
    if (status != MEAO_OK)
        status = pool_fail(p, status, std::string("meao_pool_execute_batch: member ") + std::to_string(failed) + ": " +
                                          meao_last_error(p->ctx[failed]));
    if (host) {     // the host buffers are the caller's again when the call returns: wait for every member, also one that
                    // failed -- it may have put copies from / to the caller's buffers in flight before it failed
        for (int32_t m = 0; m < G; ++m)
            if (hipSetDevice(p->device[m]) != hipSuccess || hipStreamSynchronize(p->stream[m]) != hipSuccess) {
                (void)hipGetLastError();
                if (status == MEAO_OK) status = pool_fail(p, MEAO_ERR_HIP, "meao_pool_execute_batch: stream synchronisation failed");
            }
    }
    return status;
}

int32_t meao_pool_prefetch_batch(meao_pool *p, int32_t n, const void *const *depth)
{
    if (!p || !depth) return MEAO_ERR_INVALID_ARGUMENT;
    const int32_t G = static_cast<int32_t>(p->ctx.size());
    if (n < 1 || n > p->max_batch * G)
        return pool_fail(p, MEAO_ERR_INVALID_ARGUMENT, "meao_pool_prefetch_batch: n must be 1..max_batch * members");
    DeviceGuard guard;
    int32_t failed = -1;
    bool set_device_failed = false;
    // (a context with cfg.pipelined = 1 never allocates or synchronises here: this is bookkeeping, cheaper than a hand-over
    // to the workers -- and the first call of a context created without it re-allocates, which must not race anything)
    const int32_t status = for_each_member(p, [&](int32_t m) -> int32_t {
        const void *d[MEAO_MAX_BATCH];
        const int32_t k = share_of(m, G, n, depth, d);
        if (k == 0) return MEAO_OK;
        if (hipSetDevice(p->device[m]) != hipSuccess) { (void)hipGetLastError(); set_device_failed = true; return MEAO_ERR_HIP; }
        return meao_prefetch_batch(p->ctx[m], k, d);
    }, false, &failed);
    if (status != MEAO_OK)       // (a failed hipSetDevice never reached the context: its last error would name an older failure)
        return pool_fail(p, status, std::string("meao_pool_prefetch_batch: member ") + std::to_string(failed) + ": " +
                                        (set_device_failed ? "hipSetDevice(" + std::to_string(p->device[failed]) + ") failed"
                                                           : std::string(meao_last_error(p->ctx[failed]))));
    return MEAO_OK;
}

int32_t meao_pool_composite_enqueue(meao_pool *p, int32_t mode, int32_t n, const void *const *ao, void *const *color_rgba16f,
                                    void *const *gbuffer0_rgba8)
{
    if (!p || !ao || !color_rgba16f) return MEAO_ERR_INVALID_ARGUMENT;
    const int32_t G = static_cast<int32_t>(p->ctx.size());
    if (n < 1 || n > p->max_batch * G)
        return pool_fail(p, MEAO_ERR_INVALID_ARGUMENT, "meao_pool_composite_enqueue: n must be 1..max_batch * members");
    DeviceGuard guard;
    for (int32_t m = 0; m < G; ++m) {
        const void *a[MEAO_MAX_BATCH];
        void *c[MEAO_MAX_BATCH], *g[MEAO_MAX_BATCH];
        const int32_t k = share_of(m, G, n, ao, a);
        share_of(m, G, n, color_rgba16f, c);
        if (gbuffer0_rgba8) share_of(m, G, n, gbuffer0_rgba8, g);
        if (k == 0) continue;
        const int32_t rc = meao_composite_enqueue(p->ctx[m], mode, k, a, c, gbuffer0_rgba8 ? g : nullptr);
        if (rc != MEAO_OK)
            return pool_fail(p, rc, std::string("meao_pool_composite_enqueue: member ") + std::to_string(m) + ": " +
                                        meao_last_error(p->ctx[m]));
    }
    return MEAO_OK;
}

int32_t meao_pool_composite_pending(const meao_pool *p, int32_t *out_frames)
{
    if (!p || !out_frames) return MEAO_ERR_INVALID_ARGUMENT;
    int32_t total = 0;
    for (size_t m = 0; m < p->ctx.size(); ++m) {
        int32_t k = 0;
        const int32_t rc = meao_composite_pending(p->ctx[m], &k);
        if (rc != MEAO_OK) return rc;
        total += k;
    }
    *out_frames = total;
    return MEAO_OK;
}

int32_t meao_pool_composite_flush(meao_pool *p)
{
    if (!p) return MEAO_ERR_INVALID_ARGUMENT;
    DeviceGuard guard;
    for (size_t m = 0; m < p->ctx.size(); ++m) {
        const int32_t rc = meao_composite_flush(p->ctx[m], p->stream[m]);
        if (rc != MEAO_OK)
            return pool_fail(p, rc, std::string("meao_pool_composite_flush: member ") + std::to_string(m) + ": " +
                                        meao_last_error(p->ctx[m]));
    }
    return MEAO_OK;
}

int32_t meao_pool_gather_path(const meao_pool *p, int32_t member, int32_t dst_device)
{
    if (!p || member < 0 || member >= static_cast<int32_t>(p->ctx.size()) || dst_device < 0 || dst_device >= p->device_count)
        return MEAO_ERR_INVALID_ARGUMENT;
#if MEAO_TESTING
    if (p->refuse_peer) return MEAO_POOL_PATH_STAGED;
#endif
    if (p->device[member] == dst_device) return MEAO_POOL_PATH_SAME_DEVICE;
    return p->peer_ok[static_cast<size_t>(member) * p->device_count + dst_device] ? MEAO_POOL_PATH_PEER_DIRECT : MEAO_POOL_PATH_STAGED;
}

int32_t meao_pool_gather_to_device(meao_pool *p, int32_t n, const void *const *ao_src, void *const *dst, int32_t dst_device)
{
    if (!p || !ao_src || !dst || n < 1) return MEAO_ERR_INVALID_ARGUMENT;
    if (dst_device < 0 || dst_device >= p->device_count)
        return pool_fail(p, MEAO_ERR_INVALID_ARGUMENT, "meao_pool_gather_to_device: dst_device out of range");
    const int32_t G = static_cast<int32_t>(p->ctx.size());
    DeviceGuard guard;
    for (int32_t f = 0; f < n; ++f) {
        const int32_t m = f % G;
        if (hipSetDevice(p->device[m]) != hipSuccess) return pool_fail(p, MEAO_ERR_HIP, "meao_pool_gather_to_device: hipSetDevice");
        // on the producing member's stream: ordered behind the kernels that wrote the frame
        bool same_device = p->device[m] == dst_device;
#if MEAO_TESTING
        if (p->refuse_peer) same_device = false;       // the cross-device call, on whatever devices there are (one-GPU boxes)
#endif
        const hipError_t e = same_device
                                 ? hipMemcpyAsync(dst[f], ao_src[f], p->out_bytes, hipMemcpyDeviceToDevice, p->stream[m])
                                 : hipMemcpyPeerAsync(dst[f], dst_device, ao_src[f], p->device[m], p->out_bytes, p->stream[m]);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            return pool_fail(p, MEAO_ERR_HIP, std::string("meao_pool_gather_to_device: ") + hipGetErrorString(e));
        }
    }
    return MEAO_OK;
}

int32_t meao_pool_synchronize(meao_pool *p)
{
    if (!p) return MEAO_ERR_INVALID_ARGUMENT;
    DeviceGuard guard;
    for (size_t i = 0; i < p->ctx.size(); ++i) {
        if (hipSetDevice(p->device[i]) != hipSuccess || meao_synchronize(p->ctx[i], p->stream[i]) != MEAO_OK) {
            (void)hipGetLastError();
            return pool_fail(p, MEAO_ERR_HIP, "meao_pool_synchronize: stream synchronisation failed");
        }
    }
    return MEAO_OK;
}

int32_t meao_device_numa_node(int32_t device, int32_t *out_node, char *cpulist, uint64_t cpulist_capacity)
{
    if (!out_node) return MEAO_ERR_INVALID_ARGUMENT;
    *out_node = -1;
    if (cpulist && cpulist_capacity > 0) cpulist[0] = 0;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) { (void)hipGetLastError(); return MEAO_ERR_NO_DEVICE; }
    if (device < 0 || device >= count) return MEAO_ERR_INVALID_ARGUMENT;
    const NumaPlacement np = numa_placement_of(device);
    *out_node = np.node;
    if (cpulist && cpulist_capacity > 0) {
        if (np.cpulist.size() + 1 > cpulist_capacity) return MEAO_ERR_BUFFER_TOO_SMALL;
        std::memcpy(cpulist, np.cpulist.c_str(), np.cpulist.size() + 1);
    }
    return MEAO_OK;
}

int32_t meao_pool_configure(meao_pool *p, int32_t key, int32_t value)
{
    if (!p) return MEAO_ERR_INVALID_ARGUMENT;
    switch (key) {
    case MEAO_POOL_SPIN_US:
        if (value < 0 || value > 1000000) return pool_fail(p, MEAO_ERR_INVALID_ARGUMENT, "meao_pool_configure: SPIN_US is 0..1000000");
        p->spin_us = value;
        for (auto &w : p->worker) w->spin_us.store(value, std::memory_order_relaxed);
        return MEAO_OK;
    case MEAO_POOL_BIND_NUMA:
        if (!p->worker.empty()) return pool_fail(p, MEAO_ERR_UNSUPPORTED, "meao_pool_configure: BIND_NUMA must be set before the first DEVICE batch (the workers are running)");
        p->bind_numa = value != 0;
        return MEAO_OK;
    default: return pool_fail(p, MEAO_ERR_INVALID_ARGUMENT, "meao_pool_configure: unknown key");
    }
}

int32_t meao_pool_member_placement(const meao_pool *p, int32_t member, int32_t *out_numa_node, int32_t *out_worker_bound)
{
    if (!p || member < 0 || member >= static_cast<int32_t>(p->ctx.size())) return MEAO_ERR_INVALID_ARGUMENT;
    if (out_numa_node) *out_numa_node = p->numa[member].node;
    if (out_worker_bound)
        *out_worker_bound = member < static_cast<int32_t>(p->worker.size()) ? p->worker[member]->bound.load(std::memory_order_acquire) : 0;
    return MEAO_OK;
}

#if MEAO_TESTING
// testhooks variant only: every later meao_pool_gather_to_device copy takes the cross-device call (hipMemcpyPeerAsync) and
// meao_pool_gather_path reports STAGED, as on a node whose devices offer no peer access -- testable on a one-GPU box.
__attribute__((visibility("default"))) int32_t meao_test_pool_refuse_peer(meao_pool *p, int32_t refuse)
{
    if (!p) return MEAO_ERR_INVALID_ARGUMENT;
    p->refuse_peer = refuse != 0;
    return MEAO_OK;
}
#endif

}  // extern "C"
