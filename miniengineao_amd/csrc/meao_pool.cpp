// meao_pool.cpp -- native multi-device entry point of the C ABI (include/meao.h, "pool").
//
// The path shards across independent frames only (the reference keeps no temporal
// state, AO.cs:291-308): frame f of a batch goes to pool member f mod G, every member owns a
// complete context (all intermediates) and a stream on its device, and there is no data-path
// exchange.  One host thread drives all members -- the launches are asynchronous -- so a C# host can
// bind these entry points directly ([DllImport]) instead of running one process per GPU.
// Results stay on the owning device (or go to host memory); meao_pool_gather_to_device copies them to
// one device over xGMI (hipMemcpyPeerAsync) when a single consumer wants the whole batch.
#include <hip/hip_runtime.h>

#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/meao.h"

struct meao_pool {
    std::vector<meao_ctx *> ctx;
    std::vector<int32_t> device;
    std::vector<hipStream_t> stream;
    int32_t max_batch = 1;
    uint64_t out_bytes = 0;
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
    meao_pool *p = new (std::nothrow) meao_pool();
    if (!p) return pool_fail(nullptr, MEAO_ERR_OUT_OF_MEMORY, "meao_pool_create: host allocation failed");
    p->max_batch = cfg->max_batch;
    p->out_bytes = static_cast<uint64_t>(cfg->width) * cfg->height * (cfg->ao_format == MEAO_AO_R8 ? 1 : 2);
    for (int32_t i = 0; i < num_devices; ++i) {
        // devices == NULL: members 0..n-1 on devices 0..n-1 (wrapping, so a 1-GPU box can host several members)
        const int32_t dev = devices ? devices[i] : i % count;
        meao_config c = *cfg;
        c.device = dev;
        meao_ctx *ctx = nullptr;
        int32_t rc = meao_create(&c, &ctx);
        hipStream_t s = nullptr;
        if (rc == MEAO_OK && (hipSetDevice(dev) != hipSuccess || hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess)) {
            (void)hipGetLastError();
            rc = MEAO_ERR_HIP;
        }
        if (rc != MEAO_OK) {
            const std::string why = std::string("meao_pool_create: member ") + std::to_string(i) + " on device " +
                                    std::to_string(dev) + ": " + meao_last_error(ctx);
            if (ctx) meao_destroy(ctx);
            meao_pool_destroy(p);
            return pool_fail(nullptr, rc, why);
        }
        p->ctx.push_back(ctx);
        p->device.push_back(dev);
        p->stream.push_back(s);
    }
    *out_pool = p;
    return MEAO_OK;
}

int32_t meao_pool_destroy(meao_pool *p)
{
    if (!p) return MEAO_OK;
    for (size_t i = 0; i < p->ctx.size(); ++i) {
        (void)hipSetDevice(p->device[i]);
        if (p->stream[i]) {
            (void)hipStreamSynchronize(p->stream[i]);
            (void)hipStreamDestroy(p->stream[i]);
        }
        meao_destroy(p->ctx[i]);
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
    // frame f -> member f mod G; each member runs its share as ONE batched launch
    // sequence on its own stream.  With HOST memory a member's call returns when its copies are done, so
    // members are then served one after the other (device-resident frames overlap across members).
    for (int32_t m = 0; m < G; ++m) {
        const void *d[MEAO_MAX_BATCH];
        void *o[MEAO_MAX_BATCH];
        int32_t k = 0;
        for (int32_t f = m; f < n; f += G, ++k) {
            d[k] = depth[f];
            o[k] = ao_out[f];
        }
        if (k == 0) continue;
        const int32_t rc = meao_execute_batch(p->ctx[m], k, d, depth_loc, o, out_loc, p->stream[m]);
        if (rc != MEAO_OK)
            return pool_fail(p, rc, std::string("meao_pool_execute_batch: member ") + std::to_string(m) + ": " +
                                        meao_last_error(p->ctx[m]));
    }
    return MEAO_OK;
}

int32_t meao_pool_gather_to_device(meao_pool *p, int32_t n, const void *const *ao_src, void *const *dst, int32_t dst_device)
{
    if (!p || !ao_src || !dst || n < 1) return MEAO_ERR_INVALID_ARGUMENT;
    const int32_t G = static_cast<int32_t>(p->ctx.size());
    for (int32_t f = 0; f < n; ++f) {
        const int32_t m = f % G;
        if (hipSetDevice(p->device[m]) != hipSuccess) return pool_fail(p, MEAO_ERR_HIP, "meao_pool_gather_to_device: hipSetDevice");
        // on the producing member's stream: ordered behind the kernels that wrote the frame
        const hipError_t e = hipMemcpyPeerAsync(dst[f], dst_device, ao_src[f], p->device[m], p->out_bytes, p->stream[m]);
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
    for (size_t i = 0; i < p->ctx.size(); ++i) {
        if (hipSetDevice(p->device[i]) != hipSuccess || hipStreamSynchronize(p->stream[i]) != hipSuccess) {
            (void)hipGetLastError();
            return pool_fail(p, MEAO_ERR_HIP, "meao_pool_synchronize: stream synchronisation failed");
        }
    }
    return MEAO_OK;
}

}  // extern "C"
