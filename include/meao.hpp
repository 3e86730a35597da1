// meao.hpp -- header-only C++ host side over the C ABI of libmeao_hip.so (include/meao.h).
//
// The reference's host side is a compiled C# component, MiniEngineAO.AmbientOcclusion
// (Assets/MiniEngineAO/AmbientOcclusion.cs, "AO.cs").  No C# toolchain exists in the build
// image, so the compiled-language host mirror is this C++ class (the C# P/Invoke source a
// maintainer would add is bindings/csharp/).  Same six public properties, same names and
// defaults (AO.cs:20-68); camera terms, depth input and AO output are explicit.
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "meao.h"

namespace MiniEngineAO {

class Error : public std::runtime_error {
public:
    Error(int32_t status, const std::string &what) : std::runtime_error(what), status_(status) {}
    int32_t status() const { return status_; }

private:
    int32_t status_;
};

class AmbientOcclusion {
public:
    // ---- Exposed properties (AO.cs:20-68) ---------------------------------------------------
    float noiseFilterTolerance = 0.0f;   // Range(-8, 0)
    float blurTolerance = -4.6f;         // Range(-8, -1)
    float upsampleTolerance = -12.0f;    // Range(-12, -1)
    float thicknessModifier = 1.0f;      // Range(1, 10)
    float intensity = 1.0f;              // Range(0, 2)
    bool ambientOnly = true;             // composite-side flag (AO.cs:62-68)

    // ---- camera terms Unity supplied implicitly (AO.cs:563-573) -----------------------------
    float nearClipPlane = 0.3f;
    float farClipPlane = 1000.0f;
    float projection00 = 0.9742786f;     // camera.projectionMatrix[0,0]
    bool usesReversedZBuffer = true;
    bool singlePassStereoEnabled = false;   // AO.cs:392-401: pixelWidth is then the double-wide eye pair

    // hqLevels / sampleSet: variants the reference's shaders carry but its host never dispatches
    // (Render.main wide + Upsample.main_premin*, SAMPLE_EXHAUSTIVELY); 0 = the reference's wiring.
    AmbientOcclusion(int32_t pixelWidth, int32_t pixelHeight, int32_t device = 0,
                     meao_ao_format aoFormat = MEAO_AO_R8, int32_t maxBatch = 1, int32_t numLevels = 4,
                     meao_f16_rounding f16Rounding = MEAO_F16_RTZ_CLAMP,
                     meao_depth_format depthFormat = MEAO_DEPTH_F32, int32_t hqLevels = 0,
                     meao_sample_set sampleSet = MEAO_SAMPLES_CHECKER)
    {
        meao_default_config(&cfg_);
        cfg_.device = device;
        cfg_.width = pixelWidth;
        cfg_.height = pixelHeight;
        cfg_.ao_format = aoFormat;
        cfg_.max_batch = maxBatch;
        cfg_.num_levels = numLevels;
        cfg_.f16_rounding = f16Rounding;
        cfg_.depth_format = depthFormat;
        cfg_.hq_levels = hqLevels;
        cfg_.sample_set = sampleSet;
        const int32_t rc = meao_create(&cfg_, &ctx_);
        if (rc != MEAO_OK) throw Error(rc, meao_last_error(nullptr));
    }
    AmbientOcclusion(const AmbientOcclusion &) = delete;
    AmbientOcclusion &operator=(const AmbientOcclusion &) = delete;
    ~AmbientOcclusion() { meao_destroy(ctx_); }   // OnDestroy (AO.cs:357-381); discards a composite still waiting (FlushComposite first)

    int32_t width() const { return cfg_.width; }
    int32_t height() const { return cfg_.height; }
    size_t aoTexelBytes() const { return cfg_.ao_format == MEAO_AO_R8 ? 1 : 2; }

    // Device-resident depth in -> AO texture out, asynchronous on `stream`
    // (replays what RebuildCommandBuffers records, AO.cs:496-531).
    void Render(const void *deviceDepth, void *deviceAo, meao_stream stream = nullptr)
    {
        sync();
        check(meao_execute(ctx_, deviceDepth, MEAO_MEM_DEVICE, deviceAo, MEAO_MEM_DEVICE, stream));
    }

    void RenderBatch(const std::vector<const void *> &deviceDepth, const std::vector<void *> &deviceAo,
                     meao_stream stream = nullptr)
    {
        sync();
        check(meao_execute_batch(ctx_, static_cast<int32_t>(deviceDepth.size()), deviceDepth.data(),
                                 MEAO_MEM_DEVICE, deviceAo.data(), MEAO_MEM_DEVICE, stream));
    }

    // Host arrays in and out (synchronous).
    void RenderHost(const void *depth, void *ao)
    {
        sync();
        check(meao_execute(ctx_, depth, MEAO_MEM_HOST, ao, MEAO_MEM_HOST, nullptr));
    }

    void Resize(int32_t pixelWidth, int32_t pixelHeight)   // screen-size change (AO.cs:338-341)
    {
        check(meao_resize(ctx_, pixelWidth, pixelHeight));
        cfg_.width = pixelWidth;
        cfg_.height = pixelHeight;
    }

    // Streams of frames: announce the device depth frames of the call after next; the next Render*
    // carries their downsample pass inside its last kernel (meao_prefetch_batch).
    void PrefetchBatch(const std::vector<const void *> &nextDeviceDepth)
    {
        sync();   // pending property changes first: meao_set_params would drop the announcement
        check(meao_prefetch_batch(ctx_, static_cast<int32_t>(nextDeviceDepth.size()), nextDeviceDepth.data()));
    }

    void Synchronize(meao_stream stream = nullptr) { check(meao_synchronize(ctx_, stream)); }

    // PushCompositeCommands (AO.cs:822-839).  Composite(): now, on `stream`.  CompositeWithNextFrame(): the
    // composite of device frames this component produced rides inside the render kernel of the next
    // Render* call (meao_composite_enqueue); FlushComposite() runs whatever still waits.
    void Composite(meao_composite_mode mode, const void *deviceAo, void *deviceColorRgba16f, void *deviceGBuffer0 = nullptr,
                   meao_stream stream = nullptr)
    {
        check(meao_composite(ctx_, mode, deviceAo, deviceColorRgba16f, deviceGBuffer0, MEAO_MEM_DEVICE, stream));
    }
    void CompositeWithNextFrame(meao_composite_mode mode, const std::vector<const void *> &deviceAo,
                                const std::vector<void *> &deviceColorRgba16f, const std::vector<void *> &deviceGBuffer0 = {})
    {
        check(meao_composite_enqueue(ctx_, mode, static_cast<int32_t>(deviceAo.size()), deviceAo.data(), deviceColorRgba16f.data(),
                                     deviceGBuffer0.empty() ? nullptr : deviceGBuffer0.data()));
    }
    void FlushComposite(meao_stream stream = nullptr) { check(meao_composite_flush(ctx_, stream)); }
    bool CompositePending()            // a batch given to CompositeWithNextFrame that no Render* / flush / resize has run yet
    {
        int32_t frames = 0;
        check(meao_composite_pending(ctx_, &frames));
        return frames > 0;
    }

    // roctx ranges per pass for rocprofv3 --marker-trace
    void SetTracing(bool enable) { check(meao_set_tracing(ctx_, enable ? 1 : 0)); }

    // The _debug 1..17 views (AO.cs:787-820) and OcclusionHQ1..4 (18..21), copied to the host.
    std::vector<uint8_t> DebugBuffer(int32_t debugId, meao_desc *desc = nullptr, int32_t frame = 0)
    {
        meao_desc d{};
        check(meao_get_intermediate(ctx_, frame, debugId, nullptr, 0, MEAO_MEM_HOST, &d));
        std::vector<uint8_t> data(d.bytes);
        check(meao_get_intermediate(ctx_, frame, debugId, data.data(), d.bytes, MEAO_MEM_HOST, &d));
        if (desc) *desc = d;
        return data;
    }

    // Bit f = frame f of the last Render* held NaN / inf / out-of-range depth texels and ran the IEEE-division bodies.
    uint64_t HostileFrames()
    {
        uint64_t mask = 0;
        check(meao_hostile_frames(ctx_, &mask));
        return mask;
    }

    meao_ctx *native() { return ctx_; }

private:
    void sync()   // CheckPropertiesChanged (AO.cs:104-113): push only what changed
    {
        meao_params p;
        meao_default_params(&p);
        p.noise_filter_tolerance = noiseFilterTolerance;
        p.blur_tolerance = blurTolerance;
        p.upsample_tolerance = upsampleTolerance;
        p.thickness_modifier = thicknessModifier;
        p.intensity = intensity;
        p.near_clip = nearClipPlane;
        p.far_clip = farClipPlane;
        p.proj00 = projection00;
        p.reversed_z = usesReversedZBuffer ? 1 : 0;
        p.single_pass_stereo = singlePassStereoEnabled ? 1 : 0;
        if (!applied_valid_ || !same(p, applied_)) {
            check(meao_set_params(ctx_, &p));
            applied_ = p;
            applied_valid_ = true;
        }
    }
    static bool same(const meao_params &a, const meao_params &b)
    {
        return a.noise_filter_tolerance == b.noise_filter_tolerance && a.blur_tolerance == b.blur_tolerance &&
               a.upsample_tolerance == b.upsample_tolerance && a.thickness_modifier == b.thickness_modifier &&
               a.intensity == b.intensity && a.near_clip == b.near_clip && a.far_clip == b.far_clip &&
               a.proj00 == b.proj00 && a.reversed_z == b.reversed_z && a.single_pass_stereo == b.single_pass_stereo;
    }
    void check(int32_t rc)
    {
        if (rc == MEAO_OK) return;
        std::string msg = meao_last_error(ctx_);
        if (msg.empty()) msg = meao_status_string(rc);
        throw Error(rc, msg);
    }

    meao_ctx *ctx_ = nullptr;
    meao_config cfg_{};
    meao_params applied_{};
    bool applied_valid_ = false;
};

// One context per device behind meao_pool_*: frame f of a batch runs on member f mod G (the in-process multi-GPU host;
// SURVEY 8e).  `devices` may repeat an ordinal (several members on one GPU: a throughput mode of its own).
class AmbientOcclusionPool {
public:
    AmbientOcclusionPool(int32_t pixelWidth, int32_t pixelHeight, const std::vector<int32_t> &devices, int32_t maxBatchPerMember = 1,
                         meao_ao_format aoFormat = MEAO_AO_R8, bool pipelined = false)
    {
        meao_config cfg;
        meao_default_config(&cfg);
        cfg.width = pixelWidth;
        cfg.height = pixelHeight;
        cfg.ao_format = aoFormat;
        cfg.max_batch = maxBatchPerMember;
        cfg.pipelined = pipelined ? 1 : 0;
        const int32_t rc = meao_pool_create(&cfg, devices.data(), static_cast<int32_t>(devices.size()), &pool_);
        if (rc != MEAO_OK) throw Error(rc, meao_pool_last_error(nullptr));
    }
    AmbientOcclusionPool(const AmbientOcclusionPool &) = delete;
    AmbientOcclusionPool &operator=(const AmbientOcclusionPool &) = delete;
    ~AmbientOcclusionPool() { meao_pool_destroy(pool_); }

    int32_t Size() const { return meao_pool_size(pool_); }
    int32_t DeviceOfFrame(int32_t frame) const { return meao_pool_device_of_frame(pool_, frame); }
    void SetParams(const meao_params &p) { check(meao_pool_set_params(pool_, &p)); }
    // Frame f resident on DeviceOfFrame(f); asynchronous (Synchronize()).
    void RenderDeviceBatch(const std::vector<const void *> &deviceDepth, const std::vector<void *> &deviceAo)
    {
        check(meao_pool_execute_batch(pool_, static_cast<int32_t>(deviceDepth.size()), deviceDepth.data(), MEAO_MEM_DEVICE,
                                      deviceAo.data(), MEAO_MEM_DEVICE));
    }
    void PrefetchBatch(const std::vector<const void *> &nextDeviceDepth)
    {
        check(meao_pool_prefetch_batch(pool_, static_cast<int32_t>(nextDeviceDepth.size()), nextDeviceDepth.data()));
    }
    void GatherToDevice(const std::vector<const void *> &deviceAo, const std::vector<void *> &dst, int32_t dstDevice)
    {
        check(meao_pool_gather_to_device(pool_, static_cast<int32_t>(deviceAo.size()), deviceAo.data(), dst.data(), dstDevice));
    }
    meao_pool_path GatherPath(int32_t member, int32_t dstDevice) const
    {
        return static_cast<meao_pool_path>(meao_pool_gather_path(pool_, member, dstDevice));
    }
    void Synchronize() { check(meao_pool_synchronize(pool_)); }
    // MEAO_POOL_SPIN_US / MEAO_POOL_BIND_NUMA (the workers bind themselves to their device's NUMA node; before the first device batch)
    void Configure(meao_pool_option key, int32_t value) { check(meao_pool_configure(pool_, key, value)); }
    int32_t NumaNodeOfMember(int32_t member) const
    {
        int32_t node = -1;
        (void)meao_pool_member_placement(pool_, member, &node, nullptr);
        return node;
    }
    meao_pool *native() { return pool_; }

private:
    void check(int32_t rc)
    {
        if (rc != MEAO_OK) throw Error(rc, meao_pool_last_error(pool_));
    }
    meao_pool *pool_ = nullptr;
};

}  // namespace MiniEngineAO
