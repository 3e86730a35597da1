// AmbientOcclusionOverMeao.cs -- drop-in for the hot path of MiniEngineAO.AmbientOcclusion
// (reference: Assets/MiniEngineAO/AmbientOcclusion.cs) on top of libmeao_hip.so.
//
// NOT COMPILED HERE (no C# toolchain in the build image); shipped as the reference-side
// binding a maintainer would add.  It keeps the six public properties of the reference
// component (AO.cs:22-66) with the same names, ranges and defaults; what Unity supplied
// implicitly -- the camera terms (AO.cs:339-340,563-573), the depth texture
// (AO.cs:608-641) and the "AmbientOcclusion" render texture (AO.cs:475,824) -- becomes
// explicit arguments.  Inside Unity the two IntPtrs of Render() are
// Texture.GetNativeTexturePtr() interop handles of device-resident buffers; outside Unity
// they are plain host arrays (RenderHost).

using System;
using System.Runtime.InteropServices;
using MiniEngineAO.Native;

namespace MiniEngineAO
{
    public sealed class AmbientOcclusion : IDisposable
    {
        // ---- Exposed properties (same names / defaults as AO.cs:20-68) -------------------
        float _noiseFilterTolerance = 0;       // Range(-8, 0)
        float _blurTolerance = -4.6f;          // Range(-8, -1)
        float _upsampleTolerance = -12;        // Range(-12, -1)
        float _thicknessModifier = 1;          // Range(1, 10)
        float _intensity = 1;                  // Range(0, 2)
        bool _ambientOnly = true;

        public float noiseFilterTolerance { get { return _noiseFilterTolerance; } set { _noiseFilterTolerance = value; } }
        public float blurTolerance { get { return _blurTolerance; } set { _blurTolerance = value; } }
        public float upsampleTolerance { get { return _upsampleTolerance; } set { _upsampleTolerance = value; } }
        public float thicknessModifier { get { return _thicknessModifier; } set { _thicknessModifier = value; } }
        public float intensity { get { return _intensity; } set { _intensity = value; } }
        public bool ambientOnly { get { return _ambientOnly; } set { _ambientOnly = value; } }

        // ---- Camera terms the reference read from UnityEngine.Camera ----------------------
        public float nearClipPlane = 0.3f;
        public float farClipPlane = 1000;
        public float projection00 = 0.9742786f;     // camera.projectionMatrix[0, 0]
        public bool usesReversedZBuffer = true;     // SystemInfo.usesReversedZBuffer
        public bool singlePassStereoEnabled = false; // AO.cs:392-401; pixelWidth is then the double-wide eye pair

        IntPtr _ctx;
        MeaoConfig _cfg;
        MeaoParams _applied;                        // CheckPropertiesChanged state (AO.cs:84-113)
        bool _haveApplied;

        public int width { get { return _cfg.width; } }
        public int height { get { return _cfg.height; } }

        // hqLevels / sampleSet: variants the reference's shaders carry but its host never dispatches.
        public AmbientOcclusion(int pixelWidth, int pixelHeight, int device = 0,
                                MeaoAoFormat aoFormat = MeaoAoFormat.R8, int maxBatch = 1,
                                int hqLevels = 0, MeaoSampleSet sampleSet = MeaoSampleSet.Checker)
        {
            Meao.meao_default_config(out _cfg);
            _cfg.device = device;
            _cfg.width = pixelWidth;
            _cfg.height = pixelHeight;
            _cfg.ao_format = (int)aoFormat;
            _cfg.max_batch = maxBatch;
            _cfg.hq_levels = hqLevels;
            _cfg.sample_set = (int)sampleSet;
            Check(Meao.meao_create(ref _cfg, out _ctx));
        }

        // LateUpdate (AO.cs:329-350): rebuild only when a property or the screen size changed.
        void SyncParameters(int pixelWidth, int pixelHeight)
        {
            if (pixelWidth != _cfg.width || pixelHeight != _cfg.height)
            {
                Check(Meao.meao_resize(_ctx, pixelWidth, pixelHeight));
                _cfg.width = pixelWidth;
                _cfg.height = pixelHeight;
            }
            MeaoParams p;
            Meao.meao_default_params(out p);
            p.noise_filter_tolerance = _noiseFilterTolerance;
            p.blur_tolerance = _blurTolerance;
            p.upsample_tolerance = _upsampleTolerance;
            p.thickness_modifier = _thicknessModifier;
            p.intensity = _intensity;
            p.near_clip = nearClipPlane;
            p.far_clip = farClipPlane;
            p.proj00 = projection00;
            p.reversed_z = usesReversedZBuffer ? 1 : 0;
            p.single_pass_stereo = singlePassStereoEnabled ? 1 : 0;
            if (!_haveApplied || !p.Equals(_applied))
            {
                Check(Meao.meao_set_params(_ctx, ref p));
                _applied = p;
                _haveApplied = true;
            }
        }

        // Streams of frames: announce the device depth buffer of the frame AFTER the next Render call;
        // that Render then carries its downsample pass inside its last kernel (meao_prefetch_batch).
        public void PrefetchNext(IntPtr nextDeviceDepth)
        {
            // pending property changes first: meao_set_params would drop the announcement
            SyncParameters(_cfg.width, _cfg.height);
            Check(Meao.meao_prefetch_batch(_ctx, 1, new IntPtr[] { nextDeviceDepth }));
        }

        // PushCompositeCommands (AO.cs:822-839) for streams of frames: the composite of a frame this
        // component produced rides inside the render kernel of the NEXT Render call
        // (meao_composite_enqueue); FlushComposite runs whatever still waits.
        public void CompositeWithNextFrame(IntPtr deviceAo, IntPtr deviceColorRgba16f, IntPtr deviceGBuffer0, bool debug)
        {
            int mode = debug ? (int)MeaoCompositeMode.Debug
                             : (ambientOnly && deviceGBuffer0 != IntPtr.Zero ? (int)MeaoCompositeMode.AmbientOnly
                                                                             : (int)MeaoCompositeMode.Multiply);
            Check(Meao.meao_composite_enqueue(_ctx, mode, 1, new IntPtr[] { deviceAo }, new IntPtr[] { deviceColorRgba16f },
                                              mode == (int)MeaoCompositeMode.AmbientOnly ? new IntPtr[] { deviceGBuffer0 } : null));
        }

        public void FlushComposite(IntPtr stream)
        {
            Check(Meao.meao_composite_flush(_ctx, stream));
        }

        // Device-resident depth in -> AO texture out (the recorded "SSAO" command buffer,
        // AO.cs:496-531).  Asynchronous on `stream`.
        public void Render(IntPtr deviceDepth, IntPtr deviceAo, int pixelWidth, int pixelHeight, IntPtr stream)
        {
            SyncParameters(pixelWidth, pixelHeight);
            Check(Meao.meao_execute(_ctx, deviceDepth, (int)MeaoMem.Device, deviceAo, (int)MeaoMem.Device, stream));
        }

        // Host arrays (tools, tests): raw float depth [height * width] -> R8 AO bytes.
        public byte[] RenderHost(float[] depth, int pixelWidth, int pixelHeight)
        {
            SyncParameters(pixelWidth, pixelHeight);
            int texel = _cfg.ao_format == (int)MeaoAoFormat.R8 ? 1 : 2;
            var ao = new byte[pixelWidth * pixelHeight * texel];
            var hd = GCHandle.Alloc(depth, GCHandleType.Pinned);
            var ha = GCHandle.Alloc(ao, GCHandleType.Pinned);
            try
            {
                Check(Meao.meao_execute(_ctx, hd.AddrOfPinnedObject(), (int)MeaoMem.Host,
                                        ha.AddrOfPinnedObject(), (int)MeaoMem.Host, IntPtr.Zero));
            }
            finally { hd.Free(); ha.Free(); }
            return ao;
        }

        // The _debug 1..17 views (AO.cs:787-820).
        public byte[] DebugBuffer(int debugId, out MeaoDesc desc, int frame = 0)
        {
            Check(Meao.meao_get_intermediate(_ctx, frame, debugId, IntPtr.Zero, 0, (int)MeaoMem.Host, out desc));
            var data = new byte[desc.bytes];
            var h = GCHandle.Alloc(data, GCHandleType.Pinned);
            try
            {
                Check(Meao.meao_get_intermediate(_ctx, frame, debugId, h.AddrOfPinnedObject(), desc.bytes,
                                                 (int)MeaoMem.Host, out desc));
            }
            finally { h.Free(); }
            return data;
        }

        void Check(int status)
        {
            if (status == (int)MeaoStatus.Ok) return;
            string detail = Marshal.PtrToStringAnsi(Meao.meao_last_error(_ctx));
            if (string.IsNullOrEmpty(detail)) detail = Marshal.PtrToStringAnsi(Meao.meao_status_string(status));
            throw new InvalidOperationException("meao " + (MeaoStatus)status + ": " + detail);
        }

        // OnDestroy (AO.cs:357-381)
        public void Dispose()
        {
            // (a composite still waiting for its ride is discarded by meao_destroy: call FlushComposite first if it matters)
            if (_ctx != IntPtr.Zero) { Meao.meao_destroy(_ctx); _ctx = IntPtr.Zero; }
        }
    }
}
