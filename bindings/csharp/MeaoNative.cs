// MeaoNative.cs -- P/Invoke declarations for libmeao_hip.so (include/meao.h, ABI version 6).
//
// NOT COMPILED IN THIS REPOSITORY'S ENVIRONMENT: the build image has no dotnet/mono/csc
// (SURVEY.md, "Environment facts").  tests/test_host_mirror.py cross-checks every
// [DllImport] below against the C header (names, arity, struct field order) instead.
//
// One extern per C entry point; struct layouts are sequential and mirror the C structs
// field by field (all members are 4-byte scalars except meao_desc.bytes).

using System;
using System.Runtime.InteropServices;

namespace MiniEngineAO.Native
{
    public enum MeaoStatus
    {
        Ok = 0, InvalidArgument = -1, Hip = -2, OutOfMemory = -3,
        Unsupported = -4, NoDevice = -5, BufferTooSmall = -6
    }

    public enum MeaoAoFormat { R8 = 0, F16 = 1 }
    public enum MeaoF16Rounding { RtzClamp = 0, Rtne = 1 }
    public enum MeaoMem { Host = 0, Device = 1 }
    public enum MeaoDepthFormat { F32 = 0, Unorm16 = 1, Unorm24 = 2, F16 = 3 }
    public enum MeaoCompositeMode { Multiply = 0, AmbientOnly = 1, Debug = 2 }
    public enum MeaoFormat { F32 = 0, F16 = 1, Unorm8 = 2 }
    public enum MeaoSampleSet { Checker = 0, Exhaustive = 1 }
    public enum MeaoPoolOption { SpinUs = 0, BindNuma = 1 }

    [StructLayout(LayoutKind.Sequential)]
    public struct MeaoConfig
    {
        public uint struct_size;
        public int device;
        public int width;
        public int height;
        public int num_levels;
        public int ao_format;
        public int f16_rounding;
        public int max_batch;
        public int depth_format;
        public int hq_levels;
        public int sample_set;
        public int pipelined;
    }

    [StructLayout(LayoutKind.Sequential)]
    public struct MeaoParams
    {
        public uint struct_size;
        public float noise_filter_tolerance;
        public float blur_tolerance;
        public float upsample_tolerance;
        public float thickness_modifier;
        public float intensity;
        public float near_clip;
        public float far_clip;
        public float proj00;
        public int reversed_z;
        public int single_pass_stereo;
    }

    [StructLayout(LayoutKind.Sequential)]
    public struct MeaoDesc
    {
        public int debug_id;
        public int width;
        public int height;
        public int slices;
        public int format;
        public ulong bytes;
    }

    [StructLayout(LayoutKind.Sequential)]
    public struct MeaoRenderConstants
    {
        [MarshalAs(UnmanagedType.ByValArray, SizeConst = 12)] public float[] inv_thickness_table;
        [MarshalAs(UnmanagedType.ByValArray, SizeConst = 12)] public float[] sample_weight_table;
        [MarshalAs(UnmanagedType.ByValArray, SizeConst = 2)] public float[] inv_slice_dimension;
        public float reject_fadeoff;
        public float intensity;
    }

    [StructLayout(LayoutKind.Sequential)]
    public struct MeaoUpsampleConstants
    {
        [MarshalAs(UnmanagedType.ByValArray, SizeConst = 2)] public float[] inv_low_resolution;
        [MarshalAs(UnmanagedType.ByValArray, SizeConst = 2)] public float[] inv_high_resolution;
        public float noise_filter_strength;
        public float step_size;
        public float blur_tolerance;
        public float upsample_tolerance;
    }

    public static class Meao
    {
        const string Lib = "meao_hip";   // libmeao_hip.so
        public const int AbiVersion = 6;
        public const int MaxBatch = 64;
        public const int NumPasses = 7;
        public const int DebugOcclusionHq1 = 18;
        public const int NumBuffers = 21;

        [DllImport(Lib)] public static extern int meao_abi_version();
        [DllImport(Lib)] public static extern IntPtr meao_status_string(int status);
        [DllImport(Lib)] public static extern void meao_default_config(out MeaoConfig cfg);
        [DllImport(Lib)] public static extern void meao_default_params(out MeaoParams p);

        [DllImport(Lib)] public static extern int meao_level_dims(int width, int height, int level, out int out_w, out int out_h);
        [DllImport(Lib)] public static extern int meao_zbuffer_params(ref MeaoParams p, [Out] float[] out4);
        [DllImport(Lib)] public static extern int meao_render_constants_for(int width, int height, ref MeaoParams p, int level, out MeaoRenderConstants constants);
        [DllImport(Lib)] public static extern int meao_render_constants_variant(int width, int height, ref MeaoParams p, int level, int source_tiled, int sample_set, out MeaoRenderConstants constants);
        [DllImport(Lib)] public static extern int meao_upsample_constants_for(int width, int height, ref MeaoParams p, int low_level, out MeaoUpsampleConstants constants);
        [DllImport(Lib)] public static extern int meao_describe_buffer(ref MeaoConfig cfg, int debug_id, out MeaoDesc desc);
        [DllImport(Lib)] public static extern int meao_algorithmic_bytes(ref MeaoConfig cfg, [Out] ulong[] bytes7);

        [DllImport(Lib)] public static extern int meao_create(ref MeaoConfig cfg, out IntPtr ctx);
        [DllImport(Lib)] public static extern int meao_destroy(IntPtr ctx);
        [DllImport(Lib)] public static extern int meao_resize(IntPtr ctx, int width, int height);
        [DllImport(Lib)] public static extern int meao_set_params(IntPtr ctx, ref MeaoParams p);
        [DllImport(Lib)] public static extern int meao_get_params(IntPtr ctx, out MeaoParams p);
        [DllImport(Lib)] public static extern int meao_get_config(IntPtr ctx, out MeaoConfig cfg);
        [DllImport(Lib)] public static extern IntPtr meao_last_error(IntPtr ctx);

        [DllImport(Lib)] public static extern int meao_execute(IntPtr ctx, IntPtr depth, int depth_loc, IntPtr ao_out, int out_loc, IntPtr stream);
        [DllImport(Lib)] public static extern int meao_execute_batch(IntPtr ctx, int n, IntPtr[] depth, int depth_loc, IntPtr[] ao_out, int out_loc, IntPtr stream);
        [DllImport(Lib)] public static extern int meao_prefetch_batch(IntPtr ctx, int n, IntPtr[] depth);
        [DllImport(Lib)] public static extern int meao_synchronize(IntPtr ctx, IntPtr stream);

        [DllImport(Lib)] public static extern int meao_get_intermediate(IntPtr ctx, int frame, int debug_id, IntPtr dst, ulong dst_capacity, int dst_loc, out MeaoDesc desc);
        [DllImport(Lib)] public static extern int meao_set_profiling(IntPtr ctx, int enable);
        [DllImport(Lib)] public static extern int meao_get_pass_times(IntPtr ctx, [Out] float[] ms7, out int samples);
        [DllImport(Lib)] public static extern int meao_selftest(IntPtr ctx, int which, out ulong mismatches);
        [DllImport(Lib)] public static extern int meao_set_tracing(IntPtr ctx, int enable);
        [DllImport(Lib)] public static extern int meao_composite_enqueue(IntPtr ctx, int mode, int n, IntPtr[] ao, IntPtr[] color_rgba16f, IntPtr[] gbuffer0_rgba8);
        [DllImport(Lib)] public static extern int meao_composite_flush(IntPtr ctx, IntPtr stream);
        [DllImport(Lib)] public static extern int meao_composite_pending(IntPtr ctx, out int out_frames);

        // multi-GPU pool: frame f of a batch runs on member f mod G (one context + stream per device)
        [DllImport(Lib)] public static extern int meao_pool_create(ref MeaoConfig cfg, int[] devices, int num_devices, out IntPtr pool);
        [DllImport(Lib)] public static extern int meao_pool_destroy(IntPtr pool);
        [DllImport(Lib)] public static extern int meao_pool_size(IntPtr pool);
        [DllImport(Lib)] public static extern IntPtr meao_pool_context(IntPtr pool, int member);
        [DllImport(Lib)] public static extern int meao_pool_device_of_frame(IntPtr pool, int frame);
        [DllImport(Lib)] public static extern IntPtr meao_pool_last_error(IntPtr pool);
        [DllImport(Lib)] public static extern int meao_pool_set_params(IntPtr pool, ref MeaoParams p);
        [DllImport(Lib)] public static extern int meao_pool_execute_batch(IntPtr pool, int n, IntPtr[] depth, int depth_loc, IntPtr[] ao_out, int out_loc);
        [DllImport(Lib)] public static extern int meao_pool_prefetch_batch(IntPtr pool, int n, IntPtr[] depth);
        [DllImport(Lib)] public static extern int meao_pool_composite_enqueue(IntPtr pool, int mode, int n, IntPtr[] ao, IntPtr[] color_rgba16f, IntPtr[] gbuffer0_rgba8);
        [DllImport(Lib)] public static extern int meao_pool_composite_flush(IntPtr pool);
        [DllImport(Lib)] public static extern int meao_pool_composite_pending(IntPtr pool, out int out_frames);
        [DllImport(Lib)] public static extern int meao_pool_gather_to_device(IntPtr pool, int n, IntPtr[] ao_src, IntPtr[] dst, int dst_device);
        [DllImport(Lib)] public static extern int meao_pool_gather_path(IntPtr pool, int member, int dst_device);   // 0 same device, 1 peer (xGMI), 2 staged
        [DllImport(Lib)] public static extern int meao_pool_synchronize(IntPtr pool);
        // host placement: NUMA node of a device (cpulist: the node's CPUs, "0-15,32-47"), pool options (MeaoPoolOption), where a member's worker runs
        [DllImport(Lib)] public static extern int meao_device_numa_node(int device, out int node, System.Text.StringBuilder cpulist, ulong cpulist_capacity);
        [DllImport(Lib)] public static extern int meao_pool_configure(IntPtr pool, int key, int value);
        [DllImport(Lib)] public static extern int meao_pool_member_placement(IntPtr pool, int member, out int numa_node, out int worker_bound);
        [DllImport(Lib)] public static extern int meao_hostile_frames(IntPtr ctx, out ulong mask);
        [DllImport(Lib)] public static extern int meao_debug_set(IntPtr ctx, int key, int value);   // launch-structure overrides (tests, A/B runs)
        [DllImport(Lib)] public static extern int meao_debug_view(IntPtr ctx, int frame, int debug_id, IntPtr dst, int out_loc, IntPtr stream);
        [DllImport(Lib)] public static extern int meao_composite(IntPtr ctx, int mode, IntPtr ao, IntPtr color_rgba16f, IntPtr gbuffer0_rgba8, int loc, IntPtr stream);
    }
}
