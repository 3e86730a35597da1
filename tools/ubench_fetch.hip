// ubench_fetch.hip -- calibration of the rocprofv3 PMC counters FETCH_SIZE / WRITE_SIZE against streams whose byte counts are
// known exactly, per access width (the guide says 16 B/lane coalesced reads are tallied at half size on
// gfx950; which other widths are?).  Every kernel reads (or writes) one buffer of N bytes exactly once, lane i of the grid
// touching element i, i + T, ... (T = threads of the grid): the access patterns of the product kernels.
//
//   cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d <out>/fetch -o pmc -- ubench_fetch
//   cd /tmp && rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d <out>/write -o pmc -- ubench_fetch
//   python tools/pmc_calibration.py <out>          -> profiles/r05_pmc_calibration.json
//
// Without a profiler it prints the kernels' own rates (GB/s), which is also the copy-rate picture per width.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef float float2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef unsigned short ushort4v __attribute__((ext_vector_type(4)));
typedef unsigned char uchar4v __attribute__((ext_vector_type(4)));

template <typename T> __device__ __forceinline__ float fold(T v);
template <> __device__ __forceinline__ float fold(float v) { return v; }
template <> __device__ __forceinline__ float fold(float2v v) { return v.x + v.y; }
template <> __device__ __forceinline__ float fold(float4v v) { return (v.x + v.y) + (v.z + v.w); }
template <> __device__ __forceinline__ float fold(ushort4v v) { return static_cast<float>(v.x + v.y + v.z + v.w); }
template <> __device__ __forceinline__ float fold(uchar4v v) { return static_cast<float>(v.x + v.y + v.z + v.w); }
template <> __device__ __forceinline__ float fold(unsigned char v) { return static_cast<float>(v); }
template <> __device__ __forceinline__ float fold(unsigned short v) { return static_cast<float>(v); }

// NT: non-temporal loads (what the downsample stream uses); UNROLL independent loads in flight per lane
template <typename T, bool NT, int UNROLL>
__global__ __launch_bounds__(256) void read_kernel(const T *__restrict__ src, size_t n, float *sink)
{
    const size_t threads = static_cast<size_t>(gridDim.x) * 256;
    float acc = 0.0f;
    for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < n; i += threads * UNROLL) {
        T v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const size_t j = i + threads * u < n ? i + threads * u : i;
            v[u] = NT ? __builtin_nontemporal_load(src + j) : src[j];
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += fold<T>(v[u]);
    }
    if (acc == 1234.5678f) *sink = acc;       // keeps the loads alive; never true for the zero-filled buffer
}

template <typename T, bool NT>
__global__ __launch_bounds__(256) void write_kernel(T *__restrict__ dst, size_t n, T value)
{
    const size_t threads = static_cast<size_t>(gridDim.x) * 256;
    for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < n; i += threads) {
        if (NT) __builtin_nontemporal_store(value, dst + i);
        else dst[i] = value;
    }
}

// a row segment per lane at a stride, like the level windows the render / upsample tiles gather: each 256-thread workgroup reads
// rows of 40 floats (160 B, 16 B per lane, 10 lanes per row) out of 64-float rows -- partially used 128-byte lines
__global__ __launch_bounds__(256) void read_window_kernel(const float4v *__restrict__ src, size_t rows, float *sink)
{
    float acc = 0.0f;
    for (size_t item = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; item < rows * 10; item += static_cast<size_t>(gridDim.x) * 256) {
        const size_t r = item / 10, k = item % 10;
        acc += fold<float4v>(src[r * 16 + k]);                 // quads 0..9 of a 16-quad row
    }
    if (acc == 1234.5678f) *sink = acc;
}

// The full-resolution pass's output pattern: a lane stores 4 bytes, 16 lanes make one 64-byte row segment, consecutive groups of 16
// lanes go to the NEXT ROW of a W-byte-pitch image (a 64 x 4 texel block per wave); every byte of the image is written once.
__global__ __launch_bounds__(256) void write_tile_rows_kernel(uchar4v *__restrict__ dst, int w_bytes, int h, uchar4v value, bool nt)
{
    // workgroup = a 64-byte x 16-row block; blocks tile the image
    const int blocks_x = w_bytes / 64;
    for (int b = blockIdx.x; b < blocks_x * (h / 16); b += gridDim.x) {
        const int x = (b % blocks_x) * 64 + (threadIdx.x & 15) * 4, y = (b / blocks_x) * 16 + (threadIdx.x >> 4);
        uchar4v *p = dst + (static_cast<size_t>(y) * w_bytes + x) / 4;
        if (nt) __builtin_nontemporal_store(value, p);
        else *p = value;
    }
}
// and its f16 depth input: 8 bytes per lane, 16 lanes = one 128-byte row segment, next group of lanes = next row
__global__ __launch_bounds__(256) void read_tile_rows_kernel(const ushort4v *__restrict__ src, int w_bytes, int h, float *sink)
{
    const int blocks_x = w_bytes / 128;
    float acc = 0.0f;
    for (int b = blockIdx.x; b < blocks_x * (h / 16); b += gridDim.x) {
        const int x = (b % blocks_x) * 128 + (threadIdx.x & 15) * 8, y = (b / blocks_x) * 16 + (threadIdx.x >> 4);
        acc += fold<ushort4v>(__builtin_nontemporal_load(src + (static_cast<size_t>(y) * w_bytes + x) / 8));
    }
    if (acc == 1234.5678f) *sink = acc;
}

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <typename K>
static int timed(const char *name, size_t bytes, K launch)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    launch();                                   // warm-up (the profiler sees every launch: all have the same byte count)
    CHECK(hipEventRecord(a, nullptr));
    launch();
    CHECK(hipEventRecord(b, nullptr));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    std::printf("{\"kernel\": \"%s\", \"bytes\": %zu, \"GBps\": %.1f}\n", name, bytes, bytes / (ms * 1e-3) / 1e9);
    return 0;
}

int main(int argc, char **argv)
{
    const size_t bytes = (argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 1024ull) << 20;       // MiB; default 1 GiB: four times the Infinity Cache
    void *buf = nullptr;
    float *sink = nullptr;
    CHECK(hipMalloc(&buf, bytes));
    CHECK(hipMalloc(&sink, 256));
    CHECK(hipMemset(buf, 0, bytes));
    CHECK(hipDeviceSynchronize());
    const dim3 grid(256 * 32), block(256);
    int rc = 0;
#define READ(T, NT, U, NAME) rc |= timed(NAME, bytes, [&] { read_kernel<T, NT, U><<<grid, block>>>(static_cast<const T *>(buf), bytes / sizeof(T), sink); })
    READ(unsigned char, false, 4, "read_1B_per_lane");
    READ(unsigned short, false, 4, "read_2B_per_lane");
    READ(float, false, 4, "read_4B_per_lane");
    READ(uchar4v, false, 4, "read_4B_per_lane_uchar4");
    READ(float2v, false, 4, "read_8B_per_lane");
    READ(ushort4v, true, 4, "read_8B_per_lane_nt");
    READ(float4v, false, 4, "read_16B_per_lane");
    READ(float4v, true, 4, "read_16B_per_lane_nt");
    READ(float4v, true, 1, "read_16B_per_lane_nt_one_in_flight");
    rc |= timed("read_window_160B_of_256B_rows_16B_per_lane", bytes / 256 * 160,
                [&] { read_window_kernel<<<grid, block>>>(static_cast<const float4v *>(buf), bytes / 256, sink); });
#define WRITE(T, NT, V, NAME) rc |= timed(NAME, bytes, [&] { write_kernel<T, NT><<<grid, block>>>(static_cast<T *>(buf), bytes / sizeof(T), V); })
    WRITE(unsigned char, false, (unsigned char)0, "write_1B_per_lane");
    WRITE(float, false, 0.0f, "write_4B_per_lane");
    WRITE(uchar4v, true, (uchar4v{0, 0, 0, 0}), "write_4B_per_lane_nt");
    WRITE(float2v, true, (float2v{0, 0}), "write_8B_per_lane_nt");
    WRITE(float4v, false, (float4v{0, 0, 0, 0}), "write_16B_per_lane");
    WRITE(float4v, true, (float4v{0, 0, 0, 0}), "write_16B_per_lane_nt");
    // image-shaped streams (pitch 3840 / 7680 bytes, rows as the upsample tiles touch them); 1 GiB = 3840 x 279620 rows -> use 3840 x 262144
    const int img_h = 262144;
    rc |= timed("write_64B_row_segments_4B_per_lane", static_cast<size_t>(3840) * img_h,
                [&] { write_tile_rows_kernel<<<grid, block>>>(static_cast<uchar4v *>(buf), 3840, img_h, uchar4v{0, 0, 0, 0}, false); });
    rc |= timed("write_64B_row_segments_4B_per_lane_nt", static_cast<size_t>(3840) * img_h,
                [&] { write_tile_rows_kernel<<<grid, block>>>(static_cast<uchar4v *>(buf), 3840, img_h, uchar4v{0, 0, 0, 0}, true); });
    rc |= timed("read_128B_row_segments_8B_per_lane_nt", static_cast<size_t>(7680) * (img_h / 2),
                [&] { read_tile_rows_kernel<<<grid, block>>>(static_cast<const ushort4v *>(buf), 7680, img_h / 2, sink); });
    CHECK(hipDeviceSynchronize());
    return rc;
}
