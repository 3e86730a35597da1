// ubench_lds.hip -- what LDS reads cost a VALU-bound loop on gfx950 (companion of ubench_issue.hip).
//
// The render kernel issues ~17 VALU instructions per LDS read instruction.  Each row below runs a
// loop body of 16 VALU instructions (12 independent v_fma_f32 + 4 that consume the loaded values)
// plus one LDS "read group" that returns 16 bytes per lane, in several encodings:
//   none | 2 x ds_read_b64 | 1 x ds_read2_b64 | 1 x ds_read_b128 | 4 x ds_read_b32 | 1 x ds_read_b64 (8 B)
// and prints wall-clock cycles per VALU instruction per SIMD (cycles = wall time x the clock
// measured with s_memtime / s_memrealtime inside the kernel) at 2, 4, 6 and 8 waves per SIMD.  A
// second block runs the LDS reads alone (issue / array rate of each encoding).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

struct Stamp { unsigned long long cycles, realtime; };

#define PROLOGUE                                                                          \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                           \
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();                       \
    const unsigned long long t0 = __builtin_readcyclecounter();                           \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

#define EPILOGUE                                                                          \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                           \
    const unsigned long long t1 = __builtin_readcyclecounter();                           \
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                       \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                    \
    if ((threadIdx.x & 63) == 0) {                                                        \
        Stamp s; s.cycles = t1 - t0; s.realtime = r1 - r0;                                \
        stamps[blockIdx.x * 4 + (threadIdx.x >> 6)] = s;                                  \
    }

enum { NONE = 0, TWO_B64 = 1, READ2_B64 = 2, B128 = 3, FOUR_B32 = 4, ONE_B64 = 5 };

// MIX: 0 = all v_fma_f32; 1 = the render pair mix (per 8: 4 fma, 2 mul clamp, 2 med3 ... here 16 = 2 pairs)
template <int MODE, int MIX>
__global__ __launch_bounds__(256) void k_valu_lds(float *out, Stamp *stamps, int iters, float b, float c)
{
    __shared__ __attribute__((aligned(16))) float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = b + i * 1e-6f;
    __syncthreads();
    float a[16];
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i;
    // consecutive lanes read consecutive elements of the access width (conflict-free), like the render kernel
    const unsigned stride = MODE == B128 ? 16 : (MODE == FOUR_B32 ? 4 : 8);
    const unsigned addr = (threadIdx.x & 63) * stride + (threadIdx.x >> 6) * 2048;
    const float one = 1.0f;
    PROLOGUE
    for (int it = 0; it < iters; ++it) {
        f4 v = {b, c, b, c};
        if (MODE == TWO_B64) {
            f2 lo, hi;
            asm volatile("ds_read_b64 %0, %1" : "=v"(lo) : "v"(addr));
            asm volatile("ds_read_b64 %0, %1 offset:1024" : "=v"(hi) : "v"(addr));
            v = f4{lo.x, lo.y, hi.x, hi.y};
        } else if (MODE == READ2_B64) {
            asm volatile("ds_read2_b64 %0, %1 offset1:128" : "=v"(v) : "v"(addr));
        } else if (MODE == B128) {
            asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
        } else if (MODE == FOUR_B32) {
            float x, y, z, w;
            asm volatile("ds_read_b32 %0, %1" : "=v"(x) : "v"(addr));
            asm volatile("ds_read_b32 %0, %1 offset:256" : "=v"(y) : "v"(addr));
            asm volatile("ds_read_b32 %0, %1 offset:512" : "=v"(z) : "v"(addr));
            asm volatile("ds_read_b32 %0, %1 offset:768" : "=v"(w) : "v"(addr));
            v = f4{x, y, z, w};
        } else if (MODE == ONE_B64) {
            f2 lo;
            asm volatile("ds_read_b64 %0, %1" : "=v"(lo) : "v"(addr));
            v = f4{lo.x, lo.y, lo.x, lo.y};
        }
        if (MIX == 0) {
#pragma unroll
            for (int i = 0; i < 12; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (MODE != NONE) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[12]) : "v"(v.x), "v"(b));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[13]) : "v"(v.y), "v"(b));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[14]) : "v"(v.z), "v"(b));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[15]) : "v"(v.w), "v"(b));
        } else {
            // previous iteration's values feed 1 pair while this iteration's reads are in flight
#pragma unroll
            for (int p = 0; p < 1; ++p) {
                float d1, d2, p1, p2, u1, u2, sum;
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d1) : "v"(a[0]), "v"(b), "v"(c));
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d2) : "v"(a[1]), "v"(b), "v"(c));
                asm volatile("v_mul_f32_e64 %0, %1, %2 clamp" : "=v"(p1) : "v"(d1), "v"(c));
                asm volatile("v_mul_f32_e64 %0, %1, %2 clamp" : "=v"(p2) : "v"(d2), "v"(c));
                asm volatile("v_med3_f32 %0, %1, %2, %3" : "=v"(u1) : "v"(d1), "v"(p2), "v"(one));
                asm volatile("v_med3_f32 %0, %1, %2, %3" : "=v"(u2) : "v"(d2), "v"(p1), "v"(one));
                asm volatile("v_add_f32 %0, %1, %2" : "=v"(sum) : "v"(u1), "v"(u2));
                asm volatile("v_fma_f32 %0, -%1, %2, %3 clamp" : "=v"(a[2]) : "v"(p1), "v"(p2), "v"(sum));
            }
            if (MODE != NONE) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            {
                float d1, d2, p1, p2, u1, u2, sum;
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d1) : "v"(v.x), "v"(b), "v"(c));
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d2) : "v"(v.z), "v"(b), "v"(c));
                asm volatile("v_mul_f32_e64 %0, %1, %2 clamp" : "=v"(p1) : "v"(d1), "v"(c));
                asm volatile("v_mul_f32_e64 %0, %1, %2 clamp" : "=v"(p2) : "v"(d2), "v"(c));
                asm volatile("v_med3_f32 %0, %1, %2, %3" : "=v"(u1) : "v"(d1), "v"(p2), "v"(one));
                asm volatile("v_med3_f32 %0, %1, %2, %3" : "=v"(u2) : "v"(d2), "v"(p1), "v"(one));
                asm volatile("v_add_f32 %0, %1, %2" : "=v"(sum) : "v"(u1), "v"(u2));
                asm volatile("v_fma_f32 %0, -%1, %2, %3 clamp" : "=v"(a[0]) : "v"(p1), "v"(p2), "v"(sum));
                asm volatile("v_add_f32 %0, %1, %2" : "=v"(a[1]) : "v"(v.y), "v"(v.w));   // all four components consumed
            }
        }
    }
    EPILOGUE
    float s = 0; for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// LDS reads alone: 8 read groups per iteration, one wait per iteration
template <int MODE>
__global__ __launch_bounds__(256) void k_lds_only(float *out, Stamp *stamps, int iters, float b, float c)
{
    __shared__ __attribute__((aligned(16))) float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = b + i * 1e-6f;
    __syncthreads();
    const unsigned stride = MODE == B128 ? 16 : (MODE == FOUR_B32 ? 4 : 8);
    const unsigned addr = (threadIdx.x & 63) * stride + (threadIdx.x >> 6) * 2048;
    f4 acc = {0, 0, 0, 0};
    PROLOGUE
    for (int it = 0; it < iters; ++it) {
        f4 v[8];
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if (MODE == TWO_B64) {
                f2 lo, hi;
                asm volatile("ds_read_b64 %0, %1" : "=v"(lo) : "v"(addr));
                asm volatile("ds_read_b64 %0, %1 offset:1024" : "=v"(hi) : "v"(addr));
                v[g] = f4{lo.x, lo.y, hi.x, hi.y};
            } else if (MODE == READ2_B64) {
                asm volatile("ds_read2_b64 %0, %1 offset1:128" : "=v"(v[g]) : "v"(addr));
            } else if (MODE == B128) {
                asm volatile("ds_read_b128 %0, %1" : "=v"(v[g]) : "v"(addr));
            } else {
                float x, y, z, w;
                asm volatile("ds_read_b32 %0, %1" : "=v"(x) : "v"(addr));
                asm volatile("ds_read_b32 %0, %1 offset:256" : "=v"(y) : "v"(addr));
                asm volatile("ds_read_b32 %0, %1 offset:512" : "=v"(z) : "v"(addr));
                asm volatile("ds_read_b32 %0, %1 offset:768" : "=v"(w) : "v"(addr));
                v[g] = f4{x, y, z, w};
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        acc += v[it & 7];
    }
    EPILOGUE
    out[blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

typedef void (*kfn)(float *, Stamp *, int, float, float);
struct Row { const char *name; kfn fn; double valu_per_iter, groups_per_iter; };

int main(int argc, char **argv)
{
    const double target_ms = argc > 1 ? std::atof(argv[1]) : 4.0;
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    float *out; Stamp *stamps;
    const int max_blocks = cus * 8;
    (void)hipMalloc(&out, size_t(max_blocks) * 256 * sizeof(float));
    (void)hipMalloc(&stamps, size_t(max_blocks) * 4 * sizeof(Stamp));
    std::vector<Stamp> host(size_t(max_blocks) * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 40; ++i) k_valu_lds<NONE, 0><<<max_blocks, 256>>>(out, stamps, 8192, 0.999f, 0.001f);
    (void)hipDeviceSynchronize();

    const Row rows[] = {
        {"16 v_fma, no LDS", k_valu_lds<NONE, 0>, 16, 0},
        {"16 v_fma + 1 x ds_read_b64 (8 B/lane)", k_valu_lds<ONE_B64, 0>, 16, 1},
        {"16 v_fma + 2 x ds_read_b64", k_valu_lds<TWO_B64, 0>, 16, 1},
        {"16 v_fma + 1 x ds_read2_b64", k_valu_lds<READ2_B64, 0>, 16, 1},
        {"16 v_fma + 1 x ds_read_b128", k_valu_lds<B128, 0>, 16, 1},
        {"16 v_fma + 4 x ds_read_b32", k_valu_lds<FOUR_B32, 0>, 16, 1},
        {"render mix (17 VALU), no LDS", k_valu_lds<NONE, 1>, 17, 0},
        {"render mix + 2 x ds_read_b64", k_valu_lds<TWO_B64, 1>, 17, 1},
        {"render mix + 1 x ds_read2_b64", k_valu_lds<READ2_B64, 1>, 17, 1},
        {"render mix + 1 x ds_read_b128", k_valu_lds<B128, 1>, 17, 1},
        {"LDS only: 2 x ds_read_b64 per group", k_lds_only<TWO_B64>, 0, 8},
        {"LDS only: 1 x ds_read2_b64 per group", k_lds_only<READ2_B64>, 0, 8},
        {"LDS only: 1 x ds_read_b128 per group", k_lds_only<B128>, 0, 8},
        {"LDS only: 4 x ds_read_b32 per group", k_lds_only<FOUR_B32>, 0, 8},
    };
    std::printf("# %s, %d CUs; a read group = 16 bytes per lane; cycles = wall time x in-kernel clock\n", prop.gcnArchName, cus);
    std::printf("%-42s %6s %14s %16s %10s\n", "loop body", "w/SIMD", "cyc/VALU instr", "cyc/group/SIMD", "clock MHz");
    for (const Row &r : rows) {
        for (int k : {2, 4, 6, 8}) {
            const int blocks = cus * k;
            int iters = 4096;
            r.fn<<<blocks, 256>>>(out, stamps, iters, 0.999f, 0.001f);
            (void)hipDeviceSynchronize();
            float ms = 0;
            (void)hipEventRecord(e0);
            r.fn<<<blocks, 256>>>(out, stamps, iters, 0.999f, 0.001f);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
            iters = std::max(256, int(iters * target_ms / std::max(ms, 1e-3f)));
            std::vector<double> mhz, wall;
            for (int rep = 0; rep < 7; ++rep) {
                (void)hipEventRecord(e0);
                r.fn<<<blocks, 256>>>(out, stamps, iters, 0.999f, 0.001f);
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
                (void)hipEventElapsedTime(&ms, e0, e1);
                if (rep < 2) continue;
                (void)hipMemcpy(host.data(), stamps, size_t(blocks) * 4 * sizeof(Stamp), hipMemcpyDeviceToHost);
                std::vector<double> f;
                for (int w = 0; w < blocks * 4; ++w) f.push_back(double(host[w].cycles) / double(host[w].realtime) * 100.0);
                std::nth_element(f.begin(), f.begin() + f.size() / 2, f.end());
                mhz.push_back(f[f.size() / 2]);
                wall.push_back(ms);
            }
            std::sort(mhz.begin(), mhz.end()); std::sort(wall.begin(), wall.end());
            const double clk = mhz[mhz.size() / 2], wall_ms = wall[wall.size() / 2];
            const double cycles = wall_ms * 1e-3 * clk * 1e6;            // per SIMD, whole launch
            const double per_valu = r.valu_per_iter > 0 ? cycles / (double(iters) * r.valu_per_iter * k) : 0.0;
            const double per_group = r.groups_per_iter > 0 ? cycles / (double(iters) * r.groups_per_iter * k) : 0.0;
            std::printf("%-42s %6d %14.3f %16.3f %10.0f\n", r.name, k, per_valu, per_group, clk);
        }
    }
    return 0;
}
