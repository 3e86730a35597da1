// ubench_launch.hip -- what a kernel boundary costs next to a grid-wide barrier, on this device.
// Evidence for DESIGN.md section 9.3 (one frame per call: five dependent launches vs one cooperative
// launch with barriers between the stages).
//
//   rows "launch":  N dependent empty kernels back to back in one stream -> microseconds per boundary
//   rows "barrier": one launch of G co-resident workgroups crossing R grid barriers (one atomic
//                   increment per workgroup, agent-scope acquire spin on the counter; the spin is
//                   bounded, a stuck barrier sets a flag instead of hanging) -> microseconds per barrier
//   rows "work+":   the same with ~2 us of arithmetic per stage in every workgroup, so that the arrival
//                   skew of a real stage is in the number
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

__global__ void k_empty(int *sink) { if (sink && threadIdx.x == 1024) *sink = 1; }

__global__ __launch_bounds__(256) void k_barriers(unsigned *counter, unsigned *stuck, float *out, int rounds, int work)
{
    const unsigned G = gridDim.x;
    float a = threadIdx.x * 0.001f;
    for (int r = 0; r < rounds; ++r) {
        for (int i = 0; i < work; ++i) a = __builtin_fmaf(a, 0.999f, 0.001f);
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (static_cast<unsigned>(r) + 1u) * G;
            unsigned spins = 0;
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if (++spins > (1u << 22)) { *stuck = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
    }
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

// Three-level arrival tree (groups of G/256 workgroups -> 8 groups of 32 -> 1) and one generation word that
// everybody polls: 4 + 32 + 8 serialised atomics deep instead of G on one address.
__global__ __launch_bounds__(256) void k_barriers_tree(unsigned *cells, unsigned *stuck, float *out, int rounds, int work)
{
    // cells: [0] generation, [64 * (1 + g)] level-1 counters (256, one cache line each), then 8 level-2, then 1 level-3
    const unsigned G = gridDim.x, g1 = blockIdx.x & 255u, g2 = g1 & 7u;
    const unsigned n1 = (G + 255u - g1) / 256u;           // members of level-1 group g1
    unsigned *gen = cells, *c1 = cells + 64u * (1u + g1), *c2 = cells + 64u * (257u + g2), *c3 = cells + 64u * 265u;
    float a = threadIdx.x * 0.001f;
    for (int r = 0; r < rounds; ++r) {
        for (int i = 0; i < work; ++i) a = __builtin_fmaf(a, 0.999f, 0.001f);
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned round = static_cast<unsigned>(r) + 1u;
            if (__hip_atomic_fetch_add(c1, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == n1 * round - 1u)
                if (__hip_atomic_fetch_add(c2, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == (G < 256u ? (G + 7u - g2) / 8u : 32u) * round - 1u)
                    if (__hip_atomic_fetch_add(c3, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == (G < 8u ? G : 8u) * round - 1u)
                        __hip_atomic_store(gen, round, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            unsigned spins = 0;
            while (__hip_atomic_load(gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < round) {
                if (++spins > (1u << 22)) { *stuck = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
    }
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    std::printf("# device %s, %d CUs\n", prop.gcnArchName, prop.multiProcessorCount);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    unsigned *counter, *stuck; float *out; int *sink;
    hipMalloc(&counter, 4); hipMalloc(&stuck, 4); hipMalloc(&out, 2048 * 256 * 4); hipMalloc(&sink, 4);
    // warm-up
    for (int i = 0; i < 2000; ++i) k_empty<<<1024, 256>>>(sink);
    hipDeviceSynchronize();
    for (int grid : {1, 256, 1024, 2048, 8192}) {
        std::vector<double> us;
        for (int rep = 0; rep < 7; ++rep) {
            const int n = 400;
            hipEventRecord(e0);
            for (int i = 0; i < n; ++i) k_empty<<<grid, 256>>>(sink);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            us.push_back(ms * 1e3 / n);
        }
        std::sort(us.begin(), us.end());
        std::printf("launch   grid %5d x 256 threads: %6.2f us per dependent empty kernel (median of 7 x 400)\n", grid, us[3]);
    }
    for (int work : {0, 2000}) {
        for (int G : {256, 512, 1024, 2048}) {
            std::vector<double> us;
            unsigned was_stuck = 0;
            const int rounds = 200;
            for (int rep = 0; rep < 7; ++rep) {
                hipMemset(counter, 0, 4); hipMemset(stuck, 0, 4);
                hipDeviceSynchronize();
                hipEventRecord(e0);
                k_barriers<<<G, 256>>>(counter, stuck, out, rounds, work);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms = 0; hipEventElapsedTime(&ms, e0, e1);
                us.push_back(ms * 1e3 / rounds);
                unsigned sflag = 0; hipMemcpy(&sflag, stuck, 4, hipMemcpyDeviceToHost); was_stuck |= sflag;
            }
            std::sort(us.begin(), us.end());
            std::printf("%s grid %5d x 256 threads: %6.2f us per stage of %d fma + grid barrier (median of 7 x %d)%s\n",
                        work ? "work+   " : "barrier ", G, us[3], work, rounds, was_stuck ? "  STUCK (not co-resident?)" : "");
        }
    }
    unsigned *cells;
    hipMalloc(&cells, 64 * 4 * 300);
    for (int work : {0, 2000}) {
        for (int G : {256, 512, 1024, 2048}) {
            std::vector<double> us;
            unsigned was_stuck = 0;
            const int rounds = 200;
            for (int rep = 0; rep < 7; ++rep) {
                hipMemset(cells, 0, 64 * 4 * 300); hipMemset(stuck, 0, 4);
                hipDeviceSynchronize();
                hipEventRecord(e0);
                k_barriers_tree<<<G, 256>>>(cells, stuck, out, rounds, work);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms = 0; hipEventElapsedTime(&ms, e0, e1);
                us.push_back(ms * 1e3 / rounds);
                unsigned sflag = 0; hipMemcpy(&sflag, stuck, 4, hipMemcpyDeviceToHost); was_stuck |= sflag;
            }
            std::sort(us.begin(), us.end());
            std::printf("%s grid %5d x 256 threads: %6.2f us per stage of %d fma + tree barrier (median of 7 x %d)%s\n",
                        work ? "tree+   " : "tree    ", G, us[3], work, rounds, was_stuck ? "  STUCK" : "");
        }
    }
    // 2000 fma alone, for the "+" rows
    {
        hipMemset(cells, 0, 64 * 4 * 300);
        std::vector<double> us;
        for (int rep = 0; rep < 7; ++rep) {
            hipEventRecord(e0);
            k_barriers_tree<<<1, 256>>>(cells + 64 * 299, stuck, out, 0, 0);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1); us.push_back(ms * 1e3);
        }
    }
    return 0;
}
