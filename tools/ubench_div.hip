// ubench_div.hip -- which short v_rcp_f32-based sequences reproduce IEEE division exactly?
// Exhaustive over every finite binary32 x with 2^-100 <= |x| <= 2^100 (design input for the
// strict-mode kernels; the chosen sequences are re-verified by meao_selftest in the product).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__device__ __forceinline__ float rcp_hw(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fm(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

__device__ __forceinline__ float rcp_A(float x) { float r = rcp_hw(x); float e = fm(-x, r, 1.0f); return fm(e, r, r); }
__device__ __forceinline__ float rcp_B(float x) { float r = rcp_A(x); float e = fm(-x, r, 1.0f); return fm(e, r, r); }
__device__ __forceinline__ float div_C(float k, float x) { float r = rcp_hw(x); float q = k * r; float e = fm(-x, q, k); return fm(e, r, q); }
__device__ __forceinline__ float div_D(float k, float x) { float r = rcp_A(x); float q = k * r; float e = fm(-x, q, k); return fm(e, r, q); }
__device__ __forceinline__ float div_E(float k, float x) { float r = rcp_hw(x); float q = k * r; float e = fm(-x, q, k); q = fm(e, r, q); e = fm(-x, q, k); return fm(e, r, q); }

__device__ __forceinline__ float div_F(float k, float x) { float r = rcp_A(x); float q = k * r; float e = fm(-x, q, k); q = fm(e, r, q); e = fm(-x, q, k); return fm(e, r, q); }
__device__ __forceinline__ bool safe(float x) { float a = __builtin_fabsf(x); return a >= 0x1p-100f && a <= 0x1p100f; }

__global__ void sweep(unsigned long long *bad)
{
    unsigned long long b[12] = {};
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
        const float x = __builtin_bit_cast(float, (uint32_t)i);
        if (!safe(x)) continue;
        const float ref1 = 1.0f / x;
        b[0] += rcp_hw(x) != ref1;
        b[1] += rcp_A(x) != ref1;
        b[2] += rcp_B(x) != ref1;
        const float ks[3] = {3.0f, 9.0f, 0.7853982f};
        for (int j = 0; j < 3; ++j) {
            const float ref = ks[j] / x;
            b[3 + j] += div_C(ks[j], x) != ref;
            b[6 + j] += div_D(ks[j], x) != ref;
            b[9 + j] += div_E(ks[j], x) != ref;
        }
    }
    for (int j = 0; j < 12; ++j) if (b[j]) atomicAdd(&bad[j], b[j]);
}

// general a/b: b sweeps all safe floats with a stride, a = hashed patterns with |a/b| safe
__global__ void sweep2(unsigned long long *bad, uint32_t salt)
{
    unsigned long long c = 0, d = 0, n = 0, ee = 0, ff = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
        const float x = __builtin_bit_cast(float, (uint32_t)i);
        if (!safe(x)) continue;
        { float ax = __builtin_fabsf(x); if (ax < 0x1p-60f || ax > 0x1p60f) continue; }
        uint32_t h = (uint32_t)i * 2654435761u + salt; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        // a: random mantissa/sign, exponent within +-20 of x's so the quotient is comfortably normal
        const uint32_t ex = ((uint32_t)i >> 23) & 0xff;
        const uint32_t ea = ex - 20 + (h >> 24) % 41;
        const float a = __builtin_bit_cast(float, (h & 0x807fffffu) | (ea << 23));
        const float ref = a / x;
        c += div_C(a, x) != ref;
        d += div_D(a, x) != ref;
        ee += div_E(a, x) != ref;
        ff += div_F(a, x) != ref;
        if (div_D(a, x) != ref && atomicAdd(&bad[5], 1ull) < 6) printf("fail a=%a b=%a ref=%a got=%a\n", a, x, ref, div_D(a, x));
        ++n;
    }
    if (c) atomicAdd(&bad[0], c);
    if (d) atomicAdd(&bad[1], d);
    atomicAdd(&bad[2], n);
    if (ee) atomicAdd(&bad[3], ee);
    if (ff) atomicAdd(&bad[4], ff);
}

int main()
{
    unsigned long long *bad, host[12];
    hipMalloc(&bad, sizeof host);
    hipMemset(bad, 0, sizeof host);
    sweep<<<4096, 256>>>(bad);
    hipMemcpy(host, bad, sizeof host, hipMemcpyDeviceToHost);
    const char *names[12] = {"v_rcp_f32 alone", "rcp_A (1 Newton step, 3 ops)", "rcp_B (2 steps, 5 ops)",
                             "div_C 3/x", "div_C 9/x", "div_C 0.785/x", "div_D 3/x", "div_D 9/x", "div_D 0.785/x",
                             "div_E 3/x", "div_E 9/x", "div_E 0.785/x"};
    for (int j = 0; j < 12; ++j) std::printf("%-32s mismatches vs IEEE: %llu\n", names[j], host[j]);
    for (uint32_t salt = 1; salt <= 8; ++salt) {
        hipMemset(bad, 0, sizeof host);
        sweep2<<<4096, 256>>>(bad, salt * 0x9E3779B9u);
        hipMemcpy(host, bad, sizeof host, hipMemcpyDeviceToHost);
        std::printf("general a/b salt %u: div_C %llu  div_D %llu  div_E %llu  div_F %llu of %llu\n", salt, host[0], host[1], host[3], host[4], host[2]);
    }
    return 0;
}
