// Raw material for reproducing torch.randn offline: per Philox subsequence idx the first block's words (x, y), the hardware
// log2 of u, and the hardware sin / cos of the Box-Muller angle (v = fma(y, c, c)).
#include <hip/hip_runtime.h>
#include <stdint.h>
__device__ __forceinline__ uint4 philox_block(uint4 c, unsigned k0, unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
        c = make_uint4(hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0);
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return c;
}
__global__ void kdump(float* out, unsigned long long seed, unsigned long long off, int64_t n) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    const unsigned long long c0 = off >> 2;
    const uint4 r = philox_block(make_uint4((unsigned)c0, (unsigned)(c0 >> 32), (unsigned)idx, (unsigned)((unsigned long long)idx >> 32)),
                                 (unsigned)seed, (unsigned)(seed >> 32));
    const float u = 2.3283064e-10f + ((float)r.x * 2.3283064e-10f);
    const float v = __builtin_fmaf((float)r.y, 1.46291807e-09f, 1.46291807e-09f);
    float sn, cs;
    __sincosf(v, &sn, &cs);
    float* o = out + idx * 8;
    o[0] = __uint_as_float(r.x); o[1] = __uint_as_float(r.y); o[2] = u; o[3] = v; o[4] = __log2f(u); o[5] = sn; o[6] = cs; o[7] = logf(u);
}
extern "C" int probe_dump(float* out, unsigned long long seed, unsigned long long off, int64_t n, void* stream) {
    hipLaunchKernelGGL(kdump, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, seed, off, n);
    return 0;
}
