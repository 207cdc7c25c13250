// K1: positional encoding — replaces the 41-launch mul/sin/cos/cat sequence of
// Embedding.forward (reference models/nerf.py:21-38).  HBM-bound: 4*C B in, 4*C*(2F+1) B out
// per point (264 B/point for C=3,F=10).  A workgroup encodes a tile of 64 points into LDS
// (sincos shared between the sin and cos channel) and streams the tile out as one contiguous,
// fully coalesced 16-byte-per-lane burst; nothing is re-read.
#include "common.h"

namespace nerfhip {

constexpr int PE_PTS = 64;      // points per workgroup
constexpr int PE_THREADS = 256;

// bands: NULL = the reference's default `logscale=True` bands 2^k (exact powers of two); else F explicit frequency bands (the
// reference's `logscale=False`: torch.linspace(1, 2^(F-1), F), nerf.py:16-19, passed as the module built them)
__global__ __launch_bounds__(PE_THREADS) void posenc_kernel(const float* __restrict__ x, const float* __restrict__ bands,
                                                             float* __restrict__ out, int64_t n, int C, int F) {
    extern __shared__ __attribute__((aligned(16))) float tile[];
    const int OC = C * (2 * F + 1);
    const int64_t base = (int64_t)blockIdx.x * PE_PTS;
    const int npts = (int)min((int64_t)PE_PTS, n - base);
    const int tid = threadIdx.x;

    // identity channels
    for (int i = tid; i < npts * C; i += PE_THREADS) {
        int pt = i / C, c = i - pt * C;
        tile[pt * OC + c] = x[base * C + i];
    }
    // one sincos per (point, channel, frequency); lanes run along points => LDS stride OC (odd for
    // the shapes the reference uses: 63, 27) is bank-conflict free.
    const int items = PE_PTS * C * F;
    for (int it = tid; it < items; it += PE_THREADS) {
        int pt = it & (PE_PTS - 1);
        int r = it >> 6;
        int k = r / C, c = r - k * C;
        if (pt < npts) {
            float v = x[(base + pt) * C + c];
            float arg = v * (bands ? bands[k] : __builtin_ldexpf(1.0f, k));  // freq*x in fp32 first (logscale: exact power of two)
            float s, co;
            sincosf(arg, &s, &co);
            float* row = tile + pt * OC + C + 2 * C * k + c;
            row[0] = s;
            row[C] = co;
        }
    }
    __syncthreads();
    const int total = npts * OC;
    float* dst = out + base * OC;
    if ((((uintptr_t)dst) & 15) == 0) {
        const int nv = total >> 2;
        const float4* t4 = reinterpret_cast<const float4*>(tile);
        float4* d4 = reinterpret_cast<float4*>(dst);
        for (int i = tid; i < nv; i += PE_THREADS) d4[i] = t4[i];
        for (int i = (nv << 2) + tid; i < total; i += PE_THREADS) dst[i] = tile[i];
    } else {
        for (int i = tid; i < total; i += PE_THREADS) dst[i] = tile[i];
    }
}

// gx[c] = g_id[c] + sum_k 2^k * (cos(2^k x) * g_sin[k,c] - sin(2^k x) * g_cos[k,c])
__global__ __launch_bounds__(PE_THREADS) void posenc_bwd_kernel(const float* __restrict__ x, const float* __restrict__ bands,
                                                                 const float* __restrict__ gout,
                                                                 float* __restrict__ gx, int64_t n, int C, int F) {
    extern __shared__ __attribute__((aligned(16))) float tile[];
    const int OC = C * (2 * F + 1);
    const int64_t base = (int64_t)blockIdx.x * PE_PTS;
    const int npts = (int)min((int64_t)PE_PTS, n - base);
    const int tid = threadIdx.x;
    const int total = npts * OC;
    const float* src = gout + base * OC;
    for (int i = tid; i < total; i += PE_THREADS) tile[i] = src[i];
    __syncthreads();
    for (int i = tid; i < npts * C; i += PE_THREADS) {
        int pt = i / C, c = i - pt * C;
        float v = x[base * C + i];
        const float* row = tile + pt * OC;
        float acc = row[c];
        for (int k = 0; k < F; ++k) {
            float f = bands ? bands[k] : __builtin_ldexpf(1.0f, k);
            float s, co;
            sincosf(v * f, &s, &co);
            acc += f * (co * row[C + 2 * C * k + c] - s * row[C + 2 * C * k + C + c]);
        }
        gx[base * C + i] = acc;
    }
}

}  // namespace nerfhip

extern "C" int nerfhip_posenc_bands(const float* x, const float* bands, float* out, int64_t n, int C, int n_freqs,
                                    nerfhip_stream_t stream) {
    NERFHIP_CHECK_ARG(n >= 0 && C >= 1 && C <= 8 && n_freqs >= 0 && n_freqs <= 16);
    if (n == 0) return 0;
    NERFHIP_CHECK_ARG(x && out);
    const int OC = C * (2 * n_freqs + 1);
    const int64_t blocks = (n + nerfhip::PE_PTS - 1) / nerfhip::PE_PTS;
    size_t lds = (size_t)nerfhip::PE_PTS * OC * sizeof(float);
    hipLaunchKernelGGL(nerfhip::posenc_kernel, dim3((unsigned)blocks), dim3(nerfhip::PE_THREADS), lds,
                       (hipStream_t)stream, x, bands, out, n, C, n_freqs);
    return nerfhip_launch_status();
}
extern "C" int nerfhip_posenc(const float* x, float* out, int64_t n, int C, int n_freqs, nerfhip_stream_t stream) {
    return nerfhip_posenc_bands(x, nullptr, out, n, C, n_freqs, stream);
}

extern "C" int nerfhip_posenc_bands_bwd(const float* x, const float* bands, const float* gout, float* gx, int64_t n, int C,
                                        int n_freqs, nerfhip_stream_t stream) {
    NERFHIP_CHECK_ARG(n >= 0 && C >= 1 && C <= 8 && n_freqs >= 0 && n_freqs <= 16);
    if (n == 0) return 0;
    NERFHIP_CHECK_ARG(x && gout && gx);
    const int OC = C * (2 * n_freqs + 1);
    const int64_t blocks = (n + nerfhip::PE_PTS - 1) / nerfhip::PE_PTS;
    size_t lds = (size_t)nerfhip::PE_PTS * OC * sizeof(float);
    hipLaunchKernelGGL(nerfhip::posenc_bwd_kernel, dim3((unsigned)blocks), dim3(nerfhip::PE_THREADS), lds,
                       (hipStream_t)stream, x, bands, gout, gx, n, C, n_freqs);
    return nerfhip_launch_status();
}
extern "C" int nerfhip_posenc_bwd(const float* x, const float* gout, float* gx, int64_t n, int C, int n_freqs,
                                  nerfhip_stream_t stream) {
    return nerfhip_posenc_bands_bwd(x, nullptr, gout, gx, n, C, n_freqs, stream);
}
