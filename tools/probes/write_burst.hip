// Probe (round 6): what would lift a chain-like kernel off the ~5.1 TB/s store rate of one 8-wave workgroup per CU?
// 8192 wave tiles x 160 slabs of 1 KiB, tile-major (a wave fills its own 160 KiB block), every wave: per BURST of k slabs
// { k * gap dependent FMAs; k back-to-back 1 KiB stores (64 lanes x 16 B) }.  Variants:
//   occ    resident 8-wave workgroups per CU (1, 2, 3 by the LDS request: 100 / 70 / 50 KiB)
//   burst  k = 1, 2, 4, 8 (same bytes, same FMAs: only the grouping of the stores in time changes)
//   skew   wave w of a workgroup starts w * skew FMAs late (product waves drift apart; a lock-step probe flatters interleaved layouts)
//   layout 0 tile-major; 3 section-major workgroup-interleaved (write_layout.hip's best)
//   fillN  no loop structure: a wave writes N consecutive KiB and exits, full occupancy (N = 1 is write_ceiling's `fill1`)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/bin/write_burst.bin tools/probes/write_burst.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned u4;

template <int LAYOUT, int BURST>
__global__ __launch_bounds__(512) void wr(u4* __restrict__ out, long ntiles, int gap, int skew) {
    extern __shared__ char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long tile = (long)blockIdx.x * 8 + wave, nwg = ntiles / 8;
    float x = (float)lane;
    if (gap < 0) lds[threadIdx.x] = 1;      // (keeps the LDS allocation alive)
    for (int g = 0; g < wave * skew; ++g) x = x * 1.0001f + 0.5f;
    for (int s0 = 0; s0 < 160; s0 += BURST) {
        for (int g = 0; g < gap * BURST; ++g) x = x * 1.0001f + 0.5f;
#pragma unroll
        for (int b = 0; b < BURST; ++b) {
            const int s = s0 + b, sec = s >> 4, ss = s & 15;
            long piece;
            if (LAYOUT == 0) piece = tile * 160 + s;
            else piece = (((long)sec * nwg + blockIdx.x) * 16 + ss) * 8 + wave;
            u4 v = {__float_as_uint(x), (unsigned)s, (unsigned)lane, 7u};
            __builtin_nontemporal_store(v, out + piece * 64 + lane);
        }
    }
}

template <int N>
__global__ __launch_bounds__(256) void fillN(u4* __restrict__ out) {
    const long w = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        u4 v = {(unsigned)w, (unsigned)i, (unsigned)lane, 7u};
        __builtin_nontemporal_store(v, out + (w * N + i) * 64 + lane);
    }
}


// tile-major with a runtime block pitch (KiB): wave tile T fills [T * pitch, T * pitch + 160) KiB
__global__ __launch_bounds__(512) void wr_pitch(u4* __restrict__ out, int pitch, int gap) {
    extern __shared__ char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long tile = (long)blockIdx.x * 8 + wave;
    float x = (float)lane;
    if (gap < 0) lds[threadIdx.x] = 1;
    for (int s = 0; s < 160; ++s) {
        for (int g = 0; g < gap; ++g) x = x * 1.0001f + 0.5f;
        u4 v = {__float_as_uint(x), (unsigned)s, (unsigned)lane, 7u};
        __builtin_nontemporal_store(v, out + (tile * pitch + s) * 64 + lane);
    }
}

// ---- XCD <-> memory affinity (block b runs on XCD b % 8) -------------------------------------------------------------------
// fill1 with a block -> chunk map: 0 natural (block b writes the 4 KiB chunk b: XCD x only touches chunks = x mod 8),
// 1 each XCD writes a contiguous eighth of the buffer, 2 natural shifted by one chunk (XCD x touches chunks = x + 1 mod 8)
template <int MODE>
__global__ __launch_bounds__(256) void fill1x(u4* __restrict__ out, long nblocks) {
    const long b = blockIdx.x;
    const long chunk = MODE == 0 ? b : (MODE == 1 ? (b % 8) * (nblocks / 8) + b / 8 : (b + 1) % nblocks);
    u4 v = {(unsigned)b, 1u, threadIdx.x, 7u};
    __builtin_nontemporal_store(v, out + chunk * 256 + threadIdx.x);
}
// block-filling (8 waves x 160 slabs, one workgroup per CU) with an XCD-affine layout: a group of 8 consecutive workgroups (one per
// XCD) shares 8 x 1280 KiB, interleaved in granules of G KiB: workgroup x of the group owns the granules = (x + rot) mod 8
__global__ __launch_bounds__(512) void wr_xcd(u4* __restrict__ out, int G, int rot, int gap) {
    extern __shared__ char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long grp = blockIdx.x / 8;
    const int x = (int)((blockIdx.x + rot) % 8);
    float xx = (float)lane;
    if (gap < 0) lds[threadIdx.x] = 1;
    for (int s = 0; s < 160; ++s) {
        for (int g = 0; g < gap; ++g) xx = xx * 1.0001f + 0.5f;
        const int L = wave * 160 + s, q = L / G, r = L % G;
        const long piece = grp * (8 * 1280) + ((long)q * 8 + x) * G + r;
        u4 v = {__float_as_uint(xx), (unsigned)s, (unsigned)lane, 7u};
        __builtin_nontemporal_store(v, out + piece * 64 + lane);
    }
}

static hipEvent_t e0, e1;
template <class F>
static void timeit(const char* name, size_t bytes, F launch) {
    for (int r = 0; r < 3; ++r) launch();
    float best = 1e9f, sum = 0;
    for (int r = 0; r < 12; ++r) {
        (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); sum += ms; if (ms < best) best = ms;
    }
    printf("%-72s %7.1f us avg %7.1f us min  %.2f TB/s (avg)\n", name, sum / 12 * 1e3, best * 1e3, bytes / (sum / 12 * 1e-3) / 1e12);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const long ntiles = 8192;
    const size_t bytes = (size_t)160 * ntiles * 1024;
    u4* buf; if (hipMalloc(&buf, (size_t)200 * ntiles * 1024) != hipSuccess) return 1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    char name[160];
    const long nw = (long)bytes / 1024;     // KiB pieces
    const int lds1 = 100 * 1024;            // one 8-wave workgroup per CU
    if (argc > 1 && argv[1][0] == 'x') {      // XCD <-> memory affinity
        for (int pass = 0; pass < 3; ++pass) {
            printf("# pass %d\n", pass);
            const long nb = nw / 4;
            timeit("fill1 natural (XCD x writes the 4 KiB chunks = x mod 8)", bytes, [&] { hipLaunchKernelGGL(fill1x<0>, dim3(nb), dim3(256), 0, 0, buf, nb); });
            timeit("fill1, each XCD a contiguous eighth of the buffer", bytes, [&] { hipLaunchKernelGGL(fill1x<1>, dim3(nb), dim3(256), 0, 0, buf, nb); });
            timeit("fill1 shifted by one chunk (XCD x writes chunks = x + 1 mod 8)", bytes, [&] { hipLaunchKernelGGL(fill1x<2>, dim3(nb), dim3(256), 0, 0, buf, nb); });
            timeit("tile-major  occ 1  gap  0  (reference)", bytes, [&] { hipLaunchKernelGGL(wr_pitch, dim3(ntiles / 8), dim3(512), lds1, 0, buf, 160, 0); });
            for (int gap : {0, 12}) for (int G : {1, 2, 4, 8, 16, 32}) {
                snprintf(name, sizeof name, "xcd-affine  occ 1  gap %2d  granule %2d KiB  rot 0", gap, G);
                timeit(name, bytes, [&] { hipLaunchKernelGGL(wr_xcd, dim3(ntiles / 8), dim3(512), lds1, 0, buf, G, 0, gap); });
            }
            for (int rot : {1, 2, 3, 4, 5, 6, 7}) {
                snprintf(name, sizeof name, "xcd-affine  occ 1  gap  0  granule  4 KiB  rot %d", rot);
                timeit(name, bytes, [&] { hipLaunchKernelGGL(wr_xcd, dim3(ntiles / 8), dim3(512), lds1, 0, buf, 4, rot, 0); });
            }
        }
        return 0;
    }
    if (argc > 1 && argv[1][0] == 'p') {      // block-pitch sweep: does the rate depend on how the 2048 concurrent streams fall on the channels?
        for (int pass = 0; pass < 3; ++pass) {
            printf("# pass %d\n", pass);
            for (int gap : {0, 12}) for (int pitch : {160, 161, 162, 164, 165, 167, 156, 157, 168, 169, 176, 192, 193}) {
                snprintf(name, sizeof name, "tile-major  occ 1  gap %2d  pitch %3d KiB", gap, pitch);
                timeit(name, bytes, [&] { hipLaunchKernelGGL(wr_pitch, dim3(ntiles / 8), dim3(512), lds1, 0, buf, pitch, gap); });
            }
        }
        return 0;
    }
    // three passes over the whole list (alternating order averages out clock / thermal drift between variants)
    for (int pass = 0; pass < 3; ++pass) {
        printf("# pass %d\n", pass);
        timeit("fill1  (one 1 KiB store per wave, full occupancy)", bytes, [&] { hipLaunchKernelGGL(fillN<1>, dim3(nw / 4), dim3(256), 0, 0, buf); });
        timeit("fill4  (4 consecutive KiB per wave)", bytes, [&] { hipLaunchKernelGGL(fillN<4>, dim3(nw / 16), dim3(256), 0, 0, buf); });
        timeit("fill160 (a 160 KiB block per wave, full occupancy)", bytes, [&] { hipLaunchKernelGGL(fillN<160>, dim3(nw / 640), dim3(256), 0, 0, buf); });
        for (int gap : {0, 12, 25}) for (int L : {0, 3}) for (int burst : {1, 2, 4, 8}) {
            snprintf(name, sizeof name, "%s  occ 1  gap %2d  burst %d", L == 0 ? "tile-major    " : "wg-interleaved", gap, burst);
            timeit(name, bytes, [&] {
                if (L == 0 && burst == 1) hipLaunchKernelGGL((wr<0, 1>), dim3(ntiles / 8), dim3(512), lds1, 0, buf, ntiles, gap, 0);
                if (L == 0 && burst == 2) hipLaunchKernelGGL((wr<0, 2>), dim3(ntiles / 8), dim3(512), lds1, 0, buf, ntiles, gap, 0);
                if (L == 0 && burst == 4) hipLaunchKernelGGL((wr<0, 4>), dim3(ntiles / 8), dim3(512), lds1, 0, buf, ntiles, gap, 0);
                if (L == 0 && burst == 8) hipLaunchKernelGGL((wr<0, 8>), dim3(ntiles / 8), dim3(512), lds1, 0, buf, ntiles, gap, 0);
                if (L == 3 && burst == 1) hipLaunchKernelGGL((wr<3, 1>), dim3(ntiles / 8), dim3(512), lds1, 0, buf, ntiles, gap, 0);
                if (L == 3 && burst == 2) hipLaunchKernelGGL((wr<3, 2>), dim3(ntiles / 8), dim3(512), lds1, 0, buf, ntiles, gap, 0);
                if (L == 3 && burst == 4) hipLaunchKernelGGL((wr<3, 4>), dim3(ntiles / 8), dim3(512), lds1, 0, buf, ntiles, gap, 0);
                if (L == 3 && burst == 8) hipLaunchKernelGGL((wr<3, 8>), dim3(ntiles / 8), dim3(512), lds1, 0, buf, ntiles, gap, 0);
            });
        }
    }
    return 0;
}
