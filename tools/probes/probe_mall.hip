// Does a producer -> consumer hand-over through the 256 MiB Infinity Cache beat HBM?  write(buf, n) then read(buf, n) back to back,
// for n from 32 MB to 2 GB, with plain and non-temporal stores / loads; also read-after-read.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) float f4;
template <int NT>
__global__ __launch_bounds__(256) void wr(f4* p, size_t n4, float v) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        f4 x = {v, v + 1, v + 2, (float)i};
        if (NT) __builtin_nontemporal_store(x, p + i); else p[i] = x;
    }
}
template <int NT, int REV>
__global__ __launch_bounds__(256) void rd(const f4* p, size_t n4, float* out) {
    float acc = 0.f;
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n4; k += (size_t)gridDim.x * 256) {
        const size_t i = REV ? n4 - 1 - k : k;
        f4 x = NT ? __builtin_nontemporal_load(p + i) : p[i];
        acc += x.x + x.y + x.z + x.w;
    }
    if (acc == 123.456f) *out = acc;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
int main() {
    const size_t maxb = (size_t)2 << 30;
    f4* buf; float* out;
    CK(hipMalloc(&buf, maxb)); CK(hipMalloc(&out, 4));
    f4* junk; CK(hipMalloc(&junk, (size_t)1 << 30));
    hipEvent_t e0, e1, e2; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
    const int grid = 256 * 8;
    printf("size_MB  write_GBs  read_after_write_GBs  [nt write / plain read]  [nt write / nt read]  [plain write, reversed read]  read_after_flush_GBs\n");
    for (size_t mb : {32, 64, 128, 192, 256, 384, 512, 1024, 2048}) {
        const size_t n4 = mb * 1024 * 1024 / 16;
        double res[8];
        int k = 0;
        for (int mode = 0; mode < 4; ++mode) {
            float best_w = 1e9, best_r = 1e9;
            for (int rep = 0; rep < 4; ++rep) {
                hipLaunchKernelGGL((wr<0>), dim3(grid), dim3(256), 0, 0, junk, ((size_t)1 << 30) / 16, 1.0f);     // evict
                hipEventRecord(e0);
                if (mode == 0 || mode == 3) hipLaunchKernelGGL((wr<0>), dim3(grid), dim3(256), 0, 0, buf, n4, (float)rep);
                else hipLaunchKernelGGL((wr<1>), dim3(grid), dim3(256), 0, 0, buf, n4, (float)rep);
                hipEventRecord(e1);
                if (mode == 0 || mode == 1) hipLaunchKernelGGL((rd<0, 0>), dim3(grid), dim3(256), 0, 0, buf, n4, out);
                else if (mode == 2) hipLaunchKernelGGL((rd<1, 0>), dim3(grid), dim3(256), 0, 0, buf, n4, out);
                else hipLaunchKernelGGL((rd<0, 1>), dim3(grid), dim3(256), 0, 0, buf, n4, out);
                hipEventRecord(e2);
                CK(hipDeviceSynchronize());
                float tw, tr; hipEventElapsedTime(&tw, e0, e1); hipEventElapsedTime(&tr, e1, e2);
                if (tw < best_w) best_w = tw; if (tr < best_r) best_r = tr;
            }
            if (mode == 0) res[k++] = mb / 1024.0 / (best_w * 1e-3);
            res[k++] = mb / 1024.0 / (best_r * 1e-3);
        }
        // read after the buffer was flushed out by 1 GB of other writes
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL((wr<0>), dim3(grid), dim3(256), 0, 0, junk, ((size_t)1 << 30) / 16, 1.0f);
            hipEventRecord(e1);
            hipLaunchKernelGGL((rd<0, 0>), dim3(grid), dim3(256), 0, 0, buf, n4, out);
            hipEventRecord(e2);
            CK(hipDeviceSynchronize());
            float tr; hipEventElapsedTime(&tr, e1, e2); if (tr < best) best = tr;
        }
        printf("%6zu  %9.0f  %9.0f  %9.0f  %9.0f  %9.0f  %9.0f\n", mb, res[0], res[1], res[2], res[3], res[4], mb / 1024.0 / (best * 1e-3));
    }
    return 0;
}
