// Hardware probe: lane/element mapping of ds_read_b64_tr_b16 on gfx950 (used to design mlp_bwd_dw).
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe_tr_b16.hip -o /tmp/probe_tr ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void probe(const uint32_t* lane_addr, uint16_t* out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;   // element value == element index
    __syncthreads();
    unsigned addr = (unsigned)(uintptr_t)lds + lane_addr[threadIdx.x];
    uint2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[threadIdx.x * 4 + 0] = v.x & 0xffff; out[threadIdx.x * 4 + 1] = v.x >> 16;
    out[threadIdx.x * 4 + 2] = v.y & 0xffff; out[threadIdx.x * 4 + 3] = v.y >> 16;
}
int main() {
    uint32_t h_addr[64]; uint16_t h_out[256];
    uint32_t* d_addr; uint16_t* d_out;
    hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
    for (int mode = 0; mode < 3; ++mode) {
        for (int l = 0; l < 64; ++l) {
            if (mode == 0) h_addr[l] = l * 8;                       // natural: lane l -> its own 8-byte chunk
            else if (mode == 1) h_addr[l] = 0;                      // uniform address
            else h_addr[l] = (l & 15) * 64 + (l >> 4) * 1024;       // strided chunks: 64 B apart, groups 1 KiB apart
        }
        hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_addr, d_out);
        hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
        printf("mode %d (element index read by lane:elem)\n", mode);
        for (int l = 0; l < 64; ++l) {
            printf("  l%02d: %4u %4u %4u %4u", l, h_out[l * 4], h_out[l * 4 + 1], h_out[l * 4 + 2], h_out[l * 4 + 3]);
            if ((l & 3) == 3) printf("\n");
        }
    }
    return 0;
}
