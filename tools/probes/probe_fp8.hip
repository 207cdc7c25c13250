// Probe (run on the GPU box): fragment layouts of the gfx950 fp8 instructions the fp8 backward storage relies on.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/probe_fp8.hip -o tools/probes/probe_fp8.bin && tools/probes/probe_fp8.bin
//   1. ds_read_b64_tr_b8      : which LDS byte lands in which (lane, byte) of the result
//   2. v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3 x e4m3): (lane, byte) -> (row, k) of A, (k, col) of B; scale operand
//   3. v_cvt_scalef32_pk_fp8_{bf16,f32}: direction of the scale, saturation, rounding
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((ext_vector_type(2))) int i32x2;
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) short s16x2;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void k_tr8(const uint8_t* img, const int* addr, uint32_t* out) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[4096];
    const int lane = threadIdx.x;
    for (int i = lane; i < 4096; i += 64) lds[i] = img[i];
    __syncthreads();
    i32x2 v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) i32x2*)(lds + addr[lane]));
    out[2 * lane] = (uint32_t)v[0];
    out[2 * lane + 1] = (uint32_t)v[1];
}

// D = A*B with per-lane raw register images; scales given per lane
__global__ void k_mfma(const uint32_t* a_img, const uint32_t* b_img, const uint32_t* sa, const uint32_t* sb, float* d) {
    const int lane = threadIdx.x;
    i32x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (int)a_img[lane * 8 + i]; b[i] = (int)b_img[lane * 8 + i]; }
    f32x16 c = {};
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, (int)sa[lane], 0, (int)sb[lane]);
    for (int r = 0; r < 16; ++r) d[lane * 16 + r] = c[r];
}

// exploration: for every (la, ba) one-hot in A (value 1.0 = 0x38) find (1) its row, (2) which B (lane in {0,32}, byte) one-hot
// makes D[row][0] non-zero.  Same for B against A candidates in lanes {0, 32}.
__global__ void k_explore(int* rowA, int* kA, int* colB, int* kB) {
    const int lane = threadIdx.x;
    const int s1 = 127;   // e8m0 1.0
    for (int la = 0; la < 64; ++la)
        for (int ba = 0; ba < 32; ++ba) {
            i32x8 a = {}, b;
            if (lane == la) a[ba >> 2] = 0x38 << (8 * (ba & 3));
            for (int i = 0; i < 8; ++i) b[i] = 0x38383838;
            f32x16 c = {};
            c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, s1, 0, s1);
            // D layout: lane -> col = lane&31, rows (r&3) + 8*(r>>2) + 4*(lane>>5)
            if ((lane & 31) == 0) {
                for (int r = 0; r < 16; ++r)
                    if (c[r] != 0.0f) rowA[la * 32 + ba] = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            }
            int found = -1;
            for (int cand = 0; cand < 64; ++cand) {
                const int lb = (cand >> 5) * 32, bb = cand & 31;
                i32x8 b1 = {};
                if (lane == lb) b1[bb >> 2] = 0x38 << (8 * (bb & 3));
                f32x16 c1 = {};
                c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b1, c1, 0, 0, 0, s1, 0, s1);
                float s = 0.f;
                for (int r = 0; r < 16; ++r) s += c1[r];
                const unsigned long long m = __ballot(s != 0.0f);
                if (m) found = cand;
            }
            if (lane == 0) kA[la * 32 + ba] = found;
        }
    for (int lb = 0; lb < 64; ++lb)
        for (int bb = 0; bb < 32; ++bb) {
            i32x8 a, b = {};
            if (lane == lb) b[bb >> 2] = 0x38 << (8 * (bb & 3));
            for (int i = 0; i < 8; ++i) a[i] = 0x38383838;
            f32x16 c = {};
            c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, s1, 0, s1);
            float s = 0.f;
            for (int r = 0; r < 16; ++r) s += c[r];
            const unsigned long long m = __ballot(s != 0.0f);
            if (lane == 0) colB[lb * 32 + bb] = m ? (__ffsll((long long)m) - 1) & 31 : -1;
            int found = -1;
            for (int cand = 0; cand < 64; ++cand) {
                const int la = (cand >> 5) * 32, ba = cand & 31;
                i32x8 a1 = {};
                if (lane == la) a1[ba >> 2] = 0x38 << (8 * (ba & 3));
                f32x16 c1 = {};
                c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a1, b, c1, 0, 0, 0, s1, 0, s1);
                float s2 = 0.f;
                for (int r = 0; r < 16; ++r) s2 += c1[r];
                const unsigned long long m2 = __ballot(s2 != 0.0f);
                if (m2) found = cand;
            }
            if (lane == 0) kB[lb * 32 + bb] = found;
        }
}

// which scale lane-half (k block) governs element (lane, byte)?  one-hot A (or B) against all-ones, scale x2 from lanes
// 0-31 and x0.5 from lanes 32-63 on the probed operand: the single product comes out as 2 (block 0) or 0.5 (block 1)
__global__ void k_block(float* blkA, float* blkB) {
    const int lane = threadIdx.x;
    const int sprobe = lane < 32 ? 128 : 126, s1 = 127;
    for (int half = 0; half < 2; ++half)
        for (int bq = 0; bq < 32; ++bq) {
            const int lq = 32 * half + 5;                    // row / col 5
            i32x8 hot = {}, ones;
            if (lane == lq) hot[bq >> 2] = 0x38 << (8 * (bq & 3));
            for (int i = 0; i < 8; ++i) ones[i] = 0x38383838;
            f32x16 c = {};
            c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(hot, ones, c, 0, 0, 0, sprobe, 0, s1);
            float s = 0.f;
            for (int r = 0; r < 16; ++r) s += c[r];          // lanes of col j hold D[*][j]: only row 5 is non-zero
            if (lane == 0) blkA[half * 32 + bq] = s;
            f32x16 c2 = {};
            c2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ones, hot, c2, 0, 0, 0, s1, 0, sprobe);
            float s2 = 0.f;
            for (int r = 0; r < 16; ++r) s2 += c2[r];
            const float tot = __shfl(s2, 5, 64) + __shfl(s2, 37, 64);   // col 5 lives in lanes 5 and 37
            if (lane == 0) blkB[half * 32 + bq] = tot;
        }
}

__global__ void k_cvt(const float* x, const float* scale, int n, int ns, uint32_t* out_bf, uint32_t* out_f32) {
    const int i = threadIdx.x;
    if (i >= n) return;
    for (int s = 0; s < ns; ++s) {
        bf16x2 v;
        v[0] = (__bf16)x[i];
        v[1] = (__bf16)(-x[i]);
        s16x2 old = {0, 0};
        s16x2 r = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(old, v, scale[s], false);
        out_bf[s * n + i] = (uint16_t)r[0] | ((uint32_t)(uint16_t)r[1] << 16);
        s16x2 r2 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(old, x[i], -x[i], scale[s], true);
        out_f32[s * n + i] = (uint16_t)r2[0] | ((uint32_t)(uint16_t)r2[1] << 16);
    }
}

static float e4m3_to_float(uint8_t v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float f;
    if (e == 0) f = ldexpf((float)m, -9);
    else if (e == 15 && m == 7) f = NAN;
    else f = ldexpf(1.0f + m / 8.0f, e - 7);
    return s ? -f : f;
}
static uint8_t float_to_e4m3_small_int(int v) {   // exact for 0..16
    for (int b = 0; b < 128; ++b)
        if (e4m3_to_float((uint8_t)b) == (float)v) return (uint8_t)b;
    return 0;
}

int main() {
    // ---- 1. tr_b8 -------------------------------------------------------------------------------------------------
    {
        std::vector<uint8_t> img(4096);
        std::vector<int> addr(64);
        uint8_t* d_img; int* d_addr; uint32_t* d_out;
        CK(hipMalloc(&d_img, 4096)); CK(hipMalloc(&d_addr, 256)); CK(hipMalloc(&d_out, 512));
        std::vector<uint32_t> lo(128), hi(128);
        for (int l = 0; l < 64; ++l) addr[l] = 8 * l;
        CK(hipMemcpy(d_addr, addr.data(), 256, hipMemcpyHostToDevice));
        for (int pass = 0; pass < 2; ++pass) {
            for (int i = 0; i < 4096; ++i) img[i] = pass ? (uint8_t)(i >> 8) : (uint8_t)(i & 255);
            CK(hipMemcpy(d_img, img.data(), 4096, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(k_tr8, dim3(1), dim3(64), 0, 0, d_img, d_addr, d_out);
            CK(hipMemcpy(pass ? hi.data() : lo.data(), d_out, 512, hipMemcpyDeviceToHost));
        }
        printf("== ds_read_b64_tr_b8, lane l reads LDS address 8*l: result (lane, byte) <- LDS byte index\n");
        for (int l = 0; l < 64; ++l) {
            printf("lane %2d:", l);
            for (int b = 0; b < 8; ++b) {
                const int v = ((lo[2 * l + (b >> 2)] >> (8 * (b & 3))) & 255) | (((hi[2 * l + (b >> 2)] >> (8 * (b & 3))) & 255) << 8);
                printf(" %4d", v);
            }
            printf("\n");
        }
        // second geometry: lane l reads address 64*(l&15) + 8*(l>>4)  (row pitch 64 B, 16 rows per group)
        for (int l = 0; l < 64; ++l) addr[l] = 64 * (l & 15) + 8 * (l >> 4) + 1024 * 0;
        CK(hipMemcpy(d_addr, addr.data(), 256, hipMemcpyHostToDevice));
        for (int pass = 0; pass < 2; ++pass) {
            for (int i = 0; i < 4096; ++i) img[i] = pass ? (uint8_t)(i >> 8) : (uint8_t)(i & 255);
            CK(hipMemcpy(d_img, img.data(), 4096, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(k_tr8, dim3(1), dim3(64), 0, 0, d_img, d_addr, d_out);
            CK(hipMemcpy(pass ? hi.data() : lo.data(), d_out, 512, hipMemcpyDeviceToHost));
        }
        printf("== ds_read_b64_tr_b8, lane l reads LDS address 64*(l&15) + 8*(l>>4)\n");
        for (int l = 0; l < 64; ++l) {
            printf("lane %2d:", l);
            for (int b = 0; b < 8; ++b) {
                const int v = ((lo[2 * l + (b >> 2)] >> (8 * (b & 3))) & 255) | (((hi[2 * l + (b >> 2)] >> (8 * (b & 3))) & 255) << 8);
                printf(" %4d", v);
            }
            printf("\n");
        }
    }
    // ---- 2. scaled MFMA --------------------------------------------------------------------------------------------
    {
        int *rowA, *kA, *colB, *kB;
        CK(hipMalloc(&rowA, 2048 * 4)); CK(hipMalloc(&kA, 2048 * 4)); CK(hipMalloc(&colB, 2048 * 4)); CK(hipMalloc(&kB, 2048 * 4));
        CK(hipMemset(rowA, 0xff, 2048 * 4));
        hipLaunchKernelGGL(k_explore, dim3(1), dim3(64), 0, 0, rowA, kA, colB, kB);
        CK(hipDeviceSynchronize());
        std::vector<int> hr(2048), hk(2048), hc(2048), hkb(2048);
        CK(hipMemcpy(hr.data(), rowA, 8192, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hk.data(), kA, 8192, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hc.data(), colB, 8192, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hkb.data(), kB, 8192, hipMemcpyDeviceToHost));
        printf("== mfma_scale_f32_32x32x64_f8f6f4 (e4m3): A (lane, byte) -> row, matching B slot (cand = 32*(lane>>5) + byte of B lanes {0,32})\n");
        int okA = 1, okB = 1;
        for (int la = 0; la < 64; ++la) {
            for (int ba = 0; ba < 32; ++ba) {
                const int row = hr[la * 32 + ba], k = hk[la * 32 + ba];
                if (row != (la & 31) || k != 32 * (la >> 5) + ba) okA = 0;
            }
        }
        for (int lb = 0; lb < 64; ++lb)
            for (int bb = 0; bb < 32; ++bb)
                if (hc[lb * 32 + bb] != (lb & 31) || hkb[lb * 32 + bb] != 32 * (lb >> 5) + bb) okB = 0;
        printf("hypothesis A: row = lane&31, k = 32*(lane>>5) + byte : %s\n", okA ? "CONFIRMED" : "REFUTED");
        printf("hypothesis B: col = lane&31, k = 32*(lane>>5) + byte : %s\n", okB ? "CONFIRMED" : "REFUTED");
        if (!okA || !okB) {
            for (int la = 0; la < 64; la += 1) {
                printf("A lane %2d:", la);
                for (int ba = 0; ba < 32; ++ba) printf(" (%d,%d)", hr[la * 32 + ba], hk[la * 32 + ba]);
                printf("\n");
            }
            for (int lb = 0; lb < 64; lb += 1) {
                printf("B lane %2d:", lb);
                for (int bb = 0; bb < 32; ++bb) printf(" (%d,%d)", hc[lb * 32 + bb], hkb[lb * 32 + bb]);
                printf("\n");
            }
        }
        {
            float *bA, *bB;
            CK(hipMalloc(&bA, 256)); CK(hipMalloc(&bB, 256));
            hipLaunchKernelGGL(k_block, dim3(1), dim3(64), 0, 0, bA, bB);
            std::vector<float> ha(64), hb(64);
            CK(hipMemcpy(ha.data(), bA, 256, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), bB, 256, hipMemcpyDeviceToHost));
            printf("== scale block of element (lane half, byte): product value with scale x2 from lanes 0-31 / x0.5 from lanes 32-63\n");
            for (int half = 0; half < 2; ++half) {
                printf("A half %d:", half);
                for (int b = 0; b < 32; ++b) printf(" %g", ha[half * 32 + b]);
                printf("\nB half %d:", half);
                for (int b = 0; b < 32; ++b) printf(" %g", hb[half * 32 + b]);
                printf("\n");
            }
        }
        // scale semantics: all ones; scale A lanes 0-31 = 128 (x2), lanes 32-63 = 126 (x0.5); scale B = 127, then B = 129
        std::vector<uint32_t> a(512, 0x38383838u), b(512, 0x38383838u), sa(64), sb(64);
        uint32_t *da, *db, *dsa, *dsb; float* dd;
        CK(hipMalloc(&da, 2048)); CK(hipMalloc(&db, 2048)); CK(hipMalloc(&dsa, 256)); CK(hipMalloc(&dsb, 256)); CK(hipMalloc(&dd, 4096));
        std::vector<float> d(1024);
        for (int t = 0; t < 3; ++t) {
            for (int l = 0; l < 64; ++l) {
                sa[l] = (t == 2) ? (127u | (130u << 8)) : ((l < 32) ? 128u : 126u);
                sb[l] = (t == 1) ? 129u : 127u;
            }
            CK(hipMemcpy(da, a.data(), 2048, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), 2048, hipMemcpyHostToDevice));
            CK(hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd);
            CK(hipMemcpy(d.data(), dd, 4096, hipMemcpyDeviceToHost));
            printf("scale test %d: D[0][0] = %g  D(lane 33, reg 5) = %g   (t0 expect 32*2 + 32*0.5 = 80 if the scale of lane l covers its own 32 k; t1 x4; t2: byte0 = 127 -> 64)\n",
                   t, d[0], d[33 * 16 + 5]);
        }
        // random small-integer GEMM against the hypothesised layout, distinct per-lane scales
        srand(1);
        std::vector<int> A(32 * 64), B(64 * 32);
        for (auto& v : A) v = rand() % 5;
        for (auto& v : B) v = rand() % 5;
        for (int l = 0; l < 64; ++l) {
            for (int w = 0; w < 8; ++w) {
                uint32_t wa = 0, wb = 0;
                for (int q = 0; q < 4; ++q) {
                    const int k = 32 * (l >> 5) + 4 * w + q;
                    wa |= (uint32_t)float_to_e4m3_small_int(A[(l & 31) * 64 + k]) << (8 * q);
                    wb |= (uint32_t)float_to_e4m3_small_int(B[k * 32 + (l & 31)]) << (8 * q);
                }
                a[l * 8 + w] = wa; b[l * 8 + w] = wb;
            }
            sa[l] = 127u + (l >> 5);      // k block 1 of A scaled x2
            sb[l] = 127u;
        }
        CK(hipMemcpy(da, a.data(), 2048, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), 2048, hipMemcpyHostToDevice));
        CK(hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd);
        CK(hipMemcpy(d.data(), dd, 4096, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 16; ++r) {
                const int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
                float ref = 0;
                for (int k = 0; k < 64; ++k) ref += (float)(A[row * 64 + k] * B[k * 32 + col]) * (k >= 32 ? 2.0f : 1.0f);
                if (ref != d[l * 16 + r]) ++bad;
            }
        printf("random integer GEMM with the hypothesised layout + per-block scale: %d mismatches of 1024\n", bad);
        for (int l = 0; l < 64; l += 21)
            for (int r = 0; r < 16; r += 5) {
                const int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
                float ref = 0, ref1 = 0;
                for (int k = 0; k < 64; ++k) {
                    ref += (float)(A[row * 64 + k] * B[k * 32 + col]) * (k >= 32 ? 2.0f : 1.0f);
                    ref1 += (float)(A[row * 64 + k] * B[k * 32 + col]);
                }
                printf("  lane %d reg %d (row %d col %d): got %g  ref(scaled) %g  ref(unscaled) %g\n", l, r, row, col, d[l * 16 + r], ref, ref1);
            }
    }
    // ---- 3. cvt_scalef32 ------------------------------------------------------------------------------------------
    {
        const float xs[] = {1.0f, 3.0f, 0.3f, 500.0f, 1000.0f, 1e-3f, 2.5f, 0.0f, 448.0f, 464.0f, 480.0f, 0.0021f, 17.0f, 18.0f, 19.0f, 1.0625f, 1.1875f};
        const float sc[] = {1.0f, 2.0f, 0.5f, 16.0f, 0.0625f, 3.0f};
        const int n = sizeof(xs) / 4, ns = sizeof(sc) / 4;
        float *dx, *ds; uint32_t *o1, *o2;
        CK(hipMalloc(&dx, n * 4)); CK(hipMalloc(&ds, ns * 4)); CK(hipMalloc(&o1, n * ns * 4)); CK(hipMalloc(&o2, n * ns * 4));
        CK(hipMemcpy(dx, xs, n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(ds, sc, ns * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_cvt, dim3(1), dim3(64), 0, 0, dx, ds, n, ns, o1, o2);
        std::vector<uint32_t> h1(n * ns), h2(n * ns);
        CK(hipMemcpy(h1.data(), o1, n * ns * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2.data(), o2, n * ns * 4, hipMemcpyDeviceToHost));
        printf("== v_cvt_scalef32_pk_fp8_bf16 (word_sel 0) / _f32 (word_sel 1): input (x, -x), per scale: raw dword -> decoded bytes\n");
        for (int s = 0; s < ns; ++s)
            for (int i = 0; i < n; ++i) {
                const uint32_t w1 = h1[s * n + i], w2 = h2[s * n + i];
                printf("scale %-7g x %-8g : bf16-src 0x%08x -> (%g, %g)   f32-src 0x%08x -> (%g, %g)\n", sc[s], xs[i], w1,
                       e4m3_to_float(w1 & 255), e4m3_to_float((w1 >> 8) & 255), w2, e4m3_to_float((w2 >> 16) & 255), e4m3_to_float((w2 >> 24) & 255));
            }
    }
    return 0;
}
