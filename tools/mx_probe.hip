// Lab probe (not product): operand / scale layout and issue rate of v_mfma_scale_f32_16x16x128_f8f6f4 on gfx950.
//   hipcc --offload-arch=gfx950 -O3 tools/mx_probe.hip -o tools/mx_probe && ./tools/mx_probe
// (1) exact-integer check of the lane maps the MX GEMM (csrc/gemm_mx.hip) relies on:
//       A: lane l holds A[row l&15][k = 32 (l>>4) + j], j = 0..31 (8 VGPRs, byte j of the 32-byte fragment)
//       B: lane l holds B[k = 32 (l>>4) + j][col l&15]
//       scale_a byte (op_sel) of lane l = E8M0 scale of A row l&15, K block l>>4; scale_b likewise for B column l&15
//       C/D: col = l&15, row = 4 (l>>4) + reg   (dtype independent)
//     alternatives are tried and reported if the first hypothesis fails.
// (2) issue rate: scaled fp8 K=128 vs bf16 16x16x32, one wave per SIMD, 4 independent accumulators.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int OPSEL>
__global__ void mx_one(const v8i* a, const v8i* b, const int* sa, const int* sb, v4f* c) {
    const int l = threadIdx.x;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], acc, 0, 0, OPSEL, sa[l], OPSEL, sb[l]);
    c[l] = acc;
}

__global__ void __launch_bounds__(256) rate_mx(v4f* out, int iters) {
    const int l = threadIdx.x & 63;
    v8i a, b;
    for (int i = 0; i < 8; ++i) { a[i] = 0x38303c40 + l * 0x01010101 * (i & 1); b[i] = 0x3c383040 ^ (l << 3); }
    v4f acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    const int s = 127 | (127 << 8) | (127 << 16) | (127 << 24);
    for (int it = 0; it < iters; ++it) {
        acc0 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc0, 0, 0, 0, s, 0, s);
        acc1 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc1, 0, 0, 0, s, 0, s);
        acc2 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc2, 0, 0, 0, s, 0, s);
        acc3 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc3, 0, 0, 0, s, 0, s);
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc0 + acc1 + acc2 + acc3;
}

__global__ void __launch_bounds__(256) rate_bf16(v4f* out, int iters) {
    const int l = threadIdx.x & 63;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.37f * (l + i) - 9.f); b[i] = (__bf16)(1.5f - 0.11f * (l ^ i)); }
    v4f acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    for (int it = 0; it < iters; ++it) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc3, 0, 0, 0);
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc0 + acc1 + acc2 + acc3;
}

static const float kVals[9] = {0.f, 0.5f, 1.f, 1.5f, 2.f, -0.5f, -1.f, -1.5f, -2.f};
static const uint8_t kEnc[9] = {0x00, 0x30, 0x38, 0x3C, 0x40, 0xB0, 0xB8, 0xBC, 0xC0};

// hypothesis h: k index of byte j of lane group g (= lane >> 4)
static int kmap(int h, int g, int j) {
    if (h == 0) return 32 * g + j;                              // 32 consecutive k per lane
    if (h == 1) return (j < 16 ? 0 : 64) + 16 * g + (j & 15);   // two K=64 halves, 16 per lane each
    return 8 * g + (j & 7) + 32 * (j >> 3);                     // four K=32 quarters, 8 per lane each
}

int main() {
    int dev_count = 0;
    CK(hipGetDeviceCount(&dev_count));
    srand(7);
    std::vector<int> Ai(16 * 128), Bi(128 * 16), SA(16 * 4), SB(16 * 4);
    for (auto& v : Ai) v = rand() % 9;
    for (auto& v : Bi) v = rand() % 9;
    for (auto& v : SA) v = 125 + rand() % 5;
    for (auto& v : SB) v = 125 + rand() % 5;
    v8i *da, *db; int *dsa, *dsb; v4f* dc;
    CK(hipMalloc(&da, 64 * 32)); CK(hipMalloc(&db, 64 * 32)); CK(hipMalloc(&dsa, 256)); CK(hipMalloc(&dsb, 256)); CK(hipMalloc(&dc, 64 * 16));
    int found = -1;
    for (int h = 0; h < 3 && found < 0; ++h) {
        for (int scale_mode = 0; scale_mode < 2 && found < 0; ++scale_mode) {
            // scale_mode 0: block of the lane's own group g (with h = 0 the lane's 32 bytes ARE block g);
            // scale_mode 1: scale of block (k / 32) differs per byte -> only representable if lanes hold one block; skip unless h == 0
            if (scale_mode == 1) continue;
            std::vector<uint8_t> A8(64 * 32), B8(64 * 32);
            std::vector<int> sa(64), sb(64);
            for (int l = 0; l < 64; ++l) {
                const int r = l & 15, g = l >> 4;
                for (int j = 0; j < 32; ++j) {
                    const int k = kmap(h, g, j);
                    A8[l * 32 + j] = kEnc[Ai[r * 128 + k]];
                    B8[l * 32 + j] = kEnc[Bi[k * 16 + r]];
                }
                sa[l] = SA[r * 4 + g]; sb[l] = SB[r * 4 + g];
            }
            // reference under hypothesis h: the scale of element (row, k) is the one of the lane that holds it
            std::vector<double> ref(256, 0.0);
            for (int i = 0; i < 16; ++i)
                for (int n = 0; n < 16; ++n) {
                    double s = 0;
                    for (int g = 0; g < 4; ++g)
                        for (int j = 0; j < 32; ++j) {
                            const int k = kmap(h, g, j);
                            s += (double)kVals[Ai[i * 128 + k]] * ldexp(1.0, SA[i * 4 + g] - 127) * (double)kVals[Bi[k * 16 + n]] * ldexp(1.0, SB[n * 4 + g] - 127);
                        }
                    ref[i * 16 + n] = s;
                }
            CK(hipMemcpy(da, A8.data(), 64 * 32, hipMemcpyHostToDevice));
            CK(hipMemcpy(db, B8.data(), 64 * 32, hipMemcpyHostToDevice));
            CK(hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice));
            CK(hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice));
            mx_one<0><<<1, 64>>>(da, db, dsa, dsb, dc);
            CK(hipDeviceSynchronize());
            std::vector<float> C(256);
            CK(hipMemcpy(C.data(), dc, 1024, hipMemcpyDeviceToHost));
            int bad = 0;
            for (int l = 0; l < 64; ++l)
                for (int r = 0; r < 4; ++r) {
                    const int row = 4 * (l >> 4) + r, col = l & 15;
                    if ((double)C[l * 4 + r] != ref[row * 16 + col]) ++bad;
                }
            printf("hypothesis %d: %d / 256 outputs differ\n", h, bad);
            if (bad == 0) found = h;
            if (bad == 0) {
                // op_sel: scale in byte 1 / 2 / 3 of the scale register
                for (int os = 1; os < 4; ++os) {
                    std::vector<int> sa2(64), sb2(64);
                    for (int l = 0; l < 64; ++l) { sa2[l] = (sa[l] << (8 * os)) | (0x55 * (os != 0) & 0xFF); sb2[l] = (sb[l] << (8 * os)) | 0x11; }
                    CK(hipMemcpy(dsa, sa2.data(), 256, hipMemcpyHostToDevice));
                    CK(hipMemcpy(dsb, sb2.data(), 256, hipMemcpyHostToDevice));
                    if (os == 1) mx_one<1><<<1, 64>>>(da, db, dsa, dsb, dc);
                    if (os == 2) mx_one<2><<<1, 64>>>(da, db, dsa, dsb, dc);
                    if (os == 3) mx_one<3><<<1, 64>>>(da, db, dsa, dsb, dc);
                    CK(hipDeviceSynchronize());
                    CK(hipMemcpy(C.data(), dc, 1024, hipMemcpyDeviceToHost));
                    int bad2 = 0;
                    for (int l = 0; l < 64; ++l)
                        for (int r = 0; r < 4; ++r)
                            if ((double)C[l * 4 + r] != ref[(4 * (l >> 4) + r) * 16 + (l & 15)]) ++bad2;
                    printf("  op_sel %d (scale in byte %d): %d / 256 differ\n", os, os, bad2);
                }
            }
        }
    }
    printf("LAYOUT %s (hypothesis %d)\n", found >= 0 ? "CONFIRMED" : "UNKNOWN", found);

    // ---- issue rate ----
    v4f* dout;
    CK(hipMalloc(&dout, 256 * 256 * 16));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000;
    for (int which = 0; which < 2; ++which) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            if (which == 0) rate_mx<<<256, 256>>>(dout, iters); else rate_bf16<<<256, 256>>>(dout, iters);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double flop = 256.0 * 4 * iters * 4 * 2.0 * 16 * 16 * (which == 0 ? 128 : 32);
            printf("%s: %.3f ms -> %.1f TFLOP/s (zero-ish data, registers only)\n", which == 0 ? "mx fp8 16x16x128" : "bf16 16x16x32", ms, flop / ms / 1e9);
        }
    }
    return found >= 0 ? 0 : 2;
}
