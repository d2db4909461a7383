// Lab probe (not product): DISCOVERS the operand / scale lane maps of v_mfma_scale_f32_16x16x128_f8f6f4 (fp8 e4m3 x e4m3)
// with one-hot operands, assuming nothing but the C/D map (col = lane & 15, row = 4 (lane >> 4) + reg).
//   hipcc --offload-arch=gfx950 -O3 tools/mx_probe2.hip -o tools/build/mx_probe2
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// one MFMA per block; operands of block p at a + p * 64 etc.
__global__ void mx_many(const v8i* a, const v8i* b, const int* sa, const int* sb, v4f* c) {
    const int l = threadIdx.x, p = blockIdx.x;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[p * 64 + l], b[p * 64 + l], acc, 0, 0, 0, sa[p * 64 + l], 0, sb[p * 64 + l]);
    c[p * 64 + l] = acc;
}

static float e4m3_decode(uint8_t v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float x = e == 0 ? ldexpf((float)m / 8.f, -6) : ldexpf(1.f + (float)m / 8.f, e - 7);
    return s ? -x : x;
}

int main() {
    // 128 distinct non-zero, non-NaN codes with exponent >= 1 (normals only)
    std::vector<uint8_t> code;
    for (int v = 0x08; v < 0x7F && code.size() < 64; ++v) code.push_back((uint8_t)v);
    for (int v = 0x88; v < 0xFF && code.size() < 128; ++v) code.push_back((uint8_t)v);
    const int P1 = 128;                     // probe p = (g, j): A byte j of every lane in group g = 1.0
    std::vector<uint8_t> A((size_t)P1 * 64 * 32, 0), B((size_t)P1 * 64 * 32, 0);
    std::vector<int> S((size_t)P1 * 64, 127);
    for (int p = 0; p < P1; ++p) {
        const int g = p >> 5, j = p & 31;
        for (int l = 0; l < 64; ++l) {
            if ((l >> 4) == g) A[((size_t)p * 64 + l) * 32 + j] = 0x38;
            for (int jb = 0; jb < 32; ++jb) B[((size_t)p * 64 + l) * 32 + jb] = code[(l >> 4) * 32 + jb];
        }
    }
    v8i *da, *db; int *ds; v4f* dc;
    CK(hipMalloc(&da, A.size())); CK(hipMalloc(&db, B.size())); CK(hipMalloc(&ds, S.size() * 4)); CK(hipMalloc(&dc, (size_t)P1 * 64 * 16));
    CK(hipMemcpy(da, A.data(), A.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(db, B.data(), B.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(ds, S.data(), S.size() * 4, hipMemcpyHostToDevice));
    mx_many<<<P1, 64>>>(da, db, ds, ds, dc);
    CK(hipDeviceSynchronize());
    std::vector<float> C((size_t)P1 * 256);
    CK(hipMemcpy(C.data(), dc, C.size() * 4, hipMemcpyDeviceToHost));
    // C[p][l][r] -> row 4 (l >> 4) + r, col l & 15
    int identity = 1, rows_ok = 1;
    std::vector<int> pair(128, -1);
    for (int p = 0; p < P1; ++p) {
        // all 16 rows x 16 cols should hold the value of ONE B element (gb, jb): decode from column 0 of row 0
        const float v00 = C[(size_t)p * 256 + 0 * 4 + 0];
        int hit = -1;
        for (int q = 0; q < 128; ++q) if (e4m3_decode(code[q]) == v00) hit = q;
        pair[p] = hit;
        if (hit != p) identity = 0;
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 4; ++r)
                if (C[(size_t)p * 256 + l * 4 + r] != v00) rows_ok = 0;
    }
    printf("A one-hot (g, j) in every row, B distinct codes: uniform output over the 16x16 tile: %s; pairing identity: %s\n", rows_ok ? "yes" : "NO", identity ? "yes" : "NO");
    if (!identity) {
        printf("pairing table A(g,j) -> B(g,j):\n");
        for (int p = 0; p < P1; ++p) printf("  A(%d,%2d) -> %s(%d,%2d)  value %g\n", p >> 5, p & 31, pair[p] < 0 ? "?" : "B", pair[p] >> 5, pair[p] & 31, C[(size_t)p * 256]);
    }
    // row map of A: one-hot in ONE lane only (group g, byte 0), B all ones
    {
        std::vector<uint8_t> A2((size_t)64 * 64 * 32, 0), B2((size_t)64 * 64 * 32, 0x38);
        for (int p = 0; p < 64; ++p) A2[((size_t)p * 64 + p) * 32 + 0] = 0x38;
        v8i *da2, *db2; int* ds2; v4f* dc2;
        std::vector<int> S2(64 * 64, 127);
        CK(hipMalloc(&da2, A2.size())); CK(hipMalloc(&db2, B2.size())); CK(hipMalloc(&ds2, S2.size() * 4)); CK(hipMalloc(&dc2, 64 * 64 * 16));
        CK(hipMemcpy(da2, A2.data(), A2.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(db2, B2.data(), B2.size(), hipMemcpyHostToDevice));
        CK(hipMemcpy(ds2, S2.data(), S2.size() * 4, hipMemcpyHostToDevice));
        mx_many<<<64, 64>>>(da2, db2, ds2, ds2, dc2);
        CK(hipDeviceSynchronize());
        std::vector<float> C2(64 * 256);
        CK(hipMemcpy(C2.data(), dc2, C2.size() * 4, hipMemcpyDeviceToHost));
        int ok = 1;
        for (int p = 0; p < 64; ++p) {
            int row = -1, nrows = 0;
            for (int i = 0; i < 16; ++i) {
                const float v = C2[p * 256 + ((i >> 2) * 16 + 0) * 4 + (i & 3)];     // row i, col 0
                if (v != 0.f) { row = i; ++nrows; }
            }
            if (row != (p & 15) || nrows != 1) { ok = 0; printf("  A lane %d -> row %d (%d rows hit)\n", p, row, nrows); }
        }
        printf("A lane l holds row l & 15: %s\n", ok ? "yes" : "NO");
        // same for B: one-hot lane of B, A all ones -> column
        std::vector<uint8_t> A3((size_t)64 * 64 * 32, 0x38), B3((size_t)64 * 64 * 32, 0);
        for (int p = 0; p < 64; ++p) B3[((size_t)p * 64 + p) * 32 + 0] = 0x38;
        CK(hipMemcpy(da2, A3.data(), A3.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(db2, B3.data(), B3.size(), hipMemcpyHostToDevice));
        mx_many<<<64, 64>>>(da2, db2, ds2, ds2, dc2);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(C2.data(), dc2, C2.size() * 4, hipMemcpyDeviceToHost));
        ok = 1;
        for (int p = 0; p < 64; ++p) {
            int col = -1, n = 0;
            for (int c = 0; c < 16; ++c) if (C2[p * 256 + c * 4 + 0] != 0.f) { col = c; ++n; }     // row 0, col c
            if (col != (p & 15) || n != 1) { ok = 0; printf("  B lane %d -> col %d (%d cols hit)\n", p, col, n); }
        }
        printf("B lane l holds column l & 15: %s\n", ok ? "yes" : "NO");
    }
    // scale map of A: scale lane L doubled; A one-hot (g, j) in every row; B ones.  probes: L x (g, j coarse: j in {0, 8, 16, 24, 31})
    {
        const int js[5] = {0, 8, 16, 24, 31};
        const int NP = 64 * 4 * 5;
        std::vector<uint8_t> A4((size_t)NP * 64 * 32, 0), B4((size_t)NP * 64 * 32, 0x38);
        std::vector<int> SA((size_t)NP * 64, 127), SB((size_t)NP * 64, 127);
        for (int L = 0; L < 64; ++L)
            for (int g = 0; g < 4; ++g)
                for (int jj = 0; jj < 5; ++jj) {
                    const int p = (L * 4 + g) * 5 + jj;
                    for (int l = 0; l < 64; ++l) if ((l >> 4) == g) A4[((size_t)p * 64 + l) * 32 + js[jj]] = 0x38;
                    SA[(size_t)p * 64 + L] = 128;
                }
        v8i *da4, *db4; int *dsa4, *dsb4; v4f* dc4;
        CK(hipMalloc(&da4, A4.size())); CK(hipMalloc(&db4, B4.size())); CK(hipMalloc(&dsa4, SA.size() * 4)); CK(hipMalloc(&dsb4, SB.size() * 4));
        CK(hipMalloc(&dc4, (size_t)NP * 64 * 16));
        CK(hipMemcpy(da4, A4.data(), A4.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(db4, B4.data(), B4.size(), hipMemcpyHostToDevice));
        CK(hipMemcpy(dsa4, SA.data(), SA.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dsb4, SB.data(), SB.size() * 4, hipMemcpyHostToDevice));
        mx_many<<<NP, 64>>>(da4, db4, dsa4, dsb4, dc4);
        CK(hipDeviceSynchronize());
        std::vector<float> C4((size_t)NP * 256);
        CK(hipMemcpy(C4.data(), dc4, C4.size() * 4, hipMemcpyDeviceToHost));
        int expected = 1;
        for (int L = 0; L < 64; ++L) {
            char line[512]; int n = 0;
            n += snprintf(line + n, sizeof(line) - n, "  scale_a lane %2d scales:", L);
            int any = 0;
            for (int g = 0; g < 4; ++g)
                for (int jj = 0; jj < 5; ++jj) {
                    const int p = (L * 4 + g) * 5 + jj;
                    for (int i = 0; i < 16; ++i) {
                        const float v = C4[(size_t)p * 256 + ((i >> 2) * 16 + 0) * 4 + (i & 3)];
                        if (v == 2.f) {
                            any = 1;
                            if (!(i == (L & 15) && g == (L >> 4))) expected = 0;
                            if (n < 480) n += snprintf(line + n, sizeof(line) - n, " (row %d g %d j %d)", i, g, js[jj]);
                        } else if (v != 1.f) { expected = 0; if (n < 480) n += snprintf(line + n, sizeof(line) - n, " [row %d g %d j %d = %g]", i, g, js[jj], v); }
                    }
                }
            if (!any) { expected = 0; }
            if (L < 4 || L == 17 || L == 63 || !any) printf("%s%s\n", line, any ? "" : " NOTHING");
        }
        printf("scale_a lane L applies to (row L & 15, the 32 bytes of lane group L >> 4): %s\n", expected ? "yes" : "NO");
        // scale_b the same way (swap roles): B one-hot in every column, scale_b lane L doubled
        std::fill(A4.begin(), A4.end(), 0x38); std::fill(B4.begin(), B4.end(), 0);
        std::fill(SA.begin(), SA.end(), 127); std::fill(SB.begin(), SB.end(), 127);
        for (int L = 0; L < 64; ++L)
            for (int g = 0; g < 4; ++g)
                for (int jj = 0; jj < 5; ++jj) {
                    const int p = (L * 4 + g) * 5 + jj;
                    for (int l = 0; l < 64; ++l) if ((l >> 4) == g) B4[((size_t)p * 64 + l) * 32 + js[jj]] = 0x38;
                    SB[(size_t)p * 64 + L] = 128;
                }
        CK(hipMemcpy(da4, A4.data(), A4.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(db4, B4.data(), B4.size(), hipMemcpyHostToDevice));
        CK(hipMemcpy(dsa4, SA.data(), SA.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dsb4, SB.data(), SB.size() * 4, hipMemcpyHostToDevice));
        mx_many<<<NP, 64>>>(da4, db4, dsa4, dsb4, dc4);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(C4.data(), dc4, C4.size() * 4, hipMemcpyDeviceToHost));
        expected = 1;
        for (int L = 0; L < 64; ++L)
            for (int g = 0; g < 4; ++g)
                for (int jj = 0; jj < 5; ++jj) {
                    const int p = (L * 4 + g) * 5 + jj;
                    for (int c = 0; c < 16; ++c) {
                        const float v = C4[(size_t)p * 256 + c * 4 + 0];           // row 0, col c
                        const float want = (c == (L & 15) && g == (L >> 4)) ? 2.f : 1.f;
                        if (v != want) { if (expected) printf("  scale_b lane %d: col %d g %d j %d -> %g (want %g)\n", L, c, g, js[jj], v, want); expected = 0; }
                    }
                }
        printf("scale_b lane L applies to (column L & 15, the 32 bytes of lane group L >> 4): %s\n", expected ? "yes" : "NO");
    }
    return 0;
}
