// Lab microbenchmark: what a GEMM epilogue's store instruction should look like.  One workgroup (4 waves) per CU; every wave writes
// (or reads and writes) its own 128-row x 128-column tile of a [M, ld] matrix with 16-byte accesses per lane, in three lane -> address
// maps.  A: the four-wave GEMM's bf16 map (one instruction = 16 rows x 64 contiguous bytes); B: 8 rows x 128 contiguous bytes (whole
// cache lines); C: the GEMM's fp32 map (16 rows x four 16-byte pieces 32 bytes apart).  Prints bytes per cycle and CU.
//   hipcc -O3 --offload-arch=gfx950 tools/lab/store_pattern.hip -o /tmp/store_pattern && /tmp/store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PAT, bool RMW>
__global__ void __launch_bounds__(256) k(char* base, long ld_bytes, int tiles_per_wave, long* cycles) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long t0 = __builtin_amdgcn_s_memtime();
    f32x4 v = {1.f, 2.f, 3.f, (float)lane};
    for (int t = 0; t < tiles_per_wave; ++t) {
        // the wave's tile: 128 rows; row bytes = 256 (bf16 128 columns) or 512 (fp32 128 columns)
        const long tile = ((long)blockIdx.x * 4 + w) * tiles_per_wave + t;
        char* tb = base + tile * 128 * ld_bytes;
        constexpr int ROWB = PAT == 2 ? 512 : 256;          // bytes of one tile row
        constexpr int NI = 128 * ROWB / 1024;               // instructions per tile
#pragma unroll 8
        for (int i = 0; i < NI; ++i) {
            long off;
            if (PAT == 0) {            // A: 16 rows x 64 B: row = 16 (i / 4) + (lane & 15), 64-byte quarter i & 3, 16-byte chunk lane >> 4
                off = (long)(16 * (i >> 2) + (lane & 15)) * ld_bytes + (i & 3) * 64 + (lane >> 4) * 16;
            } else if (PAT == 1) {     // B: 8 rows x 128 B: row = 8 (i / 2) + (lane & 7), half i & 1, chunk lane >> 3
                off = (long)(8 * (i >> 1) + (lane & 7)) * ld_bytes + (i & 1) * 128 + (lane >> 3) * 16;
            } else {                   // C: fp32 map: row = 16 (i / 8) + (lane & 15), 128-byte quarter (i >> 1) & 3, piece = 32 (lane >> 4) + 16 (i & 1)
                off = (long)(16 * (i >> 3) + (lane & 15)) * ld_bytes + ((i >> 1) & 3) * 128 + (lane >> 4) * 32 + (i & 1) * 16;
            }
            f32x4* p = reinterpret_cast<f32x4*>(tb + off);
            if (RMW) {
                f32x4 r = *p;
                r += v;
                __builtin_nontemporal_store(r, p);
            } else {
                __builtin_nontemporal_store(v, p);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cycles[blockIdx.x * 4 + w] = t1 - t0;
}

template <int PAT, bool RMW>
void run(const char* name, char* buf, long ld_bytes, int grid, int tpw, long* dcy) {
    std::vector<long> h(grid * 4);
    for (int rep = 0; rep < 3; ++rep) {
        k<PAT, RMW><<<grid, 256>>>(buf, ld_bytes, tpw, dcy);
        hipDeviceSynchronize();
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<PAT, RMW><<<grid, 256>>>(buf, ld_bytes, tpw, dcy);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h.data(), dcy, h.size() * sizeof(long), hipMemcpyDeviceToHost);
    std::vector<long> s(h);
    std::sort(s.begin(), s.end());
    const double cyc = (double)s[s.size() / 2];
    const double tile_bytes = 128.0 * (PAT == 2 ? 512 : 256) * (RMW ? 2 : 1);
    const double per_cu = 4.0 * tpw * tile_bytes;
    printf("%-58s grid %3d: %8.1f us, %6.2f TB/s chip-wide, median wave %9.0f cycles -> %5.1f B/cycle/CU\n", name, grid, ms * 1e3,
           grid * per_cu / (ms * 1e-3) / 1e12, cyc, per_cu / cyc);
}

int main() {
    const long ld_bytes = 7680 * 2;          // a row of the QKV output (bf16) / similar stride for the fp32 case
    const int tpw = 8;
    const long bytes = (long)256 * 4 * tpw * 128 * ld_bytes;       // every tile in its own rows (worst case footprint; only 256 / 512 B of a row are touched)
    char* buf;
    long* dcy;
    if (hipMalloc(&buf, bytes) != hipSuccess) { printf("alloc of %ld bytes failed\n", bytes); return 1; }
    hipMalloc(&dcy, 256 * 4 * sizeof(long));
    hipMemset(buf, 0, bytes);
    for (int grid : {256, 64}) {
        run<0, false>("store A: 16 rows x 64 B per instruction (bf16 epilogue)", buf, ld_bytes, grid, tpw, dcy);
        run<1, false>("store B: 8 rows x 128 B per instruction (whole lines)", buf, ld_bytes, grid, tpw, dcy);
        run<2, false>("store C: 16 rows x 4 x 16 B pieces (fp32 epilogue)", buf, ld_bytes, grid, tpw, dcy);
        run<0, true>("rmw   A: 16 rows x 64 B", buf, ld_bytes, grid, tpw, dcy);
        run<1, true>("rmw   B: 8 rows x 128 B", buf, ld_bytes, grid, tpw, dcy);
        run<2, true>("rmw   C: 16 rows x 4 x 16 B pieces (fp32 residual update)", buf, ld_bytes, grid, tpw, dcy);
    }
    return 0;
}
