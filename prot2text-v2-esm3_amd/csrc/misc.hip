// Error string, synthetic fill, cast, transpose, column sums.  All HBM-bound streaming kernels:
// 16 B per lane, grid-stride, <= 2048 blocks (cdna_hip_programming.md Guideline 11/13).
#include <stdarg.h>
#include <string.h>

#include <atomic>

#include "common.h"
#include "kernels.h"

namespace p2t {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---------------------------------------------------------------------------------------------
// Sticky fault word + epoch bookkeeping (the reference's two runtime guards, scripts/train_contrast.py:431-434, 476-480,
// kept on the device so the hot loop never pays a per-batch `.item()`).
//
// g_fault_word: one word of device-global storage per GPU, bit 0 = "a split-K consumer gave up waiting for its producer"
// (gemm_mfma.hip / gemm_w4.hip: bounded spin).  Every split-K launch points SplitFix::timeout at it; nothing clears it except
// p2t_fault_status(clear = 1), so a time-out stays visible to whoever looks next: consumers (they write NaN tiles while it is
// set, so the step's loss is NaN) and p2t_epoch_accumulate (copies it into the epoch's flag block).
__device__ unsigned g_fault_word = 0;

unsigned* fault_word_ptr() {
    static std::atomic<unsigned*> cache[32] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) return nullptr;
    unsigned* p = cache[dev].load(std::memory_order_acquire);
    if (!p) {
        void* q = nullptr;
        if (hipGetSymbolAddress(&q, HIP_SYMBOL(g_fault_word)) != hipSuccess) return nullptr;
        p = (unsigned*)q;
        cache[dev].store(p, std::memory_order_release);
    }
    return p;
}

// sums[0] += loss, sums[1] += 1 (train_contrast.py:443-444: `ddp_loss[0] += batch_loss_value; ddp_loss[1] += 1`);
// grad_norm given (an optimizer step ran): sums[2] += grad_norm, sums[3] += 1 (:461-462).
// flags[0] = number of "impossible" batch losses so far (NaN, inf or <= 0: the condition of :433), flags[1] = batch index of
// the first one (-1: none), flags[3] = its bit pattern; flags[2] |= the sticky fault word.
__global__ void epoch_accumulate_kernel(const float* __restrict__ loss, const float* __restrict__ grad_norm, int batch_idx,
                                        float* __restrict__ sums, int* __restrict__ flags, const unsigned* __restrict__ fault) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float l = loss[0];
    sums[0] += l;
    sums[1] += 1.0f;
    if (grad_norm) {
        sums[2] += grad_norm[0];
        sums[3] += 1.0f;
    }
    if (!(l > 0.0f) || isinf(l)) {            // NaN fails `l > 0`
        if (flags[0] == 0) {
            flags[1] = batch_idx;
            flags[3] = __float_as_int(l);
        }
        flags[0] += 1;
    }
    flags[2] |= (int)__hip_atomic_load(fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void fault_set_kernel(unsigned* fault, unsigned value) {
    if (threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store(fault, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) fill_hash_kernel(T* __restrict__ dst, int64_t n, uint64_t add, uint64_t xorv,
                                                        float scale23, float offset) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint64_t h = mix64(((uint64_t)(i + j) + add) ^ xorv);
            const int u = (int)(h >> 40) - 8388608;
            v[j] = __fadd_rn(__fmul_rn((float)u, scale23), offset);    // two roundings, never an FMA
        }
        if (i + 3 < n) {
            store4(dst + i, v);
        } else {
            for (int j = 0; j < 4 && i + j < n; ++j) dst[i + j] = from_f32<T>(v[j]);
        }
    }
}

template <typename S, typename D>
__global__ void __launch_bounds__(256) cast_kernel(const S* __restrict__ src, D* __restrict__ dst, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 3 < n) {
            float v[4];
            load4(src + i, v);
            store4(dst + i, v);
        } else {
            for (int j = 0; j < 4 && i + j < n; ++j) dst[i + j] = from_f32<D>(to_f32(src[i + j]));
        }
    }
}

// 64x64 tile transpose through LDS (padded: conflict-free column reads).
template <typename T>
__global__ void __launch_bounds__(256) transpose_kernel(const T* __restrict__ src, int64_t rows, int64_t cols,
                                                        int64_t ld_src, T* __restrict__ dst, int64_t ld_dst) {
    __shared__ T tile[64][65];
    const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int64_t r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < rows && c < cols) ? src[r * ld_src + c] : from_f32<T>(0.f);
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int64_t c = c0 + i, r = r0 + tx;
        if (c < cols && r < ld_dst) dst[c * ld_dst + r] = tile[tx][i];      // rows..ld_dst-1: zero K padding
    }
}


// Column sums in two deterministic stages (no atomics): stage 1, grid (cols/64, kColsumChunks): each block sums its
// row chunk of a 64-column slab (4 waves stride the rows) into partial[chunk][col]; stage 2 adds the chunks in order.
constexpr int kColsumChunks = 64;
template <typename T>
__global__ void __launch_bounds__(256) colsum_partial_kernel(const T* __restrict__ x, int64_t rows, int64_t cols, int64_t ld,
                                                             float* __restrict__ partial) {
    __shared__ float part[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t c = (int64_t)blockIdx.x * 64 + tx;
    const int64_t per = (rows + gridDim.y - 1) / gridDim.y;
    const int64_t r0 = (int64_t)blockIdx.y * per, r1 = r0 + per < rows ? r0 + per : rows;
    float acc = 0.f;
    if (c < cols)
        for (int64_t r = r0 + ty; r < r1; r += 4) acc += to_f32(x[r * ld + c]);
    part[ty][tx] = acc;
    __syncthreads();
    if (ty == 0 && c < cols) partial[(int64_t)blockIdx.y * cols + c] = part[0][tx] + part[1][tx] + part[2][tx] + part[3][tx];
}
__global__ void __launch_bounds__(256) colsum_final_kernel(const float* __restrict__ partial, int chunks, int64_t cols,
                                                           float* __restrict__ out, int accumulate) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    float s = 0.f;
    for (int k = 0; k < chunks; ++k) s += partial[(int64_t)k * cols + c];
    out[c] = accumulate ? out[c] + s : s;
}


// x[i] *= s[0]  (s lives on the device: chain-rule scaling without a host sync)
__global__ void __launch_bounds__(256) scale_kernel(float* __restrict__ x, int64_t n, const float* __restrict__ s) {
    const float f = s[0];
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) x[i] *= f;
}

static inline int stream_grid(int64_t n_items, int per_block) {
    int64_t g = ceil_div(n_items, per_block);
    return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

size_t colsum_scratch_bytes(int64_t cols) { return sizeof(float) * (size_t)kColsumChunks * (size_t)cols; }

int launch_colsum(const void* x, int dtype, int64_t rows, int64_t cols, int64_t ld, float* out, int accumulate,
                  float* scratch, hipStream_t s) {
    const int chunks = (int)(rows < kColsumChunks ? (rows < 1 ? 1 : rows) : kColsumChunks);
    const dim3 grid((unsigned)ceil_div(cols, 64), (unsigned)chunks);
    if (dtype == P2T_BF16)
        colsum_partial_kernel<bf16_t><<<grid, 256, 0, s>>>((const bf16_t*)x, rows, cols, ld, scratch);
    else
        colsum_partial_kernel<float><<<grid, 256, 0, s>>>((const float*)x, rows, cols, ld, scratch);
    P2T_LAUNCH_CHECK();
    colsum_final_kernel<<<dim3((unsigned)ceil_div(cols, 256)), 256, 0, s>>>(scratch, chunks, cols, out, accumulate);
    P2T_LAUNCH_CHECK();
    return P2T_OK;
}

}  // namespace p2t

using namespace p2t;

extern "C" int p2t_version(void) { return P2T_VERSION; }
extern "C" const char* p2t_last_error(void) { return g_err; }
extern "C" size_t p2t_struct_size(int which) {
    switch (which) {
        case 0: return sizeof(p2t_esm2_config);
        case 1: return sizeof(p2t_esm2_layer);
        case 2: return sizeof(p2t_esm2_weights);
        case 3: return sizeof(p2t_llama_config);
        case 4: return sizeof(p2t_llama_layer);
        case 5: return sizeof(p2t_llama_weights);
        case 6: return sizeof(p2t_adapter_config);
        case 7: return sizeof(p2t_adapter_weights);
        case 8: return sizeof(p2t_adapter_saved);
        case 9: return sizeof(p2t_llama_layer_t);
        case 10: return sizeof(p2t_kv_cache);
        case 11: return sizeof(p2t_llama_layer_stream);
    }
    return 0;
}

extern "C" int p2t_fill_hash(void* dst, int64_t n, uint64_t add, uint64_t xorv, float scale23, float offset, int dtype,
                             p2t_stream stream) {
    P2T_REQUIRE(dst && n >= 0, "p2t_fill_hash: bad arguments");
    if (n == 0) return P2T_OK;
    hipStream_t s = (hipStream_t)stream;
    const int grid = stream_grid(n, 1024);
    if (dtype == P2T_BF16)
        fill_hash_kernel<bf16_t><<<grid, 256, 0, s>>>((bf16_t*)dst, n, add, xorv, scale23, offset);
    else if (dtype == P2T_F32)
        fill_hash_kernel<float><<<grid, 256, 0, s>>>((float*)dst, n, add, xorv, scale23, offset);
    else
        P2T_REQUIRE(false, "p2t_fill_hash: dtype %d", dtype);
    P2T_LAUNCH_CHECK();
    return P2T_OK;
}

extern "C" int p2t_scale_by_device_scalar(float* x, int64_t n, const float* scalar, p2t_stream stream) {
    P2T_REQUIRE(x && scalar && n >= 0, "p2t_scale_by_device_scalar: bad arguments");
    if (n == 0) return P2T_OK;
    scale_kernel<<<stream_grid(n, 1024), 256, 0, (hipStream_t)stream>>>(x, n, scalar);
    P2T_LAUNCH_CHECK();
    return P2T_OK;
}

extern "C" int p2t_cast(const void* src, int sd, void* dst, int dd, int64_t n, p2t_stream stream) {
    P2T_REQUIRE(src && dst && n >= 0, "p2t_cast: bad arguments");
    if (n == 0) return P2T_OK;
    hipStream_t s = (hipStream_t)stream;
    const int grid = stream_grid(n, 1024);
    if (sd == P2T_F32 && dd == P2T_BF16)
        cast_kernel<float, bf16_t><<<grid, 256, 0, s>>>((const float*)src, (bf16_t*)dst, n);
    else if (sd == P2T_BF16 && dd == P2T_F32)
        cast_kernel<bf16_t, float><<<grid, 256, 0, s>>>((const bf16_t*)src, (float*)dst, n);
    else if (sd == P2T_F32 && dd == P2T_F32)
        cast_kernel<float, float><<<grid, 256, 0, s>>>((const float*)src, (float*)dst, n);
    else if (sd == P2T_BF16 && dd == P2T_BF16)
        cast_kernel<bf16_t, bf16_t><<<grid, 256, 0, s>>>((const bf16_t*)src, (bf16_t*)dst, n);
    else
        P2T_REQUIRE(false, "p2t_cast: dtypes %d -> %d", sd, dd);
    P2T_LAUNCH_CHECK();
    return P2T_OK;
}

extern "C" int p2t_transpose(const void* src, int64_t rows, int64_t cols, int64_t ld_src, void* dst, int64_t ld_dst,
                             int dtype, p2t_stream stream) {
    P2T_REQUIRE(src && dst && rows >= 0 && cols >= 0 && ld_src >= cols && ld_dst >= rows, "p2t_transpose: bad arguments");
    if (rows == 0 || cols == 0) return P2T_OK;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)ceil_div(cols, 64), (unsigned)ceil_div(rows, 64));
    P2T_REQUIRE(grid.y <= 65535, "p2t_transpose: too many rows (%lld)", (long long)rows);
    if (dtype == P2T_BF16)
        transpose_kernel<bf16_t><<<grid, 256, 0, s>>>((const bf16_t*)src, rows, cols, ld_src, (bf16_t*)dst, ld_dst);
    else
        transpose_kernel<float><<<grid, 256, 0, s>>>((const float*)src, rows, cols, ld_src, (float*)dst, ld_dst);
    P2T_LAUNCH_CHECK();
    return P2T_OK;
}

extern "C" int p2t_epoch_accumulate(const float* loss, const float* grad_norm, int batch_idx, float* sums, int32_t* flags,
                                    p2t_stream stream) {
    P2T_REQUIRE(loss && sums && flags, "p2t_epoch_accumulate: null argument");
    unsigned* fault = fault_word_ptr();
    P2T_REQUIRE(fault, "p2t_epoch_accumulate: no device fault word (hipGetSymbolAddress failed)");
    epoch_accumulate_kernel<<<1, 64, 0, (hipStream_t)stream>>>(loss, grad_norm, batch_idx, sums, flags, fault);
    P2T_LAUNCH_CHECK();
    return P2T_OK;
}

extern "C" int p2t_fault_status(unsigned* host_out, int clear) {
    P2T_REQUIRE(host_out, "p2t_fault_status: null argument");
    unsigned* fault = fault_word_ptr();
    P2T_REQUIRE(fault, "p2t_fault_status: no device fault word (hipGetSymbolAddress failed)");
    P2T_CHECK_HIP(hipMemcpy(host_out, fault, sizeof(unsigned), hipMemcpyDeviceToHost));     // synchronous by design
    if (clear && *host_out) {
        const unsigned zero = 0;
        P2T_CHECK_HIP(hipMemcpy(fault, &zero, sizeof(unsigned), hipMemcpyHostToDevice));
    }
    return P2T_OK;
}

extern "C" int p2t_fault_inject(unsigned value, p2t_stream stream) {
    unsigned* fault = fault_word_ptr();
    P2T_REQUIRE(fault, "p2t_fault_inject: no device fault word (hipGetSymbolAddress failed)");
    fault_set_kernel<<<1, 64, 0, (hipStream_t)stream>>>(fault, value);
    P2T_LAUNCH_CHECK();
    return P2T_OK;
}
