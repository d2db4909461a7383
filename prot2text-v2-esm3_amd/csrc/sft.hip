// Decoder half of Esm2LlamaInstructForCausalLM.forward (reference
// models/modeling_esm2llama_instruct.py:108-139, 195-215 and HF LlamaForCausalLM's shifted cross-entropy):
//   * positions_where      -- flat positions of the selected elements of an int64 array, in order (the row-major order
//                             torch's boolean-mask indexing uses), plus their count;
//   * scatter_rows         -- inputs_embeds[placeholder_mask] = encoder_hidden_states[encoder_mask];
//   * cross_entropy_shift  -- mean over (b, t < T-1, labels[b, t+1] != ignore) of logsumexp(logits[b, t]) - logits[b, t, label].
// All HBM-bound streaming kernels in fp32 arithmetic; reductions are deterministic (no float atomics).
#include "common.h"
#include "kernels.h"

namespace p2t {

// One block of 1024 threads scans the whole array: thread i owns a contiguous segment, counts its hits, the counts are
// exclusive-scanned through LDS, then each thread walks its segment again and writes the positions.  n is B*T (<= a few
// hundred thousand), so one block is microseconds; the order of `pos` is the order of the array.
__global__ void __launch_bounds__(1024) positions_where_kernel(const int64_t* __restrict__ v, int64_t n, int mode, int64_t match,
                                                               int32_t* __restrict__ pos, int32_t* __restrict__ count) {
    __shared__ int part[1024];
    const int tid = threadIdx.x;
    const int64_t seg = (n + 1023) / 1024, lo = tid * seg, hi = lo + seg < n ? lo + seg : n;
    int c = 0;
    for (int64_t i = lo; i < hi; ++i) c += mode == 0 ? (v[i] == match) : (v[i] != 0);
    part[tid] = c;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {          // Hillis-Steele inclusive scan
        const int add = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += add;
        __syncthreads();
    }
    int r = part[tid] - c;
    for (int64_t i = lo; i < hi; ++i)
        if (mode == 0 ? (v[i] == match) : (v[i] != 0)) pos[r++] = (int32_t)i;
    if (tid == 1023) *count = part[1023];
}

template <typename Tsrc>
__global__ void __launch_bounds__(256) scatter_rows_kernel(float* __restrict__ dst, int64_t ld_dst, const int32_t* __restrict__ dst_pos,
                                                           const Tsrc* __restrict__ src, int64_t ld_src,
                                                           const int32_t* __restrict__ src_pos, const int32_t* __restrict__ n_dst,
                                                           const int32_t* __restrict__ n_src, int H) {
    const int rows = min(*n_dst, *n_src);
    for (int r = blockIdx.x; r < rows; r += gridDim.x) {
        float* d = dst + (int64_t)dst_pos[r] * ld_dst;
        const Tsrc* s = src + (int64_t)src_pos[r] * ld_src;
        for (int c = threadIdx.x; c < H; c += 256) d[c] = to_f32(s[c]);
    }
}

template <typename T>
__global__ void __launch_bounds__(256) ce_rows_kernel(const T* __restrict__ logits, int64_t ld, const int64_t* __restrict__ labels,
                                                      int T_len, int V, int64_t ignore_index, float* __restrict__ row_loss,
                                                      int32_t* __restrict__ row_valid) {
    __shared__ float red[4];
    const int64_t row = blockIdx.x;                     // (b, t)
    const int t = (int)(row % T_len);
    const int64_t label = t + 1 < T_len ? labels[row + 1] : ignore_index;
    if (label == ignore_index || label < 0 || label >= V) {
        if (threadIdx.x == 0) { row_loss[row] = 0.f; row_valid[row] = 0; }
        return;
    }
    const T* x = logits + row * ld;
    float m = -INFINITY;
    for (int c = threadIdx.x; c < V; c += 256) m = fmaxf(m, to_f32(x[c]));
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float s = 0.f;
    for (int c = threadIdx.x; c < V; c += 256) s += expf(to_f32(x[c]) - m);
    s = block_sum<4>(s, red);
    if (threadIdx.x == 0) {
        row_loss[row] = logf(s) + m - to_f32(x[label]);
        row_valid[row] = 1;
    }
}

__global__ void __launch_bounds__(1024) ce_reduce_kernel(const float* __restrict__ row_loss, const int32_t* __restrict__ row_valid,
                                                         int64_t M, float* __restrict__ loss, int32_t* __restrict__ count) {
    __shared__ float red[16];
    __shared__ float redc[16];
    float s = 0.f, c = 0.f;
    for (int64_t i = threadIdx.x; i < M; i += 1024) { s += row_loss[i]; c += (float)row_valid[i]; }
    s = block_sum<16>(s, red);
    c = block_sum<16>(c, redc);
    if (threadIdx.x == 0) {
        *loss = s / c;                                  // no valid target: 0 / 0 = NaN, as torch's mean over nothing
        if (count) *count = (int32_t)c;
    }
}

}  // namespace p2t

using namespace p2t;

extern "C" int p2t_positions_where(const int64_t* values, int64_t n, int mode, int64_t match, int32_t* pos, int32_t* count,
                                   p2t_stream stream) {
    P2T_REQUIRE(values && pos && count && n > 0 && n < (1ll << 31) && (mode == 0 || mode == 1), "p2t_positions_where: bad arguments");
    positions_where_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(values, n, mode, match, pos, count);
    P2T_LAUNCH_CHECK();
    return P2T_OK;
}

extern "C" int p2t_scatter_rows(float* dst, int64_t ld_dst, const int32_t* dst_pos, const void* src, int64_t ld_src, int src_dtype,
                                const int32_t* src_pos, const int32_t* n_dst, const int32_t* n_src, int64_t max_rows, int H,
                                p2t_stream stream) {
    P2T_REQUIRE(dst && dst_pos && src && src_pos && n_dst && n_src && max_rows > 0 && H > 0 && ld_dst >= H && ld_src >= H,
                "p2t_scatter_rows: bad arguments");
    P2T_REQUIRE(src_dtype == P2T_F32 || src_dtype == P2T_BF16, "p2t_scatter_rows: unsupported dtype %d", src_dtype);
    const unsigned grid = (unsigned)(max_rows < 4096 ? max_rows : 4096);
    if (src_dtype == P2T_BF16)
        scatter_rows_kernel<bf16_t><<<grid, 256, 0, (hipStream_t)stream>>>(dst, ld_dst, dst_pos, (const bf16_t*)src, ld_src, src_pos,
                                                                          n_dst, n_src, H);
    else
        scatter_rows_kernel<float><<<grid, 256, 0, (hipStream_t)stream>>>(dst, ld_dst, dst_pos, (const float*)src, ld_src, src_pos,
                                                                         n_dst, n_src, H);
    P2T_LAUNCH_CHECK();
    return P2T_OK;
}

extern "C" int p2t_cross_entropy_shifted(const void* logits, int64_t ld, int dtype, const int64_t* labels, int B, int T, int V,
                                         int64_t ignore_index, float* row_loss, int32_t* row_valid, float* loss, int32_t* count,
                                         p2t_stream stream) {
    P2T_REQUIRE(logits && labels && row_loss && row_valid && loss && B > 0 && T > 0 && V > 0 && ld >= V,
                "p2t_cross_entropy_shifted: bad arguments");
    P2T_REQUIRE(dtype == P2T_F32 || dtype == P2T_BF16, "p2t_cross_entropy_shifted: unsupported dtype %d", dtype);
    const int64_t M = (int64_t)B * T;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == P2T_BF16)
        ce_rows_kernel<bf16_t><<<(unsigned)M, 256, 0, s>>>((const bf16_t*)logits, ld, labels, T, V, ignore_index, row_loss, row_valid);
    else
        ce_rows_kernel<float><<<(unsigned)M, 256, 0, s>>>((const float*)logits, ld, labels, T, V, ignore_index, row_loss, row_valid);
    P2T_LAUNCH_CHECK();
    ce_reduce_kernel<<<1, 1024, 0, s>>>(row_loss, row_valid, M, loss, count);
    P2T_LAUNCH_CHECK();
    return P2T_OK;
}
