// Readout (masked mean / population std / mix / last) and the row-wise InfoNCE loss, forward and
// backward.  Restates scripts/train_contrast.py:198-248 (readout_embeddings) and :72-114
// (BatchInfoNCELoss / SegmentedBatchInfoNCELoss).  fp32 arithmetic whatever the tower dtype.
// HBM-bound streaming passes; reductions by wave shuffles + a 4-wave LDS combine.
#include "common.h"
#include "kernels.h"

namespace p2t {

// grid (ceil(D/64), B), block 256: a wave reads 4 rows x 64 columns per instruction (lane = 16 column quads x 4 rows,
// 8 or 16 bytes per lane), the four waves stripe the rows.  Two passes over the (b, 64-column) slab exactly as the
// reference does (mean first, then the masked squared deviations); requires D % 4 == 0.
template <typename T>
__global__ void __launch_bounds__(256) readout_kernel(const T* __restrict__ emb, int64_t ld, const int64_t* __restrict__ mask,
                                                      int seq, int D, int mode, float* __restrict__ out) {
    __shared__ float part[4][64];
    const int b = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int cq = lane & 15, rq = lane >> 4;               // column quad, row inside the wave's 4-row group
    const int c = blockIdx.x * 64 + cq * 4;
    const T* base = emb + (int64_t)b * seq * ld;
    const int64_t* mrow = mask ? mask + (int64_t)b * seq : nullptr;
    // count of the mask (every wave computes it redundantly from its lanes)
    float cnt = 0.f;
    for (int t = lane; t < seq; t += 64) cnt += mrow ? (float)mrow[t] : 1.f;
    cnt = wave_sum(cnt);
    const int out_ld = mode == P2T_READOUT_MIX ? 2 * D : D;
    if (mode == P2T_READOUT_LAST) {
        const int idx = (int)cnt - 1;
        if (w == 0 && rq == 0 && c < D && idx >= 0) {
            float v[4];
            load4(base + (int64_t)idx * ld + c, v);
            store4(out + (int64_t)b * out_ld + c, v);
        }
        return;
    }
    // column sums over this wave's rows, then over the 4 row groups (lanes 16 apart) and the 4 waves
    auto reduce = [&](float (&v)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j] += __shfl_xor(v[j], 16, 64);
            v[j] += __shfl_xor(v[j], 32, 64);
        }
        __syncthreads();
        if (rq == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) part[w][cq * 4 + j] = v[j];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = part[0][cq * 4 + j] + part[1][cq * 4 + j] + part[2][cq * 4 + j] + part[3][cq * 4 + j];
    };
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < D)
        for (int t = w * 4 + rq; t < seq; t += 16) {
            const float mk = mrow ? (float)mrow[t] : 1.f;
            float v[4];
            load4(base + (int64_t)t * ld + c, v);
#pragma unroll
            for (int j = 0; j < 4; ++j) s[j] += v[j] * mk;
        }
    reduce(s);
    float mean[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) mean[j] = s[j] / cnt;
    if (mode == P2T_READOUT_MEAN) {
        if (w == 0 && rq == 0 && c < D) store4(out + (int64_t)b * out_ld + c, mean);
        return;
    }
    float qd[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < D)
        for (int t = w * 4 + rq; t < seq; t += 16) {
            const float mk = mrow ? (float)mrow[t] : 1.f;
            float v[4];
            load4(base + (int64_t)t * ld + c, v);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float dlt = v[j] - mean[j];
                qd[j] += dlt * dlt * mk;
            }
        }
    reduce(qd);
    if (w == 0 && rq == 0 && c < D) {
        float sd[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) sd[j] = sqrtf(qd[j] / cnt);
        if (mode == P2T_READOUT_STD) {
            store4(out + (int64_t)b * out_ld + c, sd);
        } else {
            store4(out + (int64_t)b * out_ld + c, mean);
            store4(out + (int64_t)b * out_ld + D + c, sd);
        }
    }
}

// d_emb[b,t,c] = m_t * ( dmean_c / cnt + dstd_c * (x - mean_c) / (cnt * std_c) ); "last": scatter.
// A wave owns 4 x 64 columns and walks kRowsPerBlock / 4 rows: the per-column coefficients (two divisions each) are
// formed once per wave instead of once per element, so the pass is the HBM stream it should be (2 B read + 4 B written
// per element) rather than a division loop.
constexpr int kReadoutBwdRows = 32;
template <typename T>
__global__ void __launch_bounds__(256) readout_bwd_kernel(const T* __restrict__ emb, int64_t ld, const int64_t* __restrict__ mask,
                                                          int seq, int D, int mode, const float* __restrict__ pooled,
                                                          const float* __restrict__ d_out, float* __restrict__ d_emb) {
    const int b = blockIdx.z, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t* mrow = mask ? mask + (int64_t)b * seq : nullptr;
    float cnt = 0.f;
    for (int t = lane; t < seq; t += 64) cnt += mrow ? (float)mrow[t] : 1.f;
    cnt = wave_sum(cnt);
    const int c = blockIdx.x * 256 + lane * 4;
    if (c >= D) return;
    const int out_ld = mode == P2T_READOUT_MIX ? 2 * D : D;
    const float* po = pooled ? pooled + (int64_t)b * 2 * D : nullptr;
    const float* go = d_out + (int64_t)b * out_ld;
    // r = (ca + cb * (x - cm)) * m_t
    float ca[4] = {0.f, 0.f, 0.f, 0.f}, cb[4] = {0.f, 0.f, 0.f, 0.f}, cm[4] = {0.f, 0.f, 0.f, 0.f}, glast[4] = {0.f, 0.f, 0.f, 0.f};
    if (mode == P2T_READOUT_LAST) {
        load4(go + c, glast);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (mode == P2T_READOUT_MEAN || mode == P2T_READOUT_MIX) ca[j] = go[c + j] / cnt;
            if (mode == P2T_READOUT_STD || mode == P2T_READOUT_MIX) {
                // `pooled` is the forward [mean | std] (2D wide) for both modes
                const float sd = po[D + c + j];
                const float gs = mode == P2T_READOUT_MIX ? go[D + c + j] : go[c + j];
                cm[j] = po[c + j];
                cb[j] = gs / (cnt * sd);
            }
        }
    }
    const int t0 = blockIdx.y * kReadoutBwdRows;
    for (int t = t0 + w; t < t0 + kReadoutBwdRows && t < seq; t += 4) {
        const float mk = mrow ? (float)mrow[t] : 1.f;
        float* g = d_emb + ((int64_t)b * seq + t) * D;
        float r[4];
        if (mode == P2T_READOUT_LAST) {
#pragma unroll
            for (int j = 0; j < 4; ++j) r[j] = t == (int)cnt - 1 ? glast[j] : 0.f;
        } else {
            float xv[4];
            load4(emb + ((int64_t)b * seq + t) * ld + c, xv);
#pragma unroll
            for (int j = 0; j < 4; ++j) r[j] = (ca[j] + cb[j] * (xv[j] - cm[j])) * mk;
        }
        store4(g + c, r);
    }
}

// ---------------------------------------------------------------------------------------------
// InfoNCE.  One block per row i of `seg`: logits[i, j] = <seg_i, batch_j> / tau for all j (a wave per j,
// lanes striding D with 16-byte loads), then a max-shifted log-sum-exp (the reference exponentiates
// without the shift; |l| <= 1/tau = 20, identical up to fp32 rounding).
__global__ void __launch_bounds__(256) infonce_fwd_kernel(const float* __restrict__ seg, const float* __restrict__ batch,
                                                          const int32_t* __restrict__ labels, int N, int D, float inv_tau,
                                                          float* __restrict__ logits, float* __restrict__ row_loss,
                                                          float* __restrict__ lse_out) {
    // labels == nullptr: the positive of row i is column i (column term: rows are texts, columns proteins)
    __shared__ float red[4];
    const int i = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float* p = seg + (int64_t)i * D;
    float* lrow = logits + (int64_t)i * N;
    for (int j = w; j < N; j += 4) {
        const float* tr = batch + (int64_t)j * D;
        float dot = 0.f;
        for (int c = lane * 4; c < D; c += 256) {
            float a[4], bb[4];
            load4(p + c, a);
            load4(tr + c, bb);
            dot = fmaf(a[0], bb[0], dot); dot = fmaf(a[1], bb[1], dot);
            dot = fmaf(a[2], bb[2], dot); dot = fmaf(a[3], bb[3], dot);
        }
        dot = wave_sum(dot);
        if (lane == 0) lrow[j] = dot * inv_tau;
    }
    __threadfence_block();
    __syncthreads();
    float mx = -INFINITY;
    for (int j = threadIdx.x; j < N; j += 256) mx = fmaxf(mx, lrow[j]);
    mx = wave_max(mx);
    if (lane == 0) red[w] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float se = 0.f;
    for (int j = threadIdx.x; j < N; j += 256) se += expf(lrow[j] - mx);
    se = block_sum<4>(se, red);
    if (threadIdx.x == 0) {
        const float lse = mx + logf(se);
        row_loss[i] = lse - lrow[labels ? labels[i] : i];
        if (lse_out) lse_out[i] = lse;
    }
}

__global__ void __launch_bounds__(256) infonce_reduce_kernel(const float* __restrict__ row_loss, const int32_t* __restrict__ which,
                                                             int first, int S, float weight, int accumulate,
                                                             float* __restrict__ loss_out) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < S; i += 256) s += row_loss[which ? which[i] : first + i];
    s = block_sum<4>(s, red);
    if (threadIdx.x == 0) {
        const float v = weight * s / (float)S;
        loss_out[0] = accumulate ? loss_out[0] + v : v;
    }
}

// Row term (col_lse == nullptr):  d_seg[i, c] = weight / (S tau) * sum_j (softmax_ij - [j == y_i]) batch[j, c].
// Column term (col_lse = log-sum-exp of every COLUMN j over all rows of the global batch):
//   d_seg[i, c] (+)= weight / tau * sum_j (exp(l_ij - col_lse_j) - [j == y_i]) batch[j, c]
// -- the derivative of sum_j (col_lse_j - l_jj) with respect to row i, which reaches row i through every column.
__global__ void __launch_bounds__(256) infonce_bwd_kernel(const float* __restrict__ batch, const int32_t* __restrict__ labels,
                                                          const float* __restrict__ logits, const float* __restrict__ col_lse,
                                                          int S, int N, int D, float inv_tau, float weight, int accumulate,
                                                          float* __restrict__ d_seg) {
    extern __shared__ float coef[];            // N floats
    __shared__ float red[4];
    const int i = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float* lrow = logits + (int64_t)i * N;
    float mx = -INFINITY;
    for (int j = threadIdx.x; j < N; j += 256) mx = fmaxf(mx, lrow[j]);
    mx = wave_max(mx);
    if (lane == 0) red[w] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float se = 0.f;
    for (int j = threadIdx.x; j < N; j += 256) se += expf(lrow[j] - mx);
    se = block_sum<4>(se, red);
    const float sc = col_lse ? weight * inv_tau : weight * inv_tau / (float)S;
    const int y = labels[i];
    for (int j = threadIdx.x; j < N; j += 256) {
        const float pj = col_lse ? expf(lrow[j] - col_lse[j]) : expf(lrow[j] - mx) / se;
        coef[j] = (pj - (j == y ? 1.f : 0.f)) * sc;
    }
    __syncthreads();
    for (int c = threadIdx.x * 4; c < D; c += 1024) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < N; ++j) {
            float bb[4];
            load4(batch + (int64_t)j * D + c, bb);
            const float cj = coef[j];
            acc[0] = fmaf(cj, bb[0], acc[0]); acc[1] = fmaf(cj, bb[1], acc[1]);
            acc[2] = fmaf(cj, bb[2], acc[2]); acc[3] = fmaf(cj, bb[3], acc[3]);
        }
        if (accumulate) {
            float old[4];
            load4(d_seg + (int64_t)i * D + c, old);
            acc[0] += old[0]; acc[1] += old[1]; acc[2] += old[2]; acc[3] += old[3];
        }
        store4(d_seg + (int64_t)i * D + c, acc);
    }
}

}  // namespace p2t

using namespace p2t;

extern "C" int p2t_readout(const void* emb, int dtype, int64_t ld, const int64_t* mask, int B, int T, int D, int mode,
                           float* out, p2t_stream stream) {
    P2T_REQUIRE(emb && out && B > 0 && T > 0 && D > 0 && ld >= D && mode >= 0 && mode <= 3, "p2t_readout: bad arguments");
    P2T_REQUIRE(D % 4 == 0 && ld % 4 == 0, "p2t_readout: D and ld must be multiples of 4 (D=%d ld=%lld)", D, (long long)ld);
    const dim3 grid((unsigned)ceil_div(D, 64), (unsigned)B);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == P2T_BF16)
        readout_kernel<bf16_t><<<grid, 256, 0, s>>>((const bf16_t*)emb, ld, mask, T, D, mode, out);
    else
        readout_kernel<float><<<grid, 256, 0, s>>>((const float*)emb, ld, mask, T, D, mode, out);
    P2T_LAUNCH_CHECK();
    return P2T_OK;
}

extern "C" int p2t_readout_backward(const void* emb, int dtype, int64_t ld, const int64_t* mask, int B, int T, int D,
                                    int mode, const float* pooled, const float* d_out, float* d_emb, p2t_stream stream) {
    P2T_REQUIRE(emb && d_out && d_emb && B > 0 && T > 0 && D > 0 && D % 4 == 0 && ld % 4 == 0 && mode >= 0 && mode <= 3,
                "p2t_readout_backward: bad arguments (D and ld must be multiples of 4)");
    P2T_REQUIRE(pooled || mode == P2T_READOUT_LAST || mode == P2T_READOUT_MEAN,
                "p2t_readout_backward: std/mix need the forward [mean | std] (2D wide) in `pooled`");
    const dim3 grid((unsigned)ceil_div(D, 256), (unsigned)ceil_div(T, kReadoutBwdRows), (unsigned)B);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == P2T_BF16)
        readout_bwd_kernel<bf16_t><<<grid, 256, 0, s>>>((const bf16_t*)emb, ld, mask, T, D, mode, pooled, d_out, d_emb);
    else
        readout_bwd_kernel<float><<<grid, 256, 0, s>>>((const float*)emb, ld, mask, T, D, mode, pooled, d_out, d_emb);
    P2T_LAUNCH_CHECK();
    return P2T_OK;
}

extern "C" int p2t_infonce_forward(const float* seg, const float* batch, const int32_t* labels, int S, int N, int D,
                                   float temperature, float weight, int accumulate, float* loss_out, float* logits,
                                   float* row_loss, p2t_stream stream) {
    P2T_REQUIRE(seg && batch && labels && loss_out && logits && row_loss && S > 0 && N > 0 && D > 0 && D % 4 == 0 && temperature > 0.f,
                "p2t_infonce_forward: bad arguments (logits [S,N] and row_loss [S] scratch are required, D %% 4 == 0)");
    hipStream_t s = (hipStream_t)stream;
    infonce_fwd_kernel<<<S, 256, 0, s>>>(seg, batch, labels, N, D, 1.0f / temperature, logits, row_loss, nullptr);
    P2T_LAUNCH_CHECK();
    infonce_reduce_kernel<<<1, 256, 0, s>>>(row_loss, nullptr, 0, S, weight, accumulate, loss_out);
    P2T_LAUNCH_CHECK();
    return P2T_OK;
}

extern "C" int p2t_infonce_backward(const float* batch, const int32_t* labels, const float* logits, int S, int N, int D,
                                    float temperature, float weight, float* d_seg, p2t_stream stream) {
    P2T_REQUIRE(batch && labels && logits && d_seg && S > 0 && N > 0 && N <= 8192 && D > 0 && D % 4 == 0 && temperature > 0.f,
                "p2t_infonce_backward: bad arguments (N <= 8192, D %% 4 == 0)");
    infonce_bwd_kernel<<<S, 256, (size_t)N * sizeof(float), (hipStream_t)stream>>>(batch, labels, logits, nullptr, S, N, D,
                                                                                   1.0f / temperature, weight, 0, d_seg);
    P2T_LAUNCH_CHECK();
    return P2T_OK;
}

extern "C" int p2t_infonce_col_forward(const float* p_all, const float* t_all, int N, int D, float temperature, const int32_t* cols,
                                       int first, int count, float weight, int accumulate, float* loss_out, float* col_lse,
                                       float* scratch_logits, float* scratch_col_loss, p2t_stream stream) {
    P2T_REQUIRE(p_all && t_all && loss_out && col_lse && scratch_logits && scratch_col_loss && N > 0 && D > 0 && D % 4 == 0 &&
                    temperature > 0.f && count > 0 && count <= N && (cols || (first >= 0 && first + count <= N)),
                "p2t_infonce_col_forward: bad arguments (scratch_logits [N,N], scratch_col_loss [N], D %% 4 == 0)");
    hipStream_t s = (hipStream_t)stream;
    // rows = texts, columns = proteins: the same kernel as the row term with the operands swapped (logits^T)
    infonce_fwd_kernel<<<N, 256, 0, s>>>(t_all, p_all, nullptr, N, D, 1.0f / temperature, scratch_logits, scratch_col_loss, col_lse);
    P2T_LAUNCH_CHECK();
    infonce_reduce_kernel<<<1, 256, 0, s>>>(scratch_col_loss, cols, first, count, weight, accumulate, loss_out);
    P2T_LAUNCH_CHECK();
    return P2T_OK;
}

extern "C" int p2t_infonce_col_backward(const float* t_all, const int32_t* labels, const float* logits, const float* col_lse, int S,
                                        int N, int D, float temperature, float scale, int accumulate, float* d_seg, p2t_stream stream) {
    P2T_REQUIRE(t_all && labels && logits && col_lse && d_seg && S > 0 && N > 0 && N <= 8192 && D > 0 && D % 4 == 0 && temperature > 0.f,
                "p2t_infonce_col_backward: bad arguments (N <= 8192, D %% 4 == 0)");
    infonce_bwd_kernel<<<S, 256, (size_t)N * sizeof(float), (hipStream_t)stream>>>(t_all, labels, logits, col_lse, S, N, D,
                                                                                   1.0f / temperature, scale, accumulate, d_seg);
    P2T_LAUNCH_CHECK();
    return P2T_OK;
}
