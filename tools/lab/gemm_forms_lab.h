// LAB BUILD ONLY (-DP2T_LAB, `make lab` in csrc/ -> tools/build/libp2t_lab.so): the launch-form zoo of rounds 1-2, kept for
// A/B measurements (tools/microbench.py, tools/ab.sh) and for the fuzz tests that force one kernel form on shapes the default
// policy would give to another (tests/test_gpu_lab_forms.py runs them against this build through P2T_HIP_LIB).  The product
// library (libp2t_hip.so) contains none of this: its p2t_set_gemm_policy accepts 0 and 9 only.
//
// Included at the end of csrc/gemm_mfma.hip, inside namespace p2t.  Forced forms (policy):
//   128 / 256 = tile height of the per-tile kernels; 1 = no split-K tail; 2 = per-tile kernels only; 3 = eight-wave persistent
//   kernel with the split-K fix-up whenever possible; 4 = persistent, never the fix-up; 5 = persistent, partial round as 128-row
//   halves; 6 = 64-deep single-barrier skeleton (gemm_fp8.hip's, on bf16 operands); 7 = four-wave kernel, one tile per block;
//   8 = four-wave persistent kernel with split-K pairs whenever possible; 10 = four-wave persistent, a partial last round as
//   whole tiles; 12 = 10 with the other generated instruction order of the K loop (tools/gen_w4_schedule.py, SCHED 0).
#pragma once

template <typename Epi>
static int launch_shape_lab(const void* A, int64_t lda, const void* W, int64_t ldw, int64_t M, int N, int K, int n_cover,
                        const EpiParams& ep, int tile, void* fix_ws, size_t fix_bytes, unsigned fix_epoch, hipStream_t s) {
    const bool no_w4 = tile == 9;                           // 9: the default policy without the four-wave form (A/B runs)
    if (no_w4) tile = 0;
    const int policy_in = tile;                             // (the persistent block below folds some values into 0)
    if (tile == 6) {                                        // experiment: 64-deep single-barrier skeleton of gemm_fp8.hip
        extern int launch_gemm_bf16_k64(const void*, int64_t, const void*, int64_t, int64_t, int, int, int, int, int, const EpiParams&, hipStream_t);
        const int rc = launch_gemm_bf16_k64(A, lda, W, ldw, M, N, K, n_cover, std::is_same<Epi, EpiResid>::value || std::is_same<Epi, EpiStore<float>>::value || std::is_same<Epi, EpiGelu<float>>::value ? P2T_F32 : P2T_BF16,
                                            std::is_same<Epi, EpiResid>::value ? P2T_EPI_RESID : (std::is_same<Epi, EpiGelu<bf16_t>>::value || std::is_same<Epi, EpiGelu<float>>::value ? P2T_EPI_GELU : (std::is_same<Epi, EpiStore<bf16_t>>::value || std::is_same<Epi, EpiStore<float>>::value ? P2T_EPI_STORE : -1)), ep, s);
        if (rc != P2T_ERR_UNSUPPORTED) return rc;
        tile = 0;
    }
    if (tile == 7) {                                        // four-wave form of the 256 x 256 tile (gemm_w4.hip), per-tile launch
        extern int launch_gemm_w4(const void*, int64_t, const void*, int64_t, int64_t, int, int, int, int, int, const EpiParams&, hipStream_t);
        constexpr bool f32o = std::is_same<Epi, EpiResid>::value || std::is_same<Epi, EpiStore<float>>::value;
        constexpr int code = std::is_same<Epi, EpiResid>::value ? P2T_EPI_RESID
                             : std::is_same<Epi, EpiGelu<bf16_t>>::value ? P2T_EPI_GELU
                             : std::is_same<Epi, EpiQkvRope<bf16_t>>::value ? P2T_EPI_QKV_ROPE
                             : (std::is_same<Epi, EpiStore<bf16_t>>::value || std::is_same<Epi, EpiStore<float>>::value) ? P2T_EPI_STORE : -1;
        const int rc = launch_gemm_w4(A, lda, W, ldw, M, N, K, n_cover, f32o ? P2T_F32 : P2T_BF16, code, ep, s);
        if (rc != P2T_ERR_UNSUPPORTED) return rc;
        tile = 0;
    }
    // 128 | 256: tile height; 1: no split-K tail; 2: no persistent kernel; 3 / 4 / 5 below
    const int kCUs = cu_count();
    {
        // persistent kernel: shapes without edge tiles and at least one whole round of tiles.  tile: 0 / 3 = with the
        // split-K fix-up of a partial last round when it is worth it (3: whenever possible), 4 = never, 5 = the partial
        // round always as 128-row halves
        const int64_t items = ceil_div(M, 256) * ceil_div(n_cover, 256);
        const int ns = K >> 5;
        // measured (profiles/r01_microbench_v4.log): whole rounds -> persistent (+5..13 %); a partial last round with
        // K >= 4096 -> persistent + split-K fix-up (+2..4 % over the per-tile kernel with the fix-up); K < 4096 -> plain
        // persistent if there are at least four rounds (QKV: +13 %) or the epilogue is a read-modify-write of the residual
        // stream (o-proj, 2.5 rounds: with the stream cold in HBM, as it is inside a step, the undrained stores win
        // 0.4 % of the step; with it hot in the Infinity Cache, as in the micro-benchmark, the per-tile kernel is 6 % ahead)
        const int64_t rem = items % kCUs;
        const bool eligible = items >= kCUs && (ns & 3) == 0 && ns >= 12 && M % 256 == 0 && N % 256 == 0 && n_cover == N;
        const bool pick = tile == 3 || tile == 4 || tile == 5 || tile == 8 || tile == 10 || tile == 12 || (tile == 0 && (rem == 0 || ns >= 128 || items >= 4 * kCUs || Epi::kRmw));
        if (eligible && pick) {
            int64_t n_full = items, n_tail = 0;
            int half_tail = 0;
            SplitFix fix{};
            // the fix-up tiles run un-overlapped after the tile loop (~50 us): it pays when half a tile time is well above that
            // (K = 10240: +7 %), not at K = 4096 (SwiGLU GEMM of the text tower, 3.5 rounds: -7 %)
            const bool worth = tile == 3 || ns >= 192;
            if (tile != 4 && tile != 5 && fix_ws && rem > 0 && rem <= 128 && 2 * rem <= kCUs && (ns & 7) == 0 && worth && fix_bytes >= kFixHeader + (size_t)rem * kFixSlab) {
                n_full = items - rem;
                n_tail = rem;
                fix.flag = (unsigned*)fix_ws;
                fix.timeout = fault_word_ptr();
                fix.slab = (float*)((char*)fix_ws + kFixHeader);
                fix.epoch = fix_epoch;
            } else if ((tile == 0 || tile == 5 || tile == 8) && rem > 0 && rem <= 128 && 2 * rem <= kCUs) {
                // K too short for split-K to pay: the leftover tiles as 128-row halves, one per block (measured cold, as in
                // a step: QKV 7.5 rounds -4.6 %, o-proj 2.5 rounds -3.7 %; FFN-down K = 10240 stays split-K: 787 vs 818 us)
                n_full = items - rem;
                n_tail = rem;
                half_tail = 1;
            }
            if constexpr (kHasW4<Epi>) {
                // four-wave form (gemm_w4.hip): the tiles of a partial last round run as split-K pairs inside the same
                // persistent stream (tile == 10: as whole tiles)
                const int sched = tile == 12 ? 0 : 1;             // 12: the other instruction order of the four-wave K loop (tools/gen_w4_schedule.py)
                if (tile == 12) tile = 10;
                const bool w4 = !no_w4 && (tile == 0 || tile == 8 || tile == 10) && (int64_t)256 * (lda > ldw ? lda : ldw) * 2 < ((int64_t)1 << 32) && ns >= 8;
                if (w4) {
                    SplitFix f4{};
                    int64_t t4 = 0;
                    // measured cold (profiles/r02_microbench_w4.log): the pair form pays for long K (FFN-down K = 10240: 752 vs 821 us);
                    // at K = 2560 an extra round of whole tiles is cheaper than the second ring fill + the slab (QKV 496 vs 521 us)
                    if (tile != 10 && (tile == 8 || ns >= 192) && fix_ws && rem > 0 && 2 * rem <= kCUs && ns >= 16 && fix_bytes >= kFixHeader + (size_t)rem * kFixSlab) {
                        t4 = rem;
                        f4.flag = (unsigned*)fix_ws;
                        f4.timeout = fault_word_ptr();
                        f4.slab = (float*)((char*)fix_ws + kFixHeader);
                        f4.epoch = fix_epoch;
                    }
                    return launch_gemm_w4_persist<Epi>(A, lda, W, ldw, M, N, K, (int)(items - t4), (int)t4, kCUs, ep, f4, s, sched);
                }
            }
            gemm_nt_mfma_persist_kernel<Epi><<<dim3(kCUs), 512, 0, s>>>((const bf16_t*)A, lda, (const bf16_t*)W, ldw, M, N, K,
                                                                      (int)ceil_div(M, 256), (int)ceil_div(n_cover, 256),
                                                                      (int)n_full, (int)n_tail, half_tail, n_cover, ep, fix);
            P2T_LAUNCH_CHECK();
            return P2T_OK;
        }
        if (tile == 2 || tile == 3 || tile == 4 || tile == 5 || tile == 8 || tile == 10 || tile == 12) tile = 0;
    }
    const int64_t tn = ceil_div(n_cover, 256), tm256 = ceil_div(M, 256), tm128 = ceil_div(M, 128);
    const double cost256 = (double)ceil_div(tm256 * tn, kCUs);
    const double cost128 = (double)ceil_div(tm128 * tn, kCUs) * kSmallTileCost * 1.08;
    if constexpr (std::is_same<Epi, EpiQkvRope<bf16_t>>::value || std::is_same<Epi, EpiStore<bf16_t>>::value) {
        // three quarters of a round or more, but less than one (QKV of the text tower at 2 048 tokens: 192 tiles): one tile per block
        // on the four-wave kernel (1 319 vs 1 163 TFLOP/s for the eight-wave per-tile kernel; below that fill, or with the fp32
        // read-modify-write epilogue, the eight-wave forms stay ahead)
        const int64_t total = tm256 * tn;
        if (policy_in == 0 && !no_w4 && total < kCUs && total * 4 >= kCUs * 3 && K % 128 == 0 && K >= 256 && M % 256 == 0 && N % 256 == 0 && n_cover == N) {
            extern int launch_gemm_w4(const void*, int64_t, const void*, int64_t, int64_t, int, int, int, int, int, const EpiParams&, hipStream_t);
            const int rc = launch_gemm_w4(A, lda, W, ldw, M, N, K, n_cover, P2T_BF16, std::is_same<Epi, EpiQkvRope<bf16_t>>::value ? P2T_EPI_QKV_ROPE : P2T_EPI_STORE, ep, s);
            if (rc != P2T_ERR_UNSUPPORTED) return rc;
        }
    }
    if constexpr (kHasW4Pairs<Epi>) {
        // at most half a round of 256 x 256 tiles and a long K (FFN-down of the text tower at 2 048 tokens: 128 tiles, K = 14 336):
        // every tile as a split-K pair on the four-wave kernel -- two CUs per tile, half the K loop each (gemm_w4.hip, PAIRS_ONLY):
        // 185 vs 204 us for the eight-wave pair kernel; at K = 4 096 (o-proj) the slab hand-off costs more than it saves (81 vs 73 us)
        const int64_t total = tm256 * tn;
        if (policy_in == 0 && !no_w4 && fix_ws && total * 2 <= kCUs && total * 8 >= kCUs * 3 && K % 128 == 0 && K >= 8192 && M % 256 == 0 && N % 256 == 0 &&
            n_cover == N && (int64_t)256 * (lda > ldw ? lda : ldw) * 2 < ((int64_t)1 << 32) && fix_bytes >= kFixHeader + (size_t)total * kFixSlab) {
            SplitFix f4;
            f4.flag = (unsigned*)fix_ws;
            f4.timeout = fault_word_ptr();
            f4.slab = (float*)((char*)fix_ws + kFixHeader);
            f4.epoch = fix_epoch;
            return launch_gemm_w4_pairs<Epi>(A, lda, W, ldw, M, N, K, (int)total, ep, f4, s);
        }
    }
    if (tile == 0 && fix_ws && tm256 * tn * 2 <= kCUs && tm256 * tn * 8 >= kCUs * 3 && (K >> 5) >= 256 &&
        fix_bytes >= kFixHeader + (size_t)(tm256 * tn) * kFixSlab) {
        // at most half a round of 256-row tiles and a long K (FFN-down of the text tower: 128 tiles, K = 14336): every
        // tile as two K halves on two CUs (half a tile time + the slab hand-off) instead of a full round of 128-row
        // tiles (0.675 tile times)
        SplitFix fix;
        fix.flag = (unsigned*)fix_ws;
        fix.timeout = fault_word_ptr();
        fix.slab = (float*)((char*)fix_ws + kFixHeader);
        fix.epoch = fix_epoch;
        const int64_t n_tail = tm256 * tn;
        gemm_nt_mfma_tail_kernel<Epi><<<dim3((unsigned)(2 * n_tail)), 512, 0, s>>>(
            (const bf16_t*)A, lda, (const bf16_t*)W, ldw, M, N, K, (int)tm256, (int)tn, 0, (int)n_tail, n_cover, ep, fix);
        P2T_LAUNCH_CHECK();
        return P2T_OK;
    }
    if (tile == 256 || (tile != 128 && cost256 <= cost128)) {
        // split-K tail: leftover tiles of the last partial round (at most half a round) run as two K halves each
        const int64_t total = tm256 * tn, n_full = (total / kCUs) * kCUs, n_tail = total - n_full;
        // measured (profiles/r01_microbench_v3.log): pays for long K (FFN-down +8 %) or many full rounds (QKV +5 %);
        // with K = 2560 and only two full rounds (o-proj) the slab hand-off costs more than the half round it saves
        const int ns = K >> 5;
        const bool worth = ns >= 128 || (ns >= 64 && n_full >= 4 * kCUs);
        if (tile == 0 && fix_ws && n_full > 0 && n_tail > 0 && n_tail <= 128 && worth &&
            fix_bytes >= kFixHeader + (size_t)n_tail * kFixSlab) {
            SplitFix fix;
            fix.flag = (unsigned*)fix_ws;
            fix.timeout = fault_word_ptr();
            fix.slab = (float*)((char*)fix_ws + kFixHeader);
            fix.epoch = fix_epoch;
            gemm_nt_mfma_tail_kernel<Epi><<<dim3((unsigned)(n_full + 2 * n_tail)), 512, 0, s>>>(
                (const bf16_t*)A, lda, (const bf16_t*)W, ldw, M, N, K, (int)tm256, (int)tn, (int)n_full, (int)n_tail, n_cover, ep, fix);
            P2T_LAUNCH_CHECK();
            return P2T_OK;
        }
        return launch_cfg<8, Epi>(A, lda, W, ldw, M, N, K, n_cover, ep, s);
    }
    return launch_cfg<4, Epi>(A, lda, W, ldw, M, N, K, n_cover, ep, s);
}

