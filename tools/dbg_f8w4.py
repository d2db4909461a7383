import sys, torch
sys.path.insert(0, "prot2text-v2-esm3_amd")
from p2t_hip import ops
dev = torch.device("cuda:0")
M, N, K = 4096, 4096, 512
a = torch.empty((M, K), dtype=torch.float32, device=dev); w = torch.empty((N, K), dtype=torch.float32, device=dev)
ops.fill_hash_(a, 11, "a", 1.0); ops.fill_hash_(w, 11, "w", 0.05)
a8, sa = ops.quant_rows_fp8(a); w8, sw = ops.quant_rows_fp8(w)
for epi in (0, 3):
    ref = ops.gemm_nt_fp8(a8, sa, w8, sw, None, n=N, k=K, epilogue=epi, tile=256).float()
    for rep in range(4):
        got = ops.gemm_nt_fp8(a8, sa, w8, sw, None, n=N, k=K, epilogue=epi, tile=4).float()
        ncol = got.shape[1]
        cw = 128 if epi == 3 else 256
        d = (got - ref).abs().view(M // 256, 256, ncol // cw, cw).amax(dim=(1, 3))
        bad = (d > 0).nonzero()
        print("epi", epi, "rep", rep, "bad tiles", bad.shape[0], bad[:5].tolist(), float(d.max()))
        if bad.shape[0]:
            tm, tn = bad[0].tolist()
            blk = (got - ref)[tm*256:(tm+1)*256, tn*cw:(tn+1)*cw].abs()
            print("   rows", (blk.amax(1) > 0).nonzero().flatten().tolist()[:40], "cols", (blk.amax(0) > 0).nonzero().flatten().tolist()[:40])
