"""Ragged-batch sweep: samples/s of the cfg3 step on 64 pairs with log-normal protein lengths, padded vs
length-sorted + trimmed segments, over the planner's floor_tokens / multiple (picks the defaults of
ContrastiveTrainer(trim_padding=True)).  python tools/ragged_sweep.py > gpurun_out/ragged_sweep.log"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "prot2text-v2-esm3_amd"))
import p2t_hip as P                                             # noqa: E402
from p2t_hip import specs, synth                                # noqa: E402
from p2t_hip.data import sort_batch_by_length                   # noqa: E402


def main():
    dev = torch.device("cuda:0")
    esm_name, llama_name, _, _, Tp, Tt = specs.CONFIGS["cfg3"]
    esm, llama = specs.esm_spec(esm_name), specs.llama_spec(llama_name)
    ad = specs.adapter_spec(esm, llama)
    model = P.Esm2LlamaInstructForCausalLM.from_specs(esm, llama, ad, dtype=torch.bfloat16, device=dev, seed=0)
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    rs = np.random.RandomState(0)
    lens = np.clip(np.round(rs.lognormal(5.75, 0.6, B)), 16, Tp).astype(int).tolist()
    tl = np.clip(np.round(rs.lognormal(4.0, 0.5, B)), 4, Tt).astype(int).tolist()
    pid, pm = synth.protein_batch(77, B, Tp, lens)
    tid, tm = synth.text_batch(77, B, Tt, lengths=tl)
    Tmax = int(max(lens))
    host = dict(protein_input_ids=torch.from_numpy(pid[:, :Tmax].copy()), protein_attention_mask=torch.from_numpy(pm[:, :Tmax].copy()),
                description_input_ids=torch.from_numpy(tid), description_attention_mask=torch.from_numpy(tm))
    to_dev = lambda b: {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()}
    padded, srt = to_dev(host), to_dev(sort_batch_by_length(host))

    def rate(tr, b, n=3):
        tr.step(b)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            tr.step(b)
        torch.cuda.synchronize()
        return B * n / (time.perf_counter() - t)

    print(f"B={B} mean length {np.mean(lens):.0f} longest {Tmax}", flush=True)
    print(f"padded                      : {rate(P.ContrastiveTrainer(model), padded):7.1f} samples/s", flush=True)
    for mult in (64, 128):
        for floor in (2048, 4096, 8192, 16384, 32768):
            tr = P.ContrastiveTrainer(model, trim_padding=True, trim_multiple=mult, trim_floor_tokens=floor, overlap_streams=False)
            segs = [(b - a, t) for a, b, t, _ in tr._segments(srt, B, Tmax)]
            r1 = rate(tr, srt)
            tr.overlap_streams = True
            r2 = rate(tr, srt)
            print(f"multiple {mult:3d} floor {floor:5d}: {r1:7.1f} samples/s  (two streams {r2:7.1f})  tokens {sum(n * t for n, t in segs):6d}  {segs}",
                  flush=True)
            del tr


if __name__ == "__main__":
    main()
