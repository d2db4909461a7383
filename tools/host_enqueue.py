"""Host time to ENQUEUE one cfg3 training step (ContrastiveTrainer.step with the towers on side streams and the next batch announced)
against its wall time: what 8 ranks sharing a node's CPU cores have to fit beside each other.  python tools/host_enqueue.py"""
import sys, time, torch
sys.path.insert(0, "prot2text-v2-esm3_amd")
import p2t_hip as P
from p2t_hip import specs, synth
dev = torch.device("cuda:0")
esm_name, llama_name, _, B, Tp, Tt = specs.CONFIGS["cfg3"]; layer = 16
esm, llama = specs.esm_spec(esm_name), specs.llama_spec(llama_name)
model = P.Esm2LlamaInstructForCausalLM.from_specs(esm, llama, specs.adapter_spec(esm, llama), dtype=torch.bfloat16, device=dev, seed=0)
tr = P.ContrastiveTrainer(model, output_llm_layer=layer, overlap_streams=True)
pid, pmask = synth.protein_batch(1, B, Tp); tid, tmask = synth.text_batch(1, B, Tt)
batch = {k: torch.from_numpy(v).to(dev) for k, v in dict(protein_input_ids=pid, protein_attention_mask=pmask, description_input_ids=tid, description_attention_mask=tmask).items()}
for _ in range(3): tr.step(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(10): tr.step(batch, next_batch=batch if i < 9 else None)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3*(t1-t0)/10:.1f} ms/step, wall {1e3*(t2-t0)/10:.1f} ms/step")
