"""p2t_hip -- MI355X-native contrastive-alignment hot path of Prot2Text-V2 (ESM2 + Llama).

Public surface = the reference's names for this path (models/__init__.py:1,4 and
scripts/train_contrast.py), implemented on libp2t_hip.so (hand-written HIP for gfx950):

    from p2t_hip import (Esm2LlamaInstructConfig, ModalityAdapterConfig, ModalityAdapter,
                         Esm2LlamaInstructForCausalLM, BatchInfoNCELoss, SegmentedBatchInfoNCELoss,
                         readout_embeddings, get_sequence_embeddings, get_description_embeddings,
                         teacher_forcing_forward_pass, ContrastiveTrainer,
                         train_epoch, eval_epoch, run_epochs,         # scripts/train_contrast.py:400-519, 650-701
                         iterative_generation_loop, inference_epoch)  # scripts/generate_instruct.py:50-147

`synth` and `specs` (pure numpy: synthetic weights/batches, tower shapes) and the host-side `data`
(tokeniser, collater, prefetcher) and `training_state` (lr schedule, checkpoint formats) modules import
without the library; everything else needs the built .so and a GPU and raises otherwise -- there is no
CPU fallback.
"""
from . import specs, synth  # noqa: F401

__all__ = ["specs", "synth", "Esm2LlamaInstructConfig", "ModalityAdapterConfig", "ModalityAdapter",
           "Esm2LlamaInstructForCausalLM", "EsmEncoder", "LlamaDecoder", "BatchInfoNCELoss",
           "SegmentedBatchInfoNCELoss", "readout_embeddings", "l2_normalize", "get_sequence_embeddings",
           "get_description_embeddings", "teacher_forcing_forward_pass", "ContrastiveTrainer", "ops",
           "EsmSequenceTokenizer", "ContrastiveCollater", "DevicePrefetcher", "sort_batch_by_length", "CosineWarmupSchedule", "save_checkpoint",
           "load_model_checkpoint", "load_optimizer_scheduler_checkpoint", "train_epoch", "eval_epoch", "run_epochs",
           "iterative_generation_loop", "inference_epoch", "load_and_merge_adapter"]

_LAZY = {
    "Esm2LlamaInstructConfig": "configuration", "ModalityAdapterConfig": "configuration",
    "ModalityAdapter": "modeling", "Esm2LlamaInstructForCausalLM": "modeling", "EsmEncoder": "modeling",
    "LlamaDecoder": "modeling", "BatchInfoNCELoss": "contrastive", "SegmentedBatchInfoNCELoss": "contrastive",
    "readout_embeddings": "contrastive", "l2_normalize": "contrastive", "get_sequence_embeddings": "contrastive",
    "get_description_embeddings": "contrastive", "teacher_forcing_forward_pass": "contrastive",
    "ContrastiveTrainer": "contrastive",
    "EsmSequenceTokenizer": "data", "ContrastiveCollater": "data", "DevicePrefetcher": "data", "sort_batch_by_length": "data",
    "CosineWarmupSchedule": "training_state", "save_checkpoint": "training_state",
    "load_model_checkpoint": "training_state", "load_optimizer_scheduler_checkpoint": "training_state",
    "train_epoch": "loop", "eval_epoch": "loop", "run_epochs": "loop", "iterative_generation_loop": "loop", "inference_epoch": "loop", "load_and_merge_adapter": "lora",
}


def __getattr__(name):
    if name in _LAZY:
        import importlib
        return getattr(importlib.import_module(f".{_LAZY[name]}", __name__), name)
    if name == "ops":
        import importlib
        return importlib.import_module(".ops", __name__)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
