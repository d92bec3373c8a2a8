"""MI355X-native CLIP multi-label scoring path (ViT image tower + prompt-tuned text
tower + cosine-logit matrix), behind the reference's module surface.

Import as ``leclip_amd`` (see ``/leclip_amd/__init__.py``).  Sub-modules:

* ``synth``      deterministic, counter-based synthetic weights / images
* ``hip``        ctypes binding of the C-ABI library (``include/leclip_hip.h``) + tower engines
* ``clip``       ``build_model`` / ``CLIP`` / ``tokenize`` mirror of the reference's ``clip`` package
* ``trainers``   ``TextEncoder`` / ``PromptLearner`` / ``CustomCLIP`` + the trainer plug-in surface
* ``evaluation`` mAP over 80 labels
* ``parallel``   one-process-per-GPU sharded scoring with an RCCL all-gather of logits
"""
import os as _os

__version__ = "0.1.0"



def configure(hw_queues: int = 8) -> bool:
    """Opt-in process-wide HIP runtime setting for the stream-part schedule (hip/engine.py): the image engine runs the two halves of a
    large batch on two HIP streams, which overlap only when the runtime gives them different hardware queues.  It has 4 by default and
    shares them among every stream in use (null stream, the two part streams, RCCL's, a copy stream ...); this asks for ``hw_queues``
    unless the user already chose a number (GPU_MAX_HW_QUEUES).  The variable is read when the HIP runtime starts, so call this before
    the first device call; children started afterwards (DataLoader workers, torchrun ranks) inherit it.  Entry points call it
    (train_caption.py, bench.py); importing the package does NOT.  Returns False when the runtime had already started (the setting is
    then without effect: same results, the two halves may run one after the other - the engine says so once)."""
    import sys
    started = False
    torch = sys.modules.get("torch")
    if torch is not None and hasattr(torch, "cuda"):
        started = bool(torch.cuda.is_initialized())
    global _HW_QUEUE_STATE
    if "GPU_MAX_HW_QUEUES" in _os.environ:
        _HW_QUEUE_STATE = _HW_QUEUE_STATE or "user"
    else:
        _os.environ["GPU_MAX_HW_QUEUES"] = str(int(hw_queues))
        _HW_QUEUE_STATE = "late" if started else "configured"
    return not started


_HW_QUEUE_STATE = None


def hw_queue_state() -> str:
    """"user" (GPU_MAX_HW_QUEUES came from the environment), "configured" (configure() set it before the runtime started), "late"
    (configure() ran after the runtime had started) or "default" (never configured: the runtime's own 4 queues)."""
    if _HW_QUEUE_STATE is None:
        return "user" if "GPU_MAX_HW_QUEUES" in _os.environ else "default"
    return _HW_QUEUE_STATE
