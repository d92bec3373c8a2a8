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

# The image engine runs the two halves of a large batch on two HIP streams (hip/engine.py), which overlap only when the runtime gives
# them different hardware queues.  It has 4 by default and shares them among every stream in use (null stream, the two part streams,
# RCCL's, a copy stream ...): ask for 8 unless the user chose a number.  Read when the HIP runtime starts, i.e. before the first device
# call - importing the package first is enough; set after that it has no effect (the engine still gives the same results, the halves
# may then run one after the other).
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
