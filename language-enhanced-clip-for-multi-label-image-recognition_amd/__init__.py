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
__version__ = "0.1.0"
