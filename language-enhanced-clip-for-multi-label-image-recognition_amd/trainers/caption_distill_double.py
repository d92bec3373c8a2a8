"""Learnable-prompt module surface: ``TextEncoder``, ``PromptLearner``, ``CustomCLIP`` and the trainer plug-in.

Mirrors reference ``project/my_code/trainers/Caption_distill_double.py`` (cited per class) on top of the HIP
engines.  Attribute names equal the reference's because they are also the checkpoint keys (``ctx``,
``ctx_double``, ``ctx_evidence``, ``temperature``, ``spatial_T``, ``ranking_scale``, ``token_prefix``,
``token_suffix``, ``token_suffix_nocls``).  Deliberate differences, each forced by a defect or by scope:

* ``CustomCLIP.forward`` unpacks the 6-tuple ``PromptLearner.forward`` returns (the shipped code unpacks 5
  values at :326/:341 and cannot run - SURVEY.md "mismatch" note); the arithmetic is the one written at :330-335.
* with ``CTX_INIT`` the reference leaves ``ctx_double`` / ``ctx_evidence`` undefined (:116-124 vs :158-160); here they
  start as copies of the initial context.
* ``class_token_position`` "middle"/"front" build only ``prompts`` in the reference and then fail at the return
  (:262-308); only "end" is accepted here.
* ``DenseCLIP`` (:354-559) needs the ResNet ``attnpool`` and is outside the north-star scope.
"""
from __future__ import annotations

import os
import os.path as osp
import time
from collections import OrderedDict
from typing import List, Optional

import numpy as np
import torch
from torch import nn

from .. import clip as clip_pkg
from ..clip import clip
from ..config import CfgNode
from ..registry import TRAINER_REGISTRY


def load_clip_to_cpu(cfg):
    """Build the CLIP model on the host from a local checkpoint (reference :38-54, which hard-codes an RN50 path).
    ``cfg.MODEL.BACKBONE.PATH`` names the checkpoint; ``"synthetic[:seed[:dist]]"`` builds seeded random weights of
    the named architecture (there are no pretrained weights offline)."""
    name = cfg.MODEL.BACKBONE.NAME
    if name not in clip._MODELS and name != "tiny":
        raise KeyError(f"unknown backbone {name}; available: {clip_pkg.available_models()}")
    path = str(cfg.MODEL.BACKBONE.get("PATH", "") or "")
    if path.startswith("synthetic"):
        from .. import synth
        parts = path.split(":")
        seed = int(parts[1]) if len(parts) > 1 else 0
        dist = parts[2] if len(parts) > 2 else "cond"
        state_dict = synth.make_state_dict(synth.ARCHS[name], seed=seed, dist=dist)
    else:
        if not osp.isfile(path):
            raise FileNotFoundError(f'CLIP checkpoint not found at "{path}" (set MODEL.BACKBONE.PATH)')
        try:
            state_dict = torch.jit.load(path, map_location="cpu").eval().state_dict()
        except RuntimeError:
            state_dict = torch.load(path, map_location="cpu")
    return clip_pkg.build_model(state_dict)


class TextEncoder(nn.Module):
    """Reference :72-101.  ``forward(prompts, tokenized_prompts, if_embedding=True, if_sequence=False)``:
    prompts are embeddings [n, T, d] (or token ids when ``if_embedding`` is False); returns fp32 features
    [n, E] pooled at ``tokenized_prompts.argmax(-1)``, or [n, T, E] when ``if_sequence``."""

    def __init__(self, clip_model):
        super().__init__()
        self.transformer = clip_model.transformer
        self.positional_embedding = clip_model.positional_embedding
        self.ln_final = clip_model.ln_final
        self.text_projection = clip_model.text_projection
        self.token_embedding = clip_model.token_embedding
        self.dtype = clip_model.dtype
        self._clip = [clip_model]  # not a sub-module: avoids registering the towers twice

    def forward(self, prompts, tokenized_prompts, if_embedding: bool = True, if_sequence: bool = False):
        if not prompts.is_cuda:
            raise RuntimeError("TextEncoder.forward needs HIP device tensors (no CPU fallback on the product path)")
        eng = self._clip[0].text_engine(prompts.device)
        if not if_embedding:
            return eng.encode_tokens(prompts, if_sequence=if_sequence)
        if torch.is_grad_enabled() and prompts.requires_grad and not if_sequence:
            from ..hip.autograd import TextTowerFunction     # prompt tuning: gradient w.r.t. the prompt embeddings
            return TextTowerFunction.apply(prompts, eng, tokenized_prompts)
        return eng.encode_prompts(prompts, tokenized_prompts, if_sequence=if_sequence)


class PromptLearner(nn.Module):
    """Reference :104-308 (CoOp-style context vectors in front of frozen class-name embeddings)."""

    def __init__(self, cfg, classnames: List[str], clip_model, nctx: Optional[int] = None):
        super().__init__()
        n_cls = len(classnames)
        n_ctx = cfg.TRAINER.Caption.N_CTX if nctx is None else nctx
        ctx_init = cfg.TRAINER.Caption.CTX_INIT
        csc = bool(cfg.TRAINER.Caption.CSC)
        ctx_dim = clip_model.ln_final.weight.shape[0]
        clip_imsize = clip_model.visual.input_resolution
        cfg_imsize = cfg.INPUT.SIZE[0]
        assert cfg_imsize == clip_imsize, f"cfg_imsize ({cfg_imsize}) must equal to clip_imsize ({clip_imsize})"
        table = clip_model.token_embedding.weight.detach().float()

        if ctx_init:
            ctx_init = ctx_init.replace("_", " ")
            n_ctx = len(ctx_init.split(" "))
            prompt = clip.tokenize(ctx_init, truncate=True)
            ctx_vectors = table[prompt[0, 1:1 + n_ctx].to(table.device)].clone()
            ctx_vectors_double = ctx_vectors.clone()
            ctx_vectors_evidence = ctx_vectors.clone()
            prompt_prefix = ctx_init
        else:
            shape = (n_cls, n_ctx, ctx_dim) if csc else (n_ctx, ctx_dim)
            ctx_vectors = torch.empty(shape)
            ctx_vectors_double = torch.empty(shape)
            ctx_vectors_evidence = torch.empty(n_ctx, ctx_dim)  # generic even under CSC, as :146-151
            for t in (ctx_vectors, ctx_vectors_double, ctx_vectors_evidence):
                nn.init.normal_(t, std=0.02)
            prompt_prefix = " ".join(["X"] * n_ctx)

        self.ctx = nn.Parameter(ctx_vectors)
        self.ctx_double = nn.Parameter(ctx_vectors_double)
        self.ctx_evidence = nn.Parameter(ctx_vectors_evidence)
        self.temperature = nn.Parameter(torch.tensor(3.0))
        self.spatial_T = nn.Parameter(torch.tensor(3.0))
        self.ranking_scale = nn.Parameter(torch.tensor(4.0))

        classnames = [name.replace("_", " ") for name in classnames]
        name_lens = [len(clip.encode_text(name)) for name in classnames]
        prompts = [prompt_prefix + " " + name + "." for name in classnames]
        tokenized_prompts = torch.cat([clip.tokenize(p, truncate=True) for p in prompts])
        tokenized_nocls = torch.cat([clip.tokenize(prompt_prefix + ".", truncate=True)] * n_cls)
        dev = table.device
        embedding = table[tokenized_prompts.to(dev)]
        embedding_nocls = table[tokenized_nocls.to(dev)]
        # saved with the checkpoint but ignored on load (CDD.py:929-938): recomputed from the current class names
        self.register_buffer("token_prefix", embedding[:, :1, :].clone())               # SOS
        self.register_buffer("token_suffix", embedding[:, 1 + n_ctx:, :].clone())       # class tokens, '.', EOT, pad
        self.register_buffer("token_suffix_nocls", embedding_nocls[:, 1 + n_ctx:, :].clone())

        self.n_cls, self.n_ctx = n_cls, n_ctx
        self.tokenized_prompts = tokenized_prompts
        self.name_lens = name_lens
        self.class_token_position = cfg.TRAINER.Caption.CLASS_TOKEN_POSITION
        if self.class_token_position != "end":
            raise ValueError(f'class_token_position "{self.class_token_position}" is not runnable in the reference '
                             f'either (:262-308); use "end"')

    def forward(self, neg_prompt_wcls: bool = True):
        """-> (prompts, prompts_neg, prompts_evidence, temperature, spatial_T, ranking_scale); each prompts tensor
        is fp32 [n_cls, 77, dim] = cat(prefix, ctx, suffix) (:199-259, :308)."""
        from ..hip import ops
        if not self.ctx.is_cuda:
            raise RuntimeError("PromptLearner.forward needs the module on a HIP device (no CPU fallback)")
        prefix = self.token_prefix.float().contiguous()
        suffix = self.token_suffix.float().contiguous()
        suffix_neg = suffix if neg_prompt_wcls else self.token_suffix_nocls.float().contiguous()

        def cat(ctx, suf):
            if torch.is_grad_enabled() and ctx.requires_grad:
                from ..hip.autograd import PromptAssembleFunction
                return PromptAssembleFunction.apply(ctx, prefix, suf)
            return ops.prompt_assemble(prefix, ctx.detach().float().contiguous(), suf, None, torch.float32)

        return (cat(self.ctx, suffix), cat(self.ctx_double, suffix_neg), cat(self.ctx_evidence, suffix_neg),
                self.temperature, self.spatial_T, self.ranking_scale)


class CustomCLIP(nn.Module):
    """Reference :310-352 - global cosine logits x 4.0 between the image (or caption-as-image) features and the
    learnable-prompt text features."""

    def __init__(self, cfg, classnames, clip_model):
        super().__init__()
        self.prompt_learner = PromptLearner(cfg, classnames, clip_model)
        self.tokenized_prompts = self.prompt_learner.tokenized_prompts
        self.image_encoder = clip_model.visual
        self.text_encoder = TextEncoder(clip_model)
        self.logit_scale = clip_model.logit_scale
        self.dtype = clip_model.dtype
        self.model = clip_model
        self._text_cache = None  # (ctx version, features): prompts are constant between updates
        # momentum ("EMA") copy of the prompt learner (reference :555-559 `_momentum_update`, :545-553 `copy_params`;
        # cfg.TRAIN.ema / cfg.TRAIN.momentum): m <- momentum * m + (1 - momentum) * p after every training forward
        self.ema = bool(cfg.TRAIN.get("ema", False))
        self.momentum = float(cfg.TRAIN.get("momentum", 0.999))
        if self.ema:
            import copy
            self.prompt_learner_m = copy.deepcopy(self.prompt_learner)
            self.model_pairs = [[self.prompt_learner, self.prompt_learner_m]]
            self.copy_params()

    @torch.no_grad()
    def copy_params(self):
        for model, model_m in self.model_pairs:
            for param, param_m in zip(model.parameters(), model_m.parameters()):
                param_m.data.copy_(param.data)
                param_m.requires_grad = False

    @torch.no_grad()
    def _momentum_update(self):
        for model, model_m in self.model_pairs:
            for param, param_m in zip(model.parameters(), model_m.parameters()):
                param_m.data.mul_(self.momentum).add_(param.data, alpha=1.0 - self.momentum)

    @torch.no_grad()
    def momentum_logits(self, image_features: torch.Tensor, logit_scale: float = 4.0) -> torch.Tensor:
        """Scores of the momentum prompts on already-encoded features (the `logits_m_` of reference :516-523)."""
        from ..hip import ops
        prompts = self.prompt_learner_m()[0]
        feats = self.text_encoder(prompts, self.tokenized_prompts.to(prompts.device))
        return ops.l2norm_logits(image_features, feats, logit_scale)

    def class_text_features(self) -> torch.Tensor:
        ctx = self.prompt_learner.ctx
        key = (ctx._version, ctx.data_ptr(), self.model.dtype)
        if self.training or self._text_cache is None or self._text_cache[0] != key:
            prompts = self.prompt_learner()[0]
            feats = self.text_encoder(prompts, self.tokenized_prompts.to(prompts.device))
            self._text_cache = (key, feats)
        return self._text_cache[1]

    def forward(self, image=None, captions=None, if_test: bool = False):
        from ..hip import ops
        logit_scale = 4.0  # reference :333-334 (not logit_scale.exp())
        training = torch.is_grad_enabled() and self.prompt_learner.ctx.requires_grad and self.training
        with torch.no_grad():   # both "image" encoders are frozen (reference :762-765)
            if if_test or image is not None:
                if not training and hasattr(self.image_encoder, "score"):
                    # inference: ln_post + projection + normalise + x4.0 cosine logits are one kernel at the tower's tail
                    return self.image_encoder.score(image, self.class_text_features(), logit_scale), None, None, None
                image_features = self.image_encoder(image)
            else:
                image_features = self.text_encoder(captions, None, if_embedding=False, if_sequence=False)
        if training:
            from ..hip.autograd import CosineLogitsFunction
            prompts = self.prompt_learner()[0]
            text_features = self.text_encoder(prompts, self.tokenized_prompts.to(prompts.device))
            self._text_cache = None
            logits = CosineLogitsFunction.apply(image_features, text_features, logit_scale)
            logits_m = None
            if self.ema:    # reference :516-523: update the momentum copy, score it without gradient
                self._momentum_update()
                logits_m = self.momentum_logits(image_features, logit_scale)
            return logits, None, None, logits_m
        text_features = self.class_text_features()
        logits = ops.l2norm_logits(image_features, text_features, logit_scale)
        return logits, None, None, None


@TRAINER_REGISTRY.register()
class Caption_distill_double:
    """Trainer plug-in (reference :565-938) reduced to the hot path's callers: ``build_model``,
    ``model_inference``, ``test``, ``load_model`` / ``save_model`` with the reference's checkpoint layout.
    ``forward_backward`` is the prompt-tuning step on the text-tower backward kernels."""

    def __init__(self, cfg, classnames: Optional[List[str]] = None, test_loader=None, evaluator=None):
        self.cfg = cfg
        self.check_cfg(cfg)
        self.device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        if classnames is None:
            from ..datasets import coco_object_categories
            classnames = coco_object_categories
        self.classnames = list(classnames)
        self.test_loader = test_loader
        self.evaluator = evaluator
        self._models = OrderedDict()
        self.epoch = 0
        self.build_model()

    def check_cfg(self, cfg):
        assert cfg.TRAINER.Caption.PREC in ["fp16", "fp32", "amp", "bf16"]

    def build_model(self):
        cfg = self.cfg
        print(f"Loading CLIP (backbone: {cfg.MODEL.BACKBONE.NAME})")
        clip_model = load_clip_to_cpu(cfg)
        prec = cfg.TRAINER.Caption.PREC
        if prec in ("fp32", "amp"):
            clip_model.float()          # reference :746-748
        elif prec == "bf16":
            clip_model.float()
            clip_pkg.convert_weights(clip_model, torch.bfloat16)
        if cfg.TRAIN.MODEL != "CustomCLIP":
            raise NotImplementedError(f"TRAIN.MODEL={cfg.TRAIN.MODEL}: only CustomCLIP wraps a ViT (reference :755-760)")
        name = cfg.TRAIN.get("MODEL_NAME", "default")
        model = CustomCLIP(cfg, self.classnames, clip_model)
        for pname, param in model.named_parameters():       # reference :762-765
            param.requires_grad_("prompt_learner." in pname and "prompt_learner_m." not in pname)
        model.to(self.device)
        model.eval()
        self._models[name] = model
        setattr(self, f"model_{name}", model)
        return model

    def get_model_names(self):
        return list(self._models.keys())

    def model_inference(self, input, name):
        """Reference :567-568: ``self.model_<name>(input, if_test=True)``."""
        return self._models[name](input, if_test=True)

    def parse_batch_test(self, batch):
        return batch["img"].to(self.device, non_blocking=True), batch["label"]

    def parse_batch_train(self, batch):
        return batch["img"].to(self.device), batch["label"].to(self.device)

    def build_optim(self):
        """SGD on the prompt learner only + cosine schedule with a constant-LR warm-up epoch (dassl/optim/optimizer.py:13-137,
        lr_scheduler.py:10-154 with the shipped OPTIM keys)."""
        o = self.cfg.OPTIM
        model = self._models[self.get_model_names()[0]]
        params = [p for p in model.prompt_learner.parameters() if p.requires_grad]
        self.optim = torch.optim.SGD(params, lr=o.LR, momentum=o.MOMENTUM, weight_decay=o.WEIGHT_DECAY)
        self.sched = torch.optim.lr_scheduler.CosineAnnealingLR(self.optim, T_max=max(int(o.MAX_EPOCH), 1))
        self._base_lr = o.LR
        if o.WARMUP_EPOCH > 0 and o.WARMUP_TYPE == "constant":
            for g in self.optim.param_groups:
                g["lr"] = o.WARMUP_CONS_LR
        return self.optim

    def update_lr(self):
        o = self.cfg.OPTIM
        self.epoch += 1
        if self.epoch == o.WARMUP_EPOCH and o.WARMUP_TYPE == "constant":
            for g in self.optim.param_groups:
                g["lr"] = self._base_lr
        elif self.epoch > o.WARMUP_EPOCH:
            self.sched.step()

    def forward_backward(self, batch):
        """One prompt-tuning step (reference :789-897, fp32 branch).  ``batch["img"]`` is either tokenised captions
        [B,77] int64 - the reference's texts-as-images feed, ``model(None, captions)`` - or images [B,3,R,R]
        (CoOp-style tuning on the frozen image tower, BASELINE config 3).  Loss: ``ranking_loss(scale_=1, margin_=1)``
        for LOSSFUNC == "double_ranking" (:806-808), BCE-with-logits for "bce" (trainers/utils.py:21-23)."""
        from .utils import norm_logits_BCEloss, ranking_loss
        if getattr(self, "optim", None) is None:
            self.build_optim()
        name = self.get_model_names()[0]
        model = self._models[name]
        model.train()
        inp, label = self.parse_batch_train(batch)
        if inp.dtype in (torch.int64, torch.int32):
            output = model(None, inp.long())[0]
        else:
            output = model(inp, None)[0]
        lf = self.cfg.TRAIN.LOSSFUNC
        if lf == "double_ranking":
            loss = ranking_loss(output, label, scale_=1.0, margin_=1)
        elif lf == "bce":
            loss = norm_logits_BCEloss(output, label.float())
        else:
            raise NotImplementedError(f"loss function {lf} not implemented")
        if not torch.isfinite(loss):
            raise FloatingPointError("Loss is infinite or NaN!")   # dassl/engine/trainer.py:224-226
        self.optim.zero_grad()
        loss.backward()
        self.optim.step()
        model._text_cache = None
        return {f"loss_{lf}": loss.item(), "loss": loss.item()}

    @torch.no_grad()
    def test(self, split=None, mode="test"):
        """Score every batch of the test loader with every registered model and feed the evaluator
        (reference :589-732 without the sliding-window / co-occurrence post-processing, rows N2/N3)."""
        assert self.test_loader is not None and self.evaluator is not None
        self.evaluator.reset()
        name = self.get_model_names()[0]
        for batch in self.test_loader:
            images, labels = self.parse_batch_test(batch)
            logits = self.model_inference(images, name)[0]
            self.evaluator.process(logits.float().cpu(), labels)
        return self.evaluator.evaluate()

    # ------------------------------------------------------------------ checkpoints (dassl/utils/torchtools.py:27-82)
    def save_model(self, epoch: int, directory: str, is_best: bool = False):
        for name, model in self._models.items():
            sd = OrderedDict((k, v.detach().cpu()) for k, v in model.prompt_learner.state_dict().items())
            folder = osp.join(directory, name)
            os.makedirs(folder, exist_ok=True)
            fpath = osp.join(folder, f"model.pth.tar-{epoch}")
            torch.save({"state_dict": sd, "epoch": epoch, "optimizer": None, "scheduler": None}, fpath)
            with open(osp.join(folder, "checkpoint"), "w") as f:
                f.write(osp.basename(fpath) + "\n")
            if is_best:
                torch.save({"state_dict": sd, "epoch": epoch}, osp.join(folder, "model-best.pth.tar"))

    def load_model(self, directory: str, epoch: Optional[int] = None):
        """Reference :906-938: read ``<dir>/<name>/model.pth.tar[-E]``, drop ``token_prefix`` / ``token_suffix``
        (recomputed from the current class names), load non-strict."""
        if not directory:
            print("Note that load_model() is skipped as no pretrained model is given")
            return
        for name, model in self._models.items():
            model_file = "model.pth.tar" if epoch is None else f"model.pth.tar-{epoch}"
            model_path = osp.join(directory, name, model_file)
            if not osp.exists(model_path):
                raise FileNotFoundError(f'Model not found at "{model_path}"')
            checkpoint = torch.load(model_path, map_location="cpu")
            state_dict = OrderedDict((k[7:] if k.startswith("module.") else k, v) for k, v in checkpoint["state_dict"].items())
            for key in ("token_prefix", "token_suffix", "token_suffix_nocls"):
                state_dict.pop(key, None)
            print(f'Loading weights to {name} from "{model_path}" (epoch = {checkpoint.get("epoch")})')
            model.prompt_learner.load_state_dict(state_dict, strict=False)
            model._text_cache = None
