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
* ``DenseCLIP`` (:354-559): the image side of its test branch is defined for the ResNet ``attnpool`` only; the ViT analogue used
  here is stated in the class docstring.  Its caption-as-image training branch (:473-541) involves no image tower and is built as
  written.
"""
from __future__ import annotations

import contextlib
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


def _load_caption_text_feats(path: str) -> torch.Tensor:
    """The reference pickles the tensor (`pickle.dump(all_text_feats, f)`, generate_caption_text_features.py:96-97) and reads it back with
    pickle at import time (:35-36); torch.save files are accepted too."""
    if not osp.isfile(path):
        raise FileNotFoundError(f'TEST.caption_text_feats: no file at "{path}"')
    try:
        obj = torch.load(path, map_location="cpu")
    except Exception:
        import pickle
        with open(path, "rb") as f:
            obj = pickle.load(f)
    return torch.as_tensor(obj).float()


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
        self.text_beside_image = True   # training: the text tower's forward is enqueued while the image tower's stream parts run (same values)
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
        text_features = None
        with torch.no_grad():   # both "image" encoders are frozen (reference :762-765)
            if if_test or image is not None:
                if not training and hasattr(self.image_encoder, "score"):
                    # inference: ln_post + projection + normalise + x4.0 cosine logits are one kernel at the tower's tail
                    return self.image_encoder.score(image, self.class_text_features(), logit_scale), None, None, None
                if training and self.text_beside_image and hasattr(self.image_encoder, "forward_beside"):
                    # the learnable prompts' text tower does not depend on the images: enqueued beside the frozen image tower's stream parts
                    def text_side():
                        with torch.enable_grad():
                            p = self.prompt_learner()[0]
                            return self.text_encoder(p, self.tokenized_prompts.to(p.device))
                    image_features, text_features = self.image_encoder.forward_beside(image, text_side)
                else:
                    image_features = self.image_encoder(image)
            else:
                image_features = self.text_encoder(captions, None, if_embedding=False, if_sequence=False)
        if training:
            from ..hip.autograd import CosineLogitsFunction
            if text_features is None:
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


class DenseCLIP(CustomCLIP):
    """Global + LOCAL branch for a ViT.  The reference's ``DenseCLIP`` (:354-559) exists for the ResNet
    only: its per-position features come from ``attnpool``'s v / c projections applied to the 7x7 feature map (:409-410).
    The definition used here for a ViT - the reference never wrote one - is the direct analogue: the per-position features
    are the PATCH TOKENS of the last block taken through the same ``ln_post`` and ``proj`` as the class token; everything
    downstream is the reference's arithmetic (:434-462): normalise, similarities against the "negative" (``ctx_double``)
    prompts, spatial softmax over positions at ``TRAIN.spatial_SCALE_image`` (or ``spatial_T.exp()``), optionally the
    evidence prompts' winner-take-all weighting (``TRAINER.Caption.use_evidence``), ``logits_local = sum_p scale * s * prob``.
    The top-10 caption-feature mixing of the global feature (:444-448) is applied when caption features are given
    (``set_caption_text_feats`` / ``TEST.caption_text_feats``; the reference's own file is not in its repository, ``caption_features``
    produces the same kind of table from tokenised captions).
    ``if_test=True`` returns (logits_, logits_local, None, None, None) like the reference's test branch; ``if_test=False`` is the
    reference's caption-as-image TRAINING branch (:473-541, ``_forward_captions``), which needs no image tower at all."""

    def __init__(self, cfg, classnames, clip_model, return_interm_layers=False, nctx=None):
        super().__init__(cfg, classnames, clip_model)
        self.cfg = cfg
        self.prompt_text_features = None
        # the reference's `caption_text_feats` (:35-36: normalised EOT features of its ~220 k ChatGLM captions, a module-level global
        # loaded from a pickle that is not in the repository): optional here - set_caption_text_feats() / TEST.caption_text_feats
        self.caption_text_feats = None

    def set_caption_text_feats(self, feats: Optional[torch.Tensor]):
        """[N, E] normalised caption features for the test branch's top-10 mixing (:444-448), or None to switch it off."""
        if feats is not None:
            dev = self.prompt_learner.ctx.device
            feats = feats.detach().to(device=dev, dtype=torch.float32).contiguous()
            if feats.dim() != 2 or feats.shape[0] < 10:
                raise ValueError("caption_text_feats must be [N >= 10, E]")
            n = feats.shape[0]
            npad = (n + 63) // 64 * 64               # the similarity GEMM's row granularity: padded ONCE here, not per test batch
            if npad != n:
                padded = torch.zeros((npad, feats.shape[1]), dtype=torch.float32, device=dev)
                padded[:n] = feats
                feats = padded[:n]                   # a view of the padded table: ops.topk_mix finds the pad rows behind it
        self.caption_text_feats = feats

    @torch.no_grad()
    def caption_features(self, captions: torch.Tensor) -> torch.Tensor:
        """generate_caption_text_features.py:82-88: normalised EOT-row features of tokenised captions [n, 77] (what the reference
        pickles as caption_text_feats)."""
        from ..hip import ops
        captions = captions.to(self.prompt_learner.ctx.device).long().contiguous()
        return ops.l2norm_rows_(self.text_encoder(captions, None, if_embedding=False, if_sequence=False).float().contiguous().clone())

    def _prompt_features(self):
        """text_features / text_features_neg (/ text_features_evidence), cached like the reference (:421-439)."""
        key = tuple((p._version, p.data_ptr()) for p in (self.prompt_learner.ctx, self.prompt_learner.ctx_double, self.prompt_learner.ctx_evidence))
        if self.prompt_text_features is None or self.prompt_text_features["key"] != key:
            prompts, prompts_double, prompts_evidence, _, _, _ = self.prompt_learner()
            toks = self.tokenized_prompts.to(prompts.device)
            feats = {"key": key, "text_features": self.text_encoder(prompts, toks), "text_features_neg": self.text_encoder(prompts_double, toks)}
            if self.cfg.TRAINER.Caption.get("use_evidence", False):
                feats["text_features_evidence"] = self.text_encoder(prompts_evidence, toks)
            # the similarity GEMM's W operand: rows = normalised negative (| evidence) prompt features, padded to 64 / 128 rows
            from ..hip import ops
            rows = [ops.l2norm_rows_(feats["text_features_neg"].clone())]
            if "text_features_evidence" in feats:
                rows.append(ops.l2norm_rows_(feats["text_features_evidence"].clone()))
            c = rows[0].shape[0]
            cp = (c + 63) // 64 * 64
            w = torch.zeros((cp * len(rows), rows[0].shape[1]), dtype=torch.float32, device=rows[0].device)
            for i, r in enumerate(rows):
                w[i * cp:i * cp + c] = r
            feats["w_local"], feats["c_pad"] = w, cp
            # scale * normalised class prompts, zero-padded: the B operand of the mixed global feature's plain contraction (:449)
            scale = float(self.prompt_learner.temperature.exp()) if self.cfg.TRAIN.IF_LEARN_SCALE else 4.0
            wg = torch.zeros((cp, rows[0].shape[1]), dtype=torch.float32, device=rows[0].device)
            wg[:c] = ops.l2norm_rows_(feats["text_features"].clone()) * scale
            feats["w_global"] = wg
            self.prompt_text_features = feats
        return self.prompt_text_features

    def forward(self, image=None, captions=None, if_test: bool = False, model_name: str = "ema"):
        from ..hip import ops
        if not if_test:
            return self._forward_captions(captions)
        with torch.no_grad():
            f = self._prompt_features()
            dense = self.image_encoder.dense_features(image)                 # [B, T, E] fp32
            b, t, e = dense.shape
            logit_scale = float(self.prompt_learner.temperature.exp()) if self.cfg.TRAIN.IF_LEARN_SCALE else 4.0
            if self.caption_text_feats is None:
                logits_ = ops.l2norm_logits(dense[:, 0].contiguous(), f["text_features"], logit_scale)
            else:
                # :444-449: the normalised global feature is averaged with the mean of its 10 most similar caption features (NOT
                # re-normalised), then scored against the normalised class prompts
                if self.caption_text_feats.shape[1] != e:
                    raise ValueError(f"caption_text_feats are {self.caption_text_feats.shape[1]}-wide, the model's features {e}")
                glob = ops.topk_mix(ops.l2norm_rows_(dense[:, 0].contiguous().clone()), self.caption_text_feats, 10)
                logits_ = ops.gemm(glob, f["w_global"], out_dtype=torch.float32)[:, :f["text_features"].shape[0]].contiguous()
            flat = ops.l2norm_rows_(dense.reshape(b * t, e))                  # image_features / norm (:434), every position
            sim = ops.gemm(flat, f["w_local"], out_dtype=torch.float32)       # exact-fp32 MFMA: [B*T, c_pad (x2)]
            tmp = float(self.prompt_learner.spatial_T.exp()) if self.cfg.TRAIN.IF_LEARN_spatial_SCALE else float(self.cfg.TRAIN.spatial_SCALE_image)
            evi = f["c_pad"] if "text_features_evidence" in f else -1
            logits_local = ops.local_pool(sim, b, t, 1, f["text_features"].shape[0], evi, tmp, logit_scale)
        return logits_, logits_local, None, None, None

    def _text_features_of(self, learner, with_grad: bool):
        """(text_features, text_features_neg, text_features_evidence or None) of one prompt learner: the two or three prompt sets go
        through the text tower as ONE batch of 160 / 240 prompts (one kernel sequence instead of three; with_grad: one backward)."""
        prompts, prompts_double, prompts_evidence, _, _, _ = learner()
        use_evi = bool(self.cfg.TRAINER.Caption.get("use_evidence", False))
        sets = [prompts, prompts_double] + ([prompts_evidence] if use_evi else [])
        toks = self.tokenized_prompts.to(prompts.device)
        n = prompts.shape[0]
        if with_grad:
            feats = self.text_encoder(torch.cat(sets, dim=0), torch.cat([toks] * len(sets), dim=0))
        else:
            with torch.no_grad():
                feats = self.text_encoder(torch.cat([p.detach() for p in sets], dim=0), torch.cat([toks] * len(sets), dim=0))
        return feats[:n], feats[n:2 * n], (feats[2 * n:] if use_evi else None)

    def _forward_captions(self, captions):
        """The reference's texts-as-images TRAINING branch (:473-541): a caption's 77 token positions through the frozen text tower are
        its "image" - the EOT row the global feature (:476), all rows the local ones (:477) - scored against the three learnable prompt
        sets: logits_ = scale * cos(global, ctx prompts) (:494), logits_local = spatial pooling of cos(positions, ctx_double prompts)
        under `text_mask`, with the ctx_evidence prompts' winner-take-all weighting when use_evidence (:495-513), and - TRAIN.ema - the
        same two scores from the momentum copy of the prompt learner, without gradient (:515-539).
        -> (logits_, logits_local, image_features [L, B, E] normalised, text_features normalised, logits_m_, logits_local_m)."""
        from ..hip import ops
        from ..hip.autograd import CosineLogitsFunction, LocalPoolFunction
        cfg = self.cfg
        if captions is None:
            raise ValueError("DenseCLIP.forward(if_test=False) takes tokenised captions [B, 77] (the reference's model(None, captions))")
        if cfg.TRAIN.IF_LEARN_SCALE or cfg.TRAIN.IF_LEARN_spatial_SCALE:
            # no shipped config learns them (train_caption.py:114-115, every configs/trainers/*.yaml): the gradient to temperature /
            # spatial_T is not formed here
            raise NotImplementedError("TRAIN.IF_LEARN_SCALE / IF_LEARN_spatial_SCALE: learnable scales are not supported in the tuning step")
        logit_scale, tmp = 4.0, float(cfg.TRAIN.spatial_SCALE_text)
        captions = captions.to(self.prompt_learner.ctx.device).long().contiguous()
        training = torch.is_grad_enabled() and self.prompt_learner.ctx.requires_grad
        # The step's three passes through the text tower - the frozen caption encoder, the learnable prompt sets (with gradient) and the momentum
        # copy's prompt sets - feed the scores only: with `text_beside_image` the first and the third are enqueued on side streams of their own
        # (the image engine's part streams: no image tower runs in this step) beside the second on the caller's stream, then joined.
        side = None
        if self.text_beside_image and captions.is_cuda:
            from ..hip.engine import _part_streams
            cur = torch.cuda.current_stream(captions.device)
            side = _part_streams(captions.device, 2)
            for st in side:
                st.wait_stream(cur)

        def on(st):
            return torch.cuda.stream(st) if side is not None else contextlib.nullcontext()

        def hand_over(*ts):    # allocated on a side stream, consumed on the caller's
            if side is not None:
                for t_ in ts:
                    if t_ is not None:
                        t_.record_stream(cur)

        with on(side[0] if side else None), torch.no_grad():   # the caption encoder is the frozen text tower (:762-765)
            seq = self.text_encoder(captions, None, if_embedding=False, if_sequence=True).float().contiguous()     # [B, L, E]
            b, l, e = seq.shape
            _, eot_flat = ops.eot_index(captions)
            image_feature_ = ops.gather_rows(seq.view(b * l, e), eot_flat)                                          # :476
            hand_over(seq, image_feature_)
        tf_m = tfn_m = tfe_m = None
        if self.ema:
            with on(side[1] if side else None), torch.no_grad():
                self._momentum_update()
                tf_m, tfn_m, tfe_m = self._text_features_of(self.prompt_learner_m, False)
                tf_m = tf_m.float().contiguous()
                hand_over(tf_m, tfn_m, tfe_m)
        text_features, text_features_neg, text_features_evi = self._text_features_of(self.prompt_learner, training)
        if side is not None:
            for st in side:
                cur.wait_stream(st)
        self._text_cache = None
        self.prompt_text_features = None
        logits_ = CosineLogitsFunction.apply(image_feature_, text_features, logit_scale)
        logits_local = LocalPoolFunction.apply(seq, text_features_neg, text_features_evi, captions, tmp, logit_scale)
        logits_m_, logits_local_m = None, None
        if self.ema:
            with torch.no_grad():
                logits_m_ = ops.l2norm_logits(image_feature_, tf_m, logit_scale)
                logits_local_m = LocalPoolFunction.apply(seq, tfn_m, tfe_m, captions, tmp, logit_scale)
        with torch.no_grad():
            image_features = ops.l2norm_rows_(seq.view(b * l, e).clone()).view(b, l, e).permute(1, 0, 2)
            text_features_n = ops.l2norm_rows_(text_features.detach().float().clone())
        return logits_, logits_local, image_features, text_features_n, logits_m_, logits_local_m


@TRAINER_REGISTRY.register()
class Caption_distill_double:
    """Trainer plug-in (reference :565-938 on dassl's TrainerBase / SimpleTrainer, dassl/engine/trainer.py:78-309) for the hot
    path's callers: ``build_model`` (per-name models, prompt learner registered with its optimizer and scheduler),
    ``train`` / ``before_epoch`` / ``run_epoch`` / ``after_epoch``, ``forward_backward`` (the prompt-tuning step, SURVEY.md
    §8f N1, data-parallel with ONE flat all-reduce of the context gradients), ``test`` (multi-model loop, sliding-window
    aggregation N2, co-occurrence modulation N3, global/local merge in the evaluator), ``save_model`` / ``load_model`` /
    ``resume_model_if_exist`` in the reference's checkpoint layout.  One process per GPU; the process group, if any, is the
    default one (RCCL on GPUs, gloo in the CPU tests)."""

    def __init__(self, cfg, classnames: Optional[List[str]] = None, test_loader=None, evaluator=None, train_loader=None):
        import torch.distributed as dist
        self.cfg = cfg
        self.check_cfg(cfg)
        self.device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        self.rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        if classnames is None:
            from ..datasets import coco_object_categories
            classnames = coco_object_categories
        self.classnames = list(classnames)
        self.train_loader_x = train_loader
        self.test_loader = test_loader
        self.val_loader = None
        self.evaluator = evaluator
        self._models, self._optims, self._scheds = OrderedDict(), OrderedDict(), OrderedDict()
        self.start_epoch = self.epoch = 0
        self.max_epoch = int(cfg.OPTIM.MAX_EPOCH)
        self.output_dir = cfg.OUTPUT_DIR
        self._cooc = None
        self._pipe = None                    # forward_backward(batch, next_batch=...): features handed from one call to the next (_step_pipelined)
        self.pipeline_image_tower = True     # False: every step computes its own image features first (the reference's order)
        self.build_model()

    # ------------------------------------------------------------------------------------------------ construction
    def check_cfg(self, cfg):
        assert cfg.TRAINER.Caption.PREC in ["fp16", "fp32", "amp", "bf16"]

    def register_model(self, name, model, optim=None, sched=None):
        """dassl/engine/trainer.py:80-106: what is registered is the PROMPT LEARNER (its state dict is the checkpoint)."""
        assert name not in self._models, "Found duplicate model names"
        self._models[name], self._optims[name], self._scheds[name] = model, optim, sched

    def _model_names_cfg(self):
        names = self.cfg.TEST.get("multi_model", False)      # reference :740 `names = cfg.TEST.multi_model` (a list of names)
        if isinstance(names, (list, tuple)) and len(names) > 0:
            return [str(n) for n in names]
        return [str(self.cfg.TRAIN.get("MODEL_NAME", "default"))]

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
        if cfg.TRAIN.MODEL not in ("CustomCLIP", "DenseCLIP"):
            raise NotImplementedError(f"model {cfg.TRAIN.MODEL} not implemented")      # reference :755-760
        import copy
        names = self._model_names_cfg()
        for i, name in enumerate(names):
            model_cls = DenseCLIP if cfg.TRAIN.MODEL == "DenseCLIP" else CustomCLIP
            model = model_cls(cfg, self.classnames, clip_model if i == len(names) - 1 else copy.deepcopy(clip_model))
            for pname, param in model.named_parameters():       # reference :762-765
                param.requires_grad_("prompt_learner." in pname and "prompt_learner_m." not in pname)
            if cfg.MODEL.get("INIT_WEIGHTS", ""):
                sd = torch.load(cfg.MODEL.INIT_WEIGHTS, map_location="cpu")
                model.prompt_learner.load_state_dict(sd.get("state_dict", sd), strict=False)
            model.to(self.device)
            feats_path = str(cfg.TEST.get("caption_text_feats", "") or "")
            if feats_path and isinstance(model, DenseCLIP):
                model.set_caption_text_feats(_load_caption_text_feats(feats_path))
            self._sync_prompt_learner(model)
            model.eval()
            setattr(self, f"model_{name}", model)
            optim, sched = self._build_optim_for(model)
            self.register_model(name, model.prompt_learner, optim, sched)
        self.optim, self.sched = self._optims[names[0]], self._scheds[names[0]]
        return getattr(self, f"model_{names[0]}")

    def _sync_prompt_learner(self, model):
        """World > 1: every rank must tune and score the SAME prompts.  The learnable context is drawn from torch's RNG
        (reference :128-151), which is not seeded identically on every rank unless SEED is set, so rank 0's parameters and
        buffers are broadcast once after construction - what DDP's constructor does for the reference (:786-787)."""
        if self.world <= 1:
            return
        import torch.distributed as dist
        with torch.no_grad():
            for t in list(model.prompt_learner.parameters()) + list(model.prompt_learner.buffers()):
                if t.is_floating_point() or t.dtype in (torch.int64, torch.int32):
                    dist.broadcast(t.data, src=0)
            if getattr(model, "ema", False):
                model.copy_params()
        model._text_cache = None

    def _build_optim_for(self, model):
        """SGD on the prompt learner only + cosine schedule with a constant-LR warm-up epoch (dassl/optim/optimizer.py:13-137,
        lr_scheduler.py:10-154 with the shipped OPTIM keys)."""
        o = self.cfg.OPTIM
        params = [p for p in model.prompt_learner.parameters() if p.requires_grad]
        optim = torch.optim.SGD(params, lr=o.LR, momentum=o.MOMENTUM, weight_decay=o.WEIGHT_DECAY)
        sched = torch.optim.lr_scheduler.CosineAnnealingLR(optim, T_max=max(int(o.MAX_EPOCH), 1))
        self._base_lr = o.LR
        if o.WARMUP_EPOCH > 0 and o.WARMUP_TYPE == "constant":
            for g in optim.param_groups:
                g["lr"] = o.WARMUP_CONS_LR
        return optim, sched

    def build_optim(self):
        """(Re)build optimizer and scheduler of the first model - kept for callers that construct the trainer and then
        change OPTIM keys; build_model already registered a pair."""
        name = self.get_model_names()[0]
        self._optims[name], self._scheds[name] = self._build_optim_for(getattr(self, f"model_{name}"))
        self.optim, self.sched = self._optims[name], self._scheds[name]
        return self.optim

    def get_model_names(self, names=None):
        real = list(self._models.keys())
        if names is None:
            return real
        names = [names] if isinstance(names, str) else list(names)
        for n in names:
            assert n in real
        return names

    def set_model_mode(self, mode="train", names=None):
        for name in self.get_model_names(names):
            model = getattr(self, f"model_{name}")
            if mode == "train":
                model.train()
            elif mode in ("test", "eval"):
                model.eval()
            else:
                raise KeyError(mode)

    def model_inference(self, input, name):
        """Reference :567-568: ``self.model_<name>(input, if_test=True)``."""
        return getattr(self, f"model_{name}")(input, if_test=True)

    def parse_batch_test(self, batch):
        """Reference :899-... returns (input, label, input_blocks): ``img_blocks`` is the per-scale list of window batches
        a DatasetWrapperWithBlock item carries (data_manager.py:336-341); None without TEST.multi_scale."""
        blocks = batch.get("img_blocks") if isinstance(batch, dict) else None
        if blocks is not None:
            blocks = [b.to(self.device, non_blocking=True) for b in blocks]
        return batch["img"].to(self.device, non_blocking=True), batch["label"], blocks

    def parse_batch_train(self, batch):
        return batch["img"].to(self.device), batch["label"].to(self.device)

    # ------------------------------------------------------------------------------------------------ training loop
    def update_lr(self, names=None):
        """Per-epoch schedule step (dassl/engine/trainer.py:214-219) with the constant-LR warm-up of lr_scheduler.py:41-82."""
        o = self.cfg.OPTIM
        self._lr_epoch = getattr(self, "_lr_epoch", 0) + 1
        for name in self.get_model_names(names):
            optim, sched = self._optims[name], self._scheds[name]
            if optim is None:
                continue
            if self._lr_epoch == o.WARMUP_EPOCH and o.WARMUP_TYPE == "constant":
                for g in optim.param_groups:
                    g["lr"] = self._base_lr
            elif self._lr_epoch > o.WARMUP_EPOCH:
                sched.step()

    def train(self, start_epoch=None, max_epoch=None):
        """Generic training loop (dassl/engine/trainer.py:255-264)."""
        self.start_epoch = self.start_epoch if start_epoch is None else start_epoch
        self.max_epoch = self.max_epoch if max_epoch is None else max_epoch
        self.before_train()
        last = {}
        for self.epoch in range(self.start_epoch, self.max_epoch):
            self.before_epoch()
            last = self.run_epoch()
            self.after_epoch()
        self.after_train()
        return last

    def before_train(self):
        """dassl/engine/trainer.py:409-413: ALWAYS look for a checkpoint to continue from - in OUTPUT_DIR, or in RESUME when that is
        set (RESUME only overrides the directory) - so a restarted job with the same OUTPUT_DIR continues instead of overwriting
        its checkpoints from epoch 0."""
        directory = self.cfg.get("RESUME", "") or self.output_dir or ""
        if directory:
            self.start_epoch = self.resume_model_if_exist(directory)
        self.time_start = time.time()

    def after_train(self):
        print(f"Finished training ({time.time() - self.time_start:.1f} s)")

    def before_epoch(self):
        """Reference :571-574: rank 0 announces the epoch; the distributed sampler is re-seeded with it."""
        if self.rank == 0:
            print(f"before_epoch: {self.epoch}")
        sampler = getattr(self.train_loader_x, "sampler", None)
        if sampler is not None and hasattr(sampler, "set_epoch"):
            sampler.set_epoch(self.epoch)
        elif hasattr(self.train_loader_x, "set_epoch"):
            self.train_loader_x.set_epoch(self.epoch)

    def run_epoch(self):
        """dassl/engine/trainer.py:575-612 (TrainerX.run_epoch): one pass over train_loader_x, then the schedule step."""
        assert self.train_loader_x is not None, "no training loader was given to the trainer"
        self.set_model_mode("train")
        last = {}
        it = iter(self.train_loader_x)
        batch = next(it, None)
        self.batch_idx = -1
        while batch is not None:
            nxt = next(it, None)       # one batch of lookahead: its frozen-tower features are computed beside this step's backward (forward_backward)
            self.batch_idx += 1
            last = self.forward_backward(batch, next_batch=nxt)
            batch = nxt
        self._pipe = None
        self.update_lr()
        if self.rank == 0 and last:
            print(f"epoch [{self.epoch + 1}/{self.max_epoch}] loss {last['loss']:.4f} lr {self.optim.param_groups[0]['lr']:.3e}")
        return last

    def after_epoch(self):
        """Reference :576-587: rank 0 only - checkpoint every CHECKPOINT_FREQ epochs and after the last one."""
        if self.rank == 0:
            print(f"after_epoch: {self.epoch}")
            last_epoch = (self.epoch + 1) == self.max_epoch
            freq = int(self.cfg.TRAIN.CHECKPOINT_FREQ)
            meet_checkpoint_freq = (self.epoch + 1) % freq == 0 if freq > 0 else False
            if (meet_checkpoint_freq or last_epoch) and self.output_dir:
                self.save_model(self.epoch, self.output_dir)
        if self.world > 1:
            import torch.distributed as dist
            dist.barrier()      # nobody evaluates or resumes before the checkpoint is on disk

    def _allreduce_grads(self, params):
        """Data-parallel tuning step: mean of the context gradients over ranks, ONE collective on one flat fp32 buffer
        (3 x [16, 512] = 96 KiB for the generic context; DDP's bucketed all-reduce in the reference, :786-787)."""
        if self.world <= 1:
            return
        import torch.distributed as dist
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in params]
        flat = torch.cat([g.reshape(-1).float() for g in grads])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.div_(self.world)
        o = 0
        for p, g in zip(params, grads):
            n = g.numel()
            p.grad = flat[o:o + n].view_as(g).to(g.dtype)
            o += n

    def forward_backward(self, batch, next_batch=None):
        """One prompt-tuning step (reference :789-897, fp32 branch).  ``batch["img"]`` is either tokenised captions
        [B,77] int64 - the reference's texts-as-images feed, ``model(None, captions)`` - or images [B,3,R,R]
        (CoOp-style tuning on the frozen image tower, BASELINE config 3).  Loss: ``ranking_loss(scale_=1, margin_=1)``
        for LOSSFUNC == "double_ranking" (:806-808), BCE-with-logits for "bce" (trainers/utils.py:21-23).  Under
        WORLD_SIZE > 1 the batch is this rank's shard and the gradients are averaged over ranks before the step.

        ``next_batch`` (image batches on CustomCLIP; run_epoch passes the loader's next item): the frozen image tower does not depend on the
        prompts, so the NEXT batch's image features are computed on the tower's stream parts while this step's backward, all-reduce, optimizer
        step and the next step's text-tower forward run on the caller's stream (`_step_pipelined`) - the same kernels on the same values as
        the one-batch-at-a-time order, a step no longer than the longer of the two."""
        from .utils import norm_logits_BCEloss, ranking_loss
        name = self.get_model_names()[0]
        model = getattr(self, f"model_{name}")
        optim = self._optims[name]
        model.train()
        inp, label = self.parse_batch_train(batch)
        if ((next_batch is not None or self._pipe is not None) and self.pipeline_image_tower and type(model) is CustomCLIP and not model.ema
                and inp.is_cuda and inp.dtype not in (torch.int64, torch.int32) and model.text_beside_image
                and model.prompt_learner.ctx.requires_grad and self.cfg.TRAIN.LOSSFUNC in ("double_ranking", "bce")
                and hasattr(model.image_encoder, "forward_beside")):
            return self._step_pipelined(model, optim, inp, label, next_batch)
        self._pipe = None
        output_local = output_m = output_local_m = None
        if isinstance(model, DenseCLIP):
            # the reference's step as shipped (TRAIN.MODEL = "DenseCLIP", train_caption.py:110): `model(None, captions)` (:802)
            if inp.dtype not in (torch.int64, torch.int32):
                raise TypeError("DenseCLIP is tuned on tokenised captions [B, 77] (texts as images); image batches tune CustomCLIP")
            output, output_local, _, _, output_m, output_local_m = model(None, inp.long())
        elif inp.dtype in (torch.int64, torch.int32):
            output = model(None, inp.long())[0]      # CustomCLIP returns (logits, None, None, None) (:352): no momentum scores, no KL term
        else:
            output = model(inp, None)[0]
        lf = self.cfg.TRAIN.LOSSFUNC
        summary = {}
        if lf == "double_ranking":
            # :806-815: ranking loss on the global and - when the model has one - the local head; with the momentum copy's scores the
            # distillation term kl(log_softmax(output) || softmax(output_m)) + 10000 * kl(local || local_m), batchmean
            loss = ranking_loss(output, label, scale_=1.0, margin_=1)
            if output_local is not None:
                loss = loss + ranking_loss(output_local, label, scale_=1.0, margin_=1)
            if output_m is not None:
                kl = torch.nn.KLDivLoss(reduction="batchmean")
                logp = torch.nn.functional.log_softmax
                ema_loss = kl(logp(output, dim=-1), torch.softmax(output_m, dim=-1))
                if output_local is not None and output_local_m is not None:
                    ema_loss = ema_loss + kl(logp(output_local, dim=-1), torch.softmax(output_local_m, dim=-1)) * 10000
                r_loss = loss
                loss = loss + ema_loss
                vals = torch.stack([r_loss.detach(), ema_loss.detach()]).tolist()     # one host sync for the summary scalars
                summary = {"r_loss": vals[0], "ema_loss": vals[1]}
        elif lf == "bce":
            loss = norm_logits_BCEloss(output, label.float())
            if output_local is not None:
                loss = loss + norm_logits_BCEloss(output_local, label.float())
        else:
            raise NotImplementedError(f"loss function {lf} not implemented")
        if not torch.isfinite(loss):
            raise FloatingPointError("Loss is infinite or NaN!")   # dassl/engine/trainer.py:224-226
        optim.zero_grad()
        loss.backward()
        self._allreduce_grads([p for g in optim.param_groups for p in g["params"]])
        optim.step()
        model._text_cache = None
        if not summary:
            summary = {f"loss_{lf}": loss.item()}
        summary["loss"] = loss.item()
        return summary

    def _step_pipelined(self, model, optim, inp, label, next_batch):
        """forward_backward with one batch of lookahead (CustomCLIP on image batches).  State between calls: the image features of the batch
        about to be stepped on and the text features (with their autograd graph) of the current prompts, both produced during the previous call."""
        from ..hip.autograd import CosineLogitsFunction
        from .utils import norm_logits_BCEloss, ranking_loss

        def key_of(t):
            return (t.data_ptr(), tuple(t.shape), t._version)

        def text_forward():
            with torch.enable_grad():
                p = model.prompt_learner()[0]
                return model.text_encoder(p, model.tokenized_prompts.to(p.device))

        st = self._pipe
        if st is None or st["key"] != key_of(inp) or st["ctx_version"] != key_of(model.prompt_learner.ctx):
            feats, text = model.image_encoder.forward_beside(inp, text_forward)        # first step (or a batch nobody announced): nothing to reuse
        else:
            feats, text = st["feats"], st["text"]
        self._pipe = None
        output = CosineLogitsFunction.apply(feats, text, 4.0)                          # reference :333-334: scale 4, not logit_scale.exp()
        lf = self.cfg.TRAIN.LOSSFUNC
        loss = ranking_loss(output, label, scale_=1.0, margin_=1) if lf == "double_ranking" else norm_logits_BCEloss(output, label.float())
        optim.zero_grad()

        def rest_of_step():      # on the caller's stream, beside the next batch's image tower
            loss.backward()
            self._allreduce_grads([p for g in optim.param_groups for p in g["params"]])
            optim.step()
            model._text_cache = None
            return text_forward() if next_batch is not None else None                  # the NEXT step's text features, from the updated prompts
        if next_batch is not None:
            nxt, _ = self.parse_batch_train(next_batch)
            feats_n, text_n = model.image_encoder.forward_beside(nxt, rest_of_step)
            self._pipe = {"key": key_of(nxt), "ctx_version": key_of(model.prompt_learner.ctx), "feats": feats_n, "text": text_n}
        else:
            rest_of_step()                                                             # the loader's last batch: nothing to look ahead to
        val = loss.item()
        if val != val or val in (float("inf"), float("-inf")):
            raise FloatingPointError("Loss is infinite or NaN!")   # dassl/engine/trainer.py:224-226 (found one step late: the update has been made)
        return {f"loss_{lf}": val, "loss": val}

    # ------------------------------------------------------------------------------------------------------- testing
    def cooccurrence_matrix(self):
        """Row-normalised conditional co-occurrence matrix of reference :632-634 from ``freq_stats.pkl`` ({'adj', 'nums'},
        path in cfg.TEST.freq_stats, default ./freq_stats.pkl as in the reference :620)."""
        if self._cooc is None:
            import pickle
            from ..hip import ops
            path = self.cfg.TEST.get("freq_stats", "freq_stats.pkl")
            if not osp.isfile(path):
                raise FileNotFoundError(f'TEST.use_freq needs the co-occurrence statistics at "{path}" (set TEST.freq_stats)')
            with open(path, "rb") as f:
                result = pickle.load(f)
            self._cooc = ops.cooccurrence_matrix(result["adj"], result["nums"]).to(self.device)
        return self._cooc

    def _score_blocks(self, input_blocks, name):
        """Reference :639-652: every scale's windows [B, W_s, 3, R, R] go through the model as a [B*W_s, ...] batch (in chunks
        of DATALOADER.TEST.BATCH_SIZE windows); returns global scores [B, sum W_s, C] and local scores or None."""
        chunk = max(int(self.cfg.DATALOADER.TEST.BATCH_SIZE), 1)
        outs, outs_pos = [], []
        for blk in input_blocks:
            b, w = blk.shape[0], blk.shape[1]
            flat = blk.reshape(b * w, *blk.shape[2:])
            res = [self.model_inference(flat[s:s + chunk].contiguous(), name) for s in range(0, b * w, chunk)]
            outs.append(torch.cat([r[0] for r in res]).reshape(b, w, -1))
            if res[0][1] is not None:
                outs_pos.append(torch.cat([r[1] for r in res]).reshape(b, w, -1))
        return torch.cat(outs, dim=1), (torch.cat(outs_pos, dim=1) if outs_pos else None)

    @torch.no_grad()
    def test(self, split=None, mode="test"):
        """Reference :589-732.  Per batch and model: global scores (+ local scores when the model has a local branch);
        TEST.use_freq: co-occurrence modulation of the local scores (:632-636, N3); when the batch carries ``img_blocks``
        (TEST.multi_scale): window scores aggregated as 1.4 * s_ag + output (:654-668, N2); the evaluator merges global
        and local scores (GL_merge_rate).  Returns the first metric, like the reference (:726)."""
        from ..hip import ops
        assert self.evaluator is not None
        self.set_model_mode("eval")
        self.evaluator.reset()
        if split is None:
            split = self.cfg.TEST.SPLIT
        data_loader = self.val_loader if (split == "val" and self.val_loader is not None) else self.test_loader
        assert data_loader is not None, "no test loader was given to the trainer"
        names = self.get_model_names()
        if mode == "test" and len(names) > 1:
            # reference :697-700: evaluating with several models needs a fusion strategy, which it never added
            raise NotImplementedError("Can not use multi model when evaluating, fuse strategy need to be added")
        use_freq = bool(self.cfg.TEST.get("use_freq", False))

        # The reference hands every batch's scores to the evaluator with a blocking .cpu() (:676-677), which leaves the device idle
        # while the host prepares and enqueues the next batch (about 1 ms per 86 launches).  Same calls in the same order here, one
        # batch late: the scores go to pinned host memory asynchronously behind an event, and batch i - 1 is handed over while
        # batch i runs.
        def to_host(t):
            if t is None or not t.is_cuda:
                return t
            h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            h.copy_(t, non_blocking=True)
            return h

        def hand_over(item):
            ev, out_h, pos_h, lab = item
            if ev is not None:
                ev.synchronize()
            self.evaluator.process(out_h, lab, pos_h)

        if self.world > 1:
            return self._test_sharded(data_loader, names[0], mode, use_freq, to_host)
        pending = None
        for batch in data_loader:
            input, label, input_blocks = self.parse_batch_test(batch)
            for name in names:
                res = self.model_inference(input, name)
                output, output_pos = res[0].float(), (res[1].float() if res[1] is not None else None)
                if use_freq and output_pos is not None:
                    output_pos = ops.cooccurrence_adjust(output_pos, self.cooccurrence_matrix(), 0.5)
                if mode == "test" and input_blocks is not None:
                    output_blocks, output_pos_blocks = self._score_blocks(input_blocks, name)
                    output_final = ops.window_aggregate(output, output_blocks, threshold=0.3, weight=1.4)
                    output_pos_final = output_pos
                    if output_pos_blocks is not None:
                        b, w, c = output_pos_blocks.shape
                        if use_freq:
                            output_pos_blocks = ops.cooccurrence_adjust(output_pos_blocks.reshape(b * w, c), self.cooccurrence_matrix(), 0.5).reshape(b, w, c)
                        output_pos_final = ops.window_aggregate(output_pos, output_pos_blocks, threshold=0.3, weight=1.4)
                else:
                    output_final, output_pos_final = output, output_pos
            out_h, pos_h = to_host(output_final), to_host(output_pos_final)
            ev = None
            if output_final.is_cuda:
                ev = torch.cuda.Event()
                ev.record()
            if pending is not None:
                hand_over(pending)
            pending = (ev, out_h, pos_h, label)
        if pending is not None:
            hand_over(pending)
        results = self.evaluator.evaluate()
        return list(results.values())[0]

    def _test_sharded(self, data_loader, name, mode, use_freq, to_host):
        """test() under WORLD_SIZE > 1 (the north star's sharded evaluation; the reference scores everything on every rank, :589-732).
        Every rank iterates the same loader; of each batch rank r scores the images shard_bounds(B, r, world) and - of each scale's
        flattened window list [B * W_s] (N2: ~305 forwards per image) - its contiguous share, so the window work is spread even when a
        batch holds fewer images than there are ranks.  Scores stay on the device for the whole epoch: per image one row
        [global | local | max over this rank's windows | -min | the same two for the local scores], -inf where this rank has nothing.
        Windows enter the aggregation only through their per-class maximum and minimum (:654-660), and max / min over a union of
        windows is the max / min of the parts' extrema, so ONE all-gather of the epoch's rows at the end (parallel.all_gather_rows;
        RCCL over xGMI) followed by an elementwise maximum over ranks reproduces the single-process scores bit for bit - every
        window's and every image's scores are batch-invariant.  Then the same aggregation kernel on (max, min), one
        asynchronous copy to pinned host memory, and every rank's evaluator sees the whole set."""
        from ..hip import ops
        from .. import parallel
        dev = self.device
        ninf = float("-inf")
        rows, labels = [], []
        has_pos = has_win = False
        n_cls = None
        for batch in data_loader:
            input, label, input_blocks = self.parse_batch_test(batch)
            b = input.shape[0]
            lo, hi = parallel.shard_bounds(b, self.rank, self.world)
            model = getattr(self, f"model_{name}", None)
            n_cls = model.prompt_learner.n_cls if model is not None else len(self.classnames)
            out = pos = None
            if hi > lo:      # (more ranks than images in this batch: nothing of it is this rank's)
                res = self.model_inference(input[lo:hi].contiguous(), name)
                out = res[0].float()
                pos = res[1].float() if res[1] is not None else None
                n_cls = out.shape[1]
                if use_freq and pos is not None:
                    pos = ops.cooccurrence_adjust(pos, self.cooccurrence_matrix(), 0.5)
            cols = [torch.full((b, n_cls), ninf, dtype=torch.float32, device=dev) for _ in range(6)]
            if out is not None:
                cols[0][lo:hi] = out
            if pos is not None or isinstance(model, DenseCLIP):
                has_pos = True
            if pos is not None:
                cols[1][lo:hi] = pos
            if mode == "test" and input_blocks is not None:
                has_win = True
                ext = self._window_extrema(input_blocks, name, use_freq)
                cols[2], cols[3] = ext[0], ext[1]
                if ext[2] is not None:
                    cols[4], cols[5] = ext[2], ext[3]
            rows.append(torch.cat(cols, dim=1))
            labels.append(label)
        if not rows:
            results = self.evaluator.evaluate()
            return list(results.values())[0]
        local = torch.cat(rows, dim=0).contiguous()                         # [N, 6 C] on the device
        n = local.shape[0]
        gathered = parallel.all_gather_rows(local)                          # the ONE collective of the epoch
        full = gathered.view(self.world, n, -1).amax(dim=0)
        c = n_cls
        output, output_pos = full[:, :c].contiguous(), (full[:, c:2 * c].contiguous() if has_pos else None)
        if has_win:
            stack = torch.stack([full[:, 2 * c:3 * c], -full[:, 3 * c:4 * c]], dim=1).contiguous()         # [N, 2, C] = (max_w, min_w)
            output = ops.window_aggregate(output, stack, threshold=0.3, weight=1.4)
            if has_pos:
                stack = torch.stack([full[:, 4 * c:5 * c], -full[:, 5 * c:6 * c]], dim=1).contiguous()
                output_pos = ops.window_aggregate(output_pos, stack, threshold=0.3, weight=1.4)
        out_h, pos_h = to_host(output), to_host(output_pos)
        if output.is_cuda:
            torch.cuda.current_stream(dev).synchronize()
        self.evaluator.process(out_h, torch.cat(labels, dim=0), pos_h)
        results = self.evaluator.evaluate()
        return list(results.values())[0]

    def _window_extrema(self, input_blocks, name, use_freq):
        """This rank's share of a batch's windows -> per image (max over its windows, -min) of the global scores and, when the model has
        a local branch, of the (co-occurrence-adjusted) local scores: [B, C] each, -inf for images none of whose windows fell to this rank."""
        from ..hip import ops
        from .. import parallel
        chunk = max(int(self.cfg.DATALOADER.TEST.BATCH_SIZE), 1)
        acc = None
        for blk in input_blocks:
            b, w = blk.shape[0], blk.shape[1]
            flat = blk.reshape(b * w, *blk.shape[2:])
            lo, hi = parallel.shard_bounds(b * w, self.rank, self.world)
            res = [self.model_inference(flat[s:min(s + chunk, hi)].contiguous(), name) for s in range(lo, hi, chunk)]
            parts = []
            for k in (0, 1):
                if not res or res[0][k] is None:
                    parts.append(None)
                    continue
                sc = torch.cat([r[k].float() for r in res])
                if k == 1 and use_freq:
                    sc = ops.cooccurrence_adjust(sc, self.cooccurrence_matrix(), 0.5)
                c = sc.shape[1]
                hi_buf = torch.full((b * w, c), float("-inf"), dtype=torch.float32, device=sc.device)
                lo_buf = torch.full((b * w, c), float("-inf"), dtype=torch.float32, device=sc.device)
                hi_buf[lo:hi] = sc
                lo_buf[lo:hi] = -sc
                parts += [hi_buf.view(b, w, c).amax(dim=1), lo_buf.view(b, w, c).amax(dim=1)]
            if len(parts) == 2:      # (no local branch / empty share)
                parts = [parts[0], None, None, None] if parts[0] is None else parts + [None, None]
            if parts[0] is None:
                continue
            if acc is None:
                acc = parts
            else:
                acc = [a if p is None else (p if a is None else torch.maximum(a, p)) for a, p in zip(acc, parts)]
        if acc is None:     # this rank got no window of this batch at all
            b = input_blocks[0].shape[0]
            model = getattr(self, f"model_{name}")
            c = model.prompt_learner.n_cls
            e = torch.full((b, c), float("-inf"), dtype=torch.float32, device=self.device)
            return [e, e.clone(), None, None]
        return acc

    # ------------------------------------------------------------------ checkpoints (dassl/utils/torchtools.py:27-82, 126-165)
    def save_model(self, epoch: int, directory: str, is_best: bool = False, model_name: str = ""):
        """dassl/engine/trainer.py:119-143 + torchtools.save_checkpoint: ``<dir>/<name>/model.pth.tar-<epoch+1>`` holding the
        prompt learner's state dict, ``epoch + 1``, the optimizer's and the scheduler's state, plus the ``checkpoint``
        pointer file.  Rank 0 writes; other ranks return (all ranks hold identical prompts after the gradient all-reduce)."""
        if self.rank != 0:
            return
        import shutil
        for name in self.get_model_names():
            sd = OrderedDict((k[7:] if k.startswith("module.") else k, v.detach().cpu()) for k, v in self._models[name].state_dict().items())
            optim, sched = self._optims.get(name), self._scheds.get(name)
            state = {"state_dict": sd, "epoch": epoch + 1, "optimizer": None if optim is None else optim.state_dict(),
                     "scheduler": None if sched is None else sched.state_dict(), "lr_epoch": getattr(self, "_lr_epoch", 0)}
            folder = osp.join(directory, name)
            os.makedirs(folder, exist_ok=True)
            fpath = osp.join(folder, model_name or f"model.pth.tar-{epoch + 1}")
            torch.save(state, fpath)
            print(f'Checkpoint saved to "{fpath}"')
            with open(osp.join(folder, "checkpoint"), "w+") as f:
                f.write(osp.basename(fpath) + "\n")
            if is_best:
                shutil.copy(fpath, osp.join(folder, "model-best.pth.tar"))

    def resume_model_if_exist(self, directory: str) -> int:
        """dassl/engine/trainer.py:145-170 + torchtools.resume_from_checkpoint: follow the ``checkpoint`` pointer, restore the
        prompt learner, the optimizer and the scheduler, return the epoch to continue from."""
        names = self.get_model_names()
        if any(not osp.exists(osp.join(directory, name, "checkpoint")) for name in names):
            print("No checkpoint found, train from scratch")
            return 0
        print(f'Found checkpoint in "{directory}". Will resume training')
        start_epoch = 0
        for name in names:
            folder = osp.join(directory, name)
            with open(osp.join(folder, "checkpoint")) as f:
                fpath = osp.join(folder, f.readlines()[0].strip("\n"))
            print(f'Loading checkpoint from "{fpath}"')
            ck = torch.load(fpath, map_location="cpu")
            self._models[name].load_state_dict(ck["state_dict"], strict=False)
            if self._optims[name] is not None and ck.get("optimizer") is not None:
                self._optims[name].load_state_dict(ck["optimizer"])
            if self._scheds[name] is not None and ck.get("scheduler") is not None:
                self._scheds[name].load_state_dict(ck["scheduler"])
            self._lr_epoch = int(ck.get("lr_epoch", ck["epoch"]))
            start_epoch = int(ck["epoch"])
            getattr(self, f"model_{name}")._text_cache = None
            print(f"Previous epoch: {start_epoch}")
        return start_epoch

    def load_model(self, directory: str, epoch: Optional[int] = None):
        """Reference :906-938: read ``<dir>/<name>/model.pth.tar[-E]``, drop ``token_prefix`` / ``token_suffix``
        (recomputed from the current class names), load non-strict."""
        if not directory:
            print("Note that load_model() is skipped as no pretrained model is given")
            return
        for name in self.get_model_names():
            model_file = "model.pth.tar" if epoch is None else f"model.pth.tar-{epoch}"
            model_path = osp.join(directory, name, model_file)
            if not osp.exists(model_path):
                raise FileNotFoundError(f'Model not found at "{model_path}"')
            checkpoint = torch.load(model_path, map_location="cpu")
            state_dict = OrderedDict((k[7:] if k.startswith("module.") else k, v) for k, v in checkpoint["state_dict"].items())
            for key in ("token_prefix", "token_suffix", "token_suffix_nocls"):
                state_dict.pop(key, None)
            print(f'Loading weights to {name} from "{model_path}" (epoch = {checkpoint.get("epoch")})')
            self._models[name].load_state_dict(state_dict, strict=False)
            getattr(self, f"model_{name}")._text_cache = None
