#!/usr/bin/env python3
"""DEV-ONLY fixture generator: runs the *reference's own* Python on synthetic inputs and
writes the outputs to ``tests/golden/`` (data only - no reference source is copied).

Needs ``/root/reference`` (absent on the GPU box; the script exits cleanly there).
What is imported from the reference, by file path, unmodified:

* ``project/my_code/clip/model.py``            - CLIP / VisionTransformer / Transformer / build_model
* ``project/my_code/clip/simple_tokenizer.py`` - BPE tokenizer (with an identity ``ftfy`` stub)
* ``dassl/evaluation/evaluator.py``            - ``mAP`` / ``average_precision`` (pickle5 aliased to pickle)
* ``project/my_code/trainers/utils.py``        - ``ranking_loss`` / ``norm_logits_BCEloss``

``clip.tokenize`` (clip.py:185-221, not importable: torchvision) and ``TextEncoder`` /
``PromptLearner`` / ``CustomCLIP`` (Caption_distill_double.py, not importable: mmcv, yacs, a
module-level ``.cuda()``) are pinned through the importable reference modules they are built
from, applied in the order those files state.

Usage:  python oracle/make_golden.py            (from the repo root)
"""
import importlib.util
import os
import pickle
import sys
import types

sys.dont_write_bytecode = True
REF = "/root/reference/project/my_code"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    if not os.path.isdir(REF):
        print("reference tree not present - nothing to do")
        return 0
    import numpy as np
    import torch

    sys.path.insert(0, ROOT)
    from leclip_amd import synth

    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    os.makedirs(OUT, exist_ok=True)

    ref_model = _load("ref_clip_model", os.path.join(REF, "clip", "model.py"))
    ftfy = types.ModuleType("ftfy")
    ftfy.fix_text = lambda s: s
    sys.modules["ftfy"] = ftfy
    ref_tok = _load("ref_tok", os.path.join(REF, "clip", "simple_tokenizer.py"))
    tokenizer = ref_tok.SimpleTokenizer()

    # ------------------------------------------------------------------ tokens (a13)
    sys.modules["pickle5"] = pickle
    # class names / template come from datasets/data_helpers.py:13,169-252 (not importable:
    # pycocotools); they are data - read them by evaluating just those assignments.
    src = open(os.path.join(REF, "datasets", "data_helpers.py")).read()
    ns = {}
    start = src.index("coco_classname_synonyms = [")
    end = src.index("coco_object_categories = ")
    exec(src[start:end], ns)
    classnames = [syn[0] for syn in ns["coco_classname_synonyms"]]
    template = "a photo of a {}."
    assert f'prompt_template = "{template}"' in src

    def tokenize(texts, context_length=77, truncate=False):
        # the 15 lines of clip.py:185-221 around the imported tokenizer
        sot, eot = tokenizer.encoder["<|startoftext|>"], tokenizer.encoder["<|endoftext|>"]
        res = np.zeros((len(texts), context_length), dtype=np.int64)
        for i, t in enumerate(texts):
            ids = [sot] + tokenizer.encode(t) + [eot]
            if len(ids) > context_length:
                if not truncate:
                    raise RuntimeError("too long")
                ids = ids[:context_length]
                ids[-1] = eot
            res[i, :len(ids)] = ids
        return res

    n_ctx = 16
    prefix = " ".join(["X"] * n_ctx)
    names = [c.replace("_", " ") for c in classnames]
    tok_photo = tokenize([template.format(c) for c in names])
    tok_ctx = tokenize([prefix + " " + c + "." for c in names], truncate=True)
    tok_nocls = tokenize([prefix + "."] * len(names), truncate=True)
    name_lens = np.array([len(tokenizer.encode(c)) for c in names], dtype=np.int64)
    extra_texts = ["A photo of a cat&amp;dog!!", "  hello   WORLD  ", "it's 12 o'clock; i'd say.",
                   "naïve café — 3.5€", "x" * 300]
    tok_extra = tokenize(extra_texts, truncate=True)
    np.savez_compressed(os.path.join(OUT, "tokens_coco80.npz"),
                        classnames=np.array(classnames), tokens_photo=tok_photo, tokens_ctx16=tok_ctx,
                        tokens_ctx16_nocls=tok_nocls, name_lens=name_lens,
                        eot_photo=tok_photo.argmax(-1), eot_ctx16=tok_ctx.argmax(-1),
                        extra_texts=np.array(extra_texts), tokens_extra=tok_extra)
    print("tokens:", tok_photo[0, :10], tok_ctx[0, :22])

    # ------------------------------------------------------------------ helpers
    def build_ref(arch, seed, dist):
        sd = synth.make_state_dict(arch, seed=seed, dist=dist)
        m = ref_model.CLIP(arch.embed_dim, arch.image_resolution, arch.vision_layers, arch.vision_width,
                           arch.vision_patch_size, arch.context_length, arch.vocab_size,
                           arch.transformer_width, arch.transformer_heads, arch.transformer_layers)
        missing = m.load_state_dict(sd, strict=True)
        return m.float().eval(), sd

    def nld(t):
        return t.permute(1, 0, 2).contiguous().numpy()

    def hook_blocks(blocks, store, prefix_):
        hs = []
        for i, blk in enumerate(blocks):
            def mk(key, lnd=True):
                def fn(_m, _inp, out):
                    o = out[0] if isinstance(out, tuple) else out
                    store[key] = nld(o) if lnd else o.numpy()
                return fn
            p = f"{prefix_}block{i}."
            hs += [blk.ln_1.register_forward_hook(mk(p + "ln_1")),
                   blk.attn.register_forward_hook(mk(p + "attn_out")),
                   blk.ln_2.register_forward_hook(mk(p + "ln_2")),
                   blk.mlp.gelu.register_forward_hook(mk(p + "gelu")),
                   blk.register_forward_hook(mk(p + "out"))]
        return hs

    # ------------------------------------------------------------------ tiny per-stage (fixture 2)
    arch = synth.TINY
    m, sd = build_ref(arch, seed=1, dist="cond")
    img = torch.from_numpy(synth.make_images(3, arch.image_resolution, seed=11))
    st = {}
    hs = hook_blocks(m.visual.transformer.resblocks, st, "v.")
    hs.append(m.visual.ln_pre.register_forward_hook(lambda _m, _i, o: st.__setitem__("v.ln_pre", o.numpy())))
    hs.append(m.visual.ln_post.register_forward_hook(lambda _m, _i, o: st.__setitem__("v.ln_post", o.numpy())))
    st["v.feat"] = m.encode_image(img).numpy()
    for h in hs:
        h.remove()
    toks = torch.from_numpy(tok_photo[:5])
    hs = hook_blocks(m.transformer.resblocks, st, "t.")
    hs.append(m.ln_final.register_forward_hook(lambda _m, _i, o: st.__setitem__("t.ln_final", o.numpy())))
    st["t.feat"] = m.encode_text(toks).numpy()
    for h in hs:
        h.remove()
    lpi, lpt = m(img, toks)
    st["logits_per_image"] = lpi.numpy()
    st["logits_per_text"] = lpt.numpy()
    st["images"] = img.numpy()
    st["tokens"] = toks.numpy()
    np.savez_compressed(os.path.join(OUT, "tiny_stages.npz"), **st)
    print("tiny: feat", st["v.feat"].shape, "logits", lpi.shape, "keys", len(st))

    # ------------------------------------------------------------------ ViT-B/16 cfg 1 (fixture 3)
    arch = synth.VIT_B16
    out = {}
    for dist in ("cond", "default"):
        m, sd = build_ref(arch, seed=0, dist=dist)
        img = torch.from_numpy(synth.make_images(8, 224, seed=1234))
        toks = torch.from_numpy(tok_photo)
        fi = m.encode_image(img)
        ft = m.encode_text(toks)
        lpi, _ = m(img, toks)
        # CustomCLIP(if_test=True) as intended (Caption_distill_double.py:323-337), built from the
        # reference modules in the order TextEncoder.forward states (:86-100).
        ctx = torch.from_numpy(synth.make_ctx(n_ctx, arch.transformer_width, seed=0))
        tctx = torch.from_numpy(tok_ctx)
        emb = m.token_embedding(tctx)
        prompts = torch.cat([emb[:, :1], ctx.unsqueeze(0).expand(80, -1, -1), emb[:, 1 + n_ctx:]], dim=1)
        x = prompts + m.positional_embedding
        x = m.transformer(x.permute(1, 0, 2)).permute(1, 0, 2)
        x = m.ln_final(x)
        ftp = x[torch.arange(80), tctx.argmax(-1)] @ m.text_projection
        fin = fi / fi.norm(dim=-1, keepdim=True)
        ftn = ftp / ftp.norm(dim=-1, keepdim=True)
        lcc = 4.0 * fin @ ftn.t()
        # fixed prompts with the CustomCLIP scale as well
        ftn2 = ft / ft.norm(dim=-1, keepdim=True)
        lfix = 4.0 * fin @ ftn2.t()
        # caption-as-image branch (:338-352): captions = first 6 photo prompts
        cap = toks[:6]
        fc = m.encode_text(cap)
        lcap = 4.0 * (fc / fc.norm(dim=-1, keepdim=True)) @ ftn.t()
        d = dist + "."
        out.update({d + "image_features": fi.numpy(), d + "text_features": ft.numpy(),
                    d + "logits_clip": lpi.numpy(), d + "text_features_ctx16": ftp.numpy(),
                    d + "logits_custom_ctx16": lcc.numpy(), d + "logits_custom_fixed": lfix.numpy(),
                    d + "logits_custom_captions": lcap.numpy(),
                    d + "top5_clip": torch.topk(lpi, 5, dim=1).indices.numpy(),
                    d + "top5_custom_ctx16": torch.topk(lcc, 5, dim=1).indices.numpy()})
        ii = fin @ fin.t()
        print(dist, "img-img cos mean", float((ii.sum() - 8) / 56), "argmax", lpi.argmax(1).tolist())
        del m
    # generator drift guards
    out["guard.image0_head"] = synth.make_images(1, 224, seed=1234)[0, 0, 0, :16]
    out["guard.conv1_head"] = sd["visual.conv1.weight"].numpy().reshape(-1)[:16]
    np.savez_compressed(os.path.join(OUT, "vitb16_cfg1.npz"), **out)

    # ------------------------------------------------------------------ metric / loss KATs (4, 5)
    sys.path.insert(0, os.path.join(REF, "Dassl.pytorch-master"))
    np.deprecate = getattr(np, "deprecate", lambda f=None, **k: (f if f is not None else (lambda g: g)))
    kat = {}
    try:
        ev = _load("ref_evaluator", os.path.join(REF, "Dassl.pytorch-master", "dassl", "evaluation", "evaluator.py"))
        ref_map, ref_ap = ev.mAP, ev.average_precision
    except Exception as e:  # evaluator imports dassl.* ; fall back to exec of the two functions
        print("evaluator import failed (%s); executing the two metric functions only" % type(e).__name__)
        s = open(os.path.join(REF, "Dassl.pytorch-master", "dassl", "evaluation", "evaluator.py")).read()
        a, b = s.index("def average_precision"), s.index("@EVALUATOR_REGISTRY.register()\nclass MLClassification")
        ns2 = {"np": np}
        exec(s[a:b], ns2)
        ref_map, ref_ap = ns2["mAP"], ns2["average_precision"]
    rng = np.random.RandomState(3)
    cases = {}
    p = rng.randn(64, 80)
    t = (rng.rand(64, 80) < 0.1).astype(np.int64)
    cases["random"] = (t, p)
    p2 = np.round(rng.randn(40, 6), 1)            # ties
    t2 = (rng.rand(40, 6) < 0.3).astype(np.int64)
    cases["ties"] = (t2, p2)
    t3 = t2.copy(); t3[:, 2] = 0                  # all-negative class
    cases["allneg"] = (t3, p2)
    t4 = np.zeros((40, 6), dtype=np.int64); t4[7, :] = 1   # single positive
    cases["single"] = (t4, p2)
    for k, (tt, pp) in cases.items():
        kat[f"map.{k}.targets"] = tt
        kat[f"map.{k}.preds"] = pp
        kat[f"map.{k}.value"] = np.float64(ref_map(tt, pp))
        kat[f"map.{k}.ap"] = np.array([ref_ap(pp[:, c], tt[:, c]) for c in range(pp.shape[1])])
    try:
        tu = _load("ref_tutils", os.path.join(REF, "trainers", "utils.py"))
        yp = torch.from_numpy(rng.randn(5, 80).astype(np.float32))
        yt = torch.from_numpy((rng.rand(5, 80) < 0.1).astype(np.float32))
        kat["loss.pred"] = yp.numpy().copy()
        kat["loss.target"] = yt.numpy()
        kat["loss.ranking"] = np.float64(tu.ranking_loss(yp.clone(), yt, scale_=1.0, margin_=1.0))
        kat["loss.ranking_s2"] = np.float64(tu.ranking_loss(yp.clone(), yt))
        kat["loss.bce"] = np.float64(tu.norm_logits_BCEloss(yp.clone(), yt))
    except Exception as e:
        print("trainers/utils.py import failed:", repr(e))
    np.savez_compressed(os.path.join(OUT, "metrics_kat.npz"), **kat)
    print("wrote", sorted(os.listdir(OUT)))
    return 0


if __name__ == "__main__":
    sys.exit(main())
