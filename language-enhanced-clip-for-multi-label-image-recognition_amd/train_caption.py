"""Entry point with the reference's flag and cfg-key names (reference project/my_code/train_caption.py:145-250).

    python -m leclip_amd.train_caption --eval-only --trainer Caption_distill_double --backbone ViT-B/16 \
        MODEL.BACKBONE.PATH /path/to/ViT-B-16.pt [--model-dir DIR --load-epoch E] [KEY VALUE ...]

Config assembly order is the reference's: defaults -> dataset yaml -> trainer yaml -> argparse -> free ``opts``, then
freeze (``setup_cfg``, :145-166).  One process drives one GPU; under ``torchrun`` (WORLD_SIZE > 1) the process group is
RCCL and evaluation is sharded inside the trainer's ``test()`` with one all-gather of the epoch's scores (``leclip_amd.parallel``).  The data pipeline of the
reference (Dassl dataset readers) is outside the hot path: the evaluation runs on the deterministic synthetic image set,
with targets drawn from the scores of the FIXED zero-shot prompts (not from the scores under evaluation) and the result
tagged as synthetic; ``--root`` / ``DATASET.ROOT`` is refused rather than ignored.
Without ``--eval-only`` the prompts are tuned first, the way the reference does it - on CAPTIONS fed through the text
encoder in place of images (Caption_distill_double.py:338-352, 789-897): here the caption set is synthetic too, one
templated sentence per (class, template) with that class as its label; ``OPTIM.MAX_EPOCH`` passes, the learning-rate
schedule stepped per epoch, the prompt learner saved under ``--output-dir`` in the reference's checkpoint layout.
"""
from __future__ import annotations

import argparse

import numpy as np
import torch

from . import parallel, synth
from .config import get_cfg_default
from .registry import build_evaluator, build_trainer


def reset_cfg(cfg, args):
    if args.root:
        cfg.DATASET.ROOT = args.root
    if args.output_dir:
        cfg.OUTPUT_DIR = args.output_dir
    if args.resume:
        cfg.RESUME = args.resume
    if args.seed:
        cfg.SEED = args.seed
    if args.trainer:
        cfg.TRAINER.NAME = args.trainer
    if args.backbone:
        cfg.MODEL.BACKBONE.NAME = args.backbone


def setup_cfg(args):
    cfg = get_cfg_default()
    if args.dataset_config_file:
        cfg.merge_from_file(args.dataset_config_file)
    if args.config_file:
        cfg.merge_from_file(args.config_file)
    reset_cfg(cfg, args)
    cfg.merge_from_list(args.opts)
    cfg.freeze()
    return cfg


class _SyntheticLoader:
    """Deterministic N(0,1) image batches (post-Normalize statistics); labels are filled in by the caller."""

    def __init__(self, n, batch, resolution, seed=1234):
        self.n, self.batch, self.resolution, self.seed = n, batch, resolution, seed
        self.labels = None

    def __iter__(self):
        for s in range(0, self.n, self.batch):
            b = min(self.batch, self.n - s)
            img = torch.from_numpy(synth.make_images(b, self.resolution, seed=self.seed, start=s))
            lab = torch.zeros(b, 80, dtype=torch.int64) if self.labels is None else torch.from_numpy(self.labels[s:s + b])
            yield {"img": img, "label": lab, "impath": [f"synthetic/{i}" for i in range(s, s + b)]}


_TEMPLATES = ("a photo of a {}.", "there is a {} in the scene.", "a close-up photo of a {}.", "a picture showing a {}.")


def synthetic_captions(classnames):
    """(tokens [n, 77], one-hot labels [n, C]): one sentence per (class, template).  Without the BPE merge table (offline
    image: only the prompts of clip/prompt_cache.json can be tokenised) the set shrinks to the cached template."""
    from .clip import tokenize
    names = [n.replace("_", " ") for n in classnames]
    templates = _TEMPLATES
    try:
        caps = tokenize([t.format(n) for t in templates for n in names])
    except FileNotFoundError:
        templates = _TEMPLATES[:1]
        caps = tokenize([t.format(n) for t in templates for n in names])
    return caps, torch.eye(len(names)).repeat(len(templates), 1)


class SyntheticCaptionLoader:
    """Training loader for the texts-as-images feed: yields {"img": tokens [b,77], "label": [b,C]} batches of this RANK's shard.
    Sharding follows the reference's distributed sampler (dassl/data/samplers.py:181-195): one permutation of the whole set
    per epoch, seeded by (seed, epoch) so that every rank draws the same permutation, padded to a multiple of the world
    size, rank r takes the contiguous slice r; DATALOADER.TRAIN_X.BATCH_SIZE is the per-rank batch."""

    def __init__(self, caps, labels, batch_size, seed=0, rank=0, world=1):
        self.caps, self.labels, self.batch_size, self.seed, self.rank, self.world = caps, labels, int(batch_size), int(seed), rank, world
        self.epoch = 0

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def indices(self):
        n = self.caps.shape[0]
        g = torch.Generator().manual_seed(self.seed * 100003 + self.epoch)
        order = torch.randperm(n, generator=g)
        per = (n + self.world - 1) // self.world
        order = torch.cat([order, order[:per * self.world - n]])
        return order[self.rank * per:(self.rank + 1) * per]

    def __iter__(self):
        idx = self.indices()
        for s in range(0, idx.numel(), self.batch_size):
            sel = idx[s:s + self.batch_size]
            yield {"img": self.caps[sel], "label": self.labels[sel]}


def train_on_synthetic_captions(cfg, trainer, output_dir: str = ""):
    """Prompt tuning on captions-as-images through the trainer's own epoch loop (reference: dassl's TrainerBase.train driving
    forward_backward, CDD.py:789-897): one templated sentence per (class, template), label = that class; rank-sharded
    batches, gradient all-reduce inside forward_backward, per-epoch LR schedule, rank-0 checkpoints from after_epoch."""
    caps, labels = synthetic_captions(trainer.classnames)
    trainer.train_loader_x = SyntheticCaptionLoader(caps, labels, cfg.DATALOADER.TRAIN_X.BATCH_SIZE, max(cfg.SEED, 0), trainer.rank, trainer.world)
    if output_dir:
        trainer.output_dir = output_dir
    return trainer.train()


def zero_shot_teacher_labels(trainer, images_fn, n, batch, pos_frac=0.1):
    """Targets for the synthetic evaluation set: drawn (synth.make_labels_from_logits) from the scores of the FIXED
    "a photo of a {class}." prompts - the zero-shot CLIP classifier of the same backbone - not from the scores under
    evaluation.  The printed mAP therefore measures how well the tuned prompts reproduce the zero-shot ranking; it is a
    synthetic-data figure and is tagged as such."""
    from .clip import tokenize
    model = getattr(trainer, f"model_{trainer.get_model_names()[0]}")
    names = [c.replace("_", " ") for c in trainer.classnames]
    toks = tokenize([f"a photo of a {c}." for c in names]).to(trainer.device)
    out = []
    with torch.no_grad():
        txt = model.model.encode_text(toks)
        for s in range(0, n, batch):
            b = min(batch, n - s)
            out.append(model.image_encoder.score(images_fn(s, b).to(trainer.device), txt, 4.0).float().cpu())
    return synth.make_labels_from_logits(torch.cat(out).numpy(), pos_frac=pos_frac)


def main(argv=None):
    import leclip_amd
    leclip_amd.configure()      # hardware queues for the image engine's stream parts: before the first device call (see configure())
    ap = argparse.ArgumentParser()
    ap.add_argument("--root", type=str, default="", help="path to dataset")
    ap.add_argument("--output-dir", type=str, default="", help="output directory")
    ap.add_argument("--resume", type=str, default="")
    ap.add_argument("--seed", type=int, default=-1)
    ap.add_argument("--trainer", type=str, default="", help="name of trainer")
    ap.add_argument("--backbone", type=str, default="", help="name of CNN backbone")
    ap.add_argument("--config-file", type=str, default="")
    ap.add_argument("--dataset-config-file", type=str, default="")
    ap.add_argument("--eval-only", action="store_true")
    ap.add_argument("--model-dir", type=str, default="")
    ap.add_argument("--load-epoch", type=int)
    ap.add_argument("--no-train", action="store_true")
    ap.add_argument("--num-images", type=int, default=512, help="synthetic evaluation set size")
    ap.add_argument("opts", default=None, nargs=argparse.REMAINDER)
    args = ap.parse_args(argv)

    cfg = setup_cfg(args)
    if cfg.DATASET.ROOT:
        # the reference's Dassl dataset readers (COCO / VOC / NUS-WIDE folders, caption json) are outside this build's scope:
        # refuse instead of silently scoring synthetic images under a dataset's name
        raise NotImplementedError(f"DATASET.ROOT={cfg.DATASET.ROOT}: no dataset reader in this build - the entry point runs on the "
                                  f"deterministic synthetic set only (drop --root)")
    if cfg.SEED >= 0:
        torch.manual_seed(cfg.SEED)
        np.random.seed(cfg.SEED)
    rank, world, _ = parallel.init_from_env()
    evaluator = build_evaluator(cfg)
    trainer = build_trainer(cfg, evaluator=evaluator)
    trainer.load_model(args.model_dir, epoch=args.load_epoch)
    if not args.eval_only and not args.no_train:
        train_on_synthetic_captions(cfg, trainer, args.output_dir)

    res = cfg.INPUT.SIZE[0]
    images_fn = lambda s, b: torch.from_numpy(synth.make_images(b, res, seed=1234, start=s))
    labels = zero_shot_teacher_labels(trainer, images_fn, args.num_images, cfg.DATALOADER.TEST.BATCH_SIZE)
    loader = _SyntheticLoader(args.num_images, cfg.DATALOADER.TEST.BATCH_SIZE, res)
    loader.labels = labels
    # the trainer's own test loop (reference :589-732): sharded by rank under WORLD_SIZE > 1 - every rank scores its share of each
    # batch, scores stay on the device, ONE all-gather per epoch, one asynchronous copy to pinned host memory (no per-batch .cpu())
    trainer.test_loader = loader
    trainer.test()
    out = None
    if rank == 0:
        print("=> synthetic evaluation set (N(0,1) images); targets = zero-shot fixed-prompt teacher, NOT a dataset result")
        out = evaluator.evaluate()
        out["data"] = "synthetic"
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    return out


if __name__ == "__main__":
    main()
