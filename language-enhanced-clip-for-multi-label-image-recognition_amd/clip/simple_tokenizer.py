"""Byte-level BPE tokenizer compatible with OpenAI CLIP's vocabulary (49 408 entries, SOT 49406, EOT 49407).

Behavioural mirror of the reference's ``clip/simple_tokenizer.py:62-132`` (lower-casing, html unescape,
whitespace collapse, the CLIP split pattern, greedy lowest-rank pair merging), written independently.
The merge table is *data* that ships with any CLIP checkpoint (``bpe_simple_vocab_16e6.txt.gz``); it is not
vendored here.  Resolution order: explicit ``bpe_path`` argument, ``$LECLIP_BPE_VOCAB``, a copy next to this
file.  When none exists, ``clip.tokenize`` falls back to the offline prompt cache (``prompt_cache.json``).
``ftfy`` is optional: without it ``fix_text`` is the identity, exact for ASCII prompts.
"""
from __future__ import annotations

import gzip
import html
import os
from functools import lru_cache
from typing import Dict, List, Tuple

try:  # optional dependency
    import ftfy as _ftfy

    def _fix_text(t: str) -> str:
        return _ftfy.fix_text(t)
except Exception:  # pragma: no cover - ftfy is absent in the build image
    def _fix_text(t: str) -> str:
        return t

import regex as re

_N_MERGES = 49152 - 256 - 2
_SPLIT = re.compile(r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+",
                    re.IGNORECASE)


def default_bpe() -> str:
    env = os.environ.get("LECLIP_BPE_VOCAB")
    if env:
        return env
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "bpe_simple_vocab_16e6.txt.gz")


@lru_cache()
def byte_alphabet() -> Dict[int, str]:
    """The GPT-2 byte <-> printable-unicode table: printable latin-1 bytes map to themselves, the remaining 68
    byte values to code points 256, 257, ... in increasing byte order."""
    keep = set(range(33, 127)) | set(range(161, 173)) | set(range(174, 256))
    table, extra = {}, 0
    for b in sorted(keep):
        table[b] = chr(b)
    for b in range(256):
        if b not in keep:
            table[b] = chr(256 + extra)
            extra += 1
    return table


def _symbol_order() -> List[str]:
    # vocabulary order is the reference's: kept bytes in (33..126, 161..172, 174..255) order, then the remapped ones
    kept = list(range(33, 127)) + list(range(161, 173)) + list(range(174, 256))
    rest = [b for b in range(256) if b not in set(kept)]
    alpha = byte_alphabet()
    return [alpha[b] for b in kept + rest]


class SimpleTokenizer:
    def __init__(self, bpe_path: str = None):
        bpe_path = bpe_path or default_bpe()
        if not os.path.exists(bpe_path):
            raise FileNotFoundError(f"BPE merge table not found at {bpe_path}; set LECLIP_BPE_VOCAB")
        with gzip.open(bpe_path, "rt", encoding="utf-8") as f:
            lines = f.read().split("\n")
        merges: List[Tuple[str, str]] = [tuple(ln.split()) for ln in lines[1:1 + _N_MERGES]]
        symbols = _symbol_order()
        vocab = symbols + [s + "</w>" for s in symbols] + ["".join(m) for m in merges]
        vocab += ["<|startoftext|>", "<|endoftext|>"]
        self.encoder = {tok: i for i, tok in enumerate(vocab)}
        self.decoder = {i: tok for tok, i in self.encoder.items()}
        self.rank = {m: i for i, m in enumerate(merges)}
        self.byte_encoder = byte_alphabet()
        self.byte_decoder = {c: b for b, c in self.byte_encoder.items()}
        self._cache: Dict[str, List[str]] = {}

    def _merge_word(self, token: str) -> List[str]:
        """Greedy BPE: repeatedly fuse every occurrence of the adjacent pair with the lowest merge rank."""
        if token in ("<|startoftext|>", "<|endoftext|>"):
            return [token]
        hit = self._cache.get(token)
        if hit is not None:
            return hit
        parts = list(token[:-1]) + [token[-1] + "</w>"]
        while len(parts) > 1:
            best, best_rank = None, None
            for pair in zip(parts, parts[1:]):
                r = self.rank.get(pair)
                if r is not None and (best_rank is None or r < best_rank):
                    best, best_rank = pair, r
            if best is None:
                break
            fused, i = [], 0
            while i < len(parts):
                if i + 1 < len(parts) and parts[i] == best[0] and parts[i + 1] == best[1]:
                    fused.append(parts[i] + parts[i + 1])
                    i += 2
                else:
                    fused.append(parts[i])
                    i += 1
            parts = fused
        self._cache[token] = parts
        return parts

    def encode(self, text: str) -> List[int]:
        text = html.unescape(html.unescape(_fix_text(text))).strip()
        text = re.sub(r"\s+", " ", text).strip().lower()
        ids: List[int] = []
        for word in _SPLIT.findall(text):
            mapped = "".join(self.byte_encoder[b] for b in word.encode("utf-8"))
            ids.extend(self.encoder[p] for p in self._merge_word(mapped))
        return ids

    def decode(self, tokens) -> str:
        text = "".join(self.decoder[int(t)] for t in tokens)
        raw = bytearray(self.byte_decoder[c] for c in text)
        return raw.decode("utf-8", errors="replace").replace("</w>", " ")
