"""Byte-level BPE tokenizer for CLIP text prompts (host plumbing; replaces `CLIP/clip/simple_tokenizer.py` +
`tokenize`, clip_explainability.py:237-273).

The merge table (`bpe_simple_vocab_16e6.txt.gz`, 1.3 MB, OpenAI CLIP) is third-party DATA that this repo does not
carry; it is looked up at run time (first hit wins):
    $SEMABS_BPE_VOCAB, <package>/assets/bpe_simple_vocab_16e6.txt.gz, ~/.cache/clip/bpe_simple_vocab_16e6.txt.gz.
Callers that already hold token ids (int64 [B, 77]) can pass them straight to `ClipWrapper.set_classes_tokens`.
"""
from __future__ import annotations

import gzip
import html
import os
from functools import lru_cache
from typing import List, Sequence

import regex
import torch

CONTEXT_LENGTH = 77
_CANDIDATES = (
    os.environ.get("SEMABS_BPE_VOCAB", ""),
    os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "assets", "bpe_simple_vocab_16e6.txt.gz"),
    os.path.expanduser("~/.cache/clip/bpe_simple_vocab_16e6.txt.gz"),
)


def find_vocab() -> str | None:
    for c in (os.environ.get("SEMABS_BPE_VOCAB", ""),) + _CANDIDATES[1:]:
        if c and os.path.isfile(c):
            return c
    return None


def _byte_alphabet():
    """The reversible byte -> printable-unicode map of GPT-2 style BPE."""
    keep = list(range(33, 127)) + list(range(161, 173)) + list(range(174, 256))
    codes = list(keep)
    extra = 0
    for b in range(256):
        if b not in keep:
            keep.append(b)
            codes.append(256 + extra)
            extra += 1
    return {b: chr(c) for b, c in zip(keep, codes)}


class BPETokenizer:
    def __init__(self, vocab_path: str | None = None):
        path = vocab_path or find_vocab()
        if path is None:
            raise FileNotFoundError(
                "CLIP BPE merge table not found; set SEMABS_BPE_VOCAB to bpe_simple_vocab_16e6.txt.gz "
                "(or pass token ids directly)")
        self.bytemap = _byte_alphabet()
        lines = gzip.open(path).read().decode("utf-8").split("\n")
        merges = [tuple(l.split()) for l in lines[1:49152 - 256 - 2 + 1]]
        symbols = list(self.bytemap.values())
        symbols = symbols + [s + "</w>" for s in symbols] + ["".join(m) for m in merges] + ["<|startoftext|>", "<|endoftext|>"]
        self.ids = {s: i for i, s in enumerate(symbols)}
        self.rank = {m: i for i, m in enumerate(merges)}
        self.sot, self.eot = self.ids["<|startoftext|>"], self.ids["<|endoftext|>"]
        self.splitter = regex.compile(
            r"""<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+""", regex.IGNORECASE)

    @lru_cache(maxsize=65536)
    def _merge_word(self, word: str) -> tuple:
        parts = list(word[:-1]) + [word[-1] + "</w>"]
        while len(parts) > 1:
            best, best_rank = None, None
            for i in range(len(parts) - 1):
                r = self.rank.get((parts[i], parts[i + 1]))
                if r is not None and (best_rank is None or r < best_rank):
                    best, best_rank = (parts[i], parts[i + 1]), r
            if best is None:
                break
            merged, i = [], 0
            while i < len(parts):
                if i < len(parts) - 1 and (parts[i], parts[i + 1]) == best:
                    merged.append(parts[i] + parts[i + 1])
                    i += 2
                else:
                    merged.append(parts[i])
                    i += 1
            parts = merged
        return tuple(parts)

    def encode(self, text: str) -> List[int]:
        text = html.unescape(html.unescape(text)).strip()
        text = regex.sub(r"\s+", " ", text).strip().lower()
        out: List[int] = []
        for tok in self.splitter.findall(text):
            word = "".join(self.bytemap[b] for b in tok.encode("utf-8"))
            out.extend(self.ids[p] for p in self._merge_word(word))
        return out

    def tokenize(self, texts: Sequence[str] | str, context_length: int = CONTEXT_LENGTH, truncate: bool = False) -> torch.Tensor:
        if isinstance(texts, str):
            texts = [texts]
        res = torch.zeros(len(texts), context_length, dtype=torch.long)
        for i, t in enumerate(texts):
            ids = [self.sot] + self.encode(t) + [self.eot]
            if len(ids) > context_length:
                if not truncate:
                    raise RuntimeError(f"Input {t} is too long for context length {context_length}")
                ids = ids[:context_length]
                ids[-1] = self.eot
            res[i, : len(ids)] = torch.tensor(ids)
        return res
