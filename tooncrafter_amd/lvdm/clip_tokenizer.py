"""CLIP byte-pair-encoding tokeniser: the host-side step in front of `FrozenOpenCLIPEmbedder`.

The reference calls `open_clip.tokenize(text)` (lvdm/modules/encoders/condition.py:210; third-party package
open_clip_torch == 2.22.0, requirements.txt:22, absent from this image).  This is a restatement of that package's
published algorithm (open_clip/tokenizer.py: SimpleTokenizer + tokenize, itself the OpenAI CLIP tokeniser): reversible
byte -> unicode alphabet, whitespace / HTML clean-up, lower-casing, the `<start_of_text>` / `<end_of_text>` specials, a
regex pre-tokeniser and rank-ordered BPE merges, context length 77 with truncation that keeps the end token.

**Parity**: pinned (round 5) to an independent implementation of the same tokeniser -- transformers.CLIPTokenizer, the Rust
`tokenizers` BPE -- on a merge table both are given (tests/golden/make_openclip_golden.py -> clip_bpe_hf.json;
tests/test_openclip_golden_cpu.py: case folding, whitespace, contractions, digits, punctuation runs, non-ASCII letters,
truncation).  NOT pinned: open_clip's real vocabulary file (`bpe_simple_vocab_16e6.txt.gz`, 1.3 MB, shipped inside open_clip) --
neither the package nor the file exists here.  The merges are data, not code: pass the path of that file (`CLIPTokenizer(path)`,
or `TC_CLIP_BPE_VOCAB`); without it only the empty prompt -- the interpolation scripts' default -- can be tokenised (its tokens
do not depend on the vocabulary).  open_clip also runs `ftfy.fix_text` on the input when ftfy is installed; ftfy is absent here
and plain ASCII prompts are unaffected by it.
"""
from __future__ import annotations

import gzip
import html
import os
from functools import lru_cache
from typing import Iterable, List, Union

import torch

SOT, EOT = "<start_of_text>", "<end_of_text>"
PATTERN = r"""<start_of_text>|<end_of_text>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+"""


@lru_cache()
def bytes_to_unicode():
    """The reversible byte <-> printable-unicode table of GPT-2 / CLIP: printable latin-1 bytes map to themselves, the
    other 68 byte values to code points from 256 upwards."""
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return dict(zip(bs, (chr(c) for c in cs)))


def get_pairs(word):
    return {(a, b) for a, b in zip(word[:-1], word[1:])}


def basic_clean(text: str) -> str:
    return html.unescape(html.unescape(text)).strip()


def whitespace_clean(text: str) -> str:
    return " ".join(text.split()).strip()


class CLIPTokenizer:
    """`merges`: the BPE merge rules in rank order, as an iterable of "a b" strings, or the path of open_clip's
    `bpe_simple_vocab_16e6.txt.gz` (first line is a header; the 48894 rules that follow are used: 49152 - 256 - 2)."""

    def __init__(self, merges: Union[str, Iterable[str]], context_length: int = 77):
        import regex
        if isinstance(merges, str):
            opener = gzip.open if merges.endswith(".gz") else open
            with opener(merges, "rt", encoding="utf-8") as f:
                lines = f.read().split("\n")
            merges = lines[1:49152 - 256 - 2 + 1]
        rules = [tuple(m.split()) for m in merges if m.strip()]
        self.byte_encoder = bytes_to_unicode()
        vocab = list(self.byte_encoder.values())
        vocab = vocab + [v + "</w>" for v in vocab]
        vocab += ["".join(r) for r in rules]
        vocab += [SOT, EOT]
        self.encoder = {tok: i for i, tok in enumerate(vocab)}
        self.bpe_ranks = {r: i for i, r in enumerate(rules)}
        self.cache = {SOT: SOT, EOT: EOT}
        self.pat = regex.compile(PATTERN, regex.IGNORECASE)
        self.context_length = context_length
        self.sot_token, self.eot_token = self.encoder[SOT], self.encoder[EOT]

    def bpe(self, token: str) -> str:
        if token in self.cache:
            return self.cache[token]
        word = tuple(token[:-1]) + (token[-1] + "</w>",)
        pairs = get_pairs(word)
        if not pairs:
            return token + "</w>"
        while True:
            bigram = min(pairs, key=lambda p: self.bpe_ranks.get(p, float("inf")))
            if bigram not in self.bpe_ranks:
                break
            first, second = bigram
            new, i = [], 0
            while i < len(word):
                try:
                    j = word.index(first, i)
                except ValueError:
                    new.extend(word[i:])
                    break
                new.extend(word[i:j])
                i = j
                if i < len(word) - 1 and word[i] == first and word[i + 1] == second:
                    new.append(first + second)
                    i += 2
                else:
                    new.append(word[i])
                    i += 1
            word = tuple(new)
            if len(word) == 1:
                break
            pairs = get_pairs(word)
        out = " ".join(word)
        self.cache[token] = out
        return out

    def encode(self, text: str) -> List[int]:
        ids: List[int] = []
        text = whitespace_clean(basic_clean(text)).lower()
        for tok in self.pat.findall(text):
            tok = "".join(self.byte_encoder[b] for b in tok.encode("utf-8"))
            ids.extend(self.encoder[t] for t in self.bpe(tok).split(" "))
        return ids

    def __call__(self, texts: Union[str, List[str]]) -> torch.Tensor:
        """-> int64 (B, context_length): <start_of_text> tokens <end_of_text>, zero padded; too long inputs are cut and
        the last position set to <end_of_text> (open_clip/tokenizer.py: tokenize)."""
        texts = [texts] if isinstance(texts, str) else list(texts)
        out = torch.zeros((len(texts), self.context_length), dtype=torch.long)
        for i, t in enumerate(texts):
            ids = [self.sot_token] + self.encode(t) + [self.eot_token]
            if len(ids) > self.context_length:
                ids = ids[:self.context_length]
                ids[-1] = self.eot_token
            out[i, :len(ids)] = torch.tensor(ids)
        return out


def default_vocab_path():
    """TC_CLIP_BPE_VOCAB, or the file inside an installed open_clip package; None when neither exists."""
    p = os.environ.get("TC_CLIP_BPE_VOCAB")
    if p and os.path.exists(p):
        return p
    try:
        import open_clip
        p = os.path.join(os.path.dirname(open_clip.__file__), "bpe_simple_vocab_16e6.txt.gz")
        return p if os.path.exists(p) else None
    except Exception:
        return None
