"""Token-id boundary of the Whisper path.

The English-only openai-whisper tokenizer is GPT-2 BPE plus special tokens [UPSTREAM-RECALL].  The
BPE vocabulary is not available offline, so text<->ids goes through `GPT2BPE` only when
`vocab.json`/`merges.txt` exist under MODEL_CACHE_DIR; otherwise `IdTokenizer` provides a reversible
synthetic codec (documented deviation: transcripts are strings of token placeholders).  The
special-token ids and the suppress list do not depend on the vocabulary file.
"""
from __future__ import annotations

import json
import os
import re
import zlib
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

# English-only specials (openai-whisper tokenizer.py): eot 50256, sot 50257, 99 language tags,
# translate 50357, transcribe 50358, startoflm 50359, startofprev 50360, nospeech 50361,
# notimestamps 50362, <|0.00|> 50363 ... <|30.00|> 51863
EOT = 50256
SOT = 50257
TRANSLATE = 50357
TRANSCRIBE = 50358
SOT_LM = 50359
SOT_PREV = 50360
NO_SPEECH = 50361
NO_TIMESTAMPS = 50362
TIMESTAMP_BEGIN = 50363
BLANK = 220  # GPT-2 id of " "

# Tokenizer.non_speech_tokens for the GPT-2 vocabulary.  The id list is the one the locally
# installed `transformers` package carries for the english-only checkpoints
# (transformers/models/whisper/configuration_whisper.py NON_SPEECH_TOKENS, entries < 50257).
NON_SPEECH_TEXT_TOKENS = [
    1, 2, 7, 8, 9, 10, 14, 25, 26, 27, 28, 29, 31, 58, 59, 60, 61, 62, 63, 90, 91, 92, 93, 357, 366, 438,
    532, 685, 705, 796, 930, 1058, 1220, 1267, 1279, 1303, 1343, 1377, 1391, 1635, 1782, 1875, 2162,
    2361, 2488, 3467, 4008, 4211, 4600, 4808, 5299, 5855, 6329, 7203, 9609, 9959, 10563, 10786, 11420,
    11709, 11907, 13163, 13697, 13700, 14808, 15306, 16410, 16791, 17992, 19203, 19510, 20724, 22305,
    22935, 27007, 30109, 30420, 33409, 34949, 40283, 40493, 40549, 47282, 49146,
]
# decoding.py::_get_suppress_tokens adds these specials to the "-1" default list
SUPPRESS_TOKENS = sorted(NON_SPEECH_TEXT_TOKENS + [TRANSCRIBE, TRANSLATE, SOT, SOT_PREV, SOT_LM, NO_SPEECH])


@dataclass
class DecodeRules:
    eot: int = EOT
    sot: int = SOT
    sot_prev: int = SOT_PREV
    no_speech: int = NO_SPEECH
    no_timestamps: int = NO_TIMESTAMPS
    timestamp_begin: int = TIMESTAMP_BEGIN
    blank: int = BLANK
    max_initial_timestamp_index: int = 50  # 1.0 s / 0.02 s
    suppress: Sequence[int] = field(default_factory=lambda: list(SUPPRESS_TOKENS))


class IdTokenizer:
    """Reversible placeholder codec used when no BPE vocabulary is on disk.

    decode: text ids -> " <id>" words.  encode: " <id>" words map back to the id; any other word
    maps to a stable pseudo-id (crc32 of the lower-cased word, folded below the special range and
    away from the suppressed ids) so fixed prompts are deterministic."""
    name = "id-placeholder"
    _word = re.compile(r"<(\d+)>")

    def decode(self, ids: Sequence[int]) -> str:
        return "".join(f" <{int(t)}>" for t in ids if int(t) < EOT)

    def encode(self, text: str) -> List[int]:
        out = []
        for w in text.split():
            m = self._word.fullmatch(w)
            if m:
                out.append(int(m.group(1)))
            else:
                h = zlib.crc32(w.lower().encode("utf-8")) % 40000 + 1000
                out.append(h)
        return out


class GPT2BPE:
    """Byte-level BPE from vocab.json + merges.txt (GPT-2 files) when present under the cache dir."""
    name = "gpt2-bpe"

    def __init__(self, vocab_path: str, merges_path: str):
        with open(vocab_path, encoding="utf-8") as f:
            self.enc = json.load(f)
        self.dec = {v: k for k, v in self.enc.items()}
        with open(merges_path, encoding="utf-8") as f:
            lines = [l for l in f.read().split("\n") if l and not l.startswith("#version")]
        self.ranks = {tuple(l.split()): i for i, l in enumerate(lines)}
        bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
        cs = bs[:]
        n = 0
        for b in range(256):
            if b not in bs:
                bs.append(b)
                cs.append(256 + n)
                n += 1
        self.b2u = dict(zip(bs, map(chr, cs)))
        self.u2b = {v: k for k, v in self.b2u.items()}
        self.pat = _gpt2_pattern()

    def _bpe(self, token: str) -> List[str]:
        word = list(token)
        while len(word) > 1:
            pairs = [(self.ranks.get((a, b), 1 << 30), i) for i, (a, b) in enumerate(zip(word, word[1:]))]
            r, i = min(pairs)
            if r == 1 << 30:
                break
            word[i:i + 2] = [word[i] + word[i + 1]]
        return word

    def encode(self, text: str) -> List[int]:
        ids = []
        for tok in self.pat.findall(text):
            t = "".join(self.b2u[b] for b in tok.encode("utf-8"))
            ids.extend(self.enc[p] for p in self._bpe(t))
        return ids

    def decode(self, ids: Sequence[int]) -> str:
        s = "".join(self.dec[int(t)] for t in ids if int(t) < EOT)
        return bytearray(self.u2b[c] for c in s).decode("utf-8", errors="replace")


def _gpt2_pattern():
    """GPT-2 pre-tokeniser (whisper/tokenizer.py pat_str).  Needs \\p classes: the `regex` module when importable,
    else the ASCII approximation (identical on ASCII text)."""
    try:
        import regex
        return regex.compile(r"""'s|'t|'re|'ve|'m|'ll|'d| ?\p{L}+| ?\p{N}+| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+""")
    except ImportError:
        return re.compile(r"'s|'t|'re|'ve|'m|'ll|'d| ?[A-Za-z]+| ?\d+| ?[^\sA-Za-z\d]+|\s+(?!\S)|\s+")


class TiktokenBPE:
    """Byte-level BPE from a `.tiktoken` rank file -- the format openai-whisper ships its vocabulary in
    (`whisper/assets/gpt2.tiktoken`: one `base64(token bytes) rank` pair per line).  Encoding follows tiktoken's
    byte-pair merge: within a pre-token, repeatedly merge the adjacent pair whose concatenation has the lowest rank."""
    name = "tiktoken-bpe"

    def __init__(self, path: str):
        import base64
        self.ranks: dict = {}
        with open(path, "rb") as f:
            for line in f.read().splitlines():
                if line.strip():
                    tok, rank = line.split()
                    self.ranks[base64.b64decode(tok)] = int(rank)
        self.dec = {v: k for k, v in self.ranks.items()}
        self.pat = _gpt2_pattern()

    def _bpe(self, piece: bytes) -> List[int]:
        parts = [piece[i:i + 1] for i in range(len(piece))]
        while len(parts) > 1:
            best, at = None, -1
            for i in range(len(parts) - 1):
                r = self.ranks.get(parts[i] + parts[i + 1])
                if r is not None and (best is None or r < best):
                    best, at = r, i
            if best is None:
                break
            parts[at:at + 2] = [parts[at] + parts[at + 1]]
        return [self.ranks[p] for p in parts]

    def encode(self, text: str) -> List[int]:
        ids: List[int] = []
        for tok in self.pat.findall(text):
            ids.extend(self._bpe(tok.encode("utf-8")))
        return ids

    def decode(self, ids: Sequence[int]) -> str:
        return b"".join(self.dec[int(t)] for t in ids if int(t) < EOT).decode("utf-8", errors="replace")


def get_tokenizer(cache_dir: Optional[str] = None):
    """The vocabulary files the reference's whisper package would use, if a copy sits under MODEL_CACHE_DIR:
    `gpt2.tiktoken` (openai-whisper's own asset) or GPT-2's `vocab.json` + `merges.txt`; else the placeholder codec."""
    cache_dir = cache_dir or os.environ.get("MODEL_CACHE_DIR", "models")
    for sub in ("whisper", os.path.join("whisper", "assets"), "gpt2", ""):
        t = os.path.join(cache_dir, sub, "gpt2.tiktoken")
        if os.path.exists(t):
            return TiktokenBPE(t)
        v, m = os.path.join(cache_dir, sub, "vocab.json"), os.path.join(cache_dir, sub, "merges.txt")
        if os.path.exists(v) and os.path.exists(m):
            return GPT2BPE(v, m)
    return IdTokenizer()
