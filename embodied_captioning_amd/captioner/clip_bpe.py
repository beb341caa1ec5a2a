"""CLIP byte-pair decoder: token ids -> text, what ``open_clip.decode(...)`` does for the reference's CoCa wrapper
(``experimenting_env/captioner/models/coca/coca.py:30``: ``open_clip.decode(outputs["text"][0]).split("<end_of_text>")[0]
.replace("<start_of_text>", "")``) - without open_clip installed.

The algorithm is the published one of CLIP's ``SimpleTokenizer`` (open_clip ``tokenizer.py``; third-party, not in
/root/reference): the vocabulary is 256 byte symbols (GPT-2's printable-unicode stand-ins for the 256 byte values), the same 256
with the end-of-word marker ``</w>``, one entry per merge rule (the two parts concatenated), then ``<start_of_text>`` and
``<end_of_text>`` - 49 408 ids for the 48 894 merges of ``bpe_simple_vocab_16e6.txt.gz``.  Decoding concatenates the symbols of
the ids, maps every character back to its byte, decodes UTF-8 (errors="replace") and turns ``</w>`` into a space.

The vocabulary comes from files next to the checkpoint (nothing is downloaded):
  * ``vocab.json`` (HF CLIP layout: symbol -> id), or
  * ``bpe_simple_vocab_16e6.txt.gz`` / ``.txt`` / ``merges.txt`` (merge rules, one "a b" pair per line after a header line):
    the vocabulary is rebuilt from them exactly as SimpleTokenizer does.
Only decoding is on the captioner's path (the prompt is the single start token), so there is no encoder here.
"""
from __future__ import annotations

import gzip
import json
import os
from typing import Dict, Iterable, List, Optional, Sequence

SOT, EOT = "<start_of_text>", "<end_of_text>"
_MERGE_FILES = ("bpe_simple_vocab_16e6.txt.gz", "bpe_simple_vocab_16e6.txt", "merges.txt")


def bytes_to_unicode() -> Dict[int, str]:
    """GPT-2's reversible byte <-> printable-character table: the 188 printable latin-1 bytes stand for themselves, the
    other 68 are mapped to code points 256.."""
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return dict(zip(bs, (chr(c) for c in cs)))


def vocab_from_merges(merges: Iterable[Sequence[str]], specials: Sequence[str] = (SOT, EOT)) -> List[str]:
    """SimpleTokenizer's vocabulary order: byte symbols, byte symbols + </w>, one symbol per merge, the special tokens."""
    vocab = list(bytes_to_unicode().values())
    vocab = vocab + [v + "</w>" for v in vocab]
    for a, b in merges:
        vocab.append(a + b)
    vocab.extend(specials)
    return vocab


def read_merges(path: str, limit: Optional[int] = 49152 - 256 - 2) -> List[Sequence[str]]:
    """Merge rules of a CLIP BPE file: the first line is a header, then one "left right" pair per line; SimpleTokenizer keeps
    the first 48 894 (`limit`; None = all, for HF `merges.txt` files that hold exactly their model's rules)."""
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "rt", encoding="utf-8") as f:
        lines = f.read().split("\n")
    lines = lines[1:]
    if limit is not None:
        lines = lines[:limit]
    return [tuple(ln.split()) for ln in lines if len(ln.split()) == 2]


class ClipBpeDecoder:
    def __init__(self, id_to_symbol: Dict[int, str]):
        self.decoder = dict(id_to_symbol)
        self.byte_decoder = {c: b for b, c in bytes_to_unicode().items()}

    @staticmethod
    def from_vocab_json(path: str) -> "ClipBpeDecoder":
        with open(path, encoding="utf-8") as f:
            enc = json.load(f)
        return ClipBpeDecoder({int(i): s for s, i in enc.items()})

    @staticmethod
    def from_merges_file(path: str) -> "ClipBpeDecoder":
        hf = os.path.basename(path) == "merges.txt"
        vocab = vocab_from_merges(read_merges(path, None if hf else 49152 - 256 - 2),
                                  ("<|startoftext|>", "<|endoftext|>") if hf else (SOT, EOT))
        return ClipBpeDecoder(dict(enumerate(vocab)))

    @staticmethod
    def find(*places: Optional[str]) -> Optional["ClipBpeDecoder"]:
        """First usable vocabulary in the given directories / next to the given files (None when there is none)."""
        for p in places:
            if not p:
                continue
            d = p if os.path.isdir(p) else os.path.dirname(os.path.abspath(p))
            v = os.path.join(d, "vocab.json")
            if os.path.exists(v):
                return ClipBpeDecoder.from_vocab_json(v)
            for fn in _MERGE_FILES:
                m = os.path.join(d, fn)
                if os.path.exists(m):
                    return ClipBpeDecoder.from_merges_file(m)
        return None

    def decode(self, ids: Iterable[int]) -> str:
        """open_clip `SimpleTokenizer.decode`: symbols -> bytes -> UTF-8, `</w>` -> space (special tokens stay in the text,
        as there: the caller cuts at <end_of_text> and drops <start_of_text>, coca.py:30)."""
        text = "".join(self.decoder[int(i)] for i in ids)
        return bytearray(self.byte_decoder[c] for c in text).decode("utf-8", errors="replace").replace("</w>", " ")

    def caption(self, ids: Iterable[int]) -> str:
        """The reference wrapper's post-processing of a generated row (coca.py:30), for either spelling of the specials."""
        text = self.decode(ids)
        for eot, sot in ((EOT, SOT), ("<|endoftext|>", "<|startoftext|>")):
            text = text.split(eot)[0].replace(sot, "")
        return text
