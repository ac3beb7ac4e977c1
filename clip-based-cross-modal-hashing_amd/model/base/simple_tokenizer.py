"""CLIP's byte-level BPE tokenizer with the reference's interface (model/base/simple_tokenizer.py:62-148: `encoder`,
`decoder`, `bpe`, `encode`, `decode`, `tokenize`, `convert_tokens_to_ids`) plus the batch entry the input pipeline uses:

    SimpleTokenizer().encode_captions(list_of_str, max_words) -> int64 [n, max_words]

= dataset/base.py:66-83 (_load_text) for a whole batch in one native, multi-threaded call (libcmh `cmh_bpe_encode_captions`).
Captions the native path does not take (anything outside printable ASCII, or '&': upstream sends every caption through
ftfy.fix_text and html.unescape, which only change such text) are tokenised by the Python path below, which needs `ftfy`
exactly like upstream.

The merges file is data, not code: `bpe_simple_vocab_16e6.txt.gz` (OpenAI CLIP) is looked up next to this module, as
upstream does, or at $CMH_BPE_VOCAB."""
import ctypes as C
import gzip
import html
import os
from functools import lru_cache

import numpy as np
import regex as re
import torch

import cmh_native as N

SOT, EOT = "<|startoftext|>", "<|endoftext|>"


@lru_cache()
def default_bpe():
    return os.environ.get("CMH_BPE_VOCAB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "bpe_simple_vocab_16e6.txt.gz")


@lru_cache()
def bytes_to_unicode():
    """byte -> printable unicode character (simple_tokenizer.py:15-35): '!'..'~', '¡'..'¬', '®'..'ÿ' map to themselves, the
    remaining 68 bytes to U+0100 onwards in byte order."""
    keep = list(range(33, 127)) + list(range(161, 173)) + list(range(174, 256))
    rest = [b for b in range(256) if b not in keep]
    table = {b: chr(b) for b in keep}
    table.update({b: chr(256 + i) for i, b in enumerate(rest)})
    return {b: table[b] for b in keep + rest}            # upstream's insertion order: it is the order of the vocabulary


def basic_clean(text):
    if text.isascii() and "&" not in text and not any(ord(c) < 32 and c not in "\t\n\r" for c in text) and "\x7f" not in text:
        return text.strip()                                # ftfy.fix_text / html.unescape leave such text alone
    import ftfy                                            # upstream imports it unconditionally
    return html.unescape(html.unescape(ftfy.fix_text(text))).strip()


def whitespace_clean(text):
    return re.sub(r"\s+", " ", text).strip()


class SimpleTokenizer(object):
    def __init__(self, bpe_path: str = None, threads: int = 0):
        bpe_path = bpe_path or default_bpe()
        if not os.path.exists(bpe_path):
            raise FileNotFoundError(
                f"{bpe_path}: the CLIP BPE merges file is missing — copy bpe_simple_vocab_16e6.txt.gz from the upstream repository "
                "(model/base/) next to this module or point CMH_BPE_VOCAB at it")
        raw = gzip.open(bpe_path).read() if bpe_path.endswith(".gz") else open(bpe_path, "rb").read()
        self.byte_encoder = bytes_to_unicode()
        self.byte_decoder = {v: k for k, v in self.byte_encoder.items()}
        merges = [tuple(line.split()) for line in raw.decode("utf-8").split("\n")[1:49152 - 256 - 2 + 1]]
        vocab = list(self.byte_encoder.values())
        vocab = vocab + [v + "</w>" for v in vocab] + ["".join(m) for m in merges] + [SOT, EOT]
        self.encoder = dict(zip(vocab, range(len(vocab))))
        self.decoder = {v: k for k, v in self.encoder.items()}
        self.bpe_ranks = dict(zip(merges, range(len(merges))))
        self.cache = {SOT: SOT, EOT: EOT}
        self.pat = re.compile(r"""<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+""",
                              re.IGNORECASE)
        self.threads = threads
        handle = C.c_void_p()
        N.check(N.lib().cmh_bpe_create(raw, len(raw), C.byref(handle)), "cmh_bpe_create")
        self._native = handle
        if N.lib().cmh_bpe_vocab_size(handle) != len(vocab):
            raise N.NativeError("cmh_bpe_create: vocabulary size differs from the Python tables")

    def __del__(self):
        h, self._native = getattr(self, "_native", None), None
        if h:
            try:
                N.lib().cmh_bpe_destroy(h)
            except Exception:
                pass

    # ---- Python path (also the only one that returns token STRINGS) ---------------------------------------------------
    def bpe(self, token):
        """-> the token's BPE symbols joined by blanks (simple_tokenizer.py:81-120)."""
        hit = self.cache.get(token)
        if hit is not None:
            return hit
        word = list(token[:-1]) + [token[-1] + "</w>"]
        ranks = self.bpe_ranks
        while len(word) > 1:
            best = min(zip(word, word[1:]), key=lambda p: ranks.get(p, float("inf")))
            if best not in ranks:
                break
            merged, i = [], 0
            while i < len(word):
                if i + 1 < len(word) and (word[i], word[i + 1]) == best:
                    merged.append(word[i] + word[i + 1])
                    i += 2
                else:
                    merged.append(word[i])
                    i += 1
            word = merged
        out = " ".join(word)
        self.cache[token] = out
        return out

    def tokenize(self, text):
        text = whitespace_clean(basic_clean(text)).lower()
        out = []
        for piece in re.findall(self.pat, text):
            out.extend(self.bpe("".join(self.byte_encoder[b] for b in piece.encode("utf-8"))).split(" "))
        return out

    def convert_tokens_to_ids(self, tokens):
        return [self.encoder[t] for t in tokens]

    def encode(self, text):
        return self.convert_tokens_to_ids(self.tokenize(text))

    def decode(self, tokens):
        text = "".join(self.decoder[int(t)] for t in tokens)
        return bytearray(self.byte_decoder[c] for c in text).decode("utf-8", errors="replace").replace("</w>", " ")

    def caption_ids(self, caption, max_words):
        """One caption through the Python path: dataset/base.py:66-83 without the random choice."""
        words = ([SOT] + self.tokenize(caption))[:max_words - 1] + [EOT]
        ids = self.convert_tokens_to_ids(words)
        return ids + [0] * (max_words - len(ids))

    # ---- native batch path ------------------------------------------------------------------------------------------------
    def encode_captions(self, captions, max_words=32, return_native_mask=False):
        """list of str -> int64 [n, max_words] (CPU tensor): [SOT] + BPE ids cut to max_words - 1, [EOT], zero padding."""
        enc = [str(c).encode("utf-8") for c in captions]
        n = len(enc)
        out = torch.zeros(n, max_words, dtype=torch.int64)
        if n == 0:
            return (out, np.zeros(0, dtype=bool)) if return_native_mask else out
        offsets = np.zeros(n + 1, dtype=np.int64)
        np.cumsum([len(e) for e in enc], out=offsets[1:])
        blob = b"".join(enc)
        status = np.zeros(n, dtype=np.uint8)
        N.check(N.lib().cmh_bpe_encode_captions(self._native, blob, C.c_void_p(offsets.ctypes.data), n, int(max_words),
                                                C.c_void_p(out.data_ptr()), C.c_void_p(status.ctypes.data), int(self.threads)),
                "cmh_bpe_encode_captions")
        for i in np.nonzero(status)[0]:
            out[i] = torch.tensor(self.caption_ids(captions[i], max_words), dtype=torch.int64)
        return (out, status == 0) if return_native_mask else out
