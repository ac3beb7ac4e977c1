#!/usr/bin/env python3
"""Ninth golden generator — text side of the input pipeline: the REFERENCE's SimpleTokenizer
(model/base/simple_tokenizer.py) and the body of BaseDataset._load_text (dataset/base.py:66-83, with the random caption
choice replaced by the caption itself) on the captions of tests/bpeutil.py, once with the miniature merges file written by
bpeutil.write_mini_merges (so the test runs anywhere) and once with the reference's own bpe_simple_vocab_16e6.txt.gz.
ftfy is not in this image: it is stubbed to the identity (make_golden.install_stubs), which is what it is on the native
captions; the NON_NATIVE ones are recorded with the stub as well and only their html-entity / unicode handling is compared."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import bpeutil as bu  # noqa: E402
from make_golden import install_stubs, ref_import, save  # noqa: E402


def load_text(tok, caption, max_words):
    words = tok.tokenize(caption)
    words = ["<|startoftext|>"] + words
    if len(words) > max_words - 1:
        words = words[:max_words - 1]
    words = words + ["<|endoftext|>"]
    ids = tok.convert_tokens_to_ids(words)
    while len(ids) < max_words:
        ids.append(0)
    return ids


def gen():
    st = ref_import("model.base.simple_tokenizer")
    bu.write_mini_merges()
    import gzip
    import tempfile
    out = {}
    with tempfile.TemporaryDirectory() as d:
        gz = os.path.join(d, "mini.txt.gz")
        with gzip.open(gz, "wb") as f:
            f.write(open(bu.MINI_MERGES, "rb").read())
        for tag, tok in (("mini", st.SimpleTokenizer(gz)), ("full", st.SimpleTokenizer())):
            out[f"{tag}_vocab"] = np.array(len(tok.encoder))
            for mw in (8, 32, 77):
                out[f"{tag}_ids_{mw}"] = np.array([load_text(tok, c, mw) for c in bu.CAPTIONS], dtype=np.int64)
            out[f"{tag}_nonnative_32"] = np.array([load_text(tok, c, 32) for c in bu.NON_NATIVE], dtype=np.int64)
            out[f"{tag}_encode"] = np.array([len(tok.encode(c)) for c in bu.CAPTIONS])
            out[f"{tag}_decode_3"] = np.array(tok.decode(tok.encode(bu.CAPTIONS[3])))
    save("bpe.npz", **out)


if __name__ == "__main__":
    install_stubs()
    gen()
