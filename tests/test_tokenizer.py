"""Input pipeline, text side: the native batch tokenizer (cmh_bpe_* through the C ABI — host code, runs without a GPU) and the
Python path of model/base/simple_tokenizer.py against ids the REFERENCE's SimpleTokenizer + _load_text produced
(tests/golden/make_golden9.py), with a miniature merges file everywhere and with the real CLIP vocabulary where it is found."""
import gzip
import os
import sys
import types

import numpy as np
import pytest
import torch

import bpeutil as bu

REAL_VOCAB = [p for p in (os.environ.get("CMH_BPE_VOCAB"), "/root/reference/model/base/bpe_simple_vocab_16e6.txt.gz") if p and os.path.exists(p)]


@pytest.fixture(scope="module")
def mini(tmp_path_factory):
    from model.base.simple_tokenizer import SimpleTokenizer
    gz = tmp_path_factory.mktemp("bpe") / "mini.txt.gz"
    with gzip.open(gz, "wb") as f:
        f.write(open(bu.MINI_MERGES, "rb").read())
    return SimpleTokenizer(str(gz))


def _check(tok, g, tag):
    assert len(tok.encoder) == int(g[f"{tag}_vocab"])
    for mw in (8, 32, 77):
        ids, native = tok.encode_captions(bu.CAPTIONS, mw, return_native_mask=True)
        assert native.all()                                                   # every caption of the list takes the native path
        assert ids.dtype == torch.int64 and np.array_equal(ids.numpy(), g[f"{tag}_ids_{mw}"])
        py = np.array([tok.caption_ids(c, mw) for c in bu.CAPTIONS])           # the Python path agrees too
        assert np.array_equal(py, g[f"{tag}_ids_{mw}"])
    assert [len(tok.encode(c)) for c in bu.CAPTIONS] == list(g[f"{tag}_encode"])
    assert tok.decode(tok.encode(bu.CAPTIONS[3])) == str(g[f"{tag}_decode_3"])


def test_mini_vocabulary_matches_reference(golden, mini):
    _check(mini, golden("bpe.npz"), "mini")


@pytest.mark.skipif(not REAL_VOCAB, reason="bpe_simple_vocab_16e6.txt.gz not available (set CMH_BPE_VOCAB)")
def test_real_vocabulary_matches_reference(golden):
    from model.base.simple_tokenizer import SimpleTokenizer
    tok = SimpleTokenizer(REAL_VOCAB[0])
    assert tok.encoder["<|startoftext|>"] == 49406 and tok.encoder["<|endoftext|>"] == 49407
    _check(tok, golden("bpe.npz"), "full")


def test_non_native_captions_take_the_python_path(golden, mini, monkeypatch):
    """Non-ASCII text and '&' are flagged by the native call; the Python path (ftfy stubbed to the identity, as in the golden
    generator) reproduces the reference's ids: html entities, unicode letter classes, byte-level fallback symbols."""
    ftfy = types.ModuleType("ftfy")
    ftfy.fix_text = lambda s: s
    monkeypatch.setitem(sys.modules, "ftfy", ftfy)
    ids, native = mini.encode_captions(bu.NON_NATIVE + bu.CAPTIONS[:2], 32, return_native_mask=True)
    assert list(native) == [False] * len(bu.NON_NATIVE) + [True, True]
    g = golden("bpe.npz")
    assert np.array_equal(ids.numpy()[:len(bu.NON_NATIVE)], g["mini_nonnative_32"])
    assert np.array_equal(ids.numpy()[len(bu.NON_NATIVE):], g["mini_ids_32"][:2])


def test_threads_and_cache_do_not_change_results(mini):
    rng = np.random.default_rng(3)
    words = bu.CORPUS.split() + ["it's", "we'll", "1999", "!!", "<|endoftext|>", "Zebra", "...", "x-ray"]
    caps = [" ".join(rng.choice(words, size=int(rng.integers(0, 40)))) for _ in range(3000)]
    from model.base.simple_tokenizer import SimpleTokenizer
    a = mini.encode_captions(caps, 32)
    mini.threads = 1
    b = mini.encode_captions(caps, 32)
    mini.threads = 0
    assert torch.equal(a, b)
    for i in range(0, 3000, 97):
        assert a[i].tolist() == mini.caption_ids(caps[i], 32)
    assert (a[:, 0] == mini.encoder["<|startoftext|>"]).all()
    eot = mini.encoder["<|endoftext|>"]
    assert ((a == eot).sum(1) >= 1).all()          # every row closes with an EOT (a caption may also contain the literal token)
    assert mini.encode_captions([], 32).shape == (0, 32)


def test_missing_vocabulary_fails_loudly(tmp_path):
    from model.base.simple_tokenizer import SimpleTokenizer
    with pytest.raises(FileNotFoundError):
        SimpleTokenizer(str(tmp_path / "nope.txt.gz"))


def test_random_ascii_agrees_with_the_python_path(mini):
    """Fuzz: random printable-ASCII strings (heavy on apostrophes, angle brackets, digits, blanks) through the native splitter
    and BPE against the regex-based Python path, which is pinned to the reference above."""
    rng = np.random.default_rng(11)
    alphabet = list("abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ") + list("0123456789") * 2 + list("'''  \t\n<|>|.,!?-_/\\\"#$%()*+:;=@[]^`{}~")
    pieces = ["<|startoftext|>", "<|endoftext|>", "'s", "'re", "'ll", "n't", " ", "the", "ing", "'", "<|"]
    caps = []
    for _ in range(4000):
        n = int(rng.integers(0, 60))
        s = "".join(rng.choice(pieces) if rng.random() < 0.12 else rng.choice(alphabet) for _ in range(n))
        caps.append(s)
    ids, native = mini.encode_captions(caps, 40, return_native_mask=True)
    assert native.all()
    for i, c in enumerate(caps):
        assert ids[i].tolist() == mini.caption_ids(c, 40), repr(c)


_ASAN_SCRIPT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "asan_bpe.sh")


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc") or not os.path.exists(_ASAN_SCRIPT),
                    reason="hipcc or tools/asan_bpe.sh not available (the script is CPU-only and does not travel to the GPU box)")
def test_tokenizer_under_address_sanitizer(tmp_path):
    """The host-side C++ of the input pipeline, built with -fsanitize=address,undefined (CPU build: the pool offers no GPU
    sanitizers), fed empty / huge / non-ASCII / NUL-containing / random captions on 1 and 4 threads, leak check on."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TMPDIR=str(tmp_path))
    r = subprocess.run(["bash", os.path.join(root, "tools", "asan_bpe.sh"), bu.MINI_MERGES], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "asan_bpe ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr
