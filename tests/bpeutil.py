"""Captions and the miniature merges file of the tokenizer tests (shared with tests/golden/make_golden9.py)."""
import collections
import os

HERE = os.path.dirname(os.path.abspath(__file__))
MINI_MERGES = os.path.join(HERE, "golden", "mini_bpe_merges.txt")

CORPUS = """a man riding a wave on top of a surfboard . two dogs playing in the snow near a red barn . the quick brown fox jumps over the
lazy dog . a woman is holding an umbrella while walking down the street . children's toys are scattered on the floor . it's a
beautiful day and we're going to the beach . they'll be there at 10 o'clock , i'd say . sunset over the mountains with clouds ,
flowers , trees and a lake . people sitting at tables in a restaurant eating pizza and drinking wine . a black and white photo of
an old train station . portrait of a girl with blue eyes and long hair . macro shot of a bee on a yellow flower ; nikon d80 ,
50mm f/1.8 . street art graffiti on a wall in berlin 2009 ."""

CAPTIONS = [
    "A man riding a wave on top of a surfboard.",
    "Two dogs playing in the snow",
    "  leading and trailing   spaces\tand\ttabs\nnewlines\r\n ",
    "It's a beautiful day, we're going; they'll see, I'd say: can't won't o'clock 'quoted' 'sup",
    "numbers 12345 and 3.14159, dates 2009-07-04, 50mm f/1.8 @ ISO100",
    "punctuation!!! ... --- ((nested [brackets] {braces})) <tags> \"quotes\" #hash $dollar %percent ^caret *star +plus =eq ~tilde `tick |pipe \\back /slash",
    "<|startoftext|> inside <|endoftext|> text and !<|startoftext|> glued",
    "MiXeD CaSe WoRdS and ALLCAPS",
    "",
    " ",
    "x",
    "7",
    "?",
    "a very long caption " + "with many many words that keeps going and going " * 6,
    "sunset over the mountains with clouds, flowers, trees and a lake",
    "don't can't 'tis 's 't 're 've 'm 'll 'd 'S 'LL",
    "under_score and hy-phen-ated words, e-mail@address.com http://url.example/path?x=1",
    "supercalifragilisticexpialidocious antidisestablishmentarianism zzzzzz qqqq",
]
NON_NATIVE = ["café crème brûlée", "fish &amp; chips &lt;b&gt;", "日本語のキャプション text", "emoji \U0001F600 smile",
              "form\x0cfeed and \x0bvertical"]


def train_mini_merges(n_merges=420):
    """A small byte-pair vocabulary learnt from CORPUS (plain BPE training; deterministic tie-break by pair order)."""
    words = collections.Counter(w for w in CORPUS.lower().split())
    seqs = {w: tuple(w[:-1]) + (w[-1] + "</w>",) for w in words}
    merges = []
    for _ in range(n_merges):
        pairs = collections.Counter()
        for w, c in words.items():
            s = seqs[w]
            for a, b in zip(s, s[1:]):
                pairs[(a, b)] += c
        if not pairs:
            break
        best = max(sorted(pairs), key=lambda p: pairs[p])
        merges.append(best)
        for w in words:
            s, out, i = seqs[w], [], 0
            while i < len(s):
                if i + 1 < len(s) and (s[i], s[i + 1]) == best:
                    out.append(s[i] + s[i + 1]); i += 2
                else:
                    out.append(s[i]); i += 1
            seqs[w] = tuple(out)
    return merges


def write_mini_merges(path=MINI_MERGES):
    merges = train_mini_merges()
    lines = ['"bpe_simple_vocab_mini - version: 0.1"']
    for k, (a, b) in enumerate(merges):
        lines.append(f"{a} {b}")
        if k == 40:
            lines += ["", "th e</w>", "zz", "q q q"]      # empty / repeated-pair / one-part / three-part lines: dict(zip(...)) corner cases
    lines.append(f"{merges[3][0]} {merges[3][1]}")       # a repeated merge: its LAST rank counts upstream
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
