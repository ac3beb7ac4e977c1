// Host-side native BPE tokenizer of the input pipeline (no device code): the text half of BaseDataset.__getitem__
// (reference dataset/base.py:66-83 _load_text over model/base/simple_tokenizer.py:62-148) for a whole batch of captions,
// multi-threaded, behind the C ABI.  Upstream tokenises one caption at a time in Python inside each DataLoader worker.
//
// What is native: captions made of printable ASCII / tab / CR / LF without '&' — for those ftfy.fix_text and html.unescape
// (simple_tokenizer.py:48-51) are the identity, so clean -> lower -> pattern split -> byte-level BPE -> ids is plain byte
// work.  Every other caption is flagged in `status` and goes through the Python path of model/base/simple_tokenizer.py
// (which needs ftfy, exactly like upstream).
#include <algorithm>
#include <atomic>
#include <cstring>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "cmh_common.h"

struct cmh_bpe {
  std::unordered_map<uint64_t, std::pair<int32_t, int32_t>> rank;   // (left id << 32 | right id) -> (merge rank, merged symbol id)
  int32_t byte_id[256], byte_end_id[256];        // id of the byte's symbol / of the same symbol with </w>
  int32_t n_merges = 0;
  int32_t sot = 0, eot = 0, vocab = 0;
};

namespace {

// bytes_to_unicode (simple_tokenizer.py:15-35): printable bytes keep their code point, the others get 256 + n in byte order;
// the vocabulary lists them with the printable ranges first.
void byte_tables(int32_t order[256], std::vector<std::string>& symbol) {
  std::vector<int> bs;
  for (int b = '!'; b <= '~'; ++b) bs.push_back(b);
  for (int b = 0xA1; b <= 0xAC; ++b) bs.push_back(b);
  for (int b = 0xAE; b <= 0xFF; ++b) bs.push_back(b);
  std::vector<int> cs(bs);
  int n = 0;
  for (int b = 0; b < 256; ++b)
    if (std::find(bs.begin(), bs.begin() + 188, b) == bs.begin() + 188) {
      bs.push_back(b);
      cs.push_back(256 + n++);
    }
  symbol.clear();
  for (size_t i = 0; i < bs.size(); ++i) {
    order[bs[i]] = static_cast<int32_t>(i);
    const int cp = cs[i];   // < 0x800: one or two UTF-8 bytes
    std::string u;
    if (cp < 0x80) u.push_back(static_cast<char>(cp));
    else { u.push_back(static_cast<char>(0xC0 | (cp >> 6))); u.push_back(static_cast<char>(0x80 | (cp & 0x3F))); }
    symbol.push_back(u);
  }
}

inline bool is_space(unsigned char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r'; }
inline bool is_letter(unsigned char c) { return c >= 'a' && c <= 'z'; }
inline bool is_digit(unsigned char c) { return c >= '0' && c <= '9'; }

struct Worker {
  const cmh_bpe* t;
  std::unordered_map<std::string, std::vector<int32_t>> cache;
  std::vector<int32_t> word, ids;
  std::string clean;

  // simple_tokenizer.py:81-120: merge the lowest-ranked adjacent pair (all its occurrences, left to right) until none is ranked
  const std::vector<int32_t>& bpe(const char* s, size_t n) {
    std::string key(s, n);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    word.clear();
    for (size_t i = 0; i < n; ++i) word.push_back(i + 1 == n ? t->byte_end_id[static_cast<unsigned char>(s[i])] : t->byte_id[static_cast<unsigned char>(s[i])]);
    while (word.size() > 1) {
      int32_t best = INT32_MAX, a = -1, b = -1, merged = -1;
      for (size_t i = 0; i + 1 < word.size(); ++i) {
        auto r = t->rank.find((static_cast<uint64_t>(static_cast<uint32_t>(word[i])) << 32) | static_cast<uint32_t>(word[i + 1]));
        if (r != t->rank.end() && r->second.first < best) { best = r->second.first; merged = r->second.second; a = word[i]; b = word[i + 1]; }
      }
      if (best == INT32_MAX) break;
      size_t o = 0;
      for (size_t i = 0; i < word.size();) {
        if (i + 1 < word.size() && word[i] == a && word[i + 1] == b) { word[o++] = merged; i += 2; }
        else word[o++] = word[i++];
      }
      word.resize(o);
    }
    return cache.emplace(std::move(key), word).first->second;
  }

  // -> false when the caption needs the Python path
  bool encode(const char* s, size_t n, int32_t max_words, int64_t* out) {
    for (size_t i = 0; i < n; ++i) {
      const unsigned char c = static_cast<unsigned char>(s[i]);
      if (!((c >= 0x20 && c <= 0x7E) || c == '\t' || c == '\n' || c == '\r') || c == '&') return false;
    }
    // whitespace_clean(basic_clean(text)).lower(): strip, runs of whitespace -> one blank, lower case
    clean.clear();
    bool pending = false;
    for (size_t i = 0; i < n; ++i) {
      unsigned char c = static_cast<unsigned char>(s[i]);
      if (is_space(c)) { pending = !clean.empty(); continue; }
      if (pending) { clean.push_back(' '); pending = false; }
      clean.push_back(static_cast<char>(c >= 'A' && c <= 'Z' ? c + 32 : c));
    }
    ids.clear();
    ids.push_back(t->sot);
    const char* p = clean.data();
    const size_t m = clean.size();
    static const char* kSot = "<|startoftext|>";
    static const char* kEot = "<|endoftext|>";
    static const char* kContr[7] = {"'s", "'t", "'re", "'ve", "'m", "'ll", "'d"};
    for (size_t i = 0; i < m;) {
      const unsigned char c = static_cast<unsigned char>(p[i]);
      if (c == ' ') { ++i; continue; }
      size_t len = 0;
      if (c == '<' && m - i >= 15 && std::memcmp(p + i, kSot, 15) == 0) { ids.push_back(t->sot); i += 15; continue; }
      if (c == '<' && m - i >= 13 && std::memcmp(p + i, kEot, 13) == 0) { ids.push_back(t->eot); i += 13; continue; }
      if (c == '\'') {
        for (const char* k : kContr) {
          const size_t kl = std::strlen(k);
          if (m - i >= kl && std::memcmp(p + i, k, kl) == 0) { len = kl; break; }
        }
      }
      if (!len) {
        if (is_letter(c)) { len = 1; while (i + len < m && is_letter(static_cast<unsigned char>(p[i + len]))) ++len; }
        else if (is_digit(c)) len = 1;
        else {
          len = 1;
          while (i + len < m) {
            const unsigned char d = static_cast<unsigned char>(p[i + len]);
            if (d == ' ' || is_letter(d) || is_digit(d)) break;
            ++len;
          }
        }
      }
      const std::vector<int32_t>& w = bpe(p + i, len);
      ids.insert(ids.end(), w.begin(), w.end());
      i += len;
    }
    // dataset/base.py:69-79: [CLS] + words, cut to max_words - 1, + [SEP], zero padding
    if (static_cast<int32_t>(ids.size()) > max_words - 1) ids.resize(max_words - 1);
    ids.push_back(t->eot);
    for (int32_t j = 0; j < max_words; ++j) out[j] = j < static_cast<int32_t>(ids.size()) ? ids[j] : 0;
    return true;
  }
};

}  // namespace

using namespace cmh;

extern "C" int cmh_bpe_create(const char* merges_utf8, size_t bytes, cmh_bpe** out) {
  CMH_CHECK_ARG(merges_utf8 && out, "bpe_create: null pointer");
  cmh_bpe* t = new cmh_bpe();
  std::vector<std::string> symbol;
  int32_t order[256];
  byte_tables(order, symbol);
  // simple_tokenizer.py:66-76: lines 1 .. 49152-256-2 of the file, each split on whitespace.  Every line adds one vocabulary
  // entry (the concatenation of its parts, whatever their number); two-part lines are the ranked merges.  Both upstream tables
  // are dict(zip(...)): a repeated string / pair keeps its LAST index, so ids are resolved after all lines are read.
  const int32_t max_merges = 49152 - 256 - 2;
  std::vector<std::vector<std::string>> lines;
  size_t pos = 0;
  int32_t line = 0;
  while (pos <= bytes && static_cast<int32_t>(lines.size()) < max_merges) {
    const char* nl = pos < bytes ? static_cast<const char*>(std::memchr(merges_utf8 + pos, '\n', bytes - pos)) : nullptr;
    const size_t end = nl ? static_cast<size_t>(nl - merges_utf8) : bytes;
    if (line++ > 0) {
      std::vector<std::string> parts;
      size_t i = pos;
      while (i < end) {
        while (i < end && merges_utf8[i] && std::strchr(" \t\r\f\v", merges_utf8[i])) ++i;
        size_t j = i;
        while (j < end && !(merges_utf8[j] && std::strchr(" \t\r\f\v", merges_utf8[j]))) ++j;
        if (j > i) parts.emplace_back(merges_utf8 + i, j - i);
        i = j;
      }
      lines.push_back(std::move(parts));
    }
    if (!nl) break;
    pos = end + 1;
  }
  const int32_t n = static_cast<int32_t>(lines.size());
  std::unordered_map<std::string, int32_t> id_of;
  for (int i = 0; i < 256; ++i) id_of[symbol[i]] = i;
  for (int i = 0; i < 256; ++i) id_of[symbol[i] + "</w>"] = 256 + i;
  std::vector<std::string> joined(n);
  for (int32_t k = 0; k < n; ++k) {
    for (const auto& p : lines[k]) joined[k] += p;
    id_of[joined[k]] = 512 + k;
  }
  id_of["<|startoftext|>"] = 512 + n;
  id_of["<|endoftext|>"] = 513 + n;
  for (int b = 0; b < 256; ++b) {
    t->byte_id[b] = id_of[symbol[order[b]]];
    t->byte_end_id[b] = id_of[symbol[order[b]] + "</w>"];
  }
  for (int32_t k = 0; k < n; ++k) {
    if (lines[k].size() != 2) continue;
    auto a = id_of.find(lines[k][0]), b = id_of.find(lines[k][1]);
    if (a == id_of.end() || b == id_of.end()) continue;   // a part that is no vocabulary entry can never be produced
    const uint64_t key = (static_cast<uint64_t>(static_cast<uint32_t>(a->second)) << 32) | static_cast<uint32_t>(b->second);
    t->rank[key] = std::make_pair(k, id_of[joined[k]]);
  }
  t->n_merges = n;
  t->sot = 512 + n;
  t->eot = 513 + n;
  t->vocab = 514 + n;
  *out = t;
  return CMH_OK;
}

extern "C" void cmh_bpe_destroy(cmh_bpe* t) { delete t; }

extern "C" int32_t cmh_bpe_vocab_size(const cmh_bpe* t) { return t ? t->vocab : 0; }

extern "C" int cmh_bpe_encode_captions(const cmh_bpe* t, const char* texts, const int64_t* offsets, int32_t n, int32_t max_words,
                                       int64_t* out, uint8_t* status, int32_t threads) {
  CMH_CHECK_ARG(t && texts && offsets && out && status, "bpe_encode_captions: null pointer");
  CMH_CHECK_ARG(n >= 0 && max_words >= 2, "bpe_encode_captions: bad n %d / max_words %d", n, max_words);
  for (int32_t i = 0; i < n; ++i) CMH_CHECK_ARG(offsets[i + 1] >= offsets[i], "bpe_encode_captions: offsets must not decrease");
  int nt = threads > 0 ? threads : static_cast<int>(std::thread::hardware_concurrency());
  nt = std::max(1, std::min(nt, std::min(64, n / 64 + 1)));
  std::atomic<int32_t> next{0};
  auto run = [&]() {
    Worker w;
    w.t = t;
    for (;;) {
      const int32_t lo = next.fetch_add(64);
      if (lo >= n) break;
      const int32_t hi = std::min(n, lo + 64);
      for (int32_t i = lo; i < hi; ++i) {
        int64_t* row = out + static_cast<size_t>(i) * max_words;
        const bool ok = w.encode(texts + offsets[i], static_cast<size_t>(offsets[i + 1] - offsets[i]), max_words, row);
        status[i] = ok ? 0 : 1;
        if (!ok) std::fill(row, row + max_words, static_cast<int64_t>(0));
      }
    }
  };
  if (nt == 1) run();
  else {
    std::vector<std::thread> pool;
    for (int k = 0; k < nt; ++k) pool.emplace_back(run);
    for (auto& th : pool) th.join();
  }
  return CMH_OK;
}
