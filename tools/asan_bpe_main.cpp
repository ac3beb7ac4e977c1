// AddressSanitizer / UBSan driver for the host-side tokenizer (csrc/bpe_tokenizer.hip has no device code): builds the tables
// from a merges file and pushes adversarial captions through cmh_bpe_encode_captions on several threads.
//   tools/asan_bpe.sh <merges.txt>      (run by tests/test_tokenizer.py::test_tokenizer_under_address_sanitizer)
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../include/cmh.h"

namespace cmh {
char* err_buf() { static thread_local char b[512]; return b; }
int fail(int code, const char* fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(err_buf(), 512, fmt, ap); va_end(ap);
  return code;
}
}  // namespace cmh

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 2;
  std::string merges;
  char buf[65536];
  size_t n;
  while ((n = fread(buf, 1, sizeof buf, f)) > 0) merges.append(buf, n);
  fclose(f);
  cmh_bpe* t = nullptr;
  if (cmh_bpe_create(merges.data(), merges.size(), &t) != 0) return 3;
  std::vector<std::string> caps = {"", " ", "a", "'", "<|", "<|startoftext|>", "<|endoftext|><|endoftext|>", "it's we're they'll 'd 's",
                                   "12345 6.7", "!!!...???", std::string(5000, 'x'), std::string(300, '\''), "caf\xc3\xa9", "a & b",
                                   "tab\tnew\nline\r\n", std::string("nul\0byte", 8)};
  unsigned s = 12345;
  for (int i = 0; i < 3000; ++i) {
    std::string c;
    const int len = (s = s * 1664525u + 1013904223u) % 90;
    for (int j = 0; j < len; ++j) c.push_back(static_cast<char>(32 + ((s = s * 1664525u + 1013904223u) >> 8) % 95));
    caps.push_back(c);
  }
  std::string blob;
  std::vector<int64_t> off(1, 0);
  for (auto& c : caps) { blob += c; off.push_back(static_cast<int64_t>(blob.size())); }
  long long checksum = 0;
  for (int max_words : {2, 3, 8, 32, 77}) {
    std::vector<int64_t> out(caps.size() * max_words);
    std::vector<uint8_t> st(caps.size());
    for (int threads : {1, 4}) {
      if (cmh_bpe_encode_captions(t, blob.data(), off.data(), static_cast<int32_t>(caps.size()), max_words, out.data(), st.data(), threads)) return 4;
      for (auto v : out) checksum += v;
    }
  }
  const int vocab = cmh_bpe_vocab_size(t);
  cmh_bpe_destroy(t);
  printf("asan_bpe ok vocab=%d checksum=%lld\n", vocab, checksum);
  return 0;
}
