// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
//
// CPU restatement of the reference's Hamming-ranking mAP:
//   /root/reference/utils/calc_utils.py:8-13   calc_hammingDist  -> 0.5*(q - B1.B2^T), fp32
//   /root/reference/utils/calc_utils.py:16-39  calc_map_k_matrix -> per-query loop:
//        gnd  = (query_L[i] . retrieval_L^T > 0)                       (:26)
//        tsum = sum(gnd); skipped queries still count in the mean      (:27-29, :38)
//        hamm = calc_hammingDist(qB[i], rB)                            (:30)
//        _, ind = torch.sort(hamm)      <- UNSTABLE CPU sort           (:31)
//        total  = min(k, int(tsum))                                    (:34)
//        tindex = nonzero(gnd[ind])[:total] + 1                        (:35-36)
//        AP     = mean(arange(1..total) / tindex)                      (:37)
//   mAP = sum(AP) / num_query                                          (:38)
//
// Tie order.  torch.sort(stable=False) on a CPU float tensor sorts (value,index)
// pairs with libstdc++ std::sort and a key-only comparator (ATen/native/cpu/
// SortingKernel.cpp, PyTorch 2.10).  Hamming distances take <= q+1 distinct values so
// the order inside a tie is whatever introsort leaves behind; this file reproduces it
// by running the very same std::sort on (float key, int64 idx) pairs.  Pinned against
// outputs of the reference itself: tests/golden/make_golden.py -> tests/golden/map_*.npz.
//
// Build: make -C oracle   (g++ -O2 -shared -fPIC; output oracle/_build/libcmh_oracle.so)

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

struct KV {
  float key;
  int64_t idx;
};

struct KeyLess {
  bool operator()(const KV& a, const KV& b) const { return a.key < b.key; }
};

}  // namespace

extern "C" {

// Hamming "distance" row exactly as calc_utils.py:8-13 computes it, in fp32.
// codes are arbitrary floats (the reference never checks they are +-1).
void oracle_hamming_row(const float* q, const float* rB, int64_t N, int64_t K, float* out) {
  for (int64_t j = 0; j < N; ++j) {
    const float* r = rB + j * K;
    float dot = 0.f;
    for (int64_t t = 0; t < K; ++t) dot += q[t] * r[t];
    out[j] = 0.5f * (static_cast<float>(K) - dot);
  }
}

// The permutation torch.sort(hamm) returns on CPU (unstable; libstdc++ introsort).
void oracle_sort_perm(const float* keys, int64_t N, int64_t* ind_out) {
  std::vector<KV> kv(static_cast<size_t>(N));
  for (int64_t j = 0; j < N; ++j) kv[j] = KV{keys[j], j};
  std::sort(kv.begin(), kv.end(), KeyLess());
  for (int64_t j = 0; j < N; ++j) ind_out[j] = kv[j].idx;
}

// Same, but with the introsort depth budget forced (libstdc++ uses 2*floor(log2 N)); drives the
// heapsort fallback in tests.  Uses libstdc++'s own internals, not a re-implementation.
void oracle_sort_perm_depth(const float* keys, int64_t N, int64_t depth_limit, int64_t* ind_out) {
  std::vector<KV> kv(static_cast<size_t>(N));
  for (int64_t j = 0; j < N; ++j) kv[j] = KV{keys[j], j};
  if (N > 0) {
    auto cmp = __gnu_cxx::__ops::__iter_comp_iter(KeyLess());
    std::__introsort_loop(kv.begin(), kv.end(), depth_limit, cmp);
    std::__final_insertion_sort(kv.begin(), kv.end(), cmp);
  }
  for (int64_t j = 0; j < N; ++j) ind_out[j] = kv[j].idx;
}

// Full restatement of calc_map_k_matrix.  k <= 0 means "k = N" (reference's k=None).
// ap_out[Q]  : per-query AP (0 for skipped queries)      (may be null)
// ind_out    : Q*N permutation, only written if non-null
// stable != 0: use std::stable_sort instead (regression canary, NOT the reference order)
// returns mAP as the reference would: fp32 running sum / Q.
float oracle_map_k(const float* qB, const float* rB, const float* qL, const float* rL,
                   int64_t Q, int64_t N, int64_t K, int64_t C, int64_t k, int stable,
                   float* ap_out, int64_t* ind_out) {
  if (k <= 0) k = N;
  std::vector<float> hamm(static_cast<size_t>(N));
  std::vector<uint8_t> gnd(static_cast<size_t>(N));
  std::vector<KV> kv(static_cast<size_t>(N));
  float map = 0.f;
  for (int64_t i = 0; i < Q; ++i) {
    const float* ql = qL + i * C;
    int64_t tsum = 0;
    for (int64_t j = 0; j < N; ++j) {
      const float* rl = rL + j * C;
      float dot = 0.f;
      for (int64_t c = 0; c < C; ++c) dot += ql[c] * rl[c];
      gnd[j] = dot > 0.f;
      tsum += gnd[j];
    }
    if (ap_out) ap_out[i] = 0.f;
    if (tsum == 0) {
      if (ind_out) std::memset(ind_out + i * N, 0xff, sizeof(int64_t) * N);
      continue;
    }
    oracle_hamming_row(qB + i * K, rB, N, K, hamm.data());
    for (int64_t j = 0; j < N; ++j) kv[j] = KV{hamm[j], j};
    if (stable)
      std::stable_sort(kv.begin(), kv.end(), KeyLess());
    else
      std::sort(kv.begin(), kv.end(), KeyLess());
    if (ind_out)
      for (int64_t j = 0; j < N; ++j) ind_out[i * N + j] = kv[j].idx;
    const int64_t total = std::min<int64_t>(k, tsum);
    // mean(count / tindex) in fp32 like torch (count, tindex are fp32 tensors)
    double acc = 0.0;
    int64_t found = 0;
    for (int64_t j = 0; j < N && found < total; ++j) {
      if (gnd[kv[j].idx]) {
        ++found;
        acc += static_cast<double>(static_cast<float>(found) / static_cast<float>(j + 1));
      }
    }
    const float ap = static_cast<float>(acc / static_cast<double>(total));
    if (ap_out) ap_out[i] = ap;
    map += ap;
  }
  return map / static_cast<float>(Q);
}

}  // extern "C"
