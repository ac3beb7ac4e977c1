// Hamming ranking + mAP on packed codes — replaces utils/calc_utils.py of the reference:
//   calc_hammingDist  :8-13   0.5*(K - q.r)         -> AND/XOR + popcount on bit-planes
//   calc_neighbor     :42-45  (la.lb^T > 0)         -> any(la & lb)
//   calc_map_k_matrix :16-39  per-query sort + AP   -> map_query_kernel below
//
// Tie order.  The reference ranks with torch.sort(stable=False) on the CPU, i.e. libstdc++
// std::sort (introsort) over (key, index) pairs with a key-only comparator; Hamming keys take <= 2K+1
// values, so mAP depends on where introsort leaves equal keys (stable order differs by up to 2.7e-3).
// map_query_kernel therefore EMULATES libstdc++'s introsort exactly, but breadth-first and in parallel:
//   * the recursion tree of __introsort_loop is processed level by level (both children of a
//     partition inherit the same depth budget, so a level == one depth value);
//   * __unguarded_partition (Hoare, both pointers stop on keys equal to the pivot) is replaced by its
//     closed form: with L = ascending positions whose key >= pivot and R = descending positions whose
//     key <= pivot (original values), the sequential loop swaps exactly the pairs (L_i, R_i) for
//     i < s, s = #{i : L_i < R_i} (a prefix), and returns cut = min(L_s, R_{s-1}).  L and R are built
//     with wave ballots (+ a 16-entry scan when the whole workgroup shares one big segment), the swaps
//     are independent;
//   * __move_median_to_first and the depth-exhausted heapsort (__partial_sort) are restated verbatim
//     (sequential, one lane; the latter is a rare fallback);
//   * __final_insertion_sort is a stable sort of an array whose unsorted runs are the <=16-element
//     leaves, so it equals a stable insertion sort of each leaf (one thread per leaf).
// Keys: key = K - q.r in [0, 2K] (an order-preserving integer image of 0.5*(K - q.r)), exact for codes in
// {-1,0,+1}: q.r = popc(nzq & nzr) - 2*popc((sq ^ sr) & nzq & nzr).
//
// One 1024-thread workgroup per query (grid-strided).  Elements are u32 = key << 19 | index, kept in
// LDS when the query's working set fits (N <= ~23k: MIRFlickr-scale) and in a per-workgroup slice of the
// caller's workspace otherwise (COCO / NUS-WIDE scale).  Integer/byte work, HBM/LDS-latency bound — no
// MFMA on purpose.
#include <cstdlib>
#include <cstring>

#include "cmh_common.h"

namespace cmh {

constexpr int NT = 1024;           // threads per workgroup
constexpr int NWAVE = NT / 64;
constexpr int kIdxBits = 19;       // N < 2^19 = 524288
constexpr uint32_t kIdxMask = (1u << kIdxBits) - 1;
constexpr int kMaxWords = 64;      // K <= 2048 bits
#ifndef CMH_MAP_CHUNK
#define CMH_MAP_CHUNK 1024
#endif
#ifndef CMH_MAP_SEQMAX
#define CMH_MAP_SEQMAX 32
#endif
constexpr int kChunk = CMH_MAP_CHUNK;       // a longer segment is partitioned in chunks of this many positions, one wave per chunk
#ifndef CMH_MAP_CHUNK_MASKS
#define CMH_MAP_CHUNK_MASKS 4096
#endif
// ... and of this many where the chunk passes work on bit masks (the workspace placement): fewer, longer walks per level.  A/B on one
// box, one direction: COCO 5000 x 117 218: 35.7 ms (1024) / 32.7 (2048) / 32.4 (4096), NUS-WIDE 2100 x 190 834 x 128 bit: 28.8 / 26.1 /
// 25.4; 512: 43.4 / 34.9.  (The LDS placements lose with 2048: MIRFlickr 2.77 -> 2.93 ms.)  64 groups of 64 positions is the most a
// wave's one-pass seek inside a chunk covers.
constexpr int kChunkMasks = CMH_MAP_CHUNK_MASKS;
static_assert(kChunk % 64 == 0 && kChunkMasks % 64 == 0 && kChunkMasks >= kChunk, "chunks are whole 64-position groups; the stores are sized for kChunk");
constexpr int kLeaf = 16;          // libstdc++ _S_threshold
constexpr int kSeqMax = CMH_MAP_SEQMAX;        // segments of 17..kSeqMax elements are finished sequentially, one lane each

__device__ __forceinline__ int ekey(uint32_t e) { return static_cast<int>(e >> kIdxBits); }
__device__ __forceinline__ int eidx(uint32_t e) { return static_cast<int>(e & kIdxMask); }

__device__ __forceinline__ uint64_t lanemask_lt(int lane) { return (1ull << lane) - 1ull; }

// positions map for the tmp lists: a segment [f,l) with l-f >= 17 owns tmp[g(f) .. g(l)) which holds
// 2*((l-f)/2+1) <= (l-f)+2 entries.
__device__ __host__ __forceinline__ int tmp_base(int x) { return x + 2 * (x / 17); }

struct QueryStore {
  uint32_t* elem;     // [N]
  uint32_t* tmp;      // [tmp_base(N) + 8]
  uint32_t* qa;       // segment queue A: pairs (f,l)  [2*(N/17+2)]
  uint32_t* qb;       // segment queue B
  uint32_t* leafbits; // [(N+31)/32] bit x = a segment starts at x
  uint32_t* relbits;  // [(N+31)/32] bit j = database item j is relevant to the query
  uint32_t* smallbits;// [(N+31)/32] bit x = a parked (17..kSeqMax element) segment starts at x
  uint32_t* task;     // [tcap] one level's work list: segment index << 11 | chunk
  uint32_t* cnt;      // [3*ccap] per chunk: L count, R count, swaps
  uint32_t* seginfo;  // [N/(kSeqMax+1)+2] per queued segment: first chunk | number of chunks << 16
  int tcap, ccap;
};

__host__ __device__ inline size_t store_words(int64_t N) {
  const size_t n = static_cast<size_t>(N);
  const size_t tmpw = n + 2 * (n / 17) + 8;
  const size_t qw = 2 * (n / 17 + 2);
  const size_t bw = (n + 31) / 32;
  const size_t tcap = n / (kSeqMax + 1) + n / kChunk + 8, ccap = 2 * (n / kChunk) + 8;
  return n + tmpw + 2 * qw + 3 * bw + tcap + 3 * ccap + (n / (kSeqMax + 1) + 2) + 16;
}

__device__ inline QueryStore carve_store(uint32_t* base, int N) {
  QueryStore s;
  const size_t n = static_cast<size_t>(N);
  const size_t tmpw = n + 2 * (n / 17) + 8;
  const size_t qw = 2 * (n / 17 + 2);
  const size_t bw = (n + 31) / 32;
  s.elem = base;
  s.tmp = s.elem + n;
  s.qa = s.tmp + tmpw;
  s.qb = s.qa + qw;
  s.leafbits = s.qb + qw;
  s.relbits = s.leafbits + bw;
  s.smallbits = s.relbits + bw;
  s.tcap = static_cast<int>(n / (kSeqMax + 1) + n / kChunk + 8);
  s.ccap = static_cast<int>(2 * (n / kChunk) + 8);
  s.task = s.smallbits + bw;
  s.cnt = s.task + s.tcap;
  s.seginfo = s.cnt + 3 * s.ccap;
  return s;
}

// Hybrid placement (mid-size N): the elements, the bitmaps and the task lists stay in LDS, the L / R position lists and the two segment
// queues go to the workgroup's workspace slice: ~4.4 bytes of LDS per item instead of ~9.9, so that TWO queries share a CU.
__host__ __device__ inline size_t hybrid_lds_words(int64_t N) {
  const size_t n = static_cast<size_t>(N);
  const size_t bw = (n + 31) / 32;
  const size_t tcap = n / (kSeqMax + 1) + n / kChunk + 8, ccap = 2 * (n / kChunk) + 8;
  return n + 3 * bw + tcap + 3 * ccap + (n / (kSeqMax + 1) + 2) + 16;
}
__host__ __device__ inline size_t hybrid_glob_words(int64_t N) {
  const size_t n = static_cast<size_t>(N);
  return (n + 2 * (n / 17) + 8) + 2 * (2 * (n / 17 + 2)) + 16;
}
__device__ inline QueryStore carve_hybrid(uint32_t* lds, uint32_t* glob, int N) {
  QueryStore s;
  const size_t n = static_cast<size_t>(N);
  const size_t tmpw = n + 2 * (n / 17) + 8;
  const size_t qw = 2 * (n / 17 + 2);
  const size_t bw = (n + 31) / 32;
  s.elem = lds;
  s.leafbits = s.elem + n;
  s.relbits = s.leafbits + bw;
  s.smallbits = s.relbits + bw;
  s.tcap = static_cast<int>(n / (kSeqMax + 1) + n / kChunk + 8);
  s.ccap = static_cast<int>(2 * (n / kChunk) + 8);
  s.task = s.smallbits + bw;
  s.cnt = s.task + s.tcap;
  s.seginfo = s.cnt + 3 * s.ccap;
  s.tmp = glob;
  s.qa = s.tmp + tmpw;
  s.qb = s.qa + qw;
  return s;
}

// ---- sync helpers -----------------------------------------------------------------------------------
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  __builtin_amdgcn_wave_barrier();
}

// A value every lane holds alike, said so to the compiler.  What a wave reads from the task list, the segment queue or a header is the same
// in all 64 lanes, but a vector load's result counts as divergent: without this every count, bound and loop condition derived from it
// is computed per lane (v_bcnt, v_cmp + exec masks) - in the mask walk's fill loop ~22 vector instructions per 64-position group where
// ~8 are needed, on a CU whose 32 waves all queue for the vector ALU.
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t uni(uint32_t v) { return static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(v))); }

__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- libstdc++ pieces restated -------------------------------------------------------------------------
// std::__move_median_to_first(result=f, a=f+1, b=mid, c=l-1) with a key-only '<'   (bits/stl_algo.h)
__device__ inline void median_to_first(uint32_t* e, int f, int l) {
  const int a = f + 1, b = f + (l - f) / 2, c = l - 1;
  const uint32_t ea = e[a], eb = e[b], ec = e[c];
  const int ka = ekey(ea), kb = ekey(eb), kc = ekey(ec);
  int pick;
  if (ka < kb) {
    if (kb < kc) pick = b;
    else if (ka < kc) pick = c;
    else pick = a;
  } else if (ka < kc) pick = a;
  else if (kb < kc) pick = c;
  else pick = b;
  const uint32_t ef = e[f];
  e[f] = e[pick];
  e[pick] = ef;
}

// std::__adjust_heap + std::__push_heap (bits/stl_heap.h), key-only '<'
__device__ inline void adjust_heap(uint32_t* first, int hole, int len, uint32_t value) {
  const int top = hole;
  int child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (ekey(first[child]) < ekey(first[child - 1])) child--;
    first[hole] = first[child];
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    first[hole] = first[child - 1];
    hole = child - 1;
  }
  int parent = (hole - 1) / 2;
  while (hole > top && ekey(first[parent]) < ekey(value)) {
    first[hole] = first[parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  first[hole] = value;
}

// std::__partial_sort(first, last, last) == __heap_select (make_heap) + __sort_heap
__device__ inline void heap_sort_segment(uint32_t* e, int f, int l) {
  uint32_t* first = e + f;
  const int len = l - f;
  if (len < 2) return;
  for (int parent = (len - 2) / 2;; --parent) {
    const uint32_t v = first[parent];
    adjust_heap(first, parent, len, v);
    if (parent == 0) break;
  }
  for (int last = len; last > 1;) {
    --last;
    const uint32_t v = first[last];
    first[last] = first[0];
    adjust_heap(first, 0, last, v);
  }
}

// ---- one Hoare partition, closed form -------------------------------------------------------------------
// With L = ascending positions of [f+1, l) whose key >= pivot and R = descending positions whose key <= pivot (original values),
// std::__unguarded_partition swaps exactly the pairs (L_i, R_i), i < s, s = #{i : L_i < R_i} (a prefix), and returns
// min(L_s, R_{s-1}).  A pair beyond cap = n/2+1 can never swap, so the lists are cut there.
// The partition by ONE wave, in one pass per list and without a count pass: L is filled walking up from f+1, R walking down
// from l-1 (a lane's rank = the running count + v_mbcnt of the ballot), and each walk stops once its list holds cap entries (no pair
// beyond cap can swap).  U = 64-element chunks per iteration (their reads are issued back to back); short segments take U = 1 and
// never loop.  Everything here is issue-bound - four waves share a SIMD - so instructions, not LDS round trips, are what is counted.
__device__ __forceinline__ int mbcnt64(uint64_t m) {
  return __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u));
}

// A segment whose keys are all equal (most of a Hamming ranking ends in such runs: <= 2K+1 key values) needs no data to be sorted:
// median-to-first swaps positions 0 and n/2, the partition loop swaps (1+i, n-1-i) until they meet, i.e. reverses [1, n-1], and
// returns cut = 1 + (n-1)/2; both children are all-equal again, and the final insertion sort moves nothing.  So where an element
// of the run ends up is a function of (n, its position) alone, and the whole introsort subtree is applied as ONE permutation
// (through the segment's tmp slice) - its remaining levels, parked segments and leaves are never visited.  `budget` = depth budget
// of the children: the shortcut is taken only when no heapsort fallback can occur below (levels needed <= budget).
__device__ inline bool all_equal_shortcut(const QueryStore& S, int f, int l, int lane, int budget) {
  uint32_t* e = S.elem;
  const int n = l - f;
  const int key0 = ekey(e[f]);
  for (int x0 = f + 1; x0 < l; x0 += 256) {
    bool diff = false;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int x = x0 + 64 * u + lane;
      if (x < l) diff |= ekey(e[x]) != key0;
    }
    if (__ballot(diff)) return false;
  }
  int levels = 0;
  for (int m = 1 + (n - 1) / 2; m > kLeaf; m = 1 + (m - 1) / 2) ++levels;      // the left child is never the smaller one
  if (levels > budget) return false;
  uint32_t* slice = S.tmp + tmp_base(f);
  for (int x = f + lane; x < l; x += 64) {
    const uint32_t v = e[x];
    int pos = x - f, nn = n, base = 0;
    while (nn > kLeaf) {
      const int mid = nn / 2;
      pos = pos == 0 ? mid : (pos == mid ? 0 : pos);
      pos = pos == 0 ? 0 : nn - pos;
      const int c = 1 + (nn - 1) / 2;
      if (pos < c) nn = c;
      else { pos -= c; base += c; nn -= c; }
    }
    slice[base + pos] = v;
  }
  wave_sync();
  for (int x = f + lane; x < l; x += 64) e[x] = slice[x - f];
  wave_sync();
  return true;
}

// returns the cut, or -1 when the segment was finished by the all-equal shortcut (no children)
template <int U>
__device__ inline int partition_wave(const QueryStore& S, int f, int l, int lane, int budget) {
  uint32_t* e = S.elem;
  const int n = l - f;
  if (all_equal_shortcut(S, f, l, lane, budget)) return -1;
  if (lane == 0) median_to_first(e, f, l);
  wave_sync();
  const int p = uni(ekey(e[f]));
  const int cap = n / 2 + 1;
  uint32_t* tL = S.tmp + tmp_base(f);
  uint32_t* tR = tL + cap;
  int Lc = 0, Rc = 0;
  for (int x0 = f + 1; x0 < l && Lc < cap; x0 += 64 * U) {
    int k[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int x = x0 + 64 * u + lane;
      k[u] = x < l ? ekey(e[x]) : -1;                       // -1 < every pivot
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool isL = k[u] >= p;
      const uint64_t m = __ballot(isL);
      const int rk = Lc + mbcnt64(m);
      if (isL && rk < cap) tL[rk] = static_cast<uint32_t>(x0 + 64 * u + lane);
      Lc += __popcll(m);
    }
  }
  for (int x0 = l - 1; x0 > f && Rc < cap; x0 -= 64 * U) {
    int k[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int x = x0 - 64 * u - lane;
      k[u] = x > f ? ekey(e[x]) : 0x7fffffff;               // above every pivot
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool isR = k[u] <= p;
      const uint64_t m = __ballot(isR);
      const int rk = Rc + mbcnt64(m);                        // lane order == descending position
      if (isR && rk < cap) tR[rk] = static_cast<uint32_t>(x0 - 64 * u - lane);
      Rc += __popcll(m);
    }
  }
  wave_sync();
  int npairs = Lc < Rc ? Lc : Rc;
  npairs = npairs < cap ? npairs : cap;
  int s = 0;
  for (int i0 = 0; i0 < npairs; i0 += 64) {
    const int i = i0 + lane;
    bool sw = false;
    if (i < npairs) {
      const uint32_t xl = tL[i], xr = tR[i];
      sw = xl < xr;
      if (sw) {
        const uint32_t t0 = e[xl];
        e[xl] = e[xr];
        e[xr] = t0;
      }
    }
    const int c = __popcll(__ballot(sw));
    s += c;
    if (c < 64) break;                                       // the swapping pairs are a prefix
  }
  const int lim = Lc < cap ? Lc : cap;
  const uint32_t c1 = s < lim ? tL[s] : 0x7fffffffu;
  const uint32_t c2 = s >= 1 ? tR[s - 1] : 0x7fffffffu;
  wave_sync();
  return static_cast<int>(c1 < c2 ? c1 : c2);
}

// ---- chunked segments, round 4: the two position lists as BIT MASKS -----------------------------------------------------------------
// Rounds 1-3 partitioned a segment of more than kChunk positions in four passes: count (L / R entries per chunk), write (every entry's
// POSITION into the two lists, at its global rank), swap (pairs (L_i, R_i)), cut.  Whether position x belongs to L / R is one bit each,
// and a bit needs no rank to be written: ONE pass stores, per 64 positions, the two ballots (16 bytes instead of up to 2 x 256) and
// counts per chunk; a wave-parallel prefix over the chunk counts follows; the swap pass walks the masks - L ascending, R descending -
// from the rank its piece starts at (located through the prefix: chunk, then group, then bit) and expands 64 ranks at a time into a
// per-wave LDS buffer.  Same pairs, same swaps, same cut: the permutation is unchanged.
// Used by the WORKSPACE placement only (introsort_phases<MASKS>): there the write pass and its lists are traffic (COCO 5000 x 117 218:
// 35.3 ms against 39.3, NUS-WIDE 2100 x 190 834 x 128 bit: 28.9 against 31.0, profiles/r04_n_map_*.txt); with the elements in LDS the
// walk's extra instructions cost more than the lists (MIRFlickr 5000 x 15 015: 4.4 ms against 2.8), so those placements keep the lists.
// What bounds the workspace placement after this: every level still reads the element array twice (count, swap gathers) and writes it
// once in partial lines - 60 GB per NUS-WIDE direction at 2.3-2.5 TB/s (FETCH_SIZE / WRITE_SIZE, profiles/r04_*_map_traffic_nuswide.txt);
// deeper unrolling of the count pass changes nothing (4 / 8 / 16 groups in flight: 22.9 / 22.9 / 23.5 ms), and finishing short segments
// in LDS (tried: copy in, sort, copy out) costs as many cycles per element as the levels it replaces.  What DID pay after the masks:
// chunks of 4096 positions (kChunkMasks), the swap count known before the swap pass (pass B below: equal pieces, no idle pieces), and
// telling the compiler which loaded values are wave-uniform (uni()): the walk's loops ran on vector compares and exec masks.
#ifndef CMH_MAP_MASKS
#define CMH_MAP_MASKS 1
#endif
#ifndef CMH_MAP_UNROLL_A
#define CMH_MAP_UNROLL_A 4        // 64-position groups in flight per wave in the count pass (8, 16: the same time)
#endif
#ifndef CMH_MAP_SWAP_BLOCKS
#define CMH_MAP_SWAP_BLOCKS 1     // 64-rank blocks per iteration of the swap pass.  2: the same time, 4: +11 % (COCO 40.1 ms against 35.9).
#endif                            // (Also tried: no rank buffers - what is left of the current L group paired directly with what is left of the
                                  // current R group, ~20 pairs per step: swap pass 0.5-0.65 M -> 0.75-1.26 M cycles per level, NUS-WIDE 25.2 -> 29.5 ms:
                                  // the pass costs one dependent gather -> scatter round trip per step, so fewer, fuller steps win.  And the
                                  // full fence of wave_sync() in this loop (it also waits for the stores) against an LDS-only wait: no difference.)
// mask area of a chunked segment [f, l): u32 words inside its tmp slice: [Lc, Rc, -, -], then per group g (positions f+1+64g ...) four
// words: L lo, L hi, R lo, R hi.  4 + 4 * ceil((l-f-1)/64) <= (l-f) - 3 words for every l-f > kChunk.
__device__ __forceinline__ uint32_t* mask_area(const QueryStore& S, int f) {      // 16-byte aligned ADDRESS (tmp itself may not be)
  return reinterpret_cast<uint32_t*>((reinterpret_cast<uintptr_t>(S.tmp + tmp_base(f)) + 15) & ~static_cast<uintptr_t>(15));
}

struct MaskCursor {        // one side of a segment's masks, walked by a whole wave (all fields wave-uniform except the window)
  const uint32_t* mk;      // words of group 0 of this side (L: area + 4, R: area + 6)
  int ngroups, xbase, gi, gwin;
  bool desc;               // R: groups and bits from the top
  uint32_t wlo, whi;       // per lane: the masks of group gwin + lane (gwin - lane when desc)
  uint64_t m;              // what is left of group gi's mask
};
__device__ __forceinline__ void cursor_window(MaskCursor& c, int g0, int lane) {
  c.gwin = g0;
  const int g = c.desc ? g0 - lane : g0 + lane;
  c.wlo = 0u; c.whi = 0u;
  if (g >= 0 && g < c.ngroups) {
    const uint32_t* q = c.mk + 4 * g;
    c.wlo = q[0];
    c.whi = q[1];
  }
}
__device__ __forceinline__ uint64_t cursor_mask(MaskCursor& c, int g, int lane) {   // g in [0, ngroups)
  int k = c.desc ? c.gwin - g : g - c.gwin;
  if (k < 0 || k > 63) { cursor_window(c, g, lane); k = 0; }
  // lane k's words, by v_readlane (k is wave-uniform; a __shfl here is two LDS round trips on the walk's critical path)
  const int ks = __builtin_amdgcn_readfirstlane(k);
  const uint32_t ulo = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(c.wlo), ks));      // (ints: no sign extension)
  const uint32_t uhi = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(c.whi), ks));
  return static_cast<uint64_t>(ulo) | (static_cast<uint64_t>(uhi) << 32);
}
// bits of m above lane's own
__device__ __forceinline__ int bits_above(uint64_t m, int lane) { return lane == 63 ? 0 : __popcll(m >> (lane + 1)); }

// The entry of walk rank `target` (0-based; it exists): positions the cursor on its group with every earlier entry of that group
// cleared.  pref[c] = entries in the chunks walked before chunk c (exclusive prefix, ascending for L, from the top for R).
template <int kGpc>      // 64-position groups per chunk
__device__ inline void cursor_seek(MaskCursor& c, const uint32_t* pref, int nc, int target, int lane) {
  int cle = 0;
  for (int c0 = 0; c0 < nc; c0 += 64) {
    const int ci = c0 + lane;
    cle += __popcll(__ballot(ci < nc && static_cast<int>(pref[ci]) <= target));
  }
  const int ch = c.desc ? nc - cle : cle - 1;
  const int skip = target - static_cast<int>(uni(pref[ch]));
  // the chunk's groups in walk order on the first kGpc lanes
  static_assert(kGpc >= 1 && kGpc <= 64, "a chunk is a whole number (<= 64) of 64-position groups");
  const int gfirst = c.desc ? (kGpc * ch + kGpc - 1 < c.ngroups - 1 ? kGpc * ch + kGpc - 1 : c.ngroups - 1) : kGpc * ch;
  const int g = c.desc ? gfirst - lane : gfirst + lane;
  const bool valid = lane < kGpc && g >= kGpc * ch && g < c.ngroups && g < kGpc * ch + kGpc;
  uint32_t wlo = 0u, whi = 0u;       // (plain scalars: element accesses of a vector type next to lane reads have miscompiled before)
  if (valid) {
    const uint32_t* q = c.mk + 4 * g;
    wlo = q[0];
    whi = q[1];
  }
  const int pc = __popc(wlo) + __popc(whi);
  int incl = pc;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int up = __shfl_up(incl, o, 64);
    if (lane >= o) incl += up;
  }
  const int gsel = __popcll(__ballot(valid && incl <= skip));           // first group (in walk order) whose inclusive count exceeds skip
  const int local = skip - (__builtin_amdgcn_readlane(incl, gsel) - __builtin_amdgcn_readlane(pc, gsel));
  const uint64_t m = (static_cast<uint64_t>(static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(whi), gsel))) << 32) |
                     static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(wlo), gsel));
  const bool bit = (m >> lane) & 1ull;
  const int rank = c.desc ? bits_above(m, lane) : mbcnt64(m);
  c.gi = c.desc ? gfirst - gsel : gfirst + gsel;
  c.m = __ballot(bit && rank >= local);
  cursor_window(c, c.gi, lane);
}
// the next (up to) `want` entries of the walk -> buf[0 ..]; returns how many there were
__device__ inline int cursor_fill(MaskCursor& c, uint32_t* buf, int lane, int want) {
  int have = 0;
  while (have < want) {
    if (c.m == 0ull) {
      const int g = c.desc ? c.gi - 1 : c.gi + 1;
      if (g < 0 || g >= c.ngroups) break;
      c.gi = g;
      c.m = cursor_mask(c, g, lane);
      continue;
    }
    const uint64_t m = c.m;
    const int cnt = __popcll(m);
    const bool bit = (m >> lane) & 1ull;
    const int rank = c.desc ? bits_above(m, lane) : mbcnt64(m);
    const int slot = have + rank;
    if (bit && slot < want) buf[slot] = static_cast<uint32_t>(c.xbase + 64 * c.gi + lane);
    const int take = cnt < want - have ? cnt : want - have;
    c.m = take < cnt ? __ballot(bit && rank >= take) : 0ull;
    have += take;
  }
  return have;
}
__device__ __forceinline__ MaskCursor make_cursor(const QueryStore& S, int f, int l, bool desc) {
  MaskCursor c;
  c.mk = mask_area(S, f) + 4 + (desc ? 2 : 0);
  c.ngroups = (l - f - 1 + 63) >> 6;
  c.xbase = f + 1;
  c.desc = desc;
  c.gi = 0; c.gwin = -1000000; c.wlo = 0; c.whi = 0; c.m = 0ull;
  return c;
}
// position of the entry of walk rank `target` (wave-uniform)
template <int kGpc>
__device__ inline int cursor_select(const QueryStore& S, int f, int l, bool desc, const uint32_t* pref, int nc, int target, int lane) {
  MaskCursor c = make_cursor(S, f, l, desc);
  cursor_seek<kGpc>(c, pref, nc, target, lane);
  const int b = desc ? 63 - __clzll(static_cast<long long>(c.m)) : __ffsll(static_cast<long long>(c.m)) - 1;
  return c.xbase + 64 * c.gi + b;
}

// ---- phases 1, 1b, 2 of one introsort: S.elem[0, N) with depth budget depth0 --------------------------------------------------------
// The workgroup's static LDS (the kernel owns it; the phases only borrow it)
struct WgCtx {
  int* stask;       // [3]
  int* sqcount;     // [2]
  int* shist;       // [NWAVE * 64]
  uint32_t* bufR;   // [NWAVE * 128 * CMH_MAP_SWAP_BLOCKS]: the mask walk's L and R positions of one iteration, per wave
};
// Children of a partition: > kSeqMax elements -> next breadth-first level; 17..kSeqMax -> parked for the sequential
// finisher (start bit in smallbits, (end, depth budget) in the segment's own slice of tmp); <= 16 -> leaf.
__device__ __forceinline__ void push_seg(const QueryStore& S, uint32_t* q, int* qcount, int f, int l, int depth) {
  const int n = l - f;
  if (n > kSeqMax) {
    const int i = atomicAdd(qcount, 1);
    q[2 * i] = static_cast<uint32_t>(f);
    q[2 * i + 1] = static_cast<uint32_t>(l);
  } else if (n > kLeaf) {
    uint32_t* slot = S.tmp + tmp_base(f);
    slot[0] = static_cast<uint32_t>(l);
    slot[1] = static_cast<uint32_t>(depth);
    atomicOr(&S.smallbits[f >> 5], 1u << (f & 31));
  }
}

// std::__unguarded_partition_pivot, sequential (one lane), exactly as libstdc++ walks it.
__device__ inline int seq_partition_pivot(uint32_t* e, int first, int last) {
  median_to_first(e, first, last);
  const int pk = ekey(e[first]);
  int i = first + 1, j = last;
  while (true) {
    while (ekey(e[i]) < pk) ++i;
    --j;
    while (pk < ekey(e[j])) --j;
    if (!(i < j)) return i;
    const uint32_t t = e[i];
    e[i] = e[j];
    e[j] = t;
    ++i;
  }
}

// std::__introsort_loop restricted to one parked segment [f,l) (17..kSeqMax elements) and run by ONE lane (its cuts become leaf
// boundaries, so the final insertion sort of its leaves happens in phase 2 with everybody else's): small segments are latency chains, so a wave finishes 64 of them side by side instead of spending a whole
// ballot/scan round trip on each.  Explicit stack (packed first|last|depth, relative to f) in the segment's tmp slice:
// it never holds more than min(depth, n-16) <= n entries.
__device__ inline void seq_finish_segment(uint32_t* e, uint32_t* stack, uint32_t* leafbits, int f, int l, int depth) {
  // entries: bits 0..7 first, 8..15 last (last <= kSeqMax < 256), 16..31 depth budget
  int sp = 1;
  stack[0] = 0u | (static_cast<uint32_t>(l - f) << 8) | (static_cast<uint32_t>(depth) << 16);
  while (sp > 0) {
    const uint32_t ent = stack[--sp];
    int a = static_cast<int>(ent & 0xffu), b = static_cast<int>((ent >> 8) & 0xffu), d = static_cast<int>(ent >> 16);
    while (b - a > kLeaf) {
      if (d == 0) { heap_sort_segment(e, f + a, f + b); break; }
      --d;
      const int cut = seq_partition_pivot(e, f + a, f + b) - f;
      atomicOr(&leafbits[(f + cut) >> 5], 1u << ((f + cut) & 31));   // its leaves are insertion-sorted with all the others (phase 2)
      stack[sp++] = static_cast<uint32_t>(cut) | (static_cast<uint32_t>(b) << 8) | (static_cast<uint32_t>(d) << 16);
      b = cut;
    }
  }
}

template <bool MASKS>
__device__ __forceinline__ void introsort_phases(const QueryStore& S, int N, int depth0, const WgCtx& C, unsigned long long* st) {
  constexpr int CH = MASKS ? kChunkMasks : kChunk;      // positions per chunk task
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  uint32_t* e = S.elem;
  const int bw = (N + 31) / 32;
  int* stask = C.stask;
  int* sqcount = C.sqcount;
  for (int i = tid; i < bw; i += NT) { S.leafbits[i] = 0; S.smallbits[i] = 0; }
  if (tid == 0) { sqcount[0] = 0; sqcount[1] = 0; }
  __syncthreads();
  if (st && tid == 0) st[1] = __builtin_readcyclecounter();
  // ---- phase 1: introsort loop, breadth-first ------------------------------------------------------
  int depth = depth0;
  uint32_t* qcur = S.qa;
  uint32_t* qnxt = S.qb;
  int cur = 0, lvl = 0;
  if (tid == 0) {
    S.leafbits[0] = 1u;
    push_seg(S, qcur, &sqcount[0], 0, N, depth0);
  }
  __syncthreads();
  while (true) {
    const int nseg = sqcount[cur];
    if (nseg == 0) break;
    if (st && tid == 0 && lvl < 28) {
      st[8 + 2 * lvl] = __builtin_readcyclecounter();
      st[9 + 2 * lvl] = static_cast<unsigned long long>(nseg);
    }
    ++lvl;
    if (depth == 0) {                                       // depth budget exhausted -> heapsort each segment
      for (int si = tid; si < nseg; si += NT) heap_sort_segment(e, static_cast<int>(qcur[2 * si]), static_cast<int>(qcur[2 * si + 1]));
      break;
    }
    --depth;
    if (tid == 0) { sqcount[cur ^ 1] = 0; stask[0] = 0; stask[1] = 0; stask[2] = 0; }      // tasks, chunks, next task
    __syncthreads();
    // -- work list of the level: a segment of more than CH positions is cut into chunk tasks (its median moves to the front
    //    here), a shorter one is one task.  Tasks go to the waves round-robin, so one long segment no longer holds a level up.
    for (int si = tid; si < nseg; si += NT) {
      const int f = static_cast<int>(qcur[2 * si]), l = static_cast<int>(qcur[2 * si + 1]);
      const int nc = (l - f - 1 > CH) ? (l - f - 2 + CH) / CH : 1;
      const int base = atomicAdd(&stask[0], nc);
      int cbase = 0;
      if (nc > 1) { median_to_first(e, f, l); cbase = atomicAdd(&stask[1], nc); }
      S.seginfo[si] = static_cast<uint32_t>(cbase) | (static_cast<uint32_t>(nc) << 16);
      for (int j = 0; j < nc; ++j) S.task[base + j] = (static_cast<uint32_t>(si) << 11) | static_cast<uint32_t>(j);
    }
    __syncthreads();
    if (st && tid == 0 && lvl <= 16) st[64 + 8 * (lvl - 1) + 0] = __builtin_readcyclecounter();
    const int ntask = stask[0];
    const bool chunked = stask[1] != 0;
    uint32_t* cntL = S.cnt;
    uint32_t* cntR = S.cnt + S.ccap;
    uint32_t* cntS = S.cnt + 2 * S.ccap;
    // -- A: whole short segments; L / R counts of the chunks
    int guard = 0;
    while (true) {                                            // tasks differ in length (33..CH+1 positions): first come, first served
      int t = 0;
      if (lane == 0) t = atomicAdd(&stask[2], 1);
      t = __builtin_amdgcn_readfirstlane(t);
      if (t >= ntask) break;
      if (++guard > ntask) break;                            // a wave can never be handed more tasks than exist: bounds the loop whatever happens
      const uint32_t tk = uni(S.task[t]);
      const int si = static_cast<int>(tk >> 11), j = static_cast<int>(tk & 2047u);
      const int f = static_cast<int>(uni(qcur[2 * si])), l = static_cast<int>(uni(qcur[2 * si + 1]));
      const int n = l - f;
      const uint32_t info = uni(S.seginfo[si]);
      if ((info >> 16) == 1u) {
        const int cut = n <= 65 ? partition_wave<1>(S, f, l, lane, depth) : n <= 129 ? partition_wave<2>(S, f, l, lane, depth) : partition_wave<4>(S, f, l, lane, depth);
        if (lane == 0 && cut >= 0) {
          atomicOr(&S.leafbits[cut >> 5], 1u << (cut & 31));
          push_seg(S, qnxt, &sqcount[cur ^ 1], f, cut, depth);
          push_seg(S, qnxt, &sqcount[cur ^ 1], cut, l, depth);
        }
      } else {
        const int a = f + 1 + j * CH, b = a + CH < l ? a + CH : l;
        const int p = uni(ekey(e[f]));
        int cL = 0, cR = 0;
        uint32_t* mk = MASKS ? mask_area(S, f) + 4 + 4 * (j * (CH / 64)) : nullptr;       // this chunk's groups
        constexpr int UA = CMH_MAP_UNROLL_A;                   // groups in flight per wave (the pass is latency-bound)
        for (int x0 = a; x0 < b; x0 += 64 * UA) {
          int k[UA];
#pragma unroll
          for (int u = 0; u < UA; ++u) {
            const int x = x0 + 64 * u + lane;
            k[u] = x < b ? ekey(e[x]) : -1;
          }
#pragma unroll
          for (int u = 0; u < UA; ++u) {
            const uint64_t mL = __ballot(k[u] >= p), mR = __ballot(k[u] >= 0 && k[u] <= p);
            cL += __popcll(mL);
            cR += __popcll(mR);
            if (MASKS && lane == 0 && x0 + 64 * u < b)
              *reinterpret_cast<uint4*>(mk + 4 * (((x0 - a) >> 6) + u)) =
                  uint4{static_cast<uint32_t>(mL), static_cast<uint32_t>(mL >> 32), static_cast<uint32_t>(mR), static_cast<uint32_t>(mR >> 32)};
          }
        }
        const int ci = static_cast<int>(info & 0xffffu) + j;
        if (lane == 0) { cntL[ci] = static_cast<uint32_t>(cL); cntR[ci] = static_cast<uint32_t>(cR); }
      }
    }
    if (MASKS && chunked) {
      __syncthreads();
      if (st && tid == 0 && lvl <= 16) st[64 + 8 * (lvl - 1) + 1] = __builtin_readcyclecounter();
      // -- B: per chunked segment (a wave each), the chunk counts become exclusive prefixes - L ascending, R from the top - and the
      //    totals go to the mask area's header
      for (int si = wid; si < nseg; si += NWAVE) {
        const uint32_t info = uni(S.seginfo[si]);
        const int nc = static_cast<int>(info >> 16), base = static_cast<int>(info & 0xffffu);
        if (nc == 1) continue;
        const int f = static_cast<int>(uni(qcur[2 * si]));
        int carry = 0;
        for (int c0 = 0; c0 < nc; c0 += 64) {
          const int ci = c0 + lane;
          const int v = ci < nc ? static_cast<int>(cntL[base + ci]) : 0;
          int incl = v;
#pragma unroll
          for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(incl, o, 64);
            if (lane >= o) incl += up;
          }
          if (ci < nc) cntL[base + ci] = static_cast<uint32_t>(carry + incl - v);
          carry += __builtin_amdgcn_readlane(incl, 63);
        }
        const int Lc = carry;
        carry = 0;
        for (int c0 = ((nc - 1) / 64) * 64; c0 >= 0; c0 -= 64) {
          const int ci = c0 + lane;
          const int v = ci < nc ? static_cast<int>(cntR[base + ci]) : 0;
          int incl = v;                                         // inclusive SUFFIX sum over the block
#pragma unroll
          for (int o = 1; o < 64; o <<= 1) {
            const int dn = __shfl_down(incl, o, 64);
            if (lane + o < 64) incl += dn;
          }
          if (ci < nc) cntR[base + ci] = static_cast<uint32_t>(carry + incl - v);
          carry += __builtin_amdgcn_readlane(incl, 0);
        }
        // The number of pairs that swap, s = #{i : L_i < R_i} (a prefix of the ranks), BEFORE the swap pass, so that the pass cuts
        // exactly [0, s) into equal pieces (cut over all min(|L|, |R|) candidate ranks, only the pieces below s had work and the wave
        // holding two of them set the pace).  With a(x) = L entries below position x and b(x) = R entries at or above it,
        // s = max_x min(a(x), b(x)): pair i swaps iff x = R_i has a, b >= i + 1.  a rises and b falls with x, so the maximum sits where
        // they cross - located in three wave-wide steps: the chunk (from the two prefixes just written), the 64-position group inside
        // it (from the masks' popcounts), the position inside that (from the bits).  (A bisection on the rank - two selects per probe,
        // ~34 selects - cost 0.1-0.24 M cycles per level.)
        wave_sync();                                            // the prefixes above are read back by other lanes
        const int l = static_cast<int>(uni(qcur[2 * si + 1]));
        const int Rc = carry;
        int cstar = -1;
        for (int c0 = 0; c0 < nc; c0 += 64) {
          const int ci = c0 + lane;
          bool ok = false;
          if (ci < nc) {
            const int a = static_cast<int>(cntL[base + ci]);
            const int b = ci == 0 ? Rc : static_cast<int>(cntR[base + ci - 1]);
            ok = a <= b;
          }
          cstar += __popcll(__ballot(ok));
        }
        const int a0 = static_cast<int>(uni(cntL[base + cstar])), bend = static_cast<int>(uni(cntR[base + cstar]));
        constexpr int kGpc = CH / 64;
        const uint32_t* mk = mask_area(S, f) + 4;
        const int ngroups = (l - f - 1 + 63) >> 6;
        const int g = kGpc * cstar + lane;
        const bool gvalid = lane < kGpc && g < ngroups;
        uint32_t w0 = 0u, w1 = 0u, w2 = 0u, w3 = 0u;
        if (gvalid) { const uint4 q = *reinterpret_cast<const uint4*>(mk + 4 * g); w0 = q.x; w1 = q.y; w2 = q.z; w3 = q.w; }
        const int pl = __popc(w0) + __popc(w1), pr = __popc(w2) + __popc(w3);
        int ia = pl, ib = pr;                                    // inclusive prefix of pl, inclusive suffix of pr
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const int up = __shfl_up(ia, o, 64), dn = __shfl_down(ib, o, 64);
          if (lane >= o) ia += up;
          if (lane + o < 64) ib += dn;
        }
        const int ag = a0 + ia - pl, bg = bend + ib;             // a, b at the START of group g
        const int gstar = __popcll(__ballot(gvalid && ag <= bg)) - 1;      // (lane 0 always qualifies: a0 <= b at the chunk's start)
        const int aS = __builtin_amdgcn_readlane(ag, gstar);
        const int bE = __builtin_amdgcn_readlane(bg, gstar) - __builtin_amdgcn_readlane(pr, gstar);
        const uint64_t mL = (static_cast<uint64_t>(static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(w1), gstar))) << 32) |
                            static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(w0), gstar));
        const uint64_t mR = (static_cast<uint64_t>(static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(w3), gstar))) << 32) |
                            static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(w2), gstar));
        const int ax = aS + mbcnt64(mL);                         // L entries below this lane's position
        const int bx = bE + bits_above(mR, lane) + static_cast<int>((mR >> lane) & 1ull);
        int best = ax < bx ? ax : bx;
        const int aend = aS + __popcll(mL);                      // ... and the boundary behind the group
        best = lane == 63 ? (best > (aend < bE ? aend : bE) ? best : (aend < bE ? aend : bE)) : best;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const int other = __shfl_xor(best, o, 64); best = best > other ? best : other; }
        const int lo = best;
        if (lane == 0) { uint32_t* hdr = mask_area(S, f); hdr[0] = static_cast<uint32_t>(Lc); hdr[1] = static_cast<uint32_t>(carry); hdr[2] = static_cast<uint32_t>(lo); }
      }
      __syncthreads();
      if (st && tid == 0 && lvl <= 16) st[64 + 8 * (lvl - 1) + 2] = __builtin_readcyclecounter();
      // -- C: the s swapping pairs (L_i, R_i), cut into as many equal pieces of ranks as the segment has chunks; a piece's wave walks the
      //    masks from the piece's first rank and expands 64 ranks at a time
      constexpr int CB = CMH_MAP_SWAP_BLOCKS, CW = 64 * CB;         // ranks per iteration: CB gathers of each side in flight
      uint32_t* bufL = C.bufR + wid * 2 * CW;
      uint32_t* bufR = bufL + CW;
      for (int t = wid; t < ntask; t += NWAVE) {
        const uint32_t tk = uni(S.task[t]);
        const int si = static_cast<int>(tk >> 11), j = static_cast<int>(tk & 2047u);
        const uint32_t info = uni(S.seginfo[si]);
        const int nc = static_cast<int>(info >> 16), base = static_cast<int>(info & 0xffffu);
        if (nc == 1) continue;
        const int f = static_cast<int>(uni(qcur[2 * si])), l = static_cast<int>(uni(qcur[2 * si + 1]));
        const int npairs = static_cast<int>(uni(mask_area(S, f)[2]));             // exactly the pairs that swap
        const int piece = (((npairs + nc - 1) / nc) + 63) & ~63;
        const int i0 = j * piece, i1 = i0 + piece < npairs ? i0 + piece : npairs;
        if (i0 < i1) {
          MaskCursor cl = make_cursor(S, f, l, false), cr = make_cursor(S, f, l, true);
          cursor_seek<CH / 64>(cl, cntL + base, nc, i0, lane);
          cursor_seek<CH / 64>(cr, cntR + base, nc, i0, lane);
          for (int ib = i0; ib < i1; ib += CW) {
            const int nl = cursor_fill(cl, bufL, lane, CW), nr = cursor_fill(cr, bufR, lane, CW);
            wave_sync();
            uint32_t xl[CB], xr[CB], vl[CB], vr[CB];
            bool did[CB];
#pragma unroll
            for (int u = 0; u < CB; ++u) {
              const int k = 64 * u + lane;
              did[u] = false;
              if (ib + k < i1 && k < nl && k < nr) {
                xl[u] = bufL[k];
                xr[u] = bufR[k];
                did[u] = xl[u] < xr[u];
              }
            }
#pragma unroll
            for (int u = 0; u < CB; ++u)
              if (did[u]) { vl[u] = e[xl[u]]; vr[u] = e[xr[u]]; }
#pragma unroll
            for (int u = 0; u < CB; ++u)
              if (did[u]) { e[xl[u]] = vr[u]; e[xr[u]] = vl[u]; }
            wave_sync();
          }
        }
      }
      __syncthreads();
      if (st && tid == 0 && lvl <= 16) st[64 + 8 * (lvl - 1) + 3] = __builtin_readcyclecounter();
      // -- D: cut and children of the chunked segments (a wave each): cut = min(L_s, R_{s-1}) (they wait for C: a parked child's slot
      //    lies in the mask area)
      for (int si = wid; si < nseg; si += NWAVE) {
        const uint32_t info = uni(S.seginfo[si]);
        const int nc = static_cast<int>(info >> 16), base = static_cast<int>(info & 0xffffu);
        if (nc == 1) continue;
        const int f = static_cast<int>(uni(qcur[2 * si])), l = static_cast<int>(uni(qcur[2 * si + 1]));
        const int cap = (l - f) / 2 + 1;
        const uint32_t* hdr = mask_area(S, f);
        const int Lc = static_cast<int>(uni(hdr[0])), sw = static_cast<int>(uni(hdr[2]));
        const int lim = Lc < cap ? Lc : cap;
        const uint32_t c1 = sw < lim ? static_cast<uint32_t>(cursor_select<CH / 64>(S, f, l, false, cntL + base, nc, sw, lane)) : 0x7fffffffu;
        const uint32_t c2 = sw >= 1 ? static_cast<uint32_t>(cursor_select<CH / 64>(S, f, l, true, cntR + base, nc, sw - 1, lane)) : 0x7fffffffu;
        const int cut = static_cast<int>(c1 < c2 ? c1 : c2);
        if (lane == 0) {
          atomicOr(&S.leafbits[cut >> 5], 1u << (cut & 31));
          push_seg(S, qnxt, &sqcount[cur ^ 1], f, cut, depth);
          push_seg(S, qnxt, &sqcount[cur ^ 1], cut, l, depth);
        }
      }
    }
    if (!MASKS && chunked) {
      __syncthreads();
      if (st && tid == 0 && lvl <= 16) st[64 + 8 * (lvl - 1) + 1] = __builtin_readcyclecounter();
      // -- B: the chunks' positions go to the lists (L ascending from the chunks before, R descending from the chunks after)
      for (int t = wid; t < ntask; t += NWAVE) {
        const uint32_t tk = uni(S.task[t]);
        const int si = static_cast<int>(tk >> 11), j = static_cast<int>(tk & 2047u);
        const uint32_t info = uni(S.seginfo[si]);
        const int nc = static_cast<int>(info >> 16), base = static_cast<int>(info & 0xffffu);
        if (nc == 1) continue;
        const int f = static_cast<int>(uni(qcur[2 * si])), l = static_cast<int>(uni(qcur[2 * si + 1]));
        const int cap = (l - f) / 2 + 1;
        int before = 0, after = 0;
        for (int j0 = 0; j0 < nc; j0 += 64) {
          const int jj = j0 + lane;
          if (jj < j) before += static_cast<int>(cntL[base + jj]);
          if (jj > j && jj < nc) after += static_cast<int>(cntR[base + jj]);
        }
        const int offL = wave_sum_i(before), offR = wave_sum_i(after);
        const int ownR = static_cast<int>(cntR[base + j]);
        uint32_t* tL = S.tmp + tmp_base(f);
        uint32_t* tR = tL + cap;
        const int a = f + 1 + j * CH, b = a + CH < l ? a + CH : l;
        const int p = uni(ekey(e[f]));
        int runL = offL, runR = offR + ownR - 1;
        for (int x0 = a; x0 < b; x0 += 256) {
          int k[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int x = x0 + 64 * u + lane;
            k[u] = x < b ? ekey(e[x]) : -1;
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int x = x0 + 64 * u + lane;
            const bool isL = k[u] >= p, isR = k[u] >= 0 && k[u] <= p;
            const uint64_t mL = __ballot(isL), mR = __ballot(isR);
            const int rl = runL + mbcnt64(mL), rr = runR - mbcnt64(mR);
            if (isL && rl < cap) tL[rl] = static_cast<uint32_t>(x);
            if (isR && rr < cap) tR[rr] = static_cast<uint32_t>(x);
            runL += __popcll(mL);
            runR -= __popcll(mR);
          }
        }
      }
      __syncthreads();
      if (st && tid == 0 && lvl <= 16) st[64 + 8 * (lvl - 1) + 2] = __builtin_readcyclecounter();
      // -- C: the pairs, cut into as many pieces as the segment has chunks
      for (int t = wid; t < ntask; t += NWAVE) {
        const uint32_t tk = uni(S.task[t]);
        const int si = static_cast<int>(tk >> 11), j = static_cast<int>(tk & 2047u);
        const uint32_t info = uni(S.seginfo[si]);
        const int nc = static_cast<int>(info >> 16), base = static_cast<int>(info & 0xffffu);
        if (nc == 1) continue;
        const int f = static_cast<int>(uni(qcur[2 * si])), l = static_cast<int>(uni(qcur[2 * si + 1]));
        const int cap = (l - f) / 2 + 1;
        int sl = 0, sr = 0;
        for (int j0 = 0; j0 < nc; j0 += 64) {
          const int jj = j0 + lane;
          if (jj < nc) { sl += static_cast<int>(cntL[base + jj]); sr += static_cast<int>(cntR[base + jj]); }
        }
        const int Lc = wave_sum_i(sl), Rc = wave_sum_i(sr);
        int npairs = Lc < Rc ? Lc : Rc;
        npairs = npairs < cap ? npairs : cap;
        const int piece = (((cap + nc - 1) / nc) + 63) & ~63;
        const int i0 = j * piece, i1 = i0 + piece < npairs ? i0 + piece : npairs;
        const uint32_t* tL = S.tmp + tmp_base(f);
        const uint32_t* tR = tL + cap;
        int sw = 0;
        for (int ib = i0; ib < i1; ib += 64) {
          const int i = ib + lane;
          bool did = false;
          if (i < i1) {
            const uint32_t xl = tL[i], xr = tR[i];
            did = xl < xr;
            if (did) {
              const uint32_t t0 = e[xl];
              e[xl] = e[xr];
              e[xr] = t0;
            }
          }
          const int c = __popcll(__ballot(did));
          sw += c;
          if (c < 64) break;                                 // the swapping pairs are a prefix
        }
        if (lane == 0) cntS[base + j] = static_cast<uint32_t>(sw);
      }
      __syncthreads();
      if (st && tid == 0 && lvl <= 16) st[64 + 8 * (lvl - 1) + 3] = __builtin_readcyclecounter();
      // -- D: cut and children of the chunked segments
      for (int si = tid; si < nseg; si += NT) {
        const uint32_t info = S.seginfo[si];
        const int nc = static_cast<int>(info >> 16), base = static_cast<int>(info & 0xffffu);
        if (nc == 1) continue;
        const int f = static_cast<int>(qcur[2 * si]), l = static_cast<int>(qcur[2 * si + 1]);
        const int cap = (l - f) / 2 + 1;
        int sw = 0, Lc = 0;
        for (int j = 0; j < nc; ++j) { sw += static_cast<int>(cntS[base + j]); Lc += static_cast<int>(cntL[base + j]); }
        const uint32_t* tL = S.tmp + tmp_base(f);
        const uint32_t* tR = tL + cap;
        const int lim = Lc < cap ? Lc : cap;
        const uint32_t c1 = sw < lim ? tL[sw] : 0x7fffffffu;
        const uint32_t c2 = sw >= 1 ? tR[sw - 1] : 0x7fffffffu;
        const int cut = static_cast<int>(c1 < c2 ? c1 : c2);
        atomicOr(&S.leafbits[cut >> 5], 1u << (cut & 31));
        push_seg(S, qnxt, &sqcount[cur ^ 1], f, cut, depth);
        push_seg(S, qnxt, &sqcount[cur ^ 1], cut, l, depth);
      }
    }
    __syncthreads();
    if (st && tid == 0 && lvl <= 16) st[64 + 8 * (lvl - 1) + 4] = __builtin_readcyclecounter();
    if (st && tid == 0 && lvl <= 16) { st[64 + 8 * (lvl - 1) + 5] = ntask; st[64 + 8 * (lvl - 1) + 6] = stask[1]; }
    uint32_t* t = qcur; qcur = qnxt; qnxt = t;
    cur ^= 1;
  }
  __syncthreads();

  if (st && tid == 0) st[2] = __builtin_readcyclecounter();
  // ---- phase 1b: parked segments (17..kSeqMax elements): the rest of their introsort + insertion sort, one lane each
  for (int wi = tid; wi < bw; wi += NT) {
    uint32_t bitsw = S.smallbits[wi];
    while (bitsw) {
      const int f = wi * 32 + __ffs(bitsw) - 1;
      bitsw &= bitsw - 1;
      uint32_t* slot = S.tmp + tmp_base(f);
      const int l = static_cast<int>(slot[0]), d = static_cast<int>(slot[1]);
      seq_finish_segment(e, slot, S.leafbits, f, l, d);
    }
  }
  __syncthreads();

  if (st && tid == 0) st[3] = __builtin_readcyclecounter();
  // ---- phase 2: final insertion sort == stable sort of each <=16-element leaf ---------------------------
  // One thread per 16 positions; a leaf that starts there is loaded into registers (padded with keys above every real one) and
  // sorted by the insertion sort's own compare-exchange sequence on adjacent elements (a strict '<' swaps, so equal keys keep
  // their order): no dependent LDS round trips, and a wave whose leaves are all in order already skips the network.
  for (int hw = tid; hw < 2 * bw; hw += NT) {
    const uint32_t w0 = S.leafbits[hw >> 1];
    const uint32_t w1 = (hw >> 1) + 1 < bw ? S.leafbits[(hw >> 1) + 1] : 0u;
    const uint64_t both = static_cast<uint64_t>(w0) | (static_cast<uint64_t>(w1) << 32);
    uint32_t starts = (w0 >> (16 * (hw & 1))) & 0xffffu;
    while (__ballot(starts != 0)) {                            // wave-uniform trip count: the ballots below need every lane
      int s0 = 0, n = 0;
      if (starts) {
        const int bpos = 16 * (hw & 1) + __ffs(starts) - 1;
        starts &= starts - 1;
        s0 = (hw >> 1) * 32 + bpos;
        const uint64_t rest = bpos == 63 ? 0ull : both >> (bpos + 1);
        int d = rest ? __ffsll(static_cast<long long>(rest)) : 1 << 20;      // distance to the next segment start
        d = s0 + d > N ? N - s0 : d;
        n = d <= kLeaf ? d : 0;                                  // longer: heap-sorted or an all-equal run, in its final order already
      }
      uint32_t v[kLeaf];
#pragma unroll
      for (int i = 0; i < kLeaf; ++i) v[i] = i < n ? e[s0 + i] : 0xffffffffu;
      bool unsorted = false;
#pragma unroll
      for (int i = 1; i < kLeaf; ++i) unsorted |= (v[i] >> kIdxBits) < (v[i - 1] >> kIdxBits);
      if (__ballot(unsorted) == 0) continue;
#pragma unroll
      for (int i = 1; i < kLeaf; ++i) {
#pragma unroll
        for (int j = i; j >= 1; --j) {
          const uint32_t lo = v[j - 1], hi = v[j];
          const bool sw = (hi | kIdxMask) < (lo & ~kIdxMask);       // key(hi) < key(lo)
          v[j - 1] = sw ? hi : lo;
          v[j] = sw ? lo : hi;
        }
      }
      if (unsorted) {
#pragma unroll
        for (int i = 0; i < kLeaf; ++i)
          if (i < n) e[s0 + i] = v[i];
      }
    }
  }
  __syncthreads();
}

// ---- the per-query kernel ----------------------------------------------------------------------------------
struct MapArgs {
  const uint32_t *q_sign, *q_nz, *q_label, *r_sign, *r_nz, *r_label;
  int Q, N, bits, W, LW;      // W = code words, LW = label words
  long long topk;
  int depth_limit;            // < 0: 2*floor(log2 N)
  int stable;                 // CMH_TIE_STABLE: full sort of (key, index) instead of the introsort emulation
  float* ap;
  int32_t* perm;              // may be null
  const uint32_t* r_all_nz;   // device word, nonzero: every database code has all `bits` positions nonzero (+-1 codes) - r_nz is not read
  uint32_t* gstore;           // workspace slices (MODE 0: gridDim.x * store_words(N); MODE 2: gridDim.x * hybrid_glob_words(N))
  unsigned long long* stamps; // optional: 6 cycle stamps of (workgroup 0, first query) at the phase boundaries (diagnostics)
};

// MODE 0: everything in the workspace slice, 1: everything in LDS, 2: hybrid (above).  WAVES: waves per SIMD the variant is compiled
// for - 4 = one 1024-thread workgroup per CU (<= 128 VGPRs), 8 = two (64 VGPRs, a few spills)
template <int MODE, int WAVES>
__global__ __launch_bounds__(NT, WAVES) void map_query_kernel(MapArgs A) {
  extern __shared__ __attribute__((aligned(16))) uint32_t dyn_smem[];
  __shared__ uint32_t sq[3][kMaxWords];     // query planes: sign, nz, label
  __shared__ int stask[3];                  // tasks of the level; chunks among them; the next task to hand out
  __shared__ int sqcount[2];
  __shared__ int swork[NWAVE + 2];
  __shared__ double sred[NWAVE];
  __shared__ int shist[NWAVE * 64];         // CMH_TIE_STABLE: (wave, digit) cells of the radix passes; else: 64 L positions per wave

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int N = A.N, W = A.W, LW = A.LW;
  QueryStore S;
  if constexpr (MODE == 1) S = carve_store(dyn_smem, N);
  else if constexpr (MODE == 2) S = carve_hybrid(dyn_smem, A.gstore + static_cast<size_t>(blockIdx.x) * hybrid_glob_words(N), N);
  else S = carve_store(A.gstore + static_cast<size_t>(blockIdx.x) * store_words(N), N);
  uint32_t* e = S.elem;
  const int bw = (N + 31) / 32;
  // MODE 0: the dynamic LDS holds the mask walk's position buffers
  const WgCtx C{stask, sqcount, shist, MODE == 0 ? dyn_smem : nullptr};

  for (int qi = blockIdx.x; qi < A.Q; qi += gridDim.x) {
    __syncthreads();
    if (A.stamps && blockIdx.x == 0 && qi == 0 && tid == 0) A.stamps[0] = __builtin_readcyclecounter();
    // ---- phase 0: keys + relevance ---------------------------------------------------------------
    if (tid < W) { sq[0][tid] = A.q_sign[static_cast<size_t>(qi) * W + tid]; sq[1][tid] = A.q_nz[static_cast<size_t>(qi) * W + tid]; }
    if (tid < LW) sq[2][tid] = A.q_label[static_cast<size_t>(qi) * LW + tid];
    for (int i = tid; i < bw; i += NT) S.relbits[i] = 0;
    __syncthreads();
    int myrel = 0;
    const bool r_full = *A.r_all_nz != 0u;
    for (int j0 = 0; j0 < N; j0 += NT) {
      const int j = j0 + tid;
      bool rel = false;
      if (j < N) {
        int both = 0, diff = 0;
        const uint32_t* rs = A.r_sign + static_cast<size_t>(j) * W;
        if (r_full) {               // +-1 database codes: nz(r) is all ones on the code's bits, which nz(q) has already been cut to
          for (int w = 0; w < W; ++w) {
            const uint32_t nz = sq[1][w];
            both += __popc(nz);
            diff += __popc((sq[0][w] ^ rs[w]) & nz);
          }
        } else {
        const uint32_t* rn = A.r_nz + static_cast<size_t>(j) * W;
        for (int w = 0; w < W; ++w) {
          const uint32_t nz = sq[1][w] & rn[w];
          both += __popc(nz);
          diff += __popc((sq[0][w] ^ rs[w]) & nz);
        }
        }
        const int key = A.bits - (both - 2 * diff);        // = K - q.r, in [0, 2K]
        e[j] = (static_cast<uint32_t>(key) << kIdxBits) | static_cast<uint32_t>(j);
        const uint32_t* rl = A.r_label + static_cast<size_t>(j) * LW;
        uint32_t any = 0;
        for (int w = 0; w < LW; ++w) any |= sq[2][w] & rl[w];
        rel = any != 0;
      }
      const uint64_t m = __ballot(rel);
      if (lane == 0 && j0 + wid * 64 < N) {                  // 64 consecutive j per wave -> two bitmap words
        const int wbase = (j0 + wid * 64) >> 5;
        S.relbits[wbase] = static_cast<uint32_t>(m);
        if (wbase + 1 < bw) S.relbits[wbase + 1] = static_cast<uint32_t>(m >> 32);
      }
      myrel += rel ? 1 : 0;
    }
    myrel = wave_sum_i(myrel);
    if (lane == 0) swork[wid] = myrel;
    __syncthreads();
    int tsum = 0;
    for (int w = 0; w < NWAVE; ++w) tsum += swork[w];
    if (tsum == 0) {                                          // utils/calc_utils.py:27-29 `continue`
      if (tid == 0) A.ap[qi] = 0.f;
      if (A.perm)
        for (int j = tid; j < N; j += NT) A.perm[static_cast<size_t>(qi) * N + j] = -1;
      continue;
    }

    if (A.stable) {
      // ---- CMH_TIE_STABLE: ties by ascending database index == ascending order of the (unique) u32 elements
      // key << 19 | index.  LSD radix sort on the key alone, 6 bits per pass (2 passes for 64-bit codes), order-preserving:
      // wave w owns the contiguous chunk [ca, cb) of the current array; per 64-element group a lane finds the lanes with
      // its digit from 6 ballots (no waterfall), its destination is (exclusive prefix of its (digit, wave) cell) + (# equal
      // digits in lower lanes), and the lowest lane of each digit class bumps the cell.
      const int per_s = (((N + NWAVE - 1) / NWAVE) + 63) & ~63;
      int ca = wid * per_s; ca = ca < N ? ca : N;
      int cb = ca + per_s; cb = cb < N ? cb : N;
      const int keybits = 32 - __clz(2 * A.bits);
      uint32_t* src = e;
      uint32_t* dst = S.tmp;
      auto digit_class = [&](bool valid, int d) -> uint64_t {
        uint64_t m = __ballot(valid);
#pragma unroll
        for (int bit = 0; bit < 6; ++bit) {
          const bool one = (d >> bit) & 1;
          const uint64_t bb = __ballot(valid && one);
          m &= one ? bb : ~bb;
        }
        return m;
      };
      for (int shift = 0; shift < keybits; shift += 6) {
        shist[wid * 64 + lane] = 0;
        wave_sync();
        for (int x0 = ca; x0 < cb; x0 += 64) {
          const int x = x0 + lane;
          const bool valid = x < cb;
          const int d = valid ? (ekey(src[x]) >> shift) & 63 : 0;
          const uint64_t m = digit_class(valid, d);
          if (valid && (m & lanemask_lt(lane)) == 0) shist[wid * 64 + d] += __popcll(m);
          wave_sync();
        }
        __syncthreads();
        // exclusive scan of the 64 x 16 cells in (digit major, wave minor) order: thread t <-> (digit t/16, wave t%16)
        const int cell = (tid & (NWAVE - 1)) * 64 + (tid / NWAVE);
        const int v = shist[cell];
        int inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const int up = __shfl_up(inc, o, 64);
          if (lane >= o) inc += up;
        }
        if (lane == 63) swork[wid] = inc;
        __syncthreads();
        int wbase = 0;
        for (int w = 0; w < wid; ++w) wbase += swork[w];
        shist[cell] = wbase + inc - v;
        __syncthreads();
        for (int x0 = ca; x0 < cb; x0 += 64) {
          const int x = x0 + lane;
          const bool valid = x < cb;
          const uint32_t val = valid ? src[x] : 0u;
          const int d = valid ? (ekey(val) >> shift) & 63 : 0;
          const uint64_t m = digit_class(valid, d);
          const int base = valid ? shist[wid * 64 + d] : 0;
          wave_sync();
          if (valid) {
            dst[base + __popcll(m & lanemask_lt(lane))] = val;
            if ((m & lanemask_lt(lane)) == 0) shist[wid * 64 + d] = base + __popcll(m);
          }
          wave_sync();
        }
        __syncthreads();
        uint32_t* t = src; src = dst; dst = t;
      }
      if (src != e) {                      // odd number of passes: bring the result home
        for (int x = tid; x < N; x += NT) e[x] = src[x];
      }
      __syncthreads();
    } else {
    {
      const int depth0 = A.depth_limit >= 0 ? A.depth_limit : 2 * (31 - __clz(N));   // std::__lg(n) * 2
      unsigned long long* st = A.stamps && blockIdx.x == 0 && qi == 0 ? A.stamps : nullptr;
      introsort_phases<MODE == 0 && CMH_MAP_MASKS != 0>(S, N, depth0, C, st);
    }

    }
    if (A.stamps && blockIdx.x == 0 && qi == 0 && tid == 0) A.stamps[4] = __builtin_readcyclecounter();
    // ---- phase 3: AP = mean_{r<=total} r / position_r   (utils/calc_utils.py:33-37) -----------------------
    const long long total = A.topk > 0 && A.topk < tsum ? A.topk : tsum;
    const int per = (((N + NWAVE - 1) / NWAVE) + 63) & ~63;
    int a = wid * per; a = a < N ? a : N;
    int b = a + per; b = b < N ? b : N;
    int crel = 0;
    for (int x0 = a; x0 < b; x0 += 64) {
      const int x = x0 + lane;
      bool r = false;
      if (x < b) { const int id = eidx(e[x]); r = (S.relbits[id >> 5] >> (id & 31)) & 1u; }
      const uint64_t m = __ballot(r);
      crel += __popcll(m);
      // the relevance bits of these 64 positions wait in the (now free) list area for the second pass: it then reads 8 bytes per group
      // instead of the elements and their bitmap words again
      if (lane == 0) { S.tmp[2 * (x0 >> 6)] = static_cast<uint32_t>(m); S.tmp[2 * (x0 >> 6) + 1] = static_cast<uint32_t>(m >> 32); }
    }
    __syncthreads();
    if (lane == 0) swork[wid] = crel;
    __syncthreads();
    int off = 0;
    for (int w = 0; w < wid; ++w) off += swork[w];
    double acc = 0.0;
    int run = 0;
    for (int x0 = a; x0 < b; x0 += 64) {
      const int x = x0 + lane;
      int id = 0;
      if (A.perm && x < b) id = eidx(e[x]);
      const uint64_t m = static_cast<uint64_t>(uni(S.tmp[2 * (x0 >> 6)])) | (static_cast<uint64_t>(uni(S.tmp[2 * (x0 >> 6) + 1])) << 32);
      const bool r = (m >> lane) & 1ull;
      if (r) {
        const int rank = off + run + __popcll(m & lanemask_lt(lane)) + 1;
        if (rank <= total) acc += static_cast<double>(static_cast<float>(rank) / static_cast<float>(x + 1));
      }
      run += __popcll(m);
      if (A.perm && x < b) A.perm[static_cast<size_t>(qi) * N + x] = id;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) sred[wid] = acc;
    __syncthreads();
    if (tid == 0) {
      double t = 0.0;
      for (int w = 0; w < NWAVE; ++w) t += sred[w];
      A.ap[qi] = static_cast<float>(t / static_cast<double>(total));
      if (A.stamps && blockIdx.x == 0 && qi == 0) A.stamps[5] = __builtin_readcyclecounter();
    }
  }
}

// Does every database code have all of its `bits` positions nonzero?  (codes are sign() of real features: zeros essentially never
// occur, and then the nz plane - half of the 6 MB of codes a NUS-WIDE query streams in its key phase - need not be read at all.)
// flag: preset nonzero; any word that differs from the full mask clears it.
__global__ __launch_bounds__(256) void nz_all_ones_kernel(const uint32_t* __restrict__ nz, int64_t n, int bits, int W, uint32_t* flag) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n * W) return;
  const int w = static_cast<int>(i % W);
  const int cntb = bits - w * 32 < 32 ? bits - w * 32 : 32;
  const uint32_t full = cntb == 32 ? 0xffffffffu : (1u << cntb) - 1u;
  if (nz[i] != full) *flag = 0u;
}

// map = (((ap[0] + ap[1]) + ...) / Q) in f32, query order, like the reference's `map += AP` (:37-38)
__global__ __launch_bounds__(64) void map_mean_kernel(const float* __restrict__ ap, int Q, float* __restrict__ out) {
  __shared__ float buf[1024];
  float acc = 0.f;
  for (int q0 = 0; q0 < Q; q0 += 1024) {
    const int n = Q - q0 < 1024 ? Q - q0 : 1024;
    for (int i = threadIdx.x; i < n; i += 64) buf[i] = ap[q0 + i];
    __syncthreads();
    if (threadIdx.x == 0)
      for (int i = 0; i < n; ++i) acc += buf[i];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = acc / static_cast<float>(Q);
}

// ---- packing ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_codes_kernel(const float* __restrict__ codes, int64_t n, int bits, int W,
                                                         uint32_t* __restrict__ sp, uint32_t* __restrict__ np,
                                                         int32_t* __restrict__ bad) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n * W) return;
  const int64_t row = i / W;
  const int w = static_cast<int>(i - row * W);
  const float* c = codes + row * bits + w * 32;
  const int cntb = bits - w * 32 < 32 ? bits - w * 32 : 32;
  uint32_t s = 0, z = 0;
  bool isbad = false;
  for (int t = 0; t < cntb; ++t) {
    const float v = c[t];
    if (v == 1.f) { s |= 1u << t; z |= 1u << t; }
    else if (v == -1.f) z |= 1u << t;
    else if (v != 0.f) isbad = true;      // also catches NaN
  }
  sp[i] = s;
  np[i] = z;
  if (isbad) *bad = 1;
}

// the inverse of pack_codes_kernel: bit planes -> f32 codes in {-1, 0, +1}
__global__ __launch_bounds__(256) void unpack_codes_kernel(const uint32_t* __restrict__ sp, const uint32_t* __restrict__ np, int64_t n,
                                                           int bits, int W, float* __restrict__ codes) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n * bits) return;
  const int64_t row = i / bits;
  const int t = static_cast<int>(i - row * bits);
  const uint32_t s = sp[row * W + (t >> 5)] >> (t & 31) & 1u, z = np[row * W + (t >> 5)] >> (t & 31) & 1u;
  codes[i] = z ? (s ? 1.f : -1.f) : 0.f;
}

__global__ __launch_bounds__(256) void pack_labels_kernel(const float* __restrict__ lab, int64_t n, int classes, int LW,
                                                          uint32_t* __restrict__ out, int32_t* __restrict__ bad) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n * LW) return;
  const int64_t row = i / LW;
  const int w = static_cast<int>(i - row * LW);
  const float* c = lab + row * classes + w * 32;
  const int cntb = classes - w * 32 < 32 ? classes - w * 32 : 32;
  uint32_t s = 0;
  bool isbad = false;
  for (int t = 0; t < cntb; ++t) {
    const float v = c[t];
    if (v > 0.f) s |= 1u << t;
    else if (!(v == 0.f)) isbad = true;   // negative or NaN
  }
  out[i] = s;
  if (isbad) *bad = 1;
}

__global__ __launch_bounds__(256) void hamming_dist_kernel(const uint32_t* __restrict__ qs, const uint32_t* __restrict__ qn,
                                                           const uint32_t* __restrict__ rs, const uint32_t* __restrict__ rn,
                                                           int Q, int64_t N, int bits, int W, float* __restrict__ dist) {
  const int64_t j = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int q = blockIdx.y;
  if (j >= N) return;
  int both = 0, diff = 0;
  for (int w = 0; w < W; ++w) {
    const uint32_t nz = qn[static_cast<size_t>(q) * W + w] & rn[j * W + w];
    both += __popc(nz);
    diff += __popc((qs[static_cast<size_t>(q) * W + w] ^ rs[j * W + w]) & nz);
  }
  dist[static_cast<size_t>(q) * N + j] = 0.5f * static_cast<float>(bits - (both - 2 * diff));
}

__global__ __launch_bounds__(256) void neighbor_kernel(const uint32_t* __restrict__ la, const uint32_t* __restrict__ lb,
                                                       int A, int B, int LW, float* __restrict__ sim) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = blockIdx.y;
  if (j >= B) return;
  uint32_t any = 0;
  for (int w = 0; w < LW; ++w) any |= la[static_cast<size_t>(i) * LW + w] & lb[static_cast<size_t>(j) * LW + w];
  sim[static_cast<size_t>(i) * B + j] = any ? 1.f : 0.f;
}

static size_t lds_bytes_needed(int64_t N) { return store_words(N) * 4; }
constexpr size_t kLdsOne = 150 * 1024;   // one workgroup per CU: of 160 KiB; static __shared__ of the kernel takes ~9.2 KiB
constexpr size_t kLdsTwo = 74 * 1024;    // two workgroups per CU

// Where a query's working set lives: MAP_LDS2 everything in LDS, two queries per CU (N <= ~7.6 k); MAP_HYBRID elements in LDS, position
// lists in the workspace, two per CU (N <= ~17 k: MIRFlickr); MAP_HYBRID1 the same with one query per CU (N <= ~35 k); MAP_GLOBAL
// workspace, two per CU (COCO / NUS-WIDE); MAP_LDS1 (everything in LDS, one per CU: round 2's placement) only on request.
// CMH_MAP_MODE=lds1|hybrid|hybrid1|global (diagnostic) overrides where the size allows it.
enum MapMode { MAP_GLOBAL = 0, MAP_LDS1 = 1, MAP_HYBRID = 2, MAP_LDS2 = 3, MAP_HYBRID1 = 4 };
static MapMode map_mode(int64_t N) {
  static const char* forced = getenv("CMH_MAP_MODE");
  const size_t full = lds_bytes_needed(N), hyb = hybrid_lds_words(N) * 4;
  if (forced) {
    if (!strcmp(forced, "global")) return MAP_GLOBAL;
    if (!strcmp(forced, "lds1") && full <= kLdsOne) return MAP_LDS1;
    if (!strcmp(forced, "hybrid") && hyb <= kLdsTwo) return MAP_HYBRID;
    if (!strcmp(forced, "hybrid1") && hyb <= kLdsOne) return MAP_HYBRID1;
  }
  if (full <= kLdsTwo) return MAP_LDS2;
  if (hyb <= kLdsTwo) return MAP_HYBRID;
  if (hyb <= kLdsOne) return MAP_HYBRID1;     // 17 k < N <= 35 k: one query per CU, elements in LDS (5000 x 20 015: 4.6 ms against 5.9 from the workspace)
  return MAP_GLOBAL;
}

static int map_slots(int Q) { return Q < 1024 ? Q : 1024; }

}  // namespace cmh

using namespace cmh;

extern "C" int cmh_pack_codes(const float* codes, int64_t n, int32_t bits, uint32_t* sign_plane, uint32_t* nz_plane,
                              int32_t* bad_flag, void* stream) {
  CMH_CHECK_ARG(codes && sign_plane && nz_plane && bad_flag, "pack_codes: null pointer");
  CMH_CHECK_ARG(n > 0 && bits > 0 && bits <= 32 * kMaxWords, "pack_codes: n=%lld bits=%d", static_cast<long long>(n), bits);
  const int W = (bits + 31) / 32;
  const int64_t total = n * W;
  hipLaunchKernelGGL(pack_codes_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, as_stream(stream),
                     codes, n, bits, W, sign_plane, nz_plane, bad_flag);
  CMH_CHECK_LAUNCH("pack_codes");
  return CMH_OK;
}

extern "C" int cmh_unpack_codes(const uint32_t* sign_plane, const uint32_t* nz_plane, int64_t n, int32_t bits, float* codes, void* stream) {
  CMH_CHECK_ARG(sign_plane && nz_plane && codes, "unpack_codes: null pointer");
  CMH_CHECK_ARG(n > 0 && bits > 0 && bits <= 32 * kMaxWords, "unpack_codes: n=%lld bits=%d", static_cast<long long>(n), bits);
  const int64_t total = n * bits;
  hipLaunchKernelGGL(unpack_codes_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, as_stream(stream),
                     sign_plane, nz_plane, n, bits, (bits + 31) / 32, codes);
  CMH_CHECK_LAUNCH("unpack_codes");
  return CMH_OK;
}

extern "C" int cmh_map_mean(const float* ap, int32_t Q, float* map, void* stream) {
  CMH_CHECK_ARG(ap && map && Q > 0, "map_mean: bad arguments");
  hipLaunchKernelGGL(map_mean_kernel, dim3(1), dim3(64), 0, as_stream(stream), ap, Q, map);
  CMH_CHECK_LAUNCH("map_mean");
  return CMH_OK;
}

extern "C" int cmh_pack_labels(const float* labels, int64_t n, int32_t classes, uint32_t* packed, int32_t* bad_flag,
                               void* stream) {
  CMH_CHECK_ARG(labels && packed && bad_flag, "pack_labels: null pointer");
  CMH_CHECK_ARG(n > 0 && classes > 0 && classes <= 32 * kMaxWords, "pack_labels: n=%lld classes=%d",
                static_cast<long long>(n), classes);
  const int LW = (classes + 31) / 32;
  const int64_t total = n * LW;
  hipLaunchKernelGGL(pack_labels_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0,
                     as_stream(stream), labels, n, classes, LW, packed, bad_flag);
  CMH_CHECK_LAUNCH("pack_labels");
  return CMH_OK;
}

extern "C" int cmh_hamming_dist(const uint32_t* q_sign, const uint32_t* q_nz, const uint32_t* r_sign,
                                const uint32_t* r_nz, int32_t Q, int64_t N, int32_t bits, float* dist, void* stream) {
  CMH_CHECK_ARG(q_sign && q_nz && r_sign && r_nz && dist, "hamming_dist: null pointer");
  CMH_CHECK_ARG(Q > 0 && Q <= 65535 && N > 0 && bits > 0 && bits <= 32 * kMaxWords, "hamming_dist: Q=%d N=%lld bits=%d", Q,
                static_cast<long long>(N), bits);
  const int W = (bits + 31) / 32;
  hipLaunchKernelGGL(hamming_dist_kernel, dim3(static_cast<unsigned>((N + 255) / 256), Q), dim3(256), 0,
                     as_stream(stream), q_sign, q_nz, r_sign, r_nz, Q, N, bits, W, dist);
  CMH_CHECK_LAUNCH("hamming_dist");
  return CMH_OK;
}

extern "C" int cmh_calc_neighbor(const uint32_t* la, const uint32_t* lb, int32_t A, int32_t B, int32_t classes,
                                 float* sim, void* stream) {
  CMH_CHECK_ARG(la && lb && sim, "calc_neighbor: null pointer");
  CMH_CHECK_ARG(A > 0 && A <= 65535 && B > 0 && classes > 0 && classes <= 32 * kMaxWords, "calc_neighbor: bad shape");
  const int LW = (classes + 31) / 32;
  hipLaunchKernelGGL(neighbor_kernel, dim3((B + 255) / 256, A), dim3(256), 0, as_stream(stream), la, lb, A, B, LW, sim);
  CMH_CHECK_LAUNCH("calc_neighbor");
  return CMH_OK;
}

extern "C" size_t cmh_map_workspace_bytes(int32_t Q, int64_t N, int32_t bits, int32_t tie_order) {
  (void)bits; (void)tie_order;
  if (Q <= 0 || N <= 0) return 0;
  const size_t slots = static_cast<size_t>(map_slots(Q));
  const MapMode mode = map_mode(N);       // (the same decision cmh_hamming_map takes, CMH_MAP_MODE included)
  const size_t stamps = getenv("CMH_MAP_STAMPS") ? 32768 : 0;      // diagnostics (tools/map_stamps.py): behind the slices
  // (+ 64: the "database codes have no zeros" word, behind everything else)
  if (mode == MAP_GLOBAL) return slots * store_words(N) * 4 + 256 + stamps + 64;
  if (mode == MAP_HYBRID || mode == MAP_HYBRID1) return slots * hybrid_glob_words(N) * 4 + 256 + stamps + 64;
  return 4096 + 64;                       // all-LDS placements: only the optional diagnostics stamps live here
}

extern "C" int cmh_hamming_map(const uint32_t* q_sign, const uint32_t* q_nz, const uint32_t* q_label,
                               const uint32_t* r_sign, const uint32_t* r_nz, const uint32_t* r_label, int32_t Q,
                               int64_t N, int32_t bits, int32_t classes, int64_t topk, int32_t tie_order,
                               int32_t depth_limit_override, float* ap, float* map, int32_t* perm, void* workspace,
                               size_t workspace_bytes, void* stream) {
  CMH_CHECK_ARG(q_sign && q_nz && q_label && r_sign && r_nz && r_label && ap && map, "hamming_map: null pointer");
  CMH_CHECK_ARG(Q > 0 && N > 0, "hamming_map: Q=%d N=%lld", Q, static_cast<long long>(N));
  CMH_CHECK_ARG(N < (1ll << kIdxBits), "hamming_map: N=%lld exceeds %d", static_cast<long long>(N), (1 << kIdxBits) - 1);
  CMH_CHECK_ARG(bits > 0 && bits <= 32 * kMaxWords && 2 * bits < (1 << (32 - kIdxBits)), "hamming_map: bits=%d unsupported", bits);
  CMH_CHECK_ARG(classes > 0 && classes <= 32 * kMaxWords, "hamming_map: classes=%d unsupported", classes);
  CMH_CHECK_ARG(tie_order == CMH_TIE_REFERENCE || tie_order == CMH_TIE_STABLE, "hamming_map: bad tie_order %d", tie_order);
  const size_t need = cmh_map_workspace_bytes(Q, N, bits, tie_order);
  if (workspace_bytes < need || !workspace) return fail(CMH_ERR_WORKSPACE, "hamming_map: workspace %zu < %zu bytes", workspace_bytes, need);
  hipStream_t st = as_stream(stream);
  MapArgs a;
  a.q_sign = q_sign; a.q_nz = q_nz; a.q_label = q_label; a.r_sign = r_sign; a.r_nz = r_nz; a.r_label = r_label;
  a.Q = Q; a.N = static_cast<int>(N); a.bits = bits; a.W = (bits + 31) / 32; a.LW = (classes + 31) / 32;
  a.topk = topk; a.depth_limit = depth_limit_override; a.stable = tie_order == CMH_TIE_STABLE; a.ap = ap; a.perm = perm;
  a.gstore = reinterpret_cast<uint32_t*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~static_cast<uintptr_t>(255));
  const MapMode mode = map_mode(N);
  // diagnostics stamps (tools/map_stamps.py, CMH_MAP_MODE=lds1): only where the workspace holds nothing else
  {
    uint32_t* flag = reinterpret_cast<uint32_t*>((reinterpret_cast<uintptr_t>(workspace) + need - 64 + 3) & ~static_cast<uintptr_t>(3));
    const int64_t words = N * a.W;
    if (hipMemsetAsync(flag, 1, 4, st) != hipSuccess) return fail(CMH_ERR_LAUNCH, "hamming_map: memset failed");
    hipLaunchKernelGGL(nz_all_ones_kernel, dim3(static_cast<unsigned>((words + 255) / 256)), dim3(256), 0, st, r_nz, N, bits, a.W, flag);
    CMH_CHECK_LAUNCH("nz_all_ones");
    a.r_all_nz = flag;
  }
  a.stamps = nullptr;
  if (getenv("CMH_MAP_STAMPS")) {
    if (mode == MAP_LDS1 || mode == MAP_LDS2) a.stamps = reinterpret_cast<unsigned long long*>(a.gstore);
    else      // behind the workgroups' slices (cmh_map_workspace_bytes reserved the room)
      a.stamps = reinterpret_cast<unsigned long long*>(a.gstore + static_cast<size_t>(map_slots(Q)) *
                                                       (mode == MAP_GLOBAL ? store_words(N) : hybrid_glob_words(N)));
  }
  const size_t lds = mode == MAP_GLOBAL ? (CMH_MAP_MASKS ? NWAVE * 128 * CMH_MAP_SWAP_BLOCKS * 4 : 0) : (mode == MAP_HYBRID || mode == MAP_HYBRID1 ? hybrid_lds_words(N) * 4 : lds_bytes_needed(N));
#define MAP_GO(MODE, WAVES)                                                                                                      \
  do {                                                                                                                           \
    if (lds && hipFuncSetAttribute(reinterpret_cast<const void*>(map_query_kernel<MODE, WAVES>),                                 \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)) != hipSuccess)             \
      return fail(CMH_ERR_LAUNCH, "hamming_map: cannot reserve %zu bytes of LDS", lds);                                          \
    hipLaunchKernelGGL((map_query_kernel<MODE, WAVES>), dim3(map_slots(Q)), dim3(NT), lds, st, a);                               \
  } while (0)
  if (mode == MAP_LDS2) MAP_GO(1, 8);
  else if (mode == MAP_HYBRID) MAP_GO(2, 8);
  else if (mode == MAP_HYBRID1) MAP_GO(2, 4);
  else if (mode == MAP_LDS1) MAP_GO(1, 4);
  else MAP_GO(0, 8);        // (one 1024-thread workgroup per CU with 128 registers and no spills: COCO 32.2 -> 44.0 ms, NUS-WIDE 25.5 -> 33.1)
#undef MAP_GO
  CMH_CHECK_LAUNCH("hamming_map");
  hipLaunchKernelGGL(map_mean_kernel, dim3(1), dim3(64), 0, st, ap, Q, map);
  CMH_CHECK_LAUNCH("map_mean");
  return CMH_OK;
}
