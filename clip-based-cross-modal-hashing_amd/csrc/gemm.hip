// GEMM dispatch for the CLIP towers:  out[M,N] = epi( X[M,K] . W[N,K]^T )
//
// Replaces every nn.Linear / in_proj / conv1-as-GEMM of the reference's transformer blocks
// (model/base/model.py:171-196 ResidualAttentionBlock, :215 conv1, :250 proj, :370 text_projection).
//
// This file only CHOOSES the kernel and carries the measurement hook (cmh_prof_gemm_*):
//   N % 256 == 0 (every ViT-B/32 encoder GEMM)          -> gemm_wide_kernel   (gemm_wide.hip: 160/128/96 x 256 persistent tiles)
//   ... and M <= 2048 rows (pooled tail, projections)    -> gemm_rows_kernel   (gemm_rows.hip: 64 x 64 tiles, the same bits)
//   N % 128 == 0 only (test-sized towers, width 128/384) -> gemm_glds_kernel   (gemm_glds.hip: 128 x 128 LDS-DMA tiles)
// Round 1's register-staged 128 x 128 kernel (CMH_GEMM_IMPL=regstage) and round 3's two measured-slower experiments (the 256 x 256
// "big" tile, the LayerNorm fold inside the wide kernel) left the tree in round 4; DESIGN.md 4.3 keeps their numbers, git history
// (commit 20b80d8 and before) their code.
#include <cstdlib>
#include <cstring>
#include <vector>

#include "cmh_common.h"

#include <array>
#include <map>

namespace cmh {

constexpr int kTile = 128;             // N granule of the smallest tile any GEMM kernel of the library has

// ---- optional launch timing (bench.py roofline): HIP events around every GEMM launch on its own stream ----
struct GemmProf {
  bool on = false;
  std::vector<hipEvent_t> ev;     // pairs
  std::vector<double> flops;
  std::vector<std::array<int, 4>> dims;   // M, N, K, epi of each timed launch (CMH_GEMM_PROF_DUMP breakdown)
  std::vector<int> kind;                  // kernel of each timed launch: 0 gemm_wide_kernel, 1 gemm_rows_kernel, 2 the 128 x 128 fallbacks
  // device-side row counts (packed text): copied, asynchronously, into a pinned slot at launch time and turned into FLOPs by _end()
  // - the hook itself never waits for the stream, so a profiled region keeps the launch queue of an unprofiled one
  struct Pending { size_t launch; int slot; int Mub; double flops_per_row; };
  std::vector<Pending> pending;
  std::map<const int32_t*, int> slot_of;  // one copy per device word and session: every launch that names the word shares its slot
  int32_t* rows_pinned = nullptr;
  size_t rows_cap = 0, rows_used = 0;
  size_t used = 0;
};
static GemmProf g_prof;

void launch_gemm_glds(int dt, const void* A, const void* W, const float* bias, const float* residual, void* out,
                      int M, int N, int K, int epi, hipStream_t st);   // gemm_glds.hip
bool gemm_wide_supported(int N);                                         // gemm_wide.hip
void gemm_wide_time_next(hipEvent_t start, hipEvent_t stop);             // gemm_wide.hip: the next wide launch stamps these events itself
int launch_gemm_rows(int dt, const void* A, const void* W, const float* bias, const float* residual, void* out, int M, int N, int K,
                     int epi, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1);
int launch_gemm_rows_fp8(const void* A, const void* W, const float* colscale, float alpha, const float* bias, const float* residual,
                         void* out, float oscale, int M, int N, int K, int epi, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1);
int launch_gemm_wide(int dt, const void* A, const void* W, const float* bias, const float* residual, void* out,
                     int M, int N, int K, int epi, hipStream_t st, const float* colscale = nullptr, float alpha = 1.f,
                     float oscale = 1.f, const int32_t* m_dev = nullptr, int m_hint = -1);

// the measurement hook counts algorithmic FLOPs on REAL rows: FLOPs of `flops_per_row` x the rows of launch `launch` (the entry
// g_prof.flops[launch] is created by the caller with the upper bound's FLOPs and corrected by _end() once the count has arrived)
static void prof_rows_later(size_t launch, int Mub, const int32_t* m_dev, double flops_per_row, hipStream_t st) {
  if (!m_dev || !g_prof.rows_pinned) return;
  // (a copy per LAUNCH would put ~46 four-byte copies of 4-5 us each into every profiled step of the single-stream pair mode; the
  // word is read once per profiling session - a session measures repetitions of one batch - by the first launch that names it)
  auto it = g_prof.slot_of.find(m_dev);
  int slot;
  if (it != g_prof.slot_of.end()) {
    slot = it->second;
  } else {
    if (g_prof.rows_used >= g_prof.rows_cap) return;
    slot = static_cast<int>(g_prof.rows_used);
    g_prof.rows_pinned[slot] = -1;
    if (hipMemcpyAsync(g_prof.rows_pinned + slot, m_dev, 4, hipMemcpyDeviceToHost, st) != hipSuccess) return;
    ++g_prof.rows_used;
    g_prof.slot_of[m_dev] = slot;
  }
  g_prof.pending.push_back({launch, slot, Mub, flops_per_row});
}

// CMH_GEMM_WIDE=0 (A/B against round 1's 128 x 128 kernels): launches that do not need the wide kernel's epilogues avoid it
bool gemm_wide_enabled() {
  static const bool wide = []() { const char* e = getenv("CMH_GEMM_WIDE"); return !(e && !strcmp(e, "0")); }();
  return wide;
}

int launch_gemm(int dt, const void* A, const void* W, const float* bias, const float* residual, void* out,
                int M, int N, int K, int epi, hipStream_t st, const int32_t* m_dev, int m_hint) {
  const int bk = dt == CMH_F32 ? 32 : 64;
  CMH_CHECK_ARG(dt == CMH_F32 || dt == CMH_BF16, "gemm: bad dtype %d", dt);
  CMH_CHECK_ARG(M > 0 && N > 0 && K > 0, "gemm: empty problem M=%d N=%d K=%d", M, N, K);
  CMH_CHECK_ARG(N % kTile == 0, "gemm: N=%d must be a multiple of %d", N, kTile);
  CMH_CHECK_ARG(K % bk == 0, "gemm: K=%d must be a multiple of %d", K, bk);
  CMH_CHECK_ARG(!(epi & EPI_BIAS) || bias, "gemm: EPI_BIAS without bias");
  CMH_CHECK_ARG(!(epi & EPI_RESIDUAL) || residual, "gemm: EPI_RESIDUAL without residual");
  CMH_CHECK_ARG(!(epi & (EPI_RES_F16 | EPI_OUT_F16)) || (gemm_wide_supported(N) && !(epi & EPI_OUT_BF16)),
                "gemm: fp16 residual / output needs N %% 256 == 0 (N=%d) and excludes EPI_OUT_BF16", N);
  const bool timed = g_prof.on && g_prof.used + 2 <= g_prof.ev.size();
  static const bool wide = gemm_wide_enabled();
  // the wide kernel's launch stamps the event pair with its own begin / end (gemm_wide_time_next); the fallback kernels are
  // bracketed by two recorded events; CMH_GEMM_PROF_BRACKET=1 brackets every launch (round 1-2's method, for comparison)
  static const bool bracket = []() { const char* e = getenv("CMH_GEMM_PROF_BRACKET"); return e && e[0] == '1'; }();
  const bool takes_rows = wide && !m_dev && gemm_rows_takes(M, N, K, epi);   // (it reproduces the wide kernel's bits: off with it)
  const bool takes_wide = !takes_rows && ((wide && gemm_wide_supported(N)) || (epi & (EPI_RES_F16 | EPI_OUT_F16 | EPI_MUL_DQGELU | EPI_SAVE_PRE)));
  const bool self_timed = timed && (takes_wide || takes_rows) && !bracket;
  hipEvent_t ev0 = self_timed ? g_prof.ev[g_prof.used] : nullptr, ev1 = self_timed ? g_prof.ev[g_prof.used + 1] : nullptr;
  if (timed && !self_timed) (void)hipEventRecord(g_prof.ev[g_prof.used], st);
  CMH_CHECK_ARG(!(epi & EPI_MUL_DQGELU) || (residual && gemm_wide_supported(N) && !(epi & (EPI_RESIDUAL | EPI_OUT_F16))),
                "gemm: EPI_MUL_DQGELU needs aux in the residual slot, N %% 256 == 0 (N=%d), no residual / fp16 output", N);
  CMH_CHECK_ARG(!(epi & EPI_SAVE_PRE) || (residual && dt == CMH_BF16 && (epi & EPI_OUT_BF16) && gemm_wide_supported(N) &&
                                          !(epi & (EPI_RESIDUAL | EPI_MUL_DQGELU | EPI_OUT_F16))),
                "gemm: EPI_SAVE_PRE needs the second output in the residual slot, bf16 operands and output, N %% 256 == 0 (N=%d)", N);
  if (takes_rows) {
    const int rc = launch_gemm_rows(dt, A, W, bias, residual, out, M, N, K, epi, st, ev0, ev1);
    if (rc) return rc;
  } else if (takes_wide) {
    if (self_timed) gemm_wide_time_next(ev0, ev1);
    const int rc = launch_gemm_wide(dt, A, W, bias, residual, out, M, N, K, epi, st, nullptr, 1.f, 1.f, m_dev, m_hint);
    gemm_wide_time_next(nullptr, nullptr);
    if (rc) return rc;
  } else if (m_dev) {
    return fail(CMH_ERR_INVALID, "gemm: a device-side row count needs the wide kernel (N %% 256 == 0, N=%d)", N);
  } else {
    launch_gemm_glds(dt, A, W, bias, residual, out, M, N, K, epi, st);
  }
  if (timed) {
    if (!self_timed) (void)hipEventRecord(g_prof.ev[g_prof.used + 1], st);
    g_prof.flops.push_back(2.0 * M * static_cast<double>(N) * K);   // algorithmic FLOPs (real rows only: corrected by _end)
    prof_rows_later(g_prof.flops.size() - 1, M, m_dev, 2.0 * static_cast<double>(N) * K, st);
    g_prof.dims.push_back({M, N, K, epi});
    g_prof.kind.push_back(takes_rows ? 1 : (takes_wide ? 0 : 2));
    g_prof.used += 2;
  }
  CMH_CHECK_LAUNCH("gemm");
  return CMH_OK;
}

int launch_gemm_fp8(const void* A8, const void* W8, const float* colscale, float alpha, const float* bias, const float* residual,
                    void* out, float oscale, int M, int N, int K, int epi, hipStream_t st, const int32_t* m_dev, int m_hint) {
  CMH_CHECK_ARG(M > 0 && N > 0 && K > 0, "gemm_fp8: empty problem M=%d N=%d K=%d", M, N, K);
  CMH_CHECK_ARG(gemm_wide_supported(N) && K % 128 == 0, "gemm_fp8: N=%d must be a multiple of 256 and K=%d of 128", N, K);
  CMH_CHECK_ARG(A8 && W8 && out && colscale, "gemm_fp8: null pointer");
  CMH_CHECK_ARG(!(epi & EPI_BIAS) || bias, "gemm_fp8: EPI_BIAS without bias");
  CMH_CHECK_ARG(!(epi & EPI_RESIDUAL) || residual, "gemm_fp8: EPI_RESIDUAL without residual");
  CMH_CHECK_ARG(!(epi & EPI_MUL_DQGELU), "gemm_fp8: forward epilogues only");
  const int okinds = ((epi & EPI_OUT_BF16) ? 1 : 0) + ((epi & EPI_OUT_F16) ? 1 : 0) + ((epi & EPI_OUT_FP8) ? 1 : 0);
  CMH_CHECK_ARG(okinds <= 1, "gemm_fp8: one output type at a time");
  const bool timed = g_prof.on && g_prof.used + 2 <= g_prof.ev.size();
  const bool takes_rows = gemm_wide_enabled() && !m_dev && gemm_rows_takes(M, N, K, epi | EPI_SCALE);   // few rows: 64 x 64 tiles, the same bits
  if (takes_rows) {
    const int rc = launch_gemm_rows_fp8(A8, W8, colscale, alpha, bias, residual, out, oscale, M, N, K, epi | EPI_SCALE, st,
                                        timed ? g_prof.ev[g_prof.used] : nullptr, timed ? g_prof.ev[g_prof.used + 1] : nullptr);
    if (rc) return rc;
  } else {
    if (timed) gemm_wide_time_next(g_prof.ev[g_prof.used], g_prof.ev[g_prof.used + 1]);
    const int rc = launch_gemm_wide(CMH_FP8, A8, W8, bias, residual, out, M, N, K, epi | EPI_SCALE, st, colscale, alpha, oscale, m_dev, m_hint);
    gemm_wide_time_next(nullptr, nullptr);
    if (rc) return rc;
  }
  if (timed) {
    g_prof.flops.push_back(2.0 * M * static_cast<double>(N) * K);
    prof_rows_later(g_prof.flops.size() - 1, M, m_dev, 2.0 * static_cast<double>(N) * K, st);
    g_prof.dims.push_back({M, N, K, epi | EPI_SCALE});
    g_prof.kind.push_back(takes_rows ? 1 : 0);
    g_prof.used += 2;
  }
  CMH_CHECK_LAUNCH("gemm_fp8");
  return CMH_OK;
}

int launch_gemm_wide_grouped(int dt, const GemmProblem& a, const GemmProblem& b, int epi, hipStream_t st);   // gemm_wide.hip
bool gemm_wide_grouping_pays(int dt, const GemmProblem& a, const GemmProblem& b, int epi);                   // gemm_wide.hip: cost model

static int g_grouped = -1;   // cmh_set_gemm_grouped: -1 = environment (CMH_GEMM_GROUPED=0 switches grouping off)
bool gemm_grouping_enabled() {
  static const bool env_on = []() { const char* e = getenv("CMH_GEMM_GROUPED"); return !(e && !strcmp(e, "0")); }();
  return g_grouped < 0 ? env_on : g_grouped != 0;
}

// The same layer of both towers as ONE launch of the wide kernel when it can take both problems; two plain launches otherwise
// (identical results either way).  fp8: EPI_SCALE is implied, as in launch_gemm_fp8.
int launch_gemm_grouped(int dt, const GemmProblem& a, const GemmProblem& b, int epi, hipStream_t st) {
  const int bk = dt == CMH_F32 ? 32 : (dt == CMH_FP8 ? 128 : 64);
  auto fits = [&](const GemmProblem& g) {
    return g.M > 0 && g.K > 0 && gemm_wide_supported(g.N) && g.K % bk == 0 && (g.m_dev || !gemm_rows_takes(g.M, g.N, g.K, epi));
  };
  const bool fp8 = dt == CMH_FP8;
  const bool groupable = gemm_grouping_enabled() && gemm_wide_enabled() && fits(a) && fits(b) &&
                         !(epi & (EPI_MUL_DQGELU | EPI_SAVE_PRE)) &&
                         (fp8 ? (epi & (EPI_OUT_BF16 | EPI_OUT_F16 | EPI_OUT_FP8)) != 0
                              : (dt == CMH_BF16) == ((epi & (EPI_OUT_BF16 | EPI_OUT_F16)) != 0));
  if (!groupable || !gemm_wide_grouping_pays(dt, a.K >= b.K ? a : b, a.K >= b.K ? b : a, epi)) {
    for (const GemmProblem* g : {&a, &b}) {
      const int rc = fp8 ? launch_gemm_fp8(g->A, g->W, g->colscale, g->alpha, g->bias, g->residual, g->out, g->oscale, g->M, g->N, g->K,
                                           epi, st, g->m_dev, g->m_hint)
                         : launch_gemm(dt, g->A, g->W, g->bias, g->residual, g->out, g->M, g->N, g->K, epi, st, g->m_dev, g->m_hint);
      if (rc) return rc;
    }
    return CMH_OK;
  }
  for (const GemmProblem* g : {&a, &b}) {
    CMH_CHECK_ARG(g->A && g->W && g->out, "gemm (grouped): null pointer");
    CMH_CHECK_ARG(!(epi & EPI_BIAS) || g->bias, "gemm (grouped): EPI_BIAS without bias");
    CMH_CHECK_ARG(!(epi & EPI_RESIDUAL) || g->residual, "gemm (grouped): EPI_RESIDUAL without residual");
    CMH_CHECK_ARG(!fp8 || g->colscale, "gemm (grouped): fp8 operands without weight scales");
  }
  const bool timed = g_prof.on && g_prof.used + 2 <= g_prof.ev.size();
  if (timed) gemm_wide_time_next(g_prof.ev[g_prof.used], g_prof.ev[g_prof.used + 1]);
  // the longer K first: its tiles are the long jobs of the static schedule
  const bool a_first = a.K >= b.K;
  const int rc = launch_gemm_wide_grouped(dt, a_first ? a : b, a_first ? b : a, fp8 ? epi | EPI_SCALE : epi, st);
  gemm_wide_time_next(nullptr, nullptr);
  if (rc) return rc;
  if (timed) {
    g_prof.flops.push_back(2.0 * a.M * static_cast<double>(a.N) * a.K + 2.0 * b.M * static_cast<double>(b.N) * b.K);
    prof_rows_later(g_prof.flops.size() - 1, a.M, a.m_dev, 2.0 * static_cast<double>(a.N) * a.K, st);
    prof_rows_later(g_prof.flops.size() - 1, b.M, b.m_dev, 2.0 * static_cast<double>(b.N) * b.K, st);
    g_prof.dims.push_back({a.M + b.M, a.N + b.N, a.K, epi | (1 << 20)});   // (1 << 20: a grouped launch; rows / columns summed)
    g_prof.kind.push_back(0);
    g_prof.used += 2;
  }
  CMH_CHECK_LAUNCH("gemm (grouped)");
  return CMH_OK;
}

}  // namespace cmh

extern "C" int cmh_set_gemm_grouped(int32_t on) { cmh::g_grouped = on < 0 ? -1 : (on ? 1 : 0); return CMH_OK; }

extern "C" int cmh_prof_gemm_begin(int32_t max_launches) {
  using namespace cmh;
  CMH_CHECK_ARG(max_launches > 0 && max_launches <= (1 << 20), "prof_gemm_begin: bad max_launches");
  while (g_prof.ev.size() < static_cast<size_t>(max_launches) * 2) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return fail(CMH_ERR_LAUNCH, "prof_gemm_begin: hipEventCreate failed");
    g_prof.ev.push_back(e);
  }
  if (g_prof.rows_cap < static_cast<size_t>(max_launches) * 2) {      // (profiling only: the product path never allocates here)
    if (g_prof.rows_pinned) (void)hipHostFree(g_prof.rows_pinned);
    g_prof.rows_pinned = nullptr;
    g_prof.rows_cap = 0;
    if (hipHostMalloc(reinterpret_cast<void**>(&g_prof.rows_pinned), static_cast<size_t>(max_launches) * 2 * 4, hipHostMallocDefault) == hipSuccess)
      g_prof.rows_cap = static_cast<size_t>(max_launches) * 2;
    else
      g_prof.rows_pinned = nullptr;
  }
  g_prof.rows_used = 0;
  g_prof.pending.clear();
  g_prof.slot_of.clear();
  g_prof.used = 0;
  g_prof.flops.clear();
  g_prof.dims.clear();
  g_prof.kind.clear();
  g_prof.on = true;
  return CMH_OK;
}

extern "C" int cmh_prof_gemm_by_kernel(double* ms3, double* flops3, int64_t* launches3) {
  using namespace cmh;
  CMH_CHECK_ARG(ms3 && flops3 && launches3, "prof_gemm_by_kernel: null pointer");
  CMH_CHECK_ARG(!g_prof.on, "prof_gemm_by_kernel: call cmh_prof_gemm_end first");
  for (int k = 0; k < 3; ++k) { ms3[k] = 0.0; flops3[k] = 0.0; launches3[k] = 0; }
  for (size_t i = 0; i + 1 < g_prof.used; i += 2) {
    float t = 0.f;
    if (hipEventElapsedTime(&t, g_prof.ev[i], g_prof.ev[i + 1]) != hipSuccess)
      return fail(CMH_ERR_LAUNCH, "prof_gemm_by_kernel: hipEventElapsedTime failed");
    const int k = g_prof.kind[i / 2];
    ms3[k] += t; flops3[k] += g_prof.flops[i / 2]; launches3[k] += 1;
  }
  return CMH_OK;
}

extern "C" int cmh_prof_gemm_end(double* total_ms, double* total_flops, int64_t* launches) {
  using namespace cmh;
  CMH_CHECK_ARG(total_ms && total_flops && launches, "prof_gemm_end: null pointer");
  g_prof.on = false;
  if (hipDeviceSynchronize() != hipSuccess) return fail(CMH_ERR_LAUNCH, "prof_gemm_end: device synchronisation failed");   // (the row-count copies too)
  double ms = 0.0, fl = 0.0;
  for (size_t i = 0; i + 1 < g_prof.used; i += 2) {
    if (hipEventSynchronize(g_prof.ev[i + 1]) != hipSuccess) return fail(CMH_ERR_LAUNCH, "prof_gemm_end: event sync failed");
    float t = 0.f;
    if (hipEventElapsedTime(&t, g_prof.ev[i], g_prof.ev[i + 1]) != hipSuccess)
      return fail(CMH_ERR_LAUNCH, "prof_gemm_end: hipEventElapsedTime failed");
    ms += t;
  }
  // every event has completed, so has every row-count copy queued before it: real rows instead of the upper bounds
  for (const auto& p : g_prof.pending) {
    const int got = g_prof.rows_pinned[p.slot];
    if (got > 0 && got < p.Mub) {
      g_prof.flops[p.launch] -= static_cast<double>(p.Mub - got) * p.flops_per_row;
      g_prof.dims[p.launch][0] -= p.Mub - got;
    }
  }
  g_prof.pending.clear();
  for (size_t i = 0; i + 1 < g_prof.used; i += 2) fl += g_prof.flops[i / 2];
  if (getenv("CMH_GEMM_PROF_DUMP")) {   // per-shape breakdown of the timed launches, to stderr
    std::map<std::array<int, 4>, std::array<double, 3>> by;   // dims -> {ms, flops, launches}
    for (size_t i = 0; i + 1 < g_prof.used; i += 2) {
      float t = 0.f;
      (void)hipEventElapsedTime(&t, g_prof.ev[i], g_prof.ev[i + 1]);
      auto& e = by[g_prof.dims[i / 2]];
      e[0] += t; e[1] += g_prof.flops[i / 2]; e[2] += 1;
    }
    for (const auto& kv : by)
      fprintf(stderr, "gemm M=%6d N=%5d K=%5d epi=%3d  launches %5.0f  avg %8.2f us  %7.1f TF/s  share %5.1f%%\n", kv.first[0],
              kv.first[1], kv.first[2], kv.first[3], kv.second[2], kv.second[0] * 1e3 / kv.second[2],
              kv.second[1] / (kv.second[0] * 1e-3) / 1e12, 100.0 * kv.second[0] / ms);
  }
  *total_ms = ms;
  *total_flops = fl;
  *launches = static_cast<int64_t>(g_prof.used / 2);
  return CMH_OK;
}
