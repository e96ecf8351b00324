// extern "C" entry points of libinfgen_hip.so (see include/infgen_hip.h) and the per-step
// launch sequence.  No host synchronisation, no allocation: graph-capturable.
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <string>
#include <vector>
#include <atomic>
#include <mutex>
#include <unordered_map>
#include "kernels.h"
#include "../../include/infgen_hip.h"

using namespace ig;

static thread_local std::string g_err;

static int fail(const char* where, const char* what) {
  g_err = std::string(where) + ": " + what;
  return -1;
}
static int check_launch(const char* where) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(where, hipGetErrorString(e));
  return 0;
}
#define RET_IF(x) do { int _r = (x); if (_r) return _r; } while (0)

extern "C" const char* infgen_last_error(void) { return g_err.c_str(); }

// ---------------------------------------------------------------------------------- options
// Every switch that changes what a launch does lives in an InfgenOptions.  g_def holds the process-wide defaults the
// infgen_set_* functions edit (used by the operator-level entries and by contexts with opts.use == 0); a rollout-level entry
// called with a context whose opts.use != 0 installs that context's block for the duration of the call (thread-local), so
// contexts of one process do not see each other's settings.
namespace {
int layers_p_default() { const char* e = getenv("INFGEN_LAYERS_P"); const int v = e ? atoi(e) : 1; return v < 0 ? 0 : v > 2 ? 2 : v; }
InfgenOptions g_def = {0, /*attn_mode*/ 2, /*gemm_terms*/ 3, /*fourier_mode*/ 1, /*edge_fuse*/ 1, /*edge_loop*/ 6, /*overlap*/ 0,
                       /*row_group_margin*/ 0, /*layers_p*/ layers_p_default(), /*rhat_format*/ 0, /*edge_kernel*/ 1, 0, nullptr, nullptr};
int g_def_group_rows = 0;                  // rows of the layout the process-wide group list belongs to
const int* g_def_limit_n_agents = nullptr; // process-wide row limits (infgen_set_row_limits)
int g_def_limit_A_cap = 0;
struct Active { const InfgenOptions* o; int group_rows; const int* n_agents; int A_cap; };
thread_local Active tl_active = {nullptr, 0, nullptr, 0};
thread_local InfgenOptions tl_thread_opts;         // infgen_thread_options: the calling thread's block for operator-level entries
thread_local bool tl_thread_set = false;
inline const InfgenOptions& O() { return tl_active.o ? *tl_active.o : tl_thread_set ? tl_thread_opts : g_def; }
inline int group_rows() { return tl_active.o ? tl_active.group_rows : g_def_group_rows; }
inline const int* limit_n_agents() { return tl_active.o ? tl_active.n_agents : g_def_limit_n_agents; }
inline int limit_A_cap() { return tl_active.o ? tl_active.A_cap : g_def_limit_A_cap; }
struct OptScope {
  Active prev;
  explicit OptScope(const InfgenRollout* r) : prev(tl_active) {
    if (r && r->opts.use) tl_active = Active{&r->opts, r->S * r->A_cap, r->opts.row_groups ? r->n_agents : nullptr, r->A_cap};
  }
  ~OptScope() { tl_active = prev; }
};
}  // namespace

extern "C" int infgen_get_options(InfgenOptions* out) {
  if (!out) return fail("infgen_get_options", "null pointer");
  *out = g_def;
  return 0;
}

extern "C" int infgen_thread_options(const InfgenOptions* o) {
  if (o) { tl_thread_opts = *o; tl_thread_opts.row_groups = nullptr; tl_thread_opts.n_row_groups = nullptr; }
  tl_thread_set = o != nullptr;
  return 0;
}

extern "C" int infgen_get_effective_options(InfgenOptions* out) {
  if (!out) return fail("infgen_get_effective_options", "null pointer");
  *out = O();
  return 0;
}

// ---------------------------------------------------------------------------------- profiling
// Optional, process-global, off by default: HIP events recorded on the launch stream around the
// launches of the selected kernels (bench.py's roofline leg).  Not used by the product path.
namespace {
struct Prof {
  unsigned mask = 0;
  std::vector<hipEvent_t> e0, e1;
  std::vector<int> kid;
  std::vector<int> phase_of;                // 0: prologue / operator-level call, 1: inside a decode step (infgen_decode_step, infgen_rollout_run)
  std::vector<double> macs;                 // algorithmic multiply-accumulates of the launch (0: not a GEMM kernel)
  int stride = 1;                           // of the launches inside decode steps every stride-th is bracketed (infgen_prof_set_stride)
  int seen[INFGEN_KID_COUNT] = {};          // launches of the selected kernels since infgen_prof_enable, bracketed or not
  int seen_step[INFGEN_KID_COUNT] = {};     // ... of them inside decode steps
  size_t used = 0;
  unsigned long long* rows_dev = nullptr;   // [16] device counters: [n] rows processed by k_fourier with n input dims (n < 8);
                                            // [8 + kind] edges built by k_build_edges (kind 0 temporal, 1 map, 2 agent)
} g_prof;

static thread_local int t_prof_phase = 0;       // 1: the calling thread is inside a decode step (ProfPhase below)
static std::mutex g_prof_mu;                    // slot allocation and the launch counters (two host threads may drive two contexts)
struct ProfScope {
  int slot = -1;
  hipStream_t s;
  ProfScope(int kid, void* stream, double macs = 0.0) : s((hipStream_t)stream) {
    if ((g_prof.mask >> kid) & 1u) {
      std::lock_guard<std::mutex> lk(g_prof_mu);
      ++g_prof.seen[kid];
      // an event pair costs ~5 us of launch-stream time: inside the timed region only every stride-th decode-step launch carries one
      const bool take = t_prof_phase != 1 || (g_prof.seen_step[kid]++ % g_prof.stride) == 0;
      if (take && g_prof.used < g_prof.e0.size()) {
        slot = (int)g_prof.used++;
        g_prof.kid[slot] = kid;
        g_prof.phase_of[slot] = t_prof_phase;
        g_prof.macs[slot] = macs;
        (void)hipEventRecord(g_prof.e0[slot], s);
      }
    }
  }
  ~ProfScope() { if (slot >= 0) (void)hipEventRecord(g_prof.e1[slot], s); }
};
// launches issued inside a decode step are tagged (bench.py separates the step's edge launches from the prologue's)
// (per host thread: two threads driving two contexts save / restore their own tag - a shared one could be left at 1 by an
// interleaved save / restore pair, after which every later prologue launch counted as a step launch)
struct ProfPhase {
  int old;
  explicit ProfPhase(int p) : old(t_prof_phase) { t_prof_phase = p; }
  ~ProfPhase() { t_prof_phase = old; }
};
}  // namespace

extern "C" int infgen_prof_enable(unsigned mask, int max_launches) {
  if (mask && (int)g_prof.e0.size() < max_launches) {
    const size_t old = g_prof.e0.size();
    g_prof.e0.resize(max_launches); g_prof.e1.resize(max_launches); g_prof.kid.resize(max_launches);
    g_prof.phase_of.resize(max_launches);
    g_prof.macs.resize(max_launches);
    for (size_t i = old; i < (size_t)max_launches; ++i) {
      if (hipEventCreate(&g_prof.e0[i]) != hipSuccess || hipEventCreate(&g_prof.e1[i]) != hipSuccess)
        return fail("infgen_prof_enable", "hipEventCreate failed");
    }
  }
  if (mask && !g_prof.rows_dev) {
    if (hipMalloc(&g_prof.rows_dev, 16 * sizeof(unsigned long long)) != hipSuccess)
      return fail("infgen_prof_enable", "hipMalloc failed");
  }
  if (g_prof.rows_dev) (void)hipMemset(g_prof.rows_dev, 0, 16 * sizeof(unsigned long long));
  g_prof.mask = mask;
  g_prof.used = 0;
  g_prof.stride = 1;
  for (int k = 0; k < INFGEN_KID_COUNT; ++k) g_prof.seen[k] = g_prof.seen_step[k] = 0;
  return 0;
}
// after infgen_prof_enable: bracket only every stride-th launch of the selected kernels inside decode steps (launches outside
// them are all bracketed); infgen_prof_seen returns how many launches there were, bracketed or not
extern "C" int infgen_prof_set_stride(int stride) {
  if (stride < 1) return fail("infgen_prof_set_stride", "stride must be >= 1");
  g_prof.stride = stride;
  return 0;
}
extern "C" int infgen_prof_seen(int* seen, int* seen_step) {
  for (int k = 0; k < INFGEN_KID_COUNT; ++k) {
    if (seen) seen[k] = g_prof.seen[k];
    if (seen_step) seen_step[k] = g_prof.seen_step[k];
  }
  return 0;
}

// total_ms / calls: [INFGEN_KID_COUNT]; counters: [16] device-side row / edge counts (see Prof::rows_dev).  Synchronises.
static int prof_collect_impl(double* total_ms, int* calls, double* total_macs, unsigned long long* fourier_rows,
                             double* step_ms, int* step_calls) {
  for (int k = 0; k < INFGEN_KID_COUNT; ++k) {
    total_ms[k] = 0.0; calls[k] = 0; total_macs[k] = 0.0;
    if (step_ms) step_ms[k] = 0.0;
    if (step_calls) step_calls[k] = 0;
  }
  if (hipDeviceSynchronize() != hipSuccess) return fail("infgen_prof_collect", "sync failed");
  for (size_t i = 0; i < g_prof.used; ++i) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, g_prof.e0[i], g_prof.e1[i]) != hipSuccess)
      return fail("infgen_prof_collect", "hipEventElapsedTime failed");
    total_ms[g_prof.kid[i]] += ms;
    calls[g_prof.kid[i]] += 1;
    total_macs[g_prof.kid[i]] += g_prof.macs[i];
    if (g_prof.phase_of[i] == 1) {
      if (step_ms) step_ms[g_prof.kid[i]] += ms;
      if (step_calls) step_calls[g_prof.kid[i]] += 1;
    }
  }
  if (fourier_rows) {
    for (int i = 0; i < 16; ++i) fourier_rows[i] = 0;
    if (g_prof.rows_dev) (void)hipMemcpy(fourier_rows, g_prof.rows_dev, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  }
  g_prof.used = 0;
  return 0;
}
extern "C" int infgen_prof_collect(double* total_ms, int* calls, double* total_macs, unsigned long long* fourier_rows) {
  return prof_collect_impl(total_ms, calls, total_macs, fourier_rows, nullptr, nullptr);
}
extern "C" int infgen_prof_collect_steps(double* total_ms, int* calls, double* total_macs, unsigned long long* fourier_rows,
                                         double* step_ms, int* step_calls) {
  return prof_collect_impl(total_ms, calls, total_macs, fourier_rows, step_ms, step_calls);
}

extern "C" int infgen_layout_query(int what) {
  switch (what) {
    case INFGEN_Q_ATTN_PACK_SIZE: return AL_SIZE;
    case INFGEN_Q_FOURIER_PACK_SIZE_N2: return fourier_pack_size(2);
    case INFGEN_Q_FOURIER_PACK_SIZE_N3: return fourier_pack_size(3);
    case INFGEN_Q_FOURIER_PACK_SIZE_N4: return fourier_pack_size(4);
    case INFGEN_Q_TILE_ROWS: return TR;
    case INFGEN_Q_EDGE_ATTN_CAP: return 1 << 30;   /* single-pass kernel: no per-row edge cap */
    case INFGEN_Q_MAX_AGENTS: return 1024;
    case INFGEN_Q_ABI_VERSION: return 1;
    case INFGEN_Q_SIZEOF_ROLLOUT: return (int)sizeof(InfgenRollout);
    default: return -1;
  }
}

extern "C" int infgen_attn_pack_offset(const char* f) {
#define F(name, val) if (!strcmp(f, name)) return val;
  F("ln_src_g", AL_LN_SRC_G) F("ln_src_b", AL_LN_SRC_B) F("ln_dst_g", AL_LN_DST_G) F("ln_dst_b", AL_LN_DST_B)
  F("wq", AL_WQ) F("bq", AL_BQ) F("wk", AL_WK) F("wv", AL_WV) F("bv", AL_BV)
  F("wkr", AL_WKR) F("wvr", AL_WVR) F("bvr", AL_BVR) F("ws", AL_WS) F("bs", AL_BS)
  F("wg", AL_WG) F("bg", AL_BG) F("wo", AL_WO) F("bo", AL_BO)
  F("ln_post_g", AL_LN_POST_G) F("ln_post_b", AL_LN_POST_B)
  F("ln_ffpre_g", AL_LN_FFPRE_G) F("ln_ffpre_b", AL_LN_FFPRE_B)
  F("w1", AL_W1) F("b1", AL_B1) F("w2", AL_W2) F("b2", AL_B2)
  F("ln_ffpost_g", AL_LN_FFPOST_G) F("ln_ffpost_b", AL_LN_FFPOST_B)
  F("h_hdr", AH_HDR) F("h_pre", AH_PRE) F("h_post", AH_POST)
#undef F
  return -1;
}

extern "C" int infgen_fourier_pack_offset(const char* f, int n, int dim) {
  const int d0 = FE_DIM0 + dim * FD_SIZE;
  const int t0 = FE_DIM0 + n * FD_SIZE;
#define F(name, val) if (!strcmp(f, name)) return val;
  F("freq", FE_FREQ + dim * 64)
  F("w1", d0 + FD_W1) F("w1x", d0 + FD_W1X) F("b1", d0 + FD_B1) F("ln_g", d0 + FD_LN_G) F("ln_b", d0 + FD_LN_B)
  F("w2", d0 + FD_W2)
  F("b2sum", t0 + FT_B2SUM) F("lno_g", t0 + FT_LN_G) F("lno_b", t0 + FT_LN_B) F("w3", t0 + FT_W3) F("b3", t0 + FT_B3)
  // fp16-split section (FourierHLayout)
  const int h0 = fourier_pack_size_f32(n);
  const int hd = h0 + FH_DIM0 + dim * FHD_SIZE;
  F("h_hdr", h0 + FH_HDR) F("h_freq", h0 + FH_FREQ + dim * 64)
  F("h_wx", hd + FHD_WX) F("h_b1", hd + FHD_B1) F("h_g1", hd + FHD_G1) F("h_be1", hd + FHD_BE1)
  F("h_b2sum", h0 + FH_TAIL + FHT_B2SUM) F("h_g2", h0 + FH_TAIL + FHT_G2) F("h_be2", h0 + FH_TAIL + FHT_BE2)
  F("h_b3", h0 + FH_TAIL + FHT_B3)
  F("h_mat", h0 + FH_VEC_SIZE + dim * FH_HALF_MAT_FLOATS)      /* dim = half-matrix index here */
#undef F
  return -1;
}

// ---------------------------------------------------------------------------------- launchers
static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

extern "C" int infgen_linear(const float* X, int ldx, const int* gather, int rows, int K,
                             const float* Wp, int Np, const float* bias, int N,
                             const float* pre_g, const float* pre_b, const float* post_g, const float* post_b, int relu,
                             float* Y, int ldy, void* stream) {
  if (rows <= 0) return 0;
  if (Np % 32) return fail("infgen_linear", "Np must be a multiple of 32");
  if (K > 128 && Np > 128) return fail("infgen_linear", "K > 128 requires N <= 128");
  if ((pre_g && K != 128) || (post_g && (N != 128 || Np != 128)))
    return fail("infgen_linear", "LayerNorm prologue/epilogue needs K == 128 / N == 128");
  LinearArgs a{X, ldx, gather, rows, K, ((K + 7) / 8) * 8, Wp, Np, bias, N, pre_g, pre_b, post_g, post_b, relu, Y, ldy};
  { ProfScope _ps(INFGEN_KID_LINEAR, stream, (double)rows * K * N);
    const int tiles = ceil_div(rows, TR), passes = ceil_div(Np, 128);
    const int gy = (K <= 128 && !post_g && passes > 1 && tiles < 512) ? min(passes, ceil_div(512, tiles)) : 1;
    hipLaunchKernelGGL(k_linear, dim3(tiles, gy), dim3(NT), 0, (hipStream_t)stream, a); }
  return check_launch("infgen_linear");
}

// n (<= 6) independent infgen_linear calls in one launch; desc[i] mirrors infgen_linear's arguments
extern "C" int infgen_linear_multi(const InfgenLinearDesc* desc, int n, void* stream) {
  if (n <= 0) return 0;
  if (n > LINEAR_MULTI_MAX) return fail("infgen_linear_multi", "at most 6 descriptors");
  LinearMultiArgs m;
  int tiles = 0, gy = 1;
  double flops = 0.0;
  for (int i = 0; i < n; ++i) {
    const InfgenLinearDesc& d = desc[i];
    if (d.rows <= 0) return fail("infgen_linear_multi", "empty descriptor");
    if (d.Np % 32) return fail("infgen_linear_multi", "Np must be a multiple of 32");
    if (d.K > 128 && d.Np > 128) return fail("infgen_linear_multi", "K > 128 requires N <= 128");
    if ((d.pre_g && d.K != 128) || (d.post_g && (d.N != 128 || d.Np != 128)))
      return fail("infgen_linear_multi", "LayerNorm prologue/epilogue needs K == 128 / N == 128");
    m.d[i] = LinearArgs{d.X, d.ldx, d.gather, d.rows, d.K, ((d.K + 7) / 8) * 8, d.Wp, d.Np, d.bias, d.N, d.pre_g, d.pre_b,
                        d.post_g, d.post_b, d.relu, d.Y, d.ldy};
    tiles = tiles > ceil_div(d.rows, TR) ? tiles : ceil_div(d.rows, TR);
    const int passes = ceil_div(d.Np, 128);
    if (d.K <= 128 && !d.post_g && passes > gy) gy = passes;
    flops += (double)d.rows * d.K * d.N;
  }
  if (tiles * gy > 2048) gy = 2048 / tiles > 0 ? 2048 / tiles : 1;
  { ProfScope _ps(INFGEN_KID_LINEAR, stream, flops);
    hipLaunchKernelGGL(k_linear_multi, dim3(tiles, gy, n), dim3(NT), 0, (hipStream_t)stream, m); }
  return check_launch("infgen_linear_multi");
}

// counter calibration (tools/calibrate_fetch.sh): stream n_bytes with `width` (8 or 16) bytes per lane; out: [2048] floats
extern "C" int infgen_debug_stream_read(const float* p, unsigned long long n_bytes, int width, float* out, void* stream) {
  if (width != 8 && width != 16) return fail("infgen_debug_stream_read", "width must be 8 or 16");
  if (hipMemsetAsync(out, 0, 2048 * sizeof(float), (hipStream_t)stream) != hipSuccess) return fail("infgen_debug_stream_read", "memset failed");
  if (width == 8) hipLaunchKernelGGL(k_stream_read<8>, dim3(2048), dim3(256), 0, (hipStream_t)stream, p, (size_t)(n_bytes / 4), out);
  else hipLaunchKernelGGL(k_stream_read<16>, dim3(2048), dim3(256), 0, (hipStream_t)stream, p, (size_t)(n_bytes / 4), out);
  return check_launch("infgen_debug_stream_read");
}

extern "C" int infgen_layernorm(const float* X, int rows, const float* gamma, const float* beta, float* Y, void* stream) {
  if (rows <= 0) return 0;
  hipLaunchKernelGGL(k_layernorm, dim3(ceil_div(rows, TR)), dim3(NT), 0, (hipStream_t)stream, X, rows, gamma, beta, Y);
  return check_launch("infgen_layernorm");
}

// 3 (default): fp16 three-term split = fp32 accuracy; 1: only the hi x hi product of the same packed operands, i.e. plain fp16
// arithmetic (11 bits per operand) in the split kernels - the reduced-precision mode for BASELINE config C5 ("bf16"), which
// forfeits the 1e-3 logits bar.  Governs k_fourier_h, k_attn_h, k_mlpemb_h, k_heads_h.
extern "C" int infgen_set_gemm_terms(int terms) {
  if (terms < 1 || terms > 3) return fail("infgen_set_gemm_terms", "terms must be 3 (split, fp32 accuracy), 2 (bf16 operands) or 1 (fp16 operands)");
  g_def.gemm_terms = terms;
  return 0;
}

// fourier_mode 1: fp16 three-term split (k_fourier_h), 0: fp32-input MFMA (k_fourier)
extern "C" int infgen_set_fourier_mode(int mode) {
  if (mode != 0 && mode != 1) return fail("infgen_set_fourier_mode", "mode must be 0 (fp32 MFMA) or 1 (fp16 split)");
  g_def.fourier_mode = mode;
  return 0;
}

static int qs_dbg() {
  static const int v = getenv("INFGEN_QS_DBG") ? atoi(getenv("INFGEN_QS_DBG")) : 0;
  return v;
}

// out_r24: rows in the packed 24-bit format of kernels.h (k_fourier_h only; the rollout's private rhat buffers)
static int fourier_embed_impl(const float* raw, int n, const int* count_dev, int e_cap, const float* pack,
                              const float* cat, int ldcat, float* out, int ldo, int normalize, int out_r24, void* stream,
                              const float* dt_tab = nullptr, int dt_mode = 0) {
  if (e_cap <= 0) return 0;
  if (n < 1 || n > 4) return fail("infgen_fourier_embed", "n_dims must be in 1..4");
  if (out_r24 && O().fourier_mode == 0) return fail("infgen_fourier_embed", "packed rows need the split kernel");
  if (dt_mode && (O().fourier_mode == 0 || n < 2 || (dt_mode == 1 && !dt_tab)))
    return fail("infgen_fourier_embed", "the last-dim table needs the split kernel, n_dims >= 2 and a table");
  FourierArgs a{raw, n, count_dev, e_cap, pack, cat, ldcat, out, ldo, normalize,
                (g_prof.mask >> INFGEN_KID_FOURIER) & 1u && dt_mode != 2 ? g_prof.rows_dev : nullptr, out_r24, dt_tab, dt_mode, qs_dbg()};
  if (O().fourier_mode == 0) {
    int grid = ceil_div(e_cap, TR);
    if (grid > 2048) grid = 2048;
    ProfScope _ps(INFGEN_KID_FOURIER, stream);
    hipLaunchKernelGGL(k_fourier, dim3(grid), dim3(NT), 0, (hipStream_t)stream, a);
  } else {
    // large sets: the three-wave-group variant (fourier_h12.hip); INFGEN_FH12_MIN = smallest capacity that takes it (0: never)
    static const int fh12_min = getenv("INFGEN_FH12_MIN") ? atoi(getenv("INFGEN_FH12_MIN")) : 150000;
    const bool wide = fh12_min > 0 && e_cap >= fh12_min && O().gemm_terms == 3 && FH_WAVES == 8;
    int grid = ceil_div(e_cap, wide ? FH12_TILE : FH_TILE);     // 128-row tiles (8 waves x 16 rows), persistent
    if (grid > 256 * FH_WG_PER_CU) grid = 256 * FH_WG_PER_CU;          // one workgroup per CU (fourier_h.hip explains why)
    ProfScope _ps(INFGEN_KID_FOURIER, stream);
    if (wide) hipLaunchKernelGGL(k_fourier_h12<3>, dim3(grid), dim3(FH12_NT), 0, (hipStream_t)stream, a);
    else if (O().gemm_terms == 1) hipLaunchKernelGGL(k_fourier_h<1>, dim3(grid), dim3(FH_NT), 0, (hipStream_t)stream, a);
    else if (O().gemm_terms == 2) hipLaunchKernelGGL(k_fourier_h_b16<1>, dim3(grid), dim3(FH_NT), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(k_fourier_h<3>, dim3(grid), dim3(FH_NT), 0, (hipStream_t)stream, a);
  }
  return check_launch("infgen_fourier_embed");
}

extern "C" int infgen_fourier_embed(const float* raw, int n, const int* count_dev, int e_cap, const float* pack,
                                    const float* cat, int ldcat, float* out, int ldo, int normalize, void* stream) {
  return fourier_embed_impl(raw, n, count_dev, e_cap, pack, cat, ldcat, out, ldo, normalize, 0, stream);
}

// The last input dim of a FourierEmbedding as a lookup (include/infgen_hip.h): table of its branch at the values 0, -1, ..
extern "C" int infgen_fourier_last_dim_table(const float* pack, int n, float* table, void* stream) {
  return fourier_embed_impl(nullptr, n, nullptr, DT_TAB_ROWS, pack, nullptr, 0, table, 128, 0, 0, stream, nullptr, 2);
}
extern "C" int infgen_fourier_embed_tab(const float* raw, int n, const int* count_dev, int e_cap, const float* pack,
                                        const float* table, float* out, int ldo, int normalize, void* stream) {
  return fourier_embed_impl(raw, n, count_dev, e_cap, pack, nullptr, 0, out, ldo, normalize, 0, stream, table, 1);
}

// 0: fp32-input MFMA (k_attn_pre / k_attn_post), 1: fp16 three-term split on 64-row tiles (k_attn_h), 3: the same arithmetic on
// one 16-row group per eight-wave workgroup (k_attn_hs, low latency), 2 (default): by size - k_attn_h runs one or two workgroups
// per CU on 64-row tiles and wins from ~10 k rows (16 k rows: 99 vs 121 us for the fp32 kernels); below, a launch is one tile's
// dependency chain whatever the row count (k_attn_h 61-69 us, fp32 kernels 48-53 us), which k_attn_hs cuts to 27-32 us up to 4 k
// rows (one 16-row workgroup per CU; every workgroup pulls the layer's 1.2 MB of split weights through its CU's L2 port)
extern "C" int infgen_set_attn_mode(int mode) {
  if (mode < 0 || mode > 3) return fail("infgen_set_attn_mode", "mode must be 0 (fp32 MFMA), 1 (fp16 split), 2 (by size) or 3 (fp16 split, 16-row workgroups)");
  g_def.attn_mode = mode;
  return 0;
}
static int hs_max_rows() {
  static const int v = getenv("INFGEN_ATTN_HS_MAX") ? atoi(getenv("INFGEN_ATTN_HS_MAX")) : 6144;
  return v;
}
// the kernels of the other split families (k_heads_h, k_mlpemb_h) keep the by-size rule of the 64-row tiles
static inline bool attn_split(int rows) { return O().attn_mode == 1 || (O().attn_mode >= 2 && rows > 10240); }
// 0: fp32 kernels, 1: k_attn_h, 2: k_attn_hs
static inline int attn_kind(int rows) {
  switch (O().attn_mode) {
    case 0: return 0;
    case 1: return 1;
    case 3: return 2;
    default: return rows > 10240 ? 1 : rows > hs_max_rows() ? 0 : 2;      // (8 k rows: 60 us for k_attn_hs and the fp32 kernels alike)
  }
}

// 64-row tiles, 4 waves, two workgroups per CU (attn_h.hip); INFGEN_ATTN_WAVES=8 selects the 128-row variant
// optional list of the 16-row groups that hold agents (infgen_set_row_groups): applied to every split-kernel launch over
// exactly `g_group_rows` rows, i.e. the [S][A_cap] row arrays of the rollout the caller is running
extern "C" int infgen_set_row_groups(const int* groups, const int* n_groups, int rows) {
  g_def.row_groups = groups; g_def.n_row_groups = n_groups; g_def_group_rows = groups ? rows : 0;
  if (!groups) g_def_limit_n_agents = nullptr;
  return 0;
}

// the same information for the edge kernel (one wave per row): n_agents [S] of the [S][A_cap] layout and the margin the group
// list was built with; applies to launches over exactly the rows given to infgen_set_row_groups
extern "C" int infgen_set_row_limits(const int* n_agents, int A_cap, int margin) {
  g_def_limit_n_agents = n_agents; g_def_limit_A_cap = A_cap; g_def.row_group_margin = margin;
  return 0;
}

extern "C" int infgen_active_row_groups(const int* n_agents, int S, int A_cap, int margin, int* groups, int* n_groups,
                                        void* stream) {
  if (S <= 0 || A_cap <= 0) return fail("infgen_active_row_groups", "empty layout");
  ActiveGroupsArgs a{n_agents, S, A_cap, margin, groups, n_groups};
  hipLaunchKernelGGL(k_active_groups, dim3(1), dim3(1024), 0, (hipStream_t)stream, a);
  return check_launch("infgen_active_row_groups");
}

// The next small launch's warm workgroups (tile.cuh: WarmArgs): set by the sublayer loop right before the launch that carries them,
// taken (and cleared) by launch_attn_h / edge_fused_launch.  INFGEN_WARM=0 switches them off; launches of more than
// INFGEN_WARM_MAX_GROUPS 16-row groups (default 128) never carry any - they leave no CU idle.
static thread_local WarmArgs t_warm = {};
static int warm_max_groups() {
  static const int on = getenv("INFGEN_WARM") ? atoi(getenv("INFGEN_WARM")) : 1;
  static const int mx = getenv("INFGEN_WARM_MAX_GROUPS") ? atoi(getenv("INFGEN_WARM_MAX_GROUPS")) : 128;
  return on ? mx : 0;
}
static void warm_request(const void* p0, int len0, const void* p1, int len1) {
  t_warm.p[0] = (const char*)p0; t_warm.len[0] = p0 ? len0 : 0; t_warm.p[1] = (const char*)p1; t_warm.len[1] = p1 ? len1 : 0;
}
// -> grid with the warm workgroups appended (or the grid as it was)
static int warm_take(WarmArgs& w, int grid) {
  WarmArgs req = t_warm;
  t_warm = WarmArgs{};
  w = WarmArgs{};
  if (!(req.len[0] > 0 || req.len[1] > 0) || grid > warm_max_groups()) return grid;
  const int wg0 = (grid + 7) & ~7;
  const int per = (256 - wg0) / 8 < 24 ? (256 - wg0) / 8 : 24;
  if (per <= 0) return grid;
  w = req; w.wg0 = wg0; w.per_xcd = per;
  return wg0 + 8 * per;
}

static void launch_attn_h(const AttnHArgs& a_in, void* stream) {
  AttnHArgs a = a_in;
  a.dbg = qs_dbg();
  if (O().row_groups && a.rows == group_rows()) { a.groups = O().row_groups; a.n_groups = O().n_row_groups; }
  if (attn_kind(a.rows) == 2) {            // one 16-row group per workgroup
    const int grid = warm_take(a.warm, ceil_div(a.rows, 16));
    if (O().gemm_terms == 1) hipLaunchKernelGGL(k_attn_hs<1>, dim3(grid), dim3(512), 0, (hipStream_t)stream, a);
    else if (O().gemm_terms == 2) hipLaunchKernelGGL(k_attn_hs_b16<1>, dim3(grid), dim3(512), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(k_attn_hs<3>, dim3(grid), dim3(512), 0, (hipStream_t)stream, a);
    return;
  }
  t_warm = WarmArgs{};
  static const int waves = getenv("INFGEN_ATTN_WAVES") ? atoi(getenv("INFGEN_ATTN_WAVES")) : (IG_QSU ? 8 : 4);
  if (waves != 8) {
    int grid = ceil_div(a.rows, 64);
    if (grid > 512) grid = 512;          // two workgroups per CU, persistent over the 64-row tiles beyond that
    if (O().gemm_terms == 1) hipLaunchKernelGGL((k_attn_h<4, 1>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    else if (O().gemm_terms == 2) hipLaunchKernelGGL((k_attn_h_b16<4, 1>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((k_attn_h<4, 3>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  } else {
    int grid = ceil_div(a.rows, 128);
    if (grid > 256) grid = 256;
    hipLaunchKernelGGL((k_attn_h<8, 3>), dim3(grid), dim3(512), 0, (hipStream_t)stream, a);
  }
}

extern "C" int infgen_attn_pre(const float* X, int rows, const float* pack, int use_src_ln,
                               float* Q, float* U, float* K, float* V, void* stream) {
  if (rows <= 0) return 0;
  if (attn_kind(rows)) {
    AttnHArgs h{const_cast<float*>(X), rows, nullptr, nullptr, nullptr, nullptr, 0, pack, use_src_ln, Q, U, K, V};
    { ProfScope _ps(INFGEN_KID_ATTN_PRE, stream, (double)rows * 16384.0 * ((Q || U ? 1 : 0) + (K ? 1 : 0) + (V ? 1 : 0) + (U ? 1 : 0)));
      launch_attn_h(h, stream); }
    return check_launch("infgen_attn_pre");
  }
  AttnPreArgs a{X, rows, pack, use_src_ln, Q, U, K, V};
  { ProfScope _ps(INFGEN_KID_ATTN_PRE, stream, (double)rows * 16384.0 * ((Q || U ? 1 : 0) + (K ? 1 : 0) + (V ? 1 : 0) + (U ? 1 : 0)));
    hipLaunchKernelGGL(k_attn_pre, dim3(ceil_div(rows, TR)), dim3(NT), 0, (hipStream_t)stream, a); }
  return check_launch("infgen_attn_pre");
}

static int edge_attn_impl(int rows, const float* Q, const float* U, const float* Ksrc, const float* Vsrc,
                          const int* off, const int* cnt, const int* src, const float* rhat,
                          float* AGG, float* Z, float* SIG, int wide, void* stream, const int* row_mask = nullptr) {
  if (rows <= 0) return 0;
  EdgeAttnArgs a{rows, Q, U, Ksrc, Vsrc, EdgeSet{off, cnt, src, rhat}, AGG, Z, SIG, nullptr, nullptr, 0, 0, wide ? row_mask : nullptr};
  if (limit_n_agents() && O().row_groups && rows == group_rows() && !wide) {
    a.n_agents = limit_n_agents(); a.A_cap = limit_A_cap(); a.margin = O().row_group_margin;
  }
  { ProfScope _ps(INFGEN_KID_EDGE_ATTN, stream);
    if (wide) hipLaunchKernelGGL(k_edge_attn_wide, dim3(rows), dim3(512), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(k_edge_attn, dim3(ceil_div(rows, 4)), dim3(NT), 0, (hipStream_t)stream, a); }
  return check_launch("infgen_edge_attn");
}

// 1 (default): infgen_decode_layers runs the edge side of every sublayer with k_edge_fused - the absorbed query U and the
// positional aggregate Z of a 16-row tile stay on chip (LDS) and the node kernels run with has_pos = 0; 0: the unfused
// sequence with U / Z / SIG in HBM (kept for comparison and used below 257 rows, where k_edge_attn_wide splits long edge lists)
extern "C" int infgen_set_edge_fuse(int mode) {
  if (mode < 0 || mode > 2) return fail("infgen_set_edge_fuse", "mode must be 0 (off), 1 (from 257 rows) or 2 (always)");
  g_def.edge_fuse = mode;
  return 0;
}

// row format of the step's own rhat rows: 0 fp32 (default), 1 packed 24-bit (include/infgen_hip.h)
extern "C" int infgen_set_rhat_format(int format) {
  if (format != 0 && format != 1) return fail("infgen_set_rhat_format", "format must be 0 (fp32 rows) or 1 (packed 24-bit rows)");
  g_def.rhat_format = format;
  return 0;
}

// lane layout of the fused edge kernel's loop: 0 k_edge_fused, 1 k_edge_fused3 for launches beyond 4 k rows, 2 k_edge_fused3 always
extern "C" int infgen_set_edge_kernel(int kernel) {
  if (kernel < 0 || kernel > 2) return fail("infgen_set_edge_kernel", "kernel must be 0, 1 or 2");
  g_def.edge_kernel = kernel;
  return 0;
}

// edges per trip of k_edge_fused's edge loop (their K / V / rhat rows are requested together): 4, 6 (default) or 8
extern "C" int infgen_set_edge_loop(int v) {
  if (v != 4 && v != 6 && v != 8) return fail("infgen_set_edge_loop", "edges per trip must be 4, 6 or 8");
  g_def.edge_loop = v;
  return 0;
}

// resident workgroups per CU the runtime reports for k_edge_fused<6> (diagnostics; tools/edge_probe.sh)
extern "C" int infgen_edge_fused_occupancy(void) {
  int n = -1;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_edge_fused<6, true, 1, 8>, 512, 0) != hipSuccess) return -1;
  return n;
}

static int edge_fused_launch(int rows, const float* Q, const float* pack, const float* Ksrc, const float* Vsrc,
                             const int* off, const int* cnt, const int* src, const float* rhat, float* AGG,
                             int rows_per_scene, int kv_once, void* stream, int r24 = 0) {
  if (rows <= 0) return 0;
  if (!pack || !rhat) return fail("infgen_edge_attn_fused", "needs the layer pack and rhat");
  static const int dbg = getenv("INFGEN_EDGE_DBG") ? atoi(getenv("INFGEN_EDGE_DBG")) : 0;
  static const int no_xcd = getenv("INFGEN_EDGE_NOXCD") ? atoi(getenv("INFGEN_EDGE_NOXCD")) : 0;
  static const int small_max = getenv("INFGEN_EDGE_SMALL") ? atoi(getenv("INFGEN_EDGE_SMALL")) : 4096;
  static const int wg2 = getenv("INFGEN_EDGE_WG2") ? atoi(getenv("INFGEN_EDGE_WG2")) : 1;
  const int G = O().edge_loop;
  // small launches: one 16-row group per workgroup (edge_fused.hip), twice the workgroups for the same rows
  const bool small = rows <= small_max && G == 6;
  // large launches: 8-wave workgroups of one 16-row group, two per CU - one's matrix phases run under the other's edge loop
  // (128 vs 140 us per launch at 512 scenes; INFGEN_EDGE_WG2=0: one 16-wave workgroup of two groups per CU)
  const bool two = !small && wg2 && G == 6;
  const int tr = (small || two) ? 16 : 32;          // rows per tile
  EdgeFusedArgs a{rows, Q, pack, Ksrc, Vsrc, EdgeSet{off, cnt, src, rhat}, AGG, nullptr, nullptr, dbg, 0, kv_once};
  int grid = ceil_div(rows, tr);           // with a group list at most that many
  if (O().row_groups && rows == group_rows()) { a.groups = O().row_groups; a.n_groups = O().n_row_groups; }
  else if (rows_per_scene > tr && rows_per_scene % tr == 0 && !(no_xcd & 1)) {
    a.tiles_per_scene = rows_per_scene / tr;
    const int grp = 8 * a.tiles_per_scene;
    grid = ceil_div(grid, grp) * grp;
  }
  if (no_xcd & 2) a.kv_once = 0;
  if (!r24 && (O().edge_kernel == 2 || (O().edge_kernel == 1 && !small))) {
    // k_edge_fused3 (edge_fused3.hip): one 8-wave workgroup per 16-row group at every size, fp32 rhat rows
    EdgeFusedArgs m = a;
    int mg = ceil_div(rows, 16);
    m.tiles_per_scene = 0;
    if (!m.groups && rows_per_scene > 16 && rows_per_scene % 16 == 0 && !(no_xcd & 1)) {
      m.tiles_per_scene = rows_per_scene / 16;
      const int grp = 8 * m.tiles_per_scene;
      mg = ceil_div(mg, grp) * grp;
    }
    m.n_virtual = mg;
    t_warm = WarmArgs{};
    { ProfScope _ps(INFGEN_KID_EDGE_ATTN, stream);
      auto k3 = G == 4 ? k_edge_fused3<4> : G == 8 ? k_edge_fused3<8> : k_edge_fused3<6>;
      // (INFGEN_EDGE_LDS_PAD=<bytes>, experiment: dynamic LDS on top of the kernel's 75 KB - from ~6 KB on only ONE workgroup fits a
      // CU, which leaves room for another stream's node kernel; profiles/r06_coresidency_ab.txt)
      static const int lds_pad = getenv("INFGEN_EDGE_LDS_PAD") ? atoi(getenv("INFGEN_EDGE_LDS_PAD")) : 0;
      hipLaunchKernelGGL(k3, dim3(mg), dim3(512), lds_pad, (hipStream_t)stream, m); }
    return check_launch("infgen_edge_attn_fused(k_edge_fused3)");
  }
  a.n_virtual = grid;
  if (small) grid = warm_take(a.warm, grid); else t_warm = WarmArgs{};
  static unsigned long long* ef_trace = nullptr;       // (timing experiment: INFGEN_EDGE_DBG bit 7 with an -DIG_EF_TRACE=1 build)
  if (dbg & 128) {
    if (!ef_trace && hipMalloc(&ef_trace, 64 * sizeof(unsigned long long)) != hipSuccess) return fail("infgen_edge_attn_fused", "trace buffer");
    (void)hipMemsetAsync(ef_trace, 0, 64 * sizeof(unsigned long long), (hipStream_t)stream);
    a.dbgbuf = reinterpret_cast<unsigned*>(ef_trace);
  }
  { ProfScope _ps(INFGEN_KID_EDGE_ATTN, stream);
    auto kern = small ? (r24 ? k_edge_fused<6, true, 1, 16> : k_edge_fused<6, false, 1, 16>)
              : two ? (r24 ? k_edge_fused<6, true, 1, 8> : k_edge_fused<6, false, 1, 8>)
              : r24 ? (G == 4 ? k_edge_fused<4, true, 2, 16> : G == 8 ? k_edge_fused<8, true, 2, 16> : k_edge_fused<6, true, 2, 16>)
                    : (G == 4 ? k_edge_fused<4, false, 2, 16> : G == 8 ? k_edge_fused<8, false, 2, 16> : k_edge_fused<6, false, 2, 16>);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(two ? 512 : 1024), 0, (hipStream_t)stream, a); }
  if (dbg & 128) {
    static int dumps = 0;
    if (dumps++ < 3) {
      unsigned long long hst[64];
      (void)hipStreamSynchronize((hipStream_t)stream);
      (void)hipMemcpy(hst, ef_trace, sizeof(hst), hipMemcpyDeviceToHost);
      for (int i = 1; i < 8; ++i) fprintf(stderr, "[ef trace] %d +%lld\n", i, hst[i] && hst[0] ? (long long)(hst[i] - hst[0]) : -1ll);
    }
  }
  return check_launch("infgen_edge_attn_fused");
}

extern "C" int infgen_edge_attn_fused(int rows, const float* Q, const float* pack, const float* Ksrc, const float* Vsrc,
                                      const int* off, const int* cnt, const int* src, const float* rhat,
                                      float* AGG, void* stream) {
  return edge_fused_launch(rows, Q, pack, Ksrc, Vsrc, off, cnt, src, rhat, AGG, 0, 0, stream);
}

// the same pair of operators with the rows of rhat in the packed 24-bit form (include/infgen_hip.h): what infgen_decode_layers does
// for its own edge sets, for callers that own the buffer between the two calls (the map encoder of infgen_amd/engine.py)
extern "C" int infgen_fourier_embed_r24(const float* raw, int n, const int* count_dev, int e_cap, const float* pack, void* out,
                                        void* stream) {
  return fourier_embed_impl(raw, n, count_dev, e_cap, pack, nullptr, 0, static_cast<float*>(out), 128, 1, 1, stream);
}

extern "C" int infgen_edge_attn_fused_r24(int rows, const float* Q, const float* pack, const float* Ksrc, const float* Vsrc,
                                          const int* off, const int* cnt, const int* src, const void* rhat24,
                                          float* AGG, void* stream) {
  return edge_fused_launch(rows, Q, pack, Ksrc, Vsrc, off, cnt, src, static_cast<const float*>(rhat24), AGG, 0, 0, stream, 1);
}

// one wave per destination; few destinations (<= 256 rows) get the 8-wave split so that the chip is not idle
extern "C" int infgen_edge_attn(int rows, const float* Q, const float* U, const float* Ksrc, const float* Vsrc,
                                const int* off, const int* cnt, const int* src, const float* rhat,
                                float* AGG, float* Z, float* SIG, void* stream) {
  return edge_attn_impl(rows, Q, U, Ksrc, Vsrc, off, cnt, src, rhat, AGG, Z, SIG, rows <= 256 ? 1 : 0, stream);
}

extern "C" int infgen_edge_attn_mode(int rows, const float* Q, const float* U, const float* Ksrc, const float* Vsrc,
                                     const int* off, const int* cnt, const int* src, const float* rhat,
                                     float* AGG, float* Z, float* SIG, int wide, void* stream) {
  return edge_attn_impl(rows, Q, U, Ksrc, Vsrc, off, cnt, src, rhat, AGG, Z, SIG, wide, stream);
}

static int attn_post_fused(float* X, int rows, const float* pack, const float* AGG, const float* Z, const float* SIG,
                           int has_pos, const float* next_pack, float* nQ, float* nU, float* nK, float* nV, void* stream);

extern "C" int infgen_attn_post(float* X, int rows, const float* pack, const float* AGG, const float* Z,
                                const float* SIG, int has_pos, void* stream) {
  return attn_post_fused(X, rows, pack, AGG, Z, SIG, has_pos, nullptr, nullptr, nullptr, nullptr, nullptr, stream);
}

extern "C" int infgen_attn_post_pre(float* X, int rows, const float* pack, const float* AGG, const float* Z,
                                    const float* SIG, int has_pos, const float* next_pack, float* nQ, float* nU,
                                    float* nK, float* nV, void* stream) {
  return attn_post_fused(X, rows, pack, AGG, Z, SIG, has_pos, next_pack, nQ, nU, nK, nV, stream);
}

static int attn_post_fused(float* X, int rows, const float* pack, const float* AGG, const float* Z, const float* SIG,
                           int has_pos, const float* next_pack, float* nQ, float* nU, float* nK, float* nV, void* stream) {
  if (rows <= 0) return 0;
  if (attn_kind(rows)) {
    AttnHArgs h{X, rows, pack, AGG, Z, SIG, has_pos, next_pack, 0, nQ, nU, nK, nV};
    { ProfScope _ps(INFGEN_KID_ATTN_POST, stream, (double)rows * (196608.0 + (has_pos ? 16384.0 : 0.0) +
          (next_pack ? 16384.0 * ((nQ || nU ? 1 : 0) + (nK ? 1 : 0) + (nV ? 1 : 0) + (nU ? 1 : 0)) : 0.0)));
      launch_attn_h(h, stream); }
    return check_launch("infgen_attn_post");
  }
  AttnPostArgs a{X, rows, pack, AGG, Z, SIG, has_pos, next_pack, nQ, nU, nK, nV};
  { ProfScope _ps(INFGEN_KID_ATTN_POST, stream, (double)rows * (196608.0 + (has_pos ? 16384.0 : 0.0) +
        (next_pack ? 16384.0 * ((nQ || nU ? 1 : 0) + (nK ? 1 : 0) + (nV ? 1 : 0) + (nU ? 1 : 0)) : 0.0)));
    hipLaunchKernelGGL(k_attn_post, dim3(ceil_div(rows, TR)), dim3(NT), 0, (hipStream_t)stream, a); }
  return check_launch("infgen_attn_post");
}

extern "C" int infgen_match_agent_tokens(const unsigned char* valid, const float* pos, const float* heading, const float* shape,
                                        const int* type, const float* tok, long long tok_agent_stride, int A, int T, int shift,
                                        int n_token, int* token_index, float* token_contour, void* stream) {
  if (A <= 0) return 0;
  if (shift <= 0 || T <= shift) return fail("infgen_match_agent_tokens", "need 0 < shift < T");
  if (n_token <= 0 || n_token > 2048) return fail("infgen_match_agent_tokens", "n_token must be in 1..2048");
  MatchTokensArgs a{valid, pos, heading, shape, type, tok, tok_agent_stride, A, T, shift, n_token, token_index, token_contour};
  hipLaunchKernelGGL(k_match_tokens, dim3(A), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("infgen_match_agent_tokens");
}

extern "C" int infgen_tokenize_agent(unsigned char* valid, float* pos, float* heading, float* velocity, const int* type,
                                    const float* tok, const float* shape_in, float* shape_out, float* wl_work, int A, int T,
                                    int shift, int current_step, int n_token, int invalid_state, int valid_state,
                                    int enter_state, int exit_state, int predict_state, int* token_index,
                                    float* token_contour, int* state_idx, float* token_pos, float* token_heading,
                                    unsigned char* token_valid, unsigned char* raw_token_valid, void* stream) {
  if (A <= 0) return 0;
  if (shift <= 0 || T <= shift || current_step < shift || current_step >= T)
    return fail("infgen_tokenize_agent", "need 0 < shift <= current_step < T");
  if (n_token <= 0 || n_token > 2048) return fail("infgen_tokenize_agent", "n_token must be in 1..2048");
  TokenizeArgs t{valid, pos, heading, velocity, type, wl_work, shape_in, shape_out, A, T, shift, current_step,
                 invalid_state, valid_state, enter_state, exit_state, predict_state, token_index, token_contour,
                 state_idx, token_pos, token_heading, token_valid, raw_token_valid};
  hipLaunchKernelGGL(k_tokenize_prep, dim3((A + 63) / 64), dim3(64), 0, (hipStream_t)stream, t);
  MatchTokensArgs m{valid, pos, heading, wl_work, type, tok, 0, A, T, shift, n_token, token_index, token_contour};
  hipLaunchKernelGGL(k_match_tokens, dim3(A), dim3(256), 0, (hipStream_t)stream, m);
  hipLaunchKernelGGL(k_tokenize_state, dim3((A + 63) / 64), dim3(64), 0, (hipStream_t)stream, t);
  return check_launch("infgen_tokenize_agent");
}

extern "C" int infgen_fetch_enterings(const float* token_pos, const float* token_heading, const int* state_idx,
                                     const int* agent_ptr, const int* av_index, int B, int max_agents, int T,
                                     const float* grid_xy, int grid_size, float radius, float angle_interval,
                                     int enter_state, int invalid_state, int* grid_token_idx, float* grid_offset_xy,
                                     int* heading_token_idx, int* sort_indices, unsigned char* inrange_mask,
                                     unsigned char* bos_mask, float* pos_xy, float* heading_theta, const float* pt_pos,
                                     int pt_stride, const int* pt_ptr, int M, int* pt_grid_token_idx, void* stream) {
  if (B <= 0 || T <= 0) return 0;
  if (max_agents > 2048) return fail("infgen_fetch_enterings", "more than 2048 agents in a scene");
  if (grid_size <= 0) return fail("infgen_fetch_enterings", "empty grid");
  EnteringsArgs a{token_pos, token_heading, state_idx, agent_ptr, av_index, B, T, grid_xy, grid_size, radius, angle_interval,
                  enter_state, invalid_state, grid_token_idx, grid_offset_xy, heading_token_idx, sort_indices, inrange_mask,
                  bos_mask, pos_xy, heading_theta, pt_pos, pt_stride, pt_ptr, M, pt_grid_token_idx};
  hipLaunchKernelGGL(k_fetch_enterings, dim3(B * T), dim3(256), 0, (hipStream_t)stream, a);
  if (pt_grid_token_idx && M > 0) {
    const long long waves = (long long)T * M;
    hipLaunchKernelGGL(k_pt_grid_cells, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
  }
  return check_launch("infgen_fetch_enterings");
}

extern "C" int infgen_distance_to_nearest_object(const float* cx, const float* cy, const float* length, const float* width,
                                                const float* heading, const unsigned char* valid, int B, int N, int T,
                                                int n_eval, float corner_rounding_factor, float* work, float* out,
                                                void* stream) {
  if (B <= 0 || N <= 0 || T <= 0 || n_eval <= 0) return 0;
  if (n_eval > N) return fail("infgen_distance_to_nearest_object", "n_eval > N");
  NearestArgs a{cx, cy, length, width, heading, valid, B, N, T, n_eval, corner_rounding_factor, work, out};
  const long long nbox = (long long)B * N * T, nout = (long long)B * n_eval * T;
  hipLaunchKernelGGL(k_box_corners, dim3((unsigned)((nbox + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  hipLaunchKernelGGL(k_nearest_distance, dim3((unsigned)((nout + 127) / 128)), dim3(128), 0, (hipStream_t)stream, a);
  return check_launch("infgen_distance_to_nearest_object");
}

extern "C" int infgen_kinematic_features(const float* x, const float* y, const float* z, const float* heading, int n, int T,
                                        float seconds_per_step, float* speed, float* accel, float* yaw_rate,
                                        float* yaw_accel, void* stream) {
  if (n <= 0 || T <= 0) return 0;
  if (!speed) return fail("infgen_kinematic_features", "speed output is required");
  KinematicArgs a{x, y, z, heading, n, T, seconds_per_step, speed, accel, yaw_rate, yaw_accel};
  const long long tot = (long long)n * T;
  hipLaunchKernelGGL(k_kinematic, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("infgen_kinematic_features");
}

extern "C" int infgen_time_to_collision(const float* cx, const float* cy, const float* length, const float* width,
                                       const float* heading, const float* speed, const unsigned char* valid,
                                       const int* eval_idx, int B, int N, int T, int n_eval, float* out, void* stream) {
  if (B <= 0 || N <= 0 || T <= 0 || n_eval <= 0) return 0;
  TtcArgs a{cx, cy, length, width, heading, speed, valid, eval_idx, B, N, T, n_eval, out};
  const long long tot = (long long)B * n_eval * T;
  hipLaunchKernelGGL(k_ttc, dim3((unsigned)((tot + 127) / 128)), dim3(128), 0, (hipStream_t)stream, a);
  return check_launch("infgen_time_to_collision");
}

extern "C" int infgen_distance_to_road_edge(const float* cx, const float* cy, const float* cz, const float* length,
                                          const float* width, const float* height, const float* heading,
                                          const unsigned char* valid, const int* eval_idx, int B, int N, int T, int n_eval,
                                          const float* polylines, const unsigned char* cyclic, const int* poly_off, int L,
                                          float z_stretch, float* out, void* stream) {
  if (B <= 0 || N <= 0 || T <= 0 || n_eval <= 0) return 0;
  if (L < 2) return fail("infgen_distance_to_road_edge", "polylines need at least two points");
  RoadEdgeArgs a{cx, cy, cz, length, width, height, heading, valid, eval_idx, polylines, cyclic, poly_off,
                 B, N, T, n_eval, L, z_stretch, out};
  const long long boxes = (long long)B * n_eval * T;
  hipLaunchKernelGGL(k_road_edge, dim3((unsigned)((boxes + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("infgen_distance_to_road_edge");
}

extern "C" int infgen_window_log_likelihood(const float* values, const unsigned char* valid, int n, int T, int size, int step,
                                           const float* edges, const float* logp, int num_bins, float* out_sum,
                                           int* out_cnt, void* stream) {
  if (n <= 0) return 0;
  if (size <= 0 || step <= 0 || size > T) return fail("infgen_window_log_likelihood", "need 0 < size <= T and step > 0");
  if (num_bins <= 0 || num_bins > 64) return fail("infgen_window_log_likelihood", "num_bins must be in 1..64");
  WindowLoglikArgs a{values, valid, n, T, size, step, edges, logp, num_bins, out_sum, out_cnt};
  const long long tot = (long long)n * ((T - size) / step + 1);
  hipLaunchKernelGGL(k_window_loglik, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("infgen_window_log_likelihood");
}

extern "C" int infgen_placement_features(const float* x, const float* y, const float* z, const int* state, const int* av_index,
                                        int B, int N, int T, int enter_state, int exit_state, int* num_bos, int* num_eos,
                                        float* bos_distance, float* eos_distance, void* stream) {
  if (B <= 0 || N <= 0 || T <= 0) return 0;
  PlacementArgs a{x, y, z, state, av_index, B, N, T, enter_state, exit_state, num_bos, num_eos, bos_distance, eos_distance};
  hipLaunchKernelGGL(k_placement, dim3(B * T), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("infgen_placement_features");
}

extern "C" int infgen_match_map_tokens(const float* traj_pos, const float* theta, const float* sample_pt, int P, int n_token,
                                      int* token_idx, void* stream) {
  if (P <= 0) return 0;
  if (n_token <= 0) return fail("infgen_match_map_tokens", "n_token must be positive");
  MatchMapArgs a{traj_pos, theta, sample_pt, P, n_token, token_idx};
  hipLaunchKernelGGL(k_match_map_tokens, dim3(ceil_div(P, 4)), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("infgen_match_map_tokens");
}

// scratch: optional rows x 8 bytes - with it, launches of few row tiles deal the logit chunks to several workgroups per tile
// keys_stay (infgen_rollout_run's folded tail): the split path neither clears the keys before (k_integrate of the previous step
// did) nor decodes them after (k_integrate of this step will); *split_used tells the caller whether that path ran
static int heads_impl(const float* X, int rows, const float* tok_pack, const float* st_pack, int token_size,
                      float* logits, int* next_token, int* next_state, unsigned long long* scratch, void* stream,
                      bool keys_stay = false, bool* split_used = nullptr) {
  if (split_used) *split_used = false;
  if (rows <= 0) return 0;
  if (token_size % 128) return fail("infgen_heads", "token_size must be a multiple of 128");
  HeadsArgs a{X, rows, tok_pack, st_pack, token_size, logits, next_token, next_state, nullptr, 1};
  if (scratch && !attn_split(rows)) {
    static const int no_split = getenv("INFGEN_HEADS_NOSPLIT") ? atoi(getenv("INFGEN_HEADS_NOSPLIT")) : 0;
    const int tiles = ceil_div(rows, TR), nchunk = token_size / 128;
    int ns = 1;
    while (!no_split && 2 * ns <= nchunk && nchunk % (2 * ns) == 0 && tiles * 2 * ns <= 512) ns *= 2;
    if (ns > 1) {
      if (!keys_stay && hipMemsetAsync(scratch, 0, (size_t)rows * sizeof(unsigned long long), (hipStream_t)stream) != hipSuccess)
        return fail("infgen_heads", "memset failed");
      a.part = scratch; a.nsplit = ns;
      { ProfScope _ps(INFGEN_KID_HEADS, stream, (double)rows * (2 * 16384.0 + 128.0 * token_size + 384.0));
        hipLaunchKernelGGL(k_heads, dim3(tiles, ns), dim3(NT), 0, (hipStream_t)stream, a);
        if (!keys_stay)
          hipLaunchKernelGGL(k_heads_finish, dim3(ceil_div(rows, NT)), dim3(NT), 0, (hipStream_t)stream, scratch, rows, next_token); }
      if (split_used) *split_used = true;
      return check_launch("infgen_heads");
    }
  }
  { ProfScope _ps(INFGEN_KID_HEADS, stream, (double)rows * (2 * 16384.0 + 128.0 * token_size + 384.0));
    if (attn_split(rows)) {
      int grid = ceil_div(rows, 64);
      if (grid > 512) grid = 512;
      if (O().gemm_terms == 1) hipLaunchKernelGGL(k_heads_h<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
      else if (O().gemm_terms == 2) hipLaunchKernelGGL(k_heads_h_b16<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
      else hipLaunchKernelGGL(k_heads_h<3>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    } else {
      hipLaunchKernelGGL(k_heads, dim3(ceil_div(rows, TR)), dim3(NT), 0, (hipStream_t)stream, a);
    } }
  return check_launch("infgen_heads");
}

extern "C" int infgen_heads(const float* X, int rows, const float* tok_pack, const float* st_pack, int token_size,
                            float* logits, int* next_token, int* next_state, void* stream) {
  return heads_impl(X, rows, tok_pack, st_pack, token_size, logits, next_token, next_state, nullptr, stream);
}

extern "C" int infgen_embedding_sum4(const float* tab0, const long long* idx0, int n0, const float* tab1, const long long* idx1, int n1,
                                     const float* tab2, const long long* idx2, int n2, const float* tab3, const long long* idx3, int n3,
                                     int rows, float* out, void* stream) {
  if (rows <= 0) return 0;
  if (!tab0 || !tab1 || !tab2 || !tab3 || !idx0 || !idx1 || !idx2 || !idx3 || !out || n0 < 1 || n1 < 1 || n2 < 1 || n3 < 1)
    return fail("infgen_embedding_sum4", "null table / index array or empty table");
  EmbedSum4Args a{{tab0, tab1, tab2, tab3}, {idx0, idx1, idx2, idx3}, {n0, n1, n2, n3}, rows, out};
  hipLaunchKernelGGL(k_embedding_sum4, dim3((unsigned)(((long long)rows * 32 + NT - 1) / NT)), dim3(NT), 0, (hipStream_t)stream, a);
  return check_launch("infgen_embedding_sum4");
}

extern "C" int infgen_map_graph(int S, int M_cap, const int* n_map, const float* pos, const float* orient,
                                float radius, int max_nbr, int* off, int* cnt, int* src, float* raw, int* total,
                                int cap, void* stream) {
  if (S <= 0 || M_cap <= 0) return 0;
  if (hipMemsetAsync(total, 0, sizeof(int), (hipStream_t)stream) != hipSuccess)
    return fail("infgen_map_graph", "memset failed");
  MapGraphArgs a{S, M_cap, n_map, pos, orient, radius, max_nbr, EdgeBuf{off, cnt, src, raw, total, cap}, 0};
  a.lds_tokens = (M_cap <= 4096 && M_cap % (4 * MAP_GRAPH_C) == 0) ? M_cap : 0;      // (else: the global-memory scan)
  { ProfScope _ps(INFGEN_KID_MAP_GRAPH, stream);
    hipLaunchKernelGGL(k_map_graph, dim3(ceil_div(S * M_cap, 4 * MAP_GRAPH_C)), dim3(NT), (size_t)a.lds_tokens * 12, (hipStream_t)stream, a); }
  return check_launch("infgen_map_graph");
}

// ---------------------------------------------------------------------------------- rollout context
static SceneState scene_of(const InfgenRollout* r) {
  SceneState st;
  st.S = r->S; st.A_cap = r->A_cap; st.T = r->T; st.M_cap = r->M_cap; st.W = r->W; st.ring = r->ring;
  st.n_agents = r->n_agents; st.n_map = r->n_map; st.av_index = r->av_index;
  st.pos = r->pos; st.head = r->head; st.state = r->state; st.token = r->token; st.grid = r->grid;
  st.tmask = r->tmask; st.imask = r->imask; st.catflag = r->catflag; st.type = r->type; st.bos = r->bos;
  st.map_pos = r->map_pos; st.map_orient = r->map_orient; st.map_scene = r->map_scene;
  st.first_new = r->first_new; st.hv_ovr = r->hv_ovr;
  return st;
}
static EdgeBuf ebuf(const InfgenEdgeBuf& e) { return EdgeBuf{e.off, e.cnt, e.src, e.raw, e.total, e.cap}; }

static int validate(const InfgenRollout* r, const char* where) {
  if (!r) return fail(where, "null context");
  if (r->A_cap > 1024 || r->A_cap <= 0) return fail(where, "A_cap must be in 1..1024");
  if (r->A_cap % 32) return fail(where, "A_cap must be a multiple of 32");
  if (r->num_layers <= 0 || r->num_layers > INFGEN_MAX_LAYERS) return fail(where, "bad num_layers");
  if (r->ring <= r->W) return fail(where, "ring must exceed the temporal window");
  if (r->W > 16) return fail(where, "temporal window larger than 16 columns is not supported");
  return 0;
}

static int build_edges_impl(const InfgenRollout* r, int c, int edgeless, void* stream, bool zero_totals, unsigned long long* clear_keys = nullptr,
                            bool clear_sync = false);
extern "C" int infgen_build_edges(const InfgenRollout* r, int c, int edgeless, void* stream) {
  return build_edges_impl(r, c, edgeless, stream, true);
}
// zero_totals = false: the three totals were cleared by the previous step's k_integrate (IntegrateArgs.edge_totals)
static int build_edges_impl(const InfgenRollout* r, int c, int edgeless, void* stream, bool zero_totals, unsigned long long* clear_keys,
                            bool clear_sync) {
  RET_IF(validate(r, "infgen_build_edges"));
  OptScope _opts(r);
  hipStream_t s = (hipStream_t)stream;
  if (!edgeless && zero_totals) {
    if (r->em.total == r->et.total + 1 && r->ea.total == r->et.total + 2) {      // laid out back to back: one fill
      if (hipMemsetAsync(r->et.total, 0, 3 * sizeof(int), s) != hipSuccess) return fail("infgen_build_edges", "memset failed");
    } else if (hipMemsetAsync(r->et.total, 0, sizeof(int), s) != hipSuccess ||
               hipMemsetAsync(r->em.total, 0, sizeof(int), s) != hipSuccess ||
               hipMemsetAsync(r->ea.total, 0, sizeof(int), s) != hipSuccess)
      return fail("infgen_build_edges", "memset failed");
  }
  BuildEdgesArgs a;
  a.st = scene_of(r); a.c = c; a.edgeless = edgeless; a.r_map = r->r_map; a.r_agent = r->r_agent;
  a.rows = r->S * r->A_cap; a.t = ebuf(r->et); a.m = ebuf(r->em); a.a = ebuf(r->ea);
  a.clear_keys = clear_keys;
  a.clear_sync = clear_sync ? reinterpret_cast<int*>(r->SIG) : nullptr;      // (k_layers_p's per-scene counters: layers_p_launch)
  a.prof = (g_prof.mask & ((1u << INFGEN_KID_EDGE_ATTN) | (1u << INFGEN_KID_BUILD_EDGES))) ? g_prof.rows_dev : nullptr;
  { ProfScope _ps(INFGEN_KID_BUILD_EDGES, stream);
    a.map_lds = r->M_cap < 4096 ? r->M_cap : 4096;
    const size_t lds = (size_t)a.map_lds * 8;
    // few scenes: 16 waves per workgroup whatever A_cap (INFGEN_BE_WIDE_SCENES: the largest batch that takes them, default 128)
    static const int wide_scenes = getenv("INFGEN_BE_WIDE_SCENES") ? atoi(getenv("INFGEN_BE_WIDE_SCENES")) : 128;
    if (r->A_cap <= 256 && r->S > wide_scenes) hipLaunchKernelGGL(k_build_edges<256>, dim3(r->S, 3), dim3(256), lds, s, a);
    else hipLaunchKernelGGL(k_build_edges<1024>, dim3(r->S, 3), dim3(1024), lds, s, a); }
  return check_launch("infgen_build_edges");
}

static RawFeatArgs rawfeat_args(const InfgenRollout* r, int col);
// workgroups per scene of k_integrate: 16 rows each for few scenes (INFGEN_INT_GROUP_SCENES: the largest batch, default 128), else 1.
// With more than one the arg-max keys are reset by the k_build_edges that follows (build_edges_impl: clear_keys), not by k_integrate.
static int integrate_groups(const InfgenRollout* r) {
  static const int max_scenes = getenv("INFGEN_INT_GROUP_SCENES") ? atoi(getenv("INFGEN_INT_GROUP_SCENES")) : 128;
  return (r->S <= max_scenes && r->A_cap % 16 == 0 && r->A_cap > 16) ? r->A_cap / 16 : 1;
}
static int integrate_impl(const InfgenRollout* r, int t, void* stream, unsigned long long* heads_part, bool zero_totals, bool prep, bool zero_sync = false);
extern "C" int infgen_integrate(const InfgenRollout* r, int t, void* stream) {
  return integrate_impl(r, t, stream, nullptr, false, false);
}
static int integrate_impl(const InfgenRollout* r, int t, void* stream, unsigned long long* heads_part, bool zero_totals, bool prep, bool zero_sync) {
  RET_IF(validate(r, "infgen_integrate"));
  OptScope _opts(r);
  IntegrateArgs a;
  a.heads_part = heads_part; a.next_token_w = r->next_token;
  a.edge_totals = zero_totals ? r->et.total : nullptr;
  a.zero_sync = zero_sync ? reinterpret_cast<int*>(r->SIG) : nullptr;       // (k_layers_p's per-scene counters: layers_p_launch)
  a.do_prep = prep ? 1 : 0;
  if (prep) a.prep = rawfeat_args(r, 2 + t);
  a.st = scene_of(r); a.c = 1 + t; a.t = t; a.R = r->R; a.force_valid = r->force_valid;
  a.next_token = r->next_token; a.next_state = r->next_state;
  a.teacher_token = r->teacher_token; a.teacher_state = r->teacher_state; a.teacher_grid = r->teacher_grid;
  a.teacher_pos = r->teacher_pos; a.teacher_head = r->teacher_head;
  a.vocab = r->vocab; a.token_size = r->token_size; a.grid_xy = r->grid_xy; a.grid_size = r->grid_size;
  a.pred_traj = r->pred_traj; a.pred_head = r->pred_head; a.pred_state = r->pred_state;
  a.groups = integrate_groups(r);
  { ProfScope _ps(INFGEN_KID_INTEGRATE, stream);
    // 16 waves per workgroup whatever A_cap: the grid-cell search is one wave per agent (4 instead of 16 agents per wave with 256
    // threads: 0.95 -> 0.56 ms per rollout at 8 scenes, 1.23 -> 0.98 at 512); few scenes: 16 rows per workgroup, one wave each
    hipLaunchKernelGGL(k_integrate<1024>, dim3(r->S, a.groups), dim3(1024), 0, (hipStream_t)stream, a); }
  return check_launch("infgen_integrate");
}

// MLPEmbedding pack (first Linear K0 -> 128): P(K0p,128) b ln_g ln_b | P(128,128) b ln_g ln_b | P(128,128) b
static inline int mlpemb_off2(int K0p) { return K0p * 128 + 3 * 128; }
static inline int mlpemb_off3(int K0p) { return mlpemb_off2(K0p) + 16384 + 3 * 128; }

static RawFeatArgs rawfeat_args(const InfgenRollout* r, int col) {
  RawFeatArgs a;
  a.st = scene_of(r); a.col = col; a.tok_tab = r->tok_tab; a.token_size = r->token_size;
  a.grid_tab = r->grid_tab; a.grid_size = r->grid_size; a.state_emb = r->state_emb;
  a.cat_agent = r->cat_agent; a.cat_seed = r->cat_seed; a.raw2 = r->raw2; a.cat = r->cat; a.fus_in = r->fus_in;
  a.row_list = nullptr; a.row_mask = nullptr; a.n_list = 0;
  return a;
}

static int raw_feature_fusion(const InfgenRollout* r, void* stream);
extern "C" int infgen_raw_feature(const InfgenRollout* r, int col, void* stream) {
  RET_IF(validate(r, "infgen_raw_feature"));
  OptScope _opts(r);
  const int rows = r->S * r->A_cap;
  RawFeatArgs a = rawfeat_args(r, col);
  { ProfScope _ps(INFGEN_KID_RAWFEAT, stream);
    hipLaunchKernelGGL(k_rawfeat_prep, dim3(ceil_div(rows * 32, NT)), dim3(NT), 0, (hipStream_t)stream, a); }
  RET_IF(check_launch("infgen_raw_feature/prep"));
  RET_IF(infgen_fourier_embed(r->raw2, 2, nullptr, rows, r->four_xa, r->cat, 128, r->fus_in + 128, 512, 0, stream));
  return raw_feature_fusion(r, stream);
}

// MLPEmbedding (reference infgen/modules/layers.py:163-192) with K0 = 128 j inputs: Linear LN ReLU Linear LN ReLU Linear.  Split
// arithmetic (attn_mode != 0): the three stages in one launch of k_mlpemb_h (any row count: one 25 us chain instead of three dependent
// fp32 launches of 30-40 us each); attn_mode 0: three k_linear launches through tmp1 / tmp2 [rows][128]
static int mlp_embedding_impl(const float* X, int ldx, int rows, int K0, const float* P, float* tmp1, float* tmp2, float* Y, int ldy,
                              void* stream, const char* where) {
  if (rows <= 0) return 0;
  if (O().attn_mode != 0) {
    MlpEmbHArgs m{X, ldx, rows, K0, P, Y, ldy};
    int grid = ceil_div(rows, 64);
    if (grid > 512) grid = 512;
    { ProfScope _ps(INFGEN_KID_LINEAR, stream, (double)rows * (K0 + 128 + 128) * 128.0);
      if (O().gemm_terms == 1) hipLaunchKernelGGL(k_mlpemb_h<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, m);
      else if (O().gemm_terms == 2) hipLaunchKernelGGL(k_mlpemb_h_b16<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, m);
      else hipLaunchKernelGGL(k_mlpemb_h<3>, dim3(grid), dim3(256), 0, (hipStream_t)stream, m); }
    return check_launch(where);
  }
  if (!tmp1 || !tmp2) return fail(where, "the fp32 kernels need the two scratch arrays");
  const int o2 = mlpemb_off2(K0), o3 = mlpemb_off3(K0);
  RET_IF(infgen_linear(X, ldx, nullptr, rows, K0, P, 128, P + K0 * 128, 128, nullptr, nullptr,
                       P + K0 * 128 + 128, P + K0 * 128 + 256, 1, tmp1, 128, stream));
  RET_IF(infgen_linear(tmp1, 128, nullptr, rows, 128, P + o2, 128, P + o2 + 16384, 128, nullptr, nullptr,
                       P + o2 + 16384 + 128, P + o2 + 16384 + 256, 1, tmp2, 128, stream));
  RET_IF(infgen_linear(tmp2, 128, nullptr, rows, 128, P + o3, 128, P + o3 + 16384, 128, nullptr, nullptr,
                       nullptr, nullptr, 0, Y, ldy, stream));
  return 0;
}
extern "C" int infgen_mlp_embedding(const float* X, int ldx, int rows, int K0, const float* pack, float* tmp1, float* tmp2,
                                    float* Y, int ldy, void* stream) {
  if (K0 <= 0 || K0 % 128 || K0 > 512) return fail("infgen_mlp_embedding", "K0 must be 128, 256, 384 or 512");
  return mlp_embedding_impl(X, ldx, rows, K0, pack, tmp1, tmp2, Y, ldy, stream, "infgen_mlp_embedding");
}

// fusion_emb of the gathered rows (fus_in [rows][512]) -> X
static int raw_feature_fusion(const InfgenRollout* r, void* stream) {
  return mlp_embedding_impl(r->fus_in, 512, r->S * r->A_cap, 512, r->fusion_pack, r->tmp1, r->tmp2, r->X, 128, stream,
                            "infgen_raw_feature/fusion");
}

// the same for a few rows only (insertion: the rows appended in this sub-loop iteration): row_list[k] = row, used where
// row_mask[k] != 0.  The first n rows of the scratch arrays raw2 / cat / fus_in / tmp1 / tmp2 hold the compact intermediate
// results (every all-rows call rewrites them); X is updated at the listed rows only.
extern "C" int infgen_raw_feature_rows(const InfgenRollout* r, int col, const int* row_list, const int* row_mask, int n,
                                       void* stream) {
  RET_IF(validate(r, "infgen_raw_feature_rows"));
  OptScope _opts(r);
  if (n <= 0) return 0;
  if (n > r->S * r->A_cap) return fail("infgen_raw_feature_rows", "more rows than the layout holds");
  RawFeatArgs a;
  a.st = scene_of(r); a.col = col; a.tok_tab = r->tok_tab; a.token_size = r->token_size;
  a.grid_tab = r->grid_tab; a.grid_size = r->grid_size; a.state_emb = r->state_emb;
  a.cat_agent = r->cat_agent; a.cat_seed = r->cat_seed; a.raw2 = r->raw2; a.cat = r->cat; a.fus_in = r->fus_in;
  a.row_list = row_list; a.row_mask = row_mask; a.n_list = n;
  { ProfScope _ps(INFGEN_KID_RAWFEAT, stream);
    hipLaunchKernelGGL(k_rawfeat_prep, dim3(ceil_div(n * 32, NT)), dim3(NT), 0, (hipStream_t)stream, a); }
  RET_IF(check_launch("infgen_raw_feature_rows/prep"));
  RET_IF(infgen_fourier_embed(r->raw2, 2, nullptr, n, r->four_xa, r->cat, 128, r->fus_in + 128, 512, 0, stream));
  const float* P = r->fusion_pack;
  if (O().attn_mode != 0) {
    MlpEmbHArgs m{r->fus_in, 512, n, 512, P, r->tmp1, 128};
    int grid = ceil_div(n, 64);
    if (grid > 512) grid = 512;
    { ProfScope _ps(INFGEN_KID_LINEAR, stream, (double)n * (512 + 128 + 128) * 128.0);
      if (O().gemm_terms == 1) hipLaunchKernelGGL(k_mlpemb_h<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, m);
      else if (O().gemm_terms == 2) hipLaunchKernelGGL(k_mlpemb_h_b16<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, m);
      else hipLaunchKernelGGL(k_mlpemb_h<3>, dim3(grid), dim3(256), 0, (hipStream_t)stream, m); }
    RET_IF(check_launch("infgen_raw_feature_rows/fusion"));
  } else {
    const int o2 = mlpemb_off2(512), o3 = mlpemb_off3(512);
    RET_IF(infgen_linear(r->fus_in, 512, nullptr, n, 512, P, 128, P + 512 * 128, 128, nullptr, nullptr,
                         P + 512 * 128 + 128, P + 512 * 128 + 256, 1, r->tmp1, 128, stream));
    RET_IF(infgen_linear(r->tmp1, 128, nullptr, n, 128, P + o2, 128, P + o2 + 16384, 128, nullptr, nullptr,
                         P + o2 + 16384 + 128, P + o2 + 16384 + 256, 1, r->tmp2, 128, stream));
    RET_IF(infgen_linear(r->tmp2, 128, nullptr, n, 128, P + o3, 128, P + o3 + 16384, 128, nullptr, nullptr,
                         nullptr, nullptr, 0, r->tmp1, 128, stream));
  }
  hipLaunchKernelGGL(k_scatter_rows, dim3(ceil_div(n * 32, NT)), dim3(NT), 0, (hipStream_t)stream, r->tmp1, row_list, row_mask, n, r->X);
  return check_launch("infgen_raw_feature_rows/scatter");
}

extern "C" int infgen_sample_topk(const float* logits, int rows, int n, int k, const float* uniform, int* token,
                                  void* stream) {
  if (rows <= 0) return 0;
  if (k < 1 || k > 16) return fail("infgen_sample_topk", "k must be in 1..16");
  SampleArgs a{logits, rows, n, k, uniform, token};
  hipLaunchKernelGGL(k_sample_topk, dim3(ceil_div(rows, 4)), dim3(NT), 0, (hipStream_t)stream, a);
  return check_launch("infgen_sample_topk");
}

// Optional: the Fourier embeddings of the map and agent edge sets (matrix-pipe / VALU work) run on a side stream while
// the temporal and map sublayers of the first layer (memory-bound) run on the caller's stream.
static hipStream_t g_side = nullptr;
static hipEvent_t g_ev_fork = nullptr, g_ev_m = nullptr, g_ev_a = nullptr;
extern "C" int infgen_set_overlap(int mode) {
  if (mode != 0 && mode != 1) return fail("infgen_set_overlap", "mode must be 0 or 1");
  if (mode && !g_side) {
    if (hipStreamCreateWithFlags(&g_side, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&g_ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&g_ev_m, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&g_ev_a, hipEventDisableTiming) != hipSuccess)
      return fail("infgen_set_overlap", "stream / event creation failed");
  }
  g_def.overlap = mode;
  return 0;
}

static int fourier_nomulti() {
  static const int v = getenv("INFGEN_FOURIER_NOMULTI") ? atoi(getenv("INFGEN_FOURIER_NOMULTI")) : 0;
  return v;
}

// ---- k_layers_p (layers_p.hip): every sublayer of a step in one launch, one resident workgroup per 16-row group
extern "C" int infgen_set_layers_p(int mode) {        // (process-wide default, like the other infgen_set_*: contexts carry their own copy)
  if (mode < 0 || mode > 2) return fail("infgen_set_layers_p", "mode must be 0, 1 or 2");
  g_def.layers_p = mode;
  return 0;
}
// scenes per launch when the batch needs more than one (0: not eligible): whole scenes, at most `limit` 16-row groups per launch,
// at most INFGEN_LP_MAX_CHUNKS (2) launches - measured with scenes of 64 agents: 96 / 128 scenes 36.6 / 39.6 ms per rollout through the
// per-sublayer launches, whose layers cost 26 - 28 ms there against 2 x 10 ms of two 64-scene launches; from three chunks on the
// big-batch kernels are as fast
static int lp_chunk_scenes(const InfgenRollout* r, int limit) {
  static const int max_chunks = getenv("INFGEN_LP_MAX_CHUNKS") ? atoi(getenv("INFGEN_LP_MAX_CHUNKS")) : 2;
  const int gps = r->A_cap / 16;
  if (gps <= 0 || gps > limit) return 0;
  const int per = limit / gps;                       // scenes per launch
  const int chunks = (r->S + per - 1) / per;
  return chunks <= max_chunks ? per : 0;
}
static int lp_max_groups() {
  static const int v = getenv("INFGEN_LP_MAX_GROUPS") ? atoi(getenv("INFGEN_LP_MAX_GROUPS")) : 256;
  return v;
}
// the launch shape qualifies (the kernel keeps U / Z on chip like k_edge_fused: step_mode treats it as a fused launch)
// k_layers_p's workgroups meet at counters in global memory, so a launch needs ALL of them resident at some point:
//  * the grid is checked against the device's resident capacity for this kernel (hipOccupancyMaxActiveBlocksPerMultiprocessor x CUs:
//    layers_p_shape) - what fits becomes resident as soon as kernels of other streams leave the CUs, so the wait at the counters
//    has no limit (no trap);
//  * launches of different streams of this process are ordered behind each other (two half-resident launches would wait for each
//    other for ever): g_lp_mu / g_lp_order below (an event recorded behind every launch on its own stream);
//  * layers_p == 2 additionally launches through hipLaunchCooperativeKernel (the runtime's own residency contract, and the
//    cooperative queue is device-wide - this also covers another PROCESS running such a kernel on the same GPU); measured cost of
//    the queue hand-over: 8 scenes 9.54 -> 9.97 ms per rollout, 64 scenes 17.01 -> 17.36 (tools/lp_coop_ab.sh).  A refused
//    cooperative launch falls back to the per-sublayer kernels for good (g_lp_refused).
static std::atomic<bool> g_lp_refused{false};
static std::mutex g_lp_mu;                  // orders the k_layers_p launches of the process's streams (per device: LpOrder)
constexpr int LP_MAX_DEVICES = 16;
struct LpOrder { hipEvent_t ev = nullptr; bool any = false; };     // ev: recorded behind the device's last k_layers_p launch, on the
static LpOrder g_lp_order[LP_MAX_DEVICES];                         // stream that launched it (no stream handle is kept)
struct LpDevice { bool init = false; int n_cu = 1; int coop = 0; int wg_per_cu = 0; };
static int lp_current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
  return dev < 0 || dev >= LP_MAX_DEVICES ? 0 : dev;
}
static const LpDevice& lp_device() {
  static LpDevice devs[LP_MAX_DEVICES];
  static std::mutex mu;
  const int dev = lp_current_device();
  std::lock_guard<std::mutex> lk(mu);
  LpDevice& v = devs[dev];
  if (!v.init) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) {
      v.n_cu = prop.multiProcessorCount;
      (void)hipDeviceGetAttribute(&v.coop, hipDeviceAttributeCooperativeLaunch, dev);
      // resident workgroups per CU of the largest variant (the 120 KB of LDS set it: 1)
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v.wg_per_cu, k_layers_p<true, 16>, 512, 0) != hipSuccess) v.wg_per_cu = 0;
    }
    (void)hipGetLastError();
    v.init = true;
  }
  return v;
}
extern "C" int infgen_layers_p_capacity(void) {       // workgroups one k_layers_p launch may have on the current device (0: not available)
  const LpDevice& d = lp_device();
  if (d.wg_per_cu <= 0) return 0;
  return d.n_cu * (d.wg_per_cu > 1 ? 1 : d.wg_per_cu);      // (the launch shapes assume one workgroup per CU)
}
// workgroups a launch may have: the device's resident capacity (a partition with fewer CUs than 256 - DPX / QPX / CPX modes, smaller
// parts - lowers it), and INFGEN_LP_MAX_GROUPS
static int lp_limit() {
  const int cap = infgen_layers_p_capacity(), mx = lp_max_groups();
  return cap < mx ? cap : mx;
}
// k_layers_p bounds a row's LayerNorm output with header slots 10..13 of the attention packs; a pack without them (an older or
// foreign packer: slot 14 != AH_HDR_VERSION) would scale its operands by 2^126.  Checked once per pack pointer (a 64-byte
// device -> host copy at a context's first launch); contexts with such a pack take the per-sublayer launches.
static bool lp_packs_ok(const InfgenRollout* r, bool refresh = false, void* stream = nullptr) {
  static std::mutex mu;
  static std::unordered_map<const float*, bool> seen;      // verdict per pack address, overwritten in place on refresh
  std::lock_guard<std::mutex> lk(mu);
  // a pack that was never examined needs a synchronous 64-byte copy: not while `stream` is being captured into a graph (the copy
  // would fail and the verdict would stick) - such a launch takes the per-sublayer kernels, the verdict stays open
  bool capturing = false;
  if (stream) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    capturing = hipStreamIsCapturing((hipStream_t)stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
  }
  auto ok = [&](const float* pack) {
    if (!pack) return false;
    if (!refresh) { auto it = seen.find(pack); if (it != seen.end()) return it->second; }
    if (capturing) return false;
    float hdr[16] = {};
    const bool good = hipMemcpy(hdr, pack + AH_HDR, sizeof(hdr), hipMemcpyDeviceToHost) == hipSuccess && hdr[14] == AH_HDR_VERSION &&
                      hdr[10] > 0.f && hdr[12] > 0.f;
    seen[pack] = good;
    return good;
  };
  for (int i = 0; i < r->num_layers; ++i)
    if (!ok(r->attn_t[i]) || !ok(r->attn_m[i]) || !ok(r->attn_a[i])) return false;
  return true;
}
// A context's packs looked at afresh (a device address may have held another pack before): callers that build a context call
// this once (infgen_amd/engine.py: _build_ctx); without it a pack is examined when its address is first seen.  Synchronous.
extern "C" int infgen_rollout_validate(const InfgenRollout* r) {
  RET_IF(validate(r, "infgen_rollout_validate"));
  (void)lp_packs_ok(r, true);
  return 0;
}
static bool layers_p_shape(const InfgenRollout* r, int rows, int edgeless, void* stream) {
  const LpDevice& lpd = lp_device();
  if (lpd.wg_per_cu <= 0 || (O().layers_p == 2 && (!lpd.coop || g_lp_refused.load()))) return false;
  // all workgroups must be resident at once (they meet at per-scene counters): at most one workgroup per CU (120 KB of LDS each).
  // Up to that limit the one-launch kernel wins at every size measured (scenes of 64 agents, ms per rollout, k_layers_p vs the
  // per-sublayer launches): 8 scenes 10.7 / 15.4, 16: 11.7 / 16.4, 32: 13.7 / 18.0 (8 rows per workgroup), 48: 16.8 / 20.1,
  // 64: 18.4 / 22.2 (16 rows, 256 workgroups).  (With every wave executing the scene counter's agent-scope fences the crossover
  // sat at ~40 scenes: 2,048 L2 write-backs per layer.)  INFGEN_LP_MAX_GROUPS lowers the limit.
  // (a row-group list of an insertion context is ignored: the launch visits every group - the ones without agents have empty edge
  // lists and run in parallel on CUs that would idle)
  return O().layers_p && O().edge_fuse != 0 && O().attn_mode != 0 && O().gemm_terms == 3 && O().fourier_mode != 0 &&
         !(O().overlap && g_side) && r->A_cap % 16 == 0 && lp_chunk_scenes(r, lp_limit()) > 0 &&
          r->num_layers <= LP_MAX_LAYERS && r->U && r->SIG && !r->tap_x && lp_packs_ok(r, false, stream);
}

// ---- a decode step in two halves: the edge sets of a column with their embeddings, and the 18 sublayers that consume them
struct StepMode { bool overlap, fuse, lp; int r24; int ra; const float* dt; };
static StepMode step_mode(const InfgenRollout* r, int rows, int edgeless, void* stream) {
  StepMode m;
  m.overlap = O().overlap && g_side && !edgeless;
  m.lp = layers_p_shape(r, rows, edgeless, stream);
  m.fuse = O().edge_fuse == 2 || (O().edge_fuse == 1 && (rows > 256 || m.lp));      // U / Z / SIG stay on chip inside k_edge_fused / k_layers_p
  // the step's rhat rows never leave the library: fp32 rows by default (the reference's arithmetic); rhat_format 1: packed 24-bit
  // rows (kernels.h) when both ends are the kernels that know them - a reduced-precision mode, tests/test_rollout_gpu.py compares the two
  m.r24 = m.fuse && O().fourier_mode != 0 && O().rhat_format == 1;
  m.ra = m.r24;
  // the temporal set's fourth input (the time gap, one of -1 .. -16) as a lookup of its branch (kernels.h: dt_mode)
  m.dt = O().fourier_mode != 0 ? r->four_t_dt : nullptr;
  return m;
}
// few rows: the step's Fourier embeddings side by side in one launch (k_fourier_h_multi)
static bool fourier_multi_ok(int rows, int edgeless) {
  return !edgeless && O().fourier_mode != 0 && rows <= 10240 && !fourier_nomulti() && !(O().overlap && g_side);
}

// edge sets of column c + their Fourier embeddings.  zero_totals = false: the totals were cleared by k_integrate; with_xa: the
// x_a_emb embedding of the rows' raw features (raw2 / cat -> fus_in, infgen_raw_feature's middle launch) rides along as a
// fourth set of the multi launch
static int prepare_edges(const InfgenRollout* r, int c, int edgeless, void* stream, bool zero_totals, bool with_xa,
                         unsigned long long* clear_keys = nullptr, bool clear_sync = false) {
  const int rows = r->S * r->A_cap;
  RET_IF(build_edges_impl(r, c, edgeless, stream, zero_totals, clear_keys, clear_sync));
  const StepMode sm = step_mode(r, rows, edgeless, stream);
  const bool overlap = sm.overlap; const int r24 = sm.r24, ra = sm.ra; const float* dt = sm.dt;
  if (overlap) {
    hipStream_t ms = (hipStream_t)stream;
    if (hipEventRecord(g_ev_fork, ms) != hipSuccess || hipStreamWaitEvent(g_side, g_ev_fork, 0) != hipSuccess)
      return fail("infgen_decode_layers", "fork failed");
    RET_IF(fourier_embed_impl(r->em.raw, 3, r->em.total, r->em.cap, r->four_m, nullptr, 0, r->em.rhat, 128, 1, r24, g_side));
    if (hipEventRecord(g_ev_m, g_side) != hipSuccess) return fail("infgen_decode_layers", "event failed");
    RET_IF(fourier_embed_impl(r->ea.raw, 3, r->ea.total, r->ea.cap, r->four_a, nullptr, 0, r->ea.rhat, 128, 1, ra, g_side));
    if (hipEventRecord(g_ev_a, g_side) != hipSuccess) return fail("infgen_decode_layers", "event failed");
    RET_IF(fourier_embed_impl(r->et.raw, 4, r->et.total, r->et.cap, r->four_t, nullptr, 0, r->et.rhat, 128, 1, r24, stream, dt, dt != nullptr));
  } else if (fourier_multi_ok(rows, edgeless)) {
    // few rows: the three sets side by side in one launch (each is a handful of 128-edge tiles of 35 - 45 us)
    unsigned long long* pr = (g_prof.mask >> INFGEN_KID_FOURIER) & 1u ? g_prof.rows_dev : nullptr;
    FourierMultiArgs m;
    m.set[0] = FourierArgs{r->et.raw, 4, r->et.total, r->et.cap, r->four_t, nullptr, 0, r->et.rhat, 128, 1, pr, r24, dt, dt != nullptr};
    m.set[1] = FourierArgs{r->em.raw, 3, r->em.total, r->em.cap, r->four_m, nullptr, 0, r->em.rhat, 128, 1, pr, r24, nullptr, 0};
    m.set[2] = FourierArgs{r->ea.raw, 3, r->ea.total, r->ea.cap, r->four_a, nullptr, 0, r->ea.rhat, 128, 1, pr, ra, nullptr, 0};
    if (with_xa)
      m.set[3] = FourierArgs{r->raw2, 2, nullptr, rows, r->four_xa, r->cat, 128, r->fus_in + 128, 512, 0, pr, 0, nullptr, 0};
    int cap = r->et.cap > r->em.cap ? r->et.cap : r->em.cap;
    if (r->ea.cap > cap) cap = r->ea.cap;
    // (INFGEN_FH12_MULTI_ROWS: smallest launch, in rows, whose multi-set launch takes the three-wave-group kernel; default: never)
    static const int fh12_rows = getenv("INFGEN_FH12_MULTI_ROWS") ? atoi(getenv("INFGEN_FH12_MULTI_ROWS")) : 0;
    const bool wide = fh12_rows > 0 && rows >= fh12_rows && O().gemm_terms == 3 && FH_WAVES == 8;
    int grid = ceil_div(cap, wide ? FH12_TILE : FH_TILE);
    if (grid > 256 * FH_WG_PER_CU) grid = 256 * FH_WG_PER_CU;
    { ProfScope _ps(INFGEN_KID_FOURIER, stream);
      if (wide) hipLaunchKernelGGL(k_fourier_h12_multi<3>, dim3(grid, with_xa ? 4 : 3), dim3(FH12_NT), 0, (hipStream_t)stream, m);
      else if (O().gemm_terms == 1) hipLaunchKernelGGL(k_fourier_h_multi<1>, dim3(grid, with_xa ? 4 : 3), dim3(FH_NT), 0, (hipStream_t)stream, m);
      else if (O().gemm_terms == 2) hipLaunchKernelGGL(k_fourier_h_multi_b16<1>, dim3(grid, with_xa ? 4 : 3), dim3(FH_NT), 0, (hipStream_t)stream, m);
      else hipLaunchKernelGGL(k_fourier_h_multi<3>, dim3(grid, with_xa ? 4 : 3), dim3(FH_NT), 0, (hipStream_t)stream, m); }
    RET_IF(check_launch("infgen_decode_layers(fourier)"));
  } else if (!edgeless) {
    RET_IF(fourier_embed_impl(r->et.raw, 4, r->et.total, r->et.cap, r->four_t, nullptr, 0, r->et.rhat, 128, 1, r24, stream, dt, dt != nullptr));
    RET_IF(fourier_embed_impl(r->em.raw, 3, r->em.total, r->em.cap, r->four_m, nullptr, 0, r->em.rhat, 128, 1, r24, stream));
    RET_IF(fourier_embed_impl(r->ea.raw, 3, r->ea.total, r->ea.cap, r->four_a, nullptr, 0, r->ea.rhat, 128, 1, ra, stream));
  }
  return 0;
}

constexpr int LP_REFUSED = -12345;      // layers_p_launch: the runtime refused the cooperative launch, nothing was enqueued
// sync_clear = false: the scenes' counters are already zero (the k_integrate of the previous decode step cleared them: IntegrateArgs.zero_sync)
static int layers_p_launch(const InfgenRollout* r, int c, const StepMode& sm, void* stream, bool sync_clear = true) {
  const int rows = r->S * r->A_cap;
  LayersPArgs a;
  a.rows = rows; a.A_cap = r->A_cap; a.num_layers = r->num_layers;
  a.X = r->X;
  for (int i = 0; i < r->num_layers; ++i) {
    a.attn_t[i] = r->attn_t[i]; a.attn_m[i] = r->attn_m[i]; a.attn_a[i] = r->attn_a[i];
    a.ringK[i] = r->ringK[i]; a.ringV[i] = r->ringV[i]; a.mapK[i] = r->mapK[i]; a.mapV[i] = r->mapV[i];
  }
  a.slot_off = (size_t)(c % r->ring) * rows * D;
  // the agent set's K / V rows double buffered by layer parity: the context's Ka / Va, and the first rows of its U array (4 KB per
  // row, unused while U stays on chip); the scenes' counters in its SIG array (unused for the same reason)
  a.Ka[0] = r->Ka; a.Va[0] = r->Va; a.Ka[1] = r->U; a.Va[1] = r->U + (size_t)rows * D;
  a.et = EdgeSet{r->et.off, r->et.cnt, r->et.src, r->et.rhat};
  a.em = EdgeSet{r->em.off, r->em.cnt, r->em.src, r->em.rhat};
  a.ea = EdgeSet{r->ea.off, r->ea.cnt, r->ea.src, r->ea.rhat};
  a.sync = reinterpret_cast<int*>(r->SIG);
  a.trace = nullptr;
  static const int lp_trace = getenv("INFGEN_LP_TRACE") ? atoi(getenv("INFGEN_LP_TRACE")) : 0;
  static unsigned long long* trace_dev = nullptr;
  if (lp_trace) {
    if (!trace_dev && hipMalloc(&trace_dev, 2048 * sizeof(unsigned long long)) != hipSuccess) return fail("infgen_decode_layers", "trace buffer");
    (void)hipMemsetAsync(trace_dev, 0, 2048 * sizeof(unsigned long long), (hipStream_t)stream);
    a.trace = trace_dev;
  }
  if (sync_clear && hipMemsetAsync(a.sync, 0, (size_t)r->S * sizeof(int), (hipStream_t)stream) != hipSuccess)
    return fail("infgen_decode_layers", "memset failed");
  // fewer rows per workgroup while the launch stays within the limit: 8 (one row per wave in the edge loop) up to 256 workgroups,
  // 4 (a row's edge list halved between two waves) up to 128
  static const int lp_rows_min = getenv("INFGEN_LP_ROWS8") ? (atoi(getenv("INFGEN_LP_ROWS8")) ? 8 : 16)
                               : getenv("INFGEN_LP_ROWS_MIN") ? atoi(getenv("INFGEN_LP_ROWS_MIN")) : 4;
  // (every bound below is the launch limit lp_limit() = min(resident capacity of THIS device, INFGEN_LP_MAX_GROUPS): a grid beyond the
  // capacity would leave workgroups that never become resident while the resident ones wait for them at the scene counters)
  const int limit = lp_limit();
  a.rows_per_wg = 16;
  for (int rr = 8; rr >= 4 && rr >= lp_rows_min; rr >>= 1)
    if (r->A_cap % rr == 0 && rows / rr <= (rr == 4 ? limit / 2 : limit)) a.rows_per_wg = rr;
  const int gps = r->A_cap / a.rows_per_wg;
  // a batch beyond one launch's workgroups: chunks of whole scenes, one launch after the other (each launch's workgroups are all
  // resident; the stream orders them).  Only 16-row workgroups get here with more than one chunk (fewer rows per workgroup are
  // chosen only when the whole batch fits), so lp_chunk_scenes' groups per scene are this launch's
  const int per = rows / a.rows_per_wg <= limit ? r->S : (a.rows_per_wg == 16 ? lp_chunk_scenes(r, limit) : 0);
  if (per <= 0 || per * gps > infgen_layers_p_capacity()) return fail("infgen_decode_layers", "k_layers_p: batch does not fit the device's resident capacity");
  auto kern = a.rows_per_wg == 4 ? (sm.r24 ? k_layers_p<true, 4> : k_layers_p<false, 4>)
            : a.rows_per_wg == 8 ? (sm.r24 ? k_layers_p<true, 8> : k_layers_p<false, 8>)
                                 : (sm.r24 ? k_layers_p<true, 16> : k_layers_p<false, 16>);
  // A stream that is being captured into a HIP graph cannot take a cooperative launch or the cross-stream ordering below: the
  // (opt-in) graph modes keep the plain launch with the spin limit - their caller owns the GPU
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  const bool capturing = hipStreamIsCapturing((hipStream_t)stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
  // polls at a scene counter before the kernel traps: plain launches 2^24 (tens of seconds - a legitimate wait lasts as long as other
  // streams' kernels, milliseconds; only another process's k_layers_p on the same GPU can hold a workgroup off its CU for longer,
  // and a trap is better than a hung device), cooperative launches unlimited, captured launches 2^22; INFGEN_LP_SPIN_LIMIT overrides
  static const char* spin_str = getenv("INFGEN_LP_SPIN_LIMIT");
  static const unsigned spin_env = spin_str ? (unsigned)strtoul(spin_str, nullptr, 0) : 0u;
  a.spin_limit = capturing ? (1u << 22) : spin_str ? spin_env : (O().layers_p == 2 ? 0u : (1u << 24));
  // launches of different streams are ordered behind each other (two half-resident launches would wait for each other for ever):
  // every launch is followed by an event record on ITS stream, the next launch - whatever its stream - waits for that event first
  // (a wait on an event of the same stream is a no-op for the hardware queue).  No stream handle outlives the call: a caller may
  // destroy its stream at any time.
  std::unique_lock<std::mutex> lk(g_lp_mu, std::defer_lock);
  LpOrder& ord = g_lp_order[lp_current_device()];
  if (!capturing) {
    lk.lock();
    if (!ord.ev && hipEventCreateWithFlags(&ord.ev, hipEventDisableTiming) != hipSuccess)
      return fail("infgen_decode_layers", "event creation failed");
    if (ord.any && hipStreamWaitEvent((hipStream_t)stream, ord.ev, 0) != hipSuccess) {
      (void)hipGetLastError();
      ord.any = false;            // (the event is unusable: start over rather than fail every later launch)
    }
  }
  const bool lp_coop = O().layers_p == 2;
  for (int s0 = 0; s0 < r->S; s0 += per) {
    const int ns = r->S - s0 < per ? r->S - s0 : per;
    const int n_wg = ns * gps;
    a.row0 = s0 * r->A_cap;
    a.xcd_order = n_wg % (8 * gps) == 0 ? 1 : 0;
    ProfScope _ps(INFGEN_KID_EDGE_ATTN, stream);
    if (capturing || !lp_coop) {
      hipLaunchKernelGGL(kern, dim3(n_wg), dim3(512), 0, (hipStream_t)stream, a);
    } else {
      void* params[] = {&a};
      const hipError_t e = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(kern), dim3(n_wg), dim3(512), params, 0, (hipStream_t)stream);
      if (e != hipSuccess) {
        (void)hipGetLastError();
        if (s0 > 0) return fail("infgen_decode_layers", "k_layers_p: cooperative launch of a later chunk refused");
        g_lp_refused.store(true);          // this and every later call: the per-sublayer launches
        return LP_REFUSED;
      }
    }
  }
  if (!capturing) {
    ord.any = hipEventRecord(ord.ev, (hipStream_t)stream) == hipSuccess;
    if (!ord.any) (void)hipGetLastError();
    lk.unlock();
  }
  if (lp_trace) {          // synchronous dump of the last launch's stamps (diagnostic runs only)
    static int dumps = 0;
    if (dumps++ == lp_trace) {
      std::vector<unsigned long long> h(2048);
      (void)hipStreamSynchronize((hipStream_t)stream);
      (void)hipMemcpy(h.data(), trace_dev, 2048 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
      for (int i = 0; i < 1024 && (i == 0 || h[2 * i + 1]); ++i)
        fprintf(stderr, "[lp trace] %d %llu %llu\n", i, h[2 * i], i ? h[2 * i + 1] - h[2 * i - 1] : 0ull);
    }
  }
  return check_launch("infgen_decode_layers(k_layers_p)");
}

static int layers_core(const InfgenRollout* r, int c, int edgeless, void* stream, bool lp_sync_clear = true) {
  const int rows = r->S * r->A_cap;
  const StepMode sm = step_mode(r, rows, edgeless, stream);
  if (sm.lp && sm.fuse) {
    const int rc = layers_p_launch(r, c, sm, stream, lp_sync_clear);
    if (rc != LP_REFUSED) return rc;
    // the runtime refused the cooperative launch: the per-sublayer launches below, with the fused edge kernel (the step's rhat
    // rows are already in its format); later calls do not try again (layers_p_shape)
  }
  const bool overlap = sm.overlap, fuse = sm.fuse; const int r24 = sm.r24;
  const size_t slot = (size_t)(c % r->ring) * rows * D;
  const int L = r->num_layers;
  // prologue of the first (temporal) layer; every later layer's prologue is fused into the previous
  // layer's k_attn_post
  float* U = fuse ? nullptr : r->U;
  const float* Z = fuse ? nullptr : r->Z;
  const float* SIG = fuse ? nullptr : r->SIG;
  const int has_pos = fuse ? 0 : 1;                 // the fused edge kernel already added W'vr z + b' sigma to AGG
  // warm requests (tile.cuh: WarmArgs): an edge launch pulls what the node launch after it reads (the layer's post part without
  // W'vr, the next layer's pre part), a node launch the whole post part of the layer after it (W'vr first: the next edge launch)
  const int QB = 16384;                              // bytes of a quarter-matrix (split.cuh: QUARTER fp16 elements)
  // the edgeless column-0 chain through the fused kernels: a row without edges aggregates exactly zero (agg = z = sigma = 0, so
  // agg + W'vr z + b' sigma = +0 in every column) - AGG is zeroed once and the 18 edge launches (54 us each at 1024 scenes) are skipped
  const bool skip_edges = edgeless && fuse;
  if (skip_edges && hipMemsetAsync(r->AGG, 0, (size_t)rows * D * sizeof(float), (hipStream_t)stream) != hipSuccess)
    return fail("infgen_decode_layers", "memset failed");
  auto edge = [&](const float* pack, const float* Ks, const float* Vs, const InfgenEdgeBuf& e, int kv_once = 0, const float* next_pack = nullptr, int fmt = -1) {
    if (skip_edges) return 0;
    if (fuse && O().attn_mode != 0)
      warm_request(pack + AH_POST + 4 * (QB / 4), 48 * QB, next_pack ? next_pack + AH_PRE : nullptr, 16 * QB);
    return fuse ? edge_fused_launch(rows, r->Q, pack, Ks, Vs, e.off, e.cnt, e.src, e.rhat, r->AGG, r->A_cap, kv_once, stream, fmt < 0 ? r24 : fmt)
                : infgen_edge_attn(rows, r->Q, r->U, Ks, Vs, e.off, e.cnt, e.src, e.rhat, r->AGG, r->Z, r->SIG, stream);
  };
  RET_IF(infgen_attn_pre(r->X, rows, r->attn_t[0], 0, r->Q, U, r->ringK[0] + slot, r->ringV[0] + slot, stream));
  auto warm_post = [&](const float* pack) { if (fuse && O().attn_mode != 0) warm_request(pack + AH_POST, 52 * QB, nullptr, 0); };
  for (int i = 0; i < L; ++i) {
    // temporal: K/V of this column sit in the ring (they are the cached layer inputs' projections)
    RET_IF(edge(r->attn_t[i], r->ringK[i], r->ringV[i], r->et, 1, r->attn_m[i]));
    warm_post(r->attn_m[i]);
    RET_IF(infgen_attn_post_pre(r->X, rows, r->attn_t[i], r->AGG, Z, SIG, has_pos, r->attn_m[i], r->Q, U,
                                nullptr, nullptr, stream));
    // map -> agent (bipartite: K/V of the map tokens are per-scene constants)
    if (overlap && i == 0 && hipStreamWaitEvent((hipStream_t)stream, g_ev_m, 0) != hipSuccess)
      return fail("infgen_decode_layers", "join failed");
    RET_IF(edge(r->attn_m[i], r->mapK[i], r->mapV[i], r->em, 0, r->attn_a[i]));
    warm_post(r->attn_a[i]);
    RET_IF(infgen_attn_post_pre(r->X, rows, r->attn_m[i], r->AGG, Z, SIG, has_pos, r->attn_a[i], r->Q, U,
                                r->Ka, r->Va, stream));
    // agent <-> agent
    if (overlap && i == 0 && hipStreamWaitEvent((hipStream_t)stream, g_ev_a, 0) != hipSuccess)
      return fail("infgen_decode_layers", "join failed");
    RET_IF(edge(r->attn_a[i], r->Ka, r->Va, r->ea, 0, i + 1 < L ? r->attn_t[i + 1] : nullptr, sm.ra));
    if (i + 1 < L) {
      warm_post(r->attn_t[i + 1]);
      RET_IF(infgen_attn_post_pre(r->X, rows, r->attn_a[i], r->AGG, Z, SIG, has_pos, r->attn_t[i + 1], r->Q, U,
                                  r->ringK[i + 1] + slot, r->ringV[i + 1] + slot, stream));
    } else {
      RET_IF(infgen_attn_post(r->X, rows, r->attn_a[i], r->AGG, Z, SIG, has_pos, stream));
    }
    // test hook (InfgenRollout.tap_x): the residual stream after triple i
    if (r->tap_x && hipMemcpyAsync(r->tap_x + (size_t)i * rows * D, r->X, (size_t)rows * D * sizeof(float), hipMemcpyDeviceToDevice,
                                   (hipStream_t)stream) != hipSuccess)
      return fail("infgen_decode_layers", "tap copy failed");
  }
  return 0;
}

extern "C" int infgen_decode_layers(const InfgenRollout* r, int c, int edgeless, void* stream) {
  RET_IF(validate(r, "infgen_decode_layers"));
  OptScope _opts(r);
  ProfPhase _pp(edgeless ? t_prof_phase : 1);        // (the edgeless column-0 chain belongs to the prologue)
  const StepMode sm = step_mode(r, r->S * r->A_cap, edgeless, stream);
  const bool lp = sm.lp && sm.fuse && r->SIG != nullptr;      // (k_build_edges zeroes k_layers_p's counters: infgen_rollout_run)
  RET_IF(prepare_edges(r, c, edgeless, stream, true, false, nullptr, lp));
  return layers_core(r, c, edgeless, stream, !lp);
}

extern "C" int infgen_decode_step(const InfgenRollout* r, int t, void* stream) {
  RET_IF(validate(r, "infgen_decode_step"));
  OptScope _opts(r);
  ProfPhase _pp(1);
  const int rows = r->S * r->A_cap;
  const int c = 1 + t;
  if (t < 0 || c + 1 > r->T - 1) return fail("infgen_decode_step", "step beyond the column range");
  RET_IF(infgen_decode_layers(r, c, 0, stream));
  float* lg = (r->store_logits && r->logits) ? r->logits + (size_t)t * rows * r->token_size : nullptr;
  const bool sample = r->sample_k > 1 && r->sample_u && (lg || r->logits_scratch);
  if (sample && !lg) lg = r->logits_scratch;
  // (tmp2 is scratch of the raw-feature stage, free here: the per-row keys of the split arg-max)
  RET_IF(heads_impl(r->X, rows, r->tok_head_pack, r->st_head_pack, r->token_size, lg, r->next_token,
                    r->next_state, reinterpret_cast<unsigned long long*>(r->tmp2), stream));
  if (sample)
    RET_IF(infgen_sample_topk(lg, rows, r->token_size, r->sample_k, r->sample_u + (size_t)t * rows, r->next_token, stream));
  RET_IF(infgen_integrate(r, t, stream));
  RET_IF(infgen_raw_feature(r, c + 1, stream));
  return 0;
}

// The decode steps t0 .. t1 - 1.  With few rows (the Fourier embeddings of a step fit one multi-set launch) the step's tail is
// folded: the edge sets of column c + 1 are built right after the poses of column c + 1 exist, their embeddings and the rows'
// x_a_emb embedding share one launch, and k_integrate also decodes / clears the split arg-max keys, clears the edge totals and
// gathers the raw features - 8 dependent launches between the last sublayer of a step and the first of the next instead of 13
// (memset, k_heads, k_heads_finish, k_integrate, k_rawfeat_prep, k_fourier_h, k_mlpemb_h, memset, k_build_edges,
// k_fourier_h_multi, k_attn_hs ...).  Same arithmetic on the same data as infgen_decode_step (tests compare the two).
extern "C" int infgen_rollout_run(const InfgenRollout* r, int t0, int t1, void* stream) {
  RET_IF(validate(r, "infgen_rollout_run"));
  OptScope _opts(r);
  ProfPhase _pp(1);
  const int rows = r->S * r->A_cap;
  static const int no_fold = getenv("INFGEN_NO_TAIL_FOLD") ? atoi(getenv("INFGEN_NO_TAIL_FOLD")) : 0;
  const bool sample = r->sample_k > 1 && r->sample_u;
  const bool fold = !no_fold && t1 > t0 && fourier_multi_ok(rows, 0) && O().attn_mode != 0 && !sample && !r->first_new &&
                    r->et.total && r->em.total == r->et.total + 1 && r->ea.total == r->et.total + 2;
  if (!fold) {
    for (int t = t0; t < t1; ++t) RET_IF(infgen_decode_step(r, t, stream));
    return 0;
  }
  if (t0 < 0 || 1 + t1 > r->T - 1) return fail("infgen_rollout_run", "step beyond the column range");
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(r->tmp2);     // (free scratch under attn_mode != 0)
  if (hipMemsetAsync(keys, 0, (size_t)rows * sizeof(unsigned long long), (hipStream_t)stream) != hipSuccess)
    return fail("infgen_rollout_run", "memset failed");
  const StepMode sm0 = step_mode(r, rows, 0, stream);
  const bool lp_steps = sm0.lp && sm0.fuse && r->SIG != nullptr;       // the steps' sublayers run as k_layers_p launches
  // (their per-scene counters are zeroed by the kernel in front of every launch - k_build_edges before the first step, k_integrate
  // before the later ones - not by a fill: one launch less per step, and no memset node between the kernels of a captured graph)
  RET_IF(prepare_edges(r, 1 + t0, 0, stream, true, false, nullptr, lp_steps));
  for (int t = t0; t < t1; ++t) {
    const int c = 1 + t;
    RET_IF(layers_core(r, c, 0, stream, !lp_steps));
    float* lg = (r->store_logits && r->logits) ? r->logits + (size_t)t * rows * r->token_size : nullptr;
    bool split = false;
    RET_IF(heads_impl(r->X, rows, r->tok_head_pack, r->st_head_pack, r->token_size, lg, r->next_token, r->next_state, keys, stream,
                      true, &split));
    // (the last step keeps its edge totals, like infgen_decode_step: RolloutEngine.edge_totals() / overflow checks read them)
    RET_IF(integrate_impl(r, t, stream, split ? keys : nullptr, t + 1 < t1, true, t + 1 < t1 && lp_steps));
    if (t + 1 < t1) {
      RET_IF(prepare_edges(r, c + 1, 0, stream, false, true, split && integrate_groups(r) > 1 ? keys : nullptr));
    } else {          // after the last step only the raw feature of the new column is left (kept: the context's X stays what
                      // infgen_decode_step leaves)
      RET_IF(infgen_fourier_embed(r->raw2, 2, nullptr, rows, r->four_xa, r->cat, 128, r->fus_in + 128, 512, 0, stream));
    }
    RET_IF(raw_feature_fusion(r, stream));
  }
  return 0;
}

// ---------------------------------------------------------------------------------- scenario insertion
extern "C" int infgen_occupancy(const InfgenRollout* r, int c, float* occ, void* stream) {
  RET_IF(validate(r, "infgen_occupancy"));
  OccupancyArgs a{scene_of(r), c, r->grid_size, occ};
  hipLaunchKernelGGL(k_occupancy, dim3(r->S), dim3(NT), 0, (hipStream_t)stream, a);
  return check_launch("infgen_occupancy");
}

// occupancy vector AND its embedding (seed_agent_occ_embed, an MLPLayer 1961 -> 128 -> 128) in one launch
static int occupancy_embed_impl(const InfgenRollout* r, int c, float* occ, const float* embed_pack, float* emb, const int* active,
                                void* stream) {
  RET_IF(validate(r, "infgen_occupancy_embed"));
  if (r->grid_size > 2048) return fail("infgen_occupancy_embed", "grid larger than 2048 cells");
  OccEmbedArgs a{scene_of(r), c, r->grid_size, occ, embed_pack, emb, active};
  hipLaunchKernelGGL(k_occupancy_embed, dim3(r->S), dim3(NT), 0, (hipStream_t)stream, a);
  return check_launch("infgen_occupancy_embed");
}

extern "C" int infgen_occupancy_embed(const InfgenRollout* r, int c, float* occ, const float* embed_pack, float* emb, void* stream) {
  return occupancy_embed_impl(r, c, occ, embed_pack, emb, nullptr, stream);
}

extern "C" int infgen_point_edges(const InfgenRollout* r, int c, const int* centre_row, const int* active,
                                  int exclude_centre, int which, float r_agent, int k_agent, float r_map, int k_map,
                                  const InfgenEdgeBuf* ea, const InfgenEdgeBuf* em, void* stream) {
  RET_IF(validate(r, "infgen_point_edges"));
  hipStream_t s = (hipStream_t)stream;
  if (((which & 1) && hipMemsetAsync(ea->total, 0, sizeof(int), s) != hipSuccess) ||
      ((which & 2) && hipMemsetAsync(em->total, 0, sizeof(int), s) != hipSuccess))
    return fail("infgen_point_edges", "memset failed");
  PointEdgesArgs a{scene_of(r), c, centre_row, active, exclude_centre, which, r_agent, k_agent, r_map, k_map,
                   ebuf(*ea), ebuf(*em)};
  hipLaunchKernelGGL(k_point_edges, dim3(r->S), dim3(128), 0, s, a);
  return check_launch("infgen_point_edges");
}

static int insert_decide_impl(const InfgenRollout* r, int t, int force_enter, int max_new,
                              const float* lg_state, const float* lg_type, const float* shape, const float* lg_pos,
                              const float* occ, int* active, int* n_new, int* inserted, int* new_row,
                              float* new_shape, int* new_cell, int sample_k, const float* uniform, void* stream) {
  RET_IF(validate(r, "infgen_insert_decide"));
  if (sample_k > 16) return fail("infgen_insert_decide", "sample_k must be <= 16");
  if (sample_k > 1 && !uniform) return fail("infgen_insert_decide", "cell sampling needs uniforms");
  InsertDecideArgs a;
  a.st = scene_of(r); a.c = 1 + t; a.t = t; a.R = r->R; a.grid_size = r->grid_size; a.force_enter = force_enter;
  a.max_new = max_new; a.sample_k = sample_k; a.uniform = uniform;
  a.grid_xy = r->grid_xy; a.lg_state = lg_state; a.lg_type = lg_type; a.shape = shape;
  a.lg_pos = lg_pos; a.occ = occ; a.n_agents = const_cast<int*>(r->n_agents); a.type = const_cast<int*>(r->type);
  a.active = active; a.n_new = n_new; a.inserted = inserted; a.new_row = new_row; a.new_shape = new_shape;
  a.new_cell = new_cell; a.pred_traj = r->pred_traj; a.pred_head = r->pred_head; a.pred_state = r->pred_state;
  hipLaunchKernelGGL(k_insert_decide, dim3(r->S), dim3(64), 0, (hipStream_t)stream, a);
  return check_launch("infgen_insert_decide");
}

extern "C" int infgen_insert_decide(const InfgenRollout* r, int t, int force_enter, int max_new,
                                    const float* lg_state, const float* lg_type, const float* shape, const float* lg_pos,
                                    const float* occ, int* active, int* n_new, int* inserted, int* new_row,
                                    float* new_shape, int* new_cell, void* stream) {
  return insert_decide_impl(r, t, force_enter, max_new, lg_state, lg_type, shape, lg_pos, occ, active, n_new, inserted, new_row,
                            new_shape, new_cell, 1, nullptr, stream);
}

// the same with the reference's stochastic cell choice (softmax -> topk(insert_beam_size) -> multinomial, agent_decoder.py:1900-1904)
// in a reproducible form: inverse CDF over the sample_k most probable cells with uniform[s]; a sampled cell that is occupied spends
// the iteration and leaves the scene active (:1906-1909)
extern "C" int infgen_insert_decide_topk(const InfgenRollout* r, int t, int force_enter, int max_new,
                                         const float* lg_state, const float* lg_type, const float* shape, const float* lg_pos,
                                         const float* occ, int* active, int* n_new, int* inserted, int* new_row,
                                         float* new_shape, int* new_cell, int sample_k, const float* uniform, void* stream) {
  return insert_decide_impl(r, t, force_enter, max_new, lg_state, lg_type, shape, lg_pos, occ, active, n_new, inserted, new_row,
                            new_shape, new_cell, sample_k, uniform, stream);
}

extern "C" int infgen_insert_finalize(const InfgenRollout* r, int c, float angle_interval, const int* inserted,
                                      const int* new_row, const float* lg_heading, int n_heading, const float* offset,
                                      float* hv_ovr, void* stream) {
  RET_IF(validate(r, "infgen_insert_finalize"));
  InsertFinalizeArgs a{scene_of(r), c, angle_interval, inserted, new_row, lg_heading, n_heading, offset, hv_ovr};
  hipLaunchKernelGGL(k_insert_finalize, dim3(r->S), dim3(64), 0, (hipStream_t)stream, a);
  return check_launch("infgen_insert_finalize");
}

// ---------------------------------------------------------------------------------- teacher-forced forward
extern "C" int infgen_radius_edges(const InfgenRadiusEdges* r, const InfgenEdgeBuf* e, void* stream) {
  if (!r || !e) return fail("infgen_radius_edges", "null argument");
  if (r->n_q <= 0) return 0;
  if (r->K <= 0) return fail("infgen_radius_edges", "K must be positive");
  if (r->gap_rule < 0 || r->gap_rule > 2) return fail("infgen_radius_edges", "gap_rule must be 0, 1 or 2");
  RadiusEdgesArgs a{r->n_q, r->q_node, r->q_pt, r->q_c0, r->q_c1, r->q_self, r->q_pair_off, r->p_pos, r->p_head, r->p_inv,
                    r->c_pos, r->c_head, r->c_inv, r->c_ok, r->c_src, r->pair_ok, r->radius, r->K, r->gap_rule, r->index_diff,
                    EdgeBuf{e->off, e->cnt, e->src, e->raw, e->total, e->cap}, r->e_base};
  hipLaunchKernelGGL(k_radius_edges, dim3(ceil_div(r->n_q, 4)), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("infgen_radius_edges");
}

extern "C" int infgen_motion_features(const float* pos, const float* head, const int* state, const unsigned char* gap_mask,
                                      int rows, int T, float* out, void* stream) {
  if (rows <= 0 || T <= 0) return 0;
  MotionFeatArgs a{pos, head, state, gap_mask, rows, T, out};
  hipLaunchKernelGGL(k_motion_features, dim3(ceil_div(rows * T, 256)), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("infgen_motion_features");
}

// ---------------------------------------------------------------------------------- insertion sub-loop, sequenced here
namespace {
inline int round_up_i(int x, int m) { return (x + m - 1) / m * m; }
// MLPLayer pack (packing.pack_mlp_layer, K0 = 128): two descriptors of infgen_linear_multi
void mlp_layer_descs(const float* X, int rows, const float* pack, int n_out, float* hid, float* out, InfgenLinearDesc& d1,
                     InfgenLinearDesc& d2) {
  const int o_b0 = 128 * 128, o_w3 = o_b0 + 3 * 128, npad = round_up_i(n_out, 32);
  d1 = InfgenLinearDesc{X, 128, nullptr, rows, 128, pack, 128, pack + o_b0, 128, nullptr, nullptr, pack + o_b0 + 128, pack + o_b0 + 256,
                        1, hid, 128};
  d2 = InfgenLinearDesc{hid, 128, nullptr, rows, 128, pack + o_w3, npad, pack + o_w3 + 128 * npad, n_out, nullptr, nullptr, nullptr,
                        nullptr, 0, out, n_out};
}
int gather_rows(const float* src, const int* list, const int* mask, int n, int limit, float* dst, void* stream) {
  hipLaunchKernelGGL(k_gather_rows, dim3(ceil_div(n * 32, NT)), dim3(NT), 0, (hipStream_t)stream, src, list, mask, n, limit, dst);
  return check_launch("k_gather_rows");
}
int scatter_rows2(const float* src0, const float* src1, const int* list, const int* mask, int n, float* dst0, float* dst1,
                  void* stream) {
  hipLaunchKernelGGL(k_scatter_rows2, dim3(ceil_div(n * 32, NT)), dim3(NT), 0, (hipStream_t)stream, src0, src1, list, mask, n, dst0,
                     dst1);
  return check_launch("k_scatter_rows2");
}
// a row without edges has agg = z = sigma = 0: the positional part adds exactly nothing and is skipped
int edgeless(float* X, int rows, const float* pack, const InfgenInsertion* I, void* stream) {
  return infgen_attn_post(X, rows, pack, I->zero_agg, I->zero_z, I->zero_sig, 0, stream);
}
}  // namespace

extern "C" int infgen_insert_seed(const InfgenRollout* r, const InfgenInsertion* I, int t, int it, int riders, const float* uniform,
                                  void* stream) {
  RET_IF(validate(r, "infgen_insert_seed"));
  if (!I) return fail("infgen_insert_seed", "null insertion block");
  OptScope _opts(r);
  hipStream_t hs = (hipStream_t)stream;
  const int S = r->S, rows = r->S * r->A_cap, c = 1 + t, G = r->grid_size;
  RET_IF(occupancy_embed_impl(r, c, I->occ, I->occ_embed, I->occ_emb, I->active, stream));
  for (int i = 0; i < 3; ++i) RET_IF(infgen_attn_pre(I->occ_emb, S, I->attn_occ2sa[i], 1, nullptr, nullptr, I->Kocc[i], I->Vocc[i], stream));
  // edges into the seed node (the ego's pose): agents every iteration, map tokens once per step
  RET_IF(infgen_point_edges(r, c, r->av_index, I->active, 0, it == 0 ? 3 : 1, I->r_seed, 300, I->r_seed, 2048, &I->ea_s, &I->em_s, stream));
  RET_IF(infgen_fourier_embed(I->ea_s.raw, 3, I->ea_s.total, I->ea_s.cap, I->four_a2sa, nullptr, 0, I->ea_s.rhat, 128, 1, stream));
  if (it == 0) {
    RET_IF(infgen_fourier_embed(I->em_s.raw, 3, I->em_s.total, I->em_s.cap, I->four_pt2sa, nullptr, 0, I->em_s.rhat, 128, 1, stream));
    // every agent row passes every seed sublayer edgelessly; its K / V feed the a2sa layers: all rows once per step
    if (hipMemcpyAsync(I->Xc, r->X, (size_t)rows * D * sizeof(float), hipMemcpyDeviceToDevice, hs) != hipSuccess)
      return fail("infgen_insert_seed", "copy failed");
    for (int i = 0; i < 3; ++i) {
      RET_IF(edgeless(I->Xc, rows, I->attn_occ2sa[i], I, stream));
      // the post part of the map sublayer and the K / V projections of the agent sublayer in one launch
      RET_IF(infgen_attn_post_pre(I->Xc, rows, I->attn_pt2sa[i], I->zero_agg, I->zero_z, I->zero_sig, 0, I->attn_a2sa[i], nullptr, nullptr,
                                  I->Ksa[i], I->Vsa[i], stream));
      RET_IF(edgeless(I->Xc, rows, I->attn_a2sa[i], I, stream));
    }
  }
  // the seed rows [0, S) and, edgelessly, the rows the previous iteration appended [S, 2 S) (one slot per scene)
  const int R = riders ? 2 * S : S;
  RET_IF(gather_rows(I->f_seed, nullptr, nullptr, S, 1, I->XS, stream));
  if (riders) RET_IF(gather_rows(r->X, I->prev_row, I->prev_mask, S, rows, I->XS + (size_t)S * D, stream));
  RET_IF(infgen_attn_pre(I->XS, R, I->attn_occ2sa[0], 0, I->QS, nullptr, nullptr, nullptr, stream));
  for (int i = 0; i < 3; ++i) {
    RET_IF(infgen_edge_attn(S, I->QS, nullptr, I->Kocc[i], I->Vocc[i], I->occ_off, I->occ_cnt, I->occ_src, nullptr, I->AGGS, nullptr,
                            I->SIGS, stream));
    RET_IF(infgen_attn_post_pre(I->XS, R, I->attn_occ2sa[i], I->AGGS, I->ZS, I->SIGS, 0, I->attn_pt2sa[i], I->QS, I->US, nullptr, nullptr,
                                stream));
    // (the map edges of a step are built once: scenes that stopped inserting are masked out instead, their seed rows are not read)
    RET_IF(edge_attn_impl(S, I->QS, I->US, I->mapK[i], I->mapV[i], I->em_s.off, I->em_s.cnt, I->em_s.src, I->em_s.rhat, I->AGGS,
                          I->ZS, I->SIGS, 1, stream, I->active));
    RET_IF(infgen_attn_post_pre(I->XS, R, I->attn_pt2sa[i], I->AGGS, I->ZS, I->SIGS, 1, I->attn_a2sa[i], I->QS, I->US,
                                riders ? I->KN : nullptr, riders ? I->VN : nullptr, stream));
    if (riders) {
      RET_IF(scatter_rows2(I->KN + (size_t)S * D, I->VN + (size_t)S * D, I->prev_row, I->prev_mask, S, I->Ksa[i], I->Vsa[i], stream));
    }
    RET_IF(infgen_edge_attn_mode(S, I->QS, I->US, I->Ksa[i], I->Vsa[i], I->ea_s.off, I->ea_s.cnt, I->ea_s.src, I->ea_s.rhat, I->AGGS,
                                 I->ZS, I->SIGS, 1, stream));
    if (i < 2) RET_IF(infgen_attn_post_pre(I->XS, R, I->attn_a2sa[i], I->AGGS, I->ZS, I->SIGS, 1, I->attn_occ2sa[i + 1], I->QS, nullptr,
                                           nullptr, nullptr, stream));
    else RET_IF(infgen_attn_post(I->XS, R, I->attn_a2sa[i], I->AGGS, I->ZS, I->SIGS, 1, stream));
  }
  // the four seed heads in two launches
  InfgenLinearDesc d1[4], d2[4];
  const size_t hs_ = (size_t)S * 128;
  mlp_layer_descs(I->XS, S, I->head_state, 2, I->hid, I->lg_state, d1[0], d2[0]);
  mlp_layer_descs(I->XS, S, I->head_type, 3, I->hid + hs_, I->lg_type, d1[1], d2[1]);
  mlp_layer_descs(I->XS, S, I->head_shape, 3, I->hid + 2 * hs_, I->shape, d1[2], d2[2]);
  mlp_layer_descs(I->XS, S, I->head_pos, G, I->hid + 3 * hs_, I->lg_pos, d1[3], d2[3]);
  RET_IF(infgen_linear_multi(d1, 4, stream));
  RET_IF(infgen_linear_multi(d2, 4, stream));
  RET_IF(infgen_insert_decide_topk(r, t, I->force_enter, I->max_new, I->lg_state, I->lg_type, I->shape, I->lg_pos, I->occ, I->active,
                                   I->n_new, I->inserted, I->new_row, I->new_shape, I->new_cell, I->insert_k, uniform, stream));
  // hand-over to the host: did any scene insert, into which rows, which scenes go on?
  // (one copy when the caller laid the three arrays out back to back, as infgen_amd/engine.py does)
  if (I->new_row == I->inserted + S && I->active == I->inserted + 2 * S) {
    if (hipMemcpyAsync(I->host_dec, I->inserted, 3 * (size_t)S * sizeof(int), hipMemcpyDeviceToHost, hs) != hipSuccess)
      return fail("infgen_insert_seed", "hand-over copy failed");
  } else if (hipMemcpyAsync(I->host_dec, I->inserted, S * sizeof(int), hipMemcpyDeviceToHost, hs) != hipSuccess ||
             hipMemcpyAsync(I->host_dec + S, I->new_row, S * sizeof(int), hipMemcpyDeviceToHost, hs) != hipSuccess ||
             hipMemcpyAsync(I->host_dec + 2 * S, I->active, S * sizeof(int), hipMemcpyDeviceToHost, hs) != hipSuccess)
    return fail("infgen_insert_seed", "hand-over copy failed");
  return 0;
}

extern "C" int infgen_insert_heading(const InfgenRollout* r, const InfgenInsertion* I, int t, int h_ready, int riders, void* stream) {
  RET_IF(validate(r, "infgen_insert_heading"));
  if (!I) return fail("infgen_insert_heading", "null insertion block");
  OptScope _opts(r);
  hipStream_t hs = (hipStream_t)stream;
  const int S = r->S, rows = r->S * r->A_cap, c = 1 + t;
  // categorical embedding / shape of the new rows (agent_decoder.py:1949-1950, :1993): shape_emb is an MLPEmbedding with K0 = 3
  {
    const float* P = I->shape_emb;
    const int k0p = 8, o2 = mlpemb_off2(k0p), o3 = mlpemb_off3(k0p);
    RET_IF(infgen_linear(I->new_shape, 3, nullptr, S, 3, P, 128, P + k0p * 128, 128, nullptr, nullptr, P + k0p * 128 + 128,
                         P + k0p * 128 + 256, 1, I->t1, 128, stream));
    RET_IF(infgen_linear(I->t1, 128, nullptr, S, 128, P + o2, 128, P + o2 + 16384, 128, nullptr, nullptr, P + o2 + 16384 + 128,
                         P + o2 + 16384 + 256, 1, I->t2, 128, stream));
    RET_IF(infgen_linear(I->t2, 128, nullptr, S, 128, P + o3, 128, P + o3 + 16384, 128, nullptr, nullptr, nullptr, nullptr, 0, I->shp, 128,
                         stream));
  }
  InsertCatArgs ca{S, r->A_cap, I->inserted, I->new_row, r->type, I->type_a_emb, I->shp, I->new_shape, const_cast<float*>(r->cat_agent),
                   I->shape_all, I->new_local};
  hipLaunchKernelGGL(k_insert_cat, dim3(ceil_div(S * 32, NT)), dim3(NT), 0, hs, ca);
  RET_IF(check_launch("k_insert_cat"));
  RET_IF(infgen_raw_feature_rows(r, c, I->new_row, I->inserted, S, stream));
  // heading stage: the new row attends agents / map tokens within 10 m through the motion layers 0..2
  RET_IF(infgen_point_edges(r, c, I->new_local, I->inserted, 1, 3, I->r_a2sa, 24, I->r_pl2sa, 128, &I->ea_h, &I->em_h, stream));
  RET_IF(infgen_fourier_embed(I->ea_h.raw, 3, I->ea_h.total, I->ea_h.cap, r->four_a, nullptr, 0, I->ea_h.rhat, 128, 1, stream));
  RET_IF(infgen_fourier_embed(I->em_h.raw, 3, I->em_h.total, I->em_h.cap, r->four_m, nullptr, 0, I->em_h.rhat, 128, 1, stream));
  if (!h_ready) {
    if (hipMemcpyAsync(I->Xc, r->X, (size_t)rows * D * sizeof(float), hipMemcpyDeviceToDevice, hs) != hipSuccess)
      return fail("infgen_insert_heading", "copy failed");
    for (int i = 0; i < 3; ++i) {
      RET_IF(infgen_attn_post_pre(I->Xc, rows, r->attn_m[i], I->zero_agg, I->zero_z, I->zero_sig, 0, r->attn_a[i], nullptr, nullptr,
                                  I->Kh[i], I->Vh[i], stream));
      RET_IF(edgeless(I->Xc, rows, r->attn_a[i], I, stream));
    }
  }
  const int R = riders ? 2 * S : S;
  RET_IF(gather_rows(r->X, I->new_row, nullptr, S, rows, I->XS, stream));
  if (riders) RET_IF(gather_rows(r->X, I->pend_row, I->pend_mask, S, rows, I->XS + (size_t)S * D, stream));
  RET_IF(infgen_attn_pre(I->XS, R, r->attn_m[0], 0, I->QS, I->US, nullptr, nullptr, stream));
  for (int i = 0; i < 3; ++i) {
    RET_IF(infgen_edge_attn(S, I->QS, I->US, r->mapK[i], r->mapV[i], I->em_h.off, I->em_h.cnt, I->em_h.src, I->em_h.rhat, I->AGGS, I->ZS,
                            I->SIGS, stream));
    RET_IF(infgen_attn_post_pre(I->XS, R, r->attn_m[i], I->AGGS, I->ZS, I->SIGS, 1, r->attn_a[i], I->QS, I->US, riders ? I->KN : nullptr,
                                riders ? I->VN : nullptr, stream));
    if (riders) {
      RET_IF(scatter_rows2(I->KN + (size_t)S * D, I->VN + (size_t)S * D, I->pend_row, I->pend_mask, S, I->Kh[i], I->Vh[i], stream));
    }
    RET_IF(infgen_edge_attn(S, I->QS, I->US, I->Kh[i], I->Vh[i], I->ea_h.off, I->ea_h.cnt, I->ea_h.src, I->ea_h.rhat, I->AGGS, I->ZS,
                            I->SIGS, stream));
    if (i < 2) RET_IF(infgen_attn_post_pre(I->XS, R, r->attn_a[i], I->AGGS, I->ZS, I->SIGS, 1, r->attn_m[i + 1], I->QS, I->US, nullptr,
                                           nullptr, stream));
    else RET_IF(infgen_attn_post(I->XS, R, r->attn_a[i], I->AGGS, I->ZS, I->SIGS, 1, stream));
  }
  InfgenLinearDesc d1[2], d2[2];
  const size_t hs_ = (size_t)S * 128;
  mlp_layer_descs(I->XS, S, I->head_heading, I->n_heading, I->hid + 4 * hs_, I->lg_heading, d1[0], d2[0]);
  mlp_layer_descs(I->XS, S, I->head_offset, 2, I->hid + 5 * hs_, I->offset, d1[1], d2[1]);
  RET_IF(infgen_linear_multi(d1, 2, stream));
  RET_IF(infgen_linear_multi(d2, 2, stream));
  RET_IF(infgen_insert_finalize(r, c, I->angle_interval, I->inserted, I->new_row, I->lg_heading, I->n_heading, I->offset, I->hv_ovr, stream));
  RET_IF(infgen_raw_feature_rows(r, c, I->new_row, I->inserted, S, stream));
  // the rows of this iteration ride along in the next seed chain / heading stage
  hipLaunchKernelGGL(k_note_riders, dim3(ceil_div(S, NT)), dim3(NT), 0, hs, I->new_row, I->inserted, S, I->prev_row, I->prev_mask,
                     I->pend_row, I->pend_mask);
  return check_launch("k_note_riders");
}
