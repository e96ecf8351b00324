/*
 * infgen_hip.h — C ABI of libinfgen_hip.so: the MI355X (gfx950) implementation of InfGen's
 * closed-loop rollout hot path.
 *
 * The reference (OrangeSodahub/InfGen) is pure Python and has no FFI for this path; its native
 * arithmetic lives in third-party wheels.  Each entry point below replaces one group of those
 * calls (file:line are relative to the reference repository):
 *
 *   infgen_linear            torch.nn.Linear / LayerNorm / ReLU chains: MLPEmbedding, MLPLayer
 *                            (infgen/modules/layers.py:163-215)
 *   infgen_fourier_embed     FourierEmbedding.forward (infgen/modules/layers.py:142-160)
 *   infgen_attn_pre          AttentionLayer: prenorm + to_q/to_k/to_v (+ absorbed to_k_r)
 *                            (infgen/modules/layers.py:65-73,106-108)
 *   infgen_edge_attn         MessagePassing.propagate + torch_geometric.utils.softmax
 *                            (infgen/modules/layers.py:78-92,109)
 *   infgen_attn_post         AttentionLayer.update / to_out / post-norm / FFN
 *                            (infgen/modules/layers.py:74-75,94-99,110-112)
 *   infgen_heads             token_predict_head / state_predict_head + greedy arg-max
 *                            (infgen/modules/agent_decoder.py:2161-2167)
 *   infgen_map_graph         torch_cluster.radius_graph over map tokens + relative features
 *                            (infgen/modules/map_decoder.py:91-114)
 *   infgen_build_edges       _build_temporal_edge / _build_interaction_edge / _build_map2agent_edge
 *                            incl. torch_cluster.radius(max_num_neighbors=5)
 *                            (infgen/modules/agent_decoder.py:540-758)
 *   infgen_integrate         token -> contour -> pose, Attr_Tokenizer.encode_pos, invalid handling
 *                            (infgen/modules/agent_decoder.py:2168-2239, attr_tokenizer.py:77-89)
 *   infgen_raw_feature       _build_vector_a / x_a_emb / fusion_emb for one column
 *                            (infgen/modules/agent_decoder.py:426-509,2265-2287)
 *   infgen_decode_layers     the 6 x (temporal, map->agent, agent<->agent) stack on one column
 *                            (infgen/modules/agent_decoder.py:2123-2158)
 *   infgen_decode_step       one iteration of the rollout loop (infgen/modules/agent_decoder.py:1740-2301,
 *                            insertion disabled)
 *   infgen_point_edges, infgen_occupancy, infgen_insert_decide, infgen_insert_finalize, infgen_raw_feature_rows
 *                            the insertion sub-loop (infgen/modules/agent_decoder.py:1773-2105)
 *
 * and, around the path:
 *
 *   infgen_tokenize_agent, infgen_match_agent_tokens   TokenProcessor (infgen/datasets/preprocess.py:335-653)
 *   infgen_match_map_tokens, infgen_fetch_enterings    InfGen.match_token_map / _fetch_enterings
 *                                                      (infgen/model/infgen.py:918-984, 1008-1128)
 *   infgen_distance_to_nearest_object, infgen_time_to_collision, infgen_kinematic_features,
 *   infgen_distance_to_road_edge, infgen_placement_features   infgen/metrics/{interact,trajectory,map,placement}_features.py
 *   infgen_window_log_likelihood                       the scoring of LongMetric (infgen/metrics/compute_metrics.py:845-878)
 *
 * Conventions: every pointer is a DEVICE pointer (fp32 / int32 / uint8) borrowed for the duration
 * of the call; outputs are pre-allocated by the caller; `stream` is a hipStream_t; nothing
 * synchronises with the host; functions return 0 on success or a negative code and leave a message
 * retrievable with infgen_last_error() (thread-local).
 * Re-entrancy: every switch that changes what a launch computes (kernel family, arithmetic, rhat row format, overlap, padded-row lists)
 * is a field of InfgenOptions.  A rollout context carries its own block (InfgenRollout.opts, use = 1) and a thread may install one for
 * the operator-level entries (infgen_thread_options): contexts with different arithmetic coexist in one process.  The infgen_set_*
 * functions only edit the process-wide DEFAULT block that contexts / threads without their own inherit.  Environment variables
 * (INFGEN_*) are tuning thresholds and diagnostics of the launch shapes (which kernel variant from how many rows); each is read once
 * per process and none changes the arithmetic.
 */
#ifndef INFGEN_HIP_H_
#define INFGEN_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define INFGEN_MAX_LAYERS 8

/* offsets / sizes (floats) of the packed weight layouts, see infgen_amd/csrc/layout.h */
enum {
  INFGEN_Q_ATTN_PACK_SIZE = 0,
  INFGEN_Q_FOURIER_PACK_SIZE_N2 = 1,
  INFGEN_Q_FOURIER_PACK_SIZE_N3 = 2,
  INFGEN_Q_FOURIER_PACK_SIZE_N4 = 3,
  INFGEN_Q_TILE_ROWS = 4,
  INFGEN_Q_EDGE_ATTN_CAP = 5,
  INFGEN_Q_MAX_AGENTS = 6,
  INFGEN_Q_ABI_VERSION = 7,
  INFGEN_Q_SIZEOF_ROLLOUT = 8,
};
int infgen_layout_query(int what);
/* offset (floats) of a named field inside the AttentionLayer / Fourier pack; -1 if unknown */
int infgen_attn_pack_offset(const char* field);
int infgen_fourier_pack_offset(const char* field, int n_dims, int dim);
const char* infgen_last_error(void);

typedef struct InfgenEdgeBuf {
  int* off;      /* [rows] */
  int* cnt;      /* [rows] */
  int* src;      /* [cap] */
  float* raw;    /* [cap][4] */
  float* rhat;   /* [cap][128] fp32 as written by infgen_fourier_embed.  The three sets of an InfgenRollout (et / em / ea) are scratch of
                  * infgen_decode_layers: between its Fourier and edge launches it may keep the rows in a packed 24-bit form (384 B
                  * per row in the same buffer) - do not read them back */
  int* total;    /* [1] */
  int cap;
  int _pad;
} InfgenEdgeBuf;

/* kernel-selection switches and the padded-row bookkeeping of one rollout context (see InfgenRollout.opts) */
typedef struct InfgenOptions {
  int use;              /* 0: the process-wide defaults of the infgen_set_* functions apply (single-context callers) */
  int attn_mode;        /* infgen_set_attn_mode */
  int gemm_terms;       /* infgen_set_gemm_terms */
  int fourier_mode;     /* infgen_set_fourier_mode */
  int edge_fuse;        /* infgen_set_edge_fuse */
  int edge_loop;        /* infgen_set_edge_loop */
  int overlap;          /* infgen_set_overlap (the side stream itself is shared: keep 0 when contexts run concurrently) */
  int row_group_margin; /* rows a decode step may append (infgen_set_row_limits) */
  int layers_p;         /* infgen_set_layers_p: 1 = small launches run a decode step's sublayers in one launch (k_layers_p) */
  int rhat_format;      /* infgen_set_rhat_format: 0 (default) fp32 rows of the normalised relative-position embedding between the
                         * Fourier and the edge kernels, 1 packed 24-bit rows (384 B, 2^-17 relative: a reduced-precision mode) */
  int edge_kernel;      /* infgen_set_edge_kernel: lane layout of the fused edge kernel's loop - 0: lane = two columns of every row
                         * (k_edge_fused), 1 (default): lane = (head, 16-column slice) with the rhat rows staged through LDS
                         * (k_edge_fused3) for launches beyond 4 k rows, 2: the same at every size.  Same results up to fp32
                         * summation order; packed 24-bit rhat rows (rhat_format 1) always take k_edge_fused. */
  int _pad0;
  const int* row_groups; const int* n_row_groups;   /* optional list of the 16-row groups that hold agents (infgen_set_row_groups) */
} InfgenOptions;

typedef struct InfgenRollout {
  /* sizes / hyper-parameters */
  int S, A_cap, T, M_cap, W, ring, R, token_size, grid_size, num_layers;
  int force_valid;      /* disable_insertion: predicted states are overridden with 'valid' */
  int store_logits;     /* 1: logits[t] is written every step */
  float r_map, r_agent; /* pl2a_radius, a2a_radius */
  /* scene state, column-major per scene: [S][T][A_cap] */
  const int* n_agents; const int* n_map; const int* av_index;
  float* pos; float* head; int* state; int* token; int* grid;
  uint8_t* tmask; uint8_t* imask; uint8_t* catflag;
  const int* type; int* bos;
  const float* map_pos; const float* map_orient;
  /* packed weights */
  const float* attn_t[INFGEN_MAX_LAYERS]; const float* attn_m[INFGEN_MAX_LAYERS]; const float* attn_a[INFGEN_MAX_LAYERS];
  const float* four_t; const float* four_m; const float* four_a; const float* four_xa;
  const float* fusion_pack;    /* MLPEmbedding(512): P(512,128) b ln | P(128,128) b ln | P(128,128) b */
  const float* tok_head_pack; const float* st_head_pack;
  const float* tok_tab; const float* grid_tab; const float* state_emb;
  const float* cat_agent; const float* cat_seed; const float* vocab; const float* grid_xy;
  /* caches */
  float* ringK[INFGEN_MAX_LAYERS]; float* ringV[INFGEN_MAX_LAYERS];         /* [ring][rows][128] */
  const float* mapK[INFGEN_MAX_LAYERS]; const float* mapV[INFGEN_MAX_LAYERS]; /* [S*M_cap][128] */
  /* scratch */
  float* X; float* Q; float* U; float* Ka; float* Va; float* AGG; float* Z; float* SIG;
  InfgenEdgeBuf et, em, ea;
  float* raw2; float* cat; float* fus_in; float* tmp1; float* tmp2;
  int* next_token; int* next_state;
  float* logits;               /* optional [steps][rows][token_size] */
  const int* teacher_token; const int* teacher_state;   /* optional [S][T][A_cap] */
  /* outputs */
  float* pred_traj; float* pred_head; float* pred_state;   /* [S][A_cap][R](x2) */
  /* scenario insertion (optional, NULL when unused): rows >= first_new[s] inserted in the current step use
   * hv_ovr[s] as their head vector during the motion stage (agent_decoder.py:2083) */
  const int* first_new; const float* hv_ovr;
  /* reproducible top-k sampling (optional): sample_k > 1 -> every step draws the motion token by inverse CDF over
   * the sample_k most probable tokens with the uniforms sample_u[t][row]; needs logits_scratch [rows][token_size] */
  int sample_k; int _pad1;
  const float* sample_u; float* logits_scratch;
  /* per-context options (re-entrancy): with opts.use != 0 the rollout-level entries (infgen_decode_layers / _step /
   * infgen_rollout_run, infgen_raw_feature*, infgen_build_edges) take every switch from here and never read the process-wide
   * defaults of the infgen_set_* functions - two contexts of one process may differ and run from different host threads */
  InfgenOptions opts;
  /* optional [S][T][A_cap]: grid cell of the teacher-forced state per (column, row); entries < -1 mean "none" (the cell is
   * computed).  Long teacher-forced comparisons use it where a pose sits on a cell border (arg-min of encode_pos flips with
   * the last bits of the pose) */
  const int* teacher_grid;
  /* optional [32][128], written by infgen_fourier_last_dim_table(four_t, 4, ..) under the arithmetic (infgen_set_gemm_terms /
   * opts.gemm_terms) the context runs with: the temporal edges' time-gap input takes the values -1 .. -16 only, so its branch
   * of r_t_emb (layers.py:150-153, mlps[3]) is looked up instead of evaluated per edge.  NULL: evaluated per edge. */
  const float* four_t_dt;
  /* optional [S][T][A_cap][2] / [S][T][A_cap]: pose of the teacher-forced state per (column, row).  When given, the pose a decode step
   * stores for its new column is the teacher's (rows the teacher marks invalid excepted); the step's OWN result still goes to
   * pred_traj / pred_head.  Every step then starts from the same geometry as the reference run, so a long comparison has no
   * accumulated pose drift and no radius / first-K decision can flip: logits are comparable row by row with a maximum, and the
   * one-step pose update is checked separately (tests/test_baseline_shapes_gpu.py) */
  const float* teacher_pos; const float* teacher_head;
  /* optional [S]: the map-side scene of agent-side scene s (NULL: s itself).  Several scenes may share ONE set of map tokens - n_map,
   * map_pos / map_orient, the rows [slot * M_cap, (slot + 1) * M_cap) of mapK / mapV (and of InfgenInsertion.mapK / mapV) are then
   * indexed by slot = map_scene[s]: the n rollouts of one scene (reference infgen/model/infgen.py:704-706, inference_no_map
   * infgen_decoder.py:132-134) encode their map once and keep one copy of its K / V rows */
  const int* map_scene;
  /* optional [num_layers][S * A_cap][128] (test hook): infgen_decode_layers copies the residual stream X there after every
   * (temporal, map -> agent, agent <-> agent) triple - the reference's feat_a after a2a_attn_layers[i] for the current column
   * (infgen/modules/agent_decoder.py:2133-2158).  A context with tap_x runs the per-sublayer launches (k_layers_p keeps the
   * stream in registers across the triples) */
  float* tap_x;
} InfgenRollout;

int infgen_linear(const float* X, int ldx, const int* gather, int rows, int K,
                  const float* Wp, int Np, const float* bias, int N,
                  const float* pre_g, const float* pre_b, const float* post_g, const float* post_b, int relu,
                  float* Y, int ldy, void* stream);

/* n (<= 6) independent infgen_linear calls in one launch (the heads of the insertion sub-loop); fields as infgen_linear's
 * arguments */
typedef struct InfgenLinearDesc {
  const float* X; int ldx; const int* gather; int rows; int K;
  const float* Wp; int Np; const float* bias; int N;
  const float* pre_g; const float* pre_b; const float* post_g; const float* post_b; int relu;
  float* Y; int ldy;
} InfgenLinearDesc;
int infgen_linear_multi(const InfgenLinearDesc* desc, int n, void* stream);
/* MLPEmbedding.forward (infgen/modules/layers.py:163-192) for inputs of K0 = 128 j <= 512 columns: Linear LN ReLU Linear LN ReLU
 * Linear, pack = infgen_amd.packing.pack_mlp_embedding (fp32 stages + split section).  attn_mode != 0: one launch on the fp16
 * split (k_mlpemb_h); attn_mode 0: three fp32-MFMA launches through the scratch arrays tmp1 / tmp2 [rows][128]. */
int infgen_mlp_embedding(const float* X, int ldx, int rows, int K0, const float* pack, float* tmp1, float* tmp2,
                         float* Y, int ldy, void* stream);
/* row-wise LayerNorm over 128 columns (torch.nn.LayerNorm, eps 1e-5); gamma == NULL: affine-free */
int infgen_layernorm(const float* X, int rows, const float* gamma, const float* beta, float* Y, void* stream);
int infgen_fourier_embed(const float* raw, int n_dims, const int* count_dev, int e_cap, const float* pack,
                         const float* cat, int ldcat, float* out, int ldo, int normalize, void* stream);
/* FourierEmbedding whose LAST continuous input only takes the values 0, -1, .., -31 (the time gap of the temporal edges):
 * infgen_fourier_last_dim_table writes table[32][128], row k = mlps[n-1](-k) without its bias, with the arithmetic of the
 * current gemm_terms; infgen_fourier_embed_tab evaluates dims 0 .. n-2 per row and adds row (int)(-raw[e][n-1]) of the table
 * (same result as infgen_fourier_embed up to the fp32 summation order of the per-dim branches).  Split kernel only. */
int infgen_fourier_last_dim_table(const float* pack, int n_dims, float* table, void* stream);
int infgen_fourier_embed_tab(const float* raw, int n_dims, const int* count_dev, int e_cap, const float* pack,
                             const float* table, float* out, int ldo, int normalize, void* stream);
/* arithmetic of infgen_fourier_embed: 1 (default) = fp16 MFMA with a three-term hi/lo split of both operands and fp32
 * accumulation (operands to 2^-23, round to nearest; measured at least as close to fp64 as mode 0: tests/test_precision_gpu.py; 5.3x the
 * fp32 matrix rate), 0 = fp32-input MFMA.  Process-wide default (InfgenOptions.fourier_mode per context / thread). */
int infgen_set_fourier_mode(int mode);
/* the switch for infgen_attn_pre / infgen_attn_post / infgen_attn_post_pre: 0 fp32 MFMA, 1 fp16 split, 2 (default) by
 * row count (split from ~10 k rows, where its one-workgroup-per-CU tiles fill the chip) */
int infgen_set_attn_mode(int mode);
int infgen_attn_pre(const float* X, int rows, const float* pack, int use_src_ln,
                    float* Q, float* U, float* K, float* V, void* stream);
int infgen_edge_attn(int rows, const float* Q, const float* U, const float* Ksrc, const float* Vsrc,
                     const int* off, const int* cnt, const int* src, const float* rhat,
                     float* AGG, float* Z, float* SIG, void* stream);
/* The edge side of one sublayer for 16-row tiles with the absorbed query U and the positional aggregate Z kept on chip
 * (k_edge_fused): u = q W'_kr on the matrix pipe -> LDS, the edge loop of infgen_edge_attn, then
 * AGG = sum_e a_e v_src + W'_vr z + b' sigma on the matrix pipe.  `pack` is the layer's attention pack; the node-side
 * entries that follow take this AGG with has_pos = 0 and no Z / SIG / U.  Replaces, like infgen_edge_attn, the reference's
 * MessagePassing.propagate + softmax (infgen/modules/layers.py:78-92,109).  infgen_set_edge_fuse(0) makes
 * infgen_decode_layers fall back to the unfused sequence (default 1). */
int infgen_edge_attn_fused(int rows, const float* Q, const float* pack, const float* Ksrc, const float* Vsrc,
                           const int* off, const int* cnt, const int* src, const float* rhat,
                           float* AGG, void* stream);
/* The same two operators with the normalised relative-position rows in a packed 24-bit form: per row 128 x 16 bit (bits 31..16 of
 * the fp32 values) followed by 128 x 8 bit (bits 15..8), 384 bytes, values rounded to nearest even at bit 8 (relative error 2^-17).
 * infgen_fourier_embed_r24 (normalize = 1, no categorical sum; split Fourier kernel only, i.e. infgen_set_fourier_mode != 0) writes
 * e_cap rows of 384 bytes to `out`; infgen_edge_attn_fused_r24 reads them.  infgen_decode_layers uses the form for its own sets. */
int infgen_fourier_embed_r24(const float* raw, int n_dims, const int* count_dev, int e_cap, const float* pack, void* out, void* stream);
int infgen_edge_attn_fused_r24(int rows, const float* Q, const float* pack, const float* Ksrc, const float* Vsrc,
                               const int* off, const int* cnt, const int* src, const void* rhat24, float* AGG, void* stream);
/* Which of the two row formats infgen_decode_layers / infgen_rollout_run keep their own edge sets' rhat rows in between the Fourier
 * and the edge launches: 0 (default) fp32 - the reference's arithmetic (infgen/modules/layers.py:61-113 is fp32 end to end) -,
 * 1 the packed 24-bit rows above (-25 % of the rhat bytes, ~1 % of a rollout; outside the fp32 contract, a named secondary leg of
 * bench.py).  Process-wide default; a context with opts.use != 0 takes InfgenOptions.rhat_format. */
int infgen_set_rhat_format(int format);
int infgen_set_edge_kernel(int kernel);
int infgen_set_edge_fuse(int mode);
/* the process-wide defaults (what the infgen_set_* functions edited so far), e.g. to seed a context's own InfgenOptions */
int infgen_get_options(InfgenOptions* out);
/* Re-entrancy of the OPERATOR-level entries (infgen_fourier_embed, infgen_attn_*, infgen_edge_attn*, infgen_heads, infgen_linear ...):
 * they take their switches from the calling THREAD's option block when one is installed, else from the process-wide defaults.
 * infgen_thread_options(o) installs a copy of *o for the calling thread (o == NULL removes it); a rollout-level entry called
 * with a context whose opts.use != 0 still takes the context's block.  Two engines driven from two host threads therefore never
 * see each other's settings, whatever the infgen_set_* defaults are (infgen_amd/engine.py installs the engine's block around its
 * prologue).  infgen_get_effective_options returns what an operator-level call from this thread would use right now. */
int infgen_thread_options(const InfgenOptions* o);
int infgen_get_effective_options(InfgenOptions* out);
/* diagnostics: resident workgroups per CU the runtime reports for k_edge_fused; a plain streaming read of n_bytes with 8 or
 * 16 bytes per lane for calibrating rocprofv3 FETCH_SIZE (tools/calibrate_fetch.sh; out: 2048 floats) */
int infgen_edge_fused_occupancy(void);
int infgen_debug_stream_read(const float* p, unsigned long long n_bytes, int width, float* out, void* stream);
/* edges per trip of k_edge_fused's edge loop (their K / V / rhat rows are requested together): 4, 6 (default) or 8 */
int infgen_set_edge_loop(int variant);
/* 1: infgen_decode_layers runs the Fourier embeddings of the map and agent edge sets on an internal side stream, overlapped
 * with the first temporal / map sublayers on the caller's stream (joined with events before their first use) */
int infgen_set_overlap(int mode);
/* 1 (default; INFGEN_LAYERS_P=0 in the environment: off): launches of up to 256 16-row groups run ALL 18 sublayers of a decode step
 * in one launch of k_layers_p (one resident workgroup per group, wave = feature tile; the scene's groups meet at a counter before
 * each agent sublayer) instead of 36 launches of k_edge_fused + k_attn_hs - same operators (infgen/modules/layers.py:61-113),
 * rounding-level differences only.  Needs fused edge attention (infgen_set_edge_fuse != 0) and the split GEMM kernels; otherwise
 * the per-sublayer launches run (a row-group list of an insertion context is ignored by this kernel: it visits every group).
 * Process-wide default like the other infgen_set_*: a context with opts.use != 0 takes InfgenOptions.layers_p instead.
 * Concurrency: a launch never exceeds the workgroups the CURRENT device keeps resident for this kernel (occupancy query x CUs; also on
 * partitions with fewer CUs), the launches of different streams of one process are ordered behind each other by the library (an event
 * recorded behind every launch on its own stream - no stream handle is kept), and the wait at the counters is bounded only by a
 * trap after 2^24 polls (tens of seconds; INFGEN_LP_SPIN_LIMIT) - contexts may run concurrently on several streams or host threads,
 * next to other kernels (tests/test_rollout_gpu.py).
 * mode 2: the same through hipLaunchCooperativeKernel (device-wide cooperative queue: also safe next to ANOTHER PROCESS that runs
 * such kernels on the same GPU; ~25 us more per launch; a runtime that refuses the launch gets the per-sublayer launches from
 * then on).  (A stream that is being captured into a HIP graph takes a plain launch with a bounded wait: graph replay is an
 * opt-in mode whose caller owns the GPU.)
 * infgen_layers_p_capacity: workgroups one such launch may have on the current device (0: the kernel is unavailable here). */
int infgen_set_layers_p(int mode);
int infgen_layers_p_capacity(void);
/* same with the kernel variant forced: wide = 1 -> one 8-wave workgroup per destination (long edge lists, few rows),
 * wide = 0 -> one wave per destination; infgen_edge_attn picks wide when rows <= 256 */
int infgen_edge_attn_mode(int rows, const float* Q, const float* U, const float* Ksrc, const float* Vsrc,
                          const int* off, const int* cnt, const int* src, const float* rhat,
                          float* AGG, float* Z, float* SIG, int wide, void* stream);
int infgen_attn_post(float* X, int rows, const float* pack, const float* AGG, const float* Z, const float* SIG,
                     int has_pos, void* stream);
/* infgen_attn_post followed, on the same rows, by the NEXT layer's infgen_attn_pre (one launch) */
int infgen_attn_post_pre(float* X, int rows, const float* pack, const float* AGG, const float* Z, const float* SIG,
                         int has_pos, const float* next_pack, float* nQ, float* nU, float* nK, float* nV, void* stream);
int infgen_heads(const float* X, int rows, const float* tok_pack, const float* st_pack, int token_size,
                 float* logits, int* next_token, int* next_state, void* stream);
/* out[k][:] = tab0[idx0[k]] + ((tab1[idx1[k]] + tab2[idx2[k]]) + tab3[idx3[k]]), rows of 128 floats, int64 indices (clamped to
 * the table sizes n0 .. n3): the map-token embedding plus the sum of its three nn.Embedding rows in one pass
 * (infgen/modules/map_decoder.py:87-89 and the token table of :70-86), in torch's summation order */
int infgen_embedding_sum4(const float* tab0, const long long* idx0, int n0, const float* tab1, const long long* idx1, int n1,
                          const float* tab2, const long long* idx2, int n2, const float* tab3, const long long* idx3, int n3,
                          int rows, float* out, void* stream);
/* compacted CSR: `total` (device int) receives the edge count; rows whose edges would exceed `cap`
 * get cnt = 0 and the caller must retry with a larger buffer when *total > cap */
int infgen_map_graph(int S, int M_cap, const int* n_map, const float* pos, const float* orient, float radius,
                     int max_nbr, int* off, int* cnt, int* src, float* raw, int* total, int cap, void* stream);

/* checks a freshly built context (sizes, required pointers) and examines its attention packs' headers (64-byte device -> host
 * copies: synchronous; k_layers_p needs the LayerNorm bounds of header slots 10..13, version slot 14 - contexts whose packs lack
 * them take the per-sublayer launches).  Optional: without it a pack is examined when its device address is first seen, which
 * goes wrong if that address held another pack earlier in the process. */
int infgen_rollout_validate(const InfgenRollout* r);
int infgen_build_edges(const InfgenRollout* r, int c, int edgeless, void* stream);
int infgen_integrate(const InfgenRollout* r, int t, void* stream);
int infgen_raw_feature(const InfgenRollout* r, int col, void* stream);
/* the same for the rows row_list[k] with row_mask[k] != 0 only (k < n; the rows a sub-loop iteration of the insertion appended) */
int infgen_raw_feature_rows(const InfgenRollout* r, int col, const int* row_list, const int* row_mask, int n, void* stream);
int infgen_decode_layers(const InfgenRollout* r, int c, int edgeless, void* stream);
int infgen_decode_step(const InfgenRollout* r, int t, void* stream);
/* steps t0 .. t1-1 back to back (one host call per rollout) */
int infgen_rollout_run(const InfgenRollout* r, int t0, int t1, void* stream);

/* reproducible stand-in for softmax -> topk(k) -> multinomial (agent_decoder.py:2162-2163,2194-2195): the k most
 * probable tokens, inverse-CDF over their probabilities with a caller-supplied uniform per row */
int infgen_sample_topk(const float* logits, int rows, int n, int k, const float* uniform, int* token, void* stream);

/* ---- scenario insertion (reference agent_decoder.py:1773-2105); the sub-loop is sequenced by the host ----
 *   infgen_occupancy        one-hot sum of the grid tokens of column c (:1852-1854)
 *   infgen_point_edges      _build_a2sa_edge / _build_map2sa_edge for one query point per scene (:760-904):
 *                           first-K agents / map tokens (ascending index) within a radius of centre_row's pose
 *   infgen_insert_decide    seed heads -> enter / type / shape / cell, occupied-cell rejection, row append (:1883-1999);
 *                           inserted[s] = 1 row appended, 0 none, -1 the scene has no free row left (A_cap reached: the
 *                           caller must re-run with more head-room - nothing is dropped silently)
 *   infgen_insert_finalize  heading token + xy offset of the new row (:2060-2074) */
int infgen_occupancy(const InfgenRollout* r, int c, float* occ, void* stream);
/* the same plus seed_agent_occ_embed of it (MLPLayer pack of infgen_amd.packing.pack_mlp_layer) -> emb [S][128], one launch */
int infgen_occupancy_embed(const InfgenRollout* r, int c, float* occ, const float* embed_pack, float* emb, void* stream);
int infgen_point_edges(const InfgenRollout* r, int c, const int* centre_row, const int* active, int exclude_centre,
                       int which /* bit0 agents, bit1 map */, float r_agent, int k_agent, float r_map, int k_map,
                       const InfgenEdgeBuf* ea, const InfgenEdgeBuf* em, void* stream);
int infgen_insert_decide(const InfgenRollout* r, int t, int force_enter, int max_new,
                         const float* lg_state, const float* lg_type, const float* shape, const float* lg_pos,
                         const float* occ, int* active, int* n_new, int* inserted, int* new_row, float* new_shape,
                         int* new_cell, void* stream);
/* the same with the stochastic cell choice of the reference (:1900-1904) made reproducible: inverse CDF over the sample_k (<= 16)
 * most probable cells with uniform[s] in [0, 1); an occupied sampled cell spends the iteration, the scene stays active (:1906-1909) */
int infgen_insert_decide_topk(const InfgenRollout* r, int t, int force_enter, int max_new,
                              const float* lg_state, const float* lg_type, const float* shape, const float* lg_pos,
                              const float* occ, int* active, int* n_new, int* inserted, int* new_row, float* new_shape,
                              int* new_cell, int sample_k, const float* uniform, void* stream);
int infgen_insert_finalize(const InfgenRollout* r, int c, float angle_interval, const int* inserted,
                           const int* new_row, const float* lg_heading, int n_heading, const float* offset,
                           float* hv_ovr, void* stream);

/* ---- the insertion sub-loop of one decode step, sequenced in the library (reference agent_decoder.py:1773-2105) ----
 * One iteration = infgen_insert_seed (occupancy + its embedding, edges into the seed node, the nine seed sublayers with the
 * previous iteration's new rows riding along edgelessly, the seed heads, the decision) -> the caller reads the per-scene
 * decisions (host_dec, pinned memory, valid once the stream reached this point) -> if any scene inserted,
 * infgen_insert_heading (categorical embedding and raw feature of the new rows, their 10 m neighbourhood through motion layers
 * 0..2, heading token + offset, final raw feature).  ~60 launches per iteration behind two calls; no host synchronisation inside.
 * All arrays are caller-allocated device memory (sizes in comments; S scenes, rows = S * A_cap, G grid cells). */
typedef struct InfgenInsertion {
  /* weights */
  const float* attn_occ2sa[3]; const float* attn_pt2sa[3]; const float* attn_a2sa[3];
  const float* four_a2sa; const float* four_pt2sa;
  const float* head_state; const float* head_type; const float* head_shape; const float* head_pos; const float* head_heading;
  const float* head_offset; const float* occ_embed; const float* shape_emb; const float* type_a_emb; const float* f_seed;
  /* per-step caches */
  float* occ; float* occ_emb;                 /* [S][G], [S][128] */
  float* Kocc[3]; float* Vocc[3];             /* [S][128] */
  const float* mapK[3]; const float* mapV[3]; /* [S * M_cap][128]: K / V of the map tokens for the pt2sa layers */
  float* Ksa[3]; float* Vsa[3]; float* Kh[3]; float* Vh[3];   /* [rows][128] */
  float* Xc;                                  /* [rows][128] */
  float* zero_agg; float* zero_z; float* zero_sig;            /* [rows][128], [rows][8][128], [rows][8]: stay zero */
  /* seed / new-row work arrays, 2 S rows each (rows [S, 2 S): the riders, one slot per scene) */
  float* XS; float* QS; float* US; float* AGGS; float* ZS; float* SIGS; float* KN; float* VN;
  InfgenEdgeBuf ea_s, em_s, ea_h, em_h;       /* off / cnt: [S] */
  const int* occ_off; const int* occ_cnt; const int* occ_src;   /* the one occupancy edge of every seed row */
  int* active; int* n_new; int* inserted; int* new_row; int* new_cell; int* new_local; float* new_shape;   /* [S] (new_shape [S][3]) */
  int* prev_row; int* prev_mask;              /* [S]: rows the previous iteration appended (seed-chain riders) */
  int* pend_row; int* pend_mask;              /* [S]: rows of the last heading stage (heading-chain riders) */
  float* hv_ovr; float* shape_all;            /* [S][2], [rows][3] */
  float* hid; float* lg_state; float* lg_type; float* shape; float* lg_pos; float* lg_heading; float* offset;
                                              /* [6][S][128], [S][2], [S][3], [S][3], [S][G], [S][n_heading], [S][2] */
  float* t1; float* t2; float* shp;           /* [S][128] each */
  int* host_dec;                              /* PINNED HOST memory [3][S]: inserted, new_row, active */
  float r_seed, r_a2sa, r_pl2sa, angle_interval;
  int n_heading, force_enter, insert_k, max_new;
} InfgenInsertion;
/* it: iteration of the step (0: map edges of the seed are built and the agents' edgeless chains are computed); riders != 0: the
 * previous iteration appended rows (prev_row / prev_mask); uniform: [S] for the cell draw when insert_k > 1 */
int infgen_insert_seed(const InfgenRollout* r, const InfgenInsertion* I, int t, int it, int riders, const float* uniform, void* stream);
/* h_ready == 0: the agents' edgeless chain through motion layers 0..2 is computed first; riders != 0: pend_row / pend_mask ride */
int infgen_insert_heading(const InfgenRollout* r, const InfgenInsertion* I, int t, int h_ready, int riders, void* stream);

/* ---- edge sets of the teacher-forced forward (reference agent_decoder.py:1104-1603) ----
 * infgen_radius_edges: torch_cluster.radius / radius_graph (agent_decoder.py:632, :710, :780, :875; map_decoder.py:91) for a
 * list of query points, with the filters the reference applies AFTER the radius call, as a compact CSR by destination:
 *   query q: destination node q_node[q] (its off / cnt entries are written; nodes without a query keep what the caller put
 *   there), pose p_pos / p_head / p_inv at index q_pt[q], candidates = indices [q_c0[q], q_c1[q]) of the candidate arrays in
 *   ascending order; the first K with d^2 < radius^2 are "found", of those the ones with index != q_self[q] (optional),
 *   c_ok[index] != 0 (optional) and pair_ok[q_pair_off[q] + index - q_c0[q]] != 0 (optional, q_pair_off < 0: no pair filter)
 *   become edges with src = c_src[index] (optional, else the index) and
 *   raw = (|d|, angle(heading vector of the query, d), wrap(c_head - p_head), index_diff ? index - q_pt[q] : 0),
 *   d = candidate - query; gap_rule 1: the invalid / gap overrides of :595-601 on both components, 2: :722-723 (query
 *   invalid), 0: none.  e->off entries get e_base added (several sets in one buffer); *e->total must be zeroed by the caller
 *   and is the number of edges afterwards (> e->cap: overflow, rows that did not fit have cnt = 0).
 * infgen_motion_features: (|motion vector|, angle(head vector, motion vector), 0, 0) per (row, column) of agent-major
 *   [rows][T] arrays with the state rules of _build_vector_a (:426-447); gap_mask (optional): entries forced to the gap (:1327) */
typedef struct InfgenRadiusEdges {
  int n_q; int _pad0;
  const int* q_node; const int* q_pt; const int* q_c0; const int* q_c1; const int* q_self; const int* q_pair_off;
  const float* p_pos; const float* p_head; const unsigned char* p_inv;
  const float* c_pos; const float* c_head; const unsigned char* c_inv; const unsigned char* c_ok; const int* c_src;
  const unsigned char* pair_ok;
  float radius; int K; int gap_rule; int index_diff;
  int e_base; int _pad1;
} InfgenRadiusEdges;
int infgen_radius_edges(const InfgenRadiusEdges* a, const InfgenEdgeBuf* e, void* stream);
int infgen_motion_features(const float* pos, const float* head, const int* state, const unsigned char* gap_mask, int rows, int T,
                           float* out, void* stream);

/* ---- agent tokenisation on the device ----
 * TokenProcessor._match_agent_token (infgen/datasets/preprocess.py:552-653; cal_polygon_contour :24-54), noise off:
 * valid [A][T] bytes, pos [A][T][2], heading [A][T], shape [A][2] = (width, length); tok = last contour of every token,
 * [n_type][n_token][4][2] indexed by type[a], or per agent (type == NULL, tables tok_agent_stride floats apart).
 * Writes token_index [A][T / shift] (int32) and token_contour [A][T / shift][4][2]. */
int infgen_match_agent_tokens(const unsigned char* valid, const float* pos, const float* heading, const float* shape,
                              const int* type, const float* tok, long long tok_agent_stride, int A, int T, int shift,
                              int n_token, int* token_index, float* token_contour, void* stream);

/* TokenProcessor._tokenize_agent (infgen/datasets/preprocess.py:335-550) in three launches: heading cleaning (:310-317) and
 * extrapolation to the token grid (:319-344) IN PLACE on valid [A][T], pos [A][T][2], heading [A][T], velocity [A][T][2]
 * (the reference modifies its inputs the same way); contour matching against tok [3][n_token][4][2] by type [A]; states,
 * token position / heading, token ids -1 (invalid) / -2 (entering), validity masks, shape reset ([A][T][3], optional).
 * wl_work: [A][2] scratch.  Outputs per agent and token step (T / shift): token_index, state_idx (int32), token_contour
 * [..][4][2], token_pos [..][2], token_heading, token_valid (all 1 when predict_state), raw_token_valid (bytes). */
int infgen_tokenize_agent(unsigned char* valid, float* pos, float* heading, float* velocity, const int* type, const float* tok,
                          const float* shape_in, float* shape_out, float* wl_work, int A, int T, int shift, int current_step,
                          int n_token, int invalid_state, int valid_state, int enter_state, int exit_state, int predict_state,
                          int* token_index, float* token_contour, int* state_idx, float* token_pos, float* token_heading,
                          unsigned char* token_valid, unsigned char* raw_token_valid, void* stream);

/* InfGen._fetch_enterings (infgen/model/infgen.py:1008-1128): token_pos [A][T][2], token_heading [A][T], state_idx [A][T]
 * (int32) of B scenes whose agents are rows agent_ptr[b] .. agent_ptr[b+1] (at most max_agents <= 2048 each), av_index [B]
 * = row of the ego inside its scene; grid_xy [grid_size][2] = Attr_Tokenizer.grid.  Per agent and step: grid_token_idx
 * (-1 = invalid or beyond radius), grid_offset_xy, heading_token_idx, sort_indices (entering agents by bearing, then the
 * ego's row), inrange / bos masks (bytes), pos_xy, heading_theta.  Optional: pt_pos [M][pt_stride] (x, y first) with
 * pt_ptr [B+1] -> pt_grid_token_idx [T][M]. */
int infgen_fetch_enterings(const float* token_pos, const float* token_heading, const int* state_idx, const int* agent_ptr,
                           const int* av_index, int B, int max_agents, int T, const float* grid_xy, int grid_size,
                           float radius, float angle_interval, int enter_state, int invalid_state, int* grid_token_idx,
                           float* grid_offset_xy, int* heading_token_idx, int* sort_indices, unsigned char* inrange_mask,
                           unsigned char* bos_mask, float* pos_xy, float* heading_theta, const float* pt_pos, int pt_stride,
                           const int* pt_ptr, int M, int* pt_grid_token_idx, void* stream);

/* InfGen.match_token_map, the matching core (infgen/model/infgen.py:918-936), noise off: traj_pos [P][3][2], theta [P],
 * sample_pt [n_token][3][2] -> token_idx [P] (int32) */
int infgen_match_map_tokens(const float* traj_pos, const float* theta, const float* sample_pt, int P, int n_token,
                            int* token_idx, void* stream);

/* ---- a rollout-metric feature on the device ----
 * compute_distance_to_nearest_object (infgen/metrics/interact_features.py:19-95): every array [B][N][T] with the evaluated
 * objects in the first n_eval rows of a scene (the order the reference builds, :48-50); work = B*N*T*9 floats of scratch;
 * out [B][n_eval][T]: signed distance to the nearest other valid object (negative = overlap), 1e10 where there is none. */
int infgen_distance_to_nearest_object(const float* cx, const float* cy, const float* length, const float* width,
                                      const float* heading, const unsigned char* valid, int B, int N, int T, int n_eval,
                                      float corner_rounding_factor, float* work, float* out, void* stream);

/* compute_kinematic_features (infgen/metrics/trajectory_features.py:37-51): x, y, z (NULL = 0), heading [n][T] -> linear
 * speed, linear acceleration, yaw rate, yaw acceleration [n][T] (NaN at both ends; accel / yaw outputs may be NULL) */
int infgen_kinematic_features(const float* x, const float* y, const float* z, const float* heading, int n, int T,
                              float seconds_per_step, float* speed, float* accel, float* yaw_rate, float* yaw_accel,
                              void* stream);
/* compute_time_to_collision_with_object_in_front (infgen/metrics/interact_features.py:96-219): arrays [B][N][T] in the
 * original object order, speed from infgen_kinematic_features, eval_idx [n_eval] ascending -> out [B][n_eval][T] seconds */
int infgen_time_to_collision(const float* cx, const float* cy, const float* length, const float* width, const float* heading,
                             const float* speed, const unsigned char* valid, const int* eval_idx, int B, int N, int T,
                             int n_eval, float* out, void* stream);

/* compute_distance_to_road_edge (infgen/metrics/map_features.py:27-79; signed distance :139-349, z stretch 3): boxes
 * [B][N][T] (cz / height may be NULL = 0), eval_idx [B][n_eval] rows of the evaluated objects, road edges of all scenes
 * as padded polylines [P][L][4] = x, y, z, valid with cyclic [P] (:82-136) and poly_off [B+1] = polyline range per scene
 * -> out [B][n_eval][T], -1e10 where the box is invalid; > 0 = off road */
int infgen_distance_to_road_edge(const float* cx, const float* cy, const float* cz, const float* length, const float* width,
                                 const float* height, const float* heading, const unsigned char* valid, const int* eval_idx,
                                 int B, int N, int T, int n_eval, const float* polylines, const unsigned char* cyclic,
                                 const int* poly_off, int L, float z_stretch, float* out, void* stream);

/* Scoring of a feature under its logged distribution (infgen/metrics/compute_metrics.py:845-878 and :744-762), fused over
 * the windows the reference unfolds: values / valid [n][T] (valid NULL = all), windows of `size` steps every `step`;
 * edges [num_bins + 1] (float32 linspace of the histogram), logp [num_bins] log-probabilities of the logged distribution.
 * -> out_sum / out_cnt [n][(T - size) / step + 1]: sum of log-probabilities over the valid steps of a window, their number */
int infgen_window_log_likelihood(const float* values, const unsigned char* valid, int n, int T, int size, int step,
                                 const float* edges, const float* logp, int num_bins, float* out_sum, int* out_cnt,
                                 void* stream);

/* compute_num_placement + compute_distance_placement (infgen/metrics/placement_features.py:6-48): x, y, z (NULL = 0) and
 * state [B][N][T], av_index [B] (row of the ego, excluded) -> num_bos / num_eos [B][T], bos / eos distance [B][N][T] */
int infgen_placement_features(const float* x, const float* y, const float* z, const int* state, const int* av_index,
                              int B, int N, int T, int enter_state, int exit_state, int* num_bos, int* num_eos,
                              float* bos_distance, float* eos_distance, void* stream);

/* Padded row layouts (insertion on: A_cap = agents + head-room rows per scene).  infgen_active_row_groups lists, in
 * ascending order, the 16-row groups of the [S][A_cap] layout with a row below n_agents[s] + margin (groups: capacity
 * ceil(S * A_cap / 16), n_groups: 1 int, both on the device); infgen_set_row_groups makes every split-kernel launch of the
 * attention node kernels over exactly `rows` rows visit only those groups (NULL switches it off).  Rows outside the list
 * keep their previous contents. */
int infgen_active_row_groups(const int* n_agents, int S, int A_cap, int margin, int* groups, int* n_groups, void* stream);
int infgen_set_row_groups(const int* groups, const int* n_groups, int rows);
/* the same bound for the edge kernel of those launches: n_agents [S], A_cap and the margin given to infgen_active_row_groups */
int infgen_set_row_limits(const int* n_agents, int A_cap, int margin);

/* Arithmetic of the split GEMM kernels: 3 (default) = fp16 three-term split, fp32 accuracy.  Reduced precision, for BASELINE
 * config C5 (the reference's optional bf16 trainer precision), outside the 1e-3 parity bar:
 *   1 = the hi x hi term only: fp16 operands (11 significant bits), fp32 accumulation;
 *   2 = bf16 operands: every activation operand is rounded to nearest even at 8 significant bits before it enters the matrix pipe
 *       (k_*_b16 kernels), fp32 accumulation; takes packs whose fp16 planes hold bf16-rounded weights (infgen_amd.packing.
 *       operand_bits(8) / PackedWeights(operand_bits=8)) - with default packs the weights keep 11 bits, which the library cannot
 *       see: the Python engine refuses the mismatch.
 * The edge kernels' two small GEMMs (u = q W'kr, W'vr z) stay three-term in every mode; their weights come from the same packs. */
int infgen_set_gemm_terms(int terms);

/* ---- optional profiling (bench.py roofline leg; process-global, off by default) ----
 * HIP events are recorded on the launch stream around every launch of the kernels selected by
 * `mask` (bit = INFGEN_KID_*).  infgen_prof_collect synchronises the device, returns the summed
 * duration (ms), launch count and algorithmic multiply-accumulates per kernel id and 16 device-side counters, then clears
 * the event log.  counters[n] (n < 8): rows processed by the Fourier kernel with n input dims; counters[8 + k]: edges
 * built by infgen_build_edges (k = 0 temporal, 1 map, 2 agent set; counted when INFGEN_KID_EDGE_ATTN or
 * INFGEN_KID_BUILD_EDGES is selected) - each is consumed by one edge-attention launch per layer. */
enum {
  INFGEN_KID_LINEAR = 0, INFGEN_KID_FOURIER, INFGEN_KID_ATTN_PRE, INFGEN_KID_EDGE_ATTN, INFGEN_KID_ATTN_POST,
  INFGEN_KID_HEADS, INFGEN_KID_BUILD_EDGES, INFGEN_KID_INTEGRATE, INFGEN_KID_RAWFEAT, INFGEN_KID_MAP_GRAPH,
  INFGEN_KID_COUNT
};
int infgen_prof_enable(unsigned mask, int max_launches);
int infgen_prof_collect(double* total_ms, int* calls, double* total_macs, unsigned long long* counters);
/* the same, plus the share of every kernel that was launched INSIDE a decode step (infgen_decode_step / infgen_rollout_run):
 * step_ms / step_calls [INFGEN_KID_COUNT]; the remainder is the prologue (map encoder, column-0 chain) and operator-level calls */
int infgen_prof_collect_steps(double* total_ms, int* calls, double* total_macs, unsigned long long* counters,
                              double* step_ms, int* step_calls);
/* after infgen_prof_enable: bracket only every stride-th launch of the selected kernels INSIDE decode steps (an event pair costs
 * ~5 us of launch-stream time; launches outside decode steps are all bracketed).  infgen_prof_seen: launches of every kernel since
 * infgen_prof_enable, bracketed or not - seen / seen_step [INFGEN_KID_COUNT] (either may be NULL). */
int infgen_prof_set_stride(int stride);
int infgen_prof_seen(int* seen, int* seen_step);

#ifdef __cplusplus
}
#endif
#endif  /* INFGEN_HIP_H_ */
