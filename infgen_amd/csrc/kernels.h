// Kernel argument blocks (plain structs, device pointers) and kernel declarations.
#pragma once
#include "tile.cuh"
#include "layout.h"

namespace ig {

struct LinearArgs {
  const float* X; int ldx; const int* gather;
  int rows; int K; int Kp;
  const float* Wp; int Np; const float* bias; int N;
  const float* pre_g; const float* pre_b;
  const float* post_g; const float* post_b; int relu;
  float* Y; int ldy;
};

constexpr int LINEAR_MULTI_MAX = 6;
struct LinearMultiArgs { LinearArgs d[LINEAR_MULTI_MAX]; };

struct FourierArgs {
  const float* raw;        // [E][4]
  int n;                   // continuous dims (2..4)
  const int* count_dev;    // optional device-side row count
  int e_cap;               // row count (or capacity when count_dev != null)
  const float* pack;       // FourierLayout
  const float* cat; int ldcat;   // optional [E][ldcat] categorical embedding sum
  float* out; int ldo;
  int normalize;
  unsigned long long* prof_rows;   // optional [8]: rows processed, indexed by n (profiling only)
  int out_r24;             // k_fourier_h only: 1 rows in the packed 24-bit format (R24_ROW_BYTES per row, below) instead of fp32
  // k_fourier_h only - the LAST input dim as a lookup (temporal edges: the time gap j - c is one of -1 .. -16, edge_kernels.hip):
  // dt_mode 1: dims 0 .. n - 2 are evaluated, row (int)(-raw[e][n - 1]) of dt_tab [DT_TAB_ROWS][128] is added to their sum
  //            (8 of the 4 (2 n + 1) weight quarters per tile are never staged);
  // dt_mode 2: writes that table - row e = the last dim's branch mlps[n - 1](-e) without its bias (the bias sum stays in the pack)
  const float* dt_tab; int dt_mode;
  int dbg;                 // diagnostics (INFGEN_QS_DBG; wrong results): 1 free-running waves over a static weight ring
};
constexpr int DT_TAB_ROWS = 32;
// k_fourier_h geometry (experiments: -DIG_FH_WAVES=4 -DIG_FH_RING=4 builds the two-workgroups-per-CU variant)
#ifndef IG_FH_WAVES
#define IG_FH_WAVES 8
#endif
// IG_QSU = 1: k_fourier_h and the 8-wave k_attn_h take their weights from the unit-granular stream (split.cuh: QuarterStreamU,
// one workgroup barrier per GEMM instead of one per quarter-matrix; 8 quarter buffers)
#ifndef IG_QSU
#define IG_QSU 0
#endif
// IG_FH_AP = 1 (default): k_fourier_h as two groups of four waves one phase apart (fourier_ap_body.inc: the matrix phases of
// one wave of a SIMD run under the vector phases of the other); 0: all eight waves in step, one barrier per quarter
#ifndef IG_FH_AP
#define IG_FH_AP 1
#endif
#ifndef IG_FH_RING
#define IG_FH_RING ((IG_QSU || IG_FH_AP) ? 8 : 5)
#endif
constexpr int FH_WAVES = IG_FH_WAVES;
constexpr int FH_NT = 64 * FH_WAVES;     // threads per workgroup
constexpr int FH_TILE = 16 * FH_WAVES;   // edges per workgroup tile (16 per wave)
constexpr int FH_RING = IG_FH_RING;      // quarter buffers of its weight ring
constexpr int FH_WG_PER_CU = FH_WAVES == 8 ? 1 : 2;
#ifndef IG_AH_RING4
#define IG_AH_RING4 3
#endif
#ifndef IG_AH_RING8
#define IG_AH_RING8 (IG_QSU ? 8 : 5)
#endif
struct FourierMultiArgs { FourierArgs set[4]; };      // (the fourth: the rows' x_a_emb embedding, infgen_rollout_run)

// Packed 24-bit rows of the normalised relative-position embedding (the rollout's private edge buffers; k_fourier_h writes,
// k_edge_fused reads): the upper 24 bits of every fp32 value (sign, exponent, 15 mantissa bits, round to nearest even) as two
// planes per row - 128 x 16 bit (bits 31..16) then 128 x 8 bit (bits 15..8): 384 B instead of 512.  The values are O(1)
// (affine-free LayerNorm output): relative error 2^-17, 32 x below fp16; the edge loop is bound by the bytes it gathers.
constexpr int R24_ROW_BYTES = 384;
constexpr int R24_LO_PLANE = 256;

// One edge set in CSR-by-destination form (built on the device every decode step).
struct EdgeSet {
  const int* off;          // [rows] first edge of the row
  const int* cnt;          // [rows] number of incoming edges
  const int* src;          // [E] row index into the K/V source arrays
  const float* rhat;       // [E][128] normalised relative-position embedding (nullptr: no pos emb)
};

struct AttnPreArgs {
  const float* X; int rows;          // [rows][128] layer input
  const float* pack;                 // AttnLayout
  int use_src_ln;                    // 1: LayerNorm with the *_src parameters (K/V of a bipartite source)
  float* Q;                          // [rows][128]   (scaled q), may be null
  float* U;                          // [rows][8][128] absorbed relative-position query, may be null
  float* K; float* V;                // [rows][128] each, may be null
};

struct EdgeAttnArgs {
  int rows;
  const float* Q; const float* U;
  const float* Ksrc; const float* Vsrc;    // source K/V arrays, row stride 128
  EdgeSet es;
  float* AGG;                        // [rows][128]  sum_e attn * v_src
  float* Z;                          // [rows][8][128] sum_e attn * rhat   (pos-emb layers only)
  float* SIG;                        // [rows][8]    sum_e attn
  const float* wkr;                  // unused (kept for layout stability)
  const int* n_agents; int A_cap, margin;   // optional (k_edge_attn): rows at or beyond n_agents[s] + margin of their scene are skipped
  const int* row_mask;               // optional (k_edge_attn_wide): rows with row_mask[row] == 0 are treated as edgeless
};

// k_edge_fused (edge_fused.hip): edge attention of a 16-row tile with the absorbed query and the positional aggregate kept on
// chip: q in, agg' = sum_e a_e v_src + W'_vr z + b' sigma out (what k_attn_h / k_attn_post then take with has_pos = 0)
struct EdgeFusedArgs {
  int rows;
  const float* Q;                    // [rows][128] scaled query
  const float* pack;                 // AttnLayout of the layer (AH_HDR scales, W'_kr / W'_vr quarter-matrices, AL_BVR)
  const float* Ksrc; const float* Vsrc;
  EdgeSet es;                        // rhat must be present
  float* AGG;                        // [rows][128] out
  const int* groups; const int* n_groups;   // optional: the 16-row groups to process (k_active_groups)
  int dbg;                           // timing experiments only (INFGEN_EDGE_DBG): bit 0 no edges
  int tiles_per_scene;               // > 1: XCD-aware tile order (rows of a scene are A_cap = 32 * tiles_per_scene consecutive rows)
  int kv_once;                       // 1: every K / V source row is read once (temporal ring): non-temporal loads
  int n_virtual;                     // tile slots to visit (tiles rounded up to whole XCD groups)
  unsigned* dbgbuf;                  // k_edge_fused_p only (INFGEN_EDGE_DBG bit 3): [rows][12] checksums of the hand-offs between the phases
  WarmArgs warm;                     // one-group variant only
};

// k_layers_p (layers_p.hip): every sublayer of a decode step in one launch for up to 256 16-row groups; one workgroup owns a group
constexpr int LP_MAX_LAYERS = 8;
struct LayersPArgs {
  int rows, A_cap, num_layers;
  int xcd_order;                     // 1: a scene's workgroups share an XCD (workgroups a multiple of 8 * A_cap / rows_per_wg)
  int row0;                          // first row of this launch (a batch beyond one launch's workgroup limit runs as chunks of whole scenes)
  int rows_per_wg;                   // 16, or 8 (the smallest batches: one row per wave in the edge loop, lanes j >= 8 are shadows)
  float* X;                          // [rows][128] in / out: the layer stack's input rows (raw features) -> its output
  const float* attn_t[LP_MAX_LAYERS]; const float* attn_m[LP_MAX_LAYERS]; const float* attn_a[LP_MAX_LAYERS];
  float* ringK[LP_MAX_LAYERS]; float* ringV[LP_MAX_LAYERS];      // [ring][rows][128]
  size_t slot_off;                   // floats: the current column's slot of the ring
  const float* mapK[LP_MAX_LAYERS]; const float* mapV[LP_MAX_LAYERS];
  float* Ka[2]; float* Va[2];        // agent-set K / V rows, double buffered by layer parity
  EdgeSet et, em, ea;
  int* sync;                         // [scenes] zeroed before the launch: arrivals of a scene's workgroups per layer
  unsigned spin_limit;               // 0: wait at the scene counters without limit (cooperative launch: residency is guaranteed);
                                     // n: trap after n polls (diagnostics, and launches captured into a HIP graph)
  unsigned long long* trace;         // diagnostics (INFGEN_LP_TRACE=1): [1024][2] (stamp id, s_memtime) of workgroup 0
};

struct AttnPostArgs {
  float* X; int rows;                // in/out residual stream
  const float* pack;
  const float* AGG; const float* Z; const float* SIG;
  int has_pos;
  // optional fused prologue of the NEXT layer on the freshly written rows (k_attn_pre's work)
  const float* next_pack;            // null: none
  float* nQ; float* nU; float* nK; float* nV;
};

// k_attn_h (attn_h.hip): the node side of one AttentionLayer on the fp16 matrix pipe - any of
//   post:  everything after the edge aggregation of layer `pack` (k_attn_post's work), in place on X
//   pre:   LayerNorm + q / u / k / v projections of layer `next_pack` on the (updated) rows (k_attn_pre's work)
struct AttnHArgs {
  float* X; int rows;
  const float* pack;                 // layer whose post part runs (null: none)
  const float* AGG; const float* Z; const float* SIG;
  int has_pos;
  const float* next_pack;            // layer whose pre part runs (null: none)
  int next_src_ln;                   // 1: LayerNorm with the *_src parameters (K/V of a bipartite source)
  float* nQ; float* nU; float* nK; float* nV;
  const int* groups; const int* n_groups;   // optional: the 16-row groups to process (device list + count), see k_active_groups
  int dbg;                           // diagnostics (INFGEN_QS_DBG; wrong results): 1 free-running waves over a static weight ring
  WarmArgs warm;                     // k_attn_hs only
};

struct ActiveGroupsArgs { const int* n_agents; int S, A_cap, margin; int* groups; int* n_groups; };

// metric_kernels.hip: compute_distance_to_nearest_object; all arrays [B][N][T], evaluated objects first
struct NearestArgs {
  const float* cx; const float* cy; const float* length; const float* width; const float* heading;
  const unsigned char* valid;
  int B, N, T, n_eval;
  float rounding;                    // corner_rounding_factor (0.7)
  float* work;                       // [B][N][T][9] scratch: shrunk corners + shrink radius
  float* out;                        // [B][n_eval][T]
};

struct KinematicArgs {               // compute_kinematic_features; x, y, z (may be null), heading: [n][T]
  const float* x; const float* y; const float* z; const float* heading;
  int n, T; float dt;
  float* speed; float* accel; float* yaw_rate; float* yaw_accel;      // [n][T]; all but speed may be null
};
struct TtcArgs {                     // compute_time_to_collision_with_object_in_front; arrays [B][N][T] in ORIGINAL object order
  const float* cx; const float* cy; const float* length; const float* width; const float* heading; const float* speed;
  const unsigned char* valid;
  const int* eval_idx;               // [n_eval] evaluated objects, ascending
  int B, N, T, n_eval;
  float* out;                        // [B][n_eval][T]
};

struct TokenizeArgs {                // _tokenize_agent bookkeeping (k_tokenize_prep / k_tokenize_state)
  unsigned char* valid; float* pos; float* heading; float* velocity;       // [A][T], [A][T][2], [A][T], [A][T][2] in/out
  const int* type; float* wl;                                              // [A], [A][2] out (width, length)
  const float* shape_in; float* shape_out;                                 // [A][T][3] (may be null)
  int A, T, shift, current_step;
  int invalid_state, valid_state, enter_state, exit_state, predict_state;
  int* token_index; const float* token_contour;                            // [A][T/shift] in/out, [A][T/shift][4][2]
  int* state_idx; float* token_pos; float* token_heading;                  // [A][T/shift](,2)
  unsigned char* token_valid; unsigned char* raw_token_valid;              // [A][T/shift]
};

struct EnteringsArgs {               // _fetch_enterings (k_fetch_enterings / k_pt_grid_cells)
  const float* token_pos; const float* token_heading; const int* state_idx;    // [A][T][2], [A][T], [A][T]
  const int* agent_ptr; const int* av_index;                                   // [B+1], [B] (row inside the scene)
  int B, T; const float* grid_xy; int grid_size;
  float radius, angle_interval; int enter_state, invalid_state;
  int* grid_token_idx; float* grid_offset_xy; int* heading_token_idx; int* sort_indices;
  unsigned char* inrange_mask; unsigned char* bos_mask; float* pos_xy; float* heading_theta;
  const float* pt_pos; int pt_stride; const int* pt_ptr; int M; int* pt_grid_token_idx;   // [M][pt_stride], [B+1], [T][M]
};

struct RoadEdgeArgs {                // compute_distance_to_road_edge; boxes [B][N][T]
  const float* cx; const float* cy; const float* cz; const float* length; const float* width; const float* height;
  const float* heading; const unsigned char* valid; const int* eval_idx;     // eval_idx [B][n_eval]
  const float* poly; const unsigned char* cyclic; const int* poly_off;       // [P][L][4], [P], [B+1]
  int B, N, T, n_eval, L; float z_stretch;
  float* out;                                                                // [B][n_eval][T]
};

struct WindowLoglikArgs {            // k_window_loglik
  const float* values; const unsigned char* valid;      // [n][T]; valid may be null (all)
  int n, T, size, step;
  const float* edges; const float* logp; int nb;        // [nb + 1], [nb], nb <= 64
  float* out_sum; int* out_cnt;                         // [n][(T - size) / step + 1]
};

struct PlacementArgs {               // placement_features; arrays [B][N][T]
  const float* x; const float* y; const float* z;      // z may be null
  const int* state; const int* av_index;               // [B][N][T], [B]
  int B, N, T, enter_state, exit_state;
  int* num_bos; int* num_eos;                          // [B][T]
  float* bos_distance; float* eos_distance;            // [B][N][T]
};

// k_mlpemb_h (mlp_h.hip): MLPEmbedding with K0 = 128 j on the fp16 split
struct MlpEmbHArgs {
  const float* X; int ldx; int rows; int K0;
  const float* pack;                 // packing.pack_mlp_embedding incl. its split section
  float* Y; int ldy;
};

// k_match_tokens (token_kernels.hip): TokenProcessor._match_agent_token, one workgroup per agent
struct MatchTokensArgs {
  const unsigned char* valid;        // [A][T]
  const float* pos;                  // [A][T][2]
  const float* heading;              // [A][T]
  const float* shape;                // [A][2] = (width, length)
  const int* type;                   // [A] index into tok (null: per-agent tables, tok_agent_stride floats apart)
  const float* tok;                  // [3 or A][n_token][4][2] last contour of every token
  long long tok_agent_stride;
  int A, T, shift, n_token;
  int* token_index;                  // [A][T / shift]
  float* token_contour;              // [A][T / shift][4][2]
};

// k_match_map_tokens (token_kernels.hip): InfGen.match_token_map, one wave per polyline piece
struct MatchMapArgs {
  const float* traj_pos;             // [P][3][2]
  const float* theta;                // [P]
  const float* sample_pt;            // [n_token][3][2]
  int P, n_token;
  int* token_idx;                    // [P]
};

struct HeadsArgs {
  const float* X; int rows;
  const float* tok_pack;    // MLPLayer pack: P(128,128) W0, b0, ln g/b, P(128,2048) W3, b3
  const float* st_pack;     // MLPLayer pack: P(128,128) W0, b0, ln g/b, W3 [3][128] row-major, b3[3]
  int token_size;
  float* logits;            // optional [rows][token_size]
  int* next_token;          // [rows]
  int* next_state;          // [rows] raw argmax in {0,1,2}
  // k_heads only, few rows: the token_size / 128 logit chunks dealt to gridDim.y = nsplit workgroups per row tile; each merges its
  // (max, first index) into part[row] as an ordered 64-bit key (atomicMax; zeroed by the caller), k_heads_finish decodes them
  unsigned long long* part; int nsplit;
};
__global__ void k_heads_finish(const unsigned long long* part, int rows, int* next_token);

// ---- per-scene state (column-major per scene: [S][T][A_cap]) -----------------------------------
struct SceneState {
  int S, A_cap, T, M_cap, W;        // W = temporal window (time_span / shift)
  int ring;                         // ring slots of the temporal K/V cache (> W)
  const int* n_agents;              // [S]
  const int* n_map;                 // [S]
  const int* av_index;              // [S]
  float* pos;                       // [S][T][A_cap][2]
  float* head;                      // [S][T][A_cap]
  int* state;                       // [S][T][A_cap]
  int* token;                       // [S][T][A_cap]
  int* grid;                        // [S][T][A_cap]
  unsigned char* tmask;             // [S][T][A_cap]
  unsigned char* imask;             // [S][T][A_cap]
  unsigned char* catflag;           // [S][T][A_cap] 1: own type/shape embedding, 0: seed/invalid-shape
  const int* type;                  // [S][A_cap]
  int* bos;                         // [S][A_cap] first 'enter' column (0 if none)
  const float* map_pos;             // [S][M_cap][2]
  const float* map_orient;          // [S][M_cap]
  const int* map_scene;             // optional [S]: slot of scene s in the map-side arrays (n_map, map_pos, map_orient, map K / V rows)
  // scenario insertion (optional, may be null): rows >= first_new[s] inserted in the current step carry
  // the newest row's head vector during the motion stage (reference agent_decoder.py:2083, SURVEY a-Q13)
  const int* first_new;             // [S]
  const float* hv_ovr;              // [S][2]
};

struct EdgeBuf {                    // device-side builder outputs for one edge type
  int* off; int* cnt; int* src; float* raw; int* total; int cap;
};

// forward_kernels.hip: torch_cluster.radius with an emit filter for a list of query points (the teacher-forced forward's
// edge sets and the map-token graph); include/infgen_hip.h: InfgenRadiusEdges has the field-by-field description
struct RadiusEdgesArgs {
  int n_q;
  const int* q_node; const int* q_pt; const int* q_c0; const int* q_c1; const int* q_self; const int* q_pair_off;
  const float* p_pos; const float* p_head; const unsigned char* p_inv;
  const float* c_pos; const float* c_head; const unsigned char* c_inv; const unsigned char* c_ok; const int* c_src;
  const unsigned char* pair_ok;
  float radius; int K; int gap_rule; int index_diff;
  EdgeBuf e; int e_base;
};

struct MotionFeatArgs {
  const float* pos; const float* head; const int* state; const unsigned char* gap_mask;
  int rows, T;
  float* out;                       // [rows][T][4]
};

struct BuildEdgesArgs {
  SceneState st;
  int c;                            // current column
  int edgeless;                     // 1: write empty edge sets (column 0 chain)
  float r_map, r_agent;             // pl2a_radius, a2a_radius
  int rows;                         // S * A_cap
  EdgeBuf t, m, a;
  unsigned long long* prof;         // optional profiling counters (api.hip Prof::rows_dev): [8 + kind] += edges of the scene
  int map_lds;                      // float2 slots of dynamic LDS for the scene's map-token positions (0 .. 4096)
  unsigned long long* clear_keys;   // optional [rows]: k_heads' split arg-max keys, reset here when the k_integrate before ran in row groups
  int* clear_sync;                  // optional [S]: the per-scene counters of the k_layers_p launch that follows, zeroed here
};

struct RawFeatArgs {
  SceneState st;
  int col;
  const float* tok_tab;             // [3][token_size + 2][128]
  int token_size;
  const float* grid_tab;            // [grid_size + 1][128]
  int grid_size;
  const float* state_emb;           // [4][128]
  const float* cat_agent;           // [rows][128] type_emb[type] + shape_emb(shape)
  const float* cat_seed;            // [128]
  float* raw2;                      // [rows][4]  (|mv|, angle, -, -)
  float* cat;                       // [rows][128]
  float* fus_in;                    // [rows][512]
  const int* row_list; const int* row_mask; int n_list;   // optional: only these rows (row_mask[k] != 0), outputs compact at k
};

struct IntegrateArgs {
  SceneState st;
  int c;                            // current column; writes column c + 1
  int t;                            // decode step
  int R;                            // num_recurrent_steps_val
  int force_valid;                  // disable_insertion: every state := valid
  const int* next_token; const int* next_state;   // [rows] from the heads
  const int* teacher_token; const int* teacher_state;   // optional [S][T][A_cap]
  const int* teacher_grid;                               // optional [S][T][A_cap], < -1: none
  const float* teacher_pos; const float* teacher_head;   // optional [S][T][A_cap](x2): the stored pose of column c + 1
  // the tail of a decode step folded into this launch (infgen_rollout_run with few rows; all optional):
  unsigned long long* heads_part;   // k_heads' split arg-max keys [rows]: decoded here (k_heads_finish) and reset for the next step
  int* next_token_w;                // where the decoded tokens go (the context's next_token array)
  int* edge_totals;                 // the three edge totals of the context, zeroed for the next column's k_build_edges
  int* zero_sync;                   // optional [S]: the per-scene counters of the next step's k_layers_p launch, zeroed here
  RawFeatArgs prep; int do_prep;    // the raw-feature gather (k_rawfeat_prep) of the new column, all rows of the scene
  int groups;                       // > 1: grid S x groups, A_cap / groups rows per workgroup (few scenes); the keys are then NOT reset here
  const float* vocab;               // [3][token_size][6][4][2]
  int token_size;
  const float* grid_xy; int grid_size;    // [G][2]
  float* pred_traj;                 // [S][A_cap][R][2]
  float* pred_head;                 // [S][A_cap][R]
  float* pred_state;                // [S][A_cap][R]
};


// edges into ONE query point per scene (insertion: the seed node at the ego pose, or a freshly
// inserted row during its heading stage): first-K agents / map tokens within a radius of the
// centre row's position, ascending index (torch_cluster.radius), filtered afterwards.
struct PointEdgesArgs {
  SceneState st;
  int c;
  const int* centre_row;            // [S] agent row whose column-c pose is the query point
  const int* active;                // [S] 0: emit nothing for this scene
  int exclude_centre;               // 1: the centre row is not a source (heading stage)
  int which;                        // bit 0: agents, bit 1: map tokens
  float r_agent; int k_agent;
  float r_map; int k_map;
  EdgeBuf ea, em;                   // one destination per scene: off[s], cnt[s]
};

struct InsertCatArgs {
  int S, A_cap;
  const int* inserted; const int* new_row; const int* type; const float* type_emb; const float* shp; const float* new_shape;
  float* cat_agent; float* shape_all; int* new_local;
};
struct OccupancyArgs { SceneState st; int c; int grid_size; float* occ; /* [S][grid_size] */ };
struct OccEmbedArgs { SceneState st; int c; int grid_size; float* occ; const float* pack; float* emb; /* [S][128] */
                      const int* active; /* optional: scenes with active[s] == 0 get the occupancy vector only */ };

struct InsertDecideArgs {
  SceneState st;
  int c, t, R, grid_size, force_enter, max_new;
  int sample_k; const float* uniform;   // cell sampling: top-k inverse CDF with uniform[S] (sample_k <= 1: arg-max)
  const float* grid_xy;
  const float* lg_state;            // [S][2]
  const float* lg_type;             // [S][3]
  const float* shape;               // [S][3]
  const float* lg_pos;              // [S][grid_size]
  const float* occ;                 // [S][grid_size]
  int* n_agents;                    // [S] (mutable view of st.n_agents)
  int* type;                        // [S][A_cap]
  int* active;                      // [S] in/out
  int* n_new;                       // [S] in/out
  int* inserted;                    // [S] out
  int* new_row;                     // [S] out (global row index s*A_cap + a)
  float* new_shape;                 // [S][3] out
  int* new_cell;                    // [S] out
  float* pred_traj; float* pred_head; float* pred_state;
};

struct InsertFinalizeArgs {
  SceneState st;
  int c;
  float angle_interval;
  const int* inserted; const int* new_row;
  const float* lg_heading; int n_heading;     // [S][n_heading]
  const float* offset;                        // [S][2] (pre-tanh)
  float* hv_ovr;                              // [S][2]
};

struct SampleArgs {
  const float* logits; int rows; int n;      // [rows][n]
  int k;                                     // beam size (<= 16)
  const float* uniform;                      // [rows] caller-supplied U[0,1)
  int* token;                                // [rows] out
};

constexpr int MAP_GRAPH_C = 8;          // centre tokens per wave of k_map_graph (32 per workgroup: one LDS staging of the scene's tokens, one atomic)
struct MapGraphArgs {
  int S, M_cap; const int* n_map;
  const float* pos; const float* orient; float radius; int max_nbr;
  EdgeBuf e;
  int lds_tokens;          // tokens the launch's dynamic LDS holds (12 bytes each; 0: every trip reads global memory)
};

template <int W> __global__ void k_stream_read(const float* p, size_t n_floats, float* out);   // gemm_kernels.hip
__global__ void k_linear(LinearArgs a);
__global__ void k_linear_multi(LinearMultiArgs m);
__global__ void k_fourier(FourierArgs a);
template <int TERMS> __global__ void k_fourier_h(FourierArgs a);
template <int TERMS> __global__ void k_fourier_h_multi(FourierMultiArgs m);
template <int TERMS> __global__ void k_fourier_h12(FourierArgs a);      // fourier_h12.hip: three wave groups, 192-edge tiles
template <int TERMS> __global__ void k_fourier_h12_multi(FourierMultiArgs m);
constexpr int FH12_NT = 768, FH12_TILE = 192;
__global__ void k_match_tokens(MatchTokensArgs a);   // token_kernels.hip
template <int TERMS> __global__ void k_mlpemb_h(MlpEmbHArgs a);           // mlp_h.hip
__global__ void k_box_corners(NearestArgs a);        // metric_kernels.hip
__global__ void k_nearest_distance(NearestArgs a);
__global__ void k_kinematic(KinematicArgs a);
__global__ void k_ttc(TtcArgs a);
__global__ void k_placement(PlacementArgs a);
__global__ void k_window_loglik(WindowLoglikArgs a);
__global__ void k_road_edge(RoadEdgeArgs a);
template <int TERMS> __global__ void k_heads_h(HeadsArgs a);
__global__ void k_match_map_tokens(MatchMapArgs a);
__global__ void k_tokenize_prep(TokenizeArgs a);
__global__ void k_fetch_enterings(EnteringsArgs a);
__global__ void k_pt_grid_cells(EnteringsArgs a);
__global__ void k_tokenize_state(TokenizeArgs a);
__global__ void k_active_groups(ActiveGroupsArgs a);
template <int WAVES, int TERMS> __global__ void k_attn_h(AttnHArgs a);
// the *_b16 translation units: the same kernels with bf16-precision operands (gemm_terms = 2; TERMS = 1 only)
template <int WAVES, int TERMS> __global__ void k_attn_h_b16(AttnHArgs a);
template <int TERMS> __global__ void k_attn_hs_b16(AttnHArgs a);
template <int TERMS> __global__ void k_mlpemb_h_b16(MlpEmbHArgs a);
template <int TERMS> __global__ void k_heads_h_b16(HeadsArgs a);
template <int TERMS> __global__ void k_fourier_h_b16(FourierArgs a);
template <int TERMS> __global__ void k_fourier_h_multi_b16(FourierMultiArgs m);
template <int TERMS> __global__ void k_attn_hs(AttnHArgs a);             // attn_hs.hip: the same for few rows (one 16-row group per workgroup)   // attn_h.hip     // fourier_h.hip: fp16 three-term split, register resident
__global__ void k_attn_pre(AttnPreArgs a);
__global__ void k_edge_attn(EdgeAttnArgs a);
template <int G> __global__ void k_edge_fused3(EdgeFusedArgs a);       // edge_fused3.hip: lane = (head, 16-column slice), rhat rows through LDS
template <int G, bool R24, int HALVES, int WAVES> __global__ void k_edge_fused(EdgeFusedArgs a);        // edge_fused.hip (R24: rhat rows in the packed format)
template <bool R24, int ROWS> __global__ void k_layers_p(LayersPArgs a);          // layers_p.hip
__global__ void k_edge_attn_wide(EdgeAttnArgs a);
__global__ void k_attn_post(AttnPostArgs a);
__global__ void k_heads(HeadsArgs a);
template <int BT> __global__ void k_build_edges(BuildEdgesArgs a);
template <int BT> __global__ void k_integrate(IntegrateArgs a);
__global__ void k_rawfeat_prep(RawFeatArgs a);
__global__ void k_scatter_rows(const float* src, const int* row_list, const int* row_mask, int n, float* dst);
__global__ void k_scatter_rows2(const float* src0, const float* src1, const int* row_list, const int* row_mask, int n, float* dst0,
                                float* dst1);
__global__ void k_note_riders(const int* new_row, const int* inserted, int n, int* prev_row, int* prev_mask, int* pend_row,
                              int* pend_mask);
struct EmbedSum4Args {
  const float* tab[4]; const long long* idx[4]; int n[4];
  int rows; float* out;
};
__global__ void k_embedding_sum4(EmbedSum4Args a);
__global__ void k_gather_rows(const float* src, const int* row_list, const int* row_mask, int n, int limit, float* dst);
__global__ void k_insert_cat(InsertCatArgs a);
__global__ void k_map_graph(MapGraphArgs a);
__global__ void k_point_edges(PointEdgesArgs a);
__global__ void k_occupancy(OccupancyArgs a);
__global__ void k_occupancy_embed(OccEmbedArgs a);
__global__ void k_insert_decide(InsertDecideArgs a);
__global__ void k_insert_finalize(InsertFinalizeArgs a);
__global__ void k_sample_topk(SampleArgs a);
__global__ void k_layernorm(const float* X, int rows, const float* g, const float* b, float* Y);
__global__ void k_radius_edges(RadiusEdgesArgs a);            // forward_kernels.hip
__global__ void k_motion_features(MotionFeatArgs a);

}  // namespace ig
