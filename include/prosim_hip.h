/*
 * prosim_hip.h -- C ABI of libprosim_hip.so, the MI355X (gfx950) closed-loop rollout engine.
 *
 * Drop-in boundary: the reference has no FFI of its own -- its
 * plugin API is the Python string registry (prosim/core/registry.py:54-134) through which
 * ProSim (prosim/models/traj_sam.py) looks up a scene encoder, a decoder ("generator") and a
 * policy.  The entry points below are what thin Python classes registered under that registry
 * bind with ctypes; each cites the reference call it replaces.
 *
 * Conventions: plain pointers and sizes only; all tensors are dense row-major float32 unless
 * noted; masks are uint8 (0/1); every function returns 0 on success or a negative PS_E_* code
 * and ps_last_error() then holds a message.  Host buffers are caller-owned and may be freed
 * as soon as the call returns; device buffers are engine-owned.  One engine per GPU; a handle
 * is not thread-safe.  The reference signals errors with Python exceptions/asserts -- the
 * host wrapper (prosim_amd/engine.py) turns non-zero codes into RuntimeError.
 */
#ifndef PROSIM_HIP_H
#define PROSIM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PS_OK 0
#define PS_E_ARG (-1)      /* bad argument / unsupported configuration */
#define PS_E_STATE (-2)    /* call order violated (e.g. rollout before set_scene) */
#define PS_E_HIP (-3)      /* HIP runtime error */
#define PS_E_WEIGHT (-4)   /* missing / mis-shaped weight tensor */

typedef struct ps_engine ps_engine;

/* Mirror of prosim_amd.spec.ModelSpec = the MODEL.* / DATASET.FORMAT.* / ROLLOUT.* values the
 * reference reads on this path (prosim_demo/cfg/no_text.yaml:212-279). */
typedef struct ps_config {
  int32_t hidden, heads, head_dim;             /* must be 128, 8, 16 in this build */
  int32_t scene_layers, scene_knn, agent_knn;  /* SCENE_ENCODER.ATTN; agent_knn = min(4*k, 100) */
  int32_t dec_layers, dec_max_neigh;
  int32_t goal_pred_k;                         /* MODEL.DECODER.GOAL_PRED: K goal hypotheses per prompt (decoder/base.py:22-58); 0 = disabled; K <= 64 */
  float dec_prompt_radius, dec_scene_radius;
  int32_t pol_layers, pol_max_neigh;
  float pol_agent_radius, pol_map_radius;
  int32_t cond_layers;
  int32_t drag_pre_layers, drag_mlp_layers;    /* CONDITION_ENCODER.DRAG_POINTS (default.py:531-533); mlp_layers 0 = no drag-point encoder */
  int32_t obs_fusion_mlp, obs_attn_update;     /* MODEL.OBS_UPDATE: FUSION == 'mlp', ATTN_UPDATE (default.py:499-501; both 0 in the demo) */
  float enc_agent_radius, enc_scene_radius;    /* SCENE_ENCODER.ATTN.AGENT_RADIUS / SCENE_RADIUS, used by ATTN_UPDATE only */
  int32_t hist_steps, obs_dim, map_dim;        /* 11, 24, 11 */
  int32_t map_pre_layers, map_mlp_layers, obs_pre_layers, obs_mlp_layers;
  int32_t target_steps, state_dim, motion_k, num_agent_types, prompt_dim;
  int32_t replan_freq, max_steps;
  float dt, ln_eps;
  int32_t device;                              /* HIP device ordinal */
  /* *.ATTN.LEARNABLE_PE of the scene encoder / the decoder / the policy (default.py:472, :594, :665; 0 in the demo): the
   * relative-PE rows of that part's two edge sets come from a learnable FourierEmbedding (layers/fourier_embedding.py:11-54;
   * weights "<part>.<set>_rel_pe_emb.*") instead of the fixed one.  pe_num_freq must be 64 (the reference's default): an embedding
   * with fewer bands is handed over zero-padded (freqs [3][64], mlps.i.0.weight [128][129] = cos 64 | sin 64 | x; extra bands 0). */
  int32_t enc_learnable_pe, dec_learnable_pe, pol_learnable_pe, pe_num_freq;
  /* Binary (agent-pair) tags: bit t set = V2V_MotionTag value t (Following, ParallelDriving, Merging, ByPassing, Overtaking;
   * dataset/motion_tag_utils.py:17-22) is among PROMPT.CONDITION.MOTION_TAG.USED_TAGS -> weight
   * "condition_transformers.policy_decoder.condition_encoders.v2v_tag.tag_encoder.<tag>" [2 * hidden].  0 in the demo. */
  int32_t v2v_tag_mask;
  /* MODEL.POLICY.ACT_DECODER.TRAJ.PRED_GMM (act_decoder.py:26-27; 0 in the demo): state_dim 8 = x, y, h, (std1, std2, rho), xd, yd;
   * the rollout then appends the velocity of columns 6:8 instead of 3:5 (traj_sam.py:337-340). */
  int32_t pred_gmm;
  /* MODEL.POLICY.ACT_DECODER.TRAJ.PRED_MODE (act_decoder.py:47-76, :90-110): 0 = 'anchor' (every released config) and 'cluster'
   * (whose anchors cluster_mlp(FourierEmbeddingFix(k_goals)) do not depend on the input: the host folds them into
   * "policy.act_decoder.motion_anchors.weight", K rows repeated per agent type); 1 = 'mlp': no anchors, no CG_decode, the last
   * Linear of motion_head has motion_k * target_steps * state_dim (<= 128) outputs, mode-major. */
  int32_t k_pred_mlp;
  /* !MODEL.POLICY.ACT_DECODER.TRAJ.PRED_VEL (1 = PRED_VEL False, default.py:652's default; every released yaml sets True = 0 here):
   * the predicted state has no xd, yd -- state_dim 3, or 6 with pred_gmm -- no velocity track is kept ("vel" reads zeros), and
   * step_env takes the observation's velocity / acceleration columns from position differences over hist_steps + 2 steps
   * (traj_sam.py:251-260, :552-560); needs replan_freq >= 2. */
  int32_t no_pred_vel;
  /* !LOSS.ROLLOUT_TRAJ.USE_GOAL_PRED_LOSS (1 = False, default.py:440's default; the released yamls set True = 0 here): the checkpoint
   * has no "policy.act_decoder.pred_mlp.*" tensors and there is no "reconst_pred" result (act_decoder.py:75-76, :128-130). */
  int32_t no_reconst_pred;
  /* MODEL.REL_POS_EDGE_FUNC == 'knn' (default.py:455; 'radius' in every released yaml): the generator's (p2p, s2p) and the policy's
   * (a2p, m2p) edge sets are the *_max_neigh NEAREST tokens of the scene (torch_cluster.knn / knn_graph: sym_coord.py:85-96,
   * act_decoder.py:249-261) instead of the first *_max_neigh inside the radii; at most 2560 candidate tokens per scene */
  int32_t rel_pos_knn;
  /* MODEL.SCENE_ENCODER.MAP_TYPE / OBS_TYPE == 'mlp' (scene_encoder/map_encoder.py:5, obs_encoder.py:19, selected at
   * scene_encoder/base.py:20-21; 'pointnet' in every released yaml): the flat-MLP encoders have NO engine counterpart -- ps_create
   * refuses such a config with PS_E_ARG and says so, instead of running the PointNet kernels over another model's weights. */
  int32_t map_encoder_mlp, obs_encoder_mlp;
} ps_config;

/* Create an engine and upload weights.  names[i] are reference state_dict keys
 * (scene_encoder.* / decoder.* / policy.act_decoder.* / prompt_encoder.motion_pred.* /
 * condition_transformers.policy_decoder.*; models/base.py:141-147 load_state_dict), plus the
 * constant tables "const.fourier_div32/64/128" (the dim_t of FourierEmbeddingFix,
 * models/layers/fourier_embedding.py:68-69, computed by the host with the reference's ops).
 * Replaces: ProSim.__init__/_config_models (traj_sam.py:15-57). */
int ps_create(const ps_config* cfg, int32_t n_tensors, const char* const* names,
              const float* const* data, const int64_t* numel, ps_engine** out);
void ps_destroy(ps_engine* e);
const char* ps_last_error(void);

/* Upload one batch of scenes (the batch.extras the reference reads; dataset/format_utils.py:798-815).
 *   map_input [B,M,P,map_dim], map_mask [B,M,P], map_pos [B,M,2], map_head [B,M]
 *   obs_input [B,N,hist,obs_dim] (NaN allowed where masked), obs_mask [B,N,hist,obs_dim],
 *   obs_pos [B,N,2], obs_head [B,N]
 *   prompt [B,N,prompt_dim], prompt_mask [B,N], agent_type [B,N] (1..num_agent_types),
 *   prompt_pos [B,N,2], prompt_head [B,N]
 * Observed agents = slots with a valid history step; policy agents = prompt_mask slots, a non-empty subset of
 * the observed ones (traj_sam.py:246-250 matches them by id); a policy agent that is not observed, or a scene
 * batch without any policy agent, is PS_E_ARG.  Observed agents without a prompt replay ps_set_future_log. */
int ps_set_scene(ps_engine* e, int32_t B, int32_t M, int32_t P, int32_t N,
                 const float* map_input, const uint8_t* map_mask, const float* map_pos, const float* map_head,
                 const float* obs_input, const uint8_t* obs_mask, const float* obs_pos, const float* obs_head,
                 const float* prompt, const uint8_t* prompt_mask, const int32_t* agent_type,
                 const float* prompt_pos, const float* prompt_head);

/* Agents that ENTER the scene after the initial step (get_center_obs lists an agent only while its state is finite,
 * format_utils.py:383-388): call this BEFORE ps_set_scene with rows [B,N] != 0 for every slot that needs a token row
 * although its initial history has no valid step.  Such a row is no scene token (not a neighbour of anything) until a
 * ps_set_future_log frame carries a valid step for it.  Consumed by the next ps_set_scene; NULL clears. */
int ps_declare_agent_rows(ps_engine* e, int32_t B, int32_t N, const uint8_t* rows);

/* Replace only the prompt side of the uploaded batch (prompt [B,N,prompt_dim], prompt_pos [B,N,2],
 * prompt_head [B,N], agent_type [B,N]); prompt_mask must be unchanged.  Lets the decoder be called
 * after the scene encoder with its own prompt_enc argument (decoder/sym_coord.py:112). */
int ps_set_prompt(ps_engine* e, const float* prompt, const float* prompt_pos, const float* prompt_head,
                  const int32_t* agent_type);

/* Optional unary prompt conditions (dataset/condition_utils.py:126-222): goal (gx, gy, t) and
 * vehicle action tags (tag_id, t0, t1); *_pidx = prompt slot of each condition.  C = 0 or NULL
 * clears that type.  Replaces the `condition` argument of ConditionTransformer.forward
 * (models/condition_transformer/base.py:38). */
int ps_set_conditions(ps_engine* e, int32_t C_goal, const float* goal_input, const uint8_t* goal_mask,
                      const int32_t* goal_pidx, int32_t C_tag, const float* tag_input,
                      const uint8_t* tag_mask, const int32_t* tag_pidx);

/* Optional drag-point conditions, the third type of the demo config's PROMPT.CONDITION.TYPES
 * (dataset/condition_utils.py:401-447): drag_input [B,C,T,2] points in the prompt agent's start frame, a point
 * with a NaN coordinate is absent; drag_mask [B,C]; drag_pidx [B,C] prompt slot.  T <= 32.  Independent of
 * ps_set_conditions (either order); C = 0 or NULL clears.  Replaces DragPointEncoder.forward
 * (models/condition_transformer/condition_encoders.py:164-191) and its share of the pooled condition edge
 * (condition_attns.py:114-188).  PS_E_ARG if the engine was created with drag_mlp_layers = 0. */
int ps_set_drag_points(ps_engine* e, int32_t C_drag, int32_t T, const float* drag_input, const uint8_t* drag_mask,
                       const int32_t* drag_pidx);

/* Binary conditions 'v2v_tag' (V2V_MotionTagEncoder condition_encoders.py:148-150; _construct_cond_edge_matrix
 * condition_attns.py:141-166): pair_input [B, C, 3] = (V2V tag value, t0, t1), pair_mask [B, C], pair_pidx [B, C, 2] = the
 * prompt SLOTS of (source s, target t).  A valid row puts two edges into the condition layers' graph: s -> t with the
 * source half of the tag's parameter, t -> s with the target half (+ the temporal embedding on both); entries that share
 * an edge with other condition keys are mean-pooled with them; the edges' relative-PE rows use the two prompts' poses.
 * C_pair = 0 / NULL clears.  Independent of ps_set_conditions / ps_set_drag_points (each call replaces its own types). */
int ps_set_pair_conditions(ps_engine* e, int32_t C_pair, const float* pair_input, const uint8_t* pair_mask,
                           const int32_t* pair_pidx);
/* Optional per-replan observation frames fut_obs[t] for replans 1..R-1
 * (dataset/format_utils.py:667-687): input [R-1,B,N,hist,obs_dim]; only columns 8.. (extent,
 * type, time one-hot) are used -- columns 0..7 are overwritten by step_env
 * (traj_sam.py:266-270).  Default: the init_obs columns. */
int ps_set_future_obs(ps_engine* e, const float* fut_input);
/* The full log of the later replans -- batch.extras['fut_obs'][t] (traj_sam.py:221-270): observation
 * [R-1][B][N][hist][obs_dim], its mask, position [R-1][B][N][2] and heading [R-1][B][N].  Observed agents WITHOUT a
 * prompt (prompt_mask false on an observed slot of ps_set_scene) are log-replay agents: at replan t their scene token
 * is re-encoded from this log and sits at the logged pose; a row whose history is fully masked is no token at that
 * replan.  Policy agents' rows are overwritten by the simulation (only the static features 8.. are read). */
int ps_set_future_log(ps_engine* e, const float* fut_input, const uint8_t* fut_mask, const float* fut_pos,
                      const float* fut_head);
/* The motion mode every policy agent follows at every replan, for models with TRAJ.K > 1 and ROLLOUT.POLICY.TOP_K > 1
 * (traj_sam.py:300-313: torch.topk over motion_prob, then torch.randint among the top k).  motion_prob is all ones on this
 * path (act_decoder.py:124), so the pick does not depend on the model's output: the caller draws it -- with the
 * reference's own torch.topk / torch.randint calls to replay its stream exactly -- and hands the table over before the
 * rollout: choice [R, B, N] int32 in slot layout (entries of non-policy slots are ignored), NULL = mode 0 everywhere
 * (TOP_K = 1: torch.topk of equal probabilities returns index 0).  The table survives until the next ps_set_scene.
 * With replicas (below) the table is [R, replicas, N]: every replica draws its own modes. */
int ps_set_mode_choice(ps_engine* e, const int32_t* choice);
/* MODEL.POLICY.ACT_DECODER.RANDOM_NOISE_STD (act_decoder.py:113-115: motion[..., :2] += randn_like(...) * std, before the
 * cumulative sum, in every policy call): the reference's other source of replica diversity.  Like the mode pick, the draw
 * does not depend on the model's output, so the caller draws the whole table -- with the reference's own torch.randn_like
 * call, to replay its stream -- and hands it over before the rollout: noise [R, B (or replicas), N, motion_k, target_steps, 2]
 * float32 in slot layout, ALREADY scaled by the std; NULL switches the noise off.  Survives until the next ps_set_scene. */
int ps_set_action_noise(ps_engine* e, const float* noise);
/* M replicas of ONE scene rolled out side by side -- parallel_rollout_batch / replica_batch_for_parallel_rollout
 * (rollout/gpu_utils.py:59-123, :179-228: scene_embs, policy_emds, prompt_encs, agent_trajs and fut_obs .repeat(M, ...) on
 * the batch dim, then one rollout_batch over the M-batch).  Call before ps_set_scene (B must be 1 there); it holds until
 * changed.  The replicas are identical until the first mode draw, so ps_encode_scene and ps_generate_policy compute ONE
 * replica and fan its agent rows out on the device; the map tokens (and their k | v rows for the policy layers) exist
 * once and are every replica's candidates.  Afterwards the engine has replicas * (agents of the scene) agent rows,
 * replica-major: every per-agent result of ps_get / ps_pair_metric / ps_world_trajs has that many rows.  Inputs that are
 * per scene (ps_set_prompt, ps_set_future_obs / _log, ps_set_conditions, ps_set_drag_points) stay [1, N, ...] and apply to
 * every replica; ps_set_mode_choice is per replica.  replicas = 1 switches the mode off. */
int ps_set_replicas(ps_engine* e, int32_t replicas);
int32_t ps_num_replicas(ps_engine* e);
/* Agent rows (= observed agents, the order of every per-agent result of ps_get) that are policy agents. */
int32_t ps_num_policy_agents(ps_engine* e);
int ps_policy_flags(ps_engine* e, int32_t* flags, int64_t capacity);

/* scene_encoder(batch_obs, batch_map) -> token store on device (traj_sam.py:73-77;
 * scene_encoder/base.py:31-46, attn_fusion.py:78-134). */
int ps_encode_scene(ps_engine* e);
/* prompt_encoder + decoder(scene_embs, prompt_enc) + condition transformer
 * (traj_sam.py:79-142; decoder/sym_coord.py:112-140; condition_transformer/base.py:38-60). */
int ps_generate_policy(ps_engine* e);
/* init_agent_trajs (traj_sam.py:597-633): reset the trajectory state from init_obs. */
int ps_reset_rollout(ps_engine* e);
/* One iteration of rollout_batch (traj_sam.py:159-172): step_env -> decode_output (policy.forward)
 * -> step_agent_traj, for replan index t_idx (0-based). */
int ps_policy_step(ps_engine* e, int32_t t_idx);
/* ProSim.forward(batch,'val') (traj_sam.py:59-71): encode + generate + all replans, enqueued on
 * the engine's stream (replayed from a hipGraph after the first call for a given scene shape). */
int ps_rollout(ps_engine* e);
int ps_sync(ps_engine* e);
/* scene_encoder.update_scene_emb(scene_embs, batch_obs_new, old_ids) (attn_fusion.py:238-252; OBS_UPDATE FUSION 'replace',
 * ATTN_UPDATE False as in no_text.yaml:213-215): re-encode the agents from a new observation (same layout as
 * ps_set_scene's obs_*), replace their tokens and poses, reuse the map tokens.  The observed agents must be those of
 * ps_set_scene.  ps_policy_step does this on the device from the simulated state; this entry point serves callers
 * that drive the encoder by hand.  Read the tokens back with ps_get("scene_tokens"). */
int ps_update_obs(ps_engine* e, const float* obs_input, const uint8_t* obs_mask, const float* obs_pos, const float* obs_head);
/* update_scene_emb with ANOTHER agent set (_replace_old_obs, attn_fusion.py:205-236: the map part of scene_embs is kept, the
 * agent part is whatever the new observation lists): the caller reads the map tokens (ps_get "scene_tokens", first
 * ps_num_map_tokens rows), uploads the new batch with ps_set_scene (same map arrays), hands the map tokens back with this
 * call -- tokens [ps_num_map_tokens, hidden] on the HOST -- and re-encodes the agents with ps_update_obs.  The scene then
 * counts as encoded. */
int ps_set_map_tokens(ps_engine* e, const float* tokens, int64_t count);
/* Destination rows per workgroup of the fused attention launches = the engine's operating mode:
 *   0       latency mode, ONE rollout on the GPU: from 1024 destination rows up the fused chains run on k_chain16 with 4 rows
 *           per 8-wave workgroup (256 workgroups per 1024-row policy launch, two waves per row); smaller launches on
 *           the round-1 kernel k_attn_chain (one or two rows per 4-wave workgroup);
 *   8..16   throughput mode, several engines share the GPU: k_chain16 -- node Linears as 16-row MFMA GEMMs (a layer's
 *           weights cross a CU once per workgroup), rel-PE rows recomputed per 16-edge tile from 32 B of geometry per edge
 *           -- with that many rows per workgroup; 16 rows x 4 rollouts in flight = 256 workgroups, one per CU, is what
 *           bench.py runs (with GPU_MAX_HW_QUEUES=8: the runtime's default 4 hardware queues serialise four engines'
 *           streams).
 *   1, 2, 4 rows per workgroup of either kernel (k_chain16 shares a row's edge list between 8 / rows waves; k_attn_chain
 *           takes 2 or 4) -- experiments.
 * Results do not depend on it beyond fp32 summation order. */
int ps_set_chain_rows(ps_engine* e, int32_t rows);
/* Which fused-chain kernel runs the attention layers: 0 (default) = by mode as above -- in throughput mode (ps_set_chain_rows >= 8)
 * k_chain16 for everything INCLUDING the scene encoder's s2s layers (one k | v projection + one-step chain per layer), in latency mode
 * the split launches (node halves + k_edge_small) for the s2s layers; 1 = k_attn_chain always, 2 = k_chain16 always with the split s2s
 * path, 3 = k_chain16 always and for the s2s layers too.  Every choice holds the parity bar (profiles/r04_parity.json); they differ in
 * fp32 summation order.  Resets ps_set_chain_rows to 0 and invalidates the encoded / generated stages. */
int ps_set_chain_impl(ps_engine* e, int32_t impl);
/* Which kernels run the dense per-row stacks (PointNet encoders, the node half of the split s2s layers, k | v projections):
 * 0 (default) = the row-tile kernels of round 4 (ps_rowtile.h: a wave carries 16..80 rows through the whole stack in registers,
 * chained transposed MFMA GEMMs, no barriers), 1 = the staged kernels of rounds 1-3 (k_pointnet_mfma, k_node, k_kv_proj:
 * GEMM -> LDS -> barrier -> epilogue per Linear).  Both stay in the library: each is the other's cross-check.
 * 2 = as 0 with the split layers' edge half on the 16-row workgroup kernel (k_edge16) instead of the one-wave-per-row kernel
 * (k_edge_rows); 11..13 = as 0 with 1..3 row tiles per wave forced in the node halves -- both bit-identical to 0 (tests, A/B timing). */
int ps_set_row_impl(ps_engine* e, int32_t impl);
/* How a radius search whose edges feed a geometry-record chain is launched: 0 (default) = ONE launch (k_radius_geo: one scan of the
 * candidates per query, the CSR prefix taken from counts the waves publish to each other, esrc / edst and the 32-byte records written by the
 * search's own waves), 1 = the count / fill / k_edge_geo launches of rounds 1-4, 2 = as 0 with a look-back that never waits: a count that is
 * not published yet is recomputed by the waiting wave -- the path that makes the kernel independent of the order workgroups are dispatched in,
 * taken in normal operation only after ~0.5 ms of polling (tests).  Same edges in the same order, same records: bit-identical
 * results (tests/test_round5_gpu.py).  Replaces torch_cluster.radius / radius_graph + the rel-PE construction of act_decoder.py:203-221,
 * sym_coord.py:86-110 for those edge sets; searches with a learnable rel-PE, kNN edge functions or operand-image chains keep their launches. */
int ps_set_search_impl(ps_engine* e, int32_t impl);
/* Nodes (kernel launches and copies) of the captured rollout graph; 0 before the first ps_rollout or when the rollout runs eagerly. */
int64_t ps_graph_nodes(ps_engine* e);
/* The engine's hipStream_t (every entry point enqueues on it), so a host can order its own streams against the
 * engine with events instead of ps_sync -- bench.py overlaps the RCCL metric gather of rollout k with rollout k+1. */
void* ps_stream(ps_engine* e);

/* Stateless policy.forward -- the drop-in for Policy_RelPE_Temporal.forward(policy_emd, batch_obs,
 * batch_map, batch_pos, pair_names, latent_state) (policy/base.py:19; act_decoder.py:239-283, :78-140) on
 * caller-supplied tokens: the flattened VALID agent / map tokens in the reference's scene-major order
 * (what _process_scene_token makes of batch_obs / batch_map, act_decoder.py:224-237) and the A policy
 * rows (policy_emd['emd'], batch_pos, agent_type, batch_idx).  motion_pred [A, K, target_steps, state_dim],
 * fused [A, hidden] (may be NULL).  Synchronous. */
int ps_policy_forward(ps_engine* e, int32_t n_scenes, int32_t Na, const float* a_tok, const float* a_pos,
                      const float* a_ori, const int32_t* a_scene, int32_t Nm, const float* m_tok,
                      const float* m_pos, const float* m_ori, const int32_t* m_scene, int32_t A,
                      const float* p_emd, const float* p_pos, const float* p_ori, const int32_t* p_type,
                      const int32_t* p_scene, float* motion_pred, float* fused_out);

/* Overwrite the trajectory state (test hook for open-loop parity): traj [A, steps, 4], vel
 * [A, steps, 2] over the compact policy-agent list, `steps` = hist + t_idx*replan_freq. */
int ps_set_state(ps_engine* e, int32_t steps, const float* traj, const float* vel);

/* Copy a named result to the host.  Names / shapes (A = number of valid agents, compact,
 * scene-major; Mv = valid polylines):
 *   "traj" [A, max_steps, 4] (x, y, sin, cos in the agent-init frame; traj_sam.py:588)
 *   "vel" [A, max_steps, 2]; "motion_pred" [R, A, K, target_steps, state_dim] (K = motion_k modes);
 *   "reconst_pred" [A, 2]; "policy_emd" [A, hidden];
 *   "goal_prob" [A, goal_pred_k], "goal_point" [A, goal_pred_k, 2] (decoder goal heads, when enabled); "scene_tokens" [Mv + A, hidden];
 *   "fused" [A, hidden] (last policy step); "obs_in" [A, hist, obs_dim] (last step_env -- with up to 128 agents the step_env of replan t + 1
 *   runs in the tail of replan t's head launch, so after ps_policy_step(t), t < R - 1, this is already the observation of replan t + 1);
 *   "edge_counts" [8] (a2a, s2s, p2p, s2p, a2p, m2p of the last step, cond, 0) as float.
 * Returns the number of floats written, or a negative error. */
int64_t ps_get(ps_engine* e, const char* name, float* dst, int64_t capacity);
/* Serving a stream of new batches (no counterpart in the reference, which runs its batches one after the other,
 * rollout/callbacks.py:76): ps_get without its two synchronisations.  The copy of a per-agent result ("traj", "vel",
 * "motion_pred", "reconst_pred", "policy_emd", "fused", "goal_prob", "goal_point") is enqueued on the engine's stream behind
 * whatever the stream holds -- call it right after ps_rollout -- into memory the caller owns (pinned host memory makes the call
 * return at once); the caller waits for the stream, or for an event it records on ps_stream(), before reading.  Together with
 * the upload staging of ps_set_scene (two pinned arenas in turn, no synchronisation) this lets a caller queue batch n + 1
 * behind batch n on one engine.  Returns the number of floats that will be written, or a negative error. */
int64_t ps_get_async(ps_engine* e, const char* name, float* dst, int64_t capacity);
/* out[0] = rollouts that had to capture and instantiate their hipGraph, out[1] = rollouts that followed a setter
 * (ps_set_scene, ps_set_conditions, ...) and could keep the graph they had because nothing their launch sequence is made
 * of -- row counts, flags, device pointers -- had changed (a stream of batches of one shape replays one graph). */
int ps_graph_stats(ps_engine* e, int64_t* out);
/* Closed-loop displacement metric on the device: per agent row the mean distance between the rolled-out xy and a
 * ground-truth future over the steps whose ground truth is finite, and the distance at the LAST such step (the
 * NaN-masked target / last-valid-index conventions of metrics/motion_pred.py:31-76).  gt_dev [A, max_steps, 2] is a
 * device pointer to the future in the agent-init frame (NaN = no ground truth), or NULL: then the two numbers are the
 * mean / final distance from the origin of the agent-init frame (a path length, not an error).  out_dev [A, 2] is a
 * caller-owned DEVICE buffer (e.g. a torch tensor handed to an RCCL all-gather); rows of log-replay agents and of
 * agents without a valid step are NaN.  Enqueued on the engine's stream. */
int ps_rollout_metric(ps_engine* e, const float* gt_dev, float* out_dev);
/* The reference's validation metric on the device -- PairMotionPred (metrics/motion_pred.py:111-199): per (replan,
 * agent) pair the ADE / FDE of the arg-max-probability mode and the minima over the K modes (_update_traj_error
 * :31-76), and the rollout ADE of the chained per-replan predictions against the chained targets (_compute_traj_ade
 * :125-143 over loss/loss_func.py:215-313 rollout_traj / rollout_temp_traj_preds).  Device pointers:
 * tgt_dev [R, A, target_steps, 5] local targets per replan (io_pairs_batch['tgt'] in agent-row order, NaN = missing),
 * pair_mask_dev [R, A] uint8 (io_pairs_batch['mask']), prob_dev [R, A, K] mode probabilities or NULL (mode 0),
 * out_dev [A, 10] = per agent row (sum ade, sum fde, sum min_ade, sum min_fde, the four counts of finite entries behind
 * them, rollout ade, 1 if the agent has a valid rollout step); rows of log-replay agents are NaN.  The logged scalars
 * are sums / counts over the gathered rows and the mean of the valid agents' rollout ade
 * (prosim_amd.distributed.reduce_pair_metrics).  Enqueued on the engine's stream after the rollout. */
int ps_pair_metric(ps_engine* e, const float* tgt_dev, const uint8_t* pair_mask_dev, const float* prob_dev, float* out_dev);
/* obtain_rollout_trajs_in_world (rollout/gpu_utils.py:230-281): the rolled-out steps of every agent row in the world
 * frame -- rotate the agent-init-frame xy by the initial heading and add the initial position (batch_rotate_2D,
 * models/utils/geometry.py:19-22), heading = wrap_angle(atan2(sin, cos) + initial heading) (:13-17), then the centre ->
 * world matrix on points and angles (batch_nd_transform_points_pt / _angles_pt, rollout/utils.py:347-392).
 * center_to_world: HOST pointer, row-major 3 x 3 (batch.centered_world_from_agent_tf[0]) or NULL (identity);
 * out_dev: DEVICE pointer [A, max_steps, 3] float32 (x, y, heading), or NULL to keep the result in the engine
 * (ps_get "world_traj").  fp32 in the reference's order of operations.  Enqueued on the engine's stream. */
int ps_world_trajs(ps_engine* e, const float* center_to_world, float* out_dev);
int32_t ps_num_agents(ps_engine* e);
int32_t ps_num_map_tokens(ps_engine* e);

/* Timing on the engine's own stream (HIP events): run `iters` rollouts after `warmup`, return
 * average ms per rollout in *ms_rollout and, if stage_ms != NULL, per-stage averages
 * [encode_scene, generate_policy, replan loop]. */
int ps_time_rollout(ps_engine* e, int32_t warmup, int32_t iters, float* ms_rollout, float* stage_ms);
/* Average duration (ms) of the dominant kernel (the fused policy attention chain) over the
 * launches of the last ps_time_rollout / ps_rollout, measured with HIP events around each launch. */
int ps_time_policy_kernel(ps_engine* e, int32_t iters, float* ms_kernel);

/* Launch durations of the dominant kernel INSIDE a pipelined run: with events enabled every rollout records an event pair
 * around each of its policy-chain launches on the engine's stream (such rollouts are launched eagerly instead of
 * replayed from the graph: ROCm 7.2 cannot time events recorded by graph nodes); ps_policy_event_times syncs the stream
 * and returns the R durations (ms) of the engine's LAST rollout -- while other engines' rollouts shared the GPU, if the
 * host kept them in flight.  Returns R or a negative error. */
int ps_enable_policy_events(ps_engine* e, int32_t on);
int ps_policy_event_times(ps_engine* e, float* ms, int32_t capacity);

/* Unit-test hooks for single primitives (parity against tests/golden/ref_pure_primitives.npz). */
int ps_test_pointnet(ps_engine* e, int32_t which /*0 map, 1 obs*/, int32_t n_poly, int32_t P,
                     const float* x, const uint8_t* point_mask, float* out);
/* the same with the row-tile kernel forced to `mt` row tiles per wave (1..5; 0 = the engine's choice, -1 = the staged kernel of
 * rounds 1-3), and, with iters > 0, the mean launch time in ms (HIP events on the engine's stream) */
int ps_test_pointnet_mt(ps_engine* e, int32_t which, int32_t n_poly, int32_t P, const float* x, const uint8_t* point_mask,
                        float* out, int32_t mt, int32_t iters, float* ms_out);
int ps_test_fourier(ps_engine* e, int32_t n, const float* x4, float* out128);
int ps_test_wrap(ps_engine* e, int32_t n, const float* x, float* out);
/* One AttentionLayer (models/layers/attention_layer.py:56-121) on caller-supplied tokens and a
 * CSR-by-destination edge list; rt = relative-PE rows already LayerNorm-normalised (no affine).
 * layer_index counts over [a2a | s2s | p2p | s2p | a2p | m2p | cond] layers.  T selects the kernel build:
 * 0 auto, 1 / 2 / 4 destination rows per 256-thread workgroup of the fused chain, 11 = 1 row on 4 waves built for two workgroups per CU, 18 = 1 row on 8 waves, 84 = 4 rows on 8 waves,
 * 16 = the split layer (k_node + k_edge_small + k_node; falls back to the fused chain above degree 128). */
int ps_test_attn(ps_engine* e, int32_t layer_index, int32_t Ns, int32_t Nd, int32_t E, const float* x_src,
                 const float* x_dst, const float* rt, const int32_t* eoff, const int32_t* esrc, int32_t T, float* out);
/* Read back an edge set built by the last stage: which = 0 a2a, 1 s2s, 2 p2p, 3 s2p, 4 a2p, 5 m2p.
 * Returns the edge count; rt (optional) receives the normalised rel-PE rows [E][128] rebuilt from the
 * split-fp16 operand image (columns 96..127 repeat 64..95). */
int64_t ps_test_get_edges(ps_engine* e, int32_t which, int32_t* esrc, int32_t* edst, float* rt, int64_t capacity);
/* Micro-benchmark: nwg workgroups stream the same mbytes buffer (depth x 16 float4 in flight per thread). */
int ps_test_stream(ps_engine* e, int32_t mbytes, int32_t nwg, int32_t depth, int32_t iters, float* ms_out);

#ifdef __cplusplus
}
#endif
#endif /* PROSIM_HIP_H */
