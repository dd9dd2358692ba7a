"""Model spec for the closed-loop rollout path.

A plain dataclass mirror of the ``MODEL.*`` / ``DATASET.FORMAT.*`` / ``ROLLOUT.*``
values the reference reads on this path (reference: prosim_demo/cfg/no_text.yaml:212-279
over prosim/config/default.py).  No yacs, no Lightning: the spec is the only
configuration object the engine, the oracle and the weight initialiser share.
"""
from __future__ import annotations

import dataclasses
from dataclasses import dataclass, field
from typing import List, Tuple

# V_Action_MotionTag enum order (reference: prosim/dataset/motion_tag_utils.py:4-15).
V_ACTION_TAGS: Tuple[str, ...] = (
    "Stopping", "Accelerate", "Decelerate", "KeepSpeed", "LeftLaneChange",
    "RightLaneChange", "KeepLane", "LeftTurn", "RightTurn", "Straight", "Parked",
)
# PROMPT.CONDITION.MOTION_TAG.USED_TAGS order (no_text.yaml:72) -- this is the order in
# which MotionTagEncoder.forward emits per-tag condition entries
# (condition_encoders.py:58,108).
USED_V_ACTION_TAGS: Tuple[str, ...] = (
    "Accelerate", "Decelerate", "KeepSpeed", "Stopping", "LeftLaneChange",
    "RightLaneChange", "KeepLane", "LeftTurn", "RightTurn", "Straight", "Parked",
)
# V2V_MotionTag enum order (motion_tag_utils.py:17-22): binary (agent-pair) tags
V2V_TAGS: Tuple[str, ...] = ("Following", "ParallelDriving", "Merging", "ByPassing", "Overtaking")


@dataclass(frozen=True)
class ModelSpec:
    # MODEL.HIDDEN_DIM (no_text.yaml:217)
    hidden: int = 128
    # *.ATTN.NUM_HEAD / FF_DIM -- FF_DIM is fed to AttentionLayer(head_dim=...)
    # (attn_fusion.py:32, act_decoder.py:190, sym_coord.py:29).
    heads: int = 8
    head_dim: int = 16
    # MODEL.SCENE_ENCODER.ATTN (no_text.yaml:226-232)
    scene_layers: int = 6
    scene_knn: int = 32                # MAX_NUM_NEIGH; agents use min(4*k, 100) (attn_fusion.py:107)
    # MODEL.OBS_UPDATE (default.py:499-501; no_text.yaml:213-215 keeps the defaults): how update_scene_emb folds the
    # re-encoded observation into the agent tokens at replans 1.. ('replace' | 'mlp'), and whether the agents then
    # re-attend to each other and to the map (radii: SCENE_ENCODER.ATTN.AGENT_RADIUS / SCENE_RADIUS, default.py:479-480;
    # neighbour cap = scene_knn)
    obs_fusion: str = "replace"
    obs_attn_update: bool = False
    enc_agent_radius: float = 100.0
    enc_scene_radius: float = 50.0
    # MODEL.DECODER.ATTN (no_text.yaml:243-251)
    dec_layers: int = 6
    dec_prompt_radius: float = 300.0
    dec_scene_radius: float = 300.0
    dec_max_neigh: int = 512
    # MODEL.DECODER.GOAL_PRED (default.py:610-614; ENABLE False in the demo): K goal hypotheses per prompt from the
    # decoder's embedding -- goal_prob [B, N, K], goal_point [B, N, K, 2] (decoder/base.py:22-58); 0 = disabled
    goal_pred_k: int = 0
    # MODEL.POLICY.ACT_DECODER.ATTN (no_text.yaml:270-279)
    pol_layers: int = 6
    pol_agent_radius: float = 100.0
    pol_map_radius: float = 50.0
    pol_max_neigh: int = 768
    # MODEL.CONDITION_TRANSFORMER (no_text.yaml:281-288, default.py:521-524)
    cond_layers: int = 3
    # CONDITION_ENCODER.DRAG_POINTS PointNet (default.py:531-533): [x, y] points -> hidden
    drag_pre_layers: int = 1
    drag_mlp_layers: int = 3
    # DATASET.FORMAT.HISTORY (no_text.yaml:195-200): 8 elems + 2 extent + 3 type + 11 time one-hot
    hist_steps: int = 11
    obs_dim: int = 24
    # DATASET.FORMAT.MAP: 4 coords + type + tl + 3 type one-hot + 2 dir (format_utils.py:249-261)
    map_dim: int = 11
    # MODEL.SCENE_ENCODER.MAP_TYPE / OBS_TYPE (scene_encoder/base.py:20-21 picks map_encoders[...] / obs_encoders[...]; 'pointnet' in every
    # released yaml).  'mlp' (map_encoder.py:5, obs_encoder.py:19: a flat MLP over the concatenated points) is a legal value of the
    # reference's registry that this engine does NOT build: Engine / ps_create refuse it loudly (round 6)
    map_encoder_type: str = "pointnet"
    obs_encoder_type: str = "pointnet"
    # MODEL.{MAP,OBS}_ENCODER.POINTNET (default.py:488-497)
    map_pre_layers: int = 3
    map_mlp_layers: int = 5
    obs_pre_layers: int = 1
    obs_mlp_layers: int = 3
    # DATASET.FORMAT.TARGET (no_text.yaml:186-190 + default.py:725-730): x,y,h,xd,yd
    target_steps: int = 10
    state_dim: int = 5
    # MODEL.POLICY.ACT_DECODER.TRAJ.PRED_GMM (act_decoder.py:26-27; False in the demo, no_text.yaml:262): three more regressed
    # channels per step (std1, std2, rho; read by the training loss only) -- state_dim 8, laid out x, y, h, (std1, std2, rho),
    # xd, yd: the rollout then takes the velocity from columns 6:8 instead of 3:5 (traj_sam.py:337-340)
    pred_gmm: bool = False
    # MODEL.POLICY.ACT_DECODER.TRAJ.PRED_VEL (True in every released yaml; default.py:652 says False): without it the target has no
    # xd, yd (default.py:725-730) -- state_dim 3 (6 with PRED_GMM) -- the rollout keeps no velocity track, and step_env derives the
    # observation's velocity / acceleration from position differences over hist + 2 steps (traj_sam.py:251-260, :552-560)
    pred_vel: bool = True
    # LOSS.ROLLOUT_TRAJ.USE_GOAL_PRED_LOSS (True in the released yamls, no_text.yaml:111; default.py:440 says False): with it the act
    # decoder carries pred_mlp and every policy call returns reconst_pred = pred_mlp(policy_emd) (act_decoder.py:75-76, :128-130);
    # without it a checkpoint has no pred_mlp tensors and the output no 'reconst_pred'
    use_goal_pred_loss: bool = True
    # MODEL.POLICY.ACT_DECODER.RANDOM_NOISE_STD (act_decoder.py:113-115; 0 in the demo): Gaussian noise on every predicted
    # xy step before the cumulative sum -- what makes the M replicas of parallel_rollout_batch differ when TOP_K = K = 1.
    # Drawn by the host with the reference's own torch.randn_like call (ProSimHip), handed to the engine as a table.
    action_noise_std: float = 0.0
    motion_k: int = 1                  # MODEL.POLICY.ACT_DECODER.TRAJ.K
    # MODEL.POLICY.ACT_DECODER.TRAJ.PRED_MODE (act_decoder.py:47-76, :90-110; 'anchor' in every released config, 'mlp' is
    # default.py:650's default): 'anchor' -- K learned anchors per agent type through CG_decode; 'cluster' -- the K anchors are
    # cluster_mlp(FourierEmbeddingFix(k_goals)) of the K x 2 goal clusters in TRAJ.CLUSTER_PATH, the same for every type (a
    # constant of the checkpoint: folded into the anchor table when the weights are handed to the engine); 'mlp' -- no anchors
    # and no CG_decode, motion_head regresses all K modes at once (its last Linear has K * steps * state_dim outputs)
    k_pred_mode: str = "anchor"
    rollout_top_k: int = 1             # ROLLOUT.POLICY.TOP_K (default.py:136): modes a rollout step draws from (host-side draw)
    num_agent_types: int = 3           # DATASET.USE_PED_CYCLIST -> anchors K*3 (act_decoder.py:66-68)
    prompt_dim: int = 7                # v_local(2)+extent(2)+type one-hot(3) (prompt_utils.py:111-150)
    # ROLLOUT.POLICY (no_text.yaml:44-48)
    replan_freq: int = 10
    max_steps: int = 80
    dt: float = 0.1                    # DATASET.MOTION.DT
    ln_eps: float = 1e-5
    fourier_temperature: float = 10000.0
    # *.ATTN.LEARNABLE_PE / PE_NUM_FREQ (default.py:472-473, :594-595, :665-666; False in the demo, no_text.yaml:229,246,273):
    # the relative-PE rows of the scene encoder's (a2a, s2s), the generator's (p2p, s2p) and the policy's (a2p, m2p)
    # edge sets come from a learnable FourierEmbedding (layers/fourier_embedding.py:11-54) over 3 inputs instead of the
    # fixed FourierEmbeddingFix over 4.  The condition layers always use the fixed one (condition_attns.py:93).
    # the V2V (agent-pair) tags among PROMPT.CONDITION.MOTION_TAG.USED_TAGS, in that list's order (condition_encoders.py:58):
    # a 'v2v_tag' condition between prompts s and t puts an edge s -> t and an edge t -> s into the condition layers'
    # graph (condition_attns.py:114-188).  Empty in the demo config ('v2v_tag' is not among its PROMPT.CONDITION.TYPES).
    used_v2v_tags: Tuple[str, ...] = ()
    # MODEL.REL_POS_EDGE_FUNC (default.py:455; 'radius' in every released yaml): 'knn' builds the generator's (p2p, s2p) and the policy's
    # (a2p, m2p) edge sets from the max_neigh NEAREST tokens of the scene instead of the first max_neigh inside a radius
    # (decoder/sym_coord.py:85-96, policy/act_decoder.py:249-261; the scene encoder always uses knn, attn_fusion.py:107-109)
    rel_pos_edge_func: str = "radius"
    enc_learnable_pe: bool = False
    dec_learnable_pe: bool = False
    pol_learnable_pe: bool = False
    pe_num_freq: int = 64

    def __post_init__(self):
        if self.state_dim != 3 + 2 * self.pred_vel + 3 * self.pred_gmm:
            raise ValueError(f"state_dim {self.state_dim}: 3 (x, y, h) + 2 with pred_vel + 3 with pred_gmm (use ModelSpec.replace, which keeps it in step)")
        for k in (self.map_encoder_type, self.obs_encoder_type):
            if k not in ("pointnet", "mlp"):
                raise ValueError(f"map_encoder_type / obs_encoder_type {k!r}: 'pointnet' or 'mlp' (the reference's registry keys)")
        if self.rel_pos_edge_func not in ("radius", "knn"):
            raise ValueError(f"rel_pos_edge_func {self.rel_pos_edge_func!r}: 'radius' or 'knn'")
        if self.k_pred_mode not in ("anchor", "cluster", "mlp"):
            raise ValueError(f"k_pred_mode {self.k_pred_mode!r}: 'anchor', 'cluster' or 'mlp'")

    @property
    def agent_knn(self) -> int:
        return min(self.scene_knn * 4, 100)

    @property
    def all_t_indices(self) -> List[int]:
        # format_utils.py:699-713 with TAIL_PADDING=True, SAMPLE_RATE=10, split=ROLLOUT
        return list(range(0, self.max_steps, self.replan_freq))

    @property
    def n_replans(self) -> int:
        return len(self.all_t_indices)

    @property
    def out_dim(self) -> int:
        return self.target_steps * self.state_dim

    @property
    def head_out_dim(self) -> int:
        """Outputs of motion_head's last Linear (act_decoder.py:58-61)."""
        return self.out_dim * (self.motion_k if self.k_pred_mode == "mlp" else 1)

    @property
    def vel_col(self) -> int:
        """First of the two velocity columns of a predicted step (traj_sam.py:337-340); -1 without PRED_VEL."""
        return (6 if self.pred_gmm else 3) if self.pred_vel else -1

    def replace(self, **kw) -> "ModelSpec":
        pv, pg = kw.get("pred_vel", self.pred_vel), kw.get("pred_gmm", self.pred_gmm)
        kw.setdefault("state_dim", 3 + 2 * pv + 3 * pg)   # (the head options fix the state width; __post_init__ checks)
        return dataclasses.replace(self, **kw)


def wrap_angle_np(a):
    """models/utils/geometry.py:13-17 on a float32 numpy array (torch '%' is floor-mod)."""
    import numpy as np
    a = np.asarray(a, np.float32)
    return (-np.float32(np.pi) + np.mod(a + np.float32(np.pi), np.float32(2 * np.pi))).astype(np.float32)


DEMO_SPEC = ModelSpec()
# Reduced-depth variant used for the committed golden fixtures and fast CPU tests
# (same kernels, fewer layers).
SMALL_SPEC = ModelSpec(scene_layers=2, dec_layers=2, pol_layers=2, cond_layers=1)
