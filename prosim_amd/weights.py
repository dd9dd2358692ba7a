"""Weight container for the rollout path.

Weights are a flat ``{name: float32 ndarray}`` dict keyed exactly like the reference
``ProSim.state_dict()`` (``scene_encoder.* / decoder.* / policy.act_decoder.* /
prompt_encoder.motion_pred.* / condition_transformers.policy_decoder.*``; see SURVEY.md
section 5 "checkpoint / resume"), so a released ``.ckpt`` can be loaded with
:func:`from_state_dict` and a reference module can ``load_state_dict`` what
:func:`init_weights` makes (tests/gen_golden.py does exactly that, strict=True).

No trained checkpoint ships with the reference (README.md:58 points to Google Drive), so
tests, smoke and bench use :func:`init_weights`: a seeded initialiser whose draw order is
the sorted parameter-name order.  LayerNorm affine terms are perturbed away from (1, 0)
on purpose -- an identity LayerNorm would hide a wrong gamma/beta in a kernel.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Tuple

import numpy as np
import torch

from .spec import ModelSpec, USED_V_ACTION_TAGS

Shapes = "OrderedDict[str, Tuple[int, ...]]"


def _mlp_shapes(out: Dict, prefix: str, dims: List[int], ret_before_act: bool, without_norm: bool) -> None:
    """Parameter names of the reference ``MLP`` (models/layers/mlp.py:475-494): an
    nn.Sequential of Linear [, LayerNorm], ReLU per hidden layer."""
    idx = 0
    for i in range(len(dims) - 1):
        out[f"{prefix}.mlp.{idx}.weight"] = (dims[i + 1], dims[i])
        out[f"{prefix}.mlp.{idx}.bias"] = (dims[i + 1],)
        idx += 1
        if i < len(dims) - 2:
            if not without_norm:
                out[f"{prefix}.mlp.{idx}.weight"] = (dims[i + 1],)
                out[f"{prefix}.mlp.{idx}.bias"] = (dims[i + 1],)
                idx += 1
            idx += 1  # ReLU


def mlp_layout(dims: List[int], ret_before_act: bool, without_norm: bool) -> List[Tuple[int, int]]:
    """[(linear_seq_idx, ln_seq_idx or -1)] for each Linear of a reference MLP."""
    res = []
    idx = 0
    for i in range(len(dims) - 1):
        lin = idx
        idx += 1
        ln = -1
        if i < len(dims) - 2:
            if not without_norm:
                ln = idx
                idx += 1
            idx += 1
        res.append((lin, ln))
    return res


def _pointnet_shapes(out: Dict, prefix: str, in_dim: int, hidden: int, n_pre: int, n_mlp: int) -> None:
    # pointnet_encoder.py:19-22
    _mlp_shapes(out, f"{prefix}.pre_mlps", [in_dim] + [hidden] * n_pre, False, False)
    _mlp_shapes(out, f"{prefix}.mlps", [hidden * 2] + [hidden] * (n_mlp - n_pre), False, False)
    _mlp_shapes(out, f"{prefix}.out_mlps", [hidden] * 3, True, True)


def _attn_shapes(out: Dict, prefix: str, d: int, hd: int, bipartite: bool) -> None:
    # attention_layer.py:28-54
    out[f"{prefix}.to_q.weight"] = (hd, d)
    out[f"{prefix}.to_q.bias"] = (hd,)
    out[f"{prefix}.to_k.weight"] = (hd, d)
    out[f"{prefix}.to_v.weight"] = (hd, d)
    out[f"{prefix}.to_v.bias"] = (hd,)
    out[f"{prefix}.to_k_r.weight"] = (hd, d)
    out[f"{prefix}.to_v_r.weight"] = (hd, d)
    out[f"{prefix}.to_v_r.bias"] = (hd,)
    out[f"{prefix}.to_s.weight"] = (hd, d)
    out[f"{prefix}.to_s.bias"] = (hd,)
    out[f"{prefix}.to_g.weight"] = (hd, hd + d)
    out[f"{prefix}.to_g.bias"] = (hd,)
    out[f"{prefix}.to_out.weight"] = (d, hd)
    out[f"{prefix}.to_out.bias"] = (d,)
    out[f"{prefix}.ff_mlp.0.weight"] = (4 * d, d)
    out[f"{prefix}.ff_mlp.0.bias"] = (4 * d,)
    out[f"{prefix}.ff_mlp.3.weight"] = (d, 4 * d)
    out[f"{prefix}.ff_mlp.3.bias"] = (d,)
    names = ["attn_prenorm_x_src", "attn_prenorm_r", "attn_postnorm", "ff_prenorm", "ff_postnorm"]
    if bipartite:
        names.insert(1, "attn_prenorm_x_dst")
    for n in names:
        out[f"{prefix}.{n}.weight"] = (d,)
        out[f"{prefix}.{n}.bias"] = (d,)


def _pe_emb_shapes(out: Dict, prefix: str, d: int, nf: int) -> None:
    """FourierEmbedding(input_dim=3, hidden_dim=d, num_freq_bands=nf) (layers/fourier_embedding.py:11-35)."""
    out[f"{prefix}.freqs.weight"] = (3, nf)
    for i in range(3):
        out[f"{prefix}.mlps.{i}.0.weight"] = (d, 2 * nf + 1)
        out[f"{prefix}.mlps.{i}.0.bias"] = (d,)
        out[f"{prefix}.mlps.{i}.1.weight"] = (d,)
        out[f"{prefix}.mlps.{i}.1.bias"] = (d,)
        out[f"{prefix}.mlps.{i}.3.weight"] = (d, d)
        out[f"{prefix}.mlps.{i}.3.bias"] = (d,)
    out[f"{prefix}.to_out.0.weight"] = (d,)
    out[f"{prefix}.to_out.0.bias"] = (d,)
    out[f"{prefix}.to_out.2.weight"] = (d, d)
    out[f"{prefix}.to_out.2.bias"] = (d,)


def pe_emb_prefixes(spec: ModelSpec) -> List[str]:
    """State-dict prefixes of the learnable relative-PE embeddings this spec has (attn_fusion.py:24-26,
    sym_coord.py:22-24, act_decoder.py:181-183)."""
    out = []
    if spec.enc_learnable_pe:
        out += ["scene_encoder.a2a_rel_pe_emb", "scene_encoder.s2s_rel_pe_emb"]
    if spec.dec_learnable_pe:
        out += ["decoder.p2p_rel_pe_emb", "decoder.s2p_rel_pe_emb"]
    if spec.pol_learnable_pe:
        out += ["policy.act_decoder.a2p_rel_pe_emb", "policy.act_decoder.m2p_rel_pe_emb"]
    return out


def param_shapes(spec: ModelSpec) -> "OrderedDict[str, Tuple[int, ...]]":
    """Every learnable tensor on the rollout path, reference state_dict naming.

    For non-bipartite attention layers the reference registers ONE LayerNorm under two
    attribute names (attention_layer.py:48-49), so its state_dict carries
    ``attn_prenorm_x_dst.*`` aliases; they are not separate parameters and are omitted here
    (:func:`to_reference_state_dict` adds them back).
    """
    d = spec.hidden
    hd = spec.heads * spec.head_dim
    out: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    _pointnet_shapes(out, "scene_encoder.map_encoder", spec.map_dim, d, spec.map_pre_layers, spec.map_mlp_layers)
    _pointnet_shapes(out, "scene_encoder.obs_encoder", spec.obs_dim, d, spec.obs_pre_layers, spec.obs_mlp_layers)
    for i in range(spec.scene_layers):
        _attn_shapes(out, f"scene_encoder.a2a_attn_layers.{i}", d, hd, False)
        _attn_shapes(out, f"scene_encoder.s2s_attn_layers.{i}", d, hd, False)
    if spec.obs_fusion == "mlp":    # attn_fusion.py:18-19
        _mlp_shapes(out, OBS_UPDATE_MLP, [2 * d, d, d], True, False)
    # prompt_encoder/base.py:23-34
    _mlp_shapes(out, "prompt_encoder.motion_pred.state_encoder", [spec.prompt_dim, d, d], True, False)
    for i in range(spec.dec_layers):
        _attn_shapes(out, f"decoder.p2p_attn_layers.{i}", d, hd, False)
        _attn_shapes(out, f"decoder.s2p_attn_layers.{i}", d, hd, True)
    if spec.goal_pred_k > 0:        # decoder/base.py:18-20
        _mlp_shapes(out, "decoder.goal_prob_head", [d, d // 2, spec.goal_pred_k], True, False)
        _mlp_shapes(out, "decoder.goal_point_head", [d, d // 2, spec.goal_pred_k * 2], True, False)
    pa = "policy.act_decoder"
    for i in range(spec.pol_layers):
        _attn_shapes(out, f"{pa}.a2p_attn_layers.{i}", d, hd, True)
        _attn_shapes(out, f"{pa}.m2p_attn_layers.{i}", d, hd, True)
    # act_decoder.py:47-76 (TRAJ.PRED_MODE, USE_GOAL_PRED_LOSS)
    _mlp_shapes(out, f"{pa}.motion_head", [d, d, d // 2, spec.head_out_dim], True, False)
    if spec.k_pred_mode != "mlp":
        for i in range(3):
            out[f"{pa}.CG_decode.CGs.{i}.MLP.0.weight"] = (d, d)
            out[f"{pa}.CG_decode.CGs.{i}.MLP.0.bias"] = (d,)
            out[f"{pa}.CG_decode.CGs.{i}.MLP.1.weight"] = (d,)
            out[f"{pa}.CG_decode.CGs.{i}.MLP.1.bias"] = (d,)
    if spec.k_pred_mode == "anchor":
        out[f"{pa}.motion_anchors.weight"] = (spec.motion_k * spec.num_agent_types, d)
    elif spec.k_pred_mode == "cluster":
        _mlp_shapes(out, f"{pa}.{CLUSTER_MLP}", [d, d], False, False)   # Linear + ReLU (:74)
        out[CLUSTER_GOALS] = (spec.motion_k, 2)   # the content of TRAJ.CLUSTER_PATH (:72): not a state_dict entry
    if spec.use_goal_pred_loss:
        _mlp_shapes(out, f"{pa}.pred_mlp", [d, d, d // 2, 2], True, False)
    # condition transformer at 'policy_decoder' (traj_sam.py:47-52; condition_encoders.py, condition_attns.py)
    ct = "condition_transformers.policy_decoder"
    _mlp_shapes(out, f"{ct}.condition_encoders.goal.goal_encoder", [2, d, d], True, True)
    for tag in USED_V_ACTION_TAGS:
        out[f"{ct}.condition_encoders.v_action_tag.tag_encoder.{tag}"] = (d,)
    for tag in spec.used_v2v_tags:   # V2V_MotionTagEncoder: one [2 d] parameter per used tag (source half | target half)
        out[f"{ct}.condition_encoders.v2v_tag.tag_encoder.{tag}"] = (2 * d,)
    for i in range(spec.cond_layers):
        _attn_shapes(out, f"{ct}.condition_attn.attn_layers.{i}", d, hd, False)
    # DragPointEncoder (condition_encoders.py:152-191): a PointNet over the [x, y] drag points
    if spec.drag_mlp_layers > 0:   # 0: a checkpoint trained without 'drag_point' in PROMPT.CONDITION.TYPES
        _pointnet_shapes(out, f"{ct}.{DRAG_ENCODER}", 2, d, spec.drag_pre_layers, spec.drag_mlp_layers)
    for prefix in pe_emb_prefixes(spec):
        _pe_emb_shapes(out, prefix, d, spec.pe_num_freq)
    return out


DRAG_ENCODER = "condition_encoders.drag_point.pointnet_encoder"
OBS_UPDATE_MLP = "scene_encoder.obs_update_mlp"
PE_EMB = "_rel_pe_emb."
V2V_ENCODER = "condition_encoders.v2v_tag."
CLUSTER_MLP = "cluster_mlp"
ENGINE_PE_FREQ = 64   # frequency bands of the engine's learnable relative-PE kernel (k_pe_learn)
CLUSTER_GOALS = "policy.act_decoder.k_goals"
_LATE = (DRAG_ENCODER, OBS_UPDATE_MLP, PE_EMB, V2V_ENCODER, CLUSTER_MLP, CLUSTER_GOALS)   # tensor groups added after the first fixtures: each draws from its own generator


def init_weights(spec: ModelSpec, seed: int = 0) -> Dict[str, np.ndarray]:
    """Seeded initialiser (torch CPU generator; identical on every box with this image).  The drag-point encoder and
    the observation-update MLP draw from their own generators, so the other tensors keep the values older fixtures
    saw and are the same for every variant of a spec."""
    g_main = torch.Generator(device="cpu")
    g_main.manual_seed(1_000_003 * (seed + 1))
    g_late = []
    for i in range(len(_LATE)):
        g_late.append(torch.Generator(device="cpu"))
        g_late[i].manual_seed(7_000_003 * (seed + 1) + 11 + 1000 * i)
    late = lambda n: next((i for i, key in enumerate(_LATE) if key in n), -1)
    shapes = param_shapes(spec)
    w: Dict[str, np.ndarray] = {}
    for name in sorted(shapes, key=lambda n: (late(n), n)):
        g = g_late[late(name)] if late(name) >= 0 else g_main
        shp = shapes[name]
        pe_ln = PE_EMB in name and (".1." in name.split(PE_EMB)[1] or "to_out.0." in name)
        is_ln = len(shp) == 1 and name.endswith(".weight") and (
            "norm" in name or ".MLP.1." in name or _is_mlp_ln(name, shapes) or pe_ln)
        is_ln_bias = len(shp) == 1 and name.endswith(".bias") and (
            "norm" in name or ".MLP.1." in name or _is_mlp_ln(name[:-5] + ".weight", shapes) or pe_ln)
        if is_ln:
            t = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif is_ln_bias:
            t = 0.1 * torch.randn(shp, generator=g)
        elif "motion_anchors" in name or "tag_encoder" in name:
            t = torch.randn(shp, generator=g)
        elif name == CLUSTER_GOALS:   # goal clusters: points a few tens of metres out
            t = 20.0 * torch.randn(shp, generator=g)
        elif name.endswith(".freqs.weight"):   # (the reference initialises N(0, 0.02), weight_init.py:18-19; wider here so
            t = 0.05 * torch.randn(shp, generator=g)   # that a 100 m distance sweeps several periods)
        elif len(shp) == 2:
            bound = 1.0 / np.sqrt(shp[1])
            t = (torch.rand(shp, generator=g) * 2 - 1) * bound
        else:  # Linear bias: fan-in of the sibling weight
            fan_in = shapes[name[:-5] + ".weight"][1]
            bound = 1.0 / np.sqrt(fan_in)
            t = (torch.rand(shp, generator=g) * 2 - 1) * bound
        w[name] = t.to(torch.float32).numpy().copy()
    return w


def _is_mlp_ln(weight_name: str, shapes) -> bool:
    """A 1-d ``*.mlp.N.weight`` is a LayerNorm scale (Linear weights are 2-d)."""
    return ".mlp." in weight_name and weight_name in shapes and len(shapes[weight_name]) == 1


def from_state_dict(spec: ModelSpec, state_dict, k_goals=None) -> Dict[str, np.ndarray]:
    """Pick the rollout-path tensors out of a reference ``state_dict`` (torch tensors or arrays).  ``k_goals`` [K, 2]: the content
    of TRAJ.CLUSTER_PATH with TRAJ.PRED_MODE 'cluster' (the module keeps it as a plain attribute, not in the state_dict)."""
    w = {}
    for name, shp in param_shapes(spec).items():
        if name == CLUSTER_GOALS:
            if k_goals is None:
                raise KeyError("TRAJ.PRED_MODE 'cluster': pass k_goals = np.load(TRAJ.CLUSTER_PATH)")
            state_dict = dict(state_dict, **{name: np.asarray(k_goals, np.float32)})
        if name not in state_dict:
            raise KeyError(f"checkpoint lacks '{name}'")
        t = state_dict[name]
        a = t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)
        if tuple(a.shape) != tuple(shp):
            raise ValueError(f"'{name}': checkpoint shape {a.shape} != spec shape {shp}")
        w[name] = np.ascontiguousarray(a, dtype=np.float32)
    return w


def cluster_anchors(spec: ModelSpec, w: Dict[str, np.ndarray]) -> np.ndarray:
    """TRAJ.PRED_MODE 'cluster' (act_decoder.py:70-74, :103-105): anchor_emd = cluster_mlp(FourierEmbeddingFix(d / 2)(k_goals)),
    [K, d] -- input-independent, so it is a constant of the checkpoint.  fp32 like the module would compute it."""
    g = np.asarray(w[CLUSTER_GOALS], np.float32)                                   # [K, 2]
    n = spec.hidden // 2
    dim_t = np.float32(spec.fourier_temperature) ** (2 * (np.arange(n, dtype=np.float32) // 2) / np.float32(n))   # fourier_embedding.py:68-69
    dim_t = dim_t.astype(np.float32)
    pos = g * np.float32(2 * np.pi)
    cols = []
    for i in range(2):
        a = (pos[:, i, None] / dim_t).astype(np.float32)                           # [K, n]
        cols.append(np.stack([np.sin(a[:, 0::2]), np.cos(a[:, 1::2])], axis=-1).reshape(g.shape[0], n))
    pe = np.concatenate(cols, axis=-1).astype(np.float32)                          # [K, d]
    pa = f"policy.act_decoder.{CLUSTER_MLP}.mlp.0"
    return np.maximum(pe @ np.asarray(w[pa + ".weight"], np.float32).T + np.asarray(w[pa + ".bias"], np.float32), 0).astype(np.float32)


def engine_tensors(spec: ModelSpec, w: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """The tensors ps_create takes: the checkpoint's, with its input-independent sub-graphs folded -- 'cluster' anchors become
    the (type-independent) rows of the anchor table the head kernel indexes by (agent type, mode)."""
    out = w
    if spec.k_pred_mode == "cluster":
        out = {k: v for k, v in w.items() if k != CLUSTER_GOALS and f".{CLUSTER_MLP}." not in k}
        out["policy.act_decoder.motion_anchors.weight"] = np.ascontiguousarray(np.tile(cluster_anchors(spec, w), (spec.num_agent_types, 1)))
    nf = spec.pe_num_freq
    if pe_emb_prefixes(spec) and nf != ENGINE_PE_FREQ:
        # *.ATTN.PE_NUM_FREQ below the engine's 64 bands: the embedding is the 64-band one whose extra bands have frequency 0 and
        # zero weights -- their features (cos 0 = 1, sin 0 = 0) meet zero columns of the first Linear, exact zeros in every sum
        if not 1 <= nf < ENGINE_PE_FREQ:
            raise ValueError(f"PE_NUM_FREQ {nf}: the engine's learnable relative-PE takes up to {ENGINE_PE_FREQ} frequency bands")
        out = dict(out)
        for prefix in pe_emb_prefixes(spec):
            fr = np.zeros((3, ENGINE_PE_FREQ), np.float32)
            fr[:, :nf] = w[f"{prefix}.freqs.weight"]
            out[f"{prefix}.freqs.weight"] = fr
            for i in range(3):   # columns of mlps[i][0]: [cos (nf) | sin (nf) | x]  (fourier_embedding.py:45-49)
                W0 = np.asarray(w[f"{prefix}.mlps.{i}.0.weight"], np.float32)
                P = np.zeros((W0.shape[0], 2 * ENGINE_PE_FREQ + 1), np.float32)
                P[:, :nf] = W0[:, :nf]
                P[:, ENGINE_PE_FREQ:ENGINE_PE_FREQ + nf] = W0[:, nf:2 * nf]
                P[:, 2 * ENGINE_PE_FREQ] = W0[:, 2 * nf]
                out[f"{prefix}.mlps.{i}.0.weight"] = P
    return out


def to_reference_state_dict(spec: ModelSpec, w: Dict[str, np.ndarray]) -> Dict[str, torch.Tensor]:
    """Our dict + the aliased ``attn_prenorm_x_dst`` keys the reference state_dict carries."""
    sd = {k: torch.from_numpy(np.array(v)) for k, v in w.items() if k != CLUSTER_GOALS}
    for k in list(sd):
        if ".attn_prenorm_x_src." in k:
            alias = k.replace(".attn_prenorm_x_src.", ".attn_prenorm_x_dst.")
            if alias not in sd:
                sd[alias] = sd[k].clone()
    return sd


def n_params(spec: ModelSpec) -> int:
    return int(sum(int(np.prod(s)) for s in param_shapes(spec).values()))
