"""ORACLE -- test infrastructure, NOT product code.

A CPU restatement (plain torch fp32 tensor ops, exact brute-force neighbour search) of the
reference's closed-loop rollout path ``ProSim.forward(batch, 'val')``
(reference: prosim/models/traj_sam.py:59-71).  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import this module; the product path
(``prosim_amd``) never does and fails loudly without its HIP library.

Parity pinning (see DESIGN.md "Oracle"):
  * pinned against the reference's OWN Python (imported from /root/reference by
    tests/gen_golden.py, in the build container) for every primitive that needs no
    third-party native code: PointNet (K1), Fourier/rel-PE (K5), MLP/CG head (K9-K10),
    geometry + state update (K11-K12) -- fixtures ``tests/golden/ref_pure_*.npz``;
  * the reference's AttentionLayer / neighbour search call into torch_geometric and
    torch_cluster, which are NOT vendored, NOT pinned (install_local_env.sh:4-5) and NOT
    installed here.  Fixtures that cross that boundary (``tests/golden/ref_standins_*.npz``)
    were made by running the reference's Python with builder-written stand-ins for those
    two libraries (oracle/ref_harness.py), following torch_cluster's CUDA kernels'
    semantics: strict ``d^2 < r^2``, index-order truncation to ``max_num_neighbors``, kNN
    ties to the lower index.  **At that boundary parity is unpinned** -- the reference has
    no tests or golden vectors of its own.

Every function cites the reference file:line it restates.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from prosim_amd.spec import ModelSpec, USED_V_ACTION_TAGS, V_ACTION_TAGS
from prosim_amd.weights import mlp_layout

T = torch.Tensor


class W:
    """Name -> torch tensor view of a weight dict."""

    def __init__(self, w: Dict[str, np.ndarray], dtype=torch.float32):
        self.t = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dtype) for k, v in w.items()}
        self.dtype = dtype

    def __getitem__(self, k: str) -> T:
        return self.t[k]


# --------------------------------------------------------------------------- primitives

def layer_norm(x: T, w: T, b: T, eps: float = 1e-5) -> T:
    return torch.nn.functional.layer_norm(x, (x.shape[-1],), w, b, eps)


def mlp(Wt: W, prefix: str, dims: List[int], x: T, ret_before_act: bool, without_norm: bool) -> T:
    """reference MLP (models/layers/mlp.py:475-494)."""
    lay = mlp_layout(dims, ret_before_act, without_norm)
    n = len(lay)
    for i, (lin, ln) in enumerate(lay):
        x = x @ Wt[f"{prefix}.mlp.{lin}.weight"].T + Wt[f"{prefix}.mlp.{lin}.bias"]
        if i < n - 1:
            if ln >= 0:
                x = layer_norm(x, Wt[f"{prefix}.mlp.{ln}.weight"], Wt[f"{prefix}.mlp.{ln}.bias"])
            x = torch.relu(x)
    if not ret_before_act:
        x = torch.relu(x)
    return x


def wrap_angle(a: T) -> T:
    """models/utils/geometry.py:13-17 (python-style %, result in [-pi, pi))."""
    return -math.pi + (a + math.pi) % (math.pi - (-math.pi))


def batch_rotate_2d(xy: T, theta: T) -> T:
    """models/utils/geometry.py:19-22."""
    x1 = xy[..., 0] * torch.cos(theta) - xy[..., 1] * torch.sin(theta)
    y1 = xy[..., 1] * torch.cos(theta) + xy[..., 0] * torch.sin(theta)
    return torch.stack([x1, y1], dim=-1)


def fourier_div(num_pos_feats: float, temperature: float = 10000.0, dtype=torch.float32) -> T:
    """The ``dim_t`` table of FourierEmbeddingFix (models/layers/fourier_embedding.py:68-69),
    built with the same torch ops so the divisors are bit-identical to the reference's."""
    dim_t = torch.arange(num_pos_feats, dtype=dtype)
    return temperature ** (2 * (dim_t // 2) / num_pos_feats)


def fourier_fix(x: T, num_pos_feats: float) -> T:
    """FourierEmbeddingFix.forward (fourier_embedding.py:63-78): [.., C] -> [.., C*F],
    per input channel interleaved [sin, cos, sin, cos, ...] over F = num_pos_feats slots."""
    pos = x * (2 * math.pi)
    dim_t = fourier_div(num_pos_feats, dtype=x.dtype)
    outs = []
    for i in range(x.shape[-1]):
        p = pos[..., i, None] / dim_t
        p = torch.stack((p[..., 0::2].sin(), p[..., 1::2].cos()), dim=-1).flatten(-2)
        outs.append(p)
    return torch.cat(outs, dim=-1)


def pointnet(Wt: W, prefix: str, in_dim: int, hidden: int, n_pre: int, n_mlp: int,
             polylines: T, mask: T) -> T:
    """PointNetPolylineEncoder.forward (scene_encoder/pointnet_encoder.py:24-62).
    polylines [B, M, P, C], mask [B, M, P] bool -> [B, M, hidden]; masked points pool as 0."""
    B, M, P, C = polylines.shape
    pre = mlp(Wt, f"{prefix}.pre_mlps", [in_dim] + [hidden] * n_pre, polylines[mask], False, False)
    feat = polylines.new_zeros(B, M, P, hidden)
    feat[mask] = pre
    pooled = feat.max(dim=2)[0]
    feat = torch.cat((feat, pooled[:, :, None, :].repeat(1, 1, P, 1)), dim=-1)
    mid = mlp(Wt, f"{prefix}.mlps", [hidden * 2] + [hidden] * (n_mlp - n_pre), feat[mask], False, False)
    buf = feat.new_zeros(B, M, P, hidden)
    buf[mask] = mid
    buf = buf.max(dim=2)[0]
    valid = mask.sum(dim=-1) > 0
    out_valid = mlp(Wt, f"{prefix}.out_mlps", [hidden] * 3, buf[valid], True, True)
    out = buf.new_zeros(B, M, hidden)
    out[valid] = out_valid
    return out


# --------------------------------------------------------------------------- neighbour search
# torch_cluster is a third-party dependency that is absent from /root/reference
# (install_local_env.sh:4, unpinned wheel index torch-2.4.0+cpu).  Semantics restated here
# are those of its CUDA kernels (radius_cuda.cu / knn_cuda.cu): for every query y scan the x
# of the same example in index order; radius keeps the first ``max_num_neighbors`` with
# squared distance strictly < r^2; knn keeps the k smallest by (distance, index).
# Distances are d2 = dx*dx + dy*dy in fp32 with each op rounded (no FMA contraction) so the
# HIP engine can reproduce the neighbour sets bit-for-bit.

def _d2(px: T, py: T) -> T:
    dx = px[None, :, 0] - py[:, None, 0]
    dy = px[None, :, 1] - py[:, None, 1]
    return dx * dx + dy * dy  # [Ny, Nx]


def radius_edges(pos_x: T, batch_x: T, pos_y: T, batch_y: T, r: float, max_num: int,
                 drop_self: bool = False) -> Tuple[T, T]:
    """torch_cluster.radius(x, y, r, batch_x, batch_y, max_num_neighbors) -> (y_idx, x_idx),
    sorted by y then x.  ``drop_self``: radius_graph(loop=False) = radius with cap+1, then
    remove x==y pairs (torch_cluster/radius.py radius_graph)."""
    if pos_y.shape[0] == 0 or pos_x.shape[0] == 0:
        z = torch.zeros(0, dtype=torch.long)
        return z, z
    d2 = _d2(pos_x.float(), pos_y.float())
    rr = torch.tensor(r, dtype=torch.float32) * torch.tensor(r, dtype=torch.float32)
    ok = (d2 < rr) & (batch_y[:, None] == batch_x[None, :])
    cap = max_num + 1 if drop_self else max_num
    rank = torch.cumsum(ok.to(torch.int64), dim=1)
    ok = ok & (rank <= cap)
    yi, xi = ok.nonzero(as_tuple=True)
    if drop_self:
        keep = yi != xi
        yi, xi = yi[keep], xi[keep]
    return yi, xi


def knn_edges(pos_x: T, batch_x: T, pos_y: T, batch_y: T, k: int, drop_self: bool = False) -> Tuple[T, T]:
    """torch_cluster.knn(x, y, k, batch_x, batch_y) -> (y_idx, x_idx): the k nearest x of
    each y within its example, ties to the lower x index; fewer than k if the example is small.
    ``drop_self``: knn_graph(loop=False) = knn with k + 1, then remove the x == y pairs (torch_cluster/knn.py knn_graph)."""
    if pos_y.shape[0] == 0 or pos_x.shape[0] == 0:
        z = torch.zeros(0, dtype=torch.long)
        return z, z
    d2 = _d2(pos_x.float(), pos_y.float())
    same = batch_y[:, None] == batch_x[None, :]
    d2 = torch.where(same, d2, torch.full_like(d2, float("inf")))
    order = torch.sort(d2, dim=1, stable=True)[1]  # stable: ties keep index order
    kk = min(k + 1 if drop_self else k, pos_x.shape[0])
    xi = order[:, :kk]
    yi = torch.arange(pos_y.shape[0])[:, None].expand(-1, kk)
    valid = torch.gather(same, 1, xi)
    yi, xi = yi[valid], xi[valid]
    if drop_self:
        keep = yi != xi
        yi, xi = yi[keep], xi[keep]
    return yi, xi


# --------------------------------------------------------------------------- rel-PE + attention

def rel_pe_input(src: T, dst: T, ori_dst: T, pos_dst: T, ori_src: T, pos_src: T) -> T:
    """The 4 scalars per edge (act_decoder.py:203-217; twins attn_fusion.py:44-54,
    sym_coord.py:44-55, condition_attns.py:98-108).  ori_* are [N, 1]."""
    ori_vec_dst = torch.stack([ori_dst.cos().squeeze(-1), ori_dst.sin().squeeze(-1)], dim=-1)
    rel_pos = pos_src[src] - pos_dst[dst]
    rel_ori = wrap_angle(ori_src[src] - ori_dst[dst]).squeeze(-1)
    ctr = ori_vec_dst[dst]
    # angle_between_2d_vectors (geometry.py:6-11)
    ang = torch.atan2(ctr[..., 0] * rel_pos[..., 1] - ctr[..., 1] * rel_pos[..., 0],
                      (ctr[..., :2] * rel_pos[..., :2]).sum(dim=-1))
    return torch.stack([torch.norm(rel_pos, dim=-1), rel_ori, ang, ang], dim=-1)


def fourier_learn(Wt: W, prefix: str, x: T, eps: float = 1e-5) -> T:
    """FourierEmbedding.forward with continuous inputs only (models/layers/fourier_embedding.py:37-54): x [E, 3] ->
    [E, hidden].  Per input i: [cos(x_i f_ik 2 pi), sin(...), x_i] -> Linear, LayerNorm, ReLU, Linear; summed over the
    inputs; then LayerNorm, ReLU, Linear."""
    f = Wt[f"{prefix}.freqs.weight"]
    v = x.unsqueeze(-1) * f * 2 * math.pi
    v = torch.cat([v.cos(), v.sin(), x.unsqueeze(-1)], dim=-1)
    embs = []
    for i in range(x.shape[-1]):
        q = f"{prefix}.mlps.{i}"
        h = v[:, i] @ Wt[f"{q}.0.weight"].T + Wt[f"{q}.0.bias"]
        h = torch.relu(layer_norm(h, Wt[f"{q}.1.weight"], Wt[f"{q}.1.bias"], eps))
        embs.append(h @ Wt[f"{q}.3.weight"].T + Wt[f"{q}.3.bias"])
    y = torch.stack(embs).sum(dim=0)
    y = torch.relu(layer_norm(y, Wt[f"{prefix}.to_out.0.weight"], Wt[f"{prefix}.to_out.0.bias"], eps))
    return y @ Wt[f"{prefix}.to_out.2.weight"].T + Wt[f"{prefix}.to_out.2.bias"]


def rel_pe(spec: ModelSpec, src, dst, ori_dst, pos_dst, ori_src, pos_src, Wt: W = None, emb: str = None) -> T:
    """The relative-PE rows of one edge set.  ``emb``: state-dict prefix of the set's learnable FourierEmbedding
    (LEARNABLE_PE: three inputs, attn_fusion.py:50-51) or None for the fixed one (four inputs: the angle twice)."""
    x = rel_pe_input(src, dst, ori_dst, pos_dst, ori_src, pos_src)
    if emb is not None:
        return fourier_learn(Wt, emb, x[..., :3], spec.ln_eps)
    return fourier_fix(x, spec.hidden / 4)


def _emb(spec: ModelSpec, part: str, name: str):
    """Prefix of a learnable rel-PE embedding, or None when that part of the model uses the fixed embedding."""
    on = {"scene_encoder": spec.enc_learnable_pe, "decoder": spec.dec_learnable_pe, "policy.act_decoder": spec.pol_learnable_pe}[part]
    return f"{part}.{name}_rel_pe_emb" if on else None


def attention_layer(Wt: W, p: str, spec: ModelSpec, x_src: T, x_dst: T, r: T, src: T, dst: T,
                    bipartite: bool) -> T:
    """AttentionLayer.forward (models/layers/attention_layer.py:56-121) with
    torch_geometric's propagate(aggr='add') and utils.softmax restated
    (max-shift, exp, / (sum + 1e-16), per destination and head)."""
    H, Dh = spec.heads, spec.head_dim
    nd = x_dst.shape[0]
    xs = layer_norm(x_src, Wt[f"{p}.attn_prenorm_x_src.weight"], Wt[f"{p}.attn_prenorm_x_src.bias"])
    dn = "attn_prenorm_x_dst" if bipartite else "attn_prenorm_x_src"
    xd = layer_norm(x_dst, Wt[f"{p}.{dn}.weight"], Wt[f"{p}.{dn}.bias"])
    rh = layer_norm(r, Wt[f"{p}.attn_prenorm_r.weight"], Wt[f"{p}.attn_prenorm_r.bias"])
    q = (xd @ Wt[f"{p}.to_q.weight"].T + Wt[f"{p}.to_q.bias"]).view(-1, H, Dh)
    k = (xs @ Wt[f"{p}.to_k.weight"].T).view(-1, H, Dh)
    v = (xs @ Wt[f"{p}.to_v.weight"].T + Wt[f"{p}.to_v.bias"]).view(-1, H, Dh)
    k_j = k[src] + (rh @ Wt[f"{p}.to_k_r.weight"].T).view(-1, H, Dh)
    v_j = v[src] + (rh @ Wt[f"{p}.to_v_r.weight"].T + Wt[f"{p}.to_v_r.bias"]).view(-1, H, Dh)
    sim = (q[dst] * k_j).sum(dim=-1) * (Dh ** -0.5)  # [E, H]
    smax = torch.full((nd, H), float("-inf"), dtype=sim.dtype)
    smax = smax.scatter_reduce(0, dst[:, None].expand(-1, H), sim, reduce="amax", include_self=True)
    ex = (sim - smax[dst]).exp()
    ssum = torch.zeros(nd, H, dtype=sim.dtype).index_add_(0, dst, ex) + 1e-16
    attn = ex / ssum[dst]
    agg = torch.zeros(nd, H, Dh, dtype=sim.dtype).index_add_(0, dst, v_j * attn.unsqueeze(-1))
    agg = agg.view(nd, H * Dh)
    g = torch.sigmoid(torch.cat([agg, xd], dim=-1) @ Wt[f"{p}.to_g.weight"].T + Wt[f"{p}.to_g.bias"])
    upd = agg + g * ((xd @ Wt[f"{p}.to_s.weight"].T + Wt[f"{p}.to_s.bias"]) - agg)
    x = x_dst + layer_norm(upd @ Wt[f"{p}.to_out.weight"].T + Wt[f"{p}.to_out.bias"],
                           Wt[f"{p}.attn_postnorm.weight"], Wt[f"{p}.attn_postnorm.bias"])
    h = layer_norm(x, Wt[f"{p}.ff_prenorm.weight"], Wt[f"{p}.ff_prenorm.bias"])
    h = torch.relu(h @ Wt[f"{p}.ff_mlp.0.weight"].T + Wt[f"{p}.ff_mlp.0.bias"])
    h = h @ Wt[f"{p}.ff_mlp.3.weight"].T + Wt[f"{p}.ff_mlp.3.bias"]
    x = x + layer_norm(h, Wt[f"{p}.ff_postnorm.weight"], Wt[f"{p}.ff_postnorm.bias"])
    return x


# --------------------------------------------------------------------------- scene encoder

def _flat_batch_idx(mask: T) -> T:
    B, n = mask.shape
    return torch.arange(B).unsqueeze(1).repeat(1, n).view(-1)[mask.view(-1)]


def encode_obs(Wt: W, spec: ModelSpec, obs_input: T, obs_mask: T) -> Tuple[T, T]:
    """POINTNET_OBV_ENCODER.forward (scene_encoder/obs_encoder.py:82-87)."""
    pm = obs_mask.all(dim=-1)
    emb = pointnet(Wt, "scene_encoder.obs_encoder", spec.obs_dim, spec.hidden, spec.obs_pre_layers,
                   spec.obs_mlp_layers, obs_input, pm)
    return emb, pm.any(dim=-1)


def encode_map(Wt: W, spec: ModelSpec, map_input: T, map_mask: T) -> Tuple[T, T]:
    """POINTNET_MAP_ENCODER.forward (scene_encoder/map_encoder.py:81-88)."""
    emb = pointnet(Wt, "scene_encoder.map_encoder", spec.map_dim, spec.hidden, spec.map_pre_layers,
                   spec.map_mlp_layers, map_input, map_mask)
    return emb, map_mask.any(dim=-1)


def scene_fusion(Wt: W, spec: ModelSpec, map_emb, map_mask, map_pos, map_head,
                 obs_emb, obs_mask, obs_pos, obs_head) -> Dict:
    """AttentionSceneEncoderRelPE._scene_fusion (scene_encoder/attn_fusion.py:78-134)."""
    D = spec.hidden
    map_b = _flat_batch_idx(map_mask)
    obs_b = _flat_batch_idx(obs_mask)
    scene_b = torch.cat([map_b, obs_b])
    scene_type = torch.cat([torch.zeros_like(map_b), torch.ones_like(obs_b)])
    x = torch.cat([map_emb.reshape(-1, D)[map_mask.view(-1)], obs_emb.reshape(-1, D)[obs_mask.view(-1)]])
    m_pos = map_pos.reshape(-1, 2)[map_mask.view(-1)]
    o_pos = obs_pos.reshape(-1, 2)[obs_mask.view(-1)]
    s_pos = torch.cat([m_pos, o_pos])
    m_ori = map_head.reshape(-1, 1)[map_mask.view(-1)]
    o_ori = obs_head.reshape(-1, 1)[obs_mask.view(-1)]
    s_ori = torch.cat([m_ori, o_ori])
    # knn_graph(loop=True, flow=source_to_target): edge (x-neighbour j -> query i)
    a_dst, a_src = knn_edges(o_pos, obs_b, o_pos, obs_b, spec.agent_knn)
    s_dst, s_src = knn_edges(s_pos, scene_b, s_pos, scene_b, spec.scene_knn)
    a_pe = rel_pe(spec, a_src, a_dst, o_ori, o_pos, o_ori, o_pos, Wt, _emb(spec, "scene_encoder", "a2a"))
    s_pe = rel_pe(spec, s_src, s_dst, s_ori, s_pos, s_ori, s_pos, Wt, _emb(spec, "scene_encoder", "s2s"))
    a_mask = scene_type == 1
    for i in range(spec.scene_layers):
        xa = x[a_mask]
        x = x.clone()
        x[a_mask] = attention_layer(Wt, f"scene_encoder.a2a_attn_layers.{i}", spec, xa, xa, a_pe, a_src, a_dst, False)
        x = attention_layer(Wt, f"scene_encoder.s2s_attn_layers.{i}", spec, x, x, s_pe, s_src, s_dst, False)
    return dict(scene_tokens=x, scene_pos=s_pos, scene_ori=s_ori, scene_type=scene_type,
                scene_batch_idx=scene_b, obs_mask=obs_mask, map_mask=map_mask,
                edges=dict(a2a=int(a_src.numel()), s2s=int(s_src.numel())))


def replace_obs(scene: Dict, obs_emb: T, obs_mask: T, obs_pos: T, obs_head: T) -> Dict:
    """_replace_old_obs (attn_fusion.py:205-236): keep map tokens, swap agent tokens/poses."""
    D = obs_emb.shape[-1]
    mt = scene["scene_type"] == 0
    map_b = scene["scene_batch_idx"][mt]
    obs_b = _flat_batch_idx(obs_mask)
    out = dict(scene)
    out["scene_batch_idx"] = torch.cat([map_b, obs_b])
    out["scene_type"] = torch.cat([torch.zeros_like(map_b), torch.ones_like(obs_b)])
    out["scene_tokens"] = torch.cat([scene["scene_tokens"][mt], obs_emb.reshape(-1, D)[obs_mask.view(-1)]])
    out["scene_pos"] = torch.cat([scene["scene_pos"][mt], obs_pos.reshape(-1, 2)[obs_mask.view(-1)]])
    out["scene_ori"] = torch.cat([scene["scene_ori"][mt], obs_head.reshape(-1, 1)[obs_mask.view(-1)]])
    out["obs_mask"] = obs_mask
    return out


def fuse_obs_mlp(Wt: W, spec: ModelSpec, scene: Dict, new_emb: T, new_mask: T) -> T:
    """_autoregressive_obs_fusion (attn_fusion.py:175-203) for an unchanged agent list: every agent's new token is
    obs_update_mlp(cat(previous token, re-encoded observation)); an agent that had no token at the previous replan
    contributes zeros (:193-195).  Returns new_emb [B, N, D] with the fused rows."""
    B, N, D = new_emb.shape
    old = torch.zeros(B, N, D, dtype=new_emb.dtype)
    old[scene["obs_mask"]] = scene["scene_tokens"][scene["scene_type"] == 1]
    fused = mlp(Wt, "scene_encoder.obs_update_mlp", [2 * D, D, D], torch.cat([old, new_emb], dim=-1), True, False)
    return fused


def update_scene_attn(Wt: W, spec: ModelSpec, scene: Dict) -> Dict:
    """_update_scene_emb_attn (attn_fusion.py:136-173): after the observation update the agents re-attend to each
    other (radius_graph, no self loops) and to the map (radius), through the scene encoder's own layers."""
    mt, ot = scene["scene_type"] == 0, scene["scene_type"] == 1
    m_pos, m_ori, o_pos, o_ori = scene["scene_pos"][mt], scene["scene_ori"][mt], scene["scene_pos"][ot], scene["scene_ori"][ot]
    m_b, o_b = scene["scene_batch_idx"][mt], scene["scene_batch_idx"][ot]
    a_dst, a_src = radius_edges(o_pos, o_b, o_pos, o_b, spec.enc_agent_radius, spec.scene_knn, drop_self=True)
    m_dst, m_src = radius_edges(m_pos, m_b, o_pos, o_b, spec.enc_scene_radius, spec.scene_knn)
    # (:158-159, :44-76: the agent-agent rows from a2a_rel_pe_emb, the map -> agent rows from s2s_rel_pe_emb when LEARNABLE_PE)
    a_pe = rel_pe(spec, a_src, a_dst, o_ori, o_pos, o_ori, o_pos, Wt, _emb(spec, "scene_encoder", "a2a"))
    m_pe = rel_pe(spec, m_src, m_dst, o_ori, o_pos, m_ori, m_pos, Wt, _emb(spec, "scene_encoder", "s2s"))
    x_a, x_m = scene["scene_tokens"][ot], scene["scene_tokens"][mt]
    for i in range(spec.scene_layers):
        x_a = attention_layer(Wt, f"scene_encoder.a2a_attn_layers.{i}", spec, x_a, x_a, a_pe, a_src, a_dst, False)
        # a non-bipartite layer called with (x_src, x_dst): one LayerNorm serves both sides (attention_layer.py:48-49)
        x_a = attention_layer(Wt, f"scene_encoder.s2s_attn_layers.{i}", spec, x_m, x_a, m_pe, m_src, m_dst, False)
    out = dict(scene)
    tok = scene["scene_tokens"].clone()
    tok[ot] = x_a
    out["scene_tokens"] = tok
    return out


# --------------------------------------------------------------------------- generator (decoder + conditions)

def prompt_encode(Wt: W, spec: ModelSpec, prompt: T) -> T:
    """PromptEncoder._prompt_encode (prompt_encoder/base.py:36-46)."""
    return mlp(Wt, "prompt_encoder.motion_pred.state_encoder", [spec.prompt_dim, spec.hidden, spec.hidden],
               prompt, True, False)


def decoder_fusion(Wt: W, spec: ModelSpec, scene: Dict, prompt_emd: T, prompt_mask: T,
                   prompt_pos: T, prompt_head: T) -> Tuple[T, Dict]:
    """SymCoordDecoder._fusion (decoder/sym_coord.py:63-110)."""
    B, N = prompt_mask.shape
    D = spec.hidden
    pb = _flat_batch_idx(prompt_mask)
    xp = prompt_emd.reshape(-1, D)[prompt_mask.view(-1)]
    ppos = prompt_pos.reshape(-1, 2)[prompt_mask.view(-1)]
    pori = prompt_head.reshape(-1, 1)[prompt_mask.view(-1)]
    # radius_graph(loop=False): (x=source j, y=target i)
    knn = spec.rel_pos_edge_func == "knn"   # MODEL.REL_POS_EDGE_FUNC (sym_coord.py:85-96)
    if knn:
        pp_dst, pp_src = knn_edges(ppos, pb, ppos, pb, spec.dec_max_neigh, drop_self=True)
    else:
        pp_dst, pp_src = radius_edges(ppos, pb, ppos, pb, spec.dec_prompt_radius, spec.dec_max_neigh, drop_self=True)
    pp_pe = rel_pe(spec, pp_src, pp_dst, pori, ppos, pori, ppos, Wt, _emb(spec, "decoder", "p2p"))
    if knn:
        sp_dst, sp_src = knn_edges(scene["scene_pos"], scene["scene_batch_idx"], ppos, pb, spec.dec_max_neigh)
    else:
        sp_dst, sp_src = radius_edges(scene["scene_pos"], scene["scene_batch_idx"], ppos, pb,
                                      spec.dec_scene_radius, spec.dec_max_neigh)
    sp_pe = rel_pe(spec, sp_src, sp_dst, pori, ppos, scene["scene_ori"], scene["scene_pos"], Wt, _emb(spec, "decoder", "s2p"))
    xs = scene["scene_tokens"]
    for i in range(spec.dec_layers):
        xp = attention_layer(Wt, f"decoder.p2p_attn_layers.{i}", spec, xp, xp, pp_pe, pp_src, pp_dst, False)
        xp = attention_layer(Wt, f"decoder.s2p_attn_layers.{i}", spec, xs, xp, sp_pe, sp_src, sp_dst, True)
    emd = torch.zeros(B, N, D, dtype=xp.dtype)
    emd[prompt_mask] = xp
    return emd, dict(p2p=int(pp_src.numel()), s2p=int(sp_src.numel()))


def goal_heads(Wt: W, spec: ModelSpec, emd: T, prompt_mask: T) -> Tuple[T, T]:
    """Decoder._goal_pred (decoder/base.py:22-58): goal_prob [B, N, K] and goal_point [B, N, K, 2] from the decoder's
    embedding of every prompt (zeros elsewhere); both heads are MLP([D, D/2, .]) with LayerNorm + ReLU between."""
    B, N, d = emd.shape
    K = spec.goal_pred_k
    prob = torch.zeros(B, N, K, dtype=emd.dtype)
    point = torch.zeros(B, N, K, 2, dtype=emd.dtype)
    x = emd[prompt_mask]
    prob[prompt_mask] = mlp(Wt, "decoder.goal_prob_head", [d, d // 2, K], x, True, False)
    point[prompt_mask] = mlp(Wt, "decoder.goal_point_head", [d, d // 2, 2 * K], x, True, False).view(-1, K, 2)
    return prob, point


def condition_transform(Wt: W, spec: ModelSpec, cond: Optional[Dict], emd: T, prompt_mask: T,
                        prompt_pos: T, prompt_head: T) -> T:
    """ConditionTransformer.forward at 'policy_decoder' (condition_transformer/base.py:38-60)
    with GoalConditionEncoder (condition_encoders.py:21-51), V_ActionTagEncoder (:76-141), DragPointEncoder
    (:152-191), V2V_MotionTagEncoder (:148-150) and GNNConditionAttn (condition_attns.py:114-228): unary conditions
    attach to one prompt agent (self-loop edges), binary ones to an ordered pair (edges s -> t and t -> s); mean pooling
    over the condition keys present on an edge.  ``cond`` = {'goal': {'input' [B,C,3], 'mask' [B,C],
    'prompt_idx' [B,C,1]}, 'v_action_tag': {...}}; None / {} -> identity."""
    if not cond:
        return emd
    B, N = prompt_mask.shape
    D = spec.hidden
    ct = "condition_transformers.policy_decoder"
    entries = []  # per condition key (insertion order of the reference): (emd [B,C,D], mask [B,C], pidx [B,C])
    pair_entries = []  # binary keys: (source emd [B,C,D], target emd [B,C,D], mask [B,C], pidx [B,C,2])
    if "goal" in cond and cond["goal"]["input"].shape[1] > 0:
        ci = cond["goal"]
        e = mlp(Wt, f"{ct}.condition_encoders.goal.goal_encoder", [2, D, D], ci["input"][..., :2], True, True)
        e = e + fourier_fix(ci["input"][..., 2:], D)
        entries.append((e, ci["mask"], ci["prompt_idx"][..., 0]))
    if "v_action_tag" in cond and cond["v_action_tag"]["input"].shape[1] > 0:
        ci = cond["v_action_tag"]
        for tag in USED_V_ACTION_TAGS:
            sel = ci["input"][..., 0] == V_ACTION_TAGS.index(tag)
            if sel.sum() == 0:
                continue
            e = Wt[f"{ct}.condition_encoders.v_action_tag.tag_encoder.{tag}"][None, None, :] + \
                fourier_fix(ci["input"][..., 1:3], D // 2)
            entries.append((e, ci["mask"] & sel, ci["prompt_idx"][..., 0]))
    if "drag_point" in cond and cond["drag_point"]["input"].shape[1] > 0:
        # DragPointEncoder.forward (condition_encoders.py:164-191): PointNet over the points that are not NaN
        ci = cond["drag_point"]
        pts = ci["input"].to(emd.dtype)
        pmask = ~(pts.isnan().any(-1))
        e = pointnet(Wt, f"{ct}.condition_encoders.drag_point.pointnet_encoder", 2, D, spec.drag_pre_layers,
                     spec.drag_mlp_layers, torch.nan_to_num(pts), pmask)
        entries.append((e, ci["mask"], ci["prompt_idx"][..., 0]))
    if "v2v_tag" in cond and cond["v2v_tag"]["input"].shape[1] > 0:
        # V2V_MotionTagEncoder (condition_encoders.py:148-150 over :76-141): a [2 D] parameter per tag, source half |
        # target half, the temporal embedding added to both halves
        from prosim_amd.spec import V2V_TAGS
        ci = cond["v2v_tag"]
        for tag in spec.used_v2v_tags:
            sel = ci["input"][..., 0] == V2V_TAGS.index(tag)
            if sel.sum() == 0:
                continue
            par = Wt[f"{ct}.condition_encoders.v2v_tag.tag_encoder.{tag}"]
            te = fourier_fix(ci["input"][..., 1:3], D // 2)
            pair_entries.append((par[None, None, :D] + te, par[None, None, D:] + te, ci["mask"] & sel, ci["prompt_idx"]))
    if not entries and not pair_entries:
        return emd
    # _construct_cond_edge_matrix + _pool_edges('mean') (condition_attns.py:114-188): edge_attr[b, i, j] = mean over the
    # condition keys that put an entry on (i -> j); unary keys on (s, s), binary keys s_emd on (s, t) and t_emd on (t, s)
    attr = torch.zeros(B, N, N, D, dtype=emd.dtype)
    cnt = torch.zeros(B, N, N, dtype=emd.dtype)
    # The reference fills one [B, N, N, D] plane per condition key BY ASSIGNMENT (:155-166): of several entries of one key on
    # one edge the last one (in entry order) survives, and the edge counts once in the mean pool.
    def last_wins(bi, ci_, a, b):
        keep = torch.ones(bi.numel(), dtype=torch.bool)
        seen = set()
        for k in range(bi.numel() - 1, -1, -1):
            key = (int(bi[k]), int(a[k]), int(b[k]))
            if key in seen:
                keep[k] = False
            seen.add(key)
        return keep

    for e, m, pidx in entries:
        bi, ci_ = m.nonzero(as_tuple=True)
        ni = pidx[bi, ci_]
        kp = last_wins(bi, ci_, ni, ni)
        bi, ci_, ni = bi[kp], ci_[kp], ni[kp]
        attr.index_put_((bi, ni, ni), e[bi, ci_], accumulate=True)
        cnt.index_put_((bi, ni, ni), torch.ones(bi.numel(), dtype=emd.dtype), accumulate=True)
    for es, et, m, pidx in pair_entries:
        # one [B, N, N, D] plane per binary key, filled by TWO assignment passes (:155-162): every entry's source half on (s, t) in entry
        # order, then every entry's target half on (t, s) -- so a target half overwrites a source half that a REVERSED entry of the same
        # key put on that edge, and the edge still counts once
        plane = {}
        bi, ci_ = m.nonzero(as_tuple=True)
        for k in range(bi.numel()):
            b_, c_ = int(bi[k]), int(ci_[k])
            plane[(b_, int(pidx[b_, c_, 0]), int(pidx[b_, c_, 1]))] = es[b_, c_]
        for k in range(bi.numel()):
            b_, c_ = int(bi[k]), int(ci_[k])
            plane[(b_, int(pidx[b_, c_, 1]), int(pidx[b_, c_, 0]))] = et[b_, c_]
        for (b_, i_, j_), v in plane.items():
            attr[b_, i_, j_] += v
            cnt[b_, i_, j_] += 1
    attr = attr / cnt.clamp(min=1)[..., None]
    edge_ok = (cnt > 0) & prompt_mask[:, :, None] & prompt_mask[:, None, :]
    node_index = torch.full((B, N), -1, dtype=torch.long)
    node_index[prompt_mask] = torch.arange(int(prompt_mask.sum()))
    eb, ei, ej = edge_ok.nonzero(as_tuple=True)          # row-major (b, i, j): edge i -> j  (attn_utils.py:37-48)
    e_src, e_dst = node_index[eb, ei], node_index[eb, ej]
    ppos = prompt_pos[prompt_mask]
    pori = prompt_head.reshape(B, N, 1)[prompt_mask]
    r = attr[eb, ei, ej] + rel_pe(spec, e_src, e_dst, pori, ppos, pori, ppos)
    xp = emd[prompt_mask]
    for i in range(spec.cond_layers):
        xp = attention_layer(Wt, f"{ct}.condition_attn.attn_layers.{i}", spec, xp, xp, r, e_src, e_dst, False)
    out = emd.clone()
    out[prompt_mask] = out[prompt_mask] + xp
    return out


# --------------------------------------------------------------------------- policy

def policy_forward(Wt: W, spec: ModelSpec, scene: Dict, policy_emd: T, agent_type: T, policy_b: T,
                   pos: T, head: T, noise: Optional[T] = None) -> Dict:
    """Policy_RelPE_Temporal.forward -> PolicyNoRNN.forward -> AttnRelPE.attn_fuse +
    ActDecoder._compute_traj (policy/base.py:19, temporal_ar.py:75-92, act_decoder.py:239-283,
    :78-140).  policy_emd [A, D], agent_type [A] (1..3), policy_b [A] scene index, pos [A,2],
    head [A,1].  The ragged->padded->ragged shuffles (K13) are the identity on the token set."""
    pa = "policy.act_decoder"
    st = scene["scene_type"]
    x_a, a_pos, a_ori, a_b = (scene["scene_tokens"][st == 1], scene["scene_pos"][st == 1],
                              scene["scene_ori"][st == 1], scene["scene_batch_idx"][st == 1])
    x_m, m_pos, m_ori, m_b = (scene["scene_tokens"][st == 0], scene["scene_pos"][st == 0],
                              scene["scene_ori"][st == 0], scene["scene_batch_idx"][st == 0])
    if spec.rel_pos_edge_func == "knn":   # MODEL.REL_POS_EDGE_FUNC (act_decoder.py:249-261)
        ap_dst, ap_src = knn_edges(a_pos, a_b, pos, policy_b, spec.pol_max_neigh)
        mp_dst, mp_src = knn_edges(m_pos, m_b, pos, policy_b, spec.pol_max_neigh)
    else:
        ap_dst, ap_src = radius_edges(a_pos, a_b, pos, policy_b, spec.pol_agent_radius, spec.pol_max_neigh)
        mp_dst, mp_src = radius_edges(m_pos, m_b, pos, policy_b, spec.pol_map_radius, spec.pol_max_neigh)
    ap_pe = rel_pe(spec, ap_src, ap_dst, head, pos, a_ori, a_pos, Wt, _emb(spec, "policy.act_decoder", "a2p"))
    mp_pe = rel_pe(spec, mp_src, mp_dst, head, pos, m_ori, m_pos, Wt, _emb(spec, "policy.act_decoder", "m2p"))
    xp = policy_emd
    for i in range(spec.pol_layers):
        xp = attention_layer(Wt, f"{pa}.a2p_attn_layers.{i}", spec, x_a, xp, ap_pe, ap_src, ap_dst, True)
        xp = attention_layer(Wt, f"{pa}.m2p_attn_layers.{i}", spec, x_m, xp, mp_pe, mp_src, mp_dst, True)
    out = compute_traj(Wt, spec, xp, agent_type, policy_emd, noise)
    out["fused"] = xp
    out["edges"] = dict(a2p=int(ap_src.numel()), m2p=int(mp_src.numel()))
    return out


def cg_stacked(Wt: W, prefix: str, inp: T, context: T) -> Tuple[T, T]:
    """CG_stacked(3).forward with an all-true mask (models/layers/mlp.py:207-241).
    inp [A, K, D], context [A, D]."""
    def block(i, x, c):
        y = x @ Wt[f"{prefix}.CGs.{i}.MLP.0.weight"].T + Wt[f"{prefix}.CGs.{i}.MLP.0.bias"]
        y = torch.relu(layer_norm(y, Wt[f"{prefix}.CGs.{i}.MLP.1.weight"], Wt[f"{prefix}.CGs.{i}.MLP.1.bias"]))
        y = y * c.unsqueeze(1)
        return y, y.max(dim=1)[0]
    inp_, ctx_ = block(0, inp, context)
    for i in range(1, 3):
        a, c = block(i, inp_, ctx_)
        inp_ = (inp_ * i + a) / (i + 1)
        ctx_ = (ctx_ * i + c) / (i + 1)
    return inp_, ctx_


def compute_traj(Wt: W, spec: ModelSpec, pred_feat: T, agent_type: T, policy_emd: T, noise: Optional[T] = None) -> Dict:
    """ActDecoder._compute_traj, TRAJ.PRED_MODE anchor / cluster / mlp (act_decoder.py:78-140).  ``noise`` [A, K, steps, 2]: RANDOM_NOISE_STD's draw,
    already scaled (:113-115: added to the xy steps before the cumulative sum)."""
    pa = "policy.act_decoder"
    A = pred_feat.shape[0]
    K = spec.motion_k
    d = spec.hidden
    if spec.k_pred_mode == "mlp":       # (:90-91) all K modes from one head, no anchors
        motion = mlp(Wt, f"{pa}.motion_head", [d, d, d // 2, spec.head_out_dim], pred_feat, True, False)
    else:
        if spec.k_pred_mode == "cluster":   # (:70-74, :103-105) anchors from the K goal clusters, the same for every agent
            pe = fourier_fix(Wt["policy.act_decoder.k_goals"], d // 2)
            anchor = mlp(Wt, f"{pa}.cluster_mlp", [d, d], pe, False, False)[None].repeat(A, 1, 1)
        else:                               # (:93-101) learned anchors per (agent type, mode)
            type_idx = ((agent_type - 1) * K).unsqueeze(-1).repeat(1, K)
            anchor_index = torch.arange(K)[None, :].repeat(A, 1) + type_idx
            anchor = Wt[f"{pa}.motion_anchors.weight"][anchor_index]
        pred_emd, _ = cg_stacked(Wt, f"{pa}.CG_decode", anchor, pred_feat)
        motion = mlp(Wt, f"{pa}.motion_head", [d, d, d // 2, spec.out_dim], pred_emd, True, False)
    motion = motion.view(A, K, spec.target_steps, spec.state_dim)
    if noise is not None:
        motion = torch.cat([motion[..., :2] + noise.to(motion.dtype), motion[..., 2:]], dim=-1)
    traj = motion[..., :2].cumsum(dim=-2)
    headp = wrap_angle(motion[..., 2:3].cumsum(dim=-2))
    motion_pred = torch.cat([traj, headp, motion[..., 3:]], dim=-1)
    res = dict(motion_pred=motion_pred, motion_prob=torch.ones(A, K, dtype=motion.dtype))
    if spec.use_goal_pred_loss:   # (:128-130)
        res["reconst_pred"] = mlp(Wt, f"{pa}.pred_mlp", [d, d, d // 2, 2], policy_emd, True, False)
    return res


# --------------------------------------------------------------------------- closed-loop rollout

def rel_traj_coord_to_last_step(traj: T) -> T:
    """models/utils/geometry.py:24-45."""
    th = torch.atan2(traj[..., 2], traj[..., 3])
    origin = traj[..., -1, :]
    xy = traj[..., :2] - origin[..., None, :2]
    xy = batch_rotate_2d(xy, -th[..., -1:])
    d = wrap_angle(th - th[..., -1:])
    return torch.cat([xy, torch.sin(d)[..., None], torch.cos(d)[..., None]], dim=-1)


def rollout(w: Dict[str, np.ndarray], spec: ModelSpec, scene_in: Dict, dtype=torch.float32,
            collect: bool = False) -> Dict:
    """``ProSim.forward(batch, 'val')`` (traj_sam.py:59-175, 205-349, 562-633) for a batch of
    scenes whose policy agents sit in the slots of their observations (prompt_mask[b, n] implies that slot n is
    observed); observed agents without a prompt are log-replay agents.

    scene_in (numpy or torch, batch-major, padded):
      map_input [B,M,P,11], map_mask [B,M,P], map_pos [B,M,2], map_head [B,M]
      obs_input [B,N,11,24] (NaN where masked), obs_mask [B,N,11,24], obs_pos [B,N,2], obs_head [B,N]
      prompt [B,N,7], prompt_mask [B,N], agent_type [B,N] (1..3)
      optional fut_obs_input [R-1,B,N,11,24], fut_obs_mask [R-1,...] (defaults: the init tensors) and
      fut_obs_pos [R-1,B,N,2], fut_obs_head [R-1,B,N] (defaults: the init poses): the log of every agent at the
      later replans (batch.extras['fut_obs'][t], traj_sam.py:221-270); policy agents' rows are overwritten by the
      simulation, observed agents that are not policy agents (prompt_mask False) replay the log
      optional cond = {'goal': {...}, 'v_action_tag': {...}}
      optional mode_choice [R, B, N] int: motion mode followed per replan (models with motion_k > 1)
    """
    Wt = W(w, dtype)
    tt = lambda a, dt=dtype: (a if isinstance(a, torch.Tensor) else torch.from_numpy(np.asarray(a))).to(dt)
    map_input, map_pos, map_head = tt(scene_in["map_input"]), tt(scene_in["map_pos"]), tt(scene_in["map_head"])
    map_mask = tt(scene_in["map_mask"], torch.bool)
    obs_input, obs_pos, obs_head = tt(scene_in["obs_input"]), tt(scene_in["obs_pos"]), tt(scene_in["obs_head"])
    obs_mask = tt(scene_in["obs_mask"], torch.bool)
    prompt, prompt_mask = tt(scene_in["prompt"]), tt(scene_in["prompt_mask"], torch.bool)
    agent_type = tt(scene_in["agent_type"], torch.long)
    cond = scene_in.get("cond")
    if cond:
        cond = {k: dict(input=tt(v["input"]), mask=tt(v["mask"], torch.bool), prompt_idx=tt(v["prompt_idx"], torch.long))
                for k, v in cond.items()}
    B, N = prompt_mask.shape
    H = spec.hist_steps
    R = spec.n_replans
    trace = {}

    # encode_scene (traj_sam.py:73-77)
    map_emb, map_tok_mask = encode_map(Wt, spec, map_input, map_mask)
    obs_emb, obs_tok_mask = encode_obs(Wt, spec, obs_input, obs_mask)
    scene = scene_fusion(Wt, spec, map_emb, map_tok_mask, map_pos, map_head, obs_emb, obs_tok_mask, obs_pos, obs_head)
    edges = dict(scene["edges"])
    # encode_prompt + generate_policy (traj_sam.py:79-142)
    pemd = prompt_encode(Wt, spec, prompt)
    emd, e = decoder_fusion(Wt, spec, scene, pemd, prompt_mask, obs_pos, obs_head)
    edges.update(e)
    emd_dec = emd
    goal = goal_heads(Wt, spec, emd_dec, prompt_mask) if spec.goal_pred_k > 0 else None
    emd = condition_transform(Wt, spec, cond, emd, prompt_mask, obs_pos, obs_head)
    if collect:
        trace.update(map_emb=map_emb, obs_emb=obs_emb, scene_tokens=scene["scene_tokens"], prompt_emd=pemd,
                     policy_emd_dec=emd_dec, policy_emd=emd)

    # init_agent_trajs (traj_sam.py:597-633)
    traj = torch.zeros(B, N, H + R * spec.replan_freq, 4, dtype=dtype)
    vel = torch.zeros(B, N, H + R * spec.replan_freq, 2, dtype=dtype)
    traj[:, :, :H] = torch.nan_to_num(obs_input[..., :4], nan=0.0) * prompt_mask[..., None, None]
    vel[:, :, :H] = torch.nan_to_num(obs_input[..., 4:6], nan=0.0) * prompt_mask[..., None, None]
    init_pos = obs_pos * prompt_mask[..., None]
    init_head = (obs_head * prompt_mask)[..., None]

    pb = _flat_batch_idx(prompt_mask)
    p_emd = emd[prompt_mask]
    p_type = agent_type[prompt_mask]
    fut_in = scene_in.get("fut_obs_input")
    fut_mk = scene_in.get("fut_obs_mask")
    fut_pos, fut_head = scene_in.get("fut_obs_pos"), scene_in.get("fut_obs_head")
    motion_preds, fused, step_edges = [], [], []
    last = H
    for ti in range(R):
        # step_env (traj_sam.py:205-274)
        a_pos = init_pos + traj[:, :, last - 1, :2]
        a_th = torch.atan2(traj[:, :, last - 1, 2], traj[:, :, last - 1, 3])
        a_head = wrap_angle(a_th[:, :, None] + init_head)
        if ti > 0:
            f_in = tt(fut_in[ti - 1]).clone() if fut_in is not None else obs_input.clone()
            f_mk = tt(fut_mk[ti - 1], torch.bool).clone() if fut_mk is not None else obs_mask.clone()
            abs_tr = traj[:, :, last - H - 2:last]
            rel_tr = rel_traj_coord_to_last_step(abs_tr)
            th_last = torch.atan2(abs_tr[..., 2], abs_tr[..., 3])[..., -1:]
            if spec.pred_vel:
                rel_v = batch_rotate_2d(vel[:, :, last - H - 1:last], -th_last)  # rel_vel_coord_to_last_step
            else:                                                                # (:259, :553-554) velocities from the positions
                rel_v = torch.diff(rel_tr[..., :2], dim=2) / spec.dt
            rel_acc = torch.diff(rel_v, dim=2) / spec.dt                            # _get_rel_vel_acc
            rva = torch.cat([rel_v[:, :, 1:], rel_acc], dim=-1)
            f_in[prompt_mask, :, :4] = rel_tr[prompt_mask][:, -H:]
            f_in[prompt_mask, :, 4:8] = rva[prompt_mask]
            f_mk[prompt_mask] = True
            l_pos = tt(fut_pos[ti - 1]) if fut_pos is not None else obs_pos
            l_head = tt(fut_head[ti - 1]) if fut_head is not None else obs_head
            f_pos = torch.where(prompt_mask[..., None], a_pos, l_pos)
            f_head = torch.where(prompt_mask, a_head[..., 0], l_head)
            new_emb, new_mask = encode_obs(Wt, spec, f_in, f_mk)
            if spec.obs_fusion == "mlp":
                new_emb = fuse_obs_mlp(Wt, spec, scene, new_emb, new_mask)
            scene = replace_obs(scene, new_emb, new_mask, f_pos, f_head)
            if spec.obs_attn_update:
                scene = update_scene_attn(Wt, spec, scene)
            if collect:
                trace[f"obs_in_{ti}"] = f_in
        # decode_output -> policy (traj_sam.py:178-202, 441-525)
        nz = tt(scene_in["action_noise"][ti])[prompt_mask] if scene_in.get("action_noise") is not None else None   # [R, B, N, K, S, 2]
        out = policy_forward(Wt, spec, scene, p_emd, p_type, pb, a_pos[prompt_mask], a_head[prompt_mask], nz)
        motion_preds.append(out["motion_pred"])
        fused.append(out["fused"])
        step_edges.append(out["edges"])
        # step_agent_traj (traj_sam.py:276-349): the mode picked among the top-k (scene_in['mode_choice'] [R, B, N], the
        # reference's torch.topk + torch.randint draw replayed by the caller); TOP_K = 1 -> mode 0
        if scene_in.get("mode_choice") is not None:
            pick = tt(scene_in["mode_choice"][ti], torch.long)[prompt_mask]
            pred = out["motion_pred"][torch.arange(pick.shape[0]), pick, :spec.replan_freq]
        else:
            pred = out["motion_pred"][:, 0, :spec.replan_freq]
        cur_last = traj[:, :, last - 1][prompt_mask]
        lth = torch.atan2(cur_last[:, 2], cur_last[:, 3])[:, None]
        pxy = batch_rotate_2d(pred[:, :, :2], lth) + cur_last[:, None, :2]
        pth = wrap_angle(lth + pred[:, :, 2])
        fut = torch.cat([pxy, torch.sin(pth)[..., None], torch.cos(pth)[..., None]], dim=-1)
        new_t = torch.zeros(B, N, spec.replan_freq, 4, dtype=dtype)
        new_t[prompt_mask] = fut
        new_v = torch.zeros(B, N, spec.replan_freq, 2, dtype=dtype)
        if spec.pred_vel:
            new_v[prompt_mask] = batch_rotate_2d(pred[..., spec.vel_col:spec.vel_col + 2], lth)   # (6:8 with PRED_GMM, traj_sam.py:337-340)
        traj[:, :, last:last + spec.replan_freq] = new_t
        vel[:, :, last:last + spec.replan_freq] = new_v
        last += spec.replan_freq

    res = dict(traj=traj[:, :, H:], vel=vel[:, :, H:], init_pos=init_pos, init_heading=init_head,
               motion_pred=torch.cat(motion_preds, dim=0),
               policy_emd=emd, edges=edges, step_edges=step_edges)
    if "reconst_pred" in out:
        res["reconst_pred"] = out["reconst_pred"]
    if goal is not None:
        res.update(goal_prob=goal[0], goal_point=goal[1])
    if collect:
        trace["fused"] = torch.stack(fused)
        res["trace"] = trace
    return res
