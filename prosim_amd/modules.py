"""Host-side mirror of the reference's plugin interface for the rollout path.

The reference looks its components up by string from a registry (prosim/core/registry.py:54-134):
``MODEL.SCENE_ENCODER.TYPE`` (traj_sam.py:30), ``MODEL.DECODER.TYPE`` (:40), ``MODEL.POLICY.TYPE`` (:56),
``MODEL.TYPE`` (trainer.py:160).  The classes below register under the SAME names and keep the same
call signatures, argument meaning and error behaviour (Python exceptions), so the engine drops into
``ProSim`` for this path:

  scene_encoder(batch_obs, batch_map) -> dict            (traj_sam.py:77; keys attn_fusion.py:121-134)
  scene_encoder.update_scene_emb(scene_embs, batch_obs_new, old_obs_agent_ids) -> dict   (:550)
  decoder(scene_embs, prompt_enc) -> {'emd': [B,N,D], 'agent_type'}                       (:127)
  policy(policy_emd, batch_obs, batch_map, batch_pos, pair_names, latent_state) -> dict   (:640)
  ProSimHip.forward(batch, mode) -> {'motion_pred': {...}}                                 (:59)

All arithmetic runs in libprosim_hip.so through prosim_amd.engine (ctypes); torch tensors are only
the I/O format the reference's callers expect.  There is no CPU fallback.
"""
from __future__ import annotations

import collections
from typing import Any, Callable, Dict, List, Optional

import numpy as np
import torch

from .engine import Engine
from .spec import ModelSpec, DEMO_SPEC


class Registry:
    """Same surface as the reference's Registry (core/registry.py:25-134) for the kinds this path uses."""
    mapping: Dict[str, Dict[str, Any]] = collections.defaultdict(dict)

    @classmethod
    def _register(cls, kind: str, name: Optional[str]) -> Callable:
        def wrap(obj):
            cls.mapping[kind][obj.__name__ if name is None else name] = obj
            return obj
        return wrap

    @classmethod
    def _get(cls, kind: str, name: str):
        return cls.mapping[kind].get(name, None)


for _kind in ("model", "scene_encoder", "decoder", "policy", "prompt_encoder", "metric"):
    setattr(Registry, f"register_{_kind}", classmethod(lambda cls, to_register=None, *, name=None, _k=_kind:
                                                         cls._register(_k, name) if to_register is None else cls._register(_k, name)(to_register)))
    setattr(Registry, f"get_{_kind}", classmethod(lambda cls, name, _k=_kind: cls._get(_k, name)))
registry = Registry()


def _g(obj, key):
    """The reference's batch containers are dict-like (InputMaskData, dataset/format_utils.py:31)."""
    try:
        return obj[key]
    except (TypeError, KeyError, IndexError):
        return getattr(obj, key)


def _np(t, dtype=np.float32):
    if isinstance(t, torch.Tensor):
        t = t.detach().cpu().numpy()
    return np.ascontiguousarray(t, dtype=dtype)


class _Shared:
    """One Engine shared by the components of one model (the token store lives on the device)."""

    def __init__(self, spec: ModelSpec, weights: Dict[str, np.ndarray], device: int = 0):
        self.spec = spec
        self.engine = Engine(spec, weights, device=device)
        self.scene: Optional[Dict[str, np.ndarray]] = None


def prompt_to_slots(pr, ids_o, B, N, obs_pos, obs_head):
    """Dense prompt tensors (rows = policy agents, with ``agent_ids``) -> observation-slot layout ``[B, N, ...]``.
    Returns (slots, prompt, prompt_mask, agent_type, position, heading); ``slots[b][j]`` = slot of prompt row j."""
    ids_p = _g(pr, "agent_ids")
    p_prompt, p_mask = _np(_g(pr, "prompt")), _np(_g(pr, "prompt_mask"), bool)
    p_type, p_pos = _np(_g(pr, "agent_type"), np.int64), _np(_g(pr, "position"))
    p_head = _np(_g(pr, "heading"))
    p_head = p_head.reshape(p_head.shape[0], p_head.shape[1])
    slots = []
    for b in range(B):
        n_pol = int(p_mask[b].sum())
        if not p_mask[b, :n_pol].all():
            raise ValueError("prompt rows must be dense (valid rows first), as the reference's collate builds them")
        if ids_o is None or ids_p is None:
            if n_pol > N:
                raise ValueError("more policy agents than observation slots")
            row = list(range(n_pol))                       # no ids: prompt j belongs to observation slot j
        else:
            where = {a: n for n, a in enumerate(ids_o[b])}
            missing = [a for a in list(ids_p[b])[:n_pol] if a not in where]
            if missing:
                raise ValueError(f"policy agents {missing} of scene {b} are not among the observed agents")
            row = [where[a] for a in list(ids_p[b])[:n_pol]]
        slots.append(row)
    prompt = np.zeros((B, N, p_prompt.shape[-1]), np.float32)
    prompt_mask = np.zeros((B, N), bool)
    agent_type = np.ones((B, N), np.int64)
    prompt_pos, prompt_head = np.array(obs_pos, np.float32), np.array(obs_head, np.float32).reshape(B, N)
    for b in range(B):
        for j, n in enumerate(slots[b]):
            prompt[b, n], prompt_mask[b, n], agent_type[b, n] = p_prompt[b, j], True, p_type[b, j]
            prompt_pos[b, n], prompt_head[b, n] = p_pos[b, j], p_head[b, j]
    return slots, prompt, prompt_mask, agent_type, prompt_pos, prompt_head


# PROMPT.CONDITION.TYPES of prosim_demo/cfg/no_text.yaml:65, plus the binary 'v2v_tag' (default.py:337)
COND_TYPES = ("goal", "v_action_tag", "drag_point", "v2v_tag")


def cond_to_slots(c, slots, Np):
    """A condition's ``prompt_idx`` (rows of the dense prompt tensor) -> observation slots; conditions that point
    at a padding row are masked off."""
    idx = _np(_g(c, "prompt_idx"), np.int64)
    cm = _np(_g(c, "mask"), bool).copy()
    sl = np.zeros_like(idx)
    for b in range(idx.shape[0]):
        lut = np.array(list(slots[b]) + [0] * (Np - len(slots[b])), np.int64)
        sl[b] = lut[np.clip(idx[b], 0, Np - 1)]
        cm[b] &= (idx[b] < len(slots[b])).all(-1)                # (binary conditions: both ends)
    return dict(input=_np(_g(c, "input")), mask=cm, prompt_idx=sl)


def scene_from_extras(extras, spec: ModelSpec, task: str = "motion_pred") -> Dict[str, np.ndarray]:
    """``batch.extras`` (dataset/format_utils.py:798-815) -> the engine's input dict.

    The reference matches policy agents (``prompt[task].agent_ids``) to observations (``init_obs.agent_ids``) by
    id (traj_sam.py:246-250); the engine wants a policy agent's prompt in the SLOT of its observation, so the dense
    prompt tensors are scattered into observation slots here.  Observed agents without a prompt are log-replay
    agents: ``fut_obs[t]`` (observation, mask, pose of all agents at the later replans) drives their scene tokens.
    ``scene['_policy_slots'][b]`` lists, in PROMPT order, the slot of every policy agent (the order of the outputs).
    ``fut_obs`` frames are matched to the slots by id; an agent that leaves gets a masked frame, an agent that enters
    later gets a new slot.  Raises if a policy agent is not observed, or is missing from a frame."""
    obs, mp, pr = extras["init_obs"], extras["init_map"], extras["prompt"][task]
    obs_input = _np(_g(obs, "input"))
    B, N = obs_input.shape[:2]
    ids_o = _g(obs, "agent_ids")
    Np = _np(_g(pr, "prompt_mask")).shape[1]
    slots, prompt, prompt_mask, agent_type, prompt_pos, prompt_head = prompt_to_slots(
        pr, ids_o, B, N, _np(_g(obs, "position")), _np(_g(obs, "heading")))
    scene = dict(map_input=_np(_g(mp, "input")), map_mask=_np(_g(mp, "mask"), bool), map_pos=_np(_g(mp, "position")),
                 map_head=_np(_g(mp, "heading")), obs_input=obs_input, obs_mask=_np(_g(obs, "mask"), bool),
                 obs_pos=_np(_g(obs, "position")), obs_head=_np(_g(obs, "heading")), prompt=prompt,
                 prompt_mask=prompt_mask, agent_type=agent_type, prompt_pos=prompt_pos, prompt_head=prompt_head,
                 _policy_slots=slots, _obs_ids=ids_o)
    cond = extras.get("condition") if hasattr(extras, "get") else None
    if cond:
        out = {}
        for k in COND_TYPES:
            if k in cond.keys() and _g(cond[k], "input").shape[1] > 0:
                out[k] = cond_to_slots(cond[k], slots, Np)
        unsupported = [k for k in cond.keys() if k not in COND_TYPES and _g(cond[k], "input").shape[1] > 0]
        if unsupported:
            raise NotImplementedError(f"condition types {unsupported} are not built (the demo config's unary types "
                                      f"{COND_TYPES} are; the text types are out of scope)")
        if out:
            scene["cond"] = out
    fut = extras.get("fut_obs") if hasattr(extras, "get") else None
    if fut:
        # Every frame lists its own agents (get_center_obs drops a non-target agent whose state at that step is NaN,
        # format_utils.py:383-388), so frames are matched to the observation slots BY ID.  An agent that has left the
        # scene gets a fully masked frame (no scene token at that replan); an agent that only ENTERS with a later
        # frame gets a new slot behind the init_obs ones, without history at the initial step (the engine keeps a
        # token row for it that joins the scene with its first valid frame).
        ts = sorted(int(t) for t in fut.keys())
        H, F = obs_input.shape[2], obs_input.shape[3]
        ids_all = None
        if ids_o is not None and all(_g(fut[t], "agent_ids") is not None for t in ts):
            ids_all = [list(ids_o[b_]) for b_ in range(B)]
            for t in ts:
                for b_ in range(B):
                    known = set(ids_all[b_])
                    ids_all[b_] += [a for a in _g(fut[t], "agent_ids")[b_] if a not in known]
            N2 = max(N, max(len(x) for x in ids_all))
            if N2 > N:                                           # slots for the agents that enter later
                def grow(a, fill):
                    out = np.full((B, N2) + a.shape[2:], fill, a.dtype)
                    out[:, :N] = a
                    return out
                scene.update(obs_input=grow(scene["obs_input"], np.nan), obs_mask=grow(scene["obs_mask"], False),
                             obs_pos=grow(scene["obs_pos"], 0.0), obs_head=grow(scene["obs_head"].reshape(B, N), 0.0),
                             prompt=grow(scene["prompt"], 0.0), prompt_mask=grow(scene["prompt_mask"], False),
                             agent_type=grow(scene["agent_type"], 1), prompt_pos=grow(scene["prompt_pos"], 0.0),
                             prompt_head=grow(scene["prompt_head"], 0.0))
                N = N2
            scene["_obs_ids"] = ids_all
        f_in = np.full((len(ts), B, N, H, F), np.nan, np.float32)
        f_mk = np.zeros((len(ts), B, N, H, F), bool)
        f_pos = np.repeat(scene["obs_pos"][None], len(ts), 0).copy()
        f_head = np.repeat(scene["obs_head"].reshape(1, B, N), len(ts), 0).copy()
        for k, t in enumerate(ts):
            fr = fut[t]
            ids_t = _g(fr, "agent_ids")
            fi, fm = _np(_g(fr, "input")), _np(_g(fr, "mask"), bool)
            fp_, fh = _np(_g(fr, "position")), _np(_g(fr, "heading"))
            fh = fh.reshape(fh.shape[0], -1)
            if ids_all is None:
                if fi.shape[:2] != (B, N):
                    raise ValueError("fut_obs frames without agent_ids must keep the [B, N] layout of init_obs")
                f_in[k], f_mk[k], f_pos[k], f_head[k] = fi, fm, fp_, fh
                continue
            for b_ in range(B):
                where = {a: n for n, a in enumerate(ids_all[b_])}
                for j, a in enumerate(ids_t[b_]):
                    n = where[a]
                    f_in[k, b_, n], f_mk[k, b_, n], f_pos[k, b_, n], f_head[k, b_, n] = fi[b_, j], fm[b_, j], fp_[b_, j], fh[b_, j]
                gone = [ids_all[b_][n] for n in slots[b_] if ids_all[b_][n] not in set(ids_t[b_])]
                if gone:   # target agents are listed in every frame (format_utils.py:381-385); their static columns come from it
                    raise ValueError(f"policy agents {gone} of scene {b_} are missing from fut_obs[{t}]")
        scene["fut_obs_input"], scene["fut_obs_mask"], scene["fut_obs_pos"], scene["fut_obs_head"] = f_in, f_mk, f_pos, f_head
    return scene


@registry.register_scene_encoder(name="attn_fusion_relpe")
class HipSceneEncoder:
    """AttentionSceneEncoderRelPE (scene_encoder/attn_fusion.py:11) on the HIP engine."""

    def __init__(self, shared: _Shared):
        self.s = shared

    def _result(self, scene) -> Dict[str, Any]:
        eng = self.s.engine
        mm = torch.from_numpy(scene["map_mask"].astype(bool)).any(-1)
        om = torch.from_numpy(scene["obs_mask"].astype(bool).all(-1).any(-1))    # observed agents = agent tokens
        B = mm.shape[0]
        flat = lambda m: torch.arange(B).unsqueeze(1).repeat(1, m.shape[1]).view(-1)[m.view(-1)]
        mb, ob = flat(mm), flat(om)
        Mv = int(mm.sum())
        return dict(obs_mask=om, map_mask=mm, scene_batch_idx=torch.cat([mb, ob]),
                    scene_type=torch.cat([torch.zeros_like(mb), torch.ones_like(ob)]),
                    scene_pos=torch.cat([torch.from_numpy(scene["map_pos"])[mm], torch.from_numpy(scene["obs_pos"])[om]]),
                    scene_ori=torch.cat([torch.from_numpy(scene["map_head"])[mm], torch.from_numpy(scene["obs_head"])[om]])[:, None],
                    scene_tokens=torch.from_numpy(self._tokens()), max_map_num=mm.shape[1], max_agent_num=om.shape[1],
                    _hip_resident=True, _n_map_tokens=Mv)

    def _tokens(self) -> np.ndarray:
        """[map tokens ; tokens of the agents that are in the scene at the initial step] (rows of agents that only
        enter later are engine-internal)."""
        eng = self.s.engine
        tok = eng.get("scene_tokens")
        Mv = eng.num_map_tokens
        return np.concatenate([tok[:Mv], tok[Mv:][eng.live0_rows]])

    def __call__(self, batch_obs, batch_map):
        return self.forward(batch_obs, batch_map)

    def forward(self, batch_obs, batch_map) -> Dict[str, Any]:
        spec = self.s.spec
        obs_in, obs_mask = _np(_g(batch_obs, "input")), _np(_g(batch_obs, "mask"), bool)
        B, N = obs_in.shape[:2]
        valid = obs_mask.all(-1).any(-1)
        scene = dict(map_input=_np(_g(batch_map, "input")), map_mask=_np(_g(batch_map, "mask"), bool),
                     map_pos=_np(_g(batch_map, "position")), map_head=_np(_g(batch_map, "heading")), obs_input=obs_in,
                     obs_mask=obs_mask, obs_pos=_np(_g(batch_obs, "position")), obs_head=_np(_g(batch_obs, "heading")),
                     # the prompt side arrives with decoder(); placeholders keep the engine's batch consistent
                     prompt=np.zeros((B, N, spec.prompt_dim), np.float32), prompt_mask=valid, agent_type=np.ones((B, N), np.int64))
        scene["_obs_ids"] = _g(batch_obs, "agent_ids")
        self.s.scene = scene
        self.s.engine.set_scene(scene)
        self.s.engine.encode_scene()
        return self._result(scene)

    def update_scene_emb(self, scene_embs, batch_obs, old_obs_agent_ids):
        """``update_scene_emb`` (attn_fusion.py:238-252) for the demo config's OBS_UPDATE (FUSION 'replace', ATTN_UPDATE
        False): agents re-encoded from ``batch_obs``, map tokens reused.  Inside a rollout the engine does this on
        the device (ps_policy_step); this call serves code that drives the encoder itself.  The agents of ``batch_obs``
        may be ANY set (``_replace_old_obs`` :205-236 takes whatever the new observation lists -- agents that left are
        gone, new ones are there, in the new order): the engine is then given the new batch and the old map tokens."""
        if not scene_embs.get("_hip_resident") or self.s.scene is None:
            raise ValueError("scene_embs must come from HipSceneEncoder (tokens are device-resident)")
        eng, old = self.s.engine, self.s.scene
        new_ids = _g(batch_obs, "agent_ids")
        obs_in, obs_mask = _np(_g(batch_obs, "input")), _np(_g(batch_obs, "mask"), bool)
        B, N = obs_in.shape[:2]
        same = (new_ids is None or old_obs_agent_ids is None or [list(a) for a in new_ids] == [list(a) for a in old_obs_agent_ids]) \
            and obs_in.shape[:2] == old["obs_input"].shape[:2] \
            and np.array_equal(obs_mask.all(-1).any(-1), old["obs_mask"].astype(bool).all(-1).any(-1))
        scene = dict(old)
        scene.update(obs_input=obs_in, obs_mask=obs_mask, obs_pos=_np(_g(batch_obs, "position")),
                     obs_head=_np(_g(batch_obs, "heading")).reshape(B, N))
        if not same:
            # another agent set: upload the new batch (same map), give the map tokens back, re-encode the agents
            map_tok = eng.get("scene_tokens")[:eng.num_map_tokens]
            valid = obs_mask.all(-1).any(-1)
            if not valid.any(1).all():
                raise ValueError("update_scene_emb: a scene of the new observation has no observed agent")
            scene.update(prompt=np.zeros((B, N, self.s.spec.prompt_dim), np.float32), prompt_mask=valid,
                         agent_type=np.ones((B, N), np.int64), _obs_ids=new_ids)
            for k in ("prompt_pos", "prompt_head", "cond", "mode_choice", "fut_obs_input", "fut_obs_mask", "fut_obs_pos", "fut_obs_head", "_policy_slots"):
                scene.pop(k, None)
            eng.set_scene(scene)
            eng.set_map_tokens(map_tok)
        eng.update_obs(scene["obs_input"], scene["obs_mask"], scene["obs_pos"], scene["obs_head"])
        self.s.scene = scene
        return self._result(scene)


@registry.register_decoder(name="attn_fusion_relpe")
class HipDecoder:
    """SymCoordDecoder (decoder/sym_coord.py:15) + the condition transformer at 'policy_decoder' (traj_sam.py:129-137)."""

    def __init__(self, shared: _Shared):
        self.s = shared

    def __call__(self, scene_emb, prompt_enc, condition=None):
        return self.forward(scene_emb, prompt_enc, condition)

    def forward(self, scene_emb, prompt_enc, condition=None) -> Dict[str, Any]:
        if not scene_emb.get("_hip_resident"):
            raise ValueError("scene_emb must come from HipSceneEncoder (tokens are device-resident)")
        sc, eng = self.s.scene, self.s.engine
        B, N = sc["prompt_mask"].shape
        slots, prompt, pm, a_type, p_pos, p_head = prompt_to_slots(prompt_enc, sc.get("_obs_ids"), B, N, sc["obs_pos"], sc["obs_head"])
        if not np.array_equal(pm, sc["prompt_mask"]):
            # the staged call encoded every observed agent as a policy agent; a policy SUBSET needs the log frames
            # of the others, which only ProSimHip.forward(batch) receives (extras['fut_obs'])
            raise NotImplementedError("staged decoder(): prompt agents must be the observed agents; "
                                      "use the model's forward(batch) for scenes with log-replay agents")
        eng.set_prompt(prompt, p_pos, p_head, a_type.astype(np.int32))
        sc["_policy_slots"] = slots   # (a scene that came from update_scene_emb with another agent set has none yet)
        Np = _np(_g(prompt_enc, "prompt_mask")).shape[1]
        if condition:
            unsupported = [k for k in condition.keys() if k not in COND_TYPES and np.asarray(condition[k]["input"]).shape[1] > 0]
            if unsupported:
                raise NotImplementedError(f"condition types {unsupported} are not built")
            eng.set_conditions({k: cond_to_slots(condition[k], slots, Np) for k in COND_TYPES
                                if k in condition.keys() and np.asarray(condition[k]["input"]).shape[1] > 0})
        eng.generate_policy()
        emd_slots = eng.padded("policy_emd")                       # [B, N, D] by observation slot
        emd = np.zeros((B, Np, emd_slots.shape[-1]), np.float32)   # the reference returns prompt order (sym_coord.py:60-75)
        for b in range(B):
            emd[b, :len(slots[b])] = emd_slots[b, slots[b]]
        out = dict(emd=torch.from_numpy(emd), agent_type=torch.from_numpy(_np(_g(prompt_enc, "agent_type"), np.int64)))
        out.update(_goal_outputs(eng, self.s.spec, slots, B, Np))
        return out


def _goal_outputs(eng, spec, slots, B, Np) -> Dict[str, Any]:
    """``Decoder._goal_pred`` (decoder/base.py:22-58) when MODEL.DECODER.GOAL_PRED is enabled: goal_prob [B, N, K] and
    goal_point [B, N, K, 2] in prompt order, zeros past the prompts."""
    K = spec.goal_pred_k
    if K <= 0:
        return {}
    gp, gq = eng.padded("goal_prob"), eng.padded("goal_point")
    prob, point = np.zeros((B, Np, K), np.float32), np.zeros((B, Np, K, 2), np.float32)
    for b in range(B):
        prob[b, :len(slots[b])] = gp[b, slots[b]]
        point[b, :len(slots[b])] = gq[b, slots[b]]
    return dict(goal_prob=torch.from_numpy(prob), goal_point=torch.from_numpy(point))


@registry.register_policy(name="rel_pe_temporal")
class HipPolicy:
    """Policy_RelPE_Temporal / PolicyNoRNN (policy/base.py:9, temporal_ar.py:64) on the HIP engine."""

    def __init__(self, shared: _Shared):
        self.s = shared

    def format_latent_state(self, latent_state_dict, all_batch_pair_names):
        return None   # policy_no_rnn keeps no state (temporal_ar.py:70-72)

    def __call__(self, *a, **k):
        return self.forward(*a, **k)

    def forward(self, policy_emd, batch_obs, batch_map, batch_pos, pair_names, latent_state) -> Dict[str, Any]:
        def flat(bs):
            m = _np(_g(bs, "mask"), bool)
            BT = m.shape[0]
            idx = np.repeat(np.arange(BT), m.shape[1]).reshape(m.shape)[m]
            return (_np(_g(bs, "input"))[m], _np(_g(bs, "pos"))[m], _np(_g(bs, "ori"))[m].reshape(-1), idx, BT)
        a_tok, a_pos, a_ori, a_b, BT = flat(batch_obs)
        m_tok, m_pos, m_ori, m_b, _ = flat(batch_map)
        emd = _np(_g(policy_emd, "emd"))
        mp, fused = self.s.engine.policy_forward(BT, a_tok, a_pos, a_ori, a_b, m_tok, m_pos, m_ori, m_b, emd,
                                                 _np(_g(batch_pos, "position")), _np(_g(batch_pos, "heading")).reshape(-1),
                                                 _np(_g(policy_emd, "agent_type"), np.int32), _np(_g(policy_emd, "batch_idx"), np.int32))
        if len(pair_names) != emd.shape[0]:
            raise AssertionError("pair_names must name every policy row")     # the reference indexes by it (temporal_ar.py:22-35)
        mp = torch.from_numpy(mp)
        return dict(motion_pred=mp, motion_prob=torch.ones(mp.shape[0], mp.shape[1]), latent_state=latent_state,
                    fused=torch.from_numpy(fused))


@registry.register_model(name="prosim_policy_relpe_T_step_temporal_close_loop")
class ProSimHip:
    """``ProSim.forward(batch, 'val')`` (models/traj_sam.py:59-71) with the whole closed loop on the device."""

    def __init__(self, spec: ModelSpec = DEMO_SPEC, weights: Optional[Dict[str, np.ndarray]] = None, device: int = 0):
        if weights is None:
            raise ValueError("weights required (prosim_amd.weights.from_state_dict(ckpt['state_dict']) or init_weights)")
        self.spec = spec
        self.tasks = ["motion_pred"]
        self._shared = _Shared(spec, weights, device)
        self.scene_encoder = registry.get_scene_encoder("attn_fusion_relpe")(self._shared)
        self.decoder = registry.get_decoder("attn_fusion_relpe")(self._shared)
        self.policy = registry.get_policy("rel_pe_temporal")(self._shared)

    @property
    def engine(self) -> Engine:
        return self._shared.engine

    def close(self):
        """Release the engine (device buffers, stream, graph)."""
        self._shared.engine.close()

    def eval(self):
        return self   # inference only: dropout never runs on this path

    def __call__(self, batch, mode="val"):
        return self.forward(batch, mode)

    def forward(self, batch, mode="val") -> Dict[str, Any]:
        """The whole closed loop as ONE device graph (ps_rollout).  The staged methods below give the same values
        call by call, in the reference's order (traj_sam.py:59-71)."""
        if mode == "train":
            raise NotImplementedError("training is out of scope (DESIGN.md)")
        extras = batch.extras if hasattr(batch, "extras") else batch
        scene = scene_from_extras(extras, self.spec)
        self._shared.scene = scene
        self.engine.set_scene(scene)
        self._last_mode_choice = self._draw_mode_choice()
        self.engine.set_mode_choice(self._last_mode_choice)
        self.engine.rollout()
        return self._process_rollout(extras, scene)

    # ---- the reference's stages (traj_sam.py:73-203), device-resident between calls -------------------------------
    def encode_scene(self, batch) -> Dict[str, Any]:
        """``ProSim.encode_scene`` (:73-77).  Takes the whole batch, so policy agents may be a subset of the
        observed agents here (the prompt and ``fut_obs`` are at hand), unlike the bare scene_encoder call."""
        extras = batch.extras if hasattr(batch, "extras") else batch
        scene = scene_from_extras(extras, self.spec)
        self._shared.scene = scene
        self.engine.set_scene(scene)
        self.engine.encode_scene()
        return self.scene_encoder._result(scene)

    def encode_prompt(self, batch, prompt_dict={}) -> Dict[str, Any]:
        """``ProSim.encode_prompt`` (:79-103).  The prompt MLP runs inside ps_generate_policy; the prompt data pass
        through (a ``prompt_dict`` override replaces the prompt rows that encode_scene uploaded)."""
        extras = batch.extras if hasattr(batch, "extras") else batch
        encs = {}
        for task in (prompt_dict.keys() if prompt_dict else self.tasks):
            encs[task] = prompt_dict[task] if task in prompt_dict else extras["prompt"][task]
        return encs

    def generate_policy(self, batch, scene_embs, prompt_encs) -> Dict[str, Any]:
        """``ProSim.generate_policy`` (:118-142): decoder + condition transformer -> ``{task: {'emd','agent_type'}}``."""
        extras = batch.extras if hasattr(batch, "extras") else batch
        cond = extras.get("condition") if hasattr(extras, "get") else None
        sc = self._shared.scene
        if sc is None or not scene_embs.get("_hip_resident"):
            raise ValueError("scene_embs must come from this model's encode_scene (tokens are device-resident)")
        out = {}
        for task, pe in prompt_encs.items():
            B, N = sc["prompt_mask"].shape
            slots, prompt, pm, a_type, p_pos, p_head = prompt_to_slots(pe, sc.get("_obs_ids"), B, N, sc["obs_pos"], sc["obs_head"])
            if not np.array_equal(pm, sc["prompt_mask"]):
                raise ValueError("generate_policy: the prompt agents differ from the ones encode_scene saw")
            self.engine.set_prompt(prompt, p_pos, p_head, a_type.astype(np.int32))
            sc["_policy_slots"] = slots
            Np = _np(_g(pe, "prompt_mask")).shape[1]
            if cond:
                unsupported = [k for k in cond.keys() if k not in COND_TYPES and _g(cond[k], "input").shape[1] > 0]
                if unsupported:
                    raise NotImplementedError(f"condition types {unsupported} are not built")
                sc["cond"] = {k: cond_to_slots(cond[k], slots, Np) for k in COND_TYPES
                              if k in cond.keys() and _g(cond[k], "input").shape[1] > 0}
                self.engine.set_conditions(sc["cond"])
            self.engine.generate_policy()
            emd_slots = self.engine.padded("policy_emd")
            emd = np.zeros((B, Np, emd_slots.shape[-1]), np.float32)
            for b in range(B):
                emd[b, :len(slots[b])] = emd_slots[b, slots[b]]
            out[task] = dict(emd=torch.from_numpy(emd), agent_type=torch.from_numpy(_np(_g(pe, "agent_type"), np.int64)),
                             _hip_resident=True)
            out[task].update(_goal_outputs(self.engine, self.spec, slots, B, Np))
        return out

    def init_agent_trajs(self, policy_agent_ids, batch) -> Dict[str, Any]:
        """``ProSim.init_agent_trajs`` (:597-633).  The trajectory state lives on the device (ps_reset_rollout); the dict
        that comes back mirrors the reference's (``traj`` [B, N, hist, 4], ``vel`` [B, N, hist, 2] from the history with NaN ->
        0, ``init_pos``, ``init_heading``, ``last_step`` = hist) in policy-agent order, for callers that read it."""
        self.engine.reset_rollout()
        sc, spec = self._shared.scene, self.spec
        pslots = sc["_policy_slots"]
        B, N, H = len(pslots), max(len(p) for p in pslots), spec.hist_steps
        traj, vel = torch.zeros(B, N, H, 4), torch.zeros(B, N, H, 2)
        pos, head = torch.zeros(B, N, 2), torch.zeros(B, N, 1)
        obs = torch.nan_to_num(torch.from_numpy(np.asarray(sc["obs_input"], np.float32)), nan=0.0)
        for b in range(B):
            for j, n in enumerate(pslots[b]):
                traj[b, j], vel[b, j] = obs[b, n, :, :4], obs[b, n, :, 4:6]
                pos[b, j] = torch.from_numpy(np.asarray(sc["obs_pos"][b, n], np.float32))
                head[b, j, 0] = float(np.asarray(sc["obs_head"]).reshape(len(pslots), -1)[b, n])
        self._step_state = dict(next=0, out=None)
        return {task: dict(traj=traj, vel=vel, init_pos=pos, init_heading=head, last_step=H, _hip_resident=True) for task in self.tasks}

    # ---- one iteration of rollout_batch, call by call (traj_sam.py:159-172) ---------------------------------------------------
    # The engine runs an iteration as ONE entry point (ps_policy_step: observation refresh, policy, trajectory append, all on
    # the device).  The three methods below keep the reference's call sequence and return values: step_env reports the poses the
    # iteration starts from, decode_output runs ps_policy_step and returns the policy's output, step_agent_traj mirrors the
    # appended steps into ``agent_trajs`` -- and when it is handed a ``model_output`` other than the one decode_output returned
    # (a caller that edits the prediction), it applies the reference's update (:311-347) on the host and pushes the state to
    # the device (ps_set_state).
    def _replan_index(self, t) -> int:
        want = list(self.spec.all_t_indices)
        if int(t) not in want:
            raise ValueError(f"t = {t} is not a replan step of the spec ({want})")
        return want.index(int(t))

    def step_env(self, scene_embs, a_traj, batch, policy_agent_ids, t, all_t_indices):
        """``ProSim.step_env`` (:205-274) -> (scene_embs, a_pos).  ``a_pos`` = the poses replan ``t`` starts from, by the
        reference's formula (:212-216: init_pos + last xy, wrap(atan2(last sin, last cos) + init_heading)); the observation
        refresh and the scene-token update of the replan (K12 + update_scene_emb) are the first part of ps_policy_step,
        which ``decode_output`` runs for the same ``t``."""
        from .spec import wrap_angle_np
        i = self._replan_index(t)
        st = getattr(self, "_step_state", None)
        if st is None or i != st["next"]:
            raise RuntimeError(f"step_env: replan {i} out of order (expected {None if st is None else st['next']}; call init_agent_trajs first)")
        tr = a_traj["motion_pred"]
        last = tr["traj"][..., tr["last_step"] - 1, :]
        a_pos = {"position": tr["init_pos"] + last[..., :2],
                 "heading": torch.from_numpy(wrap_angle_np((torch.atan2(last[..., 2], last[..., 3])[..., None] + tr["init_heading"]).numpy()))}
        return scene_embs, a_pos

    def decode_output(self, policy_emds, scene_embs, policy_agent_ids, batch, agent_positions=None, target_t=None, latent_state_dict=None):
        """``ProSim.decode_output`` (:178-203): the policy's output for every policy agent at replan ``target_t`` --
        ``{task: {'motion_pred' [P, K, S, D], 'motion_prob' [P, K], 'latent_state', 'pair_names'}}``, rows in the order of
        ``pair_names`` (scene-major, prompt order; :478)."""
        i = self._replan_index(target_t)
        st = self._step_state
        if i != st["next"]:
            raise RuntimeError(f"decode_output: replan {i} out of order (expected {st['next']})")
        eng, sc, spec = self.engine, self._shared.scene, self.spec
        if i == 0:
            self._last_mode_choice = self._draw_mode_choice()
            eng.set_mode_choice(self._last_mode_choice)
        eng.policy_step(i)
        B, N = sc["prompt_mask"].shape
        pslots = sc["_policy_slots"]
        row_of_slot = {int(sl): r for r, sl in enumerate(eng._slots)}
        order = [row_of_slot[b * N + n] for b in range(B) for n in pslots[b]]
        mp = torch.from_numpy(eng.get("motion_pred")[i][order])
        extras = batch.extras if hasattr(batch, "extras") else batch
        ids = _g(extras["prompt"]["motion_pred"], "agent_ids") or [[str(j) for j in range(len(pslots[b]))] for b in range(B)]
        names = [f"{b}-{ids[b][j]}-{int(target_t)}" for b in range(B) for j in range(len(pslots[b]))]
        out = {"motion_pred": mp, "motion_prob": torch.ones(mp.shape[0], mp.shape[1]), "latent_state": None, "pair_names": names}
        if self.spec.use_goal_pred_loss:   # (act_decoder.py:128-130)
            out["reconst_pred"] = torch.from_numpy(eng.get("reconst_pred")[order])
        st["out"] = (i, mp)
        return {"motion_pred": out}

    def get_action(self, policy_emb, obs_data, map_data, pos_data, pair_names, latent_state=None):
        """``ProSim.get_action`` (:635-640): the stateless policy call on explicit tokens."""
        return self.policy(policy_emb, obs_data, map_data, pos_data, [n for group in pair_names for n in group], latent_state)

    def step_agent_traj(self, a_traj, model_output, policy_agent_ids, t, mode="val"):
        """``ProSim.step_agent_traj`` (:276-349): the next ``replan_freq`` steps appended to ``a_traj`` (in place, as there)."""
        i = self._replan_index(t)
        st, spec, eng, sc = self._step_state, self.spec, self.engine, self._shared.scene
        if i != st["next"] or st["out"] is None or st["out"][0] != i:
            raise RuntimeError(f"step_agent_traj: replan {i} has not been decoded (call decode_output first)")
        tr = a_traj["motion_pred"]
        S, H = spec.replan_freq, spec.hist_steps
        pslots = sc["_policy_slots"]
        B, N = sc["prompt_mask"].shape
        mine = model_output["motion_pred"]["motion_pred"] is st["out"][1]
        if mine:   # the device has appended these very steps: mirror them
            traj, vel = eng.padded("traj"), eng.padded("vel")
            new_t, new_v = torch.zeros(tr["traj"].shape[0], tr["traj"].shape[1], S, 4), torch.zeros(tr["traj"].shape[0], tr["traj"].shape[1], S, 2)
            for b in range(len(pslots)):
                for j, n in enumerate(pslots[b]):
                    new_t[b, j] = torch.from_numpy(traj[b, n, i * S:(i + 1) * S])
                    new_v[b, j] = torch.from_numpy(vel[b, n, i * S:(i + 1) * S])
        else:      # an edited prediction: the reference's update on the host (:311-347), then the state goes to the device
            from .spec import wrap_angle_np
            mp = model_output["motion_pred"]["motion_pred"]
            mc = self._last_mode_choice
            new_t, new_v = torch.zeros(tr["traj"].shape[0], tr["traj"].shape[1], S, 4), torch.zeros(tr["traj"].shape[0], tr["traj"].shape[1], S, 2)
            p = 0
            for b in range(len(pslots)):
                for j, n in enumerate(pslots[b]):
                    k = 0 if mc is None else int(mc[i, b, n])
                    pred = mp[p, k, :S]
                    cur = tr["traj"][b, j, tr["last_step"] - 1]
                    th = torch.atan2(cur[2], cur[3])
                    c, s_ = torch.cos(th), torch.sin(th)
                    xy = torch.stack([c * pred[:, 0] - s_ * pred[:, 1], s_ * pred[:, 0] + c * pred[:, 1]], -1) + cur[:2]
                    ang = torch.from_numpy(wrap_angle_np((th + pred[:, 2]).numpy()))
                    new_t[b, j] = torch.cat([xy, torch.sin(ang)[:, None], torch.cos(ang)[:, None]], -1)
                    if spec.pred_vel:   # (traj_sam.py:337-345; without PRED_VEL no velocity track is kept)
                        v = pred[:, spec.vel_col:spec.vel_col + 2]
                        new_v[b, j] = torch.stack([c * v[:, 0] - s_ * v[:, 1], s_ * v[:, 0] + c * v[:, 1]], -1)
                    p += 1
        tr["traj"] = torch.cat([tr["traj"], new_t], 2)
        tr["vel"] = torch.cat([tr["vel"], new_v], 2)
        tr["last_step"] += S
        if not mine:
            rows_t, rows_v = [], []
            for b in range(len(pslots)):
                for j in range(len(pslots[b])):
                    rows_t.append(tr["traj"][b, j].numpy())
                    rows_v.append(tr["vel"][b, j].numpy())
            if len(rows_t) != eng.num_agents:
                raise NotImplementedError("an edited prediction with log-replay agents in the scene: ps_set_state takes the policy agents only")
            eng.set_state(np.stack(rows_t), np.stack(rows_v))
        st["next"], st["out"] = i + 1, None
        return a_traj

    def rollout_batch(self, batch, scene_embs, policy_emds, policy_agent_ids, agent_trajs, all_t_indices, mode="val"):
        """``ProSim.rollout_batch`` (:144-176): per replan step_env -> decode_output -> step_agent_traj = one
        ``ps_policy_step``; then ``_process_rollout`` (:562-595)."""
        if mode == "train":
            raise NotImplementedError("training is out of scope (DESIGN.md)")
        want = list(self.spec.all_t_indices)
        if [int(t) for t in all_t_indices] != want:
            raise ValueError(f"all_t_indices must be {want} (ROLLOUT.POLICY.REPLAN_FREQ / MAX_STEPS of the spec)")
        if not all(v.get("_hip_resident") for v in (scene_embs, *policy_emds.values(), *agent_trajs.values())):
            raise ValueError("rollout_batch needs the device-resident results of encode_scene / generate_policy / init_agent_trajs")
        self._last_mode_choice = self._draw_mode_choice()
        self.engine.set_mode_choice(self._last_mode_choice)
        self._step_state = None
        for i in range(len(want)):
            self.engine.policy_step(i)
        extras = batch.extras if hasattr(batch, "extras") else batch
        return self._process_rollout(extras, self._shared.scene)

    def _draw_mode_choice(self):
        """The rollout's random draws, made on the host with the reference's own calls in the reference's order, so that a
        seeded torch generator yields the reference's stream: per replan ``torch.randn_like`` of the xy slice of the policy's
        [P, K, S, D] output when RANDOM_NOISE_STD > 0 (act_decoder.py:113-115), then ``ProSim.step_agent_traj``'s
        ``torch.topk`` over the (all-ones) motion_prob of the P pairs and ``torch.randint`` among the top k (:300-313).
        Neither depends on the model's output, so both tables are drawn before the rollout and handed to the engine
        (ps_set_mode_choice, ps_set_action_noise).  Returns the mode table ([R, B, N] or None when TOP_K = 1) and keeps the
        noise table ([R, B, N, K, S, 2] or None) in ``self._last_action_noise`` (set on the engine here)."""
        spec, scene = self.spec, self._shared.scene
        k = min(spec.rollout_top_k, spec.motion_k)
        std = float(spec.action_noise_std)
        if k <= 1 and std <= 0:
            self._last_action_noise = None
            self.engine.set_action_noise(None)
            return None
        B, N = scene["prompt_mask"].shape
        pslots = scene["_policy_slots"]
        P = sum(len(p) for p in pslots)
        choice = np.zeros((spec.n_replans, B, N), np.int32)
        noise = np.zeros((spec.n_replans, B, N, spec.motion_k, spec.target_steps, 2), np.float32) if std > 0 else None
        for t in range(spec.n_replans):
            if std > 0:   # (the same call on a tensor of the same shape and strides: motion[..., :2] of a [P, K, S, D] view)
                nz = (torch.randn_like(torch.empty(P, spec.motion_k, spec.target_steps, spec.state_dim)[..., :2]) * std).numpy()
            _, top = torch.topk(torch.ones(P, spec.motion_k), k, dim=1)
            rnd = torch.randint(0, k, (P,))
            pick = top[torch.arange(P), rnd].numpy()
            i = 0
            for b in range(B):
                for n in pslots[b]:
                    choice[t, b, n] = pick[i]
                    if std > 0:
                        noise[t, b, n] = nz[i]
                    i += 1
        self._last_action_noise = noise
        self.engine.set_action_noise(noise)
        return choice if k > 1 else None

    def decode_batch(self, scene_embs, prompt_encs, batch, mode="val"):
        """``ProSim.decode_batch`` (:105-116)."""
        extras = batch.extras if hasattr(batch, "extras") else batch
        policy_emds = self.generate_policy(batch, scene_embs, prompt_encs)
        policy_agent_ids = {task: _g(extras["prompt"][task], "agent_ids") for task in self.tasks}
        all_t = sorted(int(t) for t in np.asarray(_np(extras["all_t_indices"], np.int64)).tolist()) if "all_t_indices" in extras \
            else list(self.spec.all_t_indices)
        agent_trajs = self.init_agent_trajs(policy_agent_ids, batch)
        return self.rollout_batch(batch, scene_embs, policy_emds, policy_agent_ids, agent_trajs, all_t, mode)

    def _process_rollout(self, extras, scene) -> Dict[str, Any]:
        """``ProSim._process_rollout`` (:562-595): outputs of the policy agents, in prompt order."""
        spec, eng = self.spec, self.engine
        B, N = scene["prompt_mask"].shape
        if "_policy_slots" not in scene:
            raise ValueError("rollout: the scene has no policy agents yet -- after update_scene_emb with another agent set, "
                             "call the decoder / generate_policy again before rolling out")
        pslots = scene["_policy_slots"]                          # per scene: observation slot of every policy agent, prompt order
        ids = _g(extras["prompt"]["motion_pred"], "agent_ids")
        if ids is None:
            ids = [[str(j) for j in range(len(pslots[b]))] for b in range(B)]
        traj, vel = eng.padded("traj"), eng.padded("vel")
        mp_rows = eng.get("motion_pred")                           # [R, rows = observed agents, K, S, D]
        rec_rows = eng.get("reconst_pred") if spec.use_goal_pred_loss else None
        slot_of_row = eng._slots                                   # flat slot b * N + n of every agent row
        row_of_slot = {int(sl): i for i, sl in enumerate(slot_of_row)}
        order = [row_of_slot[b * N + n] for b in range(B) for n in pslots[b]]   # policy agents, scene-major, prompt order
        mp = torch.from_numpy(mp_rows[:, order])
        R, A = mp.shape[0], mp.shape[1]
        names = [f"{b}-{ids[b][j]}-{t}" for t in spec.all_t_indices for b in range(B) for j in range(len(pslots[b]))]
        out = {"motion_pred": mp.reshape(R * A, *mp.shape[2:]), "motion_prob": torch.ones(R * A, mp.shape[2]),
               "pair_names": names, "rollout_trajs": {}}
        if rec_rows is not None:
            out["reconst_pred"] = torch.from_numpy(rec_rows[order]).repeat(R, 1)
        for b in range(B):
            for j, n in enumerate(pslots[b]):
                r_ = dict(traj=torch.from_numpy(traj[b, n]), init_pos=torch.from_numpy(scene["obs_pos"][b, n]),
                          init_heading=torch.from_numpy(scene["obs_head"][b, n:n + 1]))
                if spec.pred_vel:   # (traj_sam.py:592-593)
                    r_["vel"] = torch.from_numpy(vel[b, n])
                out["rollout_trajs"][f"{b}-{ids[b][j]}"] = r_
        return {"motion_pred": out}
