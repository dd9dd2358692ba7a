"""ctypes binding of libprosim_hip.so (C ABI: include/prosim_hip.h).

Thin by design: numpy arrays in, numpy arrays out, no torch types cross the boundary.  There is
no CPU fallback -- if the HIP library is missing or no GPU is visible this module raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

import numpy as np
import torch

from .spec import ModelSpec, V2V_TAGS

_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libprosim_hip.so")
_lib = None


class PsConfig(C.Structure):
    _fields_ = [
        ("hidden", C.c_int32), ("heads", C.c_int32), ("head_dim", C.c_int32),
        ("scene_layers", C.c_int32), ("scene_knn", C.c_int32), ("agent_knn", C.c_int32),
        ("dec_layers", C.c_int32), ("dec_max_neigh", C.c_int32), ("goal_pred_k", C.c_int32),
        ("dec_prompt_radius", C.c_float), ("dec_scene_radius", C.c_float),
        ("pol_layers", C.c_int32), ("pol_max_neigh", C.c_int32),
        ("pol_agent_radius", C.c_float), ("pol_map_radius", C.c_float),
        ("cond_layers", C.c_int32),
        ("drag_pre_layers", C.c_int32), ("drag_mlp_layers", C.c_int32),
        ("obs_fusion_mlp", C.c_int32), ("obs_attn_update", C.c_int32),
        ("enc_agent_radius", C.c_float), ("enc_scene_radius", C.c_float),
        ("hist_steps", C.c_int32), ("obs_dim", C.c_int32), ("map_dim", C.c_int32),
        ("map_pre_layers", C.c_int32), ("map_mlp_layers", C.c_int32),
        ("obs_pre_layers", C.c_int32), ("obs_mlp_layers", C.c_int32),
        ("target_steps", C.c_int32), ("state_dim", C.c_int32), ("motion_k", C.c_int32),
        ("num_agent_types", C.c_int32), ("prompt_dim", C.c_int32),
        ("replan_freq", C.c_int32), ("max_steps", C.c_int32),
        ("dt", C.c_float), ("ln_eps", C.c_float),
        ("device", C.c_int32),
        ("enc_learnable_pe", C.c_int32), ("dec_learnable_pe", C.c_int32), ("pol_learnable_pe", C.c_int32),
        ("pe_num_freq", C.c_int32), ("v2v_tag_mask", C.c_int32), ("pred_gmm", C.c_int32), ("k_pred_mlp", C.c_int32),
        ("no_pred_vel", C.c_int32), ("no_reconst_pred", C.c_int32), ("rel_pos_knn", C.c_int32),
        ("map_encoder_mlp", C.c_int32), ("obs_encoder_mlp", C.c_int32),
    ]


def lib_path() -> str:
    return _LIB_PATH


def load_library():
    """dlopen the engine.  Raises (never falls back) when the library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("PS_LIB", _LIB_PATH)   # (tools: A/B against another build of the same ABI)
    if not os.path.exists(path):
        raise RuntimeError(f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback for the rollout path)")
    lib = C.CDLL(path)
    fp, u8p, i32p = C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.POINTER(C.c_int32)
    vp = C.c_void_p
    lib.ps_last_error.restype = C.c_char_p
    lib.ps_create.argtypes = [C.POINTER(PsConfig), C.c_int32, C.POINTER(C.c_char_p), C.POINTER(fp), C.POINTER(C.c_int64), C.POINTER(vp)]
    lib.ps_destroy.argtypes = [vp]
    lib.ps_destroy.restype = None
    lib.ps_set_scene.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, fp, u8p, fp, fp, fp, u8p, fp, fp, fp, u8p, i32p, fp, fp]
    lib.ps_set_prompt.argtypes = [vp, fp, fp, fp, i32p]
    lib.ps_policy_forward.argtypes = [vp, C.c_int32, C.c_int32, fp, fp, fp, i32p, C.c_int32, fp, fp, fp, i32p, C.c_int32,
                                      fp, fp, fp, i32p, i32p, fp, fp]
    lib.ps_set_conditions.argtypes = [vp, C.c_int32, fp, u8p, i32p, C.c_int32, fp, u8p, i32p]
    lib.ps_set_drag_points.argtypes = [vp, C.c_int32, C.c_int32, fp, u8p, i32p]
    lib.ps_set_pair_conditions.argtypes = [vp, C.c_int32, fp, u8p, i32p]
    lib.ps_set_future_obs.argtypes = [vp, fp]
    lib.ps_set_future_log.argtypes = [vp, fp, u8p, fp, fp]
    lib.ps_set_mode_choice.argtypes = [vp, i32p]
    lib.ps_set_action_noise.argtypes = [vp, fp]
    lib.ps_set_replicas.argtypes = [vp, C.c_int32]
    lib.ps_num_replicas.argtypes = [vp]
    lib.ps_num_replicas.restype = C.c_int32
    lib.ps_world_trajs.argtypes = [vp, fp, vp]
    lib.ps_num_policy_agents.argtypes = [vp]
    lib.ps_num_policy_agents.restype = C.c_int32
    lib.ps_update_obs.argtypes = [vp, fp, u8p, fp, fp]
    lib.ps_set_map_tokens.argtypes = [vp, fp, C.c_int64]
    lib.ps_declare_agent_rows.argtypes = [vp, C.c_int32, C.c_int32, u8p]
    lib.ps_set_chain_rows.argtypes = [vp, C.c_int32]
    lib.ps_set_chain_impl.argtypes = [vp, C.c_int32]
    lib.ps_set_row_impl.argtypes = [vp, C.c_int32]
    lib.ps_set_search_impl.argtypes = [vp, C.c_int32]
    lib.ps_graph_nodes.argtypes = [vp]
    lib.ps_graph_nodes.restype = C.c_int64
    lib.ps_enable_policy_events.argtypes = [vp, C.c_int32]
    lib.ps_policy_event_times.argtypes = [vp, fp, C.c_int32]
    lib.ps_stream.argtypes = [vp]
    lib.ps_stream.restype = C.c_void_p
    lib.ps_policy_flags.argtypes = [vp, i32p, C.c_int64]
    for name in ("ps_encode_scene", "ps_generate_policy", "ps_reset_rollout", "ps_rollout", "ps_sync"):
        getattr(lib, name).argtypes = [vp]
    lib.ps_policy_step.argtypes = [vp, C.c_int32]
    lib.ps_set_state.argtypes = [vp, C.c_int32, fp, fp]
    lib.ps_get.argtypes = [vp, C.c_char_p, fp, C.c_int64]
    lib.ps_get.restype = C.c_int64
    lib.ps_get_async.argtypes = [vp, C.c_char_p, vp, C.c_int64]
    lib.ps_get_async.restype = C.c_int64
    lib.ps_graph_stats.argtypes = [vp, C.POINTER(C.c_int64)]
    lib.ps_rollout_metric.argtypes = [vp, vp, vp]
    lib.ps_pair_metric.argtypes = [vp, vp, vp, vp, vp]
    lib.ps_num_agents.argtypes = [vp]
    lib.ps_num_map_tokens.argtypes = [vp]
    lib.ps_time_rollout.argtypes = [vp, C.c_int32, C.c_int32, fp, fp]
    lib.ps_time_policy_kernel.argtypes = [vp, C.c_int32, fp]
    lib.ps_test_pointnet.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, fp, u8p, fp]
    lib.ps_test_pointnet_mt.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, fp, u8p, fp, C.c_int32, C.c_int32, fp]
    lib.ps_test_fourier.argtypes = [vp, C.c_int32, fp, fp]
    lib.ps_test_wrap.argtypes = [vp, C.c_int32, fp, fp]
    lib.ps_test_attn.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, fp, fp, fp, i32p, i32p, C.c_int32, fp]
    lib.ps_test_get_edges.argtypes = [vp, C.c_int32, i32p, i32p, fp, C.c_int64]
    lib.ps_test_get_edges.restype = C.c_int64
    lib.ps_test_stream.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, fp]
    _lib = lib
    return lib


EXPORTS = ["ps_create", "ps_destroy", "ps_last_error", "ps_set_scene", "ps_set_prompt", "ps_policy_forward", "ps_set_conditions", "ps_set_drag_points", "ps_set_pair_conditions", "ps_set_future_obs", "ps_set_future_log", "ps_set_mode_choice", "ps_set_action_noise", "ps_set_replicas", "ps_num_replicas", "ps_world_trajs", "ps_num_policy_agents", "ps_policy_flags",
           "ps_encode_scene", "ps_generate_policy", "ps_reset_rollout", "ps_policy_step", "ps_rollout", "ps_sync", "ps_stream", "ps_set_chain_rows", "ps_set_chain_impl", "ps_set_row_impl", "ps_set_search_impl", "ps_graph_nodes", "ps_enable_policy_events", "ps_policy_event_times", "ps_update_obs", "ps_set_map_tokens", "ps_declare_agent_rows",
           "ps_set_state", "ps_get", "ps_get_async", "ps_graph_stats", "ps_rollout_metric", "ps_pair_metric", "ps_num_agents", "ps_num_map_tokens", "ps_time_rollout", "ps_time_policy_kernel",
           "ps_test_pointnet", "ps_test_pointnet_mt", "ps_test_fourier", "ps_test_wrap", "ps_test_attn", "ps_test_get_edges", "ps_test_stream"]


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _u8(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def _i32(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def fourier_tables() -> Dict[str, np.ndarray]:
    """dim_t of FourierEmbeddingFix for 32/64/128 slots, computed with the reference's own torch
    ops (models/layers/fourier_embedding.py:68-69) so the divisors are bit-identical."""
    out = {}
    for n in (32.0, 64, 128):
        dim_t = torch.arange(n, dtype=torch.float32)
        out[f"const.fourier_div{int(n)}"] = (10000 ** (2 * (dim_t // 2) / n)).numpy().astype(np.float32)
    return out


class Engine:
    """One engine per GPU (``device`` = HIP ordinal).  Not thread-safe."""

    def __init__(self, spec: ModelSpec, weights: Dict[str, np.ndarray], device: int = 0):
        self.lib = load_library()
        self.spec = spec
        if spec.obs_fusion not in ("replace", "mlp"):
            raise ValueError(f"MODEL.OBS_UPDATE.FUSION must be 'replace' or 'mlp', got {spec.obs_fusion!r}")
        cfg = PsConfig(hidden=spec.hidden, heads=spec.heads, head_dim=spec.head_dim, scene_layers=spec.scene_layers,
                       scene_knn=spec.scene_knn, agent_knn=spec.agent_knn, dec_layers=spec.dec_layers,
                       dec_max_neigh=spec.dec_max_neigh, goal_pred_k=spec.goal_pred_k, dec_prompt_radius=spec.dec_prompt_radius,
                       dec_scene_radius=spec.dec_scene_radius, pol_layers=spec.pol_layers, pol_max_neigh=spec.pol_max_neigh,
                       pol_agent_radius=spec.pol_agent_radius, pol_map_radius=spec.pol_map_radius, cond_layers=spec.cond_layers,
                       drag_pre_layers=spec.drag_pre_layers, drag_mlp_layers=spec.drag_mlp_layers,
                       obs_fusion_mlp=int(spec.obs_fusion == "mlp"), obs_attn_update=int(bool(spec.obs_attn_update)),
                       enc_agent_radius=spec.enc_agent_radius, enc_scene_radius=spec.enc_scene_radius,
                       hist_steps=spec.hist_steps, obs_dim=spec.obs_dim, map_dim=spec.map_dim,
                       map_pre_layers=spec.map_pre_layers, map_mlp_layers=spec.map_mlp_layers,
                       obs_pre_layers=spec.obs_pre_layers, obs_mlp_layers=spec.obs_mlp_layers,
                       target_steps=spec.target_steps, state_dim=spec.state_dim, motion_k=spec.motion_k,
                       num_agent_types=spec.num_agent_types, prompt_dim=spec.prompt_dim, replan_freq=spec.replan_freq,
                       max_steps=spec.max_steps, dt=spec.dt, ln_eps=spec.ln_eps, device=device,
                       enc_learnable_pe=int(spec.enc_learnable_pe), dec_learnable_pe=int(spec.dec_learnable_pe),
                       pol_learnable_pe=int(spec.pol_learnable_pe),
                       pe_num_freq=64 if spec.pe_num_freq <= 64 else spec.pe_num_freq,   # (fewer bands: zero-padded by weights.engine_tensors)
                       v2v_tag_mask=sum(1 << V2V_TAGS.index(t) for t in spec.used_v2v_tags), pred_gmm=int(spec.pred_gmm),
                       k_pred_mlp=int(spec.k_pred_mode == "mlp"), no_pred_vel=int(not spec.pred_vel),
                       no_reconst_pred=int(not spec.use_goal_pred_loss), rel_pos_knn=int(spec.rel_pos_edge_func == "knn"),
                       map_encoder_mlp=int(spec.map_encoder_type == "mlp"), obs_encoder_mlp=int(spec.obs_encoder_type == "mlp"))
        from .weights import engine_tensors
        tensors = dict(engine_tensors(spec, weights))   # ('cluster' anchors folded into the anchor table)
        tensors.update(fourier_tables())
        names = sorted(tensors)
        arrs = [np.ascontiguousarray(tensors[n], dtype=np.float32) for n in names]
        c_names = (C.c_char_p * len(names))(*[n.encode() for n in names])
        c_ptrs = (C.POINTER(C.c_float) * len(names))(*[_f(a) for a in arrs])
        c_numel = (C.c_int64 * len(names))(*[a.size for a in arrs])
        h = C.c_void_p()
        rc = self.lib.ps_create(C.byref(cfg), len(names), c_names, c_ptrs, c_numel, C.byref(h))
        self.h = h if rc == 0 else None
        self._check(rc)
        self._shape = self._in_shape = None
        self._slots = None

    def _check(self, rc: int):
        if rc != 0:
            raise RuntimeError(f"libprosim_hip error {rc}: {self.lib.ps_last_error().decode()}")

    def close(self):
        if getattr(self, "h", None):
            self.lib.ps_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- scene
    def set_scene(self, s: Dict[str, np.ndarray]):
        f32 = lambda k: np.ascontiguousarray(s[k], dtype=np.float32)
        msk = lambda k: np.ascontiguousarray(s[k]).astype(np.uint8)
        map_input, obs_input, prompt = f32("map_input"), f32("obs_input"), f32("prompt")
        B, M, P, _ = map_input.shape
        N = obs_input.shape[1]
        map_mask, obs_mask, pm = msk("map_mask"), msk("obs_mask"), msk("prompt_mask")
        at = np.ascontiguousarray(s["agent_type"], dtype=np.int32)
        ppos = f32("prompt_pos") if "prompt_pos" in s else f32("obs_pos")
        phead = f32("prompt_head") if "prompt_head" in s else f32("obs_head")
        keep = [map_input, obs_input, prompt, map_mask, obs_mask, pm, at, ppos, phead, f32("map_pos"), f32("map_head"),
                f32("obs_pos"), f32("obs_head")]
        # agent rows = agents observed at the initial step (a history step with every feature valid) plus agents that
        # only ENTER with a later fut_obs frame (declared to the engine: token rows that are not in the scene yet)
        seen0 = obs_mask.astype(bool).all(-1).any(-1)
        seen = seen0.copy()
        if s.get("fut_obs_mask") is not None:
            seen |= np.asarray(s["fut_obs_mask"]).astype(bool).all(-1).any(-1).any(0)
        if (seen & ~seen0).any():
            rows = np.ascontiguousarray(seen).astype(np.uint8)
            self._check(self.lib.ps_declare_agent_rows(self.h, B, N, _u8(rows)))
        else:
            self._check(self.lib.ps_declare_agent_rows(self.h, 0, 0, None))
        self._check(self.lib.ps_set_scene(self.h, B, M, P, N, _f(map_input), _u8(map_mask), _f(keep[9]), _f(keep[10]),
                                          _f(obs_input), _u8(obs_mask), _f(keep[11]), _f(keep[12]), _f(prompt), _u8(pm),
                                          _i32(at), _f(ppos), _f(phead)))
        Mrep = self.replicas
        self._shape = (B, N) if Mrep == 1 else (Mrep, N)   # layout of the per-agent RESULTS (replica-major with replicas)
        self._in_shape = (B, N)                              # layout of the per-scene INPUTS (replicas share one scene)
        # rows in slot order; policy agents are the rows whose slot carries a prompt, the others replay the log
        # (ps_set_future_log); live0_rows: the row is a scene token at the initial step
        in_slots = np.nonzero(seen.reshape(-1))[0]
        self.policy_rows = np.tile(pm.reshape(-1).astype(bool)[in_slots], Mrep)
        self.live0_rows = np.tile(seen0.reshape(-1)[in_slots], Mrep)
        # with replicas the rows are replica-major and slot r * N + n is agent n of replica r
        self._slots = in_slots if Mrep == 1 else (np.arange(Mrep)[:, None] * N + in_slots[None, :]).reshape(-1)
        self.set_conditions(s.get("cond"), _fresh_scene=True)
        if s.get("mode_choice") is not None:
            self.set_mode_choice(s["mode_choice"])
        if s.get("action_noise") is not None:
            self.set_action_noise(s["action_noise"])
        if s.get("fut_obs_input") is not None:
            fo = np.ascontiguousarray(s["fut_obs_input"], dtype=np.float32)
            if s.get("fut_obs_mask") is not None and s.get("fut_obs_pos") is not None and s.get("fut_obs_head") is not None:
                fm = np.ascontiguousarray(s["fut_obs_mask"]).astype(np.uint8)
                fp_ = np.ascontiguousarray(s["fut_obs_pos"], dtype=np.float32)
                fh = np.ascontiguousarray(s["fut_obs_head"], dtype=np.float32)
                self._check(self.lib.ps_set_future_log(self.h, _f(fo), _u8(fm), _f(fp_), _f(fh)))
            else:
                self._check(self.lib.ps_set_future_obs(self.h, _f(fo)))

    @property
    def replicas(self) -> int:
        return int(getattr(self, "_replicas", 1))

    def set_replicas(self, m: int):
        """Roll the next scenes out as ``m`` replicas side by side (parallel_rollout_batch, rollout/gpu_utils.py:179-228):
        ``set_scene`` then takes a ONE-scene batch; encode / generate run once and fan out on the device; every per-agent
        result has ``m * agents`` rows (``padded`` returns [m, N, ...]); ``mode_choice`` is [R, m, N]."""
        self._check(self.lib.ps_set_replicas(self.h, int(m)))
        self._replicas = int(m)
        self._shape = self._in_shape = self._slots = None

    def world_trajs(self, center_to_world=None, out_dev_ptr: int = 0):
        """obtain_rollout_trajs_in_world (rollout/gpu_utils.py:230-281) on the device.  ``center_to_world``: 3 x 3 or None
        (identity).  With ``out_dev_ptr`` (a device buffer [A, max_steps, 3] float32) nothing is returned; without, the
        result is read back: [A, max_steps, 3] (x, y, heading)."""
        tf = None if center_to_world is None else np.ascontiguousarray(center_to_world, dtype=np.float32).reshape(9)
        self._check(self.lib.ps_world_trajs(self.h, None if tf is None else _f(tf), C.c_void_p(out_dev_ptr or None)))
        if out_dev_ptr:
            return None
        return self.get("world_traj")

    def set_mode_choice(self, choice):
        """``choice`` [R, B, N] int ([R, replicas, N] with replicas): the motion mode each policy agent follows at each
        replan (TOP_K > 1), or None."""
        if choice is None:
            self._check(self.lib.ps_set_mode_choice(self.h, None))
            return
        c = np.ascontiguousarray(choice, dtype=np.int32)
        if c.shape != (self.spec.n_replans,) + tuple(self._shape):
            raise ValueError(f"mode choice must be [R, B, N] = {(self.spec.n_replans,) + tuple(self._shape)}, got {c.shape}")
        self._check(self.lib.ps_set_mode_choice(self.h, _i32(c)))

    def set_action_noise(self, noise):
        """``noise`` [R, B, N, K, target_steps, 2] float ([R, replicas, N, ...] with replicas), already scaled by
        RANDOM_NOISE_STD: added to every predicted xy step of the policy agents before the cumulative sum
        (act_decoder.py:113-115); None switches it off."""
        if noise is None:
            self._check(self.lib.ps_set_action_noise(self.h, None))
            return
        a = np.ascontiguousarray(noise, dtype=np.float32)
        want = (self.spec.n_replans,) + tuple(self._shape) + (self.spec.motion_k, self.spec.target_steps, 2)
        if a.shape != want:
            raise ValueError(f"action noise must be {want}, got {a.shape}")
        self._check(self.lib.ps_set_action_noise(self.h, _f(a)))

    def set_conditions(self, cond, _fresh_scene: bool = False):
        """``cond`` = {'goal' | 'v_action_tag' | 'drag_point': {'input', 'mask', 'prompt_idx' [B,C,1] = prompt SLOT},
        'v2v_tag': {'input' [B,C,3], 'mask', 'prompt_idx' [B,C,2] = the SLOTS of (source, target)}} or None; replaces every
        condition of the uploaded batch (batch.extras['condition'] after the id -> slot mapping)."""
        cond = cond or {}
        unknown = [k for k in cond if k not in ("goal", "v_action_tag", "drag_point", "v2v_tag")]
        if unknown:
            raise NotImplementedError(f"condition types {unknown}: goal, v_action_tag, drag_point and v2v_tag are built")
        args, keep = [], []
        for c in (cond.get("goal"), cond.get("v_action_tag")):
            if c is None or np.asarray(c["input"]).shape[1] == 0:
                args += [0, None, None, None]
            else:
                ci = np.ascontiguousarray(c["input"], dtype=np.float32)
                cm = np.ascontiguousarray(c["mask"]).astype(np.uint8)
                cp = np.ascontiguousarray(np.asarray(c["prompt_idx"])[..., 0], dtype=np.int32)
                keep += [ci, cm, cp]
                args += [ci.shape[1], _f(ci), _u8(cm), _i32(cp)]
        # (ps_set_scene has just cleared every condition: a type the batch does not carry needs no call -- each one rebuilds and
        # uploads the condition graph)
        absent = lambda c: c is None or np.asarray(c["input"]).shape[1] == 0
        if not (_fresh_scene and absent(cond.get("goal")) and absent(cond.get("v_action_tag"))):
            self._check(self.lib.ps_set_conditions(self.h, *args))
        if not (_fresh_scene and absent(cond.get("drag_point"))):
            self.set_drag_points(cond.get("drag_point"))
        if not (_fresh_scene and absent(cond.get("v2v_tag"))):
            self.set_pair_conditions(cond.get("v2v_tag"))

    def set_pair_conditions(self, c):
        """``c`` = {'input' [B,C,3] (V2V tag value, t0, t1), 'mask' [B,C], 'prompt_idx' [B,C,2] (source, target slots)} or None."""
        if c is None or np.asarray(c["input"]).shape[1] == 0:
            self._check(self.lib.ps_set_pair_conditions(self.h, 0, None, None, None))
            return
        ci = np.ascontiguousarray(c["input"], dtype=np.float32)
        cm = np.ascontiguousarray(c["mask"]).astype(np.uint8)
        cp = np.ascontiguousarray(np.asarray(c["prompt_idx"]), dtype=np.int32)
        if cp.shape != ci.shape[:2] + (2,):
            raise ValueError("v2v_tag prompt_idx must be [B, C, 2]")
        self._check(self.lib.ps_set_pair_conditions(self.h, ci.shape[1], _f(ci), _u8(cm), _i32(cp)))

    def set_drag_points(self, c):
        """``c`` = {'input' [B,C,T,2] (NaN = absent point), 'mask' [B,C], 'prompt_idx' [B,C,1]} or None (clears)."""
        if c is None or np.asarray(c["input"]).shape[1] == 0:
            self._check(self.lib.ps_set_drag_points(self.h, 0, 0, None, None, None))
            return
        ci = np.ascontiguousarray(c["input"], dtype=np.float32)
        cm = np.ascontiguousarray(c["mask"]).astype(np.uint8)
        cp = np.ascontiguousarray(np.asarray(c["prompt_idx"])[..., 0], dtype=np.int32)
        self._check(self.lib.ps_set_drag_points(self.h, ci.shape[1], ci.shape[2], _f(ci), _u8(cm), _i32(cp)))

    def set_prompt(self, prompt, prompt_pos, prompt_head, agent_type):
        a = [np.ascontiguousarray(prompt, np.float32), np.ascontiguousarray(prompt_pos, np.float32),
             np.ascontiguousarray(prompt_head, np.float32).reshape(self._in_shape), np.ascontiguousarray(agent_type, np.int32)]
        self._check(self.lib.ps_set_prompt(self.h, _f(a[0]), _f(a[1]), _f(a[2]), _i32(a[3])))

    def policy_forward(self, n_scenes, a_tok, a_pos, a_ori, a_scene, m_tok, m_pos, m_ori, m_scene, p_emd, p_pos, p_ori,
                       p_type, p_scene):
        """Stateless policy.forward on explicit (valid, scene-major) tokens -> (motion_pred [A,K,S,D], fused [A,128])."""
        f = lambda x, shp: np.ascontiguousarray(x, np.float32).reshape(shp)
        i = lambda x: np.ascontiguousarray(x, np.int32).reshape(-1)
        Na, Nm, A = len(i(a_scene)), len(i(m_scene)), len(i(p_scene))
        arrs = [f(a_tok, (Na, 128)), f(a_pos, (Na, 2)), f(a_ori, (Na,)), i(a_scene), f(m_tok, (Nm, 128)), f(m_pos, (Nm, 2)),
                f(m_ori, (Nm,)), i(m_scene), f(p_emd, (A, 128)), f(p_pos, (A, 2)), f(p_ori, (A,)), i(p_type), i(p_scene)]
        sp = self.spec
        mp = np.empty((A, sp.motion_k, sp.target_steps, sp.state_dim), np.float32)
        fused = np.empty((A, 128), np.float32)
        P = lambda a: _f(a) if a.dtype == np.float32 else _i32(a)
        self._check(self.lib.ps_policy_forward(self.h, n_scenes, Na, P(arrs[0]), P(arrs[1]), P(arrs[2]), P(arrs[3]), Nm, P(arrs[4]),
                                               P(arrs[5]), P(arrs[6]), P(arrs[7]), A, P(arrs[8]), P(arrs[9]), P(arrs[10]),
                                               P(arrs[11]), P(arrs[12]), _f(mp), _f(fused)))
        return mp, fused

    # ---- stages (each enqueues on the engine's stream; results are read with get())
    def encode_scene(self):
        self._check(self.lib.ps_encode_scene(self.h))

    def generate_policy(self):
        self._check(self.lib.ps_generate_policy(self.h))

    def reset_rollout(self):
        self._check(self.lib.ps_reset_rollout(self.h))

    def update_obs(self, obs_input, obs_mask, obs_pos, obs_head):
        """Re-encode the agents from a new observation ([B,N,hist,obs_dim], mask, [B,N,2], [B,N]); map tokens are kept."""
        a = [np.ascontiguousarray(obs_input, np.float32), np.ascontiguousarray(obs_mask).astype(np.uint8),
             np.ascontiguousarray(obs_pos, np.float32), np.ascontiguousarray(obs_head, np.float32).reshape(self._in_shape)]
        if a[0].shape[:2] != self._in_shape or a[1].shape != a[0].shape:
            raise ValueError("update_obs: the observation must keep the [B, N] layout of set_scene")
        self._check(self.lib.ps_update_obs(self.h, _f(a[0]), _u8(a[1]), _f(a[2]), _f(a[3])))

    def set_map_tokens(self, tokens: np.ndarray):
        """Hand the map tokens of an encoded scene back after a ``set_scene`` with the same map and another agent set
        (update_scene_emb with a changing agent set); the agents are then re-encoded with ``update_obs``."""
        t = np.ascontiguousarray(tokens, np.float32)
        self._check(self.lib.ps_set_map_tokens(self.h, _f(t), t.size))

    def policy_step(self, t_idx: int):
        self._check(self.lib.ps_policy_step(self.h, t_idx))

    def rollout(self):
        self._check(self.lib.ps_rollout(self.h))

    def sync(self):
        self._check(self.lib.ps_sync(self.h))

    def set_chain_rows(self, rows: int):
        """0: latency mode (one rollout on the GPU); 8..16: throughput mode for several engines sharing the GPU (16 in bench.py)."""
        self._check(self.lib.ps_set_chain_rows(self.h, rows))

    def set_chain_impl(self, impl: int):
        """0: by mode (default); 1: k_attn_chain always; 2: k_chain16 always (A/B measurements, cross-checks); 3: k_chain16
        always and for the scene encoder's s2s layers as well (faster; another fp32 evaluation order of the scene tokens)."""
        self._check(self.lib.ps_set_chain_impl(self.h, impl))

    @property
    def graph_nodes(self) -> int:
        """Launches + copies in the captured rollout graph (0 before the first rollout)."""
        return int(self.lib.ps_graph_nodes(self.h))

    def set_row_impl(self, impl: int):
        """0: the row-tile kernels (default); 1: the staged row kernels of rounds 1-3 (cross-checks, A/B measurements); 2: as 0 with
        the workgroup edge kernel (k_edge16) in the split layers; 11..13: as 0 with 1..3 row tiles per wave forced (2, 11..13: same bits as 0)."""
        self._check(self.lib.ps_set_row_impl(self.h, impl))

    def set_search_impl(self, impl: int):
        """0: a radius search whose edges feed a geometry-record chain is ONE launch (k_radius_geo, default); 1: the count / fill /
        record launches of rounds 1-4 (cross-checks, A/B measurements); 2: as 0 with a look-back that recomputes unpublished counts instead of
        waiting for them (the no-forward-progress fallback, forced: tests) -- same bits."""
        self._check(self.lib.ps_set_search_impl(self.h, impl))

    @property
    def stream_handle(self) -> int:
        """The engine's hipStream_t as an integer (``torch.cuda.ExternalStream(handle)``)."""
        return int(self.lib.ps_stream(self.h) or 0)

    def set_state(self, traj: np.ndarray, vel: np.ndarray):
        traj = np.ascontiguousarray(traj, dtype=np.float32)
        vel = np.ascontiguousarray(vel, dtype=np.float32)
        self._check(self.lib.ps_set_state(self.h, traj.shape[1], _f(traj), _f(vel)))

    @property
    def num_agents(self) -> int:
        """Agent rows = observed agents (policy agents and log-replay agents)."""
        return self.lib.ps_num_agents(self.h)

    @property
    def num_policy_agents(self) -> int:
        return self.lib.ps_num_policy_agents(self.h)

    @property
    def num_map_tokens(self) -> int:
        return self.lib.ps_num_map_tokens(self.h)

    def get(self, name: str) -> np.ndarray:
        sp, A, Mv = self.spec, self.num_agents, self.num_map_tokens
        R, S = sp.n_replans, sp.n_replans * sp.replan_freq
        shapes = {"traj": (A, S, 4), "vel": (A, S, 2), "motion_pred": (R, A, sp.motion_k, sp.target_steps, sp.state_dim),
                  "reconst_pred": (A, 2), "policy_emd": (A, sp.hidden), "scene_tokens": (Mv + A, sp.hidden),
                  "fused": (A, sp.hidden), "obs_in": (A, sp.hist_steps, sp.obs_dim), "cur_pos": (A, 2), "edge_counts": (8,),
                  "goal_prob": (A, max(sp.goal_pred_k, 1)), "goal_point": (A, max(sp.goal_pred_k, 1), 2), "world_traj": (A, S, 3)}
        out = np.empty(shapes[name], np.float32)
        n = self.lib.ps_get(self.h, name.encode(), _f(out), out.size)
        if n < 0:
            self._check(int(n))
        return out

    ASYNC_RESULTS = ("traj", "vel", "motion_pred", "reconst_pred", "policy_emd", "fused", "goal_prob", "goal_point")

    def result_shape(self, name: str):
        sp, A = self.spec, self.num_agents
        R, S = sp.n_replans, sp.n_replans * sp.replan_freq
        return {"traj": (A, S, 4), "vel": (A, S, 2), "motion_pred": (R, A, sp.motion_k, sp.target_steps, sp.state_dim),
                "reconst_pred": (A, 2), "policy_emd": (A, sp.hidden), "fused": (A, sp.hidden),
                "goal_prob": (A, max(sp.goal_pred_k, 1)), "goal_point": (A, max(sp.goal_pred_k, 1), 2)}[name]

    def get_async(self, name: str, host_ptr: int, capacity: int) -> int:
        """``ps_get`` without its synchronisations: enqueue the copy of a per-agent result behind the work already on the engine's
        stream (call right after ``rollout()``) into caller-owned memory of ``capacity`` floats at ``host_ptr`` -- pinned memory
        (``torch.empty(n, pin_memory=True).data_ptr()``) makes the call return at once.  Wait for the stream (``sync()``) or for an
        event recorded on ``stream_handle`` before reading.  Returns the number of floats that will arrive."""
        n = self.lib.ps_get_async(self.h, name.encode(), C.c_void_p(host_ptr), capacity)
        if n < 0:
            self._check(int(n))
        return int(n)

    def graph_stats(self):
        """(rollouts that captured a graph, rollouts after a setter that kept the graph they had)."""
        out = (C.c_int64 * 2)()
        self._check(self.lib.ps_graph_stats(self.h, out))
        return int(out[0]), int(out[1])

    def padded(self, name: str) -> np.ndarray:
        """Per-agent result scattered back to the padded [B, N, ...] slot layout of the inputs."""
        a = self.get(name)
        if name in ("traj", "vel", "policy_emd", "reconst_pred", "fused", "goal_prob", "goal_point"):
            a = np.where(self.policy_rows.reshape((-1,) + (1,) * (a.ndim - 1)), a, 0.0).astype(np.float32)   # log-replay rows
        B, N = self._shape
        out = np.zeros((B * N,) + a.shape[1:], np.float32)
        out[self._slots] = a
        return out.reshape((B, N) + a.shape[1:])

    def rollout_metric(self, out_dev_ptr: int, gt_dev_ptr: int = 0):
        """Per-agent (ADE, FDE) into a caller-owned device buffer [A, 2] (pass tensor.data_ptr())."""
        self._check(self.lib.ps_rollout_metric(self.h, C.c_void_p(gt_dev_ptr or None), C.c_void_p(out_dev_ptr)))

    def pair_metric(self, out_dev_ptr: int, tgt_dev_ptr: int, pair_mask_dev_ptr: int, prob_dev_ptr: int = 0):
        """The reference's PairMotionPred sums per agent row into a caller-owned device buffer [A, 10]; tgt
        [R, A, target_steps, 5] float32 and pair_mask [R, A] uint8 are device pointers in agent-row order."""
        self._check(self.lib.ps_pair_metric(self.h, C.c_void_p(tgt_dev_ptr), C.c_void_p(pair_mask_dev_ptr),
                                            C.c_void_p(prob_dev_ptr or None), C.c_void_p(out_dev_ptr)))

    @property
    def row_slots(self) -> np.ndarray:
        """Flat slot index b * N + n of every agent row (the order of all per-agent results)."""
        return self._slots.copy()

    def enable_policy_events(self, on: bool = True):
        """Record an event pair around every policy-chain launch of the following rollouts (see policy_event_times)."""
        self._check(self.lib.ps_enable_policy_events(self.h, int(bool(on))))

    def policy_event_times(self) -> np.ndarray:
        """Durations (ms) of the policy-chain launches of this engine's last rollout, as they ran (pipelined or alone)."""
        out = np.zeros(self.spec.n_replans, np.float32)
        n = self.lib.ps_policy_event_times(self.h, _f(out), out.size)
        if n < 0:
            self._check(int(n))
        return out[:n]

    def time_rollout(self, warmup: int, iters: int):
        ms = C.c_float()
        st = (C.c_float * 3)()
        self._check(self.lib.ps_time_rollout(self.h, warmup, iters, C.byref(ms), st))
        return ms.value, list(st)

    def time_policy_kernel(self, iters: int) -> float:
        ms = C.c_float()
        self._check(self.lib.ps_time_policy_kernel(self.h, iters, C.byref(ms)))
        return ms.value

    # ---- primitive test hooks
    def test_pointnet(self, which: int, x: np.ndarray, mask: np.ndarray) -> np.ndarray:
        x = np.ascontiguousarray(x, np.float32)
        m = np.ascontiguousarray(mask).astype(np.uint8)
        n, P = m.shape
        out = np.empty((n, self.spec.hidden), np.float32)
        self._check(self.lib.ps_test_pointnet(self.h, which, n, P, _f(x), _u8(m), _f(out)))
        return out

    def test_pointnet_mt(self, which: int, x: np.ndarray, mask: np.ndarray, mt: int, iters: int = 0):
        """Test hook: the PointNet with ``mt`` row tiles per wave (0: the engine's choice, -1: the staged round-3 kernel);
        returns (features, ms per launch over ``iters`` launches or None)."""
        x = np.ascontiguousarray(x, np.float32)
        m = np.ascontiguousarray(mask).astype(np.uint8)
        n, P = m.shape
        out = np.empty((n, self.spec.hidden), np.float32)
        ms = C.c_float(0)
        self._check(self.lib.ps_test_pointnet_mt(self.h, which, n, P, _f(x), _u8(m), _f(out), mt, iters, C.byref(ms)))
        return out, (float(ms.value) if iters > 0 else None)

    def test_fourier(self, x4: np.ndarray) -> np.ndarray:
        x4 = np.ascontiguousarray(x4, np.float32)
        out = np.empty((x4.shape[0], 128), np.float32)
        self._check(self.lib.ps_test_fourier(self.h, x4.shape[0], _f(x4), _f(out)))
        return out

    def test_wrap(self, x: np.ndarray) -> np.ndarray:
        x = np.ascontiguousarray(x, np.float32)
        out = np.empty_like(x)
        self._check(self.lib.ps_test_wrap(self.h, x.size, _f(x), _f(out)))
        return out

    def layer_index(self, group: str, i: int) -> int:
        sp = self.spec
        order = [("a2a", sp.scene_layers), ("s2s", sp.scene_layers), ("p2p", sp.dec_layers), ("s2p", sp.dec_layers),
                 ("a2p", sp.pol_layers), ("m2p", sp.pol_layers), ("cond", sp.cond_layers)]
        base = 0
        for g, n in order:
            if g == group:
                return base + i
            base += n
        raise KeyError(group)

    def test_attn(self, layer_index: int, x_src, x_dst, rt, eoff, esrc, T: int = 0) -> np.ndarray:
        x_src, x_dst, rt = (np.ascontiguousarray(a, np.float32) for a in (x_src, x_dst, rt))
        eoff, esrc = np.ascontiguousarray(eoff, np.int32), np.ascontiguousarray(esrc, np.int32)
        out = np.empty_like(x_dst)
        self._check(self.lib.ps_test_attn(self.h, layer_index, x_src.shape[0], x_dst.shape[0], esrc.shape[0], _f(x_src),
                                          _f(x_dst), _f(rt), _i32(eoff), _i32(esrc), T, _f(out)))
        return out

    def get_edges(self, which: int, cap: int = 1 << 20):
        esrc, edst = np.empty(cap, np.int32), np.empty(cap, np.int32)
        rt = np.empty((cap, 128), np.float32)
        n = self.lib.ps_test_get_edges(self.h, which, _i32(esrc), _i32(edst), _f(rt), cap)
        if n < 0:
            self._check(int(n))
        return esrc[:n].copy(), edst[:n].copy(), rt[:n].copy()

    def test_stream(self, mbytes: int, nwg: int, depth: int, iters: int = 5) -> float:
        ms = C.c_float()
        self._check(self.lib.ps_test_stream(self.h, mbytes, nwg, depth, iters, C.byref(ms)))
        return ms.value
