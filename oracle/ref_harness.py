"""ORACLE TOOLING -- build-container only; never imported by the product, never runs on the GPU box.

Imports the reference's own Python from /root/reference so that tests/gen_golden.py can
(a) check oracle/prosim_oracle.py against it and (b) write the committed golden fixtures.

The reference cannot be imported as shipped in this image:
pytorch_lightning, torchmetrics, wandb, peft, yacs, trajdata, torch_cluster and
torch_geometric are absent.  This module registers stand-ins for them in ``sys.modules``:

  * STRUCTURAL stand-ins (no arithmetic): LightningModule = nn.Module, Metric = nn.Module,
    a minimal yacs CfgNode, empty wandb/peft/trajdata namespaces, package shells for
    ``prosim.*`` so the reference's ``__init__`` import chains (dataset -> trajdata) do not run.
  * ARITHMETIC-BEARING stand-ins, written by the builder from torch_cluster's / PyG's
    documented behaviour (NOT the real libraries -- fixtures that depend on them are
    labelled ``ref_standins``): ``torch_cluster.{radius, radius_graph, knn, knn_graph}``
    (brute force; strict d^2 < r^2; first ``max_num_neighbors`` in index order; kNN by
    (distance, index)) and ``torch_geometric`` ``MessagePassing.propagate(aggr='add',
    node_dim=0)`` + ``utils.softmax`` (max-shift, / (sum + 1e-16)).
"""
from __future__ import annotations

import importlib
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import yaml

REF_ROOT = "/root/reference"


def available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "prosim"))


# --------------------------------------------------------------------------- yacs stand-in
class CfgNode(dict):
    def __init__(self, init_dict=None, key_list=None, new_allowed=False):
        super().__init__()
        self.__dict__["_frozen"] = False
        if init_dict:
            for k, v in init_dict.items():
                self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, k):
        if k in self:
            return self[k]
        raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self):
        out = type(self)()
        for k, v in self.items():
            dict.__setitem__(out, k, v.clone() if isinstance(v, CfgNode) else (list(v) if isinstance(v, list) else v))
        return out

    def merge_from_other_cfg(self, other):
        for k, v in other.items():
            if isinstance(v, dict) and k in self and isinstance(self[k], CfgNode):
                self[k].merge_from_other_cfg(v if isinstance(v, CfgNode) else CfgNode(v))
            else:
                self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def merge_from_file(self, path):
        with open(path) as f:
            self.merge_from_other_cfg(CfgNode(yaml.safe_load(f)))

    def merge_from_list(self, lst):
        for k, v in zip(lst[0::2], lst[1::2]):
            node = self
            parts = k.split(".")
            for p in parts[:-1]:
                node = node[p]
            node[parts[-1]] = v

    def freeze(self):
        pass

    def defrost(self):
        pass

    def register_renamed_key(self, *a, **k):
        pass


# --------------------------------------------------------------------------- torch_cluster stand-in
def _d2(x, y):
    dx = x[None, :, 0] - y[:, None, 0]
    dy = x[None, :, 1] - y[:, None, 1]
    return dx * dx + dy * dy


def _radius(x, y, r, batch_x=None, batch_y=None, max_num_neighbors=32, num_workers=1):
    if batch_x is None:
        batch_x = torch.zeros(x.shape[0], dtype=torch.long)
    if batch_y is None:
        batch_y = torch.zeros(y.shape[0], dtype=torch.long)
    d2 = _d2(x.float(), y.float())
    rr = torch.tensor(float(r), dtype=torch.float32) ** 2
    ok = (d2 < rr) & (batch_y[:, None] == batch_x[None, :])
    ok = ok & (torch.cumsum(ok.long(), dim=1) <= max_num_neighbors)
    yi, xi = ok.nonzero(as_tuple=True)
    return torch.stack([yi, xi], dim=0)


def _radius_graph(x, r, batch=None, loop=False, max_num_neighbors=32, flow="source_to_target", num_workers=1):
    ei = _radius(x, x, r, batch, batch, max_num_neighbors if loop else max_num_neighbors + 1)
    row, col = (ei[1], ei[0]) if flow == "source_to_target" else (ei[0], ei[1])
    if not loop:
        m = row != col
        row, col = row[m], col[m]
    return torch.stack([row, col], dim=0)


def _knn(x, y, k, batch_x=None, batch_y=None, cosine=False, num_workers=1):
    if batch_x is None:
        batch_x = torch.zeros(x.shape[0], dtype=torch.long)
    if batch_y is None:
        batch_y = torch.zeros(y.shape[0], dtype=torch.long)
    d2 = _d2(x.float(), y.float())
    same = batch_y[:, None] == batch_x[None, :]
    d2 = torch.where(same, d2, torch.full_like(d2, float("inf")))
    order = torch.sort(d2, dim=1, stable=True)[1][:, :min(k, x.shape[0])]
    yi = torch.arange(y.shape[0])[:, None].expand_as(order)
    valid = torch.gather(same, 1, order)
    return torch.stack([yi[valid], order[valid]], dim=0)


def _knn_graph(x, k, batch=None, loop=False, flow="source_to_target", cosine=False, num_workers=1):
    ei = _knn(x, x, k if loop else k + 1, batch, batch)
    row, col = (ei[1], ei[0]) if flow == "source_to_target" else (ei[0], ei[1])
    if not loop:
        m = row != col
        row, col = row[m], col[m]
    return torch.stack([row, col], dim=0)


# --------------------------------------------------------------------------- torch_geometric stand-in
def _pyg_softmax(src, index, ptr=None, num_nodes=None, dim=0):
    if index.numel() == 0:
        return src
    n = int(index.max()) + 1 if num_nodes is None else num_nodes
    idx = index.view(-1, *([1] * (src.dim() - 1))).expand_as(src)
    smax = torch.full((n,) + src.shape[1:], float("-inf"), dtype=src.dtype)
    smax = smax.scatter_reduce(0, idx, src.detach(), reduce="amax", include_self=True)
    out = (src - smax[index]).exp()
    ssum = torch.zeros((n,) + src.shape[1:], dtype=src.dtype).index_add_(0, index, out) + 1e-16
    return out / ssum[index]


class _MessagePassing(nn.Module):
    """propagate() for the one usage in the reference (attention_layer.py:117):
    flow source_to_target, aggr='add', node_dim=0; kwargs ending in _i/_j are gathered by
    target/source index; ``index`` = target index, ``ptr`` = None."""

    def __init__(self, aggr="add", node_dim=0, **kwargs):
        super().__init__()
        assert aggr == "add" and node_dim == 0

    def propagate(self, edge_index, **kwargs):
        src, dst = edge_index[0], edge_index[1]
        x_dst = kwargs["x_dst"]
        msg = self.message(q_i=kwargs["q"][dst], k_j=kwargs["k"][src], v_j=kwargs["v"][src],
                           r=kwargs["r"], index=dst, ptr=None)
        out = torch.zeros((x_dst.shape[0],) + msg.shape[1:], dtype=msg.dtype).index_add_(0, dst, msg)
        return self.update(out, x_dst=x_dst)


# --------------------------------------------------------------------------- install
_INSTALLED = False


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _pkg(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    sys.modules[name] = m
    return m


def install():
    global _INSTALLED
    if _INSTALLED:
        return
    if not available():
        raise RuntimeError("reference tree not present (this harness only runs in the build container)")

    class _Lightning(nn.Module):
        @property
        def device(self):
            return torch.device("cpu")

        def log(self, *a, **k):
            pass

    pl = _mod("pytorch_lightning", LightningModule=_Lightning, Callback=object)
    _mod("torchmetrics", Metric=nn.Module, MeanMetric=nn.Module)
    _mod("wandb")
    _mod("peft")
    yc = _mod("yacs")
    yc.config = _mod("yacs.config", CfgNode=CfgNode)
    _mod("torch_cluster", radius=_radius, radius_graph=_radius_graph, knn=_knn, knn_graph=_knn_graph)
    tg = _mod("torch_geometric")
    tg.nn = _mod("torch_geometric.nn")
    tg.nn.conv = _mod("torch_geometric.nn.conv", MessagePassing=_MessagePassing)
    tg.utils = _mod("torch_geometric.utils", softmax=_pyg_softmax)

    # package shells: let submodules import without running the reference's __init__ chains
    root = os.path.join(REF_ROOT, "prosim")
    for name in ["prosim", "prosim.core", "prosim.config", "prosim.dataset", "prosim.models", "prosim.models.layers",
                 "prosim.models.utils", "prosim.models.scene_encoder", "prosim.models.decoder",
                 "prosim.models.policy", "prosim.models.prompt_encoder", "prosim.models.condition_transformer",
                 "prosim.rollout", "prosim.loss", "prosim.metrics"]:
        _pkg(name, os.path.join(REF_ROOT, *name.split(".")))
    # modules of the reference that only drag in absent deps and carry no arithmetic on this path
    _mod("prosim.models.utils.visualization", vis_agent_traj_pred=None, vis_scene_traj_pred=None,
         visualization_callback=None)
    _mod("prosim.rollout.distributed_utils", check_mem_usage=lambda *a, **k: None,
         print_system_mem_usage=lambda *a, **k: None, get_gpu_memory_usage=lambda *a, **k: None)
    _mod("prosim.loss.loss_func", loss_func_dict={})
    _mod("prosim.models.condition_transformer.text_attns", text_attns={})
    wi = importlib.import_module("prosim.models.utils.weight_init")
    sys.modules["prosim.models.utils"].weight_init = wi.weight_init
    # condition_transformer/__init__ is empty in the reference; traj_sam imports the class from the package
    base = importlib.import_module("prosim.models.condition_transformer.base")
    sys.modules["prosim.models.condition_transformer"].ConditionTransformer = base.ConditionTransformer
    _INSTALLED = True


def get_config(cond_types=("goal", "v_action_tag", "drag_point"), overrides=None):
    """The demo config (prosim_demo/cfg/no_text.yaml over config/default.py)."""
    install()
    default = importlib.import_module("prosim.config.default")
    opts = ["PROMPT.CONDITION.TYPES", list(cond_types)]
    cfg = default.get_config(os.path.join(REF_ROOT, "prosim_demo/cfg/no_text.yaml"), opts + list(overrides or []))
    return cfg


def build_model(cfg):
    """Instantiate the reference ``ProSim`` (traj_sam.py:14) without Lightning's trainer
    scaffolding: BaseModel.__init__'s metric/loss wiring is skipped, the sub-modules are the
    reference's own classes built by its own ``_config_models``."""
    install()
    for m in ["prosim.models.scene_encoder.base", "prosim.models.scene_encoder.attn_fusion",
              "prosim.models.decoder.base", "prosim.models.decoder.sym_coord",
              "prosim.models.prompt_encoder.base", "prosim.models.policy.base"]:
        importlib.import_module(m)
    traj_sam = importlib.import_module("prosim.models.traj_sam")

    class RefProSim(traj_sam.ProSim):
        def __init__(self, config):
            nn.Module.__init__(self)
            self.config = config
            self.tasks = config.TASK.TYPES
            self._config_models()
            self.rollout_steps = config.ROLLOUT.POLICY.REPLAN_FREQ
            self.rollout_top_k = config.ROLLOUT.POLICY.TOP_K
            self.rollout_top_k_train = config.ROLLOUT.POLICY.TOP_K_TRAIN
            self.hist_step = config.DATASET.FORMAT.HISTORY.STEPS
            self.pred_gmm = config.MODEL.POLICY.ACT_DECODER.TRAJ.PRED_GMM
            self.pred_vel = config.MODEL.POLICY.ACT_DECODER.TRAJ.PRED_VEL

    return RefProSim(cfg).eval()


from .ref_batch import Extras, make_batch  # noqa: E402,F401  (data only: shared with the -m gpu tests)


# --------------------------------------------------------------------------- the reference's rollout metric
class _MetricBase(nn.Module):
    """torchmetrics.Metric stand-in (structural): a module that knows its device."""

    @property
    def device(self):
        return torch.device("cpu")


class _MeanMetric(_MetricBase):
    """torchmetrics.MeanMetric as the reference uses it (update(value) with the default weight 1, nan_strategy
    'warn'): NaN entries of an update are dropped, compute() = sum of the kept values / their count."""

    def __init__(self):
        super().__init__()
        self.total = torch.zeros((), dtype=torch.float64)
        self.count = torch.zeros((), dtype=torch.float64)

    def update(self, value):
        v = torch.as_tensor(value, dtype=torch.float32).reshape(-1)
        v = v[~v.isnan()]
        self.total += v.double().sum()
        self.count += v.numel()

    def compute(self):
        return (self.total / self.count).float()

    def reset(self):
        self.total.zero_()
        self.count.zero_()


def load_pair_metric():
    """The reference's own PairMotionPred class (metrics/motion_pred.py:111) over its own loss/loss_func.py, with
    stand-ins for torchmetrics (above) and for prosim.dataset.data_utils.rotate (trajdata-bound, not on this path)."""
    install()
    import importlib.util

    def rotate(x, y, angle):
        return torch.stack([x * torch.cos(angle) - y * torch.sin(angle), x * torch.sin(angle) + y * torch.cos(angle)], dim=-1)

    _mod("torchmetrics", Metric=_MetricBase, MeanMetric=_MeanMetric, Accuracy=_MetricBase)
    saved = {k: sys.modules.get(k) for k in ("prosim.dataset.data_utils", "prosim.loss.loss_func")}
    _mod("prosim.dataset.data_utils", rotate=rotate)
    spec = importlib.util.spec_from_file_location("prosim.loss.loss_func", os.path.join(REF_ROOT, "prosim", "loss", "loss_func.py"))
    lf = importlib.util.module_from_spec(spec)
    sys.modules["prosim.loss.loss_func"] = lf
    spec.loader.exec_module(lf)
    importlib.import_module("prosim.core.registry")
    mp = importlib.import_module("prosim.metrics.motion_pred")
    for k, v in saved.items():   # the rollout harness keeps its own (empty) loss_func stand-in
        if v is not None and k == "prosim.loss.loss_func":
            sys.modules[k] = v
    return mp.PairMotionPred, lf


def load_world_output():
    """The reference's own ``obtain_rollout_trajs_in_world`` and ``replica_batch_for_parallel_rollout``
    (rollout/gpu_utils.py:59-123, :230-281) over its own rollout/utils.py transforms and models/utils/geometry.py.
    Structural stand-ins only: the Waymo packaging helpers, trajdata's array utilities and the dataset classes those
    files import by name carry no arithmetic on these two functions."""
    install()
    saved = {k: sys.modules.get(k) for k in ("prosim.dataset.format_utils", "prosim.dataset.data_utils", "prosim.models.utils.data")}
    names = ("get_waymo_file_template", "get_waymo_scene_object", "joint_scene_from_states", "plot_waymo_gt_trajectory",
             "rollout_states_to_joint_scene", "plot_waymo_rollout_trajectory")
    _mod("prosim.rollout.waymo_utils", **{n: None for n in names})
    _mod("prosim.dataset.format_utils", BatchDataDict=dict, InputMaskData=type("InputMaskData", (), {"from_dict": staticmethod(lambda d: d)}))
    _mod("prosim.dataset.data_utils", rotate=None, transform_to_frame_offset_rot=None)
    _mod("prosim.models.utils.data", get_agent_pos_dict=None, extract_agent_obs_from_center_obs=None)
    sys.modules["prosim.models.utils.visualization"].vis_rollout_traj_pred = None
    td = _mod("trajdata")
    td.data_structures = _mod("trajdata.data_structures")
    td.data_structures.state = _mod("trajdata.data_structures.state", StateArray=None)
    td.utils = _mod("trajdata.utils")
    td.utils.arr_utils = _mod("trajdata.utils.arr_utils", transform_coords_np=None, transform_angles_np=None, transform_matrices=None)
    for k in ("prosim.rollout.utils", "prosim.rollout.gpu_utils"):
        sys.modules.pop(k, None)
    gu = importlib.import_module("prosim.rollout.gpu_utils")
    for k, v in saved.items():
        if v is not None:
            sys.modules[k] = v
    return gu.obtain_rollout_trajs_in_world, gu.replica_batch_for_parallel_rollout


# --------------------------------------------------------------------------- the reference's input formatters
class _StateArray(np.ndarray):
    """trajdata's ``StateArray`` (trajdata/utils/state_utils.py -- third party, absent here) restated from its published
    interface for the calls the reference's formatters make: a float array whose last axis follows a comma-separated
    ``_format`` ('x,y,z,xd,yd,xdd,ydd,s,c' | '...,h'); ``position`` = (x, y), ``velocity`` = (xd, yd), ``acceleration`` =
    (xdd, ydd), ``heading_vector`` = (c, s), ``heading`` = h or atan2(s, c) with a trailing axis of one; ``as_format``
    re-orders / derives columns (s = sin h, c = cos h).  ARITHMETIC-BEARING stand-in: what it pins is labelled
    'ref + trajdata stand-ins'."""
    _format = ""

    @classmethod
    def from_array(cls, a, format):
        out = np.asarray(a).view(cls)
        out._format = format
        return out

    def __array_finalize__(self, obj):
        self._format = getattr(obj, "_format", "")

    @property
    def _format_dict(self):
        return {k: i for i, k in enumerate(self._format.split(","))}

    def _cols(self, names):
        d = self._format_dict
        return np.asarray(self)[..., [d[n] for n in names]]

    def _set(self, names, v):
        d = self._format_dict
        np.asarray(self)[..., [d[n] for n in names]] = v

    position = property(lambda s: s._cols("xy"), lambda s, v: s._set("xy", v))
    velocity = property(lambda s: s._cols(["xd", "yd"]), lambda s, v: s._set(["xd", "yd"], v))
    acceleration = property(lambda s: s._cols(["xdd", "ydd"]), lambda s, v: s._set(["xdd", "ydd"], v))
    heading_vector = property(lambda s: s._cols("cs"), lambda s, v: s._set("cs", v))

    @property
    def heading(self):
        d = self._format_dict
        if "h" in d:
            return self._cols("h")
        sc = self._cols("sc")
        return np.arctan2(sc[..., :1], sc[..., 1:])

    @heading.setter
    def heading(self, v):
        self._set("h", v)

    def as_format(self, fmt):
        d = self._format_dict
        a = np.asarray(self)
        cols = []
        for k in fmt.split(","):
            if k in d:
                cols.append(a[..., d[k]])
            elif k == "s":
                cols.append(np.sin(a[..., d["h"]]))
            elif k == "c":
                cols.append(np.cos(a[..., d["h"]]))
            elif k == "h":
                cols.append(np.arctan2(a[..., d["s"]], a[..., d["c"]]))
            else:
                raise KeyError(k)
        return _StateArray.from_array(np.stack(cols, -1), fmt)

    def copy(self, *a, **k):
        return _StateArray.from_array(np.array(self), self._format)


class _StateTensor:
    """trajdata's ``StateTensor`` for the same calls: a thin wrapper around a torch tensor with a ``_format``."""

    def __init__(self, t, fmt):
        self.t, self._format = t, fmt

    @classmethod
    def from_array(cls, a, format):
        return cls(a.t if isinstance(a, _StateTensor) else torch.as_tensor(a), format)

    @classmethod
    def from_numpy(cls, a):
        return cls(torch.from_numpy(np.array(a)), a._format)

    def numpy(self):
        return _StateArray.from_array(self.t.numpy().copy(), self._format)

    @property
    def shape(self):
        return self.t.shape

    @property
    def device(self):
        return self.t.device

    def __getitem__(self, idx):
        return _StateTensor(self.t[idx], self._format)

    def _cols(self, names):
        d = {k: i for i, k in enumerate(self._format.split(","))}
        return self.t[..., [d[n] for n in names]]

    @property
    def position(self):
        return self._cols("xy")

    @property
    def heading(self):
        d = self._format.split(",")
        if "h" in d:
            return self._cols("h")
        sc = self._cols("sc")
        return torch.atan2(sc[..., :1], sc[..., 1:])

    def as_format(self, fmt):
        return _StateTensor(torch.from_numpy(np.asarray(self.numpy().as_format(fmt))), fmt)

    def as_tensor(self):
        return self.t

    def float(self):
        return self.t.float()

    def isnan(self):
        return self.t.isnan()

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):   # torch.cat([StateTensor, ...]) -> plain tensor
        unwrap = lambda x: x.t if isinstance(x, _StateTensor) else ([unwrap(y) for y in x] if isinstance(x, (list, tuple)) else x)
        return func(*unwrap(args), **(kwargs or {}))


class _SceneBatch:
    """Duck type of trajdata's ``SceneBatch`` with the attributes the formatters read."""

    def __init__(self, **kw):
        self.__dict__.update(kw)


def load_format():
    """The reference's own ``dataset/format_utils.py`` and ``dataset/data_utils.py`` (the rollout path's input formatters)
    under stand-ins for trajdata: structural for the batch / dataset classes, arithmetic-bearing for ``StateArray`` /
    ``StateTensor`` (above) and for three two-line helpers of trajdata.utils.arr_utils (``rotation_matrix``,
    ``angle_wrap``, ``transform_coords_np``).  ``data_utils.py`` unpickles two id lists at import, one of which
    (waymo_train_IDs.pkl) is not in the tree: ``open`` is shimmed for that one file.  Returns (format_utils, data_utils,
    SceneBatch, StateTensor)."""
    install()
    import builtins
    import io
    import pickle
    from enum import IntEnum

    class AgentType(IntEnum):
        UNKNOWN = 0
        VEHICLE = 1
        PEDESTRIAN = 2
        BICYCLE = 3
        MOTORCYCLE = 4

    def rotation_matrix(angle):
        c, s = np.cos(angle), np.sin(angle)
        return np.stack([np.stack([c, -s], -1), np.stack([s, c], -1)], -2)

    def angle_wrap(r):
        return (r + np.pi) % (2 * np.pi) - np.pi

    def transform_coords_np(coords, tf, translate=True):
        out = np.einsum("...ij,...j->...i", tf[..., :-1, :-1], coords)
        return out + tf[..., :-1, -1] if translate else out

    td = _mod("trajdata", AgentBatch=type("AgentBatch", (), {}), AgentType=AgentType)
    td.maps = _mod("trajdata.maps")
    td.maps.map_api = _mod("trajdata.maps.map_api", MapAPI=None)
    td.data_structures = _mod("trajdata.data_structures")
    td.data_structures.batch_element = _mod("trajdata.data_structures.batch_element", AgentBatchElement=object, SceneBatchElement=object)
    td.data_structures.batch = _mod("trajdata.data_structures.batch", AgentBatch=td.AgentBatch, SceneBatch=_SceneBatch)
    td.utils = _mod("trajdata.utils")
    td.utils.arr_utils = _mod("trajdata.utils.arr_utils", transform_coords_np=transform_coords_np, rotation_matrix=rotation_matrix,
                              angle_wrap=angle_wrap)
    td.utils.state_utils = _mod("trajdata.utils.state_utils", StateArray=_StateArray, StateTensor=_StateTensor, transform_state_np_2d=None)
    td.augmentation = _mod("trajdata.augmentation", BatchAugmentation=object)
    _mod("prosim.models.utils.data", get_agent_pos_dict=None, extract_agent_obs_from_center_obs=None)
    _mod("prosim.dataset.condition_utils", ConditionGenerator=None, BatchCondition=None)
    real_open = builtins.open

    def shim(path, *a, **k):
        if str(path).endswith("waymo_train_IDs.pkl") and not os.path.exists(path):
            return io.BytesIO(pickle.dumps([]))
        return real_open(path, *a, **k)

    for k in ("prosim.dataset.data_utils", "prosim.dataset.format_utils", "prosim.dataset.prompt_utils", "prosim.dataset.motion_tag_utils"):
        sys.modules.pop(k, None)
    builtins.open = shim
    try:
        du = importlib.import_module("prosim.dataset.data_utils")
        fu = importlib.import_module("prosim.dataset.format_utils")
        pu = importlib.import_module("prosim.dataset.prompt_utils")
    finally:
        builtins.open = real_open
    return fu, du, pu, _SceneBatch, _StateTensor
