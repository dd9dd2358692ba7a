"""Test infrastructure (like everything under oracle/): where does the reference's arithmetic graze its +-pi discontinuities?
``near_cut_edges`` runs the fp64 oracle on a scene and returns every relative-PE edge whose ``rel_ori`` (wrap_angle of the heading
difference, act_decoder.py:203-217 and twins) or bearing (the atan2 of angle_between_2d_vectors, geometry.py:6-11) lies within
``thr`` rad of the cut -- with the call it belongs to and its (destination row, source row).  An fp32 evaluation whose upstream
values differ by about the margin times the edge length lands on the other side, and the destination's Fourier features change at
order 1: such rows are the cut agents of a workload (tests/golden/known_cut_agents.json)."""
import math
from typing import Dict, List, Tuple

import torch

from . import prosim_oracle as orc


def near_cut_edges(w: Dict, spec, scene: Dict, thr: float = 1e-5) -> Tuple[List[tuple], Dict[str, int]]:
    """[(margin rad, kind 'rel_ori' | 'atan2', call number, dst row, src row, edges of the call)], sorted by margin; call counts."""
    found: List[tuple] = []
    calls = {"pe": 0}
    real_pe = orc.rel_pe_input

    def pe_rec(src, dst, ori_dst, pos_dst, ori_src, pos_src):
        calls["pe"] += 1
        out = real_pe(src, dst, ori_dst, pos_dst, ori_src, pos_src)
        ang = out[..., 2].detach().double()
        m = math.pi - ang.abs()
        for i in torch.nonzero(m < thr)[:256]:
            e = int(i[0])
            found.append((float(m[e]), "atan2", calls["pe"], int(dst[e]), int(src[e]), int(src.numel())))
        u = ((ori_src[src] - ori_dst[dst]).detach().double().reshape(-1) + math.pi) % (2 * math.pi)
        m = torch.minimum(u, 2 * math.pi - u)
        for i in torch.nonzero(m < thr)[:256]:
            e = int(i[0])
            found.append((float(m[e]), "rel_ori", calls["pe"], int(dst[e]), int(src[e]), int(src.numel())))
        return out

    orc.rel_pe_input = pe_rec
    try:
        with torch.no_grad():
            orc.rollout(w, spec, scene, dtype=torch.float64)
    finally:
        orc.rel_pe_input = real_pe
    found.sort(key=lambda t: t[0])
    return found, calls
