"""``DCVC_HEM`` -- the reference's training/evaluation wrapper around ``DMC``
(/root/reference/core/model/dcvc_hem.py:10-631, ``build_model`` core/model/__init__.py:9-11),
for evaluation and for training steps.

What is kept: the ``forward(forward_method, ...)`` dispatch (:605-631) with its five methods, the
loss assembly ``rate + lambda * (dist * dist_lambda + p_dist * pl_lambda)`` (:208, :300, :436,
:554), the P-frame loops and DPB hand-over of ``single`` (:160-236) / ``single_multi`` (:284-322)
/ ``cascade`` (:383-483) / ``cascade_multi`` (:520-587) / ``forward_simple`` (:589-603), the
result dictionaries with their shapes, ``lambdas`` / ``dmc`` attributes, ``state_dict`` keys
prefixed ``dmc.`` and the ``activate_modules_*`` parameter groups (:23-102).

Training (``is_train=True``): ``self.dmc.forward_one_frame`` in ``.train()`` mode is one autograd
node whose backward runs the kernels of include/dcvc_hip_grad.h (vcm_ts_amd/grad.py), so
``loss_to_opt.backward()`` and ``optimizer.step()`` happen inside ``forward`` exactly where the
reference has them (:224-229, :466-471): once per P picture in ``single``, once per sub-sequence
with the gradient flowing through the un-detached DPB in ``cascade``; the ``*_multi`` variants
return ``loss_to_opt`` for the caller (trainer_multi.py) to step.

Not built: the detector-based perceptual losses (core/engine/losses.py, out of scope: third-party
networks and weights) -- replaced by a pluggable callable ``perceptual_loss(target, recon) -> (N,)``.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Callable, List, Optional

import torch
from torch import nn

from .dmc import DMC

_INTER_PREFIXES = ("bit_estimator_z_mv", "mv_decoder", "mv_encoder", "mv_hyper_prior_decoder", "mv_hyper_prior_encoder",
                   "mv_y_spatial_prior", "mv_y_prior_fusion", "optic_flow")  # dcvc_hem.py:23-32
_INTER_RATE = ("mv_y_q_basic", "mv_y_q_scale")  # :34-37
_RECON_RATE = ("y_q_basic", "y_q_scale")  # :39-42


def make_cfg(lambdas=(85.0, 170.0, 380.0, 840.0), pl_lambda=0.0, dist_lambda=1.0, pl_layers=()):
    """Minimal stand-in for the yacs node the reference reads (cfg.SOLVER.*, cfg.MODEL.ARCHITECTURE)."""
    return SimpleNamespace(SOLVER=SimpleNamespace(LAMBDAS=list(lambdas), PL_LAMBDA=pl_lambda, DIST_LAMBDA=dist_lambda,
                                                  PL_MODEL=None, PL_LAYERS=list(pl_layers)),
                           MODEL=SimpleNamespace(ARCHITECTURE="DCVC_HEM"))


class DCVC_HEM(nn.Module):
    def __init__(self, cfg, perceptual_loss: Optional[Callable] = None, precision=None):
        super().__init__()
        self.cfg = cfg
        self.dmc = DMC(anchor_num=len(cfg.SOLVER.LAMBDAS), precision=precision)
        self.register_buffer("lambdas", torch.tensor(cfg.SOLVER.LAMBDAS, dtype=torch.float32), persistent=False)
        self.register_buffer("pl_lambda", torch.tensor(float(cfg.SOLVER.PL_LAMBDA)), persistent=False)
        self.register_buffer("dist_lambda", torch.tensor(float(cfg.SOLVER.DIST_LAMBDA)), persistent=False)
        self.perceptual_loss = perceptual_loss  # callable(target (N,3,H,W), recon) -> (N,) or None

    # ------------------------------------------------------------------ parameter groups (:59-102)
    def _set(self, pred, flag):
        for name, p in self.dmc.named_parameters():
            if pred(name):
                p.requires_grad = flag

    def activate_modules_inter_dist(self):
        self._set(lambda n: True, False)
        self._set(lambda n: n.startswith(_INTER_PREFIXES), True)

    def activate_modules_inter_dist_rate(self):
        self.activate_modules_inter_dist()
        self._set(lambda n: n in _INTER_RATE, True)

    def activate_modules_recon_dist(self):
        self._set(lambda n: True, True)
        self._set(lambda n: n.startswith(_INTER_PREFIXES) or n in _INTER_RATE or n in _RECON_RATE, False)

    def activate_modules_recon_dist_rate(self):
        self._set(lambda n: True, True)
        self._set(lambda n: n.startswith(_INTER_PREFIXES) or n in _INTER_RATE, False)

    def activate_modules_all(self):
        self._set(lambda n: True, True)

    # ------------------------------------------------------------------ shared pieces
    def _step(self, optimizer, loss_to_opt, is_train):
        """:224-229 / :466-471."""
        if is_train:
            if optimizer is None:
                raise ValueError("is_train=True needs an optimizer")
            optimizer.zero_grad()
            loss_to_opt.backward()
            optimizer.step()

    def _first_dpb(self, input, t_i, i_frame_net, i_frame_q_scales):
        """I-frame initialisation of a sub-sequence (:160-180)."""
        if i_frame_net is None:
            ref = input[:, t_i]
        else:
            with torch.no_grad():
                ref = torch.stack([i_frame_net(input[i, t_i].unsqueeze(0), i_frame_q_scales[i])["x_hat"].squeeze(0).clone()
                                   for i in range(input.shape[0])], 0)
        return {"ref_frame": ref, "ref_feature": None, "ref_y": None, "ref_mv_y": None}

    def _p_frame(self, frame, target_frame, dpb, loss_dist_key, loss_rate_keys, perceptual_loss, keep_graph=False):
        """One forward_one_frame plus the loss terms; returns (rate, dist, p_dist, loss, new dpb)."""
        out = self.dmc.forward_one_frame(frame, dpb, self.dmc.mv_y_q_scale, self.dmc.y_q_scale)
        lambdas = self.lambdas if len(loss_rate_keys) else torch.ones_like(self.lambdas)
        rate = torch.zeros_like(self.lambdas)
        for key in loss_rate_keys:
            rate = rate + out[key]
        dist = out[loss_dist_key]
        if perceptual_loss:
            if self.perceptual_loss is None:
                raise RuntimeError("perceptual_loss=True needs a perceptual_loss callable (detector losses are out of scope)")
            p_dist = self.perceptual_loss(target_frame, out["dpb"]["ref_frame"])
        else:
            p_dist = torch.zeros_like(self.lambdas)
        loss = rate + lambdas * (dist * self.dist_lambda + p_dist * self.pl_lambda)
        if keep_graph and self.dmc.training:   # cascade: the next picture back-propagates into this one (:418)
            new_dpb = dict(out["dpb"])
        elif self.dmc.training:                # a recorded forward owns its buffers: detaching is enough (:195-196)
            new_dpb = {k: v.detach() for k, v in out["dpb"].items()}
        else:                                  # inference views are recycled two calls later: keep a copy
            new_dpb = {k: v.detach().clone() for k, v in out["dpb"].items()}
        return rate, dist, p_dist, loss, new_dpb

    @staticmethod
    def _seqs(stacked):
        """list over t_i of (N, C, H, W, p+1) -> (N, T-p, p+1, C, H, W) (:246-251)."""
        return torch.stack(stacked, -1).permute(0, 5, 4, 1, 2, 3)

    # ------------------------------------------------------------------ forward methods
    def forward_single(self, input, target, optimizer, loss_dist_key, loss_rate_keys, p_frames, perceptual_loss,
                       is_train=True, i_frame_net=None, i_frame_q_scales=None):
        n, t = input.shape[:2]
        assert 0 < p_frames < t and self.lambdas.shape[0] == n
        res = {k: [] for k in ("rate", "dist", "p_dist", "loss", "loss_seq", "input_seqs", "decod_seqs")}
        res["single_forwards"] = 0
        for t_i in range(t - p_frames):
            dpb = self._first_dpb(input, t_i, i_frame_net, i_frame_q_scales)
            ins, decs, losses = [target[:, t_i]], [input[:, t_i]], []
            for p in range(p_frames):
                rate, dist, p_dist, loss, dpb = self._p_frame(input[:, t_i + 1 + p], target[:, t_i + 1 + p], dpb,
                                                              loss_dist_key, loss_rate_keys, perceptual_loss)
                self._step(optimizer, torch.mean(loss), is_train)
                for k, v in (("rate", rate), ("dist", dist), ("p_dist", p_dist), ("loss", loss)):
                    res[k].append(v.detach() if is_train else v)
                losses.append(loss.detach() if is_train else loss)
                res["single_forwards"] += 1
                ins.append(target[:, t_i + 1 + p])
                decs.append(dpb["ref_frame"])
            res["loss_seq"].append(torch.stack(losses, -1).mean(-1))
            res["input_seqs"].append(torch.stack(ins, -1))
            res["decod_seqs"].append(torch.stack(decs, -1))
        for k in ("rate", "dist", "p_dist", "loss", "loss_seq"):
            res[k] = torch.stack(res[k], -1)
        res["input_seqs"], res["decod_seqs"] = self._seqs(res["input_seqs"]), self._seqs(res["decod_seqs"])
        return res

    def forward_single_multi(self, input, target, loss_dist_key, loss_rate_keys, dpb, perceptual_loss):
        assert self.lambdas.shape[0] == input.shape[0]
        rate, dist, p_dist, loss, new_dpb = self._p_frame(input, target, dpb, loss_dist_key, loss_rate_keys, perceptual_loss)
        for k in dpb.keys():
            dpb[k] = new_dpb[k]
        return {"rate": rate, "dist": dist, "p_dist": p_dist, "loss": loss, "loss_to_opt": loss.mean(),
                "input_seqs": target, "decod_seqs": dpb["ref_frame"], "dpb": dpb}

    def _cascade_span(self, input, target, dpb, t_i, p_frames, loss_dist_key, loss_rate_keys, perceptual_loss):
        ins, decs = [target[:, t_i]], [input[:, t_i]]
        acc = {"rate": [], "dist": [], "p_dist": [], "loss": []}
        for p in range(p_frames):
            rate, dist, p_dist, loss, dpb = self._p_frame(input[:, t_i + 1 + p], target[:, t_i + 1 + p], dpb,
                                                          loss_dist_key, loss_rate_keys, perceptual_loss, keep_graph=True)
            for k, v in (("rate", rate), ("dist", dist), ("p_dist", p_dist), ("loss", loss)):
                acc[k].append(v)
            ins.append(target[:, t_i + 1 + p])
            decs.append(dpb["ref_frame"])
        means = {k: torch.stack(v, -1).mean(-1) for k, v in acc.items()}
        return means, torch.stack(ins, -1), torch.stack(decs, -1), dpb

    def forward_cascade(self, input, target, optimizer, loss_dist_key, loss_rate_keys, p_frames, perceptual_loss,
                        is_train=True, i_frame_net=None, i_frame_q_scales=None):
        n, t = input.shape[:2]
        assert 0 < p_frames < t and self.lambdas.shape[0] == n
        res = {k: [] for k in ("rate", "dist", "p_dist", "loss", "input_seqs", "decod_seqs")}
        res["single_forwards"] = 0
        for t_i in range(t - p_frames):
            dpb = self._first_dpb(input, t_i, i_frame_net, i_frame_q_scales)
            means, ins, decs, _ = self._cascade_span(input, target, dpb, t_i, p_frames, loss_dist_key, loss_rate_keys,
                                                     perceptual_loss)
            self._step(optimizer, torch.mean(means["loss"]), is_train)
            for k in ("rate", "dist", "p_dist", "loss"):
                res[k].append(means[k].detach() if is_train else means[k])
            res["single_forwards"] += 1
            res["input_seqs"].append(ins)
            res["decod_seqs"].append(decs.detach())
        for k in ("rate", "dist", "p_dist", "loss"):
            res[k] = torch.stack(res[k], -1)
        res["loss_seq"] = res["loss"]
        res["input_seqs"], res["decod_seqs"] = self._seqs(res["input_seqs"]), self._seqs(res["decod_seqs"])
        return res

    def forward_cascade_multi(self, input, target, loss_dist_key, loss_rate_keys, dpb, p_frames, t_i, perceptual_loss):
        assert self.lambdas.shape[0] == input.shape[0]
        means, ins, decs, dpb = self._cascade_span(input, target, dpb, t_i, p_frames, loss_dist_key, loss_rate_keys,
                                                   perceptual_loss)
        return {"rate": means["rate"], "dist": means["dist"], "p_dist": means["p_dist"], "loss": means["loss"],
                "loss_to_opt": means["loss"].mean(-1), "input_seqs": ins, "decod_seqs": decs, "dpb": dpb}

    def forward_simple(self, input, dpb):
        n = input.shape[0]
        assert self.lambdas.shape[0] == n
        out = []
        for i in range(n):
            r = self.dmc.forward_one_frame(input[i], dpb[i], self.dmc.mv_y_q_scale[i], self.dmc.y_q_scale[i])
            out.append({k: v.clone() for k, v in r["dpb"].items()})
        return out

    def forward(self, forward_method: str, input, target=None, loss_dist_key=None, loss_rate_keys: List[str] = None,
                p_frames=None, perceptual_loss=None, optimizer=None, is_train=True, dpb=None, t_i=None, i_frame_net=None,
                i_frame_q_scales=None):
        if forward_method == "single":
            return self.forward_single(input, target, optimizer, loss_dist_key, loss_rate_keys, p_frames, perceptual_loss,
                                       is_train, i_frame_net, i_frame_q_scales)
        if forward_method == "single_multi":
            return self.forward_single_multi(input, target, loss_dist_key, loss_rate_keys, dpb, perceptual_loss)
        if forward_method == "cascade":
            return self.forward_cascade(input, target, optimizer, loss_dist_key, loss_rate_keys, p_frames, perceptual_loss,
                                        is_train, i_frame_net, i_frame_q_scales)
        if forward_method == "cascade_multi":
            return self.forward_cascade_multi(input, target, loss_dist_key, loss_rate_keys, dpb, p_frames, t_i,
                                              perceptual_loss)
        if forward_method == "forward_simple":
            return self.forward_simple(input, dpb)
        raise ValueError(f"unknown forward_method {forward_method!r}")


_MODEL_ARCHITECTURES = {"DCVC_HEM": DCVC_HEM}


def build_model(cfg, **kw):
    """core/model/__init__.py:9-11."""
    return _MODEL_ARCHITECTURES[cfg.MODEL.ARCHITECTURE](cfg, **kw)
