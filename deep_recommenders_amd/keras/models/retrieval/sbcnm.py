"""Sampling-bias-corrected two-tower retrieval task — same surface as the reference's
keras/models/retrieval/sbcnm.py (helpers :15-86, `Retrieval` :89-163), on the K9 kernels.

The reference's optional branches name a module that does not exist (`deep_recommenders.keras.layers.
embedding.loss.*`, :136-146): only its all-None path can run.  Here those branches call the
module-local layers below (the classes the reference's own tests exercise, tests/keras/test_sbcnm.py)."""
from typing import Optional, Tuple

import numpy as np
import torch
from torch import nn

from deep_recommenders_amd import ops
from deep_recommenders_amd.keras.models.retrieval import FactorizedTopK
from deep_recommenders_amd.keras.models.retrieval.factorized_top_k import _dev

MAX_FLOAT = np.finfo(np.float32).max / 100.0     # sbcnm.py:9
MIN_FLOAT = np.finfo(np.float32).min / 100.0     # sbcnm.py:10


def _gather_elements_along_row(data, column_indices):
    """same as factorized_top_k._take_long_axis (sbcnm.py:15-30)"""
    data = _dev(data, torch.float32)
    column_indices = _dev(column_indices, torch.int64)
    assert data.shape[0] == column_indices.shape[0]                                  # :18-19
    return ops.take_along_rows(data, column_indices)


class HardNegativeMining(nn.Module):
    """keeps the positive and the `num_hard_negatives` highest-scoring negatives per row (:33-49)"""

    def __init__(self, num_hard_negatives: int, **kwargs):
        super().__init__()
        self._num_hard_negatives = num_hard_negatives

    def forward(self, logits, labels) -> Tuple[torch.Tensor, torch.Tensor]:
        logits = _dev(logits, torch.float32)
        labels = _dev(labels, torch.float32)
        num_sampled = min(self._num_hard_negatives + 1, logits.shape[1])
        boosted = ops.logits_adjust(logits, labels, add_label_scale=MAX_FLOAT)       # logits + labels * MAX_FLOAT (:44)
        _, indices = ops.topk_select(boosted, num_sampled)
        return _gather_elements_along_row(logits, indices), _gather_elements_along_row(labels, indices)

    call = forward


class RemoveAccidentalNegative(nn.Module):
    """pushes in-batch negatives that share the positive's identifier to MIN_FLOAT (:52-75)"""

    def forward(self, logits, labels, identifiers):
        return ops.logits_adjust(_dev(logits, torch.float32), _dev(labels, torch.float32),
                                 cand_ids=_dev(identifiers, torch.int64).reshape(-1))

    call = forward


class SamplingProbabilityCorrection(nn.Module):
    """logits - log(candidate_sampling_probability) (:78-86)"""

    def forward(self, logits, candidate_sampling_probability):
        return ops.logits_adjust(_dev(logits, torch.float32),
                                 cand_prob=_dev(candidate_sampling_probability, torch.float32).reshape(-1))

    call = forward


class _InBatchSoftmaxFn(torch.autograd.Function):
    """K9: loss = sum_i w_i (logsumexp_j s_ij - s_ii); backward = softmax-gradient epilogue + two GEMMs."""

    @staticmethod
    def forward(ctx, q, c, sample_weight, cand_prob, cand_ids, inv_t):
        loss, row_lse, _ = ops.inbatch_softmax_fwd(q, c, cand_prob, cand_ids, sample_weight, inv_t)
        ctx.inv_t = inv_t
        ctx.save_for_backward(q, c, row_lse, sample_weight, cand_prob, cand_ids)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, d_loss):
        q, c, row_lse, w, cp, ci = ctx.saved_tensors
        G = ops.inbatch_softmax_grad_scores(q, c, row_lse, float(d_loss), cp, ci, w, ctx.inv_t)
        dq = ops.linear_fwd(G, c) if ctx.needs_input_grad[0] else None                  # dq = G @ c
        dc = None
        if ctx.needs_input_grad[1]:
            dc = torch.zeros_like(c)
            ops.linear_bwd_dw(G, q, 1.0, dc)                                            # dc = G^T @ q
        return dq, dc, None, None, None, None


class _AdjustFn(torch.autograd.Function):
    """scores - log p_j + (dup_ij - labels_ij) * MIN_FLOAT (sbcnm.py:66-86) on an explicit matrix: additive constants, so the
    gradient passes through unchanged."""

    @staticmethod
    def forward(ctx, scores, labels, cand_prob, cand_ids):
        return ops.logits_adjust(scores.contiguous(), labels, cand_prob=cand_prob, cand_ids=cand_ids)

    @staticmethod
    def backward(ctx, g):
        return g, None, None, None


class _ScoresFn(torch.autograd.Function):
    """scores = q c^T as an explicit [B, B] matrix with the MFMA GEMMs in both directions (for user-supplied losses)."""

    @staticmethod
    def forward(ctx, q, c):
        ctx.save_for_backward(q, c)
        return ops.scores_nt(q, c).contiguous()

    @staticmethod
    def backward(ctx, G):
        q, c = ctx.saved_tensors
        G = G.contiguous()
        dq = ops.linear_fwd(G, c) if ctx.needs_input_grad[0] else None
        dc = None
        if ctx.needs_input_grad[1]:
            dc = torch.zeros_like(c)
            ops.linear_bwd_dw(G, q, 1.0, dc)
        return dq, dc


class _ExplicitSoftmaxFn(torch.autograd.Function):
    """num_candidates != num_queries (labels = tf.eye(num_queries, num_candidates), sbcnm.py:131-134): the same call sequence
    on the explicit [Bq, Nc] score matrix -- scores -> corrections -> CCE(from_logits, SUM) -- with the row-softmax kernels
    and the MFMA GEMMs in both directions.  (The fused K9 kernels are square: one candidate per query.)"""

    @staticmethod
    def forward(ctx, q, c, sample_weight, cand_prob, cand_ids, inv_t):
        scores = ops.scores_nt(q, c).contiguous()
        labels = torch.eye(scores.shape[0], scores.shape[1], device=scores.device)
        if cand_prob is not None or cand_ids is not None:
            scores = ops.logits_adjust(scores, labels, cand_prob=cand_prob, cand_ids=cand_ids)
        ctx.inv_t = inv_t
        ctx.save_for_backward(q, c, scores, labels, sample_weight)
        return ops.softmax_ce_rows(scores, labels, inv_t, sample_weight)

    @staticmethod
    def backward(ctx, d_loss):
        q, c, scores, labels, w = ctx.saved_tensors
        G = ops.softmax_ce_rows_bwd(scores, labels, ctx.inv_t, w, float(d_loss))
        dq = ops.linear_fwd(G, c) if ctx.needs_input_grad[0] else None              # dq = G @ c
        dc = None
        if ctx.needs_input_grad[1]:
            dc = torch.zeros_like(c)
            ops.linear_bwd_dw(G, q, 1.0, dc)                                          # dc = G^T @ q
        return dq, dc, None, None, None, None


class _HardNegativeSoftmaxFn(torch.autograd.Function):
    """num_hard_negatives branch (sbcnm.py:145-151): scores -> corrections -> top-(h+1) by `logits + labels * MAX_FLOAT`
    -> CCE(from_logits, SUM).  The selection is piecewise constant, so the gradient flows only through the kept scores:
    G = scatter(softmax-CE gradient of the kept columns) into a zero [B, B] matrix, dq = G c, dc = G^T q."""

    @staticmethod
    def forward(ctx, q, c, sample_weight, cand_prob, cand_ids, inv_t, num_hard_negatives):
        scores = ops.scores_nt(q, c).contiguous()
        B = scores.shape[0]
        labels = torch.eye(B, scores.shape[1], device=scores.device)
        if cand_prob is not None or cand_ids is not None:
            scores = ops.logits_adjust(scores, labels, cand_prob=cand_prob, cand_ids=cand_ids)
        num_sampled = min(num_hard_negatives + 1, scores.shape[1])
        boosted = ops.logits_adjust(scores, labels, add_label_scale=MAX_FLOAT)       # logits + labels * MAX_FLOAT (:44)
        _, indices = ops.topk_select(boosted, num_sampled)
        s_sel = _gather_elements_along_row(scores, indices)
        l_sel = _gather_elements_along_row(labels, indices)
        ctx.inv_t = inv_t
        ctx.save_for_backward(q, c, s_sel, l_sel, indices, sample_weight)
        return ops.softmax_ce_rows(s_sel, l_sel, inv_t, sample_weight)              # :148-151

    @staticmethod
    def backward(ctx, d_loss):
        q, c, s_sel, l_sel, indices, w = ctx.saved_tensors
        B = q.shape[0]
        G = torch.zeros((B, c.shape[0]), dtype=torch.float32, device=q.device)
        ops.softmax_ce_rows_bwd(s_sel, l_sel, ctx.inv_t, w, float(d_loss), cols=indices, out=G)
        dq = ops.linear_fwd(G, c) if ctx.needs_input_grad[0] else None              # dq = G @ c
        dc = None
        if ctx.needs_input_grad[1]:
            dc = torch.zeros_like(c)
            ops.linear_bwd_dw(G, q, 1.0, dc)                                          # dc = G^T @ q
        return dq, dc, None, None, None, None, None


class Retrieval(nn.Module):
    """Retrieval(loss=None, metrics=None, temperature=None, num_hard_negatives=None).call(query_embeddings,
    candidate_embeddings, sample_weight=None, candidate_sampling_probability=None, candidate_ids=None,
    compute_metrics=True) -> loss   (sbcnm.py:92-163)."""

    def __init__(self, loss=None, metrics: Optional[FactorizedTopK] = None, temperature: Optional[float] = None,
                 num_hard_negatives: Optional[int] = None, **kwargs):
        super().__init__()
        # loss=None: CategoricalCrossentropy(from_logits=True, reduction=SUM) (sbcnm.py:100-102), fused.  Any other loss is a
        # callable loss(y_true=labels, y_pred=scores, sample_weight=...) on torch tensors, evaluated on the explicit score matrix.
        self._loss = loss
        self._factorized_metrics = metrics
        self._temperature = temperature
        self._num_hard_negatives = num_hard_negatives

    @property
    def factorized_metrics(self) -> Optional[FactorizedTopK]:
        return self._factorized_metrics

    @factorized_metrics.setter
    def factorized_metrics(self, value: Optional[FactorizedTopK]) -> None:
        self._factorized_metrics = value

    def call(self, query_embeddings, candidate_embeddings, sample_weight=None, candidate_sampling_probability=None,
             candidate_ids=None, compute_metrics: bool = True):
        q = _dev(query_embeddings, torch.float32).contiguous()
        c = _dev(candidate_embeddings, torch.float32).contiguous()
        w = _dev(sample_weight, torch.float32).reshape(-1).contiguous() if sample_weight is not None else None
        cp = _dev(candidate_sampling_probability, torch.float32).reshape(-1).contiguous() \
            if candidate_sampling_probability is not None else None
        ci = _dev(candidate_ids, torch.int64).reshape(-1).contiguous() if candidate_ids is not None else None
        inv_t = 1.0 / self._temperature if self._temperature is not None else 1.0       # :148-149
        if self._loss is not None:
            loss = self._custom_loss(q, c, w, cp, ci, inv_t)
        elif self._num_hard_negatives is None:
            for name, t, n in (("sample_weight", w, q.shape[0]), ("candidate_sampling_probability", cp, c.shape[0]),
                               ("candidate_ids", ci, c.shape[0])):
                if t is not None and t.numel() != n:
                    raise ValueError("%s must have %d elements, got %d" % (name, n, t.numel()))
            if c.shape[0] == q.shape[0]:
                loss = _InBatchSoftmaxFn.apply(q, c, w, cp, ci, inv_t)                   # :129-151 fused
            else:
                loss = _ExplicitSoftmaxFn.apply(q, c, w, cp, ci, inv_t)                  # eye(num_queries, num_candidates)
        else:
            loss = self._hard_negative_loss(q, c, w, cp, ci, inv_t)
        if compute_metrics is False or not self._factorized_metrics:                     # :153-157
            return loss
        self._factorized_metrics.update_state(q.detach(), c.detach())                    # :159-163
        return loss

    forward = call

    def _custom_loss(self, q, c, w, cp, ci, inv_t):
        """user-supplied loss object (sbcnm.py:100-103,151): the call sequence of :129-151 on an explicit score matrix.  The
        score GEMMs, the corrections (dr_logits_adjust) and the hard-negative selection (dr_topk_select) are the kernels of the
        default path; the loss itself is the user's torch callable, so the kept scores are handed to it as autograd tensors."""
        scores = _ScoresFn.apply(q, c)
        B = scores.shape[0]
        labels = torch.eye(B, scores.shape[1], device=scores.device)
        if cp is not None or ci is not None:
            scores = _AdjustFn.apply(scores, labels, cp, ci)                                  # :78-86, :66-75
        if self._num_hard_negatives is not None:
            k = min(self._num_hard_negatives + 1, scores.shape[1])
            boosted = ops.logits_adjust(scores.detach(), labels, add_label_scale=MAX_FLOAT)   # logits + labels * MAX_FLOAT (:44)
            _, idx = ops.topk_select(boosted, k)                                              # :41-44
            scores, labels = torch.gather(scores, 1, idx), torch.gather(labels, 1, idx)       # :45-47 (selection only)
        scores = scores * inv_t
        return self._loss(y_true=labels, y_pred=scores, sample_weight=w)

    def _hard_negative_loss(self, q, c, w, cp, ci, inv_t):
        return _HardNegativeSoftmaxFn.apply(q, c, w, cp, ci, inv_t, int(self._num_hard_negatives))
