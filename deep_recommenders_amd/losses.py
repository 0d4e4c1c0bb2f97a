"""Sigmoid + the three binary losses the reference's examples use, as autograd functions over the K11
kernels.  Reference call sites (reference root): examples/train_fm_on_movielens_estimator.py:46
(tf.losses.sigmoid_cross_entropy), examples/train_deepfm_on_movielens_estimator.py:47 (tf.losses.log_loss),
examples/train_deepfm_on_movielens_keras.py:43 (tf.keras.losses.binary_crossentropy)."""
import torch

from . import ops


class _SigmoidFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y = ops.sigmoid_fwd(x)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return ops.sigmoid_bwd(y, dy.contiguous())


class _LogitLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, mode):
        loss, _, g = ops.bce_fwd_bwd(logits, labels, mode, want_prob=False, want_grad=True)
        ctx.save_for_backward(g)
        ctx.shape = logits.shape
        return loss.reshape(())

    @staticmethod
    def backward(ctx, d_loss):
        (g,) = ctx.saved_tensors
        return (g * d_loss).reshape(ctx.shape), None, None


class _ProbLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, prob, labels, mode):
        loss, g = ops.bce_prob_fwd_bwd(prob, labels, mode)
        ctx.save_for_backward(g)
        ctx.shape = prob.shape
        return loss.reshape(())

    @staticmethod
    def backward(ctx, d_loss):
        (g,) = ctx.saved_tensors
        return (g * d_loss).reshape(ctx.shape), None, None


def sigmoid(x: torch.Tensor) -> torch.Tensor:
    return _SigmoidFn.apply(x.contiguous())


def sigmoid_cross_entropy(labels, logits):
    """tf.losses.sigmoid_cross_entropy(labels, logits): mean over all elements ([TF] B9)."""
    return _LogitLossFn.apply(logits.contiguous(), labels.to(torch.float32).contiguous(), ops.LOSS_SIGMOID_CE)


def log_loss(labels, predictions):
    """tf.losses.log_loss(labels, predictions) on probabilities, eps = 1e-7 ([TF] B10)."""
    return _ProbLossFn.apply(predictions.contiguous(), labels.to(torch.float32).contiguous(), ops.LOSS_LOG_LOSS)


def binary_crossentropy(y_true, y_pred):
    """tf.keras.losses.binary_crossentropy on probabilities, mean over the batch ([TF] B11)."""
    return _ProbLossFn.apply(y_pred.contiguous(), y_true.to(torch.float32).contiguous(), ops.LOSS_KERAS_BCE)
