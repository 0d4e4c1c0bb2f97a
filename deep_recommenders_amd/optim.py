"""The optimizers the reference's example scripts construct, as torch.optim.Optimizer classes over the models of this package
(dense-gradient mode), each step one HIP kernel launch per parameter (dr_adam_step / dr_ftrl_step):

  Adam  tf.train.AdamOptimizer(0.01)  (examples/train_fm_on_movielens_estimator.py:51-52, ..._deepfm_...:52-53, ..._wdl_...:72)
        / tf.keras.optimizers.Adam()  (examples/train_deepfm_on_movielens_keras.py:44): [TF] B15 formula, epsilon OUTSIDE the
        square root and un-corrected ("epsilon hat"), lr_t = lr sqrt(1-b2^t)/(1-b1^t).  Applied to the DENSE gradient of an
        embedding slab this is TF's non-lazy sparse behaviour exactly: every row's moments decay every step.
  Ftrl  tf.train.FtrlOptimizer(0.01, l1_regularization_strength=0.5)  (examples/train_wdl_on_movielens_estimator.py:66-70).

For 10 M-row tables use the engine's fused row-wise Adam (DeepFMEngine(optimizer="adam")) instead."""
import torch

from . import ops


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
        """Defaults are tf.keras.optimizers.Adam's; tf.train.AdamOptimizer: Adam(params, lr, epsilon=1e-8)."""
        super().__init__(params, dict(lr=lr, beta_1=beta_1, beta_2=beta_2, epsilon=epsilon))

    @torch.no_grad()
    def step(self, closure=None):
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["t"] = 0
                    st["m"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["v"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["t"] += 1
                lr_t = ops.adam_lr_t(group["lr"], group["beta_1"], group["beta_2"], st["t"])
                ops.adam_step(p.data, p.grad.contiguous(), st["m"], st["v"], lr_t, group["beta_1"], group["beta_2"],
                              group["epsilon"])


class Ftrl(torch.optim.Optimizer):
    def __init__(self, params, lr, learning_rate_power=-0.5, initial_accumulator_value=0.1, l1_regularization_strength=0.0,
                 l2_regularization_strength=0.0):
        super().__init__(params, dict(lr=lr, lr_power=learning_rate_power, init_accum=initial_accumulator_value,
                                      l1=l1_regularization_strength, l2=l2_regularization_strength))

    @torch.no_grad()
    def step(self, closure=None):
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["accum"] = torch.full_like(p, group["init_accum"], memory_format=torch.contiguous_format)
                    st["linear"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                ops.ftrl_step(p.data, p.grad.contiguous(), st["accum"], st["linear"], group["lr"], group["lr_power"],
                              group["l1"], group["l2"])
