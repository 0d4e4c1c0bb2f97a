"""MI355X-native embedding-lookup + feature-interaction hot path of deep_recommenders.

Package layout mirrors the reference's import paths for the hot-path classes only:
    deep_recommenders_amd.keras.models.ranking     FM, FactorizationMachine, DeepFM, dcn.Cross
    deep_recommenders_amd.keras.models.retrieval   sbcnm.Retrieval, factorized_top_k.*
    deep_recommenders_amd.estimator.models...      fm, FM, dnn, DeepFM
    deep_recommenders_amd.feature_column           the slice of tf.feature_column those classes consume
All arithmetic runs in hand-written gfx950 kernels behind the C-ABI of include/dr_hotpath.h.
"""
import os as _os

# HIP maps streams onto 4 hardware queues by default; the sharded engines drive 5+ streams (training, routing, exchange, the later
# micro-batches' forwards, RCCL's own) and two streams that share a queue run in line: 1.97 instead of 1.67 ms per step at world 1
# (profiles/r05_sharded_step_boundary.log).  Read when the HIP runtime initialises, so it has to be in the environment before the first
# device call; a value the user set wins.
# NOTE -- an import side effect on the whole process (every HIP user in it gets 8 queues), and a no-op when HIP is already up:
if "GPU_MAX_HW_QUEUES" not in _os.environ:
    _os.environ["GPU_MAX_HW_QUEUES"] = "8"
    try:
        import sys as _sys
        _t = _sys.modules.get("torch")
        if _t is not None and _t.cuda.is_initialized():
            import warnings as _w
            _w.warn("deep_recommenders_amd: the HIP runtime was initialised before this import, so GPU_MAX_HW_QUEUES=8 cannot take effect; "
                    "export it before the first device call (the sharded engines' streams otherwise share 4 hardware queues)")
    except Exception:
        pass

__version__ = "0.1.0"
