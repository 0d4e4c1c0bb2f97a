"""MI355X-native embedding-lookup + feature-interaction hot path of deep_recommenders.

Package layout mirrors the reference's import paths for the hot-path classes only:
    deep_recommenders_amd.keras.models.ranking     FM, FactorizationMachine, DeepFM, dcn.Cross
    deep_recommenders_amd.keras.models.retrieval   sbcnm.Retrieval, factorized_top_k.*
    deep_recommenders_amd.estimator.models...      fm, FM, dnn, DeepFM
    deep_recommenders_amd.feature_column           the slice of tf.feature_column those classes consume
All arithmetic runs in hand-written gfx950 kernels behind the C-ABI of include/dr_hotpath.h.
"""
__version__ = "0.1.0"
