# same exports as the reference's keras/models/retrieval/__init__.py
from deep_recommenders_amd.keras.models.retrieval.factorized_top_k import FactorizedTopK
