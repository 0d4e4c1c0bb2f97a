# same exports as the reference's keras/models/ranking/__init__.py:4-6
from deep_recommenders_amd.keras.models.ranking.fm import FM
from deep_recommenders_amd.keras.models.ranking.fm import FactorizationMachine
from deep_recommenders_amd.keras.models.ranking.deepfm import DeepFM
