# same exports as the reference's estimator/models/feature_interaction/__init__.py
from deep_recommenders_amd.estimator.models.feature_interaction.fm import fm
from deep_recommenders_amd.estimator.models.feature_interaction.fm import FM
from deep_recommenders_amd.estimator.models.feature_interaction.dnn import dnn
