from deep_recommenders_amd.estimator.models.ranking.deepfm import DeepFM
